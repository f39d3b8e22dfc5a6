"""Random n-gram language models (ARPA) and word-spelling logits shared by the CTC decoder tests."""
import numpy as np


def _random_lm(tmp_path, rng, alphabet, order=3, n_words=14):
  letters = [c for c in alphabet if c != " "]
  words = set()
  while len(words) < n_words:
    words.add("".join(rng.choice(letters, size=rng.integers(1, 4))))
  words = sorted(words)
  vocab = ["<unk>", "<s>", "</s>"] + words
  grams = [dict() for _ in range(order)]
  for w in vocab:
    grams[0][(w,)] = (-float(rng.uniform(0.5, 3.0)), -float(rng.uniform(0.0, 1.0)))
  ctx_words = ["<s>"] + words
  for n in range(2, order + 1):
    prev = [g for g in grams[n - 2] if g[-1] != "</s>" and g[0] != "</s>"]
    for g in prev:
      if g[0] == "<unk>":
        continue
      for w in words + ["</s>"]:
        if rng.uniform() < (0.45 if n == 2 else 0.25):
          # an n-gram needs its suffix (n-1)-gram as well as its prefix to be well formed
          if n > 2 and g[1:] + (w,) not in grams[n - 2]:
            continue
          grams[n - 1][g + (w,)] = (-float(rng.uniform(0.1, 2.0)),
                                    -float(rng.uniform(0.0, 0.8)) if n < order else 0.0)
  path = str(tmp_path / "lm.arpa")
  with open(path, "w") as f:
    f.write("\\data\\\n" + "".join("ngram %d=%d\n" % (n + 1, len(grams[n])) for n in range(order)))
    for n in range(order):
      f.write("\n\\%d-grams:\n" % (n + 1))
      for g, (p, b) in grams[n].items():
        f.write("%.7f\t%s" % (p, " ".join(g)) + ("\t%.7f\n" % b if n + 1 < order else "\n"))
    f.write("\n\\end\\\n")
  vocab_path = str(tmp_path / "vocab.txt")
  with open(vocab_path, "w") as f:
    f.write(" ".join(words) + "\n" + " ".join(words[:5]) + "\n")
  return path, vocab_path, words


def _peaky_logits(rng, T, B, C, words, alphabet):
  """Frames that mostly spell vocabulary words, with confusable runner-ups."""
  lab = {c: i for i, c in enumerate(alphabet)}
  out = rng.normal(0, 1.0, size=(T, B, C)).astype(np.float32)
  for b in range(B):
    t = 0
    while t < T:
      w = words[rng.integers(len(words))] + " "
      for ch in w:
        for _ in range(rng.integers(1, 3)):
          if t < T:
            out[t, b, lab[ch]] += rng.uniform(1.0, 4.0)
            t += 1
        if t < T and rng.uniform() < 0.5:
          out[t, b, C - 1] += rng.uniform(1.0, 4.0)
          t += 1
  return out

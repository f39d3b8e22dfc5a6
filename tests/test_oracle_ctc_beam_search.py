"""Pins oracle/ctc_beam_search.py to the reference's own known answers
(ctc_decoder_with_lm/ctc-test.py:29-124; fixtures made by tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest

from oracle import ctc_beam_search as cb

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def kat():
  with open(os.path.join(GOLD, "ctc_test_meta.json")) as f:
    meta = json.load(f)
  seq = np.load(os.path.join(GOLD, "ctc_test_logits.npy"))[:, 0, :]
  return meta, seq


def _text(meta, ids):
  return "".join(meta["vocab"][c] for c in ids)


def _lm_scorer(meta, alpha, beta, trie_weight):
  lm = cb.load_lm(os.path.join(GOLD, "ctc_test_lm.binary"))
  trie = cb.read_letter_trie(os.path.join(GOLD, "ctc_test_lm.trie"), len(meta["vocab"]))
  return cb.WordLMScorer(lm, trie, meta["vocab"], alpha, beta, trie_weight)


def test_beam_search_without_scorer_matches_tf_known_answer(kat):
  # ctc-test.py:42-45,69-74: tf.nn.ctc_beam_search_decoder(beam_width=16, merge_repeated=False)
  meta, seq = kat
  paths, lp = cb.ctc_beam_search(seq, meta["beam_width"])
  assert _text(meta, paths[0]) == meta["beam_text"]
  assert abs(lp[0] - meta["beam_log_prob"]) < meta["tol"]


def test_beam_search_with_language_model_known_answer(kat):
  # ctc-test.py:47-59,76-78: the custom op with alpha=2.0, beta=0.5, trie_weight=0.1
  meta, seq = kat
  sc = _lm_scorer(meta, meta["lm_alpha"], meta["lm_beta"], meta["lm_trie_weight"])
  paths, lp = cb.ctc_beam_search(seq, meta["beam_width"], sc)
  assert _text(meta, paths[0]) == meta["lm_text"] == meta["label"]
  assert abs(lp[0] - meta["lm_log_prob"]) < meta["tol"]


def test_zero_weights_equal_plain_beam_search(kat):
  # ctc-test.py:81-124: random logits, alpha = beta = trie_weight = 0
  meta, seq = kat
  np.random.seed(1234)
  logits = np.random.uniform(size=seq.shape).astype(np.float32)
  p1, lp1 = cb.ctc_beam_search(logits, meta["beam_width"])
  p2, lp2 = cb.ctc_beam_search(logits, meta["beam_width"], _lm_scorer(meta, 0.0, 0.0, 0.0))
  assert p1[0] == p2[0]
  assert abs(lp1[0] - lp2[0]) < meta["tol"] and lp2[0] < 0


def test_kenlm_sample_model_contents(kat):
  # the sample is a bigram model over "<s> ten seconds </s>" (hash-ranked word ids)
  lm = cb.load_lm(os.path.join(GOLD, "ctc_test_lm.binary"))
  assert lm.order == 2 and sorted(lm.vocab) == sorted(["<unk>", "<s>", "ten", "seconds", "</s>"])
  ids = {w: i for i, w in enumerate(lm.vocab)}
  assert set(lm.ngrams[1]) == {(ids["<s>"], ids["ten"]), (ids["ten"], ids["seconds"]),
                               (ids["seconds"], ids["</s>"])}
  assert abs(lm.ngrams[0][(ids["<unk>"],)][0] - np.log10(1 / 8.0)) < 1e-5
  # back-off: P(seconds | seconds) = backoff(seconds) + P(seconds)
  p = lm.score([ids["seconds"]], ids["seconds"])
  assert abs(p - (lm.ngrams[0][(ids["seconds"],)][1] + lm.ngrams[0][(ids["seconds"],)][0])) < 1e-6


def test_arpa_reader_agrees_with_binary_reader(tmp_path, kat):
  lm = cb.load_lm(os.path.join(GOLD, "ctc_test_lm.binary"))
  path = str(tmp_path / "lm.arpa")
  with open(path, "w") as f:
    f.write("\\data\\\nngram 1=%d\nngram 2=%d\n\n\\1-grams:\n" % (len(lm.ngrams[0]), len(lm.ngrams[1])))
    for (w,), (p, b) in lm.ngrams[0].items():
      f.write("%.9g\t%s\t%.9g\n" % (p, lm.vocab[w], b))
    f.write("\n\\2-grams:\n")
    for (a, w), (p, _) in lm.ngrams[1].items():
      f.write("%.9g\t%s %s\n" % (p, lm.vocab[a], lm.vocab[w]))
    f.write("\n\\end\\\n")
  lm2 = cb.load_lm(path)
  for hist in ([], ["<s>"], ["ten"], ["seconds"], ["</s>"]):
    for w in lm.vocab[1:]:
      a = lm.score([lm.index(h) for h in hist], lm.index(w))
      b = lm2.score([lm2.index(h) for h in hist], lm2.index(w))
      assert abs(a - b) < 1e-6, (hist, w)


def test_letter_trie_rebuild_matches_file(kat):
  # generate_trie.cpp:53-61 over the vocabulary "ten seconds"
  meta, _ = kat
  lm = cb.load_lm(os.path.join(GOLD, "ctc_test_lm.binary"))
  ref = cb.read_letter_trie(os.path.join(GOLD, "ctc_test_lm.trie"), len(meta["vocab"]))
  mine = cb.build_letter_trie([(lm.index(w), w) for w in ("ten", "seconds")], meta["vocab"],
                              lambda wid: lm.score([], wid))

  def same(a, b):
    assert (a is None) == (b is None)
    if a is None:
      return
    assert a.prefix_count == b.prefix_count and a.min_score_word == b.min_score_word
    assert abs(a.min_unigram_score - b.min_unigram_score) < 1e-5
    assert set(a.children) == set(b.children)
    for k in a.children:
      same(a.children[k], b.children[k])
  same(ref, mine)

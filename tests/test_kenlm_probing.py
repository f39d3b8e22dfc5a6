"""KenLM binaries in the probing layout (kenlm's default; the `ctc_decoders` scorer of the
reference loads any KenLM type): reader validated on the reference's own sample
open_seq2seq/test_utils/toy_speech_data/toy_data-lm.binary (trigram, 91 words) — structural
invariants of the file, probabilities that sum to one in every context, product == oracle."""
import os
import sys

import numpy as np
import pytest

from oracle import ctc_beam_search as cb
from oracle import ctc_decoders as cd
from _ctc_helpers import _peaky_logits

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LM = os.path.join(REPO, "tests", "golden", "toy_data_lm.binary")


@pytest.fixture(scope="module")
def lm():
  return cb.load_lm(LM)


def test_file_invariants(lm):
  assert lm.order == 3 and len(lm.vocab) == 91 and lm.vocab[:3] == ["<unk>", "<s>", "</s>"]
  assert len(lm.hashed[1]) == 115 and len(lm.hashed[2]) == 108        # the header's n-gram counts
  # every stored bigram / trigram key is the hash chain of some word-id sequence
  V = len(lm.vocab)
  bigrams = [(a, b) for b in range(V) for a in range(V)
             if cb.combine_word_hash(b, a) in lm.hashed[1]]
  assert len(bigrams) == 115
  tri = sum(1 for (b, c) in bigrams for a in range(V)
            if cb.combine_word_hash(cb.combine_word_hash(c, b), a) in lm.hashed[2])
  assert tri == 108
  # the training text of the toy corpus starts with "there was no autopsy period"
  ids = [lm.index(w) for w in ["<s>", "there", "was", "no", "autopsy", "period", "</s>"]]
  assert 0 not in ids
  for i in range(2, len(ids)):
    assert (cb.combine_word_hash(cb.combine_word_hash(ids[i], ids[i - 1]), ids[i - 2]) in lm.hashed[2])


def test_every_context_is_a_probability_distribution(lm):
  V = len(lm.vocab)
  rng = np.random.default_rng(0)
  contexts = [[], [lm.bos], [lm.index("there")], [lm.bos, lm.index("there")],
              [lm.index("there"), lm.index("was")], [lm.index("the"), lm.index("new")]]
  contexts += [list(rng.integers(2, V, size=2)) for _ in range(10)]
  for h in contexts:
    total = sum(10.0 ** lm.score(h, w) for w in range(V) if w != lm.bos)
    assert abs(total - 1.0) < 1e-5, (h, total)


def test_product_reader_matches_oracle(tmp_path, lm):
  from openseq2seq_amd import capi
  alphabet = [" "] + [chr(ord("a") + i) for i in range(26)] + ["'"]
  alpha_path = str(tmp_path / "alphabet.txt")
  with open(alpha_path, "w") as f:
    f.write("\n".join(alphabet) + "\n")
  words = [w for w in lm.vocab[3:] if all(c in alphabet for c in w)]
  vocab_path = str(tmp_path / "words.txt")
  with open(vocab_path, "w") as f:
    f.write(" ".join(words) + "\n")
  trie_path = str(tmp_path / "lm.trie")
  capi.ctc_generate_trie(alpha_path, LM, vocab_path, trie_path)
  sc = capi.CtcScorer(LM, trie_path, alpha_path, 1.0, 0.0, 0.0)
  osc = cb.WordLMScorer(lm, None, alphabet, 1.0, 0.0, 0.0)
  rng = np.random.default_rng(1)
  seqs = [["there"], ["there", "was"], ["there", "was", "no"], ["was", "no", "autopsy", "period"],
          ["the", "new", "york"], ["zzz"], ["there", "zzz", "was"]]
  seqs += [[words[i] for i in rng.integers(0, len(words), size=rng.integers(1, 6))] for _ in range(30)]
  for ws in seqs:
    assert abs(sc.ngram_score(ws) - osc.score_ngram(tuple(ws))) < 1e-5, ws


def test_ctc_decoders_scorer_on_a_probing_model(lm):
  sys.path.insert(0, os.path.join(REPO, "decoders"))
  from ctc_decoders import Scorer, ctc_beam_search_decoder
  alphabet = [" "] + [chr(ord("a") + i) for i in range(26)] + ["'"]
  scorer = Scorer(1.2, 0.4, LM, alphabet)
  oscorer = cd.Scorer(1.2, 0.4, lm, alphabet)
  assert scorer.get_max_order() == 3 and not scorer.is_character_based()
  assert scorer.get_dict_size() == oscorer.dictionary.size
  words = ["there", "was", "no", "autopsy", "period", "the", "new", "york"]
  rng = np.random.default_rng(3)
  logits = _peaky_logits(rng, 60, 1, len(alphabet) + 1, words, alphabet)[:, 0].astype(np.float64)
  e = np.exp(logits - logits.max(-1, keepdims=True))
  probs = e / e.sum(-1, keepdims=True)
  got = ctc_beam_search_decoder(probs.astype(np.float32), alphabet, 24, ext_scoring_func=scorer)
  ref = cd.ctc_beam_search_decoder(probs, alphabet, 24, ext_scorer=oscorer)
  assert got[0][1] == ref[0][1] and abs(got[0][0] - ref[0][0]) < 2e-3 * max(1.0, abs(ref[0][0]))
  assert any(w in got[0][1].split() for w in words)

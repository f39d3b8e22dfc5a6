"""The drop-in `ctc_decoders` module (openseq2seq_amd/ctc_decoders.py on the host entry points
os2s_ctc_dict_*) against oracle/ctc_decoders.py and the reference's known answer
(scripts/ctc_decoders_test.py). Host code: runs without a GPU."""
import json
import os
import sys

import numpy as np
import pytest

from oracle import ctc_beam_search as cb
from oracle import ctc_decoders as cd
from _ctc_helpers import _peaky_logits, _random_lm

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, os.path.join(REPO, "decoders"))       # as the reference's scripts do
from ctc_decoders import (Scorer, ctc_beam_search_decoder, ctc_beam_search_decoder_batch,  # noqa: E402
                          ctc_greedy_decoder)


def _softmax(x):
  m = x.max(-1, keepdims=True)
  e = np.exp(x - m)
  return e / e.sum(-1, keepdims=True)


def test_reference_test_script_known_answers():
  # scripts/ctc_decoders_test.py:36-80 verbatim in its decoder calls
  with open(os.path.join(GOLD, "ctc_test_meta.json")) as f:
    meta = json.load(f)
  seq = np.load(os.path.join(GOLD, "ctc_test_logits.npy"))
  vocab = meta["vocab"] + ["_"]
  scorer = Scorer(alpha=2.0, beta=0.5, model_path=os.path.join(GOLD, "ctc_test_lm.binary"),
                  vocabulary=vocab[:-1])
  res = ctc_beam_search_decoder(_softmax(seq.squeeze()), vocab[:-1], beam_size=16, ext_scoring_func=scorer)
  res_prob, decoded_text = res[0]
  assert abs(4.0845 + res_prob) < 1e-3
  assert decoded_text == meta["label"]
  assert len(res) == 16
  assert not scorer.is_character_based() and scorer.get_max_order() == 2 and scorer.get_dict_size() == 2
  assert ctc_greedy_decoder(_softmax(seq.squeeze()), vocab[:-1]) == meta["greedy_text"]
  plain = ctc_beam_search_decoder(_softmax(seq.squeeze()), vocab[:-1], beam_size=16)
  assert plain[0][1] == meta["beam_text"] and abs(plain[0][0] - meta["beam_log_prob"]) < 1e-3
  # reset_params: no LM weight, no word bonus -> only the dictionary constraint is left
  scorer.reset_params(0.0, 0.0)
  res0 = ctc_beam_search_decoder(_softmax(seq.squeeze()), vocab[:-1], beam_size=16, ext_scoring_func=scorer)
  assert res0[0][1].strip() in ("ten seconds", "ten")


@pytest.mark.parametrize("order,beam,cutoff_prob,top_n", [(3, 8, 1.0, 40), (2, 16, 0.99, 4), (4, 12, 1.0, 3)])
def test_random_word_models_match_oracle(tmp_path, order, beam, cutoff_prob, top_n):
  rng = np.random.default_rng(7 * order + beam)
  alphabet = [" ", "a", "b", "c", "d", "e", "'"]
  lm_path, _, words = _random_lm(tmp_path, rng, alphabet, order=order)
  lm = cb.load_lm(lm_path)
  oscorer = cd.Scorer(1.3, 0.7, lm, alphabet)
  scorer = Scorer(1.3, 0.7, lm_path, alphabet)
  assert scorer.get_dict_size() == oscorer.dictionary.size == len(words)
  T, B, C = 36, 4, len(alphabet) + 1
  probs = _softmax(_peaky_logits(rng, T, B, C, words, alphabet).astype(np.float64)).astype(np.float32)
  split = [probs[:T, 0], probs[:29, 1], probs[:2, 2], probs[:T - 1, 3]]
  res = ctc_beam_search_decoder_batch(split, alphabet, beam, 3, cutoff_prob=cutoff_prob, cutoff_top_n=top_n,
                                      ext_scoring_func=scorer)
  scored_words = 0
  for b, p in enumerate(split):
    ref = cd.ctc_beam_search_decoder(p.astype(np.float64), alphabet, beam, cutoff_prob, top_n, oscorer)
    got = res[b]
    n = min(len(ref), len(got), 3)
    for k in range(n):
      assert abs(ref[k][0] - got[k][0]) <= 2e-3 * max(1.0, abs(ref[k][0])), (b, k, ref[:3], got[:3])
      near_tie = (k + 1 < len(ref) and abs(ref[k][0] - ref[k + 1][0]) < 1e-3) or \
          (k > 0 and abs(ref[k][0] - ref[k - 1][0]) < 1e-3)
      if not near_tie:
        assert ref[k][1] == got[k][1], (b, k, ref[:3], got[:3])
    scored_words += ref[0][1].count(" ")
  assert scored_words > 0


def test_character_based_model(tmp_path):
  """All LM 'words' are single characters -> the scorer is character based: no dictionary, the
  LM is consulted on every new character (scorer.cpp:63-69, ctc_beam_search_decoder.cpp:112-120)."""
  alphabet = [" ", "a", "b", "c"]
  arpa = str(tmp_path / "char.arpa")
  with open(arpa, "w") as f:
    f.write("\\data\\\nngram 1=6\nngram 2=4\n\n\\1-grams:\n-1.0\t<unk>\n-99\t<s>\t-0.3\n-0.6\t</s>\n"
            "-0.5\ta\t-0.2\n-0.7\tb\t-0.25\n-0.9\tc\t-0.1\n\n\\2-grams:\n-0.2\t<s> a\n-0.3\ta b\n"
            "-0.4\tb c\n-0.6\tc a\n\n\\end\\\n")
  rng = np.random.default_rng(5)
  probs = _softmax(rng.normal(0, 1.5, size=(20, 5))).astype(np.float32)
  scorer = Scorer(0.8, 0.3, arpa, alphabet)
  oscorer = cd.Scorer(0.8, 0.3, cb.load_lm(arpa), alphabet)
  assert scorer.is_character_based() and oscorer.is_character_based
  got = ctc_beam_search_decoder(probs, alphabet, 10, ext_scoring_func=scorer)
  ref = cd.ctc_beam_search_decoder(probs.astype(np.float64), alphabet, 10, ext_scorer=oscorer)
  assert got[0][1] == ref[0][1] and abs(got[0][0] - ref[0][0]) < 2e-3 * max(1.0, abs(ref[0][0]))


def test_argument_checks(tmp_path):
  probs = np.full((4, 5), 0.2, dtype=np.float32)
  with pytest.raises(ValueError):
    ctc_beam_search_decoder(probs, ["a", "b"], 4)                      # vocabulary / classes mismatch
  with pytest.raises(ValueError):
    ctc_beam_search_decoder_batch([probs], ["a", "b", "c", "d"], 4, 0)  # num_processes
  from openseq2seq_amd import _lib
  with pytest.raises(_lib.Os2sError):
    Scorer(1.0, 0.0, str(tmp_path / "missing.arpa"), ["a"])

"""Pins oracle/ctc_decoders.py (the reference's second CTC decoder, decoders/*.cpp) to the known
answer of the reference's own test scripts/ctc_decoders_test.py:73-80."""
import json
import os

import numpy as np

from oracle import ctc_beam_search as cb
from oracle import ctc_decoders as cd

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _softmax(x):
  m = x.max(-1, keepdims=True)
  e = np.exp(x - m)
  return e / e.sum(-1, keepdims=True)


def _kat():
  with open(os.path.join(GOLD, "ctc_test_meta.json")) as f:
    meta = json.load(f)
  seq = np.load(os.path.join(GOLD, "ctc_test_logits.npy"))[:, 0, :].astype(np.float64)
  return meta, _softmax(seq)


def test_scorer_known_answer():
  meta, probs = _kat()
  lm = cb.load_lm(os.path.join(GOLD, "ctc_test_lm.binary"))
  scorer = cd.Scorer(2.0, 0.5, lm, meta["vocab"])
  assert not scorer.is_character_based and scorer.max_order == 2 and scorer.dictionary.size == 2
  res = cd.ctc_beam_search_decoder(probs, meta["vocab"], 16, ext_scorer=scorer)
  score, text = res[0]
  assert text == meta["label"] == "ten seconds"
  assert abs(4.0845 + score) < 1e-3            # scripts/ctc_decoders_test.py:79
  assert [s for s, _ in res] == sorted((s for s, _ in res), reverse=True)


def test_without_scorer_equals_plain_prefix_search():
  # no scorer: the plain CTC prefix search — same best path / score as tf.nn.ctc_beam_search_decoder
  meta, probs = _kat()
  res = cd.ctc_beam_search_decoder(probs, meta["vocab"], 16)
  assert res[0][1] == meta["beam_text"] and abs(res[0][0] - meta["beam_log_prob"]) < 1e-3


def test_make_ngram_and_cond_prob():
  meta, _ = _kat()
  lm = cb.load_lm(os.path.join(GOLD, "ctc_test_lm.binary"))
  sc = cd.Scorer(1.0, 0.0, lm, meta["vocab"])
  lab = {c: i for i, c in enumerate(meta["vocab"])}
  root = cd.PathTrie()
  node = root
  for ch in "ten seconds":
    k = cd.PathTrie()
    k.character, k.parent = lab[ch], node
    node = k
  assert sc.make_ngram(node) == ["ten", "seconds"]
  assert sc.make_ngram(node.parent.parent.parent.parent.parent.parent.parent.parent) == ["<s>", "ten"]
  assert abs(sc.get_log_cond_prob(["<s>", "ten"]) + 0.1898795) < 1e-6
  assert sc.get_log_cond_prob(["ten", "foo"]) == cd.OOV_SCORE
  # the dictionary refuses a letter that continues no word, and re-arms after a finished word
  d = sc.dictionary
  s = 0
  for ch in "ten ":
    s = d.children[s][lab[ch]]
  assert d.final[s] and not d.children[s] and lab["x"] not in d.children[0]

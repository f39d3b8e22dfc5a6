"""The host entry points os2s_ctc_* (CTC prefix beam search + n-gram language model) against
oracle/ctc_beam_search.py and the reference's known answers (ctc_decoder_with_lm/ctc-test.py).
They are host code of the C-ABI library (the reference's op is CPU-only too), so these run
without a GPU."""
import json
import os

import numpy as np
import pytest
import torch

from openseq2seq_amd import _lib, capi
from oracle import ctc_beam_search as cb

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def kat(tmp_path_factory):
  with open(os.path.join(GOLD, "ctc_test_meta.json")) as f:
    meta = json.load(f)
  seq = torch.from_numpy(np.load(os.path.join(GOLD, "ctc_test_logits.npy")))
  alpha_path = str(tmp_path_factory.mktemp("kat") / "alphabet.txt")
  with open(alpha_path, "w") as f:
    f.write("\n".join(meta["vocab"]) + "\n")
  return meta, seq, alpha_path


def _text(meta, ids, n):
  return "".join(meta["vocab"][c] for c in ids[:n].tolist())


def test_known_answers(kat):
  meta, seq, alpha_path = kat
  T = seq.shape[0]
  sl = torch.tensor([T], dtype=torch.int32)
  ids, lens, lp = capi.ctc_beam_search(seq, sl, meta["beam_width"])
  assert _text(meta, ids[0, 0], lens[0, 0]) == meta["beam_text"]
  assert abs(float(lp[0, 0]) - meta["beam_log_prob"]) < meta["tol"]
  assert (ids[0, 0, lens[0, 0]:] == -1).all()
  sc = capi.CtcScorer(os.path.join(GOLD, "ctc_test_lm.binary"), os.path.join(GOLD, "ctc_test_lm.trie"),
                      alpha_path, meta["lm_alpha"], meta["lm_beta"], meta["lm_trie_weight"])
  ids, lens, lp = capi.ctc_beam_search(seq, sl, meta["beam_width"], sc)
  assert _text(meta, ids[0, 0], lens[0, 0]) == meta["lm_text"]
  assert abs(float(lp[0, 0]) - meta["lm_log_prob"]) < meta["tol"]
  # ctc-test.py:81-124
  np.random.seed(1234)
  rnd = torch.from_numpy(np.random.uniform(size=tuple(seq.shape)).astype(np.float32))
  zero = capi.CtcScorer(os.path.join(GOLD, "ctc_test_lm.binary"), os.path.join(GOLD, "ctc_test_lm.trie"),
                        alpha_path, 0.0, 0.0, 0.0)
  a = capi.ctc_beam_search(rnd, sl, meta["beam_width"])
  b = capi.ctc_beam_search(rnd, sl, meta["beam_width"], zero)
  assert a[1][0, 0] == b[1][0, 0] and (a[0] == b[0]).all()
  assert abs(float(a[2][0, 0]) - float(b[2][0, 0])) < meta["tol"] and float(b[2][0, 0]) < 0


def test_ngram_score_matches_oracle(kat):
  meta, _, alpha_path = kat
  sc = capi.CtcScorer(os.path.join(GOLD, "ctc_test_lm.binary"), os.path.join(GOLD, "ctc_test_lm.trie"),
                      alpha_path, 1.0, 0.0, 0.0)
  lm = cb.load_lm(os.path.join(GOLD, "ctc_test_lm.binary"))
  osc = cb.WordLMScorer(lm, None, meta["vocab"], 1.0, 0.0, 0.0)
  for words in (["ten"], ["seconds"], ["ten", "seconds"], ["seconds", "ten"], ["ten", "ten", "seconds"],
                ["foo"], ["ten", "foo"], ["foo", "ten"], [""]):
    assert abs(sc.ngram_score(words) - osc.score_ngram(tuple(words))) < 1e-5, words


from _ctc_helpers import _peaky_logits, _random_lm  # noqa: E402


@pytest.mark.parametrize("order,beam,top,merge", [(3, 8, 1, False), (3, 24, 3, False), (2, 16, 2, True),
                                                   (4, 12, 1, False)])
def test_random_language_models_match_oracle(tmp_path, order, beam, top, merge):
  rng = np.random.default_rng(100 * order + beam)
  alphabet = [" ", "a", "b", "c", "d", "e", "'"]
  alpha_path = str(tmp_path / "alphabet.txt")
  with open(alpha_path, "w") as f:
    f.write("# comment line\n" + "\n".join(alphabet) + "\n")
  lm_path, vocab_path, words = _random_lm(tmp_path, rng, alphabet, order=order)
  trie_path = str(tmp_path / "lm.trie")
  f = _lib.bind("os2s_ctc_generate_trie", [_lib.ctypes.c_char_p] * 4, _lib.c_int)
  _lib.check(f(alpha_path.encode(), lm_path.encode(), vocab_path.encode(), trie_path.encode()), "generate_trie")
  # the written trie equals the oracle's build over the same word list (words[:5] inserted twice)
  lm = cb.load_lm(lm_path)
  ref = cb.build_letter_trie([(lm.index(w), w) for w in words + words[:5]], alphabet,
                             lambda wid: lm.score([], wid))
  got = cb.read_letter_trie(trie_path, len(alphabet))

  def same(a, b):
    assert (a is None) == (b is None)
    if a is not None:
      assert a.prefix_count == b.prefix_count
      assert abs(a.min_unigram_score - b.min_unigram_score) < 1e-4
      assert set(a.children) == set(b.children)
      for k in a.children:
        same(a.children[k], b.children[k])
  same(ref, got)

  alpha, beta, tw = 1.5, 0.8, 0.2
  sc = capi.CtcScorer(lm_path, trie_path, alpha_path, alpha, beta, tw)
  osc = cb.WordLMScorer(lm, got, alphabet, alpha, beta, tw)
  T, B, C = 40, 5, len(alphabet) + 1
  logits = _peaky_logits(rng, T, B, C, words, alphabet)
  seq_len = np.array([T, T - 7, 1, T - 1, 13], dtype=np.int32)
  ids, lens, lp = capi.ctc_beam_search(torch.from_numpy(logits), torch.from_numpy(seq_len), beam, sc,
                                       top_paths=top, merge_repeated=merge, n_threads=3)
  n_words_seen = 0
  for b in range(B):
    if seq_len[b] * (C - 1) + 1 < top:
      continue
    paths, olp = cb.ctc_beam_search(logits[:seq_len[b], b], beam, osc, top_paths=top, merge_repeated=merge)
    for k in range(top):
      # float (product) vs double (oracle): a near-tie between two beams may swap them
      if abs(olp[k] - float(lp[b, k])) > 2e-3 * max(1.0, abs(olp[k])):
        raise AssertionError((b, k, olp, lp[b]))
      if k + 1 < top and abs(olp[k] - olp[k + 1]) < 1e-3 or k > 0 and abs(olp[k] - olp[k - 1]) < 1e-3:
        continue
      assert ids[b, k, :lens[b, k]].tolist() == paths[k], (b, k)
      n_words_seen += "".join(alphabet[c] for c in paths[k]).count(" ")
  assert n_words_seen > 0      # the language model was actually consulted


def test_ragged_and_degenerate_inputs():
  rng = np.random.default_rng(3)
  logits = torch.from_numpy(rng.normal(size=(6, 3, 5)).astype(np.float32))
  sl = torch.tensor([6, 0, 2], dtype=torch.int32)
  ids, lens, lp = capi.ctc_beam_search(logits, sl, 4)
  assert int(lens[1, 0]) == 0 and float(lp[1, 0]) == 0.0 and (ids[1] == -1).all()
  for b in (0, 2):
    paths, olp = cb.ctc_beam_search(logits[:int(sl[b]), b].numpy(), 4)
    assert ids[b, 0, :lens[b, 0]].tolist() == paths[0] and abs(float(lp[b, 0]) - olp[0]) < 1e-4
  with pytest.raises(_lib.Os2sError):          # more paths than leaves exist after 0 frames
    capi.ctc_beam_search(logits, sl, 4, top_paths=2)
  with pytest.raises(_lib.Os2sError):          # top_paths > beam_width (beam_search.cc:405-407)
    capi.ctc_beam_search(logits, sl, 2, top_paths=3)
  with pytest.raises(_lib.Os2sError):
    capi.ctc_beam_search(logits, torch.tensor([7, 1, 1], dtype=torch.int32), 4)


def test_scorer_errors(tmp_path, kat):
  meta, _, alpha_path = kat
  trie = os.path.join(GOLD, "ctc_test_lm.trie")
  with pytest.raises(_lib.Os2sError):
    capi.CtcScorer(str(tmp_path / "missing.arpa"), trie, alpha_path, 1.0, 0.0)
  # a KenLM binary of another model type / order is refused, not misread
  with open(os.path.join(GOLD, "ctc_test_lm.binary"), "rb") as f:
    d = bytearray(f.read())
  d[0x60] = 2          # model type: unquantised trie — a layout the reference ships no sample of
  other = str(tmp_path / "trie.binary")
  with open(other, "wb") as f:
    f.write(bytes(d))
  with pytest.raises(_lib.Os2sError, match="(?i)unsupported"):
    capi.CtcScorer(other, trie, alpha_path, 1.0, 0.0)
  d[0x60] = 0          # claims to be a probing model but is not laid out as one: refused as well
  with open(other, "wb") as f:
    f.write(bytes(d))
  with pytest.raises(_lib.Os2sError):
    capi.CtcScorer(other, trie, alpha_path, 1.0, 0.0)
  # a hostile ARPA header (absurd n-gram counts) comes back as a status code: nothing may be
  # allocated from it and no C++ exception may cross the C ABI
  bad = str(tmp_path / "bad.arpa")
  with open(bad, "w") as f:
    f.write("\\data\\\nngram 1=3\nngram 2=99999999999999\n\n\\1-grams:\n-1.0\t<unk>\n-1.0\ta\t-0.5\n-1.0\tb\t-0.5\n\n"
            "\\2-grams:\n-0.3\ta b\n\n\\end\\\n")
  with pytest.raises(_lib.Os2sError):
    capi.CtcScorer(bad, trie, alpha_path, 1.0, 0.0)
  # alphabet / trie size mismatch (trie_node.h:73-79)
  short = str(tmp_path / "short_alphabet.txt")
  with open(short, "w") as f:
    f.write(" \na\nb\n")
  with pytest.raises(_lib.Os2sError):
    capi.CtcScorer(os.path.join(GOLD, "ctc_test_lm.binary"), trie, short, 1.0, 0.0)


def test_raw_ctypes_binding_as_documented(kat):
  """The stub of INTEGRATION.md §3 verbatim: plain ctypes + NumPy, no torch, no capi."""
  import ctypes
  from openseq2seq_amd.build import LIB_PATH
  meta, seq, alphabet_path = kat
  lib = ctypes.CDLL(LIB_PATH)
  lib.os2s_ctc_scorer_create.argtypes = [ctypes.c_char_p] * 3 + [ctypes.c_float] * 3 + [ctypes.POINTER(ctypes.c_void_p)]
  lib.os2s_ctc_beam_search.argtypes = ([ctypes.c_void_p, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_void_p]
                                       + [ctypes.c_int] * 6 + [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 3)
  lib.os2s_ctc_scorer_destroy.argtypes = [ctypes.c_void_p]
  lib.os2s_ctc_scorer_destroy.restype = None
  scorer = ctypes.c_void_p()
  assert lib.os2s_ctc_scorer_create(os.path.join(GOLD, "ctc_test_lm.binary").encode(),
                                    os.path.join(GOLD, "ctc_test_lm.trie").encode(), alphabet_path.encode(),
                                    meta["lm_alpha"], meta["lm_beta"], meta["lm_trie_weight"],
                                    ctypes.byref(scorer)) == 0
  logits = np.ascontiguousarray(seq.numpy(), dtype=np.float32)
  T, B, C = logits.shape
  seq_len = np.array([T], dtype=np.int32)
  ids = np.empty((B, 1, T), np.int32)
  lens = np.empty((B, 1), np.int32)
  logp = np.empty((B, 1), np.float32)
  rc = lib.os2s_ctc_beam_search(logits.ctypes.data, B * C, C, seq_len.ctypes.data, T, B, C, meta["beam_width"], 1, 0,
                                scorer, 0, ids.ctypes.data, lens.ctypes.data, logp.ctypes.data)
  assert rc == 0
  assert "".join(meta["vocab"][c] for c in ids[0, 0, :lens[0, 0]]) == meta["lm_text"]
  assert abs(float(logp[0, 0]) - meta["lm_log_prob"]) < meta["tol"]
  lib.os2s_ctc_scorer_destroy(scorer)

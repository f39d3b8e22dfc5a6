"""Drop-in for the reference's `ctc_decoders` Python module (decoders/ctc_decoders.py, the swig
wrapper around decoders/*.cpp): same names and arguments —

  Scorer(alpha, beta, model_path, vocabulary)      decoders/ctc_decoders.py:9-31
  ctc_greedy_decoder(probs_seq, vocabulary)         :34-47
  ctc_beam_search_decoder(probs_seq, vocabulary, beam_size, cutoff_prob=1.0, cutoff_top_n=40,
                          ext_scoring_func=None)    :63-98
  ctc_beam_search_decoder_batch(probs_split, vocabulary, beam_size, num_processes,
                                cutoff_prob=1.0, cutoff_top_n=40, ext_scoring_func=None)  :101-141

on the host entry points os2s_ctc_dict_* of the C-ABI library (include/os2s.h). `vocabulary` is
the list of labels WITHOUT the blank (the blank is the last class of probs_seq); results are
lists of (score, text) tuples, best first, `beam_size` of them like the reference's."""
from __future__ import absolute_import, division, print_function

import ctypes

import numpy as np

from . import _lib

_c_pp = ctypes.POINTER(ctypes.c_char_p)


def _fn(name, argtypes, restype=ctypes.c_int):
  return _lib.bind(name, list(argtypes), restype)


class Scorer(object):
  """External scorer: n-gram language model + word insertion bonus + dictionary.
  alpha weighs log10 P(word | history), beta is added per word."""

  def __init__(self, alpha, beta, model_path, vocabulary):
    self._h = ctypes.c_void_p(0)
    self._destroy = _fn("os2s_ctc_dict_scorer_destroy", (ctypes.c_void_p,), None)
    self.alpha, self.beta = float(alpha), float(beta)
    self.vocabulary = list(vocabulary)
    arr = (ctypes.c_char_p * len(self.vocabulary))(*[v.encode("utf-8") for v in self.vocabulary])
    f = _fn("os2s_ctc_dict_scorer_create", (ctypes.c_char_p, _c_pp, ctypes.c_int, ctypes.c_double,
                                            ctypes.c_double, ctypes.POINTER(ctypes.c_void_p)))
    _lib.check(f(str(model_path).encode(), arr, len(self.vocabulary), self.alpha, self.beta,
                 ctypes.byref(self._h)), "os2s_ctc_dict_scorer_create(%s)" % model_path)

  @property
  def handle(self):
    return self._h

  def reset_params(self, alpha, beta):
    self.alpha, self.beta = float(alpha), float(beta)
    _lib.check(_fn("os2s_ctc_dict_scorer_set_weights", (ctypes.c_void_p, ctypes.c_double, ctypes.c_double))(
        self._h, self.alpha, self.beta), "os2s_ctc_dict_scorer_set_weights")

  def _info(self):
    a, b, c = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(_fn("os2s_ctc_dict_scorer_info", (ctypes.c_void_p,) + (ctypes.POINTER(ctypes.c_int),) * 3)(
        self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)), "os2s_ctc_dict_scorer_info")
    return bool(a.value), b.value, c.value

  def is_character_based(self):
    return self._info()[0]

  def get_max_order(self):
    return self._info()[1]

  def get_dict_size(self):
    return self._info()[2]

  def __del__(self):
    h, self._h = getattr(self, "_h", None), None
    if h and getattr(self, "_destroy", None) is not None:
      self._destroy(h)


def ctc_greedy_decoder(probs_seq, vocabulary):
  """Best path: argmax per frame, merge repeats, drop blanks (decoders/ctc_greedy_decoder.cpp)."""
  probs = np.asarray(probs_seq)
  best = probs.argmax(-1).tolist()
  blank = len(vocabulary)
  out, prev = [], -1
  for c in best:
    if c != prev and c != blank:
      out.append(vocabulary[c])
    prev = c
  return "".join(out)


def _decode(probs, lens, vocabulary, beam_size, cutoff_prob, cutoff_top_n, scorer, n_threads):
  T, B, C = probs.shape
  if C != len(vocabulary) + 1:
    raise ValueError("The shape of probs_seq does not match with the shape of the vocabulary")
  ids = np.empty((B, beam_size, T), dtype=np.int32)
  ln = np.empty((B, beam_size), dtype=np.int32)
  sc = np.empty((B, beam_size), dtype=np.float32)
  f = _fn("os2s_ctc_dict_beam_search",
          (ctypes.c_void_p, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_void_p) + (ctypes.c_int,) * 4
          + (ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int) + (ctypes.c_void_p,) * 3)
  # fewer than beam_size prefixes can exist (short utterances): ask for what is there
  top = beam_size
  while True:
    rc = f(probs.ctypes.data, B * C, C, lens.ctypes.data, T, B, C, int(beam_size), float(cutoff_prob),
           int(cutoff_top_n), int(top), scorer.handle if scorer is not None else None, int(n_threads),
           ids.ctypes.data, ln.ctypes.data, sc.ctypes.data)
    if rc == 0 or top == 1:
      break
    top = max(1, top // 2)
    ids = np.empty((B, top, T), dtype=np.int32)
    ln = np.empty((B, top), dtype=np.int32)
    sc = np.empty((B, top), dtype=np.float32)
  _lib.check(rc, "os2s_ctc_dict_beam_search")
  return [[(float(sc[b, k]), "".join(vocabulary[c] for c in ids[b, k, :ln[b, k]])) for k in range(top)]
          for b in range(B)]


def ctc_beam_search_decoder(probs_seq, vocabulary, beam_size, cutoff_prob=1.0, cutoff_top_n=40,
                            ext_scoring_func=None):
  probs = np.ascontiguousarray(np.asarray(probs_seq, dtype=np.float32)[:, None, :])
  lens = np.array([probs.shape[0]], dtype=np.int32)
  return _decode(probs, lens, list(vocabulary), beam_size, cutoff_prob, cutoff_top_n, ext_scoring_func, 1)[0]


def ctc_beam_search_decoder_batch(probs_split, vocabulary, beam_size, num_processes, cutoff_prob=1.0,
                                  cutoff_top_n=40, ext_scoring_func=None):
  if num_processes <= 0:
    raise ValueError("Number of processes must be positive!")
  seqs = [np.asarray(p, dtype=np.float32) for p in probs_split]
  T, C = max(s.shape[0] for s in seqs), seqs[0].shape[1]
  probs = np.full((T, len(seqs), C), 1.0 / C, dtype=np.float32)
  lens = np.zeros(len(seqs), dtype=np.int32)
  for b, s in enumerate(seqs):
    probs[:s.shape[0], b] = s
    lens[b] = s.shape[0]
  return _decode(probs, lens, list(vocabulary), beam_size, cutoff_prob, cutoff_top_n, ext_scoring_func,
                 num_processes)

"""ParallelTextDataLayer — parameter schema and batch contract of
open_seq2seq/data/text2text/text2text.py:36-298.

Batch contract: source_tensors = [ids int32 [B, Ls] (pad 0), lengths [B]], target_tensors
likewise; special ids PAD=0, EOS=1, S=2, UNK=3 (SpecialTextTokens, :14-20); with
pad_vocab_to_eight the vocabulary sizes are rounded up to a multiple of 8.
`iterate_batches` is the host-side reader for real line files (vocab load :96-150, token ->
id with <S> ... </S> and max_length truncation :162-182, per-worker shard, shuffle, repeat,
padded batches :184-247) without tf.data; `synthetic_batch` draws the workload SURVEY.md §8d
defines (lengths U[8, max_length], ids U{4..V-1}). Both also emit the packed index vectors
the Transformer kernels consume (parts/transformer/packing.py)."""
from __future__ import absolute_import, division, print_function

import enum
import io
import os

import numpy as np
import torch

from ..data_layer import DataLayer
from ...parts.transformer import packing


class SpecialTextTokens(enum.Enum):
  PAD_ID = 0   # special padding token
  EOS_ID = 1   # special end of sentence token
  S_ID = 2     # special start of sentence token
  UNK_ID = 3   # out of vocabulary
  OUT_OF_BUCKET = 1234567890
  END_OF_CHOICE = -100

  @staticmethod
  def to_string(s_token):
    return {0: "<PAD>", 1: "</S>", 2: "<S>", 3: "<UNK>"}.get(s_token)


class ParallelTextDataLayer(DataLayer):
  @staticmethod
  def get_required_params():
    return dict(DataLayer.get_required_params(), **{
        'source_file': str, 'src_vocab_file': None, 'tgt_vocab_file': None,
        'max_length': int, 'shuffle': bool, 'repeat': bool,
    })

  @staticmethod
  def get_optional_params():
    return dict(DataLayer.get_optional_params(), **{
        'use_targets': bool, 'delimiter': str, 'target_file': str, 'map_parallel_calls': int,
        'prefetch_buffer_size': int, 'pad_lengths_to_eight': bool, 'pad_vocab_to_eight': bool,
        'shuffle_buffer_size': int, 'special_tokens_already_in_vocab': bool,
        'use_start_token': bool, 'synthetic_vocab_size': int,
    })

  def __init__(self, params, model, num_workers=1, worker_id=0):
    super(ParallelTextDataLayer, self).__init__(params, model, num_workers, worker_id)
    p = self.params
    self.max_len = p['max_length']

    def vocab_size(path):
      if path:
        with open(path, "r", encoding="utf-8") as f:
          n = sum(1 for line in f if line.strip("\n"))
        if not p.get('special_tokens_already_in_vocab', True):
          n += SpecialTextTokens.UNK_ID.value + 1
        return n
      return p.get('synthetic_vocab_size', 32768)

    sv, tv = vocab_size(p['src_vocab_file']), vocab_size(p['tgt_vocab_file'])
    if p.get('pad_vocab_to_eight', False):
      sv += (8 - sv % 8) % 8
      tv += (8 - tv % 8) % 8
    p['src_vocab_size'], p['tgt_vocab_size'] = sv, tv

  def build_graph(self):
    return self

  # ---- real line files ---------------------------------------------------------------
  def _load_vocab(self, path):
    """text2text.py:96-150: one token per line (first tab-separated column); ids start after
    the 4 special tokens unless the file already contains them."""
    offset = 0 if self.params.get('special_tokens_already_in_vocab', True) else \
        SpecialTextTokens.UNK_ID.value + 1
    seq2idx, idx2seq = {}, {}
    with io.open(path, "r", encoding="utf-8") as f:
      n = 0
      for line in f:
        if not line.strip("\n"):
          continue
        tok = line.rstrip("\n").split("\t")[0]
        seq2idx[tok] = n + offset
        idx2seq[n + offset] = tok
        n += 1
    if offset:
      for tkn in (SpecialTextTokens.PAD_ID, SpecialTextTokens.EOS_ID, SpecialTextTokens.S_ID,
                  SpecialTextTokens.UNK_ID):
        idx2seq[tkn.value] = SpecialTextTokens.to_string(tkn.value)
        seq2idx[SpecialTextTokens.to_string(tkn.value)] = tkn.value
    return seq2idx, idx2seq

  def _line_to_ids(self, line, seq2idx):
    toks = line.rstrip("\n").split(self.params.get('delimiter', ' '))
    ids = [seq2idx.get(t, SpecialTextTokens.UNK_ID.value) for t in toks[:self.max_len - 2]]
    ids = ([SpecialTextTokens.S_ID.value] if self.params.get('use_start_token', True) else []) + \
        ids + [SpecialTextTokens.EOS_ID.value]
    if self.params.get('pad_lengths_to_eight', False) and len(ids) % 8:
      ids += [SpecialTextTokens.PAD_ID.value] * (8 - len(ids) % 8)
    return np.asarray(ids, np.int32)

  def has_files(self):
    p = self.params
    return bool(p.get('source_file')) and os.path.exists(p['source_file']) and \
        bool(p.get('src_vocab_file')) and os.path.exists(p['src_vocab_file'])

  def load(self):
    """Reads and tokenises this worker's shard (lines worker_id::num_workers)."""
    if getattr(self, "_examples", None) is not None:
      return self._examples
    p = self.params
    self.src_seq2idx, self.src_idx2seq = self._load_vocab(p['src_vocab_file'])
    self.tgt_seq2idx, self.tgt_idx2seq = self._load_vocab(p['tgt_vocab_file'])
    with io.open(p['source_file'], "r", encoding="utf-8") as f:
      src = f.read().splitlines()
    tgt_path = p.get('target_file') or p['source_file']
    with io.open(tgt_path, "r", encoding="utf-8") as f:
      tgt = f.read().splitlines()
    if len(src) != len(tgt):
      raise ValueError("source and target files have different numbers of lines")
    self._size = len(src)
    sel = range(self._worker_id, len(src), self._num_workers)
    self._examples = [(self._line_to_ids(src[i], self.src_seq2idx),
                       self._line_to_ids(tgt[i], self.tgt_seq2idx)) for i in sel]
    return self._examples

  def get_size_in_samples(self):
    self.load()
    return self._size

  def _collate(self, pairs, device):
    def pad(seqs):
      lens = np.asarray([len(s) for s in seqs], np.int32)
      out = np.zeros((len(seqs), int(lens.max())), np.int32)
      for i, s in enumerate(seqs):
        out[i, :len(s)] = s
      return out, lens
    src, sl = pad([a for a, _ in pairs])
    tgt, tl = pad([b for _, b in pairs])
    return {
        'source_tensors': [torch.from_numpy(src).to(device), torch.from_numpy(sl).to(device)],
        'target_tensors': [torch.from_numpy(tgt).to(device), torch.from_numpy(tl).to(device)],
        'packed_source': packing.to_device(packing.pack_ids(src, sl), device),
        'packed_target': packing.to_device(packing.pack_ids(tgt, tl, shift_right=True), device),
        'num_tokens': int(sl.sum() + tl.sum()),
        'padded_tokens': int(src.size + tgt.size),
    }

  def iterate_batches(self, device, seed=0, drop_remainder=None):
    """Yields batch dicts forever (repeat=True) or for one pass."""
    p = self.params
    ex = self.load()
    B = p['batch_size']
    if drop_remainder is None:
      drop_remainder = p.get('mode', 'train') == 'train'
    rng = np.random.RandomState(seed)
    while True:
      order = rng.permutation(len(ex)) if p.get('shuffle', False) else np.arange(len(ex))
      for i in range(0, len(order), B):
        idx = order[i:i + B]
        if len(idx) < B and drop_remainder:
          break
        yield self._collate([ex[j] for j in idx], device)
      if not p.get('repeat', False):
        return

  @property
  def input_tensors(self):
    return {}

  def synthetic_batch(self, device, seed, fixed_len=None, min_len=8):
    p = self.params
    B = p['batch_size']
    rng = np.random.RandomState(seed)

    def draw(V):
      lens = (np.full(B, fixed_len) if fixed_len else
              rng.randint(min_len, self.max_len + 1, size=B)).astype(np.int32)
      L = int(lens.max())
      ids = np.zeros((B, L), np.int32)
      for b in range(B):
        ids[b, :lens[b] - 1] = rng.randint(4, V, size=lens[b] - 1)
        ids[b, lens[b] - 1] = SpecialTextTokens.EOS_ID.value
      return ids, lens

    src, sl = draw(p['src_vocab_size'])
    tgt, tl = draw(p['tgt_vocab_size'])
    return {
        'source_tensors': [torch.from_numpy(src).to(device), torch.from_numpy(sl).to(device)],
        'target_tensors': [torch.from_numpy(tgt).to(device), torch.from_numpy(tl).to(device)],
        'packed_source': packing.to_device(packing.pack_ids(src, sl), device),
        'packed_target': packing.to_device(packing.pack_ids(tgt, tl, shift_right=True), device),
        'num_tokens': int(sl.sum() + tl.sum()),
        'padded_tokens': int(B * (src.shape[1] + tgt.shape[1])),
    }

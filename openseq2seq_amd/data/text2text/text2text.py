"""ParallelTextDataLayer — parameter schema and batch contract of
open_seq2seq/data/text2text/text2text.py:36-298.

Batch contract: source_tensors = [ids int32 [B, Ls] (pad 0), lengths [B]], target_tensors
likewise; special ids PAD=0, EOS=1, S=2, UNK=3 (SpecialTextTokens, :14-20); with
pad_vocab_to_eight the vocabulary sizes are rounded up to a multiple of 8. The line-file /
tf.data plumbing is host-side and not re-created; `synthetic_batch` draws the workload
SURVEY.md §8d defines (lengths U[8, max_length], ids U{4..V-1}) and also emits the packed
index vectors the kernels consume (parts/transformer/packing.py)."""
from __future__ import absolute_import, division, print_function

import enum

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

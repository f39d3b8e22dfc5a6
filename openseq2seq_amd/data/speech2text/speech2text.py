"""Speech2TextDataLayer — parameter schema and batch contract of
open_seq2seq/data/speech2text/speech2text.py:22-485.

Batch contract (what build_graph()/input_tensors deliver to the model):
  source_tensors = [features [B, T, num_audio_features], src_length int32 [B]]
      T = max frames in the batch padded up to a multiple of `pad_to`
      (speech2text.py:313-317), frames = 1 + num_samples // hop (librosa backend)
  target_tensors = [transcript ids int32 [B, Lmax] (pad value 0), tgt_length int32 [B]]
      vocabulary = characters of vocab_file; tgt_vocab_size = len(vocab) + 1 (CTC blank
      is the LAST id, speech2text.py:123-125).
The tf.data/py_func plumbing of the reference is host-side and not re-created; the
feature math (speech_utils.get_speech_features_librosa) runs on the GPU
(csrc/logmel.hip) — see data/speech2text/speech_utils.py. `synthetic_batch` draws
the synthetic workload SURVEY.md §8d defines for benchmarking.
"""
from __future__ import absolute_import, division, print_function

import numpy as np
import torch

from ..data_layer import DataLayer

DEFAULT_VOCAB = [" "] + [chr(ord("a") + i) for i in range(26)] + ["'"]


def load_pre_existing_vocabulary(path, read_chars=True):
  """open_seq2seq/data/utils.py: one token per line; first char when read_chars."""
  idx = 0
  vocab = {}
  with open(path, "r", encoding="utf-8") as f:
    for line in f:
      if len(line.rstrip("\n")) == 0 and not line.startswith(" "):
        continue
      tok = line[0] if read_chars else line.rstrip("\n").split("\t")[0]
      vocab[tok] = idx
      idx += 1
  return vocab


class Speech2TextDataLayer(DataLayer):
  @staticmethod
  def get_required_params():
    return dict(DataLayer.get_required_params(), **{
        'num_audio_features': int,
        'input_type': ['spectrogram', 'mfcc', 'logfbank'],
        'vocab_file': None,    # str; None selects the 28-character toy vocabulary
        'dataset_files': list,
    })

  @staticmethod
  def get_optional_params():
    return dict(DataLayer.get_optional_params(), **{
        'backend': ['psf', 'librosa'], 'augmentation': dict, 'pad_to': int,
        'max_duration': float, 'min_duration': float, 'bpe': bool, 'autoregressive': bool,
        'syn_enable': bool, 'syn_subdirs': list, 'window_size': float,
        'window_stride': float, 'dither': float, 'norm_per_feature': bool,
        'window': ['hanning', 'hamming', 'none'], 'num_fft': int,
        'precompute_mel_basis': bool, 'sample_freq': int, 'gain': float,
        'features_mean': np.ndarray, 'features_std_dev': np.ndarray,
    })

  def __init__(self, params, model, num_workers, worker_id):
    super(Speech2TextDataLayer, self).__init__(params, model, num_workers, worker_id)
    p = self.params
    if p.get('bpe', False) or p.get('autoregressive', False):
      raise NotImplementedError("bpe / autoregressive targets")
    if p['vocab_file']:
      p['char2idx'] = load_pre_existing_vocabulary(p['vocab_file'], read_chars=True)
    else:
      p['char2idx'] = {c: i for i, c in enumerate(DEFAULT_VOCAB)}
    p['idx2char'] = {i: c for c, i in p['char2idx'].items()}
    # add one for the implied blank token (speech2text.py:123-125)
    p['tgt_vocab_size'] = len(p['char2idx']) + 1
    self._input_tensors = {}

  def build_graph(self):
    return self

  @property
  def input_tensors(self):
    return self._input_tensors

  def get_size_in_samples(self):
    return None

  # ------------------------------------------------------------------------
  def frames_for_samples(self, n_samples):
    hop = int(self.params.get('sample_freq', 16000) * self.params.get('window_stride', 10e-3))
    return 1 + n_samples // hop

  def synthetic_batch(self, device, seed, fixed_frames=None, min_dur=2.0):
    """Synthetic training batch: durations U[min_dur, max_duration] s,
    features N(0,1) in bf16 (what per-feature normalisation produces), T padded to
    `pad_to`; labels U{0..V-2} of length U[10, T'/2] where T' = frames after the
    stride-2 encoder layer. Returns the batch dict for Model.train_step."""
    p = self.params
    B = p['batch_size']
    rng = np.random.RandomState(seed)
    sr = p.get('sample_freq', 16000)
    if fixed_frames is None:
      dur = rng.uniform(min_dur, p.get('max_duration', 16.7), size=B)
      frames = np.array([self.frames_for_samples(int(d * sr)) for d in dur], np.int32)
    else:
      frames = np.full(B, fixed_frames, np.int32)
    pad_to = p.get('pad_to', 8)
    T = int(-(-frames.max() // pad_to) * pad_to)
    F = p['num_audio_features']
    g = torch.Generator(device="cpu").manual_seed(seed)
    feats = torch.randn(B, T, F, generator=g).to(torch.bfloat16)
    V = p['tgt_vocab_size']
    tlen = np.array([rng.randint(10, max(11, f // 4)) for f in frames], np.int32)
    Lmax = int(tlen.max())
    tgt = rng.randint(0, V - 1, size=(B, Lmax)).astype(np.int32)
    for b in range(B):
      tgt[b, tlen[b]:] = 0
      feats[b, frames[b]:] = 0
    return {
        'source_tensors': [feats.to(device), torch.from_numpy(frames).to(device)],
        'target_tensors': [torch.from_numpy(tgt).to(device), torch.from_numpy(tlen).to(device)],
        'num_frames': int(frames.sum()), 'padded_frames': int(B * T),
    }

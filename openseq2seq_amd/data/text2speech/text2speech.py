"""Text2SpeechDataLayer — parameter schema and batch contract of
open_seq2seq/data/text2speech/text2speech.py:22-545.

Batch contract (train): source_tensors = [text ids int32 [B,S] (pad <p>=0, </s>=2),
text lengths [B]] (+ [style mel fp32 [B,T,n_mel], style lengths] with style_input "wav");
target_tensors = [spec fp32 [B,T,n_mel(+n_mag)], stop token fp32 [B,T], spec lengths [B]].
With output_type "both" the target is concat(log-mel, magnitude) and the magnitude is
exp(log_mag) when exp_mag (:486-488); pad_EOS appends frames carrying log(data_min) /
data_min and stop token 1 and the returned lengths include them (:494-545). The character
vocabulary is the file + 3 special symbols (:137-148). The csv/wav reading path is
host-side; feature extraction itself is the GPU kernel behind
data/text2speech/speech_utils.py. `synthetic_batch` draws SURVEY.md 8d's cfg-5 workload."""
from __future__ import absolute_import, division, print_function

import io
import math

import numpy as np
import torch

from ..data_layer import DataLayer


class Text2SpeechDataLayer(DataLayer):
  @staticmethod
  def get_required_params():
    return dict(DataLayer.get_required_params(), **{
        'dataset_location': str, 'dataset': ['LJ', 'MAILABS'], 'num_audio_features': None,
        'output_type': ['magnitude', 'mel', 'both'], 'vocab_file': str, 'dataset_files': list,
        'feature_normalize': bool,
    })

  @staticmethod
  def get_optional_params():
    return dict(DataLayer.get_optional_params(), **{
        'pad_to': int, 'mag_power': int, 'pad_EOS': bool, 'pad_value': float,
        'feature_normalize_mean': float, 'feature_normalize_std': float, 'trim': bool,
        'data_min': None, 'duration_min': int, 'duration_max': int,
        'mel_type': ['slaney', 'htk'], 'exp_mag': bool, 'style_input': [None, 'wav'],
        'n_samples_train': int, 'n_samples_eval': int, 'n_fft': int, 'fmax': float,
        'max_normalization': bool, 'use_cache': bool,
    })

  def __init__(self, params, model, num_workers=None, worker_id=None):
    super(Text2SpeechDataLayer, self).__init__(params, model, num_workers or 1, worker_id or 0)
    p = self.params
    chars = []
    try:
      with io.open(p['vocab_file'], "r", encoding="utf-8") as f:
        chars = [line.rstrip("\n") for line in f if line.rstrip("\n") != ""]
    except IOError:
      chars = [chr(32 + i) for i in range(91)]
    # ids start at 3: <p> = 0, <s> = 1, </s> = 2  (text2speech.py:137-148)
    p['char2idx'] = {c: i + 3 for i, c in enumerate(chars)}
    p['char2idx'].update({'<p>': 0, '<s>': 1, '</s>': 2})
    p['idx2char'] = {i: c for c, i in p['char2idx'].items()}
    p['src_vocab_size'] = len(p['char2idx'])
    self._both = p['output_type'] == "both"
    self._exp_mag = bool(p.get('exp_mag', False)) and self._both

  def build_graph(self):
    return self

  @property
  def input_tensors(self):
    return {}

  def feature_sizes(self):
    naf = self.params['num_audio_features']
    if self._both:
      return naf['mel'], naf['magnitude']
    return (naf, 0) if self.params['output_type'] == 'mel' else (0, naf)

  def iterate_batches(self, device, seed=0, drop_remainder=None, num_batches=2):
    """eval / infer passes (run.py). The csv + wav reader of the reference's TTS data layer
    (data/text2speech/text2speech.py:150-420) is outside SURVEY 8f's file-reading row (ASR only): the
    pass runs over `num_batches` synthetic batches of the configured shapes."""
    for i in range(num_batches):
      yield self.synthetic_batch(device, seed + i)

  def synthetic_batch(self, device, seed, fixed_text=None, fixed_frames=None):
    p = self.params
    B = p['batch_size']
    rng = np.random.RandomState(seed)
    pad_to = p.get('pad_to', 8)
    V = p['src_vocab_size']
    sl = (np.full(B, fixed_text) if fixed_text else rng.randint(20, 201, size=B)).astype(np.int32)
    S = int(-(-sl.max() // pad_to) * pad_to)
    text = np.zeros((B, S), np.int32)
    for b in range(B):
      text[b, :sl[b] - 1] = rng.randint(3, V, size=sl[b] - 1)
      text[b, sl[b] - 1] = 2
    dmin, dmax = p.get('duration_min', 24), min(p.get('duration_max', 1024), 1024)
    raw = (np.full(B, fixed_frames) if fixed_frames else rng.randint(dmin, dmax + 1, size=B)).astype(np.int64)
    # pad_EOS: at least one EOS frame, total a multiple of pad_to (:494-536)
    num_pad = pad_to - ((raw + 1) % pad_to) + 1
    tl = (raw + num_pad).astype(np.int32) if p.get('pad_EOS', True) else raw.astype(np.int32)
    T = int(tl.max())
    n_mel, n_mag = self.feature_sizes()
    dm = p.get('data_min', 1e-5)
    dm_mel = dm['mel'] if isinstance(dm, dict) else dm
    dm_mag = dm['magnitude'] if isinstance(dm, dict) else dm
    feats = []
    if n_mel:
      mel = np.maximum(rng.randn(B, T, n_mel).astype(np.float32) - 2.0, math.log(dm_mel))
      feats.append(mel)
    if n_mag:
      lm = np.maximum(rng.randn(B, T, n_mag).astype(np.float32) - 3.0, math.log(dm_mag))
      feats.append(np.exp(lm) if self._exp_mag else lm)
    spec = np.concatenate(feats, -1)
    stop = np.zeros((B, T), np.float32)
    for b in range(B):
      if p.get('pad_EOS', True):
        spec[b, raw[b]:, :n_mel] = math.log(dm_mel)
        if n_mag:
          spec[b, raw[b]:, n_mel:] = dm_mag if self._exp_mag else math.log(dm_mag)
        stop[b, raw[b]:] = 1.0
      spec[b, tl[b]:] = 0.0
    t = lambda a: torch.from_numpy(a).to(device)
    src = [t(text), t(sl)]
    if p.get('style_input', None) == 'wav':
      src += [t(np.ascontiguousarray(spec[:, :, :n_mel])), t(tl)]
    return {'source_tensors': src, 'target_tensors': [t(spec), t(stop), t(tl)],
            'num_frames': int(tl.sum())}

"""TTS feature extraction front end — host side of
open_seq2seq/data/text2speech/speech_utils.py:98-182 (get_speech_features). The
per-utterance arithmetic runs on the GPU (csrc/tts_features.hip, os2s_tts_spectrogram);
the host prepares the constant tables: the periodic Hann window librosa.stft uses by default
(scipy.signal.get_window('hann', n_fft, fftbins=True)) and the mel filterbank
librosa.filters.mel(sr, n_fft, n_mels, htk=True, norm=None) (mel_type 'htk', :160-172) or the
Slaney / area-normalised one (mel_type 'slaney')."""
from __future__ import absolute_import, division, print_function

import math

import numpy as np
import torch

from ... import capi
from ..speech2text.speech_utils import mel_basis_slaney


def mel_basis_htk(sample_freq, n_fft, n_mels, fmin=0.0, fmax=None):
  """librosa.filters.mel(..., htk=True, norm=None): HTK mel scale 2595 log10(1 + f/700),
  triangular filters with unit peak."""
  fmax = sample_freq / 2.0 if fmax is None else fmax
  hz2mel = lambda f: 2595.0 * np.log10(1.0 + np.asarray(f, np.float64) / 700.0)
  mel2hz = lambda m: 700.0 * (10.0 ** (np.asarray(m, np.float64) / 2595.0) - 1.0)
  fftfreqs = np.linspace(0, sample_freq / 2.0, 1 + n_fft // 2)
  mel_f = mel2hz(np.linspace(hz2mel(fmin), hz2mel(fmax), n_mels + 2))
  fdiff = np.diff(mel_f)
  ramps = np.subtract.outer(mel_f, fftfreqs)
  w = np.zeros((n_mels, 1 + n_fft // 2))
  for i in range(n_mels):
    w[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
  return w.astype(np.float32)


class TTSFeatureFrontEnd(object):
  """Constant tables + launcher for a Text2SpeechDataLayer configuration."""

  def __init__(self, device, sample_freq, n_fft, num_features, features_type="both", hop_length=None,
               mag_power=1, data_min=1e-5, mel_type="htk", mel_basis=None):
    self.device, self.n_fft = device, n_fft
    self.hop = hop_length if hop_length is not None else n_fft // 4
    self.mag_power = mag_power
    self.features_type = features_type
    dm = data_min
    self.dm_mel = dm["mel"] if isinstance(dm, dict) else dm
    self.dm_mag = dm["magnitude"] if isinstance(dm, dict) else dm
    if isinstance(num_features, dict):
      self.n_mels, self.n_mag = num_features["mel"], num_features["magnitude"]
    else:
      self.n_mels = num_features if "mel" in features_type else 0
      self.n_mag = num_features if features_type == "magnitude" else 0
    n = np.arange(n_fft)
    self.window = torch.from_numpy((0.5 - 0.5 * np.cos(2 * np.pi * n / n_fft)).astype(np.float32)).to(device)
    self.mel_start = self.mel_len = self.mel_wt = None
    if self.n_mels:
      if mel_basis is None:
        mel_basis = (mel_basis_htk if mel_type == "htk" else mel_basis_slaney)(sample_freq, n_fft, self.n_mels)
      self.mel_basis = np.asarray(mel_basis, np.float32)
      starts, lens = [], []
      for m in range(self.n_mels):
        nz = np.nonzero(self.mel_basis[m])[0]
        starts.append(int(nz[0]) if len(nz) else 0)
        lens.append(int(nz[-1] - nz[0] + 1) if len(nz) else 0)
      wt = np.zeros((max(max(lens), 1), self.n_mels), np.float32)
      for m in range(self.n_mels):
        wt[:lens[m], m] = self.mel_basis[m, starts[m]:starts[m] + lens[m]]
      self.mel_start = torch.tensor(starts, dtype=torch.int32, device=device)
      self.mel_len = torch.tensor(lens, dtype=torch.int32, device=device)
      self.mel_wt = torch.from_numpy(wt).to(device)

  def frames(self, n_samples):
    return 1 + int(n_samples) // self.hop

  def __call__(self, signal, n_samples, max_samples=None):
    """signal fp32 [B, Nmax] (device), n_samples int32 [B]. Returns (mel, log_mag) fp32
    [B, T, *] (None for the part a features_type does not produce); frames past an
    utterance's end hold log(data_min)."""
    nmax = int(max_samples) if max_samples is not None else signal.shape[1]
    T = self.frames(nmax)
    return capi.tts_spectrogram(
        signal, n_samples, self.window, n_fft=self.n_fft, hop=self.hop, T=T, mag_power=self.mag_power,
        data_min_mag=self.dm_mag, data_min_mel=self.dm_mel, n_mag=self.n_mag, n_mels=self.n_mels,
        mel_start=self.mel_start, mel_len=self.mel_len, mel_wt=self.mel_wt,
        pad_mel=math.log(self.dm_mel), pad_mag=math.log(self.dm_mag))

"""Speech feature extraction front end — host side of
open_seq2seq/data/speech2text/speech_utils.py (get_speech_features :275-319,
get_speech_features_librosa :322-441). The per-sample arithmetic runs on the GPU
(csrc/logmel.hip, os2s_logmel); the host only prepares constant tables once:
the analysis window and the mel filterbank (what the reference precomputes with
librosa.filters.mel in speech2text.py:167-183).

librosa 0.6.3 conventions used by the reference's call sites (librosa itself is not
vendored in the reference): window passed as the CALLABLE np.hanning => symmetric
Hann of win_length, zero-padded centred to n_fft; filters.mel defaults htk=False,
norm=1 => Slaney mel scale with area normalisation.
"""
from __future__ import absolute_import, division, print_function

import math

import numpy as np
import torch

from ... import capi

WINDOWS_FNS = {"hanning": np.hanning, "hamming": np.hamming, "none": None}


def _hz_to_mel(f):
  f = np.asanyarray(f, dtype=np.float64)
  f_sp, min_log_hz = 200.0 / 3, 1000.0
  logstep = np.log(6.4) / 27.0
  return np.where(f >= min_log_hz,
                  min_log_hz / f_sp + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep,
                  f / f_sp)


def _mel_to_hz(m):
  m = np.asanyarray(m, dtype=np.float64)
  f_sp, min_log_hz = 200.0 / 3, 1000.0
  min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
  return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_basis_slaney(sample_freq, n_fft, n_mels, fmin=0.0, fmax=None):
  """Equivalent of librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) (htk=False, norm=1)."""
  fmax = sample_freq / 2.0 if fmax is None else fmax
  fftfreqs = np.linspace(0, sample_freq / 2.0, 1 + n_fft // 2)
  mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
  fdiff = np.diff(mel_f)
  ramps = np.subtract.outer(mel_f, fftfreqs)
  w = np.zeros((n_mels, 1 + n_fft // 2))
  for i in range(n_mels):
    w[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
  w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
  return w.astype(np.float32)


class LogMelFrontEnd(object):
  """Constant tables + launcher for the 'logfbank' features of a Speech2TextDataLayer
  configuration (params as in speech_utils.get_speech_features :275-306)."""

  def __init__(self, params, device):
    self.device = device
    sr = params.get('sample_freq', 16000)
    self.sample_freq = sr
    if params.get('backend', 'psf') != 'librosa' or params.get('input_type') != 'logfbank':
      raise NotImplementedError("GPU front ends: backend='librosa' + input_type='logfbank' (the Jasper "
                                "configs), backend='psf' + input_type='spectrogram' (DeepSpeech2) or 'logfbank'")
    self.n_mels = params['num_audio_features']
    window_size = params.get('window_size', 20e-3)
    window_stride = params.get('window_stride', 10e-3)
    self.win_length = int(sr * window_size)
    self.hop = int(sr * window_stride)
    self.n_fft = params.get('num_fft', None) or 2 ** math.ceil(math.log2(window_size * sr))
    self.dither = params.get('dither', 0.0)
    self.norm_per_feature = params.get('norm_per_feature', False)
    self.gain = params.get('gain', None)
    self.pad_to = params.get('pad_to', 8)
    wfn = WINDOWS_FNS[params.get('window', 'hanning')]
    win = wfn(self.win_length) if wfn is not None else np.ones(self.win_length)
    full = np.zeros(self.n_fft, np.float32)
    lp = (self.n_fft - self.win_length) // 2
    full[lp:lp + self.win_length] = win
    basis = params.get('mel_basis', None)
    if basis is None:
      basis = mel_basis_slaney(sr, self.n_fft, self.n_mels, 0, int(sr / 2))
    self.mel_basis = np.asarray(basis, np.float32)
    starts, lens = [], []
    for m in range(self.n_mels):
      nz = np.nonzero(self.mel_basis[m])[0]
      starts.append(int(nz[0]) if len(nz) else 0)
      lens.append(int(nz[-1] - nz[0] + 1) if len(nz) else 0)
    maxlen = max(max(lens), 1)
    wt = np.zeros((maxlen, self.n_mels), np.float32)
    for m in range(self.n_mels):
      wt[:lens[m], m] = self.mel_basis[m, starts[m]:starts[m] + lens[m]]
    self.window = torch.from_numpy(full).to(device)
    self.mel_start = torch.tensor(starts, dtype=torch.int32, device=device)
    self.mel_len = torch.tensor(lens, dtype=torch.int32, device=device)
    self.mel_wt = torch.from_numpy(wt).to(device)

  def frames(self, n_samples):
    return 1 + int(n_samples) // self.hop

  def __call__(self, signal, n_samples, max_samples=None, seed=0, want_f32=False):
    """signal [B,Nmax] (float32 or int16, device), n_samples int32 [B] (device).
    max_samples: host int = max(n_samples) (avoids a device sync); defaults to Nmax.
    Returns (features bf16 [B,Tpad,F], frames int32 [B], fp32 copy or None)."""
    nmax = int(max_samples) if max_samples is not None else signal.shape[1]
    tmax = self.frames(nmax)
    tpad = -(-tmax // self.pad_to) * self.pad_to if self.pad_to > 0 else tmax
    return capi.logmel(signal, n_samples, self.window, self.mel_start, self.mel_len,
                       self.mel_wt, hop=self.hop, n_mels=self.n_mels, tmax=tmax, tpad=tpad,
                       dither=self.dither, seed=seed,
                       fixed_gain=self.gain if self.gain is not None else -1.0,
                       norm_per_feature=self.norm_per_feature, want_f32=want_f32,
                       n_fft=self.n_fft)


class PsfSpectrogramFrontEnd(object):
  """Launcher for the 'spectrogram' features of the python_speech_features backend
  (get_speech_features_psf, speech_utils.py:444-535; the DeepSpeech2 configs). Same call
  signature as LogMelFrontEnd; dither / gain / norm_per_feature do not exist on this path."""

  def __init__(self, params, device):
    self.device = device
    sr = params.get('sample_freq', 16000)
    self.sample_freq = sr
    if params.get('backend', 'psf') != 'psf' or params.get('input_type') != 'spectrogram':
      raise NotImplementedError("PsfSpectrogramFrontEnd implements backend='psf', input_type='spectrogram'")
    self.num_features = params['num_audio_features']
    self.win_length = int(sr * params.get('window_size', 20e-3))
    self.hop = int(sr * params.get('window_stride', 10e-3))
    self.pad_to = params.get('pad_to', 8)
    self.gain = None
    if self.num_features > self.win_length // 2 + 1:        # speech_utils.py:501-502
      raise AssertionError("num_features for spectrogram should be <= (sample_freq * window_size // 2 + 1)")

  def frames(self, n_samples):
    n = int(n_samples)
    f = 1 if n <= self.win_length else 1 + -(-(n - self.win_length) // self.hop)
    if self.pad_to > 0 and f % self.pad_to:
      f += self.pad_to - f % self.pad_to
    return f

  def __call__(self, signal, n_samples, max_samples=None, seed=0, want_f32=False):
    nmax = int(max_samples) if max_samples is not None else signal.shape[1]
    return capi.psf_spectrogram(signal, n_samples, n_win=self.win_length, n_step=self.hop,
                                pad_to=self.pad_to, num_features=self.num_features,
                                tpad=self.frames(nmax), want_f32=want_f32)


def psf_filterbanks(nfilt, nfft, samplerate, lowfreq=0.0, highfreq=None):
  """python_speech_features.get_filterbanks (0.6): nfilt triangular filters on the HTK mel scale
  (2595 log10(1 + f / 700)), corner bins floor((nfft + 1) * hz / samplerate), unnormalised: [nfilt, nfft/2 + 1]."""
  highfreq = highfreq or samplerate / 2.0
  hz2mel = lambda hz: 2595.0 * np.log10(1.0 + hz / 700.0)
  mel2hz = lambda mel: 700.0 * (10.0 ** (mel / 2595.0) - 1.0)
  melpoints = np.linspace(hz2mel(lowfreq), hz2mel(highfreq), nfilt + 2)
  bins = np.floor((nfft + 1) * mel2hz(melpoints) / samplerate)
  fb = np.zeros([nfilt, nfft // 2 + 1])
  for j in range(nfilt):
    for i in range(int(bins[j]), int(bins[j + 1])):
      fb[j, i] = (i - bins[j]) / (bins[j + 1] - bins[j])
    for i in range(int(bins[j + 1]), int(bins[j + 2])):
      fb[j, i] = (bins[j + 2] - i) / (bins[j + 2] - bins[j + 1])
  return fb


class PsfLogfbankFrontEnd(PsfSpectrogramFrontEnd):
  """Launcher for the 'logfbank' features of the python_speech_features backend (get_speech_features_psf,
  speech_utils.py:517-535: psf.logfbank with nfft = 512, preemph = 0.97; the toy Wave2Letter / TDNN test
  configurations). Framing and padding as the spectrogram path."""

  def __init__(self, params, device):
    self.device = device
    sr = params.get('sample_freq', 16000)
    self.sample_freq = sr
    if params.get('backend', 'psf') != 'psf' or params.get('input_type') != 'logfbank':
      raise NotImplementedError("PsfLogfbankFrontEnd implements backend='psf', input_type='logfbank'")
    self.num_features = params['num_audio_features']
    self.win_length = int(sr * params.get('window_size', 20e-3))
    self.hop = int(sr * params.get('window_stride', 10e-3))
    self.pad_to = params.get('pad_to', 8)
    self.gain = None
    self.nfft = 512
    if self.win_length > self.nfft:
      raise NotImplementedError("psf.logfbank truncates frames longer than nfft = 512 (window_size > 32 ms)")
    fb = psf_filterbanks(self.num_features, self.nfft, sr, 0.0, sr / 2.0)
    self.fb = torch.from_numpy(np.ascontiguousarray(fb, np.float32)).to(device)

  def __call__(self, signal, n_samples, max_samples=None, seed=0, want_f32=False):
    nmax = int(max_samples) if max_samples is not None else signal.shape[1]
    return capi.psf_logfbank(signal, n_samples, self.fb, n_win=self.win_length, n_step=self.hop,
                             pad_to=self.pad_to, nfft=self.nfft, tpad=self.frames(nmax), want_f32=want_f32)


def make_front_end(params, device):
  """The GPU front end of a Speech2TextDataLayer configuration: log-mel (librosa backend, the
  Jasper / wav2letter configs) or psf spectrogram (the DeepSpeech2 configs)."""
  if params.get('backend', 'psf') == 'psf' and params.get('input_type') == 'spectrogram':
    return PsfSpectrogramFrontEnd(params, device)
  if params.get('backend', 'psf') == 'psf' and params.get('input_type') == 'logfbank':
    return PsfLogfbankFrontEnd(params, device)
  return LogMelFrontEnd(params, device)


# ---- speed perturbation filter (resampy 'kaiser_best') --------------------------------------------
KAISER_BEST = dict(num_zeros=64, precision=9, rolloff=0.9475937167399596, beta=14.769656459379492)


def sinc_window(num_zeros=64, precision=9, rolloff=0.945, beta=14.769656459379492):
  """resampy.filters.sinc_window with a Kaiser taper: the right half of the band-limited
  interpolation filter, num_zeros * 2^precision + 1 taps (resampy's published construction;
  'kaiser_best' = KAISER_BEST). Returns (interp_win float32, num_table = 2^precision)."""
  num_bits = 2 ** precision
  n = num_bits * num_zeros
  sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
  taper = np.kaiser(2 * n + 1, beta)[n:]
  return (taper * sinc_win).astype(np.float32), num_bits


def read_wav(filename):
  """scipy.io.wavfile.read as used by get_speech_features_from_file (speech_utils.py:186-196):
  (sample_freq, int16 / float32 mono signal)."""
  from scipy.io import wavfile
  sample_freq, signal = wavfile.read(filename)
  if signal.ndim > 1:
    signal = signal[:, 0]
  return sample_freq, signal

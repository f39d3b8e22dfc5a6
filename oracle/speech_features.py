"""ORACLE (test infrastructure only): NumPy restatement of the ASR log-mel front end
open_seq2seq/data/speech2text/speech_utils.py:322-441 (get_speech_features_librosa,
'logfbank' + 'spectrogram' branches), normalize_signal :216-222, preemphasis :271-272.

The arithmetic the reference delegates to librosa 0.6.3 (requirements.txt:8; NOT
vendored under /root/reference) is restated here from librosa's documented
behaviour:
  * librosa.core.stft(y, n_fft, hop_length, win_length, window=np.hanning (callable
    => SYMMETRIC Hann of win_length), center=True): reflect-pad n_fft//2, window
    zero-padded CENTRED to n_fft, frames = 1 + len(y)//hop, rfft of each frame;
  * librosa.filters.mel(sr, n_fft, n_mels, fmin=0, fmax=sr/2, htk=False, norm=1):
    Slaney mel scale + area normalisation, float32 result.
PARITY STATUS (round 5): the GLUE of all four paths (signal normalisation, pre-emphasis, FFT-size rule, log floors,
slicing, padding to a multiple of pad_to, normalisation) is pinned to the reference's OWN get_speech_features,
executed from its file with independent stand-ins for librosa / python_speech_features
(oracle/ref_shim/audio_libs; tests/test_ref_exec_frontend.py: 3e-5 / 2e-5). The libraries' own arithmetic stays a
restatement (now written twice, independently). Before that:
the reference's own test (speech_utils_test.py:45-85) pins only
shapes and mean~0/std~1 — no feature values — and only for the psf backend:
"parity unpinned" for the values; the pieces are cross-checked against
scipy.signal.stft and closed forms in tests/test_oracle_speech_features.py.
"""
import math

import numpy as np


def normalize_signal(signal, gain=None):
  if gain is None:
    gain = 1.0 / (np.max(np.abs(signal)) + 1e-5)
  return signal * gain


def preemphasis(signal, coeff=0.97):
  return np.append(signal[0], signal[1:] - coeff * signal[:-1])


def hz_to_mel(f):
  f = np.asanyarray(f, dtype=np.float64)
  f_sp = 200.0 / 3
  mels = f / f_sp
  min_log_hz = 1000.0
  min_log_mel = min_log_hz / f_sp
  logstep = np.log(6.4) / 27.0
  return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep,
                  mels)


def mel_to_hz(m):
  m = np.asanyarray(m, dtype=np.float64)
  f_sp = 200.0 / 3
  freqs = f_sp * m
  min_log_hz = 1000.0
  min_log_mel = min_log_hz / f_sp
  logstep = np.log(6.4) / 27.0
  return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_filterbank(sr, n_fft, n_mels, fmin=0.0, fmax=None):
  """librosa.filters.mel(..., htk=False, norm=1) -> float32 [n_mels, 1+n_fft//2]."""
  if fmax is None:
    fmax = sr / 2.0
  fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
  mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
  fdiff = np.diff(mel_f)
  ramps = np.subtract.outer(mel_f, fftfreqs)
  w = np.zeros((n_mels, 1 + n_fft // 2))
  for i in range(n_mels):
    lower = -ramps[i] / fdiff[i]
    upper = ramps[i + 2] / fdiff[i + 1]
    w[i] = np.maximum(0, np.minimum(lower, upper))
  enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
  w *= enorm[:, None]
  return w.astype(np.float32)


def stft_power(y, n_fft, hop, win_length, window_fn=np.hanning):
  """|librosa.core.stft(...)|**2 -> [1+n_fft//2, frames] (float64)."""
  y = np.asarray(y, np.float64)
  win = window_fn(win_length) if window_fn is not None else np.ones(win_length)
  lpad = (n_fft - win_length) // 2
  fft_window = np.zeros(n_fft)
  fft_window[lpad:lpad + win_length] = win
  yp = np.pad(y, n_fft // 2, mode="reflect")
  nfr = 1 + (len(yp) - n_fft) // hop
  idx = np.arange(n_fft)[None, :] + hop * np.arange(nfr)[:, None]
  frames = yp[idx] * fft_window[None, :]
  spec = np.fft.rfft(frames, axis=1)
  return (np.abs(spec) ** 2).T


def get_speech_features_librosa(signal, sample_freq, num_features, features_type="logfbank",
                                window_size=20e-3, window_stride=10e-3, window_fn=np.hanning,
                                num_fft=None, dither=0.0, norm_per_feature=False,
                                mel_basis=None, gain=None, mean=None, std_dev=None,
                                dither_noise=None):
  """speech_utils.py:322-441 without augmentation. dither_noise: optional N(0,1) vector
  (the reference draws it from np.random; passing it makes the function deterministic)."""
  signal = normalize_signal(np.asarray(signal).astype(np.float32), gain)
  audio_duration = len(signal) * 1.0 / sample_freq
  n_window_size = int(sample_freq * window_size)
  n_window_stride = int(sample_freq * window_stride)
  num_fft = num_fft or 2 ** math.ceil(math.log2(window_size * sample_freq))
  if dither > 0:
    noise = dither_noise if dither_noise is not None else np.random.randn(*signal.shape)
    signal = signal + dither * noise
  if features_type == "spectrogram":
    powspec = stft_power(signal, n_window_size, n_window_stride, n_window_size, window_fn)
    powspec[powspec <= 1e-30] = 1e-30
    features = 10 * np.log10(powspec.T)
    assert num_features <= n_window_size // 2 + 1
    features = features[:, :num_features]
  elif features_type == "logfbank":
    signal = preemphasis(signal, coeff=0.97)
    S = stft_power(signal, num_fft, n_window_stride, n_window_size, window_fn)
    if mel_basis is None:
      mel_basis = mel_filterbank(sample_freq, num_fft, num_features, 0, int(sample_freq / 2))
    features = np.log(np.dot(mel_basis.astype(np.float64), S) + 1e-20).T
  else:
    raise ValueError("Unknown features type: {}".format(features_type))
  norm_axis = 0 if norm_per_feature else None
  if mean is None:
    mean = np.mean(features, axis=norm_axis)
  if std_dev is None:
    std_dev = np.std(features, axis=norm_axis)
  features = (features - mean) / std_dev
  return features, audio_duration


# ---- psf backend, 'spectrogram' features (the DeepSpeech2 configs) ---------------------------------
# open_seq2seq/data/speech2text/speech_utils.py:444-535 (get_speech_features_psf). The frame /
# spectrum arithmetic is delegated to python_speech_features (requirements.txt:9, version 0.6 on
# PyPI; NOT vendored under /root/reference) and restated here from its published sigproc module:
#   framesig(sig, frame_len, frame_step, winfunc): numframes = 1 if len <= frame_len else
#       1 + ceil((len - frame_len) / frame_step); the signal is zero-padded to
#       (numframes-1)*frame_step + frame_len; frame i = padded[i*step : i*step + frame_len] * winfunc(frame_len)
#   magspec  = |numpy.fft.rfft(frames, NFFT)|;  powspec = magspec**2 / NFFT
#   logpowspec(frames, NFFT, norm=1): ps = powspec; ps[ps <= 1e-30] = 1e-30; lps = 10*log10(ps);
#       return lps - max(lps)
# PARITY STATUS: "parity unpinned" for the values (the reference's test, speech_utils_test.py:45-85,
# pins shapes, mean ~ 0, std ~ 1 and the num_features assertion — reproduced in
# tests/test_oracle_speech_features.py); the DFT is cross-checked against a direct O(N^2) sum.
def psf_framesig(sig, frame_len, frame_step, winfunc=np.hanning):
  slen = len(sig)
  numframes = 1 if slen <= frame_len else 1 + int(math.ceil((1.0 * slen - frame_len) / frame_step))
  padlen = int((numframes - 1) * frame_step + frame_len)
  padsignal = np.concatenate((np.asarray(sig, np.float64), np.zeros((padlen - slen,))))
  idx = np.arange(frame_len)[None, :] + (np.arange(numframes) * frame_step)[:, None]
  return padsignal[idx] * winfunc(frame_len)[None, :]


def psf_logpowspec(frames, nfft, norm=True):
  ps = np.square(np.abs(np.fft.rfft(frames, nfft))) / nfft
  ps[ps <= 1e-30] = 1e-30
  lps = 10.0 * np.log10(ps)
  return lps - np.max(lps) if norm else lps


def get_speech_features_psf_spectrogram(signal, sample_freq, num_features, pad_to=8,
                                        window_size=20e-3, window_stride=10e-3):
  """speech_utils.py:473-535, features_type='spectrogram' (no augmentation): returns
  (features float32 [frames, num_features], audio_duration)."""
  signal = (normalize_signal(np.asarray(signal).astype(np.float32)) * 32767.0).astype(np.int16)
  audio_duration = len(signal) * 1.0 / sample_freq
  n_window_size = int(sample_freq * window_size)
  n_window_stride = int(sample_freq * window_stride)
  length = 1 + int(math.ceil((1.0 * signal.shape[0] - n_window_size) / n_window_stride))
  if pad_to > 0 and length % pad_to != 0:
    pad_size = (pad_to - length % pad_to) * n_window_stride
    signal = np.pad(signal, (0, pad_size), mode='constant')
  frames = psf_framesig(signal, n_window_size, n_window_stride, np.hanning)
  features = psf_logpowspec(frames, n_window_size)
  assert num_features <= n_window_size // 2 + 1, \
      "num_features for spectrogram should be <= (sample_freq * window_size // 2 + 1)"
  features = features[:, :num_features]
  if pad_to > 0:
    assert features.shape[0] % pad_to == 0
  mean = np.mean(features)
  std_dev = np.std(features)
  return ((features - mean) / std_dev).astype(np.float32), audio_duration


# ---- python_speech_features 0.6: logfbank (published algorithm restated; the package is not in the
# reference tree — requirements.txt names it) -------------------------------------------------------------
#   preemphasis(signal, coeff): append(signal[0], signal[1:] - coeff * signal[:-1])
#   fbank: frames = framesig(signal, winlen*sr, winstep*sr, winfunc = ones)   (rectangular)
#          pspec  = |rfft(frames, nfft)|^2 / nfft;  feat = pspec . get_filterbanks(nfilt, nfft, sr, low, high)^T
#          feat == 0 -> float eps;  logfbank = log(feat)
#   get_filterbanks: HTK mel scale 2595 log10(1 + f / 700), corner bins floor((nfft + 1) * hz / sr),
#          rising / falling unnormalised triangles
# PARITY STATUS: "parity unpinned" for the values (the reference's speech_utils_test.py pins shapes, mean ~ 0 and
# std ~ 1 for this backend); cross-checked in tests/test_oracle_speech_features.py against a direct O(N^2) DFT.
def psf_get_filterbanks(nfilt, nfft, samplerate, lowfreq=0.0, highfreq=None):
  highfreq = highfreq or samplerate / 2.0
  lowmel = 2595.0 * np.log10(1.0 + lowfreq / 700.0)
  highmel = 2595.0 * np.log10(1.0 + highfreq / 700.0)
  melpoints = np.linspace(lowmel, highmel, nfilt + 2)
  bins = np.floor((nfft + 1) * (700.0 * (10.0 ** (melpoints / 2595.0) - 1.0)) / samplerate)
  fbank = np.zeros([nfilt, nfft // 2 + 1])
  for j in range(nfilt):
    for i in range(int(bins[j]), int(bins[j + 1])):
      fbank[j, i] = (i - bins[j]) / (bins[j + 1] - bins[j])
    for i in range(int(bins[j + 1]), int(bins[j + 2])):
      fbank[j, i] = (bins[j + 2] - i) / (bins[j + 2] - bins[j + 1])
  return fbank


def psf_logfbank(signal, samplerate, winlen, winstep, nfilt, nfft, lowfreq, highfreq, preemph):
  signal = np.asarray(signal, np.float64)
  signal = np.append(signal[0], signal[1:] - preemph * signal[:-1])
  frames = psf_framesig(signal, int(round(winlen * samplerate)), int(round(winstep * samplerate)),
                        lambda n: np.ones((n,)))
  pspec = np.square(np.abs(np.fft.rfft(frames, nfft))) / nfft
  feat = np.dot(pspec, psf_get_filterbanks(nfilt, nfft, samplerate, lowfreq, highfreq).T)
  feat = np.where(feat == 0, np.finfo(float).eps, feat)
  return np.log(feat)


def get_speech_features_psf_logfbank(signal, sample_freq, num_features, pad_to=8, window_size=20e-3,
                                     window_stride=10e-3):
  """speech_utils.py:473-535, features_type='logfbank' (no augmentation): returns
  (features float32 [frames, num_features], audio_duration)."""
  signal = (normalize_signal(np.asarray(signal).astype(np.float32)) * 32767.0).astype(np.int16)
  audio_duration = len(signal) * 1.0 / sample_freq
  n_window_size = int(sample_freq * window_size)
  n_window_stride = int(sample_freq * window_stride)
  length = 1 + int(math.ceil((1.0 * signal.shape[0] - n_window_size) / n_window_stride))
  if pad_to > 0 and length % pad_to != 0:
    pad_size = (pad_to - length % pad_to) * n_window_stride
    signal = np.pad(signal, (0, pad_size), mode='constant')
  features = psf_logfbank(signal, sample_freq, window_size, window_stride, num_features, 512, 0,
                          sample_freq / 2, 0.97)
  if pad_to > 0:
    assert features.shape[0] % pad_to == 0
  mean = np.mean(features)
  std_dev = np.std(features)
  return ((features - mean) / std_dev).astype(np.float32), audio_duration

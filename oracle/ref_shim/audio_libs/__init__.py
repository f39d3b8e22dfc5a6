"""TEST INFRASTRUCTURE ONLY — stand-ins for the two third-party audio libraries the reference's ASR front end calls
(open_seq2seq/data/speech2text/speech_utils.py), so that the REFERENCE'S OWN get_speech_features_librosa /
get_speech_features_psf can be executed where they lie (neither library is installed, there is no network):

  * librosa 0.6.3 (requirements.txt:8): core.stft, filters.mel — the two calls of the 'logfbank' / 'spectrogram' paths
  * python_speech_features 0.6 (requirements.txt:9): sigproc.framesig / magspec / powspec / logpowspec,
    get_filterbanks, fbank, logfbank

Written from the two libraries' published algorithms, independently of oracle/speech_features.py (which restates the same
library arithmetic inside its own functions): the fixtures produced through these stand-ins therefore check the
oracle's library arithmetic a second way, and pin the reference's GLUE — signal normalisation, pre-emphasis, the FFT size
rule, log floors, feature slicing, the padding to a multiple of `pad_to`, global / per-feature normalisation — to its
own source. Never imported by the product; installed under the names `librosa` / `python_speech_features` only inside
tests/golden/make_ref_exec.py.
"""
import sys
import types

import numpy as np

# ------------------------------------------------------------------------------------------------ librosa


def _pad_center(w, size):
  lpad = (size - len(w)) // 2
  return np.pad(w, (lpad, size - len(w) - lpad), mode="constant")


def stft(y, n_fft=2048, hop_length=None, win_length=None, window="hann", center=True, dtype=np.complex64,
         pad_mode="reflect"):
  """librosa.core.stft: window of win_length (a callable is called with win_length: np.hanning gives the SYMMETRIC
  Hann; the string 'hann' the periodic one) zero-padded on both sides to n_fft; center=True reflect-pads n_fft // 2
  samples at both ends; frame t starts at t * hop_length; 1 + len(y) // hop frames; rfft -> [1 + n_fft / 2, frames]."""
  win_length = n_fft if win_length is None else win_length
  hop_length = win_length // 4 if hop_length is None else hop_length
  if callable(window):
    w = np.asarray(window(win_length), dtype=np.float64)
  elif window == "hann":
    w = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win_length) / win_length)
  else:
    raise NotImplementedError(window)
  w = _pad_center(w, n_fft)
  y = np.asarray(y)
  if center:
    y = np.pad(y, n_fft // 2, mode=pad_mode)
  n_frames = 1 + (len(y) - n_fft) // hop_length
  idx = np.arange(n_fft)[:, None] + hop_length * np.arange(n_frames)[None, :]
  frames = y[idx] * w[:, None]
  return np.fft.rfft(frames, axis=0).astype(dtype)


def _hz_to_mel(f):
  f = np.atleast_1d(np.asarray(f, dtype=np.float64))
  m = f / (200.0 / 3)
  big = f >= 1000.0
  m[big] = 15.0 + np.log(f[big] / 1000.0) / (np.log(6.4) / 27.0)
  return m


def _mel_to_hz(m):
  m = np.atleast_1d(np.asarray(m, dtype=np.float64))
  f = m * (200.0 / 3)
  big = m >= 15.0
  f[big] = 1000.0 * np.exp((np.log(6.4) / 27.0) * (m[big] - 15.0))
  return f


def mel(sr, n_fft, n_mels=128, fmin=0.0, fmax=None, htk=False, norm=1):
  """librosa.filters.mel (0.6): triangles over the rfft bin frequencies with corners equally spaced on the mel scale
  — Slaney's (linear below 1 kHz, logarithmic above) or, with htk=True, 2595 log10(1 + f / 700); norm=1: each filter
  scaled by 2 / (its bandwidth in Hz), norm=None: peak 1."""
  fmax = sr / 2.0 if fmax is None else fmax
  fft_f = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
  if htk:
    to_mel = lambda f: 2595.0 * np.log10(1.0 + np.asarray(f, dtype=np.float64) / 700.0)       # noqa: E731
    corners = 700.0 * (10.0 ** (np.linspace(to_mel(fmin), to_mel(fmax), n_mels + 2) / 2595.0) - 1.0)
  else:
    corners = _mel_to_hz(np.linspace(_hz_to_mel(fmin)[0], _hz_to_mel(fmax)[0], n_mels + 2))
  fb = np.zeros((n_mels, len(fft_f)))
  for i in range(n_mels):
    lo, ce, hi = corners[i], corners[i + 1], corners[i + 2]
    up = (fft_f - lo) / (ce - lo)
    down = (hi - fft_f) / (hi - ce)
    fb[i] = np.maximum(0.0, np.minimum(up, down))
    if norm == 1:
      fb[i] *= 2.0 / (hi - lo)
  return fb.astype(np.float32)


def magphase(D, power=1):             # noqa: N803
  """librosa.magphase: (|D| ^ power, D / |D|)."""
  mag = np.abs(D)
  phase = np.exp(1.j * np.angle(D))
  return mag ** power, phase


# ------------------------------------------------------------------------------------------------ python_speech_features


def framesig(sig, frame_len, frame_step, winfunc=lambda x: np.ones((x,)), stride_trick=True):
  slen = len(sig)
  frame_len, frame_step = int(round(frame_len)), int(round(frame_step))
  numframes = 1 if slen <= frame_len else 1 + int(np.ceil((1.0 * slen - frame_len) / frame_step))
  padlen = int((numframes - 1) * frame_step + frame_len)
  padded = np.concatenate((sig, np.zeros((padlen - slen,))))
  idx = np.arange(frame_len)[None, :] + frame_step * np.arange(numframes)[:, None]
  return padded[idx] * winfunc(frame_len)[None, :]


def magspec(frames, NFFT):          # noqa: N803
  return np.absolute(np.fft.rfft(frames, NFFT))


def powspec(frames, NFFT):          # noqa: N803
  return 1.0 / NFFT * np.square(magspec(frames, NFFT))


def logpowspec(frames, NFFT, norm=1):          # noqa: N803
  ps = powspec(frames, NFFT)
  ps[ps <= 1e-30] = 1e-30
  lps = 10 * np.log10(ps)
  return lps - np.max(lps) if norm else lps


def preemphasis(signal, coeff=0.95):
  return np.append(signal[0], signal[1:] - coeff * signal[:-1])


def get_filterbanks(nfilt=20, nfft=512, samplerate=16000, lowfreq=0, highfreq=None):
  """psf's HTK-style triangles: corners equally spaced on mel = 2595 log10(1 + f / 700), snapped to FFT bins
  floor((nfft + 1) f / samplerate)."""
  highfreq = highfreq or samplerate / 2
  hz2mel = lambda hz: 2595 * np.log10(1 + hz / 700.0)          # noqa: E731
  mel2hz = lambda m: 700 * (10 ** (m / 2595.0) - 1)            # noqa: E731
  melpoints = np.linspace(hz2mel(lowfreq), hz2mel(highfreq), nfilt + 2)
  bins = np.floor((nfft + 1) * mel2hz(melpoints) / samplerate)
  fbank = np.zeros([nfilt, nfft // 2 + 1])
  for j in range(nfilt):
    for i in range(int(bins[j]), int(bins[j + 1])):
      fbank[j, i] = (i - bins[j]) / (bins[j + 1] - bins[j])
    for i in range(int(bins[j + 1]), int(bins[j + 2])):
      fbank[j, i] = (bins[j + 2] - i) / (bins[j + 2] - bins[j + 1])
  return fbank


def fbank(signal, samplerate=16000, winlen=0.025, winstep=0.01, nfilt=26, nfft=512, lowfreq=0, highfreq=None,
          preemph=0.97, winfunc=lambda x: np.ones((x,))):
  highfreq = highfreq or samplerate / 2
  signal = preemphasis(signal, preemph)
  frames = framesig(signal, winlen * samplerate, winstep * samplerate, winfunc)
  pspec = powspec(frames, nfft)
  energy = np.sum(pspec, 1)
  energy = np.where(energy == 0, np.finfo(float).eps, energy)
  feat = np.dot(pspec, get_filterbanks(nfilt, nfft, samplerate, lowfreq, highfreq).T)
  feat = np.where(feat == 0, np.finfo(float).eps, feat)
  return feat, energy


def logfbank(signal, samplerate=16000, winlen=0.025, winstep=0.01, nfilt=26, nfft=512, lowfreq=0, highfreq=None,
             preemph=0.97):
  feat, _ = fbank(signal, samplerate, winlen, winstep, nfilt, nfft, lowfreq, highfreq, preemph)
  return np.log(feat)


def lifter(cepstra, L=22):            # noqa: N803
  if L > 0:
    n = np.arange(cepstra.shape[1])
    return (1 + (L / 2.0) * np.sin(np.pi * n / L)) * cepstra
  return cepstra


def mfcc(signal, samplerate=16000, winlen=0.025, winstep=0.01, numcep=13, nfilt=26, nfft=None, lowfreq=0,
         highfreq=None, preemph=0.97, ceplifter=22, appendEnergy=True, winfunc=lambda x: np.ones((x,))):   # noqa: N803
  """python_speech_features.mfcc: log filterbank energies -> DCT-II (orthonormal) -> the first numcep coefficients
  -> sinusoidal lifter; with appendEnergy the first coefficient is the log of the frame energy."""
  from scipy.fftpack import dct
  feat, energy = fbank(signal, samplerate, winlen, winstep, nfilt, nfft or 512, lowfreq, highfreq, preemph, winfunc)
  feat = dct(np.log(feat), type=2, axis=1, norm="ortho")[:, :numcep]
  feat = lifter(feat, ceplifter)
  if appendEnergy:
    feat[:, 0] = np.log(energy)
  return feat


def resample(x, sr_orig, sr_new, axis=-1, filter="kaiser_best", **kwargs):   # noqa: A002
  """resampy.resample's CONTRACT (output length int(n * sr_new / sr_orig), band-limited) through scipy's Fourier
  resampler — not resampy's windowed-sinc interpolation, which oracle/augment.py restates: this stand-in only lets the
  reference's augmentation code run (its own test checks lengths)."""
  from scipy.signal import resample as fft_resample
  n = int(x.shape[axis] * float(sr_new) / float(sr_orig))
  return fft_resample(x, n, axis=axis).astype(x.dtype)


def install():
  """Registers `librosa`, `python_speech_features`, `resampy` (the stand-ins above) and an empty `h5py` module (imported
  by speech_utils.py at module level, used only by the caching path that is not executed)."""
  lib = types.ModuleType("librosa")
  lib.core = types.ModuleType("librosa.core")
  lib.core.stft = stft
  lib.stft = stft
  lib.magphase = magphase
  lib.filters = types.ModuleType("librosa.filters")
  lib.filters.mel = mel
  lib.feature = types.ModuleType("librosa.feature")
  psf = types.ModuleType("python_speech_features")
  psf.sigproc = types.ModuleType("python_speech_features.sigproc")
  for n, f in (("framesig", framesig), ("magspec", magspec), ("powspec", powspec), ("logpowspec", logpowspec),
               ("preemphasis", preemphasis)):
    setattr(psf.sigproc, n, f)
  psf.get_filterbanks, psf.fbank, psf.logfbank, psf.mfcc, psf.lifter = get_filterbanks, fbank, logfbank, mfcc, lifter
  rsm = types.ModuleType("resampy")
  rsm.resample = resample
  mods = {"librosa": lib, "librosa.core": lib.core, "librosa.filters": lib.filters, "librosa.feature": lib.feature,
          "python_speech_features": psf, "python_speech_features.sigproc": psf.sigproc,
          "h5py": types.ModuleType("h5py"), "resampy": rsm}
  sys.modules.update(mods)
  return mods

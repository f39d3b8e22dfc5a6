"""ORACLE (test infrastructure only): NumPy restatement of the TTS feature extraction
open_seq2seq/data/text2speech/speech_utils.py:98-182 (get_speech_features) and of the target
assembly of data/text2speech/text2speech.py:459-545 ("both" = concat(log-mel, magnitude),
exp_mag, pad_EOS).

The arithmetic delegated to librosa 0.6.3 (requirements.txt:8; not under /root/reference) is
restated from its documented behaviour:
  * librosa.stft(y, n_fft): hop_length = n_fft // 4, win_length = n_fft, window 'hann' =
    scipy.signal.get_window('hann', n_fft, fftbins=True) (PERIODIC Hann), center=True with
    np.pad(mode='reflect') of n_fft // 2, frames = 1 + len(y) // hop;
  * librosa.magphase(D, power) -> |D| ** power;
  * librosa.filters.mel(sr, n_fft, n_mels, htk=True, norm=None): HTK mel scale, unit-peak
    triangles (the TTS call site, :160-172).
PARITY STATUS (round 5): the reference's own get_speech_features ("both": mel + magnitude, two parameter sets) is
executed from its file on independent stand-ins for librosa.stft / magphase / filters.mel
(oracle/ref_shim/audio_libs) and this restatement matches it to 1e-4 (tests/test_ref_exec_frontend.py); librosa's own
arithmetic stays a restatement (no value tests in the reference); the STFT is cross-checked against
scipy.signal.stft in tests/test_oracle_tts_features.py."""
import math

import numpy as np


def mel_filterbank_htk(sr, n_fft, n_mels, fmin=0.0, fmax=None):
  fmax = sr / 2.0 if fmax is None else fmax
  hz2mel = lambda f: 2595.0 * np.log10(1.0 + np.asarray(f, np.float64) / 700.0)
  mel2hz = lambda m: 700.0 * (10.0 ** (np.asarray(m, np.float64) / 2595.0) - 1.0)
  fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
  mel_f = mel2hz(np.linspace(hz2mel(fmin), hz2mel(fmax), n_mels + 2))
  fdiff = np.diff(mel_f)
  ramps = np.subtract.outer(mel_f, fftfreqs)
  w = np.zeros((n_mels, 1 + n_fft // 2))
  for i in range(n_mels):
    lower = -ramps[i] / fdiff[i]
    upper = ramps[i + 2] / fdiff[i + 1]
    w[i] = np.maximum(0, np.minimum(lower, upper))
  return w.astype(np.float32)


def stft(y, n_fft, hop=None):
  """complex [1 + n_fft//2, frames] as librosa.stft(y, n_fft)."""
  hop = n_fft // 4 if hop is None else hop
  y = np.asarray(y, np.float64)
  win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)
  yp = np.pad(y, n_fft // 2, mode="reflect")
  frames = 1 + len(y) // hop
  out = np.empty((1 + n_fft // 2, frames), np.complex128)
  for t in range(frames):
    out[:, t] = np.fft.rfft(yp[t * hop:t * hop + n_fft] * win)
  return out


def get_speech_features(signal, fs, num_features, features_type="both", n_fft=1024, hop_length=None,
                        mag_power=2, data_min=1e-5, mel_basis=None):
  dm_mel = data_min["mel"] if isinstance(data_min, dict) else data_min
  dm_mag = data_min["magnitude"] if isinstance(data_min, dict) else data_min
  n_mel = num_features["mel"] if isinstance(num_features, dict) else num_features
  n_mag = num_features["magnitude"] if isinstance(num_features, dict) else num_features
  mag = np.abs(stft(signal, n_fft, hop_length)) ** mag_power
  mag_features = None
  if features_type in ("magnitude", "both"):
    mag_features = np.log(np.clip(mag, dm_mag, None)).T[:, :n_mag]
    if features_type == "magnitude":
      return mag_features
  if mel_basis is None:
    mel_basis = mel_filterbank_htk(fs, n_fft, n_mel)
  feats = np.log(np.clip(np.dot(mel_basis.astype(np.float64), mag), dm_mel, None)).T
  return [feats, mag_features] if features_type == "both" else feats

"""ORACLE (test infrastructure only): NumPy restatement of the speech data layer's signal
augmentation, open_seq2seq/data/speech2text/speech_utils.py:
  normalize_signal :225-231, augment_audio_signal :234-272 (speed perturbation through
  resampy.resample(x, sr, int(sr * a), filter='kaiser_best'); additive noise),
  SpecAugment boxes :419-433.
resampy is a third-party dependency (requirements.txt: `resampy`, unpinned; the 0.2.x API)
that is NOT vendored in /root/reference and not installable here: its published algorithm is
restated — filters.sinc_window (Kaiser-tapered sinc half-window) and interpn.resample_f (the
interpolation loop) — so this piece is "parity unpinned" by the reference; it is cross-checked
against scipy.signal.resample_poly on band-limited signals (tests/test_oracle_augment.py).
"""
import numpy as np

KAISER_BEST = dict(num_zeros=64, precision=9, rolloff=0.9475937167399596, beta=14.769656459379492)


def normalize_signal(signal, gain=None):
  if gain is None:
    gain = 1.0 / (np.max(np.abs(signal)) + 1e-5)
  return signal * gain


def sinc_window(num_zeros=64, precision=9, rolloff=0.945, beta=14.769656459379492):
  num_bits = 2 ** precision
  n = num_bits * num_zeros
  sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
  taper = np.kaiser(2 * n + 1, beta)[n:]
  return taper * sinc_win, num_bits


def resample(x, sr_orig, sr_new, **filt):
  """resampy.resample for a 1-D signal (core.py + interpn.resample_f)."""
  sample_ratio = float(sr_new) / sr_orig
  n_out = int(x.shape[0] * sample_ratio)
  interp_win, num_table = sinc_window(**(filt or KAISER_BEST))
  if sample_ratio < 1:
    interp_win = interp_win * sample_ratio
  interp_delta = np.zeros_like(interp_win)
  interp_delta[:-1] = np.diff(interp_win)
  y = np.zeros(n_out, dtype=x.dtype)
  scale = min(1.0, sample_ratio)
  time_increment = 1.0 / sample_ratio
  index_step = int(scale * num_table)
  nwin = interp_win.shape[0]
  n_orig = x.shape[0]
  for t in range(n_out):
    time_register = t * time_increment
    n = int(time_register)
    frac = scale * (time_register - n)
    index_frac = frac * num_table
    offset = int(index_frac)
    eta = index_frac - offset
    i_max = min(n + 1, (nwin - offset) // index_step)
    idx = offset + np.arange(i_max) * index_step
    acc = np.dot(interp_win[idx] + eta * interp_delta[idx], x[n - np.arange(i_max)])
    frac = scale - frac
    index_frac = frac * num_table
    offset = int(index_frac)
    eta = index_frac - offset
    k_max = min(n_orig - n - 1, (nwin - offset) // index_step)
    idx = offset + np.arange(k_max) * index_step
    acc += np.dot(interp_win[idx] + eta * interp_delta[idx], x[n + 1 + np.arange(k_max)])
    y[t] = acc
  return y


def spec_augment(features, boxes):
  """boxes: (t0, t1, f0, f1) half-open; the reference zeroes features[t0:t1, f0:f1]."""
  out = features.copy()
  for t0, t1, f0, f1 in boxes:
    out[t0:t1, f0:f1] = 0
  return out

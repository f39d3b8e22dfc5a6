"""CPU checks of the TTS feature oracle: its STFT restatement against scipy.signal.stft
(periodic Hann, reflect padding, hop n_fft/4 — scipy scales by 1/sum(window)), the HTK mel
basis against closed-form properties, shapes/clipping of get_speech_features."""
import math

import numpy as np
import scipy.signal

from oracle import tts_features as otf


def test_stft_matches_scipy():
  rng = np.random.RandomState(0)
  y = rng.randn(4000)
  for n_fft in (800, 1024, 256):
    D = otf.stft(y, n_fft)
    win = scipy.signal.get_window("hann", n_fft, fftbins=True)
    _, _, Z = scipy.signal.stft(y, window=win, nperseg=n_fft, noverlap=n_fft - n_fft // 4,
                                boundary=None, padded=False)
    yp = np.pad(y, n_fft // 2, mode="reflect")
    _, _, Z = scipy.signal.stft(yp, window=win, nperseg=n_fft, noverlap=n_fft - n_fft // 4,
                                boundary=None, padded=False)
    Z = Z * win.sum()
    n = min(D.shape[1], Z.shape[1])
    assert D.shape == (n_fft // 2 + 1, 1 + len(y) // (n_fft // 4))
    np.testing.assert_allclose(D[:, :n], Z[:, :n], rtol=1e-9, atol=1e-8)


def test_htk_mel_basis_properties():
  w = otf.mel_filterbank_htk(16000, 800, 80)
  assert w.shape == (80, 401) and w.dtype == np.float32
  assert (w >= 0).all() and w.max() <= 1.0 + 1e-6
  peaks = w.argmax(1)
  assert (np.diff(peaks) > 0).all()                       # centres increase monotonically
  centres_hz = peaks * 16000 / 800.0
  mels = 2595 * np.log10(1 + centres_hz / 700.0)
  assert np.abs(np.diff(mels) - np.diff(mels).mean()).max() < 25.0     # equally spaced on the HTK scale


def test_features_shapes_and_clipping():
  rng = np.random.RandomState(1)
  y = np.concatenate([rng.randn(3000) * 0.1, np.zeros(1000)])
  mel, mag = otf.get_speech_features(y, 16000, {"mel": 80, "magnitude": 401}, "both", n_fft=800,
                                     mag_power=1, data_min={"mel": 1e-2, "magnitude": 1e-5})
  T = 1 + len(y) // 200
  assert mel.shape == (T, 80) and mag.shape == (T, 401)
  assert mel.min() >= math.log(1e-2) - 1e-6 and mag.min() >= math.log(1e-5) - 1e-6
  assert abs(mel[-1].max() - math.log(1e-2)) < 1e-6        # trailing silence clips to data_min

"""CPU: pins for the log-mel oracle (librosa 0.6.3 semantics restated):
shape contract of the reference's own test (speech_utils_test.py:45-85: frames,
mean~0, std~1), scipy.signal.stft cross-check of the STFT, and closed-form
properties of the Slaney mel filterbank."""
import numpy as np
import scipy.signal

from oracle import speech_features as sf


def test_shape_and_whitening():
  rng = np.random.RandomState(0)
  sig = (rng.randn(16000 * 3 + 77) * 3000).astype(np.int16)
  f, dur = sf.get_speech_features_librosa(sig, 16000, 64, "logfbank", norm_per_feature=True)
  assert f.shape == (1 + len(sig) // 160, 64)
  assert abs(dur - len(sig) / 16000.0) < 1e-9
  np.testing.assert_allclose(f.mean(0), 0, atol=1e-6)   # reference: places=6
  np.testing.assert_allclose(f.std(0), 1, atol=1e-6)


def test_stft_vs_scipy():
  rng = np.random.RandomState(1)
  y = rng.randn(4000)
  P = sf.stft_power(y, 512, 160, 320, np.hanning)
  win = np.zeros(512); win[96:416] = np.hanning(320)
  # scipy: same framing when boundary handled manually
  yp = np.pad(y, 256, mode="reflect")
  _, _, Z = scipy.signal.stft(yp, window=win, nperseg=512, noverlap=512 - 160, nfft=512,
                              boundary=None, padded=False, return_onesided=True)
  Z = Z * win.sum()   # scipy normalises by the window sum
  assert Z.shape[1] == P.shape[1]
  np.testing.assert_allclose(np.abs(Z) ** 2, P, rtol=1e-8, atol=1e-8)


def test_mel_filterbank_properties():
  M = sf.mel_filterbank(16000, 512, 64, 0, 8000)
  assert M.shape == (64, 257) and M.dtype == np.float32
  assert (M >= 0).all()
  # Slaney scale: linear below 1 kHz => first filters have equal width 200/3 * step
  pk = M.argmax(1)
  assert (np.diff(pk) >= 0).all()
  # area normalisation: each filter integrates to ~1 over Hz (bin width 31.25 Hz),
  # up to the sampling of the narrowest triangles on the 31.25 Hz bin grid
  area = M.sum(1) * (8000 / 256.0)
  np.testing.assert_allclose(area[8:], 1.0, rtol=0.15)
  # hz<->mel round trip
  f = np.array([0., 500., 1000., 4000., 8000.])
  np.testing.assert_allclose(sf.mel_to_hz(sf.hz_to_mel(f)), f, atol=1e-9)
  assert abs(sf.hz_to_mel(1000.0) - 15.0) < 1e-12

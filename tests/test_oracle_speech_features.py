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


# ---- psf backend, 'spectrogram' (the DeepSpeech2 configs) -----------------------------------------
def _speechlike(n, seed):
  rng = np.random.RandomState(seed)
  t = np.arange(n) / 16000.0
  return (0.3 * np.sin(2 * np.pi * 220 * t) + 0.1 * rng.randn(n)).astype(np.float32)


def test_psf_spectrogram_reference_relations():
  """speech_utils_test.py:45-85 restated for the psf 'spectrogram' branch: shape
  [frames % pad_to == 0, num_features], mean ~ 0, std ~ 1, and the num_features assertion."""
  import pytest
  for n, pad_to, F in [(16000, 8, 161), (23456, 8, 160), (5000, 16, 96), (48000, 0, 160)]:
    feats, dur = sf.get_speech_features_psf_spectrogram(_speechlike(n, n), 16000, F, pad_to=pad_to)
    frames = 1 + int(np.ceil((n - 320) / 160.0))
    if pad_to:
      frames = -(-frames // pad_to) * pad_to
    assert feats.shape == (frames, F)
    assert abs(feats.mean()) < 1e-3 and abs(feats.std() - 1.0) < 1e-3
    assert dur == n / 16000.0
  with pytest.raises(AssertionError):
    sf.get_speech_features_psf_spectrogram(_speechlike(16000, 0), 16000, 162)


def test_psf_logpowspec_against_direct_dft():
  """psf_logpowspec (rfft) == 10 log10(|sum x e^{-2 pi i k n / N}|^2 / N) by the definition."""
  x = _speechlike(2000, 3) * 1000
  frames = sf.psf_framesig(x, 320, 160)
  assert frames.shape == (1 + int(np.ceil((2000 - 320) / 160.0)), 320)
  np.testing.assert_allclose(frames[0], x[:320].astype(np.float64) * np.hanning(320))
  # the tail frame is zero-padded
  last = (frames.shape[0] - 1) * 160
  assert np.all(frames[-1][2000 - last:] == 0)
  n = np.arange(320)
  k = np.arange(161)
  dft = frames @ np.exp(-2j * np.pi * np.outer(n, k) / 320)
  want = 10 * np.log10(np.maximum(np.abs(dft) ** 2 / 320, 1e-30))
  got = sf.psf_logpowspec(frames, 320, norm=False)
  np.testing.assert_allclose(got, want, atol=1e-6)
  np.testing.assert_allclose(sf.psf_logpowspec(frames, 320), want - want.max(), atol=1e-6)


def test_psf_spectrogram_int16_and_padding_rows():
  """int16 input goes through the same gain normalisation; the pad_to rows are the -300 dB
  floor of an all-zero frame and enter mean / std."""
  x = _speechlike(16000 + 37, 5)
  xi = (x / np.abs(x).max() * 20000).astype(np.int16)
  fa, _ = sf.get_speech_features_psf_spectrogram(xi, 16000, 160, pad_to=8)
  assert fa.shape[0] % 8 == 0
  real = 1 + int(np.ceil((len(xi) - 320) / 160.0))
  assert real < fa.shape[0]
  # frame `real` still overlaps the last 37 samples; the rows after it are all-zero frames and
  # hold one constant (the floor), far below every other row
  assert np.ptp(fa[real + 1:]) == 0.0 and fa[real + 1:].max() < fa[:real + 1].min()


def test_psf_logfbank_reference_relations_and_direct_dft():
  """The psf 'logfbank' branch (speech_utils.py:517-535): shapes / whitening as the reference's
  speech_utils_test.py pins them for this backend; the filter table and the spectrum against their
  definitions (a direct O(N^2) DFT of the pre-emphasised rectangular frames)."""
  for n, pad_to, F in [(16000, 8, 40), (23456, 8, 64), (5000, 16, 26), (48000, 0, 40)]:
    feats, dur = sf.get_speech_features_psf_logfbank(_speechlike(n, n), 16000, F, pad_to=pad_to)
    frames = 1 + int(np.ceil((n - 320) / 160.0))
    if pad_to:
      frames = -(-frames // pad_to) * pad_to
    assert feats.shape == (frames, F)
    assert abs(feats.mean()) < 1e-3 and abs(feats.std() - 1.0) < 1e-3
  fb = sf.psf_get_filterbanks(40, 512, 16000, 0, 8000)
  assert fb.shape == (40, 257) and fb.min() >= 0 and fb.max() <= 1.0
  assert np.all(fb.sum(1) > 0)                                   # every triangle has support
  peaks = fb.argmax(1)
  assert np.all(np.diff(peaks) > 0)                              # centres rise with the filter index
  x = (_speechlike(3000, 7) * 8000).astype(np.int16).astype(np.float64)
  y = np.append(x[0], x[1:] - 0.97 * x[:-1])
  got = sf.psf_logfbank(x, 16000, 0.02, 0.01, 40, 512, 0, 8000, 0.97)
  nfr = 1 + int(np.ceil((3000 - 320) / 160.0))
  pad = np.concatenate([y, np.zeros((nfr - 1) * 160 + 320 - 3000)])
  fr = np.stack([pad[i * 160:i * 160 + 320] for i in range(nfr)])
  dft = fr @ np.exp(-2j * np.pi * np.outer(np.arange(320), np.arange(257)) / 512)
  want = np.log(np.maximum((np.abs(dft) ** 2 / 512) @ fb.T, np.finfo(float).eps))
  np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-9)


def test_psf_logfbank_padding_rows():
  """pad_to rows: the first padding sample is -0.97 x[n-1] (pre-emphasis runs over the padded signal), rows
  past it are ln(eps) before the whitening — one constant, below every real row."""
  x = _speechlike(16000 + 37, 5)
  xi = (x / np.abs(x).max() * 20000).astype(np.int16)
  fa, _ = sf.get_speech_features_psf_logfbank(xi, 16000, 40, pad_to=8)
  real = 1 + int(np.ceil((len(xi) - 320) / 160.0))
  assert fa.shape[0] % 8 == 0 and real < fa.shape[0]
  assert np.ptp(fa[real + 1:]) == 0.0 and fa[real + 1:].max() < fa[:real + 1].min()

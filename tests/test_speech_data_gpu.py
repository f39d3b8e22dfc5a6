"""GPU tests of the real-file speech pipeline: the augmentation kernels
(os2s_augment_signal, os2s_spec_augment) against the NumPy oracle (oracle/augment.py), the
Speech2TextDataLayer batches against the float64 feature oracle, and a train -> eval (WER)
loop on generated wav files through the model / data-layer plugin API.

Tolerances: resampled samples are sums of <= 128 fp32 products of O(1) values — atol 2e-4 on
signals normalised to [-1, 1]; features as in test_logmel_gpu (bf16 output: atol 2e-2 on
N(0,1) values); SpecAugment boxes are exact."""
import csv
import os

import numpy as np
import pytest
import torch
from scipy.io import wavfile

pytestmark = pytest.mark.gpu

from oracle import augment as oa  # noqa: E402
from oracle import speech_features as osf  # noqa: E402


def _speechlike(n, rng, sr=16000):
  t = np.arange(n) / sr
  x = np.zeros(n)
  for f in (180.0, 950.0, 2700.0, 5200.0):
    x += rng.rand() * np.sin(2 * np.pi * f * t * (1 + 0.05 * t) + rng.rand())
  x += 0.05 * rng.randn(n)
  return (x / np.abs(x).max() * 12000).astype(np.int16)


def test_augment_signal_vs_oracle(cuda):
  from openseq2seq_amd import capi
  from openseq2seq_amd.data.speech2text.speech_utils import KAISER_BEST, sinc_window
  rng = np.random.RandomState(0)
  lens = [5000, 3211, 4800, 2000]
  stretch = [1.1, 0.9, 1.0, 0.95]
  sigs = [_speechlike(n, rng) for n in lens]
  ratios = [float(int(16000 * s)) / 16000 if s != 1.0 else 1.0 for s in stretch]
  n_out = [int(n * r) if r != 1.0 else n for n, r in zip(lens, ratios)]
  buf = np.zeros((4, max(lens)), np.int16)
  for b, s in enumerate(sigs):
    buf[b, :len(s)] = s
  win, num_table = sinc_window(**KAISER_BEST)
  ow, ot = oa.sinc_window(**oa.KAISER_BEST)
  assert num_table == ot and np.allclose(win, ow, atol=1e-7)
  dev = cuda
  out = capi.augment_signal(torch.from_numpy(buf).to(dev), torch.tensor(lens, dtype=torch.int32, device=dev),
                            torch.tensor(n_out, dtype=torch.int32, device=dev),
                            torch.tensor(ratios, dtype=torch.float64, device=dev), None,
                            torch.from_numpy(win).to(dev), num_table, max(n_out)).cpu().numpy()
  for b, s in enumerate(sigs):
    x = oa.normalize_signal(s.astype(np.float32))
    ref = oa.resample(x, 16000, int(16000 * stretch[b])) if stretch[b] != 1.0 else x
    assert len(ref) == n_out[b]
    np.testing.assert_allclose(out[b, :n_out[b]], ref, atol=2e-4, rtol=0)
    assert np.all(out[b, n_out[b]:] == 0)
  # additive noise: level and whiteness
  amp = [0.0, 1e-2, 1e-3, 0.0]
  noisy = capi.augment_signal(torch.from_numpy(buf).to(dev), torch.tensor(lens, dtype=torch.int32, device=dev),
                              torch.tensor(n_out, dtype=torch.int32, device=dev),
                              torch.tensor(ratios, dtype=torch.float64, device=dev),
                              torch.tensor(amp, dtype=torch.float32, device=dev),
                              torch.from_numpy(win).to(dev), num_table, max(n_out), seed=9).cpu().numpy()
  for b in range(4):
    d = (noisy[b] - out[b])[:n_out[b]]
    if amp[b] == 0:
      assert np.all(d == 0)
    else:
      assert abs(d.std() / amp[b] - 1) < 0.08 and abs(d.mean()) < 0.1 * amp[b]
      assert abs(np.corrcoef(d[:-1], d[1:])[0, 1]) < 0.08
  # fixed gain
  fg = capi.augment_signal(torch.from_numpy(buf).to(dev), torch.tensor(lens, dtype=torch.int32, device=dev),
                           torch.tensor(lens, dtype=torch.int32, device=dev),
                           torch.ones(4, dtype=torch.float64, device=dev), None, torch.from_numpy(win).to(dev),
                           num_table, max(lens), fixed_gain=0.5).cpu().numpy()
  np.testing.assert_allclose(fg[1, :lens[1]], sigs[1].astype(np.float32) * 0.5, rtol=1e-6)


def test_spec_augment_vs_oracle(cuda):
  from openseq2seq_amd import capi
  g = torch.Generator().manual_seed(0)
  B, T, F = 3, 50, 64
  x = torch.randn(B, T, F, generator=g).to(torch.bfloat16)
  boxes = [[(0, 50, 3, 9), (10, 22, 0, 64), (0, 0, 0, 0)], [(0, 40, 60, 64), (39, 40, 0, 64), (5, 6, 0, 64)],
           [(0, 50, 0, 0), (0, 0, 0, 64), (49, 50, 0, 64)]]
  got = capi.spec_augment(x.clone().to(cuda), torch.tensor(boxes, dtype=torch.int32, device=cuda)).cpu()
  for b in range(B):
    ref = oa.spec_augment(x[b].float().numpy(), boxes[b])
    assert np.array_equal(got[b].float().numpy(), ref)


def _dataset(tmp, n, seed, words):
  rng = np.random.RandomState(seed)
  rows = []
  for i in range(n):
    txt = " ".join(words[(i + j) % len(words)] for j in range(1 + i % 3))
    x = _speechlike(int((0.5 + 0.08 * len(txt)) * 16000), rng)
    path = os.path.join(tmp, "s%d_%d.wav" % (seed, i))
    wavfile.write(path, 16000, x)
    rows.append((path, os.path.getsize(path), txt))
  csv_path = os.path.join(tmp, "set%d.csv" % seed)
  with open(csv_path, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["wav_filename", "wav_filesize", "transcript"])
    w.writerows(rows)
  return csv_path, rows


def test_data_layer_batches_vs_oracle(cuda, tmp_path):
  from openseq2seq_amd.data.speech2text.speech2text import Speech2TextDataLayer
  csv_path, rows = _dataset(str(tmp_path), 5, 1, ["ten", "seconds", "it's"])
  dl = Speech2TextDataLayer(dict(mode="eval", batch_size=2, num_audio_features=64, input_type="logfbank",
                                 vocab_file=None, dataset_files=[csv_path], backend="librosa",
                                 norm_per_feature=True, window="hanning", pad_to=16, dither=0.0,
                                 sample_freq=16000), None, 1, 0)
  seen = 0
  for batch in dl.iterate_batches(cuda):
    feats, frames = batch['source_tensors']
    tgt, tl = batch['target_tensors']
    assert feats.shape[1] % 16 == 0
    # the host copy of the lengths (the convolution launcher's tile hint) is the device's lengths
    assert batch['source_lengths_host'].tolist() == frames.cpu().tolist()
    for b in range(feats.shape[0]):
      path, _, txt = rows[int(batch['source_ids'][b])]
      _, sig = wavfile.read(path)
      ref, dur = osf.get_speech_features_librosa(sig, 16000, 64, "logfbank", norm_per_feature=True)
      T = ref.shape[0]
      assert int(frames[b]) == T
      np.testing.assert_allclose(feats[b, :T].float().cpu().numpy(), ref, rtol=1e-2, atol=2e-2)
      assert float(feats[b, T:].abs().max()) == 0.0 if T < feats.shape[1] else True
      ids = tgt[b, :int(tl[b])].cpu().tolist()
      assert "".join(dl.params['idx2char'][i] for i in ids) == txt
      seen += 1
  assert seen == 5      # eval: no remainder dropped


def test_train_eval_on_wav_files(cuda, tmp_path):
  """Files -> GPU features (with speed perturbation + SpecAugment) -> training steps -> WER
  evaluation with a second model in eval mode (run.py --mode=train_eval)."""
  from openseq2seq_amd.configs.jasper import jasper10x5_config
  words = ["alpha", "bravo", "charlie", "delta"]
  train_csv, _ = _dataset(str(tmp_path), 8, 2, words)
  model_cls, params = jasper10x5_config(batch_size_per_gpu=4, use_horovod=False, max_steps=40)
  params["encoder_params"]["convnet_layers"] = [
      {"type": "conv1d", "repeat": 1, "kernel_size": [11], "stride": [2], "num_channels": 128,
       "padding": "SAME", "dilation": [1], "dropout_keep_prob": 1.0},
      {"type": "conv1d", "repeat": 2, "kernel_size": [11], "stride": [1], "num_channels": 128,
       "padding": "SAME", "dilation": [1], "dropout_keep_prob": 1.0, "residual": True},
      {"type": "conv1d", "repeat": 1, "kernel_size": [1], "stride": [1], "num_channels": 256,
       "padding": "SAME", "dilation": [1], "dropout_keep_prob": 1.0},
  ]
  params["lr_policy_params"] = {"learning_rate": 0.02, "min_lr": 1e-4, "power": 2.0}
  params["data_layer_params"].update(dataset_files=[train_csv], max_duration=8.0, dither=0.0)
  import copy
  tp = copy.deepcopy(params)
  tp["data_layer_params"]["augmentation"] = dict(speed_perturbation_ratio=0.05, n_freq_mask=1, n_time_mask=1,
                                                 width_freq_mask=4, width_time_mask=5)
  model = model_cls(tp, mode="train", hvd=None, device=cuda)
  model.compile()
  ep = copy.deepcopy(params)
  ep["data_layer_params"]["shuffle"] = False
  emodel = model_cls(ep, mode="eval", hvd=None, device=cuda)
  emodel.compile()
  dl = model.get_data_layer()
  assert dl.has_files()
  losses = []
  it = dl.iterate_batches(cuda, seed=3)
  for step in range(40):
    losses.append(float(model.train_step(next(it)).cpu()[0]))
  assert np.isfinite(losses).all()
  assert np.mean(losses[-5:]) < 0.7 * np.mean(losses[:5]), (losses[:5], losses[-5:])
  emodel.copy_weights_from(model)
  res = emodel.evaluate()
  assert res["samples_batches"] == 2 and 0.0 <= res["Eval WER"] <= 5.0


CONFIG_TEMPLATE = '''
from openseq2seq_amd.models import Speech2Text
from openseq2seq_amd.encoders import TDNNEncoder
from openseq2seq_amd.decoders import FullyConnectedCTCDecoder
from openseq2seq_amd.data import Speech2TextDataLayer
from openseq2seq_amd.losses import CTCLoss
from openseq2seq_amd.optimizers.lr_policies import poly_decay
from openseq2seq_amd.optimizers.novograd import NovoGrad

base_model = Speech2Text
base_params = {
  "random_seed": 0, "use_horovod": False, "num_gpus": 1, "batch_size_per_gpu": 4, "max_steps": 30,
  "print_loss_steps": 10, "eval_steps": 20, "save_checkpoint_steps": 20, "logdir": %(logdir)r,
  "optimizer": NovoGrad,
  "optimizer_params": {"beta1": 0.95, "beta2": 0.98, "epsilon": 1e-08, "weight_decay": 0.001,
                       "grad_averaging": False},
  "lr_policy": poly_decay, "lr_policy_params": {"learning_rate": 0.02, "min_lr": 1e-4, "power": 2.0},
  "larc_params": {"larc_eta": 0.001}, "dtype": "mixed", "loss_scaling": "Backoff",
  "encoder": TDNNEncoder,
  "encoder_params": {
    "convnet_layers": [
      {"type": "conv1d", "repeat": 1, "kernel_size": [11], "stride": [2], "num_channels": 128,
       "padding": "SAME", "dilation": [1], "dropout_keep_prob": 1.0},
      {"type": "conv1d", "repeat": 1, "kernel_size": [1], "stride": [1], "num_channels": 128,
       "padding": "SAME", "dilation": [1], "dropout_keep_prob": 1.0},
    ],
    "dropout_keep_prob": 1.0, "initializer": "xavier_initializer", "initializer_params": {"uniform": False},
    "normalization": "batch_norm", "activation_fn": "relu", "data_format": "channels_last",
    "use_conv_mask": True,
  },
  "decoder": FullyConnectedCTCDecoder,
  "decoder_params": {"initializer": "xavier_initializer", "use_language_model": False,
                     "infer_logits_to_pickle": False, "beam_width": 16, "alpha": 1.0, "beta": 0.0,
                     "trie_weight": 0.1, "lm_path": "", "trie_path": "", "alphabet_config_path": "",
                     "decoder_library_path": ""},
  "loss": CTCLoss, "loss_params": {},
  "data_layer": Speech2TextDataLayer,
  "data_layer_params": {"num_audio_features": 64, "input_type": "logfbank", "vocab_file": None,
                        "norm_per_feature": True, "window": "hanning", "sample_freq": 16000,
                        "pad_to": 16, "dither": 1e-5, "backend": "librosa"},
}
train_params = {"data_layer_params": {"dataset_files": [%(train)r], "max_duration": 8.0, "shuffle": True,
                                      "augmentation": {"n_freq_mask": 1, "width_freq_mask": 4}}}
eval_params = {"data_layer_params": {"dataset_files": [%(dev)r], "shuffle": False}}
infer_params = {"data_layer_params": {"dataset_files": [%(dev)r], "shuffle": False}}
'''


def test_run_py_train_eval_infer_checkpoints(cuda, tmp_path):
  """run.py end to end on wav files: train_eval writes checkpoints under the reference's
  variable names, eval / infer restore the latest one."""
  import subprocess
  import sys
  words = ["alpha", "bravo", "charlie"]
  train_csv, _ = _dataset(str(tmp_path), 8, 4, words)
  dev_csv, dev_rows = _dataset(str(tmp_path), 5, 5, words)
  logdir = str(tmp_path / "log")
  cfg = str(tmp_path / "cfg.py")
  with open(cfg, "w") as f:
    f.write(CONFIG_TEMPLATE % dict(logdir=logdir, train=train_csv, dev=dev_csv))
  repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

  def run(*extra):
    r = subprocess.run([sys.executable, os.path.join(repo, "run.py"), "--config_file=" + cfg] + list(extra),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=repo, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    return r.stdout

  out = run("--mode=train_eval")
  assert "Eval WER" in out and "Saved checkpoint" in out
  from openseq2seq_amd.utils import tensor_bundle
  assert os.path.exists(os.path.join(logdir, "model.ckpt-30.index"))         # tf.train.Saver V2 files
  assert os.path.exists(os.path.join(logdir, "model.ckpt-30.data-00000-of-00001"))
  ck = tensor_bundle.BundleReader(os.path.join(logdir, "model.ckpt-30"))
  names = set(ck.keys())
  k = "ForwardPass/w2l_encoder/conv11/kernel"
  assert k in names and "Loss_Optimization/FP32-master-copy/" + k in names
  assert ck[k].shape == (11, 64, 128)                       # TF conv1d layout [K, Cin, Cout]
  assert ck["ForwardPass/w2l_encoder/conv11/bn/moving_mean"].shape == (128,)
  assert "ForwardPass/w2l_encoder/conv11/bn/moving_variance" in names and int(ck["global_step"]) == 30
  fc = [n for n in names if n.endswith("fully_connected/kernel") and not n.startswith("Loss_")]
  assert len(fc) == 1 and ck[fc[0]].shape[0] == 128        # tf.layers.dense layout [Cin, Cout]
  assert open(os.path.join(logdir, "checkpoint")).read().splitlines() == [
      'model_checkpoint_path: "model.ckpt-30"', 'all_model_checkpoint_paths: "model.ckpt-30"']
  out = run("--mode=eval")
  assert "Restored checkpoint" in out and "Eval WER" in out
  inf = str(tmp_path / "infer.csv")
  out = run("--mode=infer", "--infer_output_file=" + inf)
  rows = list(csv.reader(open(inf)))
  assert rows[0] == ["wav_filename", "predicted_transcript"] and len(rows) == 6
  assert [r[0] for r in rows[1:]] == [r[0] for r in dev_rows]
  # language-model beam search as the decoder's text generation (fc_decoders.py:197-240): the
  # reference's sample bigram model only knows "ten seconds", so every word it emits is one of those
  gold = os.path.join(repo, "tests", "golden")
  alphabet = str(tmp_path / "alphabet.txt")
  with open(alphabet, "w") as f:
    f.write("\n".join([" "] + [chr(ord("a") + i) for i in range(26)] + ["'"]) + "\n")
  inf_lm = str(tmp_path / "infer_lm.csv")
  run("--mode=infer", "--infer_output_file=" + inf_lm, "--decoder_params/use_language_model=True",
      "--decoder_params/lm_path=" + os.path.join(gold, "ctc_test_lm.binary"),
      "--decoder_params/trie_path=" + os.path.join(gold, "ctc_test_lm.trie"),
      "--decoder_params/alphabet_config_path=" + alphabet, "--decoder_params/beam_width=64",
      "--decoder_params/alpha=4.0", "--decoder_params/beta=0.0", "--decoder_params/trie_weight=1.0",
      "--decoder_params/decoder_library_path=ctc_decoder_with_lm/libctc_decoder_with_kenlm.so")
  rows_lm = list(csv.reader(open(inf_lm)))
  assert len(rows_lm) == 6 and [r[0] for r in rows_lm[1:]] == [r[0] for r in dev_rows]
  assert rows_lm[1:] != rows[1:]
  # logits dump for the offline rescoring script (speech2text.py:327-346)
  import pickle
  dump_path = str(tmp_path / "logits.pkl")
  run("--mode=infer", "--infer_output_file=" + dump_path, "--decoder_params/infer_logits_to_pickle=True")
  dump = pickle.load(open(dump_path, "rb"))
  assert sorted(dump["logits"]) == sorted(r[0] for r in dev_rows) and abs(dump["step_size"] - 0.02) < 1e-9
  lg = dump["logits"][dev_rows[0][0]]
  assert lg.ndim == 2 and lg.shape[1] == 29 and dump["vocab"][0] == " "
  # --continue_learning resumes from the stored global step
  out = run("--mode=train", "--continue_learning", "--max_steps=35")
  assert "Restored checkpoint" in out and "(step 30)" in out


def test_checkpoint_roundtrip_transformer_layout(cuda, tmp_path):
  """Fused qkv / kv projections are exported as the reference's separate q, k, v Dense kernels
  ([in, out]) and re-imported bit-exactly."""
  from openseq2seq_amd.configs.transformer import transformer_config
  from openseq2seq_amd.utils import checkpoint
  cls, params = transformer_config(d_model=512, num_layers=1, num_heads=8, batch_size_per_gpu=4, vocab_size=96)
  m1 = cls(params, mode="train", hvd=None, device=cuda)
  m1.compile()
  prefix = checkpoint.save(m1, str(tmp_path), 7)
  ck = checkpoint.open_checkpoint(prefix)
  base = "ForwardPass/transformer_encoder/layer_0/self_attention/self_attention"
  for t in "qkv":
    assert ck["%s/%s/kernel" % (base, t)].shape == (512, 512)
  assert base + "/qkv/kernel" not in ck
  assert ck["ForwardPass/transformer_encoder/embedding_shared_weights/embedding_and_softmax/weights"].shape == (96, 512)
  qkv = m1.store.by_name(base + "/qkv/kernel").master.cpu().numpy()
  # mixed precision: the plain name holds DT_HALF (what a reference fp16 graph stores), the fp32 value
  # lives under the master-copy name (mp_wrapper.py:55-82)
  assert ck[base + "/k/kernel"].dtype == np.float16
  assert np.array_equal(ck[base + "/k/kernel"], qkv[0, 512:1024].T.astype(np.float16))
  assert np.array_equal(ck[checkpoint.MASTER_PREFIX + base + "/k/kernel"], qkv[0, 512:1024].T)
  m2 = cls(params, mode="train", hvd=None, device=cuda)
  m2.compile()
  m2.store.master.mul_(0.5)             # same seed => same init: make it differ before the restore
  m2.store.refresh_compute_copies()
  assert not torch.equal(m1.store.master, m2.store.master)
  assert checkpoint.load(m2, checkpoint.latest_checkpoint(str(tmp_path))) == []
  assert torch.equal(m1.store.master, m2.store.master) and torch.equal(m1.store.w16, m2.store.w16)

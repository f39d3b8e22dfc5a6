"""The reference's own ASR acceptance tests on its ten-utterance toy corpus (VERDICT round 4,
missing #2): open_seq2seq/models/speech2text_test.py:29-227 with speech2text_ds2_test.py /
speech2text_w2l_test.py, and the block-dropout runs of scripts/run_all_tests.sh:71-83.

The three configurations are loaded from open_seq2seq/test_utils/test_speech_configs/ (the reference's values,
tests/test_config_dropin.py compares them with the reference's files) and trained through run.py's loop, from
the repository root as the reference does. What the reference asserts is asserted here:

  convergence_test(5.0, 30.0, 0.1)   train loss < 5, eval loss < 30, WER < 0.1 — for dtype float32 AND "mixed"
                                     (this engine computes in bf16 with fp32 masters in both; "mixed" adds the
                                     loss scaler and the half / master twins of the checkpoint)
  mp_collection_test(14, 7 / 6)      trainable variables / variables a mixed-precision graph of the reference
                                     keeps as fp16 with an fp32 master copy, counted in the checkpoint the model
                                     writes under the reference's names
  infer_test                         250 epochs, then infer: file names in order, every transcript within 5
                                     characters of the truth
  run_all_tests.sh:71-83             train_eval with drop_block_prob 0.98, then eval with drop_block_index 0 / 1 / 2
"""
import copy
import csv
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = "open_seq2seq/test_utils/test_speech_configs/%s_test_config.py"


def _models(name, dtype, logdir, extra=()):
  sys.path.insert(0, REPO)
  from openseq2seq_amd.utils.utils import create_model, get_base_config
  args, base, model_cls, mod = get_base_config(
      ["--config_file=" + CFG % name, "--mode=train_eval", "--logdir=" + logdir] + list(extra))
  if dtype is not None:
    base["dtype"] = dtype
  base["random_seed"] = 0
  return args, create_model(args, base, mod, model_cls, None)


def _train_and_measure(name, dtype, logdir, extra=()):
  import run
  from openseq2seq_amd.parts.cnns.conv_blocks import Tape
  args, model = _models(name, dtype, logdir, extra)
  run.train(model, args)
  # the train graph's loss on one more batch with the trained weights (speech2text_test.py:45: no update)
  dl = model.get_data_layer()
  batch = next(dl.iterate_batches(model._device, seed=99))
  loss = float(model._forward_backward(batch, Tape()).cpu()[0])
  res = run.run_eval(model, model.eval_model, 0)
  return model, loss, res


@pytest.mark.parametrize("dtype", [None, "mixed"])         # None: the config's own tf.float32
@pytest.mark.parametrize("name", ["ds2", "w2l"])
def test_convergence(cuda, tmp_path, monkeypatch, name, dtype):
  """speech2text_ds2_test.py:29 / speech2text_w2l_test.py:24: convergence_test(5.0, 30.0, 0.1)."""
  monkeypatch.chdir(REPO)
  model, loss, res = _train_and_measure(name, dtype, str(tmp_path / "log"))
  print(name, dtype, "train loss %.3f" % loss, res)
  assert loss < 5.0
  assert res["Eval loss"] < 30.0
  assert res["Eval WER"] < 0.1


@pytest.mark.parametrize("name,n_vars,n_master", [("ds2", 14, 7), ("w2l", 14, 6)])
def test_mp_collection(cuda, tmp_path, monkeypatch, name, n_vars, n_master):
  """mp_collection_test: len(tf.trainable_variables()) and len(FP32_MASTER_COPIES) of the mixed-precision
  graph (excluded: BatchNorm beta / gamma and the row convolution), read off the checkpoint this model writes
  under the reference's variable names. The recurrent layer is ONE variable in the reference (the opaque
  CudnnGRU parameter blob); here it is stored per direction as wx / wh / bias / bias_h and counted once."""
  monkeypatch.chdir(REPO)
  from openseq2seq_amd.utils import checkpoint
  _, model = _models(name, "mixed", str(tmp_path / "log"))
  names = checkpoint.model_variables(model)
  masters = {n[len(checkpoint.MASTER_PREFIX):] for n in names if n.startswith(checkpoint.MASTER_PREFIX)}
  state = set(model.store.state.keys())
  trainable = {n for n in names if not n.startswith(checkpoint.MASTER_PREFIX) and n not in state}
  fold = lambda ns: {re.sub(r"/cudnn_(gru|lstm)/.*", r"/cudnn_\1", n) for n in ns}
  assert len(fold(trainable)) == n_vars, sorted(fold(trainable))
  assert len(fold(masters)) == n_master, sorted(fold(masters))
  assert not any("/bn/" in n or "/row_conv/" in n for n in masters)


def test_ds2_infer(cuda, tmp_path, monkeypatch):
  """infer_test (speech2text_test.py:171-213): 250 epochs of training, inference with batch size 4 from the
  latest checkpoint; the csv names the files in dataset order and every transcript is within 5 characters."""
  monkeypatch.chdir(REPO)
  import run
  from openseq2seq_amd.models.speech2text import levenshtein
  from openseq2seq_amd.utils.utils import create_model, get_base_config
  logdir = str(tmp_path / "log")
  args, model = _models("ds2", None, logdir, ["--num_epochs=250"])
  run.train(model, args)
  out = str(tmp_path / "infer_out.csv")
  args, base, model_cls, mod = get_base_config(
      ["--config_file=" + CFG % "ds2", "--mode=infer", "--logdir=" + logdir, "--batch_size_per_gpu=4",
       "--infer_output_file=" + out])
  # the reference's test feeds the eval data layer parameters to the infer model (prepare_config)
  mod = dict(mod, infer_params=copy.deepcopy(mod["eval_params"]))
  imodel = create_model(args, base, mod, model_cls, None)
  run.restore_latest(imodel, 0)
  run.infer(imodel, args, 0)
  pred = list(csv.reader(open(out)))[1:]
  true = list(csv.reader(open("open_seq2seq/test_utils/toy_speech_data/toy_data.csv")))[1:]
  assert len(pred) == len(true) == 10
  for p, t in zip(pred, true):
    assert p[0] == t[0]
    assert levenshtein(p[-1], t[-1]) < 5, (p[-1], t[-1])


def test_block_dropout_cli_runs(cuda, tmp_path):
  """scripts/run_all_tests.sh:71-83, the four commands as written there (plus a log directory and a shorter
  schedule: 60 epochs instead of 500 — the reference only checks that the runs complete)."""
  cfg = CFG % "jasper_res_blockout"
  logdir = str(tmp_path / "log")

  def run_py(*extra):
    r = subprocess.run([sys.executable, "run.py", "--config_file=" + cfg, "--logdir=" + logdir] + list(extra),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=REPO, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    return r.stdout

  out = run_py("--mode=train_eval", "--encoder_params/drop_block_prob=0.98", "--num_epochs=60")
  assert "Saved checkpoint" in out and "Eval WER" in out
  for idx in (0, 1, 2):
    out = run_py("--mode=eval", "--encoder_params/drop_block_index=%d" % idx)
    assert "Eval WER" in out

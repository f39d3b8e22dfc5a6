"""Convergence test on the toy reversal corpus with the en-de-nmt-small architecture
(BASELINE.json configs[0]) — the reference's own acceptance test for its RNN NMT path is of
this kind (toy reversal task, BLEU > 0.9: SURVEY.md 4 / 8c). 1000 steps of the full config with a
fixed random_seed (the config leaves it to the clock, as the reference's does: 400 steps reach BLEU
0.89-0.95 and 600 steps 0.86-0.97 depending on the initialisation draw — one in-suite run of round 5
drew 0.86)
through run.py's train loop, then greedy decoding of the dev set."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _deterministic_kernels():
  """Convergence tests run with the single-contributor launch geometries (os2s_set_deterministic: no fp32 atomics
  across workgroups — embedding scatter, narrow weight gradients): a seed's trajectory is then the same on every run
  and every box, so a seed that clears its bar clears it always. (With the default kernels the same seed drew BLEU
  0.971 ... 1.0 over nine runs and failed once in a full-suite run of round 6.)"""
  from openseq2seq_amd import capi
  prev = capi.deterministic()
  capi.set_deterministic(True)
  try:
    yield
  finally:
    capi.set_deterministic(prev)


@pytest.mark.parametrize("seed", [7, 11, 2024])
def test_nmt_small_learns_reversal(cuda, tmp_path, monkeypatch, seed):
  """Three seeds, each must clear the bar (round 5 drew BLEU 0.86 once under a clock-derived seed: a convergence
  test that passes on one lucky seed is a flake on the next box)."""
  sys.path.insert(0, REPO)
  import run
  from openseq2seq_amd.test_utils.create_reversed_examples import create_data
  from openseq2seq_amd.utils.utils import create_model, get_base_config
  monkeypatch.chdir(tmp_path)
  create_data(train_corpus_size=10000, dev_corpus_size=256, test_corpus_size=8,
              data_path="toy_text_data", seed=0)
  cfg = os.path.join(REPO, "example_configs/text2text/toy-reversal/nmt-small-reversal.py")
  args, base_config, base_model, config_module = get_base_config(
      ["--config_file=" + cfg, "--mode=train_eval", "--max_steps=1000", "--print_loss_steps=100"])
  base_config["random_seed"] = seed     # (command-line overrides exist only for keys the config file has)
  model = create_model(args, base_config, config_module, base_model, None)
  run.train(model, args)
  res = run.run_eval(model, model.eval_model, 0)
  print("nmt-small reversal: %r" % (res,))
  assert res["samples"] == 256
  assert res["bleu"] > 0.9, res
  # infer mode of the same config = BeamSearchRNNDecoderWithAttention (beam 10, GNMT length
  # penalty) restored from the checkpoint the training loop wrote; output file vs reversed sources
  args, base_config, base_model, config_module = get_base_config(
      ["--config_file=" + cfg, "--mode=infer", "--infer_output_file=out.txt"])
  imodel = create_model(args, base_config, config_module, base_model, None)
  run.restore_latest(imodel, 0)
  run.infer(imodel, args, 0)
  src = [l.split() for l in open("toy_text_data/test/source.txt").read().strip().splitlines()]
  hyp = [l.split() for l in open("out.txt").read().strip().splitlines()]
  assert len(hyp) == len(src) == 8
  assert sum(h == list(reversed(s_)) for h, s_ in zip(hyp, src)) >= 6, (hyp, src)


def test_transformer_learns_reversal_with_beam_search(cuda, tmp_path, monkeypatch):
  """The Transformer counterpart (the reference's toy-reversal/nmt-reversal-TT.py at d_model 512):
  training path + beam-search inference (beam 5, alpha 1.0) end to end; BLEU on the dev set."""
  sys.path.insert(0, REPO)
  import run
  from openseq2seq_amd.test_utils.create_reversed_examples import create_data
  from openseq2seq_amd.utils.utils import create_model, get_base_config
  monkeypatch.chdir(tmp_path)
  create_data(train_corpus_size=10000, dev_corpus_size=256, test_corpus_size=8,
              data_path="toy_text_data", seed=0)
  cfg = os.path.join(REPO, "example_configs/text2text/toy-reversal/transformer-reversal-512.py")
  args, base_config, base_model, config_module = get_base_config(
      ["--config_file=" + cfg, "--mode=train_eval", "--max_steps=800", "--print_loss_steps=200",
       "--eval_steps=10000"])
  model = create_model(args, base_config, config_module, base_model, None)
  run.train(model, args)
  res = run.run_eval(model, model.eval_model, 0)
  assert res["samples"] == 256
  assert res["bleu"] > 0.9, res


def test_luong_rr_config_learns_reversal(cuda, tmp_path, monkeypatch):
  """The reference's own RNN acceptance configuration (toy-reversal/nmt-reversal-RR.py:
  LSTM-128, Luong attention, Adam + gradient clipping): greedy BLEU after 800 steps, then beam
  search (infer mode) from the checkpoint."""
  sys.path.insert(0, REPO)
  import run
  from openseq2seq_amd.test_utils.create_reversed_examples import create_data
  from openseq2seq_amd.utils.utils import create_model, get_base_config
  monkeypatch.chdir(tmp_path)
  create_data(train_corpus_size=10000, dev_corpus_size=256, test_corpus_size=8,
              data_path="toy_text_data", seed=0)
  cfg = os.path.join(REPO, "example_configs/text2text/toy-reversal/nmt-reversal-RR.py")
  args, base_config, base_model, config_module = get_base_config(
      ["--config_file=" + cfg, "--mode=train_eval", "--max_steps=1000", "--print_loss_steps=200",
       "--eval_steps=10000"])
  model = create_model(args, base_config, config_module, base_model, None)
  run.train(model, args)
  res = run.run_eval(model, model.eval_model, 0)
  assert res["samples"] == 256 and res["bleu"] > 0.9, res
  args, base_config, base_model, config_module = get_base_config(
      ["--config_file=" + cfg, "--mode=infer", "--infer_output_file=out.txt"])
  imodel = create_model(args, base_config, config_module, base_model, None)
  run.restore_latest(imodel, 0)
  run.infer(imodel, args, 0)
  src = [l.split() for l in open("toy_text_data/test/source.txt").read().strip().splitlines()]
  hyp = [l.split() for l in open("out.txt").read().strip().splitlines()]
  assert sum(h == list(reversed(s_)) for h, s_ in zip(hyp, src)) >= 6, (hyp, src)

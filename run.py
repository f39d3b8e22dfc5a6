#!/usr/bin/env python
"""run.py — same entry point and flags as the reference's run.py (run.py:18-101):
`python run.py --config_file=<cfg> --mode=train [--benchmark --bench_steps N
--bench_start K] [--<nested/param>=value ...]`; under torchrun (one process per GPU) the
data-parallel path runs over RCCL. The reference's own example_configs load unchanged.
Implemented modes: train (incl. --benchmark; synthetic batches when dataset files are
absent). The hot loop mirrors utils/funcs.py:172-218 (objects/sec accounting)."""
from __future__ import print_function

import sys
import time

import torch

from openseq2seq_amd.utils import distributed as dist_utils
from openseq2seq_amd.utils.utils import create_model, deco_print, get_base_config


def train(model, args):
  """utils/funcs.py:22-220 reduced to the hot loop + benchmark timing."""
  p = model.params
  max_steps = p.get('max_steps', 100)
  bench_start = p.get('bench_start', 10)
  dl = model.get_data_layer()
  rank = model.hvd.rank() if model.hvd else 0
  batches = None
  if hasattr(dl, "has_files") and dl.has_files():   # real line files (e.g. the toy reversal corpus)
    batches = dl.iterate_batches(model._device, seed=1234 + rank)
  else:
    batch = dl.synthetic_batch(model._device, seed=1234 + rank)
  eval_model = getattr(model, "eval_model", None)
  eval_steps = p.get('eval_steps', None)
  total_time, total_objects = 0.0, 0.0
  for step in range(max_steps):
    if batches is not None:
      batch = next(batches)
    if eval_model is not None and eval_steps and step > 0 and step % eval_steps == 0:
      run_eval(model, eval_model, rank)
    torch.cuda.synchronize()
    t0 = time.time()
    loss = model.train_step(batch)
    torch.cuda.synchronize()
    dt = time.time() - t0
    if step >= bench_start:
      total_time += dt
      total_objects += float(model._get_num_objects_per_step(batch))
    ps = p.get('print_loss_steps', None)
    if ps and step % ps == 0 and rank == 0:
      deco_print("step %d loss %.4f time per step %.3fs" % (step, float(loss.cpu()[0]), dt))
  if model.hvd and model.hvd.size() > 1:
    t = torch.tensor([total_objects], dtype=torch.float64, device=model._device)
    torch.distributed.all_reduce(t)
    total_objects = float(t.item())
  if rank == 0 and total_time > 0:
    n = max(max_steps - bench_start, 1)
    deco_print("Finished training")
    deco_print("Avg time per step: {:.3f}s".format(total_time / n))
    deco_print("Avg objects per second: {:.3f}".format(total_objects / total_time))


def run_eval(model, eval_model, rank):
  eval_model.copy_weights_from(model)
  res = eval_model.evaluate()
  if rank == 0:
    deco_print("Validation: Eval BLUE score: %.4f  exact match: %.4f  (%d samples)"
               % (res["bleu"], res["exact_match"], res["samples"]))
  return res


def main():
  args, base_config, base_model, config_module = get_base_config(sys.argv[1:])
  hvd = dist_utils.init_from_env() if base_config.get('use_horovod', False) else None
  if args.mode not in ("train", "train_eval"):
    raise NotImplementedError("mode %s: only the training path is re-hosted so far" % args.mode)
  model = create_model(args, base_config, config_module, base_model, hvd)
  train(model, args)
  if getattr(model, "eval_model", None) is not None:
    run_eval(model, model.eval_model, model.hvd.rank() if model.hvd else 0)


if __name__ == '__main__':
  main()

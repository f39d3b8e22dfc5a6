#!/usr/bin/env python
"""run.py — same entry point and flags as the reference's run.py (run.py:18-101):
`python run.py --config_file=<cfg> --mode=train [--benchmark --bench_steps N
--bench_start K] [--<nested/param>=value ...]`; under torchrun (one process per GPU) the
data-parallel path runs over RCCL. The reference's own example_configs load unchanged.
Modes: train / train_eval (incl. --benchmark; synthetic batches when the dataset files are
absent), eval and infer (from the latest checkpoint in logdir). The hot loop mirrors
utils/funcs.py:172-218 (objects/sec accounting); checkpoints are written under the reference's
variable names (openseq2seq_amd/utils/checkpoint.py)."""
from __future__ import print_function

import os
import sys
import time

import torch

from openseq2seq_amd.utils import checkpoint
from openseq2seq_amd.utils import distributed as dist_utils
from openseq2seq_amd.utils.utils import create_model, deco_print, get_base_config


def train(model, args):
  """utils/funcs.py:22-220 reduced to the hot loop + benchmark timing."""
  p = model.params
  # models/model.py:346-365 / utils/funcs.py:45: stop at max_steps, else num_epochs * steps_in_epoch
  # (100 steps when neither can be known: synthetic batches without dataset files)
  max_steps = model._last_step(default=100)
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
  logdir = p.get('logdir', None)
  save_steps = p.get('save_checkpoint_steps', None)
  first_step = 0
  if logdir and (args.continue_learning or p.get('load_model')):
    prefix = checkpoint.latest_checkpoint(p.get('load_model') or logdir)
    if prefix is not None:
      checkpoint.load(model, prefix, restore_optimizer=args.continue_learning)
      if args.continue_learning:
        first_step = checkpoint.read_step(prefix)
      if rank == 0:
        deco_print("Restored checkpoint %s (step %d)" % (prefix, first_step))
  total_time, total_objects = 0.0, 0.0
  for step in range(first_step, max_steps):
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
    if logdir and save_steps and rank == 0 and step > 0 and step % save_steps == 0:
      checkpoint.save(model, logdir, step)
  if logdir and rank == 0 and not args.benchmark:
    deco_print("Saved checkpoint %s" % checkpoint.save(model, logdir, max_steps))
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
  if model is not None:
    eval_model.copy_weights_from(model)
  res = eval_model.evaluate()
  if rank == 0:
    if "bleu" in res:
      deco_print("Validation: Eval BLUE score: %.4f  exact match: %.4f  (%d samples)"
                 % (res["bleu"], res["exact_match"], res["samples"]))
    else:
      deco_print("Validation: " + "  ".join("%s: %s" % kv for kv in sorted(res.items())))
  return res


def restore_latest(model, rank):
  logdir = model.params.get('load_model') or model.params.get('logdir')
  prefix = checkpoint.latest_checkpoint(logdir) if logdir else None
  if prefix is None:
    raise IOError("no checkpoint found in %r (eval / infer need a trained model)" % (logdir,))
  checkpoint.load(model, prefix, restore_optimizer=False)
  if rank == 0:
    deco_print("Restored checkpoint %s" % prefix)


def infer(model, args, rank):
  """utils/funcs.py:223-290: run the infer data layer once, hand the per-batch results to
  finalize_inference."""
  dl = model.get_data_layer()
  results = [model.infer_batch(b) for b in dl.iterate_batches(model._device, drop_remainder=False)]
  if rank == 0:
    model.finalize_inference(results, args.infer_output_file)
    deco_print("Finished inference: %d batches -> %s" % (len(results), args.infer_output_file))


def launch_dry_run(base_config, hvd):
  """OS2S_LAUNCH_DRY_RUN=1: stop after the process group exists and report what the job would
  train on (no model, no GPU work) — the check that a tower config gets all its replicas."""
  import json
  world = hvd.size() if hvd else 1
  seen = torch.ones(1)
  if world > 1:
    torch.distributed.all_reduce(seen)
  if (hvd.rank() if hvd else 0) == 0:
    print(json.dumps({"world_size": world, "ranks_seen": int(seen.item()),
                      "use_horovod": bool(base_config.get('use_horovod', False)),
                      "batch_size_per_gpu": base_config['batch_size_per_gpu'],
                      "global_batch": base_config['batch_size_per_gpu'] * world,
                      "backend": torch.distributed.get_backend() if world > 1 else "none"}))
  if world > 1:
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def main():
  args, base_config, base_model, config_module = get_base_config(sys.argv[1:])
  # Horovod mode: the ranks come from the launcher. Tower mode (`use_horovod False, num_gpus N`,
  # models/model.py:386-427): N replicas = N ranks, started here when no launcher did
  world, spawn = dist_utils.plan_workers(base_config)
  if spawn:
    deco_print("num_gpus=%d without Horovod: starting %d ranks (one process per GPU, RCCL)"
               % (world, world))
    sys.exit(dist_utils.spawn_ranks(world, __file__, sys.argv[1:]))
  hvd = dist_utils.init_from_env() if world > 1 else None
  if os.environ.get("OS2S_LAUNCH_DRY_RUN", "") == "1":
    launch_dry_run(base_config, hvd)
    return
  model = create_model(args, base_config, config_module, base_model, hvd)
  rank = hvd.rank() if hvd else 0
  if args.mode == "eval":
    restore_latest(model, rank)
    run_eval(None, model, rank)
    return
  if args.mode == "infer":
    restore_latest(model, rank)
    infer(model, args, rank)
    return
  if args.mode not in ("train", "train_eval"):
    raise NotImplementedError("mode %s" % args.mode)
  train(model, args)
  if getattr(model, "eval_model", None) is not None:
    run_eval(model, model.eval_model, model.hvd.rank() if model.hvd else 0)


if __name__ == '__main__':
  main()

#!/usr/bin/env python
"""Times Transformer-big beam-search inference (beam 4, alpha 0.6, extra_decode_length 50;
transformer-big.py decoder_params) on synthetic sentences with random-init weights:
ms per decode step and decoded positions/sec. Usage: python tools/bench_transformer_infer.py
[--batch 64] [--reps 3] [--poll 8]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--batch", type=int, default=64)
  ap.add_argument("--reps", type=int, default=3)
  ap.add_argument("--small", action="store_true")
  args = ap.parse_args()
  from openseq2seq_amd.configs.transformer import transformer_config
  dev = torch.device("cuda:0")
  kw = dict(d_model=512, num_layers=2, num_heads=8, vocab_size=4096) if args.small else {}
  model_cls, params = transformer_config(batch_size_per_gpu=args.batch, **kw)
  model = model_cls(params, mode="infer", hvd=None, device=dev)
  model.compile()
  batch = model.get_data_layer().synthetic_batch(dev, seed=7)
  S = int(batch['source_tensors'][0].shape[1])
  for r in range(args.reps + 1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ids, lens = model.infer_batch(batch)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = int(ids.shape[1])
    print("rep %d: %.1f ms total, %d steps (src len %d), %.3f ms/step, %.0f beam-positions/s"
          % (r, dt * 1e3, steps, S, dt * 1e3 / steps, args.batch * 4 * steps / dt), flush=True)


if __name__ == "__main__":
  main()

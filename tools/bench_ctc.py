"""CTC loss (log-softmax + alpha/beta + gradient) on its own at the Jasper bench shape (T = 835 frames
after the stride-2 layer, B = 32, V = 29, labels up to 250 characters): microseconds per call.
OS2S_CTC_WAVE=0 selects the 256-thread alpha/beta kernel (a workgroup barrier per frame) instead of
the one-wave-per-(sample, direction) kernel.

  for w in 0 1; do OS2S_CTC_WAVE=$w python tools/bench_ctc.py; done
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
  from openseq2seq_amd import capi
  dev = torch.device("cuda:0")
  rng = np.random.RandomState(0)
  T, B, V, L = 835, 32, 29, 250
  logits = torch.from_numpy(rng.randn(T, B, V).astype(np.float32)).to(dev)
  in_len = torch.from_numpy(rng.randint(100, T + 1, size=B).astype(np.int32))
  in_len[0] = T
  label_len = torch.minimum(torch.from_numpy(rng.randint(10, L + 1, size=B).astype(np.int32)), in_len // 3)
  labels = torch.from_numpy(rng.randint(0, V - 1, size=(B, L)).astype(np.int32)).to(dev)
  in_len, label_len = in_len.to(dev), label_len.to(dev)
  fn = lambda: capi.ctc_loss(logits, in_len, labels, label_len, want_grad=False, want_grad_bf16=True)
  for _ in range(3):
    out = fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(20):
    fn()
  e1.record()
  torch.cuda.synchronize()
  print("OS2S_CTC_WAVE=%s  ctc_loss %.1f us per call  (loss %.4f)" %
        (os.environ.get("OS2S_CTC_WAVE", "1"), e0.elapsed_time(e1) * 1e3 / 20, float(out["loss_mean"][0])))


if __name__ == "__main__":
  main()

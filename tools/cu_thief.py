"""CU-thief experiment: how does the Jasper train step degrade when N compute units
are held by another resident kernel — what the RCCL all-reduce of an 8-GPU run does to the 216-256-tile
convolution launches — measured on ONE GPU.

  python tools/cu_thief.py [--steps 12] [--thieves 0,8,16,32,48,64]

Per setting: the thief (tools/cu_thief.hip: N workgroups x 256 threads x 96 KB LDS — one per CU, placement read back
from HW_ID — spinning on the 100 MHz wall clock) is launched on its own stream at the start of every backward pass, for (a) the whole backward
pass ("full": the worst case, a collective that never ends) and (b) 10 bursts of 0.45 ms = what ten 128 MB
fp32 buckets need at ~300 GB/s bus bandwidth on 8 GPUs ("bursts"). Prints one JSON line with ms/step per
setting; the table goes into DESIGN.md section 4."""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import torch  # noqa: E402


def build():
  so = os.path.join(HERE, "libcu_thief.so")
  src = os.path.join(HERE, "cu_thief.hip")
  if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", src, "-o", so])
  lib = ctypes.CDLL(so)
  lib.cu_thief_launch.restype = ctypes.c_int
  lib.cu_thief_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
  return lib


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--steps", type=int, default=12)
  ap.add_argument("--warmup", type=int, default=4)
  ap.add_argument("--thieves", default="0,16,32,48,64,96")
  ap.add_argument("--model", default="jasper", choices=["jasper", "ds2", "transformer"],
                  help="round 5: the same experiment on DeepSpeech2-large (whose persistent GRU launches need 32 "
                       "co-resident workgroups per XCD: a held CU makes them give up and the step is redone on the "
                       "launch-per-step kernels) and on Transformer-big")
  args = ap.parse_args()
  lib = build()
  from openseq2seq_amd import capi
  from openseq2seq_amd.parts.cnns import conv_blocks
  dev = torch.device("cuda", 0)
  torch.cuda.set_device(dev)
  if args.model == "jasper":
    from openseq2seq_amd.configs.jasper import jasper10x5_config
    model_cls, params = jasper10x5_config(batch_size_per_gpu=32, use_horovod=True)
  elif args.model == "ds2":
    from openseq2seq_amd.configs.ds2 import ds2_large_config
    model_cls, params = ds2_large_config()
  else:
    from openseq2seq_amd.configs.transformer import transformer_config
    model_cls, params = transformer_config()
  model = model_cls(params, mode="train", hvd=None, device=dev)
  model.compile()
  batch = model.get_data_layer().synthetic_batch(dev, seed=1234)
  sink = torch.zeros(4, dtype=torch.int32, device=dev)
  where = torch.full((256,), -1, dtype=torch.int32, device=dev)
  thief_stream = torch.cuda.Stream(device=dev)
  state = {"n": 0, "mode": "off", "bwd_us": 25000.0}
  orig_backward = conv_blocks.Tape.backward

  def backward(self):
    n, mode = state["n"], state["mode"]
    if n > 0:
      thief_stream.wait_stream(torch.cuda.current_stream())     # starts when backward starts
      with torch.cuda.stream(thief_stream):
        if mode == "full":
          lib.cu_thief_launch(ctypes.c_void_p(thief_stream.cuda_stream), n, state["bwd_us"], ctypes.c_void_p(sink.data_ptr()),
                              ctypes.c_void_p(where.data_ptr()))
        else:
          gap = max(state["bwd_us"] / 10.0 - 450.0, 0.0)
          for _ in range(10):
            lib.cu_thief_launch(ctypes.c_void_p(thief_stream.cuda_stream), n, 450.0, ctypes.c_void_p(sink.data_ptr()), None)
            lib.cu_thief_launch(ctypes.c_void_p(thief_stream.cuda_stream), 1, gap, ctypes.c_void_p(sink.data_ptr()), None)
    return orig_backward(self)
  conv_blocks.Tape.backward = backward

  fell_back = {}

  def run(n, mode):
    state["n"], state["mode"] = n, mode
    capi.gru_xcd_set_mode(-1)       # a setting that made the persistent GRU give up must not decide the next one
    import warnings
    with warnings.catch_warnings(record=True) as caught:
      warnings.simplefilter("always")
      for _ in range(args.warmup):
        model.train_step(batch)
    if args.model == "ds2":
      fell_back["%s/%d" % (mode, n)] = any("persistent GRU" in str(w.message) for w in caught)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
      model.train_step(batch)
    torch.cuda.synchronize()
    return 1000.0 * (time.perf_counter() - t0) / args.steps

  base = run(0, "off")
  state["bwd_us"] = 1000.0 * base * 0.62          # backward is ~62 % of the step (trace_phases)
  out = {"model": args.model, "baseline_ms": base, "backward_us_assumed": state["bwd_us"], "full": {}, "bursts": {}, "distinct_cus_held": {}}
  for n in [int(v) for v in args.thieves.split(",") if int(v) > 0]:
    where.fill_(-1)
    out["full"][n] = run(n, "full")
    out["distinct_cus_held"][n] = len(set(where[:n].cpu().tolist()))
    out["bursts"][n] = run(n, "bursts")
  out["baseline_again_ms"] = run(0, "off")
  if args.model == "ds2":
    out["gru_fell_back_to_step_launches"] = fell_back
  print(json.dumps(out))


if __name__ == "__main__":
  main()

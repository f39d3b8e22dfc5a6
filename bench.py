#!/usr/bin/env python
"""bench.py — the reference's headline measurement on MI355X.

`python bench.py --gpus N --steps K --warmup W` times K full training steps
(forward + backward + all-reduce + mixed-precision NovoGrad/LARC update) of
Jasper 10x5 Dense-Residual (BASELINE.json configs[1]) on synthetic 16 kHz-shaped
feature batches already resident in HBM, one process per GPU (RCCL through
torch.distributed under torchrun), and prints ONE JSON line on rank 0.

metric  = what `run.py --benchmark` prints as "Avg objects per second"
          (open_seq2seq/utils/funcs.py:192-218): sum over ranks of INPUT feature
          frames (sum of src_length, models/speech2text.py:356-360) / wall time.
roofline = the dominant kernel (implicit-GEMM conv1d on the matrix cores, used by
          forward and data-gradient): algorithmic FLOPs of its launches / their
          measured durations (HIP events on the launch stream, live in the timed
          region) vs the dense bf16 MFMA peak.
cpu_baseline = the CPU oracle (fp32 restatement of the reference's path, torch-CPU
          + NumPy NovoGrad) timed on the host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
  sys.path.insert(0, REPO)

# the host driver only supports dmabuf IPC (RCCL / cross-process tensors need it)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

BF16_DENSE_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA peak
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def parse():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=20)
  ap.add_argument("--warmup", type=int, default=5)
  ap.add_argument("--batch", type=int, default=32, help="batch_size_per_gpu (reference: 32)")
  ap.add_argument("--fixed-frames", type=int, default=0,
                  help="0: durations U[2,16.7]s (default workload); >0: every utterance has "
                       "this many frames (1680 = worst-case fixed shape)")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-transformer", action="store_true",
                  help="skip the secondary Transformer-big tokens/sec measurement")
  ap.add_argument("--transformer-batch", type=int, default=256)
  ap.add_argument("--only-transformer-infer", action="store_true",
                  help="Transformer-big beam-search inference only: decoded positions/sec")
  ap.add_argument("--only-nmt", action="store_true", help="en-de-nmt-small train step only: tokens/sec")
  ap.add_argument("--no-other-configs", action="store_true",
                  help="skip the short measurements of the other BASELINE configs (N=1 only)")
  ap.add_argument("--only-quartznet", action="store_true",
                  help="QuartzNet 15x5 (separable convolutions) train step only: frames/sec")
  ap.add_argument("--only-tacotron", action="store_true",
                  help="Tacotron2 (tacotron_gst.py shapes) train step only: mel frames/sec")
  ap.add_argument("--only-tacotron-decode", action="store_true",
                  help="Tacotron2-GST free-running decode (BASELINE configs[4]) only: mel frames/sec, us per step")
  ap.add_argument("--decode-steps", type=int, default=1000)
  ap.add_argument("--no-style", action="store_true", help="Tacotron2 without the GST style encoder")
  ap.add_argument("--no-fp8", action="store_true",
                  help="Tacotron2 with bf16 decoder weights (default: e4m3 copies, BASELINE configs[4])")
  ap.add_argument("--only-ds2", action="store_true",
                  help="run only the DeepSpeech2-large train step (BASELINE configs[2])")
  ap.add_argument("--only-frontend", action="store_true", help="profiling aid: the log-mel front end only")
  ap.add_argument("--cpu-nmt-leg", default=None, help=argparse.SUPPRESS)     # child process of cpu_baseline_nmt_guarded
  ap.add_argument("--only-transformer", action="store_true",
                  help="profiling aid: run only the Transformer-big measurement")
  ap.add_argument("--no-kernel-timing", action="store_true")
  ap.add_argument("--pp-cost", default=None,
                  help="A/B aid: 'c256,c2x128,c3x128[,dgrad_penalty[,prio]]' (os2s_set_option conv1d.pp_*) = microseconds per step of the three ping-pong convolution "
                       "tiles in the device-side tile choice (os2s_conv1d_set_pp_cost; 1e6 removes a tile)")
  ap.add_argument("--no-host-lens", action="store_true",
                  help="A/B: drop the batch's host copy of the sequence lengths (the convolution launcher then "
                       "enqueues both ping-pong kernels and the device chooses)")
  ap.add_argument("--one-rank-group", action="store_true",
                  help="N=1 only: run the data-parallel path (RCCL process group, bucketed all-reduce on "
                       "the side stream, comm diagnostics) on a one-rank group")
  ap.add_argument("--set-option", action="append", default=[], metavar="NAME=VALUE",
                  help="os2s_set_option(NAME, VALUE) before anything runs (A/B runs of a library knob, e.g. "
                       "conv1d_wgrad.xcd_order=0); repeatable")
  ap.add_argument("--bucket-mb", type=float, default=None,
                  help="size of the gradient all-reduce buckets in MB of fp32 (default 128; OS2S_BUCKET_MB)")
  ap.add_argument("--allreduce-dtype", choices=["fp32", "bf16"], default=None,
                  help="what crosses the wire in the gradient all-reduce (default fp32; OS2S_ALLREDUCE_DTYPE)")
  ap.add_argument("--backend", choices=["nccl", "gloo"], default=None,
                  help="process-group backend for --gpus N (default: nccl = RCCL)")
  ap.add_argument("--one-device", action="store_true",
                  help="rehearsal of --gpus N on a one-GPU box: every rank runs on cuda:0 over gloo (the numbers "
                       "describe nothing; the line, the comm block and the collectives are the real ones)")
  ap.add_argument("--launcher-dry-run", action="store_true",
                  help="exercise only the multi-rank launcher and the timing collectives (no GPU "
                       "work: runs on CPU over gloo); prints the JSON line with value null")
  return ap.parse_args()


_JSON_FD = None


def claim_stdout():
  """The contract is ONE JSON line on stdout. RCCL and the ROCm runtime print through C stdio ('Librccl path : ...'
  on the first communicator), and that buffer is flushed at process exit — AFTER Python's own print, on every
  rank. So each rank keeps a private duplicate of the original stdout for the JSON line and points fd 1 at
  stderr for everything else (C and Python alike)."""
  global _JSON_FD
  if _JSON_FD is None:
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)


def emit(obj):
  sys.stdout.flush()
  data = (json.dumps(obj) + "\n").encode()
  fd = _JSON_FD if _JSON_FD is not None else 1
  while data:
    data = data[os.write(fd, data):]


def spawn_ranks(n):
  """`python bench.py --gpus N` without a launcher: re-execute this script as N ranks of ONE node
  (openseq2seq_amd/utils/distributed.py:spawn_ranks). Returns the launcher's exit code."""
  from openseq2seq_amd.utils.distributed import spawn_ranks as _spawn
  return _spawn(n, __file__, sys.argv[1:])


class ConvTimer(object):
  """Wraps capi.conv1d_fwd: HIP events around every launch + the FLOPs it EXECUTES: the dense
  2*B*Tout*Cin*Cout*K scaled by the fraction of 128-row time tiles that are not skipped as
  exact zeros (input window past in_len) or as never-read outputs (tile past out_len)."""

  def __init__(self, capi):
    self.capi = capi
    self.orig = capi.conv1d_fwd
    self.records = []
    self.enabled = False
    self.in_backward = False
    self.untimed = 0      # launches of the timed region that were not bracketed (see install)
    self.seen = 0
    self.every = int(os.environ.get("OS2S_BENCH_CONV_EVERY", "4"))

  def install(self):
    timer = self
    # launches issued inside Tape.backward share the GPU with the weight-gradient kernels of the
    # side stream (parts/cnns/conv_blocks.py): their event-bracketed time is not the kernel's own
    from openseq2seq_amd.parts.cnns import conv_blocks
    orig_backward = conv_blocks.Tape.backward
    self._orig_backward = orig_backward

    def backward(tape):
      timer.in_backward = True
      try:
        return orig_backward(tape)
      finally:
        timer.in_backward = False

    conv_blocks.Tape.backward = backward
    self.overlap = os.environ.get("OS2S_WGRAD_STREAM", "1") != "0"

    def wrapped(x, w, **kw):
      # launches that share the GPU with side-stream work are not the kernel alone: backward
      # (weight gradients) and the forward launches next to early residual branches
      if not timer.enabled or (timer.in_backward and timer.overlap) or conv_blocks.forward_side_busy():
        timer.untimed += int(timer.enabled)
        return timer.orig(x, w, **kw)
      # an event pair costs a few microseconds of dispatch bubble on the stream: bracket every
      # 4th eligible launch (109 forward launches per step -> the sample rotates over the layers)
      timer.seen += 1
      if timer.seen % timer.every:
        timer.untimed += 1
        return timer.orig(x, w, **kw)
      e0 = torch.cuda.Event(enable_timing=True)
      e1 = torch.cuda.Event(enable_timing=True)
      e0.record()
      out = timer.orig(x, w, **kw)
      e1.record()
      B, tin, Cin = x.shape
      K, Cout, _ = w.shape
      tout = out.shape[0] if kw.get("time_major") else out.shape[1]
      stride, dil = kw.get("stride", 1), kw.get("dil", 1)
      pl = kw.get("pad_left")
      if pl is None:
        pl = timer.capi.same_padding(tin, K, stride, dil)[1]
      timer.records.append((e0, e1, 2.0 * B * tout * Cin * Cout * K,
                            (kw.get("in_len"), kw.get("out_len"), tin, tout, stride, pl),
                            timer.in_backward and timer.overlap, (Cin, Cout, K, stride)))
      return out

    self.capi.conv1d_fwd = wrapped
    # modules that imported the symbol through `capi.` pick the wrapper up automatically

    # the dense-residual 1x1 branches of a block end go out as ONE grouped launch of the same tile
    # code: it is part of the dominant kernel family and is timed and counted with it
    orig_grouped = self.capi.conv1x1_fwd_grouped

    def wrapped_grouped(items, in_len=None, out_len=None, out_f32=False):
      if out_f32:        # weight-sized products of the dense-residual statistics: not an activation launch
        return orig_grouped(items, in_len=in_len, out_len=out_len, out_f32=True)
      if not timer.enabled or (timer.in_backward and timer.overlap) or conv_blocks.forward_side_busy():
        timer.untimed += int(timer.enabled)
        return orig_grouped(items, in_len=in_len, out_len=out_len)
      timer.seen += 1
      if timer.seen % timer.every:
        timer.untimed += 1
        return orig_grouped(items, in_len=in_len, out_len=out_len)
      e0 = torch.cuda.Event(enable_timing=True)
      e1 = torch.cuda.Event(enable_timing=True)
      e0.record()
      orig_grouped(items, in_len=in_len, out_len=out_len)
      e1.record()
      B, T, _ = items[0]["x"].shape
      fl = sum(2.0 * B * T * it["x"].shape[2] * it["w"].shape[1] for it in items)
      timer.records.append((e0, e1, fl, (in_len, out_len, T, T, 1, 0),
                            timer.in_backward and timer.overlap, (0, len(items), 1, 1)))

    self.capi.conv1x1_fwd_grouped = wrapped_grouped

    # the dense-residual sums and their data gradients as GEMMs over the concatenated block inputs
    # (parts/cnns/dense_residual.py): the same family, timed and counted like the grouped launches they replace
    orig_cat = self.capi.conv1x1_cat_fwd

    def wrapped_cat(x, w, y, in_len=None, out_len=None, bias=None, accumulate=False):
      if not timer.enabled or (timer.in_backward and timer.overlap) or conv_blocks.forward_side_busy():
        timer.untimed += int(timer.enabled)
        return orig_cat(x, w, y, in_len=in_len, out_len=out_len, bias=bias, accumulate=accumulate)
      timer.seen += 1
      if timer.seen % timer.every:
        timer.untimed += 1
        return orig_cat(x, w, y, in_len=in_len, out_len=out_len, bias=bias, accumulate=accumulate)
      e0 = torch.cuda.Event(enable_timing=True)
      e1 = torch.cuda.Event(enable_timing=True)
      e0.record()
      out = orig_cat(x, w, y, in_len=in_len, out_len=out_len, bias=bias, accumulate=accumulate)
      e1.record()
      B, T, Cin = x.shape
      timer.records.append((e0, e1, 2.0 * B * T * Cin * y.shape[2], (in_len, out_len, T, T, 1, 0),
                            timer.in_backward and timer.overlap, (0, 1, 1, 1)))
      return out

    self.capi.conv1x1_cat_fwd = wrapped_cat
    self._restore = (("conv1d_fwd", self.orig), ("conv1x1_fwd_grouped", orig_grouped), ("conv1x1_cat_fwd", orig_cat))

  def uninstall(self):
    """The entry points and Tape.backward as they were: the configurations measured after the headline model run
    without the wrappers (a Python call layer per launch is measurable in a step that ends with the Python thread)."""
    from openseq2seq_amd.parts.cnns import conv_blocks
    for name, fn in getattr(self, "_restore", ()):
      setattr(self.capi, name, fn)
    if getattr(self, "_orig_backward", None) is not None:
      conv_blocks.Tape.backward = self._orig_backward
    self._restore = ()

  @staticmethod
  def _live_fraction(geom, cache):
    in_len, out_len, tin, tout, stride, pl = geom
    if in_len is None and out_len is None:
      return 1.0
    ntile = -(-tout // 128)
    t0 = torch.arange(ntile)[None, :] * 128
    live = torch.ones((1, ntile), dtype=torch.bool)
    for ln, is_in in ((in_len, True), (out_len, False)):
      if ln is None:
        continue
      key = ln.data_ptr()
      if key not in cache:
        cache[key] = ln.cpu().to(torch.int64)
      l = cache[key][:, None]
      live = live & ((t0 * stride - pl < l.clamp(max=tin)) if is_in else (t0 < l))
    return float(live.float().mean())

  def summary(self):
    """(ms, executed flops, launches) of the launches that had the GPU to themselves (forward
    pass; everything when the side stream is off); the totals over ALL launches are kept in
    self.all_ms / all_flops / all_n."""
    cache = {}
    ms = fl = dense = 0.0
    n = 0
    self.all_ms = self.all_flops = 0.0
    self.all_n = len(self.records)
    for r in self.records:
      t = r[0].elapsed_time(r[1])
      f = r[2] * self._live_fraction(r[3], cache)
      self.all_ms += t
      self.all_flops += f
      if not r[4]:
        ms += t
        fl += f
        dense += r[2]
        n += 1
    self.dense_flops = dense
    # the same split by the kernel a launch dispatches to (the tile choice is a fixed function of the
    # shape, conv1d_igemm.hip: ping-pong kernel for stride 1, >= 320 output channels, Cin a multiple
    # of 64 and K long enough to spread the X prefetch — every K >= 11 layer of the Jasper configs;
    # the lockstep tiles — 256-channel layers, stride 2, K = 1, grouped 1x1 — otherwise)
    self.by_kernel = {}
    for r in self.records:
      if r[4]:
        continue
      cin, cout, k, stride = r[5]
      name = ("conv1d_pp_kernel / conv1d_ppn_kernel (ping-pong tiles: 2 x 256, 2 x 128, 3 x 128 by the cost model)"
              if (cin > 0 and stride == 1 and k >= 8 and cout >= 320 and cin % 64 == 0)
              else "conv1d_igemm_kernel + conv1d_igemm_grouped_kernel (lockstep tiles)")
      e = self.by_kernel.setdefault(name, [0, 0.0, 0.0])
      e[0] += 1
      e[1] += r[0].elapsed_time(r[1])
      e[2] += r[2] * self._live_fraction(r[3], cache)
    if os.environ.get("OS2S_BENCH_CONV_TABLE"):      # per-shape breakdown of the timed launches
      tab = {}
      for r in self.records:
        if r[4]:
          continue
        e = tab.setdefault(r[5], [0, 0.0, 0.0])
        e[0] += 1
        e[1] += r[0].elapsed_time(r[1])
        e[2] += r[2] * self._live_fraction(r[3], cache)
      for k in sorted(tab, key=lambda k: -tab[k][1]):
        c, t, f = tab[k]
        print("conv %4d->%4d K %2d s%d: %4d launches %8.3f ms total %7.1f us/launch %6.0f TF/s executed"
              % (k[0], k[1], k[2], k[3], c, t, 1000 * t / c, f / (t * 1e-3) / 1e12 if t > 0 else 0.0),
              file=sys.stderr)
    return ms, fl, n


class StepBreakdown(object):
  """Per-family kernel time of the REST of the train step, measured live but OUTSIDE the timed
  region: after the timed steps, `steps` more steps run with the weight-gradient side stream off
  (OS2S_WGRAD_STREAM=0: every kernel alone on the GPU) and a HIP-event pair around every launch of
  the conv weight-gradient kernels (executed FLOPs vs the MFMA peak), the BatchNorm kernels and the
  optimizer (algorithmic bytes vs the HBM peak). The event pairs cost a few microseconds each, so
  these figures are lower bounds of the kernels' own rates; profiles/ holds the rocprofv3 view."""

  def __init__(self, capi):
    self.capi = capi
    self.rec = {}       # family -> [launches, [event pairs], work]
    self.saved = {}

  def _bracket(self, family, work, call):
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    out = call()
    e1.record()
    r = self.rec.setdefault(family, [0, [], 0.0])
    r[0] += 1
    r[1].append((e0, e1))
    r[2] += work
    return out

  @staticmethod
  def _live(lens, T, cache, quantum=1, margin=0):
    if lens is None:
      return 1.0
    key = (lens.data_ptr(), T, quantum, margin)
    if key not in cache:
      l = (lens.cpu().to(torch.int64) + margin).clamp(max=T)
      l = ((l + quantum - 1) // quantum * quantum).clamp(max=T)
      cache[key] = float(l.sum()) / float(l.numel() * T)
    return cache[key]

  def install(self):
    capi, bd, cache = self.capi, self, {}
    for name in ("conv1d_wgrad", "conv1d_wgrad_grouped", "conv1x1_wgrad_grouped", "bn_act_fwd", "bn_act_bwd_reduce",
                 "bn_bwd_apply", "opt_step"):
      self.saved[name] = getattr(capi, name)
    o = self.saved

    def wgrad(x, dy, K, **kw):
      B, Tin, Cin = x.shape
      _, Tout, Cout = dy.shape
      fl = 2.0 * B * Tout * Cin * Cout * K * bd._live(kw.get("in_len"), Tin, cache, quantum=64)
      return bd._bracket("conv1d weight gradient (conv1d_wgrad_pp_kernel + lockstep conv1d_wgrad_kernel)",
                         fl, lambda: o["conv1d_wgrad"](x, dy, K, **kw))

    def wgrad_same_shape(items, K, **kw):      # several layers of one shape in one launch (round 6)
      B, Tin, Cin = items[0]["x"].shape
      _, Tout, Cout = items[0]["dy"].shape
      fl = 2.0 * B * Tout * Cin * Cout * K * len(items) * bd._live(kw.get("in_len"), Tin, cache, quantum=64)
      return bd._bracket("conv1d weight gradient (conv1d_wgrad_pp_kernel + lockstep conv1d_wgrad_kernel)",
                         fl, lambda: o["conv1d_wgrad_grouped"](items, K, **kw))

    def wgrad_grouped(items, in_len=None, **kw):
      B, T, _ = items[0]["x"].shape
      fl = sum(2.0 * B * T * it["x"].shape[2] * it["dy"].shape[2] for it in items) * \
          bd._live(in_len, T, cache, quantum=64)
      return bd._bracket("conv1d weight gradient, K = 1 residual branches grouped per block end "
                         "(conv1d_wgrad_grouped_kernel)", fl,
                         lambda: o["conv1x1_wgrad_grouped"](items, in_len=in_len, **kw))

    def bn_fwd(ys, scales, shifts, out, out_len, act, keep_prob, seed):
      B, T, C = out.shape
      by = B * T * C * 2.0 * (len(ys) * bd._live(out_len, T, cache) + 1.0)
      return bd._bracket("batchnorm", by, lambda: o["bn_act_fwd"](ys, scales, shifts, out, out_len, act,
                                                                  keep_prob, seed))

    def bn_red(dout, out, ys, means, rstds, dz, partial, out_len, act, keep_prob, seed):
      B, T, C = out.shape
      by = B * T * C * 2.0 * ((2 + len(ys)) * bd._live(out_len, T, cache) + 1.0)
      return bd._bracket("batchnorm", by, lambda: o["bn_act_bwd_reduce"](
          dout, out, ys, means, rstds, dz, partial, out_len, act, keep_prob, seed))

    def bn_apply(dz, y, gamma, mean, rstd, c1, c2, dy, out_len=None, margin=0, **kw):
      T = dz.shape[1] if dz.dim() == 3 else 1
      by = dz.numel() * 2.0 * (2.0 * bd._live(out_len, T, cache, margin=margin) + 1.0)
      return bd._bracket("batchnorm", by, lambda: o["bn_bwd_apply"](dz, y, gamma, mean, rstd, c1, c2, dy,
                                                                    out_len=out_len, margin=margin, **kw))

    def opt(cfg, state, grads, weights, *a, **kw):
      # per parameter: gradient + master + moment read (12 B), master + moment + bf16 copy written (10 B)
      return bd._bracket("optimizer", 22.0 * weights.numel(),
                         lambda: o["opt_step"](cfg, state, grads, weights, *a, **kw))

    capi.conv1d_wgrad, capi.bn_act_fwd, capi.bn_act_bwd_reduce = wgrad, bn_fwd, bn_red
    capi.conv1x1_wgrad_grouped = wgrad_grouped
    capi.conv1d_wgrad_grouped = wgrad_same_shape
    capi.bn_bwd_apply, capi.opt_step = bn_apply, opt

  def remove(self):
    for name, fn in self.saved.items():
      setattr(self.capi, name, fn)

  def run(self, model, batch, steps=2):
    prev = os.environ.get("OS2S_WGRAD_STREAM")
    os.environ["OS2S_WGRAD_STREAM"] = "0"
    self.install()
    try:
      for _ in range(steps):
        model.train_step(batch)
      torch.cuda.synchronize()
    finally:
      self.remove()
      if prev is None:
        os.environ.pop("OS2S_WGRAD_STREAM", None)
      else:
        os.environ["OS2S_WGRAD_STREAM"] = prev
    out = {}
    for fam, (n, evs, work) in self.rec.items():
      ms = sum(a.elapsed_time(b) for a, b in evs)
      hbm = not fam.startswith("conv1d")
      rate = work / (ms * 1e-3) / (1e9 if hbm else 1e12) if ms > 0 else 0.0
      peak = HBM_PEAK_GBS if hbm else BF16_DENSE_PEAK_TFLOPS
      out[fam] = {"bound": "hbm" if hbm else "mfma", "launches_per_step": n / float(steps),
                  "ms_per_step": ms / steps, "avg_launch_ms": ms / max(n, 1), "achieved": rate,
                  "unit": "GB/s" if hbm else "TFLOP/s", "peak": peak, "frac": rate / peak,
                  "measured": "serial pass after the timed region (side stream off), HIP events per launch"}
    return out


class FamilyBrackets(object):
  """HIP-event pairs around every call of some capi entry points during `steps` extra train steps with the
  side stream off (every kernel alone on the GPU): what the roofline blocks of the other configurations
  are computed from. spec: {capi function name: (family label, work(args, kwargs) -> number)}."""

  def __init__(self, capi, spec):
    self.capi, self.spec, self.rec, self.saved = capi, spec, {}, {}

  def run(self, model, batch, steps=2):
    capi = self.capi
    prev = os.environ.get("OS2S_WGRAD_STREAM")
    os.environ["OS2S_WGRAD_STREAM"] = "0"
    for name, (fam, work) in self.spec.items():
      orig = getattr(capi, name)
      self.saved[name] = orig

      def wrapped(*a, _orig=orig, _fam=fam, _work=work, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = _orig(*a, **kw)
        e1.record()
        r = self.rec.setdefault(_fam, [0, [], 0.0])
        r[0] += 1
        r[1].append((e0, e1))
        r[2] += float(_work(a, kw))
        return out
      setattr(capi, name, wrapped)
    try:
      for _ in range(steps):
        model.train_step(batch)
      torch.cuda.synchronize()
    finally:
      for name, fn in self.saved.items():
        setattr(capi, name, fn)
      if prev is None:
        os.environ.pop("OS2S_WGRAD_STREAM", None)
      else:
        os.environ["OS2S_WGRAD_STREAM"] = prev
    out = {}
    for fam, (n, evs, work) in self.rec.items():
      ms = sum(a.elapsed_time(b) for a, b in evs)
      out[fam] = {"launches_per_step": n / float(steps), "ms_per_step": ms / steps, "work_per_step": work / steps}
    return out


def _roofline(bound, kernel, achieved, how, traffic=None, **extra):
  peak = HBM_PEAK_GBS if bound == "hbm" else BF16_DENSE_PEAK_TFLOPS
  d = {"bound": bound, "kernel": kernel, "achieved": achieved, "peak": peak,
       "unit": "GB/s" if bound == "hbm" else "TFLOP/s", "frac": achieved / peak, "traffic": traffic, "how": how}
  d.update(extra)
  return d


def other_config_roofline(key, model, batch, res):
  """The roofline block of one `other_configs` entry: what bounds the configuration's
  dominant kernel family, measured in a serial pass after the timed steps."""
  from openseq2seq_amd import capi
  if key == "ds2":
    # the recurrence: one persistent launch per GRU layer and pass (csrc/rnn_xcd.hip) or one launch per step
    def steps_of(a, kw):
      dirs = a[1]
      t = dirs[0]["gx"] if "gx" in dirs[0] else dirs[0]["dy"]
      return t.shape[1]           # sequential steps of the call (both directions advance together)
    fb = FamilyBrackets(capi, {"rnn_layer_fwd_multi": ("recurrence forward", steps_of),
                               "rnn_layer_bwd_multi": ("recurrence backward", steps_of)}).run(model, batch)
    enc = model.get_encoder()
    H = enc.params["rnn_cell_dim"]
    B = batch["source_tensors"][0].shape[0]
    f, b = fb.get("recurrence forward"), fb.get("recurrence backward")
    us_f = 1e3 * f["ms_per_step"] / max(f["work_per_step"], 1) if f else None
    us_b = 1e3 * b["ms_per_step"] / max(b["work_per_step"], 1) if b else None
    rec_ms = (f["ms_per_step"] if f else 0.0) + (b["ms_per_step"] if b else 0.0)
    # per sequential step and direction: recurrent product 2 * B * 3H * H FLOP forward, twice that backward
    flop = 2.0 * 2 * B * 3 * H * H * ((f["work_per_step"] if f else 0) + 2 * (b["work_per_step"] if b else 0))
    return _roofline(
        "mfma", "gru_xcd_fwd_kernel / gru_xcd_bwd_kernel (persistent, one XCD per direction)",
        flop / (rec_ms * 1e-3) / 1e12 if rec_ms else 0.0,
        "recurrent-product FLOPs of the bracketed launches / their time (serial pass). The recurrence is bound "
        "by its per-step exchange, not by the matrix pipe: see us_per_step",
        us_per_step_forward=us_f, us_per_step_backward=us_b, recurrence_ms_per_train_step=rec_ms,
        share_of_step=rec_ms / res["ms_per_step"],
        exchange_bytes_per_step_per_direction=B * H * 2,
        exchange="all-gather (forward) / reduce-scatter (backward) of the bf16 state among the 32 workgroups of one "
                 "XCD through that XCD's L2, 16-byte tagged granules; the guide prices a 16-32 KB all-gather at "
                 "2.4-4.2 us (MI355X_MICROARCH.md, row allgather)",
        )
  if key == "quartznet":
    def dw_bytes(a, kw):
      x = a[0]
      return 2.0 * x.numel() * 2          # read x, write y (bf16); the [K, C] filter is noise
    fb = FamilyBrackets(capi, {"depthwise_conv1d_fwd": ("depthwise forward / data gradient", dw_bytes),
                               "depthwise_conv1d_wgrad": ("depthwise weight gradient", dw_bytes)}).run(model, batch)
    ms = sum(v["ms_per_step"] for v in fb.values())
    by = sum(v["work_per_step"] for v in fb.values())
    return _roofline("hbm", "depthwise_mfma_fwd_kernel / depthwise_mfma_wgrad_kernel (stride 1, dilation 1: Toeplitz band x "
                            "segmented time series on the matrix cores) + the register-window kernels of the dilated / "
                            "strided layers (csrc/depthwise.hip)",
                     by / (ms * 1e-3) / 1e9 if ms else 0.0,
                     "algorithmic bytes (input + output, bf16) of the bracketed depthwise launches / their time (event "
                     "pairs around single launches: the 256-channel ones sit at the ~8 us host floor of a call)",
                     depthwise_ms_per_train_step=ms, share_of_step=ms / res["ms_per_step"],
                     families=fb)
  if key in ("nmt", "tacotron"):
    # sequential decoder loops: price the loop launches
    def steps_fwd(a, kw):
      self_ = a[0]
      t0 = a[1] if len(a) > 1 else kw.get("t_begin", 0)
      t1 = a[2] if len(a) > 2 else kw.get("t_end")
      return (self_.dims["T"] if t1 is None else t1) - t0
    saved_f, saved_b = capi.AttnDecoder.forward, capi.AttnDecoder.backward
    rec = {"f": [0.0, [], 0], "b": [0.0, [], 0]}

    def fwd(self_, *a, **kw):
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record(); out = saved_f(self_, *a, **kw); e1.record()
      rec["f"][1].append((e0, e1)); rec["f"][2] += steps_fwd((self_,) + a, kw)
      return out

    def bwd(self_, *a, **kw):
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record(); out = saved_b(self_, *a, **kw); e1.record()
      rec["b"][1].append((e0, e1)); rec["b"][2] += self_.dims["T"]
      return out
    capi.AttnDecoder.forward, capi.AttnDecoder.backward = fwd, bwd
    try:
      for _ in range(2):
        model.train_step(batch)
      torch.cuda.synchronize()
    finally:
      capi.AttnDecoder.forward, capi.AttnDecoder.backward = saved_f, saved_b
    msf = sum(a.elapsed_time(b) for a, b in rec["f"][1]) / 2
    msb = sum(a.elapsed_time(b) for a, b in rec["b"][1]) / 2
    nf, nb = rec["f"][2] / 2.0, rec["b"][2] / 2.0
    dec = model.get_decoder()
    cell = dec.cell
    # bytes a decoder step must stream: the recurrent weight matrices of the loop (bf16, or e4m3 forward)
    wbytes = sum(w.numel for w in cell.wcat) * (1.0 if getattr(cell, "fp8_weights", False) else 2.0)
    us_f = 1e3 * msf / max(nf, 1)
    return _roofline("hbm", "attention-decoder loop (ad_cell_fwd / ad_loc_* / ad_attn_* kernels, csrc/attn_decoder.hip)",
                     wbytes / (us_f * 1e-6) / 1e9 if us_f else 0.0,
                     "recurrent weight bytes a forward decoder step streams / its measured time; the loop is a chain of "
                     "dependent launches (latency-bound), the weights are re-read from L2 / MALL every step",
                     us_per_decoder_step_forward=us_f, us_per_decoder_step_backward=1e3 * msb / max(nb, 1),
                     decoder_steps_per_train_step=nf, loop_ms_per_train_step=msf + msb,
                     share_of_step=(msf + msb) / res["ms_per_step"], weight_bytes_per_step=wbytes)
  return None



def tensorflow_probe():
  """SURVEY 8d plan (1): the reference's own TF1 CPU path as the baseline when TensorFlow is
  importable on the bench host. It is not part of this image (and /root/reference does not travel
  to the GPU box), so the probe documents its absence in the JSON line and the oracle port is
  timed instead."""
  try:
    import tensorflow as tf  # noqa: F401
    return {"tensorflow": getattr(tf, "__version__", "?")}
  except Exception as e:
    return {"tensorflow": None, "reason": type(e).__name__}


def cpu_baseline(batch, vocab_size=29, budget_s=30.0):
  """Oracle training step (fp32 torch-CPU + NumPy NovoGrad/LARC/Backoff: oracle/tdnn.py,
  oracle/optim.py) on a bounded SUB-BATCH OF THE SAME synthetic batch the GPU was timed on: FOUR
  utterances at the 0, 1/3, 2/3 and 1 quantiles of its length distribution (features, lengths and
  labels as in the batch, padded to the longest of the four, so BatchNorm is a batch statistic and
  the padding / masking work of a ragged step is in the number). 3 warm-up steps, then >= 5 timed
  steps (as many as fit the time budget)."""
  from oracle import tdnn as otdnn, optim as oopt
  from openseq2seq_amd.configs.jasper import jasper_convnet_layers
  torch.manual_seed(0)
  layers = jasper_convnet_layers()
  ncores = os.cpu_count() or 1
  nthreads = min(ncores, 64)
  torch.set_num_threads(nthreads)
  # weights in TF layout
  w = {}
  cin, res = 64, []
  for ib, blk in enumerate(layers):
    if blk.get("residual"):
      res.append(cin)
    for ir in range(blk["repeat"]):
      n = "conv%d%d" % (ib + 1, ir + 1)
      k, c = blk["kernel_size"][0], blk["num_channels"]
      w[n + "/kernel"] = (torch.randn(k, cin, c) * (2.0 / (k * (cin + c))) ** 0.5).requires_grad_(True)
      w[n + "/bn/gamma"] = torch.ones(c, requires_grad=True)
      w[n + "/bn/beta"] = torch.zeros(c, requires_grad=True)
      if blk.get("residual") and ir == blk["repeat"] - 1:
        for i, rc in enumerate(res):
          w[n + "/res_%d/kernel" % i] = (torch.randn(1, rc, c) * (2.0 / (rc + c)) ** 0.5).requires_grad_(True)
          w[n + "/res_bn_%d/gamma" % i] = torch.ones(c, requires_grad=True)
          w[n + "/res_bn_%d/beta" % i] = torch.zeros(c, requires_grad=True)
      cin = c
  fcw = (torch.randn(1024, vocab_size) * 0.03).requires_grad_(True)
  fcb = torch.zeros(vocab_size, requires_grad=True)
  names = sorted(w.keys())
  tensors = [w[n] for n in names] + [fcw, fcb]
  opt = oopt.RefOptimizer([t.detach().numpy() for t in tensors], optimizer="NovoGrad",
                          opt_params=dict(beta1=0.95, beta2=0.98, epsilon=1e-8, weight_decay=0.001),
                          lr_fn=lambda s: oopt.poly_decay(s, 0.02, 100000, power=2.0, min_lr=1e-5),
                          larc_params=dict(larc_eta=0.001), scaler=oopt.BackoffScaler())
  feats, frames = [t.cpu() for t in batch["source_tensors"]]
  tgt, tlen = [t.cpu() for t in batch["target_tensors"]]
  order = torch.argsort(frames)
  nb = int(order.numel())
  pick = order[sorted({0, (nb - 1) // 3, (2 * (nb - 1)) // 3, nb - 1})]
  lens = frames[pick].to(torch.int64)
  T = int(-(-int(lens.max()) // 8) * 8)
  x = feats[pick, :T].float()
  label_len = tlen[pick].to(torch.int64)
  labels = tgt[pick, :int(label_len.max())].to(torch.int64)

  split = {"model": 0.0, "optimizer": 0.0}

  def step():
    t_a = time.time()
    for t in tensors:
      t.grad = None
    out, olen = otdnn.tdnn_encode(x, lens, layers, w)
    _, loss = otdnn.fc_ctc(out, olen, fcw, fcb, labels, label_len)
    scale = float(opt.loss_scale)
    (loss * scale).backward()
    t_b = time.time()
    opt.step([t.grad.numpy() for t in tensors])
    with torch.no_grad():
      for t, nw in zip(tensors, opt.w):
        t.copy_(torch.from_numpy(nw))
    split["model"] += t_b - t_a
    split["optimizer"] += time.time() - t_b         # batch-independent: 333 M parameters through NumPy
    return float(loss.detach())

  t_w = time.time()
  for _ in range(3):
    step()
  per = (time.time() - t_w) / 3
  nt = max(5, min(12, int(max(budget_s - 3 * per, 0.0) / max(per, 1e-3))))
  split["model"] = split["optimizer"] = 0.0
  t0 = time.time()
  for _ in range(nt):
    step()
  dt = (time.time() - t0) / nt
  nframes = int(lens.sum())
  out = {"value": nframes / dt, "unit": "frames/sec", "cores": nthreads, "kind": "port",
         # the step's two halves: forward + backward scale with the frames, the NumPy optimizer pass over the
         # 333 M parameters does not — at the bench batch (32 utterances) the second term is amortised 8 x further
         "seconds_per_step": {"forward_backward": split["model"] / nt, "optimizer": split["optimizer"] / nt},
         "frames_per_sec_forward_backward_only": nframes / max(split["model"] / nt, 1e-9),
         "sample": "Jasper10x5 oracle train step (fp32 torch-CPU + NumPy NovoGrad) on %d utterances of "
                   "the bench batch at the 0, 1/3, 2/3, 1 quantiles of its lengths (%s frames, padded "
                   "to %d), 3 warm-up + %d timed steps, %.2f s/step"
                   % (int(lens.numel()), "/".join(str(int(v)) for v in lens), T, nt, dt)}
  out.update(tensorflow_probe())
  return out


def cpu_baseline_nmt(batch, vocab=32768, sentences=16, budget_s=20.0):
  """BASELINE.json configs[0] (en-de-nmt-small, CPU fp32, world_size 1) as a timed CPU leg: the oracle port of the
  model (oracle/nmt.py: 2 x bi-LSTM-512 encoder, GNMT-v2 attention decoder, 32 k softmax, BasicSequenceLoss) +
  torch Adam, forward + backward + update on the first `sentences` sentence pairs of the synthetic batch the GPU
  leg ran (a bounded sample: the full batch of 128 is about 8 x the work). TensorFlow — the reference's own CPU
  path — is not importable on the bench host (tensorflow_probe)."""
  from oracle import nmt as onmt
  torch.manual_seed(0)
  n = sentences
  src, sl = batch['source_tensors'][0][:n].cpu(), batch['source_tensors'][1][:n].cpu()
  tgt, tl = batch['target_tensors'][0][:n].cpu(), batch['target_tensors'][1][:n].cpu()
  src, tgt = src[:, :int(sl.max())], tgt[:, :int(tl.max())]
  H, E, U, M, V = 512, 512, 512, 1024, vocab
  leaves = []

  def w(*shape):
    t = (torch.rand(*shape) * 2 - 1) * (6.0 / (shape[0] + shape[-1])) ** 0.5
    t.requires_grad_(True)
    leaves.append(t)
    return t

  def lstm(inp):
    return dict(wx=w(4 * H, inp), wh=w(4 * H, H), b=w(4 * H))
  P = {"emb": w(V, E), "fw": [lstm(E), lstm(H)], "bw": [lstm(E), lstm(H)]}
  cell = dict(wcat=[w(4 * H, M + H)], bias=[None], wq=w(U, H), wmem=w(U, M), v=w(U), g=w(1), b=w(U),
              w_in=w(4 * H, E), b0=w(4 * H))
  D = {"demb": w(V, E), "cell": cell,
       "upper": [dict(wx_h=w(4 * H, H), wx_a=w(4 * H, M), wh=w(4 * H, H), b=w(4 * H))], "proj": w(V, H)}
  opt = torch.optim.Adam(leaves, lr=1e-3)
  # hundreds of small sequential ops per step: more threads than this only add barrier time (with every core of a
  # 2-socket host the step did not finish in 10 minutes)
  torch.set_num_threads(min(os.cpu_count() or 1, 16))

  def step():
    opt.zero_grad()
    enc = onmt.encoder(P, src, sl)
    logits = onmt.decoder_logits(D, enc, sl, tgt, tl, "gnmt_v2")
    loss = onmt.basic_sequence_loss(logits, tgt, tl, n)
    loss.backward()
    opt.step()
    return float(loss.detach())
  step()
  t0 = time.perf_counter()
  k = 0
  while k < 3 or (time.perf_counter() - t0 < budget_s and k < 10):
    step()
    k += 1
  dt = (time.perf_counter() - t0) / k
  toks = float(sl.sum() + tl.sum())
  probe = tensorflow_probe()
  return {"value": toks / dt, "unit": "tokens/sec", "cores": torch.get_num_threads(), "kind": "port",
          "sample": "en-de-nmt-small oracle train step (oracle/nmt.py fp32 torch-CPU + torch Adam) on the first %d "
                    "sentence pairs of the GPU leg's batch (%d tokens), %d timed steps of %.2f s" % (n, toks, k, dt),
          "tensorflow": probe.get("tensorflow"), "reason": probe.get("reason")}


def cpu_baseline_nmt_guarded(batch, limit_s=120):
  """cpu_baseline_nmt in a child process with a hard time limit: a CPU leg must never cost the bench line."""
  import subprocess
  import tempfile
  fail = lambda why: {"value": None, "unit": "tokens/sec", "cores": 0, "kind": "port", "sample": "failed: " + why}
  try:
    with tempfile.TemporaryDirectory() as d:
      f = os.path.join(d, "batch.pt")
      torch.save({"source_tensors": [t[:16].cpu() for t in batch["source_tensors"][:2]],
                  "target_tensors": [t[:16].cpu() for t in batch["target_tensors"][:2]]}, f)
      env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
      r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-nmt-leg", f], capture_output=True, text=True,
                         timeout=limit_s, env=env)
      lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
      return json.loads(lines[-1]) if lines else fail("no output (rc %d): %s" % (r.returncode, r.stderr[-200:]))
  except subprocess.TimeoutExpired:
    return fail("time limit of %d s" % limit_s)
  except Exception as e:
    return fail(repr(e))


def bench_frontend(dev, batch_size, seed=1234, reps=20):
  """The log-mel front end (get_speech_features_librosa, speech_utils.py:322-441) as
  its own timed stage: int16 PCM of the bench batch's durations resident in HBM -> normalised
  bf16 features [B, Tpad, 64]. Algorithmic bytes = PCM read once + features written once (+ the
  fp32 pass of the per-feature normalisation). At the bench batch (B = 32: 31 MB, four launches since round 6)
  the stage is launch-/latency-bound; `saturated` times the same kernels on 16 bench batches in one
  call (the kernels' rate when the chip is full: ~56 kFLOP of fp32 FFT / mel work per frame make
  the frames kernel VALU-bound, not HBM-bound)."""
  import numpy as np
  from openseq2seq_amd.data.speech2text.speech_utils import LogMelFrontEnd
  params = dict(sample_freq=16000, backend="librosa", input_type="logfbank", num_audio_features=64,
                window_size=20e-3, window_stride=10e-3, dither=1e-5, norm_per_feature=True,
                window="hanning", num_fft=512, pad_to=16)
  fe = LogMelFrontEnd(params, dev)

  def run(bs):
    rng = np.random.RandomState(seed)
    dur = np.tile(rng.uniform(2.0, 16.7, size=batch_size), bs // batch_size)
    ns = (dur * 16000).astype(np.int32)
    nmax = int(ns.max())
    pcm = torch.randint(-20000, 20000, (bs, nmax), dtype=torch.int16, device=dev)
    n_samples = torch.from_numpy(ns).to(dev)
    for _ in range(3):
      feats, frames, _ = fe(pcm, n_samples, max_samples=nmax, seed=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
      feats, frames, _ = fe(pcm, n_samples, max_samples=nmax, seed=i)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nframes = int((1 + ns // fe.hop).sum())
    # PCM read (2 B/sample) + bf16 features written (128 B/frame) + the fp32 feature plane written
    # and re-read by the per-feature normalisation (2 x 256 B/frame)
    algo = float(ns.sum()) * 2.0 + nframes * (128.0 + 512.0)
    return ms, nframes, int(bs * feats.shape[1]), algo

  ms, nframes, padded, algo = run(batch_size)
  traffic, traffic_src = committed_pmc_traffic("frontend")
  out = {"metric": "audio-frames/sec log-mel front end (int16 PCM -> normalised bf16 features)",
         "value": nframes / (ms * 1e-3), "unit": "frames/sec", "ms_per_batch": ms,
         "frames_per_batch": nframes, "padded_frames": padded,
         "roofline": {"bound": "hbm", "achieved": algo / (ms * 1e-3) / 1e9, "peak": 8000.0,
                      "unit": "GB/s", "frac": algo / (ms * 1e-3) / 1e9 / 8000.0, "traffic": traffic,
                      "traffic_source": traffic_src, "traffic_unit": "HBM bytes per batch (all launches of the stage)",
                      "algorithmic_bytes": algo,
                      "note": "launch-bound at the bench batch (4 launches, 31 MB); see `saturated`"}}
  try:
    if os.environ.get("OS2S_FRONTEND_NO_SATURATED"):      # the PMC pass counts the bench batch alone (tools/pmc_frontend.sh)
      raise RuntimeError("skipped (OS2S_FRONTEND_NO_SATURATED)")
    ms16, nf16, _, algo16 = run(16 * batch_size)
    out["saturated"] = {"batch": 16 * batch_size, "value": nf16 / (ms16 * 1e-3), "unit": "frames/sec",
                        "ms_per_call": ms16, "hbm_GBps": algo16 / (ms16 * 1e-3) / 1e9,
                        "fp32_tflops": 56e3 * nf16 / (ms16 * 1e-3) / 1e12}
  except Exception as e:  # noqa
    out["saturated"] = {"error": repr(e)}
  return out


def bench_transformer(args, hvd, dev, rank, world):
  """Secondary headline: tokens/sec of Transformer-big (transformer-big.py: B=256 pairs/GPU,
  lengths U[8,56], V=32768) full train step; objects = src+tgt tokens (text2text.py:227-241)."""
  from openseq2seq_amd.configs.transformer import transformer_config
  model_cls, params = transformer_config(batch_size_per_gpu=args.transformer_batch)
  model = model_cls(params, mode="train", hvd=hvd, device=dev)
  model.compile()
  batch = model.get_data_layer().synthetic_batch(dev, seed=1234 + rank)

  def barrier():
    if world > 1:
      torch.distributed.barrier()
    torch.cuda.synchronize()

  for _ in range(args.warmup):
    model.train_step(batch)
  barrier()
  reducer = getattr(model, "_reducer", None)
  if reducer is not None:
    reducer.timing = True
  t0 = time.perf_counter()
  for _ in range(args.steps):
    loss = model.train_step(batch)
  barrier()
  dt = time.perf_counter() - t0
  comm = None
  if reducer is not None:
    reducer.timing = False
    comm = reducer.pop_timing()
    if comm is not None:
      comm["backend"] = torch.distributed.get_backend()
      comm["exposed_share_of_step"] = comm["exposed_ms_per_step"] / (1000.0 * dt / args.steps)
  tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
  toks = torch.tensor([float(batch['num_tokens'])], dtype=torch.float64, device=dev)
  if world > 1:
    torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    torch.distributed.all_reduce(toks, op=torch.distributed.ReduceOp.SUM)
  dt = float(tmax.item())
  st = model.train_op.read_state()
  tps = float(toks.item()) * args.steps / dt
  res = {
      "metric": "tokens/sec Transformer-big bf16 (train step, objects = src+tgt tokens)",
      "value": tps, "unit": "tokens/sec", "ms_per_step": 1000.0 * dt / args.steps,
      "workload": "Transformer-big (transformer-big.py): B=%d pairs/GPU, lengths U[8,56], "
                  "V=32768, packed tokens, fwd+bwd+all-reduce+Adam" % args.transformer_batch,
      "tokens_per_step": float(toks.item()),
      # 0.629 GFLOP per counted token (train), SURVEY 8d: the packed layout executes exactly the
      # counted tokens, so the whole-step rate IS an executed-FLOP rate (GEMMs + attention)
      "roofline": {"bound": "mfma", "kernel": "whole train step (the in-tree MFMA GEMMs dominate: gemm_pp_kernel, "
                             "conv1d_wgrad1x1_pp_kernel; no vendor GEMM in the product library)",
                   "achieved": 0.629e-3 * tps, "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                   "frac": 0.629e-3 * tps / BF16_DENSE_PEAK_TFLOPS, "traffic": committed_pmc_traffic("transformer")[0],
                   "traffic_source": committed_pmc_traffic("transformer")[1],
                   "traffic_unit": "HBM bytes per gemm_pp_kernel launch (FETCH_SIZE x 2 + WRITE_SIZE, committed PMC pass)"},
      "params_M": model.store.num_trainable() / 1e6,
      "loss": float(loss.cpu()[0]), "skipped_steps": st["num_skipped"],
  }
  if comm is not None:
    res["comm"] = comm
  del model
  torch.cuda.empty_cache()
  return res


def bench_simple(spec, steps, warmup, hvd, dev, rank, world, roofline_key=None, cpu_leg=False):
  """One model of BASELINE.json's other configs: K timed train steps on a synthetic batch."""
  import importlib
  mod, fn, kw, metric, count_key, unit = spec
  model_cls, params = getattr(importlib.import_module(mod), fn)(**kw)
  model = model_cls(params, mode="train", hvd=hvd, device=dev)
  model.compile()
  batch = model.get_data_layer().synthetic_batch(dev, seed=1234 + rank)
  for _ in range(warmup):
    model.train_step(batch)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    loss = model.train_step(batch)
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  res = {"metric": metric, "value": batch[count_key] * steps / dt, "unit": unit,
         "dtype": "fp8-weights" if kw.get("fp8_weights") else "bf16",
         "ms_per_step": 1000 * dt / steps, "n_gpus": world, "steps": steps,
         "params_M": model.store.num_trainable() / 1e6, "loss": float(loss.cpu()[0])}
  if roofline_key is not None:
    try:
      res["roofline"] = other_config_roofline(roofline_key, model, batch, res)
    except Exception as e:      # a diagnostic must not lose the measurement
      res["roofline"] = {"error": repr(e)}
  if roofline_key == "nmt" and cpu_leg:
    res["cpu_baseline"] = cpu_baseline_nmt_guarded(batch)
  del model
  torch.cuda.empty_cache()
  return res


def bench_transformer_infer(dev, batch=64, reps=2):
  """Transformer-big beam-search inference (beam 4, alpha 0.6, extra_decode_length 50 as in
  transformer-big.py): random-init weights almost never emit EOS, so every sentence decodes
  the full input_length + 50 positions — the worst case. Reports decoded beam positions/sec
  (B * beam * steps / time) and ms per decoding step, encoder included."""
  from openseq2seq_amd.configs.transformer import transformer_config
  model_cls, params = transformer_config(batch_size_per_gpu=batch)
  model = model_cls(params, mode="infer", hvd=None, device=dev)
  model.compile()
  data = model.get_data_layer().synthetic_batch(dev, seed=7)
  best = None
  for _ in range(reps + 1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ids, _ = model.infer_batch(data)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
  steps = int(ids.shape[1])
  beam = params["decoder_params"]["beam_size"]
  res = {"metric": "beam positions/sec Transformer-big bf16 beam-search inference",
         "value": batch * beam * steps / best, "unit": "positions/sec",
         "ms_per_decode_step": 1000 * best / steps, "decode_steps": steps, "sentences": batch,
         "beam_size": beam, "ms_per_batch": 1000 * best}
  del model
  torch.cuda.empty_cache()
  return res


def bench_tacotron_decode(dev, style=True, fp8=True, batch=32, steps=1000, reps=2):
  """BASELINE.json configs[4] as it is named: Tacotron2-GST free-running DECODE with fp8 weights (eval / infer
  mode of tacotron_gst.py: TacotronHelper, parts/tacotron/tacotron_helper.py:138-226). B = 32 utterances, text
  lengths U[20, 200], random-init weights. The reference decodes until every stop token has fired or 10 x
  max(src_len) steps; random weights make the stop token meaningless, so the loop is run for a FIXED number of
  steps (mask_decoder_sequence off: the full per-step work, including the stop projection and the device-side
  bookkeeping, still runs). Timed: the whole infer call — encoder + style tokens, memory keys, the step loop,
  post-net, magnitude branch — with the batch resident in HBM; us_per_step comes from the difference of two run
  lengths (it excludes everything outside the loop)."""
  from openseq2seq_amd.configs.tacotron import tacotron_gst_config
  model_cls, params = tacotron_gst_config(batch_size_per_gpu=batch, style=style, fp8_weights=fp8)
  params["decoder_params"]["mask_decoder_sequence"] = False
  model = model_cls(params, mode="infer", hvd=None, device=dev)
  model.compile()
  data = model.get_data_layer().synthetic_batch(dev, seed=4321)

  def run(n):
    best = None
    for _ in range(reps + 1):
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      out = model.infer_batch(data, max_decoder_steps=n)
      torch.cuda.synchronize()
      dt = time.perf_counter() - t0
      best = dt if best is None else min(best, dt)
    return best, out

  t_full, out = run(steps)
  t_half, _ = run(steps // 2)
  us = 1e6 * (t_full - t_half) / (steps - steps // 2)
  dec = model.get_decoder()
  cell = dec.cell
  src_len = data["source_tensors"][1].float()
  S = int(data["source_tensors"][0].shape[1])
  live = float(src_len.sum())
  H, M, U, P, nm = cell.H, cell.M, cell.U, dec.prenet[0].cout, dec.n_mel
  wb = 1.0 if fp8 else 2.0
  by = {"lstm_weights": wb * (4 * H * (P + M + H) + 4 * H * 2 * H) + (4.0 * 8 * H if fp8 else 0.0),
        "query_frame_prenet_weights": 2.0 * (U * H + nm * H + P * nm + P * P),
        "memory_values_live_rows": 2.0 * live * M, "memory_keys_live_rows": 2.0 * live * U,
        "frame_projection_of_values_live_rows": 2.0 * live * nm,
        "state_vectors": 2.0 * batch * (P + M + H + 2 * H) * 2 + 4.0 * batch * (2 * H + 3 * S)}
  total = sum(by.values())
  fused = bool(out.get("fused"))
  if not fused:
    raise SystemExit("bench.py: the free-running Tacotron2 decode did not run on the fused step kernels "
                     "(TacotronInfer.supported() is false or OS2S_TACOTRON_FUSED_DECODE=0): the line below would "
                     "describe kernels that did not run")
  res = {"metric": "mel-frames/sec Tacotron2-GST free-running decode (%s decoder weights)" % ("fp8 e4m3" if fp8 else "bf16"),
         "value": batch * steps / t_full, "unit": "frames/sec", "dtype": "fp8-weights" if fp8 else "bf16",
         "ms_per_batch": 1e3 * t_full, "decoder_steps": steps, "utterances": batch, "us_per_step": us,
         "launches_per_step": 4, "host_syncs_per_step": 1.0 / (2 * dec.POLL_STEPS),
         "workload": "tacotron_gst.py infer: B=%d, text U[20,200] (padded S=%d), %d free-running steps, "
                     "encoder + GST + post-net + magnitude branch included in ms_per_batch" % (batch, S, steps),
         "roofline": _roofline(
             "hbm", "one decoder step = ti_lstm_kernel x 2 + ti_scores_kernel + ti_context_kernel (csrc/tacotron_infer.hpp)",
             total / (us * 1e-6) / 1e9 if us > 0 else 0.0,
             "algorithmic bytes one step must read (recurrent weights once, live memory rows once, state vectors) / "
             "measured time per step. ~%.0f MB per step: it is re-read every step and fits the 256 MB MALL (and most "
             "of the per-XCD slices of the weight stream fit the 4 MB L2s), so after the first step little of it is "
             "HBM traffic: 'achieved' is an effective (algorithmic) rate against the HBM peak, as for every re-read "
             "working set" % (total / 1e6),
             bytes_per_step=by, bytes_per_step_total=total, floor_us_at_hbm_peak=total / (HBM_PEAK_GBS * 1e3),
             weights="streamed every step (e4m3: 17.8 MB; L2 / MALL resident across steps, not register-stationary)")}
  del model
  torch.cuda.empty_cache()
  return res


def committed_pmc_traffic(kind="bench"):
  """HBM bytes per launch of a configuration's dominant kernel from the committed PMC pass of the same command
  (tools/profile_round.sh: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 correction applied). Counters cannot
  be collected inside the timed run, so the JSON line carries the number of the newest
  profiles/*_pmc_<kind>_traffic.json (kind: bench = Jasper conv kernels, transformer = gemm_pp_kernel,
  frontend = the log-mel kernels per batch), or null if there is none."""
  import glob
  here = os.path.dirname(os.path.abspath(__file__))
  files = sorted(glob.glob(os.path.join(here, "profiles", "*_pmc_%s_traffic.json" % kind)))
  if not files:
    return None, None
  try:
    with open(files[-1]) as f:
      d = json.load(f)
    return float(d["hbm_bytes_per_launch"]), "profiles/" + os.path.basename(files[-1])
  except Exception:
    return None, None


def launcher_dry_run(args, hvd, rank, world):
  """The multi-rank plumbing of the bench without GPU work: barrier, MAX-over-ranks of the timed
  interval, SUM of the per-rank units — on whatever backend the process group has (gloo on CPU)."""
  import torch.distributed as dist
  dev = torch.device("cpu")
  if world > 1:
    dist.barrier()
  t0 = time.perf_counter()
  time.sleep(0.01 * (rank + 1))            # ranks finish at different times: MAX must pick the slowest
  if world > 1:
    dist.barrier()
  dt = time.perf_counter() - t0
  tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
  units = torch.tensor([1000.0 * (rank + 1)], dtype=torch.float64, device=dev)
  if world > 1:
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(units, op=dist.ReduceOp.SUM)
  if rank == 0:
    emit({"metric": "launcher dry run (no GPU work)", "value": None, "unit": "frames/sec",
                      "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                      "ms_per_step": 1000.0 * float(tmax.item()), "higher_is_better": True,
                      "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                      "config": {"workload": "launcher dry run", "global_batch": args.batch * world,
                                 "parallelism": "dp%d" % world,
                                 "backend": dist.get_backend() if world > 1 else "none",
                                 "units_sum_over_ranks": float(units.item())}})


def main():
  args = parse()
  if args.cpu_nmt_leg:
    emit(cpu_baseline_nmt(torch.load(args.cpu_nmt_leg)))
    return
  if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
    sys.exit(spawn_ranks(args.gpus))      # one process per GPU; this process only waits for them
  claim_stdout()
  if args.bucket_mb is not None:
    os.environ["OS2S_BUCKET_MB"] = repr(args.bucket_mb)
  if args.allreduce_dtype is not None:
    os.environ["OS2S_ALLREDUCE_DTYPE"] = args.allreduce_dtype
  from openseq2seq_amd.utils import distributed as dist_utils
  hvd = dist_utils.init_from_env(backend=args.backend, one_device=args.one_device)
  if args.one_rank_group and hvd is None:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    os.environ["OS2S_FORCE_REDUCER"] = "1"
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("nccl", rank=0, world_size=1)
    hvd = dist_utils.HvdAdapter()
  rank = hvd.rank() if hvd else 0
  world = hvd.size() if hvd else 1
  if args.gpus != world:
    raise SystemExit("bench.py: --gpus %d but the process group has %d rank(s) "
                     "(WORLD_SIZE=%s)" % (args.gpus, world, os.environ.get("WORLD_SIZE")))
  if args.launcher_dry_run:
    launcher_dry_run(args, hvd, rank, world)
    return
  dev = torch.device("cuda", 0 if args.one_device else int(os.environ.get("LOCAL_RANK", 0)))
  torch.cuda.set_device(dev)
  if os.environ.get("OS2S_MAIN_PRIO"):        # experiment: the step's own stream at a HIP stream priority (-1 = high)
    _hp = torch.cuda.stream(torch.cuda.Stream(device=dev, priority=int(os.environ["OS2S_MAIN_PRIO"])))
    _hp.__enter__()
  for kv in args.set_option:
    from openseq2seq_amd import _lib as _l
    name, _, val = kv.partition("=")
    _l.set_option(name, float(val))
  if args.pp_cost:
    import ctypes
    from openseq2seq_amd import _lib
    f = _lib.lib().os2s_set_option
    f.argtypes, f.restype = [ctypes.c_char_p, ctypes.c_double], ctypes.c_int
    names = ["conv1d.pp_cost_256", "conv1d.pp_cost_2x128", "conv1d.pp_cost_3x128", "conv1d.pp_dgrad_penalty",
             "conv1d.pp_prio"]
    for n, v in zip(names, args.pp_cost.split(",")):
      assert f(n.encode(), float(v)) == 0
  if rank == 0 and world > 1:
    print("bench.py: %d ranks, backend %s%s" % (
        world, torch.distributed.get_backend(),
        ", ALL ON cuda:0 (rehearsal)" if args.one_device else " (RCCL), one process per GPU"), file=sys.stderr)

  simple = {
      "quartznet": ("openseq2seq_amd.configs.quartznet", "quartznet15x5_config", {},
                    "audio-frames/sec QuartzNet15x5 bf16 (train step)", "num_frames", "frames/sec"),
      "tacotron": ("openseq2seq_amd.configs.tacotron", "tacotron_gst_config",
                   {"style": not args.no_style, "fp8_weights": not args.no_fp8},
                   "mel-frames/sec Tacotron2-GST (train step; bf16 activations, %s decoder LSTM weights)"
                   % ("bf16" if args.no_fp8 else "fp8 e4m3"), "num_frames", "frames/sec"),
      "ds2": ("openseq2seq_amd.configs.ds2", "ds2_large_config", {},
              "audio-frames/sec DeepSpeech2-large bf16 (train step)", "num_frames", "frames/sec"),
      "nmt": ("openseq2seq_amd.configs.nmt", "nmt_small_config", {},
              "tokens/sec en-de-nmt-small bf16 (train step, synthetic 32k-vocab batches)", "num_tokens",
              "tokens/sec"),
  }
  for key, flag in (("quartznet", args.only_quartznet), ("tacotron", args.only_tacotron),
                    ("ds2", args.only_ds2), ("nmt", args.only_nmt)):
    if flag:
      res = bench_simple(simple[key], args.steps, args.warmup, hvd, dev, rank, world, roofline_key=key,
                         cpu_leg=not args.no_cpu_baseline)
      if rank == 0:
        emit(res)
      return
  if args.only_transformer_infer:
    if rank == 0:
      emit(bench_transformer_infer(dev))
    return
  if args.only_frontend:
    if rank == 0:
      emit(bench_frontend(dev, args.batch))
    return
  if args.only_tacotron_decode:
    if rank == 0:
      emit(bench_tacotron_decode(dev, style=not args.no_style, fp8=not args.no_fp8,
                                             batch=args.batch, steps=args.decode_steps))
    return
  if args.only_transformer:
    tr = bench_transformer(args, hvd, dev, rank, world)
    if rank == 0:
      emit(tr)
    return
  from openseq2seq_amd import capi
  from openseq2seq_amd.configs.jasper import jasper10x5_config
  timer = ConvTimer(capi)
  if not args.no_kernel_timing:
    timer.install()
  model_cls, params = jasper10x5_config(batch_size_per_gpu=args.batch, use_horovod=True,
                                        max_steps=100000)
  model = model_cls(params, mode="train", hvd=hvd, device=dev)
  model.compile()
  dl = model.get_data_layer()
  batch = dl.synthetic_batch(dev, seed=1234 + rank,
                             fixed_frames=args.fixed_frames if args.fixed_frames > 0 else None)

  if args.no_host_lens:
    batch.pop('source_lengths_host', None)

  def barrier():
    if world > 1:
      torch.distributed.barrier()
    torch.cuda.synchronize()

  # A step starts from PCM resident in HBM (north_star puts the log-mel front end on the path): int16 audio of
  # the batch's durations -> os2s_logmel -> features -> train step, all inside the timed region. The synthetic
  # batch above fixes the durations / labels; its N(0,1) feature tensor is replaced by the front end's output.
  step_from_pcm = None
  if not args.fixed_frames:
    try:
      import numpy as np
      from openseq2seq_amd.data.speech2text.speech_utils import make_front_end
      fe = make_front_end(dl.params, dev)
      sr = dl.params.get('sample_freq', 16000)
      dur = np.random.RandomState(1234 + rank).uniform(2.0, dl.params.get('max_duration', 16.7), size=args.batch)
      ns = (dur * sr).astype(np.int32)
      nmax = int(ns.max())
      pcm = torch.randint(-20000, 20000, (args.batch, nmax), dtype=torch.int16, device=dev)
      n_samples = torch.from_numpy(ns).to(dev)
      feats, frames_dev, _ = fe(pcm, n_samples, max_samples=nmax, seed=0)
      if tuple(feats.shape) == tuple(batch['source_tensors'][0].shape) and \
         torch.equal(frames_dev.cpu(), batch['source_tensors'][1].cpu()):
        def step_from_pcm(i):
          f, fr, _ = fe(pcm, n_samples, max_samples=nmax, seed=i)
          batch['source_tensors'] = [f, fr]
          return model.train_step(batch)
    except Exception as e:      # never lose the headline to the front-end plumbing: fall back, and say so
      print("bench.py: front end not in the timed step (%r)" % (e,), file=sys.stderr)
      step_from_pcm = None

  def one_step(i):
    return step_from_pcm(i) if step_from_pcm is not None else model.train_step(batch)

  warm_ms = None
  for i in range(args.warmup):
    tw = time.perf_counter()
    one_step(i)
    if i >= args.warmup - 2:        # the last two warm-up steps are timed one by one: they size the clock probe
      torch.cuda.synchronize()
      w = 1000.0 * (time.perf_counter() - tw)
      warm_ms = w if warm_ms is None else min(warm_ms, w)
  barrier()
  # one wave on a private stream reads the shader clock the chip sustains UNDER the timed steps (it ends before
  # they do: half the expected region at 2.4 GHz is at most 0.8 of it at any clock the part runs)
  clock_probe = None
  if rank == 0 and warm_ms is not None and not args.no_kernel_timing:
    try:
      clock_probe = capi.clock_probe_start(0.5 * args.steps * warm_ms * 2.4e6)
    except Exception as e:
      print("bench.py: no shader-clock probe (%r)" % (e,), file=sys.stderr)
  timer.enabled = not args.no_kernel_timing and rank == 0
  reducer = getattr(model, "_reducer", None)
  if reducer is not None:
    reducer.timing = True           # a few event records per step (one pair per 128 MB bucket)
  t0 = time.perf_counter()
  for i in range(args.steps):
    loss = one_step(args.warmup + i)
  barrier()
  dt = time.perf_counter() - t0
  timer.enabled = False
  shader_mhz = None
  if clock_probe is not None:
    try:
      shader_mhz = capi.clock_probe_read(clock_probe)
    except Exception as e:
      print("bench.py: shader-clock probe failed (%r)" % (e,), file=sys.stderr)
  # every forward AND data-gradient launch of the dominant kernel family, each alone on the GPU: two more
  # (untimed-by-the-clock) steps with the weight-gradient stream folded into the main stream and every launch
  # bracketed — the all-launch figure next to the sampled one of the timed region
  all_launch = None
  if not args.no_kernel_timing and rank == 0 and world == 1:
    try:
      main_state = (timer.records, timer.every, timer.overlap, timer.untimed, timer.seen)
      timer.records, timer.every, timer.overlap, timer.untimed, timer.seen = [], 1, False, 0, 0
      model.params['os2s_side_stream'] = False
      timer.enabled = True
      for i in range(2):
        model.train_step(batch)
      torch.cuda.synchronize()
      timer.enabled = False
      ms_a, fl_a, n_a = timer.summary()
      all_launch = {"launches_per_step": n_a / 2.0, "ms_per_step": ms_a / 2.0,
                    "achieved": fl_a / (ms_a * 1e-3) / 1e12 if ms_a > 0 else 0.0,
                    "frac": (fl_a / (ms_a * 1e-3) / 1e12 if ms_a > 0 else 0.0) / BF16_DENSE_PEAK_TFLOPS,
                    "what": "every forward + data-gradient conv launch (ping-pong, lockstep, dense-residual 1x1 GEMMs) of two "
                            "steps run with the side stream off (each kernel alone on the GPU), HIP events"}
    except Exception as e:
      all_launch = {"error": repr(e)}
    finally:
      model.params.pop('os2s_side_stream', None)
      timer.records, timer.every, timer.overlap, timer.untimed, timer.seen = main_state
      timer.enabled = False
  comm = None
  if reducer is not None:
    reducer.timing = False
    comm = reducer.pop_timing()
    if comm is not None:
      comm["backend"] = torch.distributed.get_backend()
      comm["exposed_share_of_step"] = comm["exposed_ms_per_step"] / (1000.0 * dt / args.steps)
      if rank == 0:
        print("bench.py: world %d (%s), %d all-reduce bucket(s)/step = %.1f MB in %.2f ms "
              "(bus %.0f GB/s), exposed (not hidden by backward) %.2f ms/step of %.2f"
              % (comm["world_size"], comm["backend"], len(comm["bucket_ms"]),
                 comm["allreduce_bytes_per_step"] / 1e6, comm["allreduce_ms_per_step"],
                 comm["bus_GBps"] or 0.0, comm["exposed_ms_per_step"], 1000.0 * dt / args.steps),
              file=sys.stderr)
  breakdown = None
  if not args.no_kernel_timing and rank == 0 and world == 1:
    try:       # untimed: the other kernel families of the step, each alone on the GPU
      breakdown = StepBreakdown(capi).run(model, batch, steps=2)
    except Exception as e:
      breakdown = {"error": repr(e)}

  tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
  frames = torch.tensor([float(batch['num_frames'])], dtype=torch.float64, device=dev)
  padded = torch.tensor([float(batch['padded_frames'])], dtype=torch.float64, device=dev)
  if world > 1:
    torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    torch.distributed.all_reduce(frames, op=torch.distributed.ReduceOp.SUM)
    torch.distributed.all_reduce(padded, op=torch.distributed.ReduceOp.SUM)
  dt = float(tmax.item())
  total_frames = float(frames.item()) * args.steps
  value = total_frames / dt
  st = model.train_op.read_state()
  out = {
      "metric": "audio-frames/sec/node Jasper10x5 bf16 (train step, objects = input feature frames)",
      "value": value, "unit": "frames/sec", "n_gpus": world, "steps": args.steps,
      "warmup": args.warmup, "ms_per_step": 1000.0 * dt / args.steps,
      "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
      "data": "synthetic",
      "config": {
          "workload": "Jasper 10x5 DR (jasper10x5_LibriSpeech_nvgrad_masks) full train step: %s"
                      "fwd+bwd+%sNovoGrad/LARC/Backoff, B=%d/GPU, %s, F=64, V=29" % (
                          "log-mel front end from int16 PCM resident in HBM+" if step_from_pcm is not None else "",
                          ("RCCL all-reduce+" if torch.distributed.get_backend() == "nccl"
                           else "%s all-reduce (rehearsal on one device)+" % torch.distributed.get_backend())
                          if world > 1 else "",
                          args.batch, ("T=%d fixed" % args.fixed_frames) if args.fixed_frames
                          else "durations U[2,16.7]s padded to the batch max"),
          "front_end_in_timed_step": step_from_pcm is not None,
          "global_batch": args.batch * world,
          "frames_per_step": float(frames.item()),
          "padded_frames_per_step": float(padded.item()),
          "padded_frames_per_sec": float(padded.item()) * args.steps / dt,
          "train_gflop_per_padded_frame": 0.9975,
          # dense-equivalent rate: counts the FLOPs of padded frames too, 36 % of which are never
          # executed (all-padding tiles are skipped); roofline.achieved counts executed FLOPs only
          "dense_equivalent_tflops_whole_step": 0.9975e-3 * float(padded.item()) * args.steps / dt,
          "params_M": model.store.num_trainable() / 1e6,
          "parallelism": "dp%d" % world,
          "loss": float(loss.cpu()[0]), "loss_scale": st["loss_scale"],
          "skipped_steps": st["num_skipped"],
      },
  }
  if comm is not None:
    out["comm"] = comm
  if not args.no_kernel_timing:
    ms, fl, n = timer.summary()
    ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    traffic, traffic_src = committed_pmc_traffic()
    # headline figure = EVERY forward + data-gradient launch of the family (all_launch, each alone on the GPU);
    # the sampled forward launches of the timed region ride along as sampled_*. Without the all-launch pass
    # (world > 1, or it failed) the sampled figure is the only one there is and `frac_is` says so.
    have_all = bool(all_launch) and "achieved" in all_launch and all_launch["achieved"] > 0
    head_ach = all_launch["achieved"] if have_all else ach
    wg = None
    if isinstance(breakdown, dict):
      wg = next((v for k, v in breakdown.items() if k.startswith("conv1d weight gradient (")), None)
    out["roofline"] = {
        "bound": "mfma", "kernel": "conv1d implicit GEMM (conv1d_pp_kernel + conv1d_ppn_kernel + conv1d_igemm_kernel tiles, incl. the dense-residual GEMMs over the concatenated block inputs: conv1x1_pp_kernel)",
        "achieved": head_ach, "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": head_ach / BF16_DENSE_PEAK_TFLOPS,
        "frac_is": "all forward + data-gradient launches of two serial steps" if have_all
                   else "sampled forward launches only (no all-launch pass in this run)",
        "traffic": traffic, "traffic_source": traffic_src,
        # with the dense-residual chains on their own stream next to the forward convolutions almost no forward
        # launch has the GPU to itself any more: a sample of fewer than 4 launches per step is not reported
        "sampled_achieved": ach if n >= 4 * max(args.steps, 1) else None,
        "sampled_frac": ach / BF16_DENSE_PEAK_TFLOPS if n >= 4 * max(args.steps, 1) else None,
        "all_launch_ms_per_step": all_launch.get("ms_per_step") if have_all else None,
        "wgrad_frac": wg.get("frac") if wg else None,
        "wgrad_ms_per_step": wg.get("ms_per_step") if wg else None,
        "rest_of_step_ok": isinstance(breakdown, dict) and "error" not in breakdown,
        # the peak is quoted at 2.4 GHz; under matrix load on random data the part clocks lower
        # (MI355X_MICROARCH.md "DVFS give-back"): the clock one probe wave read next to the timed steps
        "shader_clock_mhz": shader_mhz,
        "peak_at_measured_clock": BF16_DENSE_PEAK_TFLOPS * shader_mhz / 2400.0 if shader_mhz else None,
        "frac_of_peak_at_measured_clock": head_ach / (BF16_DENSE_PEAK_TFLOPS * shader_mhz / 2400.0) if shader_mhz else None,
        "timed_launches_per_step": n / max(args.steps, 1),
        "avg_launch_ms": (all_launch["ms_per_step"] / max(all_launch["launches_per_step"], 1.0)) if have_all
                         else ms / max(n, 1),
        "sampled_avg_launch_ms": ms / max(n, 1),
        "timed_launch_time_share_of_step": timer.all_ms / (1000.0 * dt),
        "executed_flop_fraction": fl / max(timer.dense_flops, 1.0),
        "dense_equivalent_tflops": timer.dense_flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
        "launches_timed": ("every 4th forward-pass launch that has the GPU to itself (not the launches next to "
                           "the early residual branches of the side stream)" if timer.overlap
                           else "every 4th forward / data-gradient launch"),
        "all_launches_per_step": (timer.all_n + timer.untimed) / max(args.steps, 1),
        # the same family over ALL its launches (forward + data gradient), each alone on the GPU
        "all_launch": all_launch,
        # the WHOLE step against the same peak: FLOPs of the real (unpadded) frames of the batch /
        # wall time per step — convolutions at `frac`, weight gradients, BatchNorm, optimizer, CTC
        # and every bubble between them
        "whole_step_achieved": 0.9975e-3 * float(frames.item()) * args.steps / dt,
        "whole_step_frac": 0.9975e-3 * float(frames.item()) * args.steps / dt / (BF16_DENSE_PEAK_TFLOPS * world),
        "by_kernel": {name: {"launches": c, "avg_launch_ms": t / max(c, 1),
                             "achieved": f / (t * 1e-3) / 1e12 if t > 0 else 0.0,
                             "frac": (f / (t * 1e-3) / 1e12 if t > 0 else 0.0) / BF16_DENSE_PEAK_TFLOPS,
                             "share_of_timed_ms": t / max(ms, 1e-9)}
                      for name, (c, t, f) in timer.by_kernel.items()},
        "rest_of_step": breakdown,
        "note": "achieved = FLOPs of the executed (non-skipped) time tiles / HIP-event time; "
                "tiles whose input window is all padding are exact zeros and are not multiplied. "
                "The data-gradient launches of the same kernel run concurrently with the "
                "weight-gradient kernels of the side stream (OS2S_WGRAD_STREAM=1): an event "
                "bracket around them would not measure the kernel alone, they are not timed",
    }
  if not args.no_transformer:
    # free the Jasper model first
    del model
    torch.cuda.empty_cache()
    timer.enabled = False
    timer.uninstall()
    try:
      tr = bench_transformer(args, hvd, dev, rank, world)
      if rank == 0:
        out["secondary"] = tr
    except Exception as e:
      if rank == 0:
        out["secondary"] = {"metric": "tokens/sec Transformer-big bf16", "value": None,
                            "error": repr(e)}
  if rank != 0:
    return
  if not args.no_other_configs and world == 1:
    others = {}
    timer.uninstall()
    for key in ("nmt", "ds2", "tacotron", "quartznet"):
      try:
        others[key] = bench_simple(simple[key], 10, 5, hvd, dev, rank, world, roofline_key=key,
                                   cpu_leg=not args.no_cpu_baseline)
      except Exception as e:   # never lose the headline line to a secondary measurement
        others[key] = {"error": repr(e)}
    try:
      others["transformer_beam_search"] = bench_transformer_infer(dev)
    except Exception as e:
      others["transformer_beam_search"] = {"error": repr(e)}
    try:
      others["tacotron_decode"] = bench_tacotron_decode(dev)
    except Exception as e:
      others["tacotron_decode"] = {"error": repr(e)}
    out["other_configs"] = others
  if world == 1:
    try:
      out["frontend"] = bench_frontend(dev, args.batch)
    except Exception as e:
      out["frontend"] = {"error": repr(e)}
  if world == 1 and not args.no_cpu_baseline:
    try:
      out["cpu_baseline"] = cpu_baseline(batch)
    except Exception as e:  # the baseline must never break the bench line
      out["cpu_baseline"] = {"value": None, "unit": "frames/sec", "cores": 0, "kind": "port",
                             "sample": "failed: %r" % (e,)}
  emit(with_headline(out))


def with_headline(out):
  """The driver keeps the last ~2 KB of stdout and, of the parsed line, the flat scalars of the contract's own
  objects. So (a) the secondary metric and the loop latencies are repeated as flat scalars inside `config`, and
  (b) a compact `headline` object is the LAST key of the one JSON line: value, ms_per_step, roofline,
  cpu_baseline and secondary.{value, ms_per_step, roofline.frac} end up in the preserved tail whatever the
  length of the line before them."""
  def pick(d, *keys):
    return {k: d.get(k) for k in keys if isinstance(d, dict) and d.get(k) is not None}
  sec = out.get("secondary") if isinstance(out.get("secondary"), dict) else {}
  oc = out.get("other_configs") if isinstance(out.get("other_configs"), dict) else {}
  cfg = out.get("config", {})
  if sec:
    cfg["secondary_metric"] = "tokens/sec Transformer-big bf16 (train step)"
    cfg["secondary_value"] = sec.get("value")
    cfg["secondary_ms_per_step"] = sec.get("ms_per_step")
    cfg["secondary_roofline_frac"] = (sec.get("roofline") or {}).get("frac")
  for key, short in (("nmt", "nmt"), ("ds2", "ds2"), ("tacotron", "tacotron_train"), ("quartznet", "quartznet")):
    if isinstance(oc.get(key), dict) and oc[key].get("ms_per_step") is not None:
      cfg[short + "_ms_per_step"] = oc[key]["ms_per_step"]
  td = oc.get("tacotron_decode")
  if isinstance(td, dict) and td.get("us_per_step") is not None:
    cfg["tacotron_decode_us_per_step"] = td["us_per_step"]
  head = pick(out, "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype")
  head["roofline"] = pick(out.get("roofline"), "bound", "achieved", "peak", "unit", "frac", "traffic", "sampled_frac",
                          "whole_step_frac", "wgrad_frac", "wgrad_ms_per_step", "all_launch_ms_per_step",
                          "shader_clock_mhz", "rest_of_step_ok")
  head["cpu_baseline"] = pick(out.get("cpu_baseline"), "value", "unit", "cores", "kind")
  if sec:
    head["secondary"] = pick(sec, "value", "unit", "ms_per_step")
    head["secondary"]["roofline_frac"] = (sec.get("roofline") or {}).get("frac")
  head["other_ms_per_step"] = {k: oc[k]["ms_per_step"] for k in ("nmt", "ds2", "tacotron", "quartznet")
                               if isinstance(oc.get(k), dict) and oc[k].get("ms_per_step") is not None}
  if isinstance(td, dict) and td.get("us_per_step") is not None:
    head["tacotron_decode_us_per_step"] = td["us_per_step"]
  keys = ("world_size", "backend", "allreduce_ms_per_step", "exposed_ms_per_step", "bus_GBps", "bucket_mb",
          "buckets_per_step", "allreduce_dtype")
  if isinstance(out.get("comm"), dict):
    head["comm"] = pick(out["comm"], *keys)
    for k in ("allreduce_ms_per_step", "exposed_ms_per_step", "bus_GBps", "bucket_mb", "allreduce_dtype"):
      if out["comm"].get(k) is not None:
        cfg["comm_" + k] = out["comm"][k]
  if isinstance(sec.get("comm"), dict):
    head["secondary"]["comm"] = pick(sec["comm"], *keys)
  out.pop("headline", None)
  out["headline"] = head
  return out


if __name__ == "__main__":
  try:
    main()
  finally:
    if torch.distributed.is_available() and torch.distributed.is_initialized():
      torch.distributed.destroy_process_group()      # RCCL communicators released before interpreter exit

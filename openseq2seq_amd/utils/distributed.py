"""Data-parallel plumbing: RCCL over xGMI through torch.distributed (backend
"nccl" is RCCL on ROCm; "gloo" for the CPU tests), one process per GPU.

Replaces the Horovod pieces of the reference:
  * hvd.init()/rank()/size()                (run.py:43-49)         -> HvdAdapter
  * BroadcastGlobalVariablesHook(0)         (utils/hooks.py:15-55)  -> broadcast_parameters
  * reduce_gradients: hvd.allreduce per gradient tensor
                                            (optimizers.py:77-104)  -> GradientReducer:
    the gradients already live in ONE flat fp32 buffer in reverse-topological
    order of production, so the all-reduce is a handful of large contiguous
    buckets (sized for xGMI: few, big messages) launched on a side stream. The
    1/world average is folded into the optimizer kernel (os2s_opt_config_t.world_size).
  * collect_if_horovod (mpi4py gather)      (utils/utils.py:47-82)  -> gather_objects
"""
import os

import torch
import torch.distributed as dist


class HvdAdapter(object):
  """Minimal stand-in for the `hvd` module object the reference passes around."""

  def __init__(self):
    assert dist.is_initialized()

  def rank(self):
    return dist.get_rank()

  def size(self):
    return dist.get_world_size()

  def local_rank(self):
    return int(os.environ.get("LOCAL_RANK", 0))


def init_from_env(backend=None, one_device=False):
  """Initialises torch.distributed from torchrun's env (RANK/WORLD_SIZE/MASTER_*).
  Returns an HvdAdapter, or None for a single-process run. one_device: every rank uses cuda:0 (the rehearsal of
  an N-rank run on a one-GPU box: the wire then has to be gloo, two ranks cannot open one device through RCCL)."""
  world = int(os.environ.get("WORLD_SIZE", "1"))
  if world <= 1:
    return None
  if not dist.is_initialized():
    if backend is None:
      backend = "nccl" if (torch.cuda.is_available() and not one_device) else "gloo"
    if one_device and backend == "nccl":
      raise ValueError("ranks sharing one device need the gloo backend")
    if torch.cuda.is_available():
      torch.cuda.set_device(0 if one_device else int(os.environ.get("LOCAL_RANK", 0)))
    dist.init_process_group(backend=backend)
  return HvdAdapter()


def configured_towers(config):
  """Number of model replicas a config asks for WITHOUT Horovod: `gpu_ids` wins over `num_gpus`
  (models/model.py:293-303; the reference raises when neither is given, a single replica is the
  default here)."""
  if 'gpu_ids' in config:
    return max(len(config['gpu_ids']), 1)
  return max(int(config.get('num_gpus', 1)), 1)


def plan_workers(config, environ=None):
  """How many ranks a config means and whether this process still has to launch them.

  The reference has two data-parallel modes (models/model.py:386-427): Horovod (`use_horovod
  True`: the ranks come from `mpirun -np N`, `num_gpus` is ignored) and tower mode (`use_horovod
  False, num_gpus N`: ONE process builds N replicas with shared variables, loss = mean over
  replicas, data layer i of N per replica). Here both are one process per GPU over RCCL, so
  tower mode with N > 1 means N ranks: if no launcher provided them (`WORLD_SIZE` unset) the
  caller re-executes itself under torch.distributed.run (`spawn` True); a launcher whose world
  size contradicts `num_gpus` is an error — a tower config never silently trains on fewer
  replicas than it names.

  Returns (world, spawn)."""
  env = os.environ if environ is None else environ
  launched = "WORLD_SIZE" in env
  world_env = int(env.get("WORLD_SIZE", "1"))
  if config.get('use_horovod', False):
    return world_env, False
  n = configured_towers(config)
  if n == 1:
    if launched and world_env > 1:
      raise ValueError("use_horovod is False and the config names one GPU, but the launcher "
                       "started %d ranks: set use_horovod True (or num_gpus %d)" % (world_env, world_env))
    return 1, False
  if not launched:
    return n, True
  if world_env != n:
    raise ValueError("config asks for num_gpus=%d replicas but the launcher started %d rank(s) "
                     "(WORLD_SIZE=%s)" % (n, world_env, env.get("WORLD_SIZE")))
  return n, False


def spawn_ranks(n, script, argv):
  """Re-execute `script argv` as n ranks of ONE node under torch.distributed.run (one process per
  GPU, RCCL over xGMI; rendezvous on 127.0.0.1 — the reference relies on an external
  `mpirun -np N`, run.py:43-49, or on in-process towers). Returns the launcher's exit code."""
  import socket
  import subprocess
  import sys
  with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
         "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(script)] + list(argv)
  return subprocess.call(cmd)


def broadcast_parameters(store, extra_tensors=(), root=0):
  """Rank-0 values for every variable (masters + non-trainable state)."""
  dist.broadcast(store.master, src=root)
  for t in extra_tensors:
    dist.broadcast(t, src=root)
  store.refresh_compute_copies()


class GradientReducer(object):
  """Bucketed all-reduce(SUM) of the flat fp32 gradient buffer on a side stream,
  overlapped with backward.

  Variables are created in forward order, so backward finalises their gradients from
  the END of the flat buffer towards the start. Each backward closure reports the
  lowest parameter offset it has finalised (`mark_done`); every bucket that lies
  entirely above the watermark is all-reduced immediately on the side stream while the
  remaining backward kernels keep running on the compute stream. `finish()` reduces
  whatever is left and makes the compute stream wait for the side stream."""

  def __init__(self, store, world_size, bucket_bytes=None):
    self.store, self.world = store, world_size
    # bucket size: 128 MB of fp32 gradients unless OS2S_BUCKET_MB says otherwise (bench.py --bucket-mb: the size
    # has never met xGMI — the first 8-GPU run can sweep it without a code change)
    if bucket_bytes is None:
      bucket_bytes = int(float(os.environ.get("OS2S_BUCKET_MB", "128")) * (1 << 20))
    self.bucket_bytes = bucket_bytes
    n = store.grads.numel()
    per = max(bucket_bytes // 4, store.chunk)
    per = (per // store.chunk) * store.chunk
    self.bounds = [(s, min(s + per, n)) for s in range(0, n, per)]
    self.stream = torch.cuda.Stream() if store.grads.is_cuda else None
    # OS2S_CHECK_REDUCER=1 (one-rank debugging aid): keep a copy of every bucket as it is handed
    # to the all-reduce and verify at finish() that no backward closure wrote to it afterwards —
    # i.e. that variables really become final in the order mark_done() assumes
    self.check = os.environ.get("OS2S_CHECK_REDUCER", "") == "1" and world_size == 1
    self._snap = []
    # OS2S_ALLREDUCE_DTYPE=bf16: the payload crosses xGMI as bf16 (half the bytes: 0.67 GB instead
    # of 1.33 GB per Jasper step); the loss-scaled fp32 gradients are rounded to bf16 before the
    # sum and the sum itself is rounded once more — NOT bit-compatible with the fp32 reduction
    # (relative error <= 2^-8 per element), off by default
    self.wire_dtype = torch.bfloat16 if os.environ.get("OS2S_ALLREDUCE_DTYPE", "fp32") == "bf16" \
        else torch.float32
    # timing=True: HIP events around every bucket's all-reduce on the side stream + the time the
    # compute stream sits in finish() waiting for it (bench.py --gpus N reads `pop_timing()`)
    self.timing = False
    self._events, self._exposed = [], []
    self.reset()

  def reset(self):
    self.watermark = self.store.grads.numel()
    self.next_bucket = len(self.bounds) - 1

  def _reduce(self, s, e):
    g = self.store.grads
    if self.stream is None:
      if self.check:
        self._snap.append((s, e, g[s:e].clone()))
      self._all_reduce(g[s:e])
      return
    from ..parts.cnns.conv_blocks import side_streams
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    self.stream.wait_event(ev)
    for st in side_streams():       # weight-gradient kernels enqueued off the main stream
      self.stream.wait_stream(st)
    with torch.cuda.stream(self.stream):
      if self.check:
        self._snap.append((s, e, g[s:e].clone()))
      if self.timing:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(self.stream)
      self._all_reduce(g[s:e])
      if self.timing:
        e1.record(self.stream)
        self._events.append((e0, e1, (e - s) * (2 if self.wire_dtype == torch.bfloat16 else 4)))

  def _all_reduce(self, view):
    if self.wire_dtype == torch.float32:
      dist.all_reduce(view, op=dist.ReduceOp.SUM)
      return
    wire = view.to(self.wire_dtype)
    dist.all_reduce(wire, op=dist.ReduceOp.SUM)
    view.copy_(wire)

  def mark_done(self, offset):
    """All gradients at flat offsets >= `offset` are final."""
    if offset < self.watermark:
      self.watermark = offset
    while self.next_bucket >= 0 and self.bounds[self.next_bucket][0] >= self.watermark:
      s, e = self.bounds[self.next_bucket]
      self._reduce(s, e)
      self.next_bucket -= 1

  def finish(self):
    self.mark_done(0)
    if self.stream is not None:
      if self.timing:     # how long the compute stream waits here = communication NOT hidden by backward
        w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        w0.record(torch.cuda.current_stream())
      torch.cuda.current_stream().wait_stream(self.stream)
      if self.timing:
        w1.record(torch.cuda.current_stream())
        self._exposed.append((w0, w1))
    if self.check:
      for s, e, snap in self._snap:
        if not torch.equal(snap, self.store.grads[s:e]):
          bad = (snap != self.store.grads[s:e]).nonzero()[0].item() + s
          names = [p.name for p in self.store.params if p.offset <= bad < p.offset + p.numel]
          raise RuntimeError("gradient bucket [%d, %d) was reduced before it was final (offset %d, %s)"
                             % (s, e, bad, names))
      self._snap = []
    self.reset()

  def pop_timing(self):
    """Per-bucket all-reduce durations and the exposed (non-overlapped) wait, in ms, since the last
    call; synchronises. {"buckets": [{"bytes", "ms", "GBps"}...] averaged per step, "steps",
    "allreduce_ms_per_step", "exposed_ms_per_step", "bus_GBps"} — bus bandwidth = the ring
    all-reduce convention 2 (N-1)/N x bytes / time, what one xGMI link has to carry."""
    if self.stream is None or not self._exposed:
      return None
    torch.cuda.synchronize()
    steps = len(self._exposed)
    nb = len(self.bounds)
    per = [[0.0, 0] for _ in range(nb)]
    for i, (e0, e1, nbytes) in enumerate(self._events):
      per[i % nb][0] += e0.elapsed_time(e1)
      per[i % nb][1] = nbytes
    total_ms = sum(p[0] for p in per) / steps
    total_bytes = sum(p[1] for p in per)
    exposed = sum(a.elapsed_time(b) for a, b in self._exposed) / steps
    f = 2.0 * (self.world - 1) / max(self.world, 1)
    out = {"steps": steps, "world_size": self.world, "wire_dtype": str(self.wire_dtype).replace("torch.", ""),
           "allreduce_dtype": "bf16" if self.wire_dtype == torch.bfloat16 else "fp32",
           "bucket_mb": self.bucket_bytes / float(1 << 20), "buckets_per_step": nb,
           "bucket_bytes": [p[1] for p in per],
           "bucket_ms": [p[0] / steps for p in per],
           "allreduce_bytes_per_step": total_bytes, "allreduce_ms_per_step": total_ms,
           "exposed_ms_per_step": exposed,
           "bus_GBps": f * total_bytes / (total_ms * 1e-3) / 1e9 if total_ms > 0 else None}
    self._events, self._exposed = [], []
    return out

  def all_reduce(self):
    """Non-overlapped form (everything after backward)."""
    self.reset()
    self.finish()


def gather_objects(obj, root=0):
  """collect_if_horovod(..., mode='gather') (utils/utils.py:47-82)."""
  if not dist.is_initialized() or dist.get_world_size() == 1:
    return [obj]
  out = [None] * dist.get_world_size() if dist.get_rank() == root else None
  dist.gather_object(obj, out, dst=root)
  return out

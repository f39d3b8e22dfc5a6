"""Thin Python wrappers: one function per C-ABI entry point of include/os2s.h.

Arguments are torch CUDA tensors (used as plain device buffers); every wrapper
validates dtype / contiguity / device, passes raw pointers + the current HIP
stream, and raises Os2sError on a non-zero status.
"""
import functools
import os

import torch

from . import _lib
from ._lib import c_void_p, c_int, c_int64, c_size_t, c_float, c_uint64


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
  """The current HIP stream of the current device as a raw handle. torch.cuda.current_stream() builds a Stream
  object through three Python layers (7 us a call, 850 calls = 6 ms of host time per Jasper step); the two C
  entry points underneath it return the same handle in well under a microsecond."""
  if _raw_stream is not None and _cur_device is not None:
    return c_void_p(_raw_stream(_cur_device()))
  return c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t, dtype=None, allow_none=False):
  """Device address of a checked tensor, as the plain int (None for NULL) a c_void_p parameter converts itself —
  no c_void_p object per argument on the launch path."""
  if t is None:
    if allow_none:
      return None
    raise ValueError("tensor required")
  if not t.is_cuda:
    raise _lib.Os2sError("os2s kernels need CUDA(HIP) tensors; got %s" % t.device)
  if dtype is not None and t.dtype is not dtype:
    raise TypeError("expected %s, got %s" % (dtype, t.dtype))
  if not t.is_contiguous():
    raise ValueError("tensor must be contiguous")
  return t.data_ptr()


_FN_CACHE = {}


def _fn(name, argtypes, restype=c_int):
  """The bound entry point `name` (bound once; every call site of a name passes the same argument types). A dict
  lookup by name: functools.lru_cache hashed the 10 - 25-element argtypes tuple on every call — 1 - 2 us for each
  of the ~2 000 launches of a step, on the thread whose pace bounds the launch-heavy models. The call sites read
  `_FN_CACHE.get(name) or _fn(name, argtypes)`: after the first call not even the argtypes tuple is built (0.2 - 0.4
  us of LOAD_GLOBALs per launch)."""
  f = _FN_CACHE.get(name)
  if f is None:
    f = _FN_CACHE[name] = _lib.bind(name, list(argtypes), restype)
  return f


def abi_version():
  return _lib.lib().os2s_abi_version()


def clock_probe_start(spin_cycles):
  """Enqueue the one-wave shader-clock probe (os2s_clock_probe) next to whatever runs on the other streams;
  returns the device buffer clock_probe_read() reads."""
  out = torch.zeros((2,), dtype=torch.int64, device="cuda")
  torch.cuda.current_stream().synchronize()      # the zero fill lands before the probe stream writes
  f = (_FN_CACHE.get("os2s_clock_probe") or _fn("os2s_clock_probe", (c_void_p, c_uint64)))
  _lib.check(f(_ptr(out), int(spin_cycles)), "os2s_clock_probe")
  return out


def clock_probe_read(out):
  """MHz of the shader clock over the probe's run (100 MHz reference counter), or None if it did not run."""
  _lib.check((_FN_CACHE.get("os2s_clock_probe_wait") or _fn("os2s_clock_probe_wait", ()))(), "os2s_clock_probe_wait")
  cyc, ref = (int(v) for v in out.cpu())
  return 100.0 * cyc / ref if ref > 0 else None


# --------------------------------------------------------------------------
# CTC greedy decode
# --------------------------------------------------------------------------
def ctc_greedy_decode(logits, seq_len, blank=None, merge_repeated=True):
  """logits [T,B,V] fp32 (time-major), seq_len [B] int32.

  Returns (ids [B,T] int32 padded with -1, lens [B] int32, neg_sum_logits [B] fp32).
  Mirrors tf.nn.ctc_greedy_decoder as used in fc_decoders.py:244-251.
  """
  T, B, V = logits.shape
  if blank is None:
    blank = V - 1
  dev = logits.device
  ids = torch.empty((B, T), dtype=torch.int32, device=dev)
  lens = torch.empty((B,), dtype=torch.int32, device=dev)
  neg = torch.empty((B,), dtype=torch.float32, device=dev)
  wsf = (_FN_CACHE.get("os2s_ctc_greedy_decode_workspace_bytes") or _fn("os2s_ctc_greedy_decode_workspace_bytes", (c_int, c_int), c_size_t))
  nbytes = int(wsf(T, B))
  ws = torch.empty((max(nbytes, 16),), dtype=torch.uint8, device=dev)
  f = (_FN_CACHE.get("os2s_ctc_greedy_decode") or _fn("os2s_ctc_greedy_decode",
          (c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
           c_void_p, c_void_p, c_void_p, c_void_p, c_size_t)))
  _lib.check(f(_stream(), _ptr(logits, torch.float32), _ptr(seq_len, torch.int32),
               T, B, V, int(blank), int(bool(merge_repeated)), _ptr(ids),
               _ptr(lens), _ptr(neg), _ptr(ws), nbytes),
             "os2s_ctc_greedy_decode")
  return ids, lens, neg


# --------------------------------------------------------------------------
# conv1d (implicit GEMM)
# --------------------------------------------------------------------------
c_ll = _lib.ctypes.c_longlong


def same_padding(tin, k, stride, dil):
  """TF 'SAME' padding (asymmetric when stride > 1): returns (tout, pad_left)."""
  tout = (tin + stride - 1) // stride
  total = max((tout - 1) * stride + (k - 1) * dil + 1 - tin, 0)
  return tout, total // 2


def valid_padding(tin, k, stride, dil):
  return (tin - (k - 1) * dil - 1) // stride + 1, 0


def conv1d_num_mtiles(B, tout):
  return int((_FN_CACHE.get("os2s_conv1d_num_mtiles") or _fn("os2s_conv1d_num_mtiles", (c_int, c_int)))(B, tout))


_conv_ws = {}


def set_deterministic(on):
  """Deterministic mode of the library (os2s_set_deterministic; default = environment
  OS2S_DETERMINISTIC): parameter-gradient kernels that use fp32 atomics across workgroups run in a
  single-contributor launch geometry — slower, bit-identical run to run."""
  (_FN_CACHE.get("os2s_set_deterministic") or _fn("os2s_set_deterministic", (c_int,), None))(int(bool(on)))


def deterministic():
  return bool((_FN_CACHE.get("os2s_deterministic") or _fn("os2s_deterministic", (), c_int))())


def upsample_rows(x, stride, tup):
  """[B, T, C] bf16 -> [B, tup, C] with x's rows at the multiples of `stride` and zeros between
  (os2s_upsample_rows_bf16: the output gradient of a strided convolution, made stride-1)."""
  B, T, C = x.shape
  y = torch.empty((B, tup, C), dtype=torch.bfloat16, device=x.device)
  f = (_FN_CACHE.get("os2s_upsample_rows_bf16") or _fn("os2s_upsample_rows_bf16", (c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p)))
  _lib.check(f(_stream(), _ptr(x, torch.bfloat16), B, T, C, int(stride), int(tup), _ptr(y, torch.bfloat16)),
             "os2s_upsample_rows_bf16")
  return y


def conv1d_set_host_lens(lens):
  """Hands the launcher a HOST copy (sequence of ints / numpy int32) of the lengths the following forward
  convolutions receive as in_len, or withdraws it (None). See include/os2s.h: a hint that saves the second
  (null) launch of the device-side tile choice; it cannot change results."""
  import ctypes
  f = (_FN_CACHE.get("os2s_conv1d_set_host_lens") or _fn("os2s_conv1d_set_host_lens", (c_void_p, c_int)))
  if lens is None:
    _lib.check(f(None, 0), "os2s_conv1d_set_host_lens")
    return
  arr = (ctypes.c_int32 * len(lens))(*[int(v) for v in lens])
  _lib.check(f(ctypes.cast(arr, c_void_p), len(lens)), "os2s_conv1d_set_host_lens")


def conv1d_workspace(device):
  """Caller-owned workspace of os2s_conv1d_fwd_ws (tickets zeroed once): one per (device,
  stream) — launches that may overlap must not share one."""
  if _raw_stream is not None and device.index is not None:
    key = (device.index, _raw_stream(device.index))
  else:
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
  ws = _conv_ws.get(key)
  if ws is None:
    n = int((_FN_CACHE.get("os2s_conv1d_workspace_bytes") or _fn("os2s_conv1d_workspace_bytes", (), c_size_t))())
    ws = torch.zeros((n,), dtype=torch.uint8, device=device)
    _conv_ws[key] = ws
  return ws


def conv1d_fwd(x, w, *, stride=1, dil=1, pad_left=None, tout=None, in_len=None,
               bias=None, stats=None, out=None, out_f32=False, accumulate=False,
               time_major=False, act=0, keep_prob=1.0, seed=0, residual=None, out_len=None,
               use_workspace=True):
  """x [B,Tin,Cin] bf16, w [K,Cout,Cin] bf16 -> y [B,Tout,Cout] (or [Tout,B,Cout]
  when time_major). pad_left/tout default to TF 'SAME'. use_workspace=False: no split of the
  last partial round of workgroups (same results up to fp32 summation order)."""
  B, Tin, Cin = x.shape
  K, Cout, Cin2 = w.shape
  assert Cin == Cin2
  if pad_left is None or tout is None:
    tout, pad_left = same_padding(Tin, K, stride, dil)
  dt = torch.float32 if out_f32 else torch.bfloat16
  if out is None:
    shape = (tout, B, Cout) if time_major else (B, tout, Cout)
    out = torch.empty(shape, dtype=dt, device=x.device)
  if time_major:
    ysb, yst = Cout, B * Cout
  else:
    ysb, yst = tout * Cout, Cout
  ws = conv1d_workspace(x.device) if use_workspace else None
  f = (_FN_CACHE.get("os2s_conv1d_fwd_ws") or _fn("os2s_conv1d_fwd_ws",
          (c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
           c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
           c_ll, c_ll, c_int, c_int, c_int, c_float, c_uint64, c_void_p, c_void_p,
           c_void_p, c_size_t)))
  _lib.check(f(_stream(), _ptr(x, torch.bfloat16), _ptr(w, torch.bfloat16),
               _ptr(out, dt), _ptr(in_len, torch.int32, True),
               _ptr(bias, torch.float32, True), _ptr(stats, torch.float32, True),
               B, Tin, Cin, Cout, K, stride, dil, pad_left, tout, ysb, yst,
               int(out_f32), int(accumulate), int(act), float(keep_prob),
               int(seed) & (2**64 - 1), _ptr(residual, torch.bfloat16, True),
               _ptr(out_len, torch.int32, True), _ptr(ws, None, True),
               ws.numel() if ws is not None else 0),
             "os2s_conv1d_fwd_ws")
  return out


class _ZeroArena(object):
  """fp32 scratch that must read as zeros when handed out (per-window statistic partials that dead
  windows never write): ONE fill per backward pass instead of one 7 us fill launch per layer. take()
  falls back to torch.zeros until reset() has seen the pass's total demand once."""

  def __init__(self):
    self.buf, self.cursor, self.demand = None, 0, 0

  def reset(self):
    if self.demand == 0 and self.buf is None:
      return
    if self.buf is None or self.demand > self.buf.numel():
      self.buf = torch.zeros(self.demand, dtype=torch.float32, device=self._device) if self.demand else None
    elif self.cursor:
      self.buf[:self.cursor].zero_()
    self.cursor, self.demand = 0, 0

  def take(self, shape, device):
    n = 1
    for d in shape:
      n *= int(d)
    n = (n + 63) & ~63
    self.demand += n
    self._device = device
    if self.buf is None or self.buf.device != device or self.cursor + n > self.buf.numel():
      return torch.zeros(shape, dtype=torch.float32, device=device)
    out = self.buf[self.cursor:self.cursor + n]
    self.cursor += n
    m = 1
    for d in shape:
      m *= int(d)
    return out[:m].view(shape)


# One arena per NESTING DEPTH of Tape.backward: a backward pass started from inside a closure of another one
# (depth 1, 2, ...) takes its zeroed scratch from its own arena and cannot re-zero the partials the outer pass
# still has in flight. Arenas persist across steps (they learn a pass's demand once).
_zero_arenas = [_ZeroArena()]
_zero_arena = _zero_arenas[0]


def zero_arena_enter(depth):
  """Start of a backward pass at nesting depth `depth` (Tape.backward): selects that depth's arena; every
  slice it handed out so far is dead (the pass that used them joined its side streams), re-zero what was
  used."""
  global _zero_arena
  while len(_zero_arenas) <= depth:
    _zero_arenas.append(_ZeroArena())
  _zero_arena = _zero_arenas[depth]
  _zero_arena.reset()


def zero_arena_leave(depth):
  """End of the pass at `depth`: the enclosing pass (if any) gets its arena back, untouched."""
  global _zero_arena
  _zero_arena = _zero_arenas[max(depth - 1, 0)]


def zero_arena_reset():
  zero_arena_enter(0)


def zero_scratch(shape, device):
  """Zeroed fp32 scratch that lives until the end of the running backward pass (the pass's arena)."""
  return _zero_arena.take(tuple(shape), device)


def conv1d_dgrad_bnact(dy, wt, dx, *, dil, pad_left, accumulate, out_len, mask_ref, mask_scale, stat_ref):
  """dx (+)= conv(dy, wt) (stride 1), then dz = (mask_ref > 0) ? dx * mask_scale : 0 written to dx;
  returns the BatchNorm-backward partials [num_mtiles(B, T), 2, C] = (sum dz, sum dz * stat_ref) per
  128-row window (os2s_conv1d_dgrad_bnact_ws). dy [B,Tin,Cin] bf16, wt [K,Cout,Cin] bf16 (tap-flipped
  transposed weights), dx / mask_ref / stat_ref [B,Tout,Cout] bf16 contiguous."""
  B, Tin, Cin = dy.shape
  K, Cout, _ = wt.shape
  Tout = dx.shape[1]
  assert tuple(dx.shape) == (B, Tout, Cout) == tuple(mask_ref.shape) == tuple(stat_ref.shape)
  assert dx.is_contiguous() and mask_ref.is_contiguous() and stat_ref.is_contiguous() and dy.is_contiguous()
  stats = _zero_arena.take((conv1d_num_mtiles(B, Tout), 2, Cout), dy.device)
  ws = conv1d_workspace(dy.device)
  f = (_FN_CACHE.get("os2s_conv1d_dgrad_bnact_ws") or _fn("os2s_conv1d_dgrad_bnact_ws",
          (c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
           c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_size_t)))
  _lib.check(f(_stream(), _ptr(dy, torch.bfloat16), _ptr(wt, torch.bfloat16), _ptr(dx, torch.bfloat16),
               _ptr(stats, torch.float32), B, Tin, Cin, Cout, K, int(dil), int(pad_left), Tout,
               int(bool(accumulate)), _ptr(out_len, torch.int32, True), _ptr(mask_ref, torch.bfloat16),
               float(mask_scale), _ptr(stat_ref, torch.bfloat16), _ptr(ws), ws.numel()),
             "os2s_conv1d_dgrad_bnact_ws")
  return stats


def bn_bwd_finalize_raw(partial, count, mean, rstd, dgamma, dbeta, accumulate, c1, c2):
  """dgamma / dbeta / c1 / c2 from raw partials [nparts, 2, C] = (sum dz, sum dz * y)."""
  nparts, two, C = partial.shape
  assert two == 2
  f = (_FN_CACHE.get("os2s_bn_bwd_finalize_raw") or _fn("os2s_bn_bwd_finalize_raw", (c_void_p, c_void_p, c_int, c_int, c_ll, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_int, c_void_p, c_void_p)))
  _lib.check(f(_stream(), _ptr(partial, torch.float32), nparts, C, int(count), _ptr(mean, torch.float32),
               _ptr(rstd, torch.float32), _ptr(dgamma, torch.float32, True), _ptr(dbeta, torch.float32, True),
               int(accumulate), _ptr(c1, torch.float32), _ptr(c2, torch.float32)), "os2s_bn_bwd_finalize_raw")


class _ConvGroup(_lib.ctypes.Structure):
  _fields_ = [("x", c_void_p), ("w", c_void_p), ("y", c_void_p), ("stats", c_void_p),
              ("Cin", c_int), ("Cout", c_int), ("accumulate", c_int)]


def conv1x1_fwd_grouped(items, in_len=None, out_len=None, out_f32=False):
  """items: list of dict(x [B,T,Cin] bf16, w [1,Cout,Cin] bf16, y [B,T,Cout] bf16, stats or None,
  accumulate) — up to 16 independent 1x1 convolutions over the same (B, T, lengths) in ONE
  launch (os2s_conv1x1_fwd_grouped); longer lists are cut into several launches. out_f32: every y is
  fp32 (os2s_conv1x1_fwd_grouped_ex; no statistics)."""
  B, T, _ = items[0]["x"].shape
  ydt = torch.float32 if out_f32 else torch.bfloat16
  f = (_FN_CACHE.get("os2s_conv1x1_fwd_grouped_ex") or _fn("os2s_conv1x1_fwd_grouped_ex",
          (c_void_p, _lib.ctypes.POINTER(_ConvGroup), c_int, c_void_p, c_void_p, c_int, c_int, c_int)))
  outs = [it["y"].data_ptr() for it in items]
  assert len(set(outs)) == len(outs), "two groups of one launch must not write the same tensor"
  for i0 in range(0, len(items), 16):
    part = items[i0:i0 + 16]
    arr = (_ConvGroup * len(part))()
    for g, it in zip(arr, part):
      x, w, y = it["x"], it["w"], it["y"]
      assert x.shape[0] == B and x.shape[1] == T and w.shape[0] == 1 and w.shape[2] == x.shape[2]
      assert tuple(y.shape) == (B, T, w.shape[1]) and x.is_contiguous() and y.is_contiguous()
      g.x, g.w, g.y = _ptr(x, torch.bfloat16), _ptr(w, torch.bfloat16), _ptr(y, ydt)
      g.stats = _ptr(it.get("stats"), torch.float32, True)
      g.Cin, g.Cout, g.accumulate = x.shape[2], w.shape[1], int(bool(it.get("accumulate", False)))
    _lib.check(f(_stream(), arr, len(part), _ptr(in_len, torch.int32, True),
                 _ptr(out_len, torch.int32, True), B, T, int(bool(out_f32))), "os2s_conv1x1_fwd_grouped_ex")


# --------------------------------------------------------------------------
# Dense-residual block ends without branch tensors (csrc/dense_residual.hip)
# --------------------------------------------------------------------------
def _rows_view_ok(t):
  """[B, T, C] bf16 whose rows are a channel slice of a wider [B, T, W] tensor (or contiguous)."""
  return t.dim() == 3 and t.stride(2) == 1 and t.stride(0) == t.shape[1] * t.stride(1) and t.stride(1) % 8 == 0


def dres_copy_cols(src, dst, lens=None, want_colsum=False):
  """dst[b, t, :] = t < lens[b] ? src[b, t, :] : 0 (both [B, T, C] bf16, either may be a channel-slice view);
  returns the per-block column sums [nparts, C] fp32 when asked."""
  B, T, C = src.shape
  assert tuple(dst.shape) == (B, T, C) and _rows_view_ok(src) and _rows_view_ok(dst)
  assert src.dtype == torch.bfloat16 and dst.dtype == torch.bfloat16
  part = None
  if want_colsum:
    n = int((_FN_CACHE.get("os2s_dres_copy_num_parts") or _fn("os2s_dres_copy_num_parts", (c_int, c_int)))(B, T))
    part = torch.empty((n, C), dtype=torch.float32, device=src.device)
  f = (_FN_CACHE.get("os2s_dres_copy_cols") or _fn("os2s_dres_copy_cols", (c_void_p, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_int, c_int, c_int, c_void_p)))
  _lib.check(f(_stream(), c_void_p(src.data_ptr()), src.stride(1), c_void_p(dst.data_ptr()), dst.stride(1),
               _ptr(lens, torch.int32, True), B, T, C, _ptr(part, torch.float32, True)), "os2s_dres_copy_cols")
  return part


def dres_cov(colsum_partial, gram, count, s, m, chl):
  """s, m [C] fp32 and chl [2C, C] bf16 (covariance hi / lo) from the column-sum partials and gram [C, C] fp32."""
  nparts, C = colsum_partial.shape
  assert tuple(gram.shape[-2:]) == (C, C) and chl.numel() == 2 * C * C and s.numel() == C and m.numel() == C
  f = (_FN_CACHE.get("os2s_dres_cov") or _fn("os2s_dres_cov", (c_void_p, c_void_p, c_int, c_void_p, c_int, c_ll, c_void_p, c_void_p, c_void_p)))
  _lib.check(f(_stream(), _ptr(colsum_partial, torch.float32), nparts, _ptr(gram, torch.float32), C, int(count),
               _ptr(s, torch.float32), _ptr(m, torch.float32), _ptr(chl, torch.bfloat16)), "os2s_dres_cov")


class _DresSeg(_lib.ctypes.Structure):
  _fields_ = [("w", c_void_p), ("tt", c_void_p), ("m", c_void_p), ("s", c_void_p), ("gamma", c_void_p),
              ("beta", c_void_p), ("moving_mean", c_void_p), ("moving_var", c_void_p), ("mean", c_void_p),
              ("rstd", c_void_p), ("dgamma", c_void_p), ("dbeta", c_void_p), ("dw", c_void_p), ("wd1", c_void_p),
              ("wd2", c_void_p), ("wt", c_void_p), ("ld", c_ll), ("c", c_int), ("koff", c_int)]


def dres_seg_table(segs, device):
  """segs: list of dict with the fields of os2s_dres_seg_t (tensors -> their data pointers; wd1 / wd2 / wt may be
  views at a column offset). Returns the table as a DEVICE uint8 tensor (the kernels read it from memory)."""
  import ctypes
  arr = (_DresSeg * len(segs))()
  for g, d in zip(arr, segs):
    for k, _ in _DresSeg._fields_:
      v = d[k]
      if k in ("ld", "c", "koff"):
        setattr(g, k, int(v))
      else:
        setattr(g, k, None if v is None else v.data_ptr())
  raw = bytes(ctypes.string_at(ctypes.addressof(arr), ctypes.sizeof(arr)))
  return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)


def dres_bn_fwd(table, nseg, Cout, Kk, wp, shift, count, eps, momentum, training):
  assert wp.numel() == Cout * Kk and shift.numel() == Cout and table.numel() == nseg * _lib.ctypes.sizeof(_DresSeg)
  f = (_FN_CACHE.get("os2s_dres_bn_fwd") or _fn("os2s_dres_bn_fwd", (c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_ll, c_float, c_float,
                               c_int)))
  _lib.check(f(_stream(), _ptr(table, torch.uint8), nseg, Cout, Kk, _ptr(wp, torch.bfloat16),
               _ptr(shift, torch.float32), int(count), float(eps), float(momentum), int(bool(training))),
             "os2s_dres_bn_fwd")


def dres_bn_bwd(table, nseg, Cout, Kk, P, mean_dz, count, coef):
  assert P.numel() == Cout * Kk and mean_dz.numel() == Cout and coef.numel() >= nseg * 4 * Cout
  f = (_FN_CACHE.get("os2s_dres_bn_bwd") or _fn("os2s_dres_bn_bwd", (c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_ll, c_void_p)))
  _lib.check(f(_stream(), _ptr(table, torch.uint8), nseg, Cout, Kk, _ptr(P, torch.float32),
               _ptr(mean_dz, torch.float32), int(count), _ptr(coef, torch.float32)), "os2s_dres_bn_bwd")


def conv1x1_cat_fwd(x, w, y, in_len=None, out_len=None, bias=None, accumulate=False):
  """y[b,t,:] (+)= x[b,t,:] . w^T (+ bias): x [B,T,Cin], y [B,T,Cout] bf16, either a channel-slice view of a wider
  tensor; w [Cout, Cin] bf16 contiguous (os2s_conv1x1_cat_fwd)."""
  B, T, Cin = x.shape
  Cout = w.shape[-2]
  assert w.shape[-1] == Cin and w.is_contiguous() and w.numel() == Cout * Cin and w.dtype == torch.bfloat16
  assert tuple(y.shape) == (B, T, Cout) and _rows_view_ok(x) and _rows_view_ok(y)
  assert x.dtype == torch.bfloat16 and y.dtype == torch.bfloat16
  ws = conv1d_workspace(x.device)
  f = (_FN_CACHE.get("os2s_conv1x1_cat_fwd") or _fn("os2s_conv1x1_cat_fwd", (c_void_p, c_void_p, c_ll, c_void_p, c_void_p, c_ll, c_void_p, c_void_p, c_void_p,
                                   c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t)))
  _lib.check(f(_stream(), c_void_p(x.data_ptr()), x.stride(1), c_void_p(w.data_ptr()), c_void_p(y.data_ptr()),
               y.stride(1), _ptr(in_len, torch.int32, True), _ptr(out_len, torch.int32, True),
               _ptr(bias, torch.float32, True), B, T, Cin, Cout, int(bool(accumulate)), _ptr(ws), ws.numel()),
             "os2s_conv1x1_cat_fwd")
  return y


def conv1d_wgrad(x, dy, K, *, stride=1, dil=1, pad_left=None, in_len=None,
                 out=None, accumulate=False, use_workspace=True):
  """x [B,Tin,Cin] bf16 (may be a channel-slice view of a wider tensor),
  dy [B,Tout,Cout] bf16 -> dW [K,Cout,Cin] fp32."""
  B, Tin, Cin = x.shape
  B2, Tout, Cout = dy.shape
  assert B == B2
  if pad_left is None:
    tout, pad_left = same_padding(Tin, K, stride, dil)
    assert tout == Tout
  if out is None:
    assert not accumulate
    out = torch.empty((K, Cout, Cin), dtype=torch.float32, device=x.device)
  assert x.stride(2) == 1 and x.stride(0) == Tin * x.stride(1)
  ws = conv1d_workspace(x.device) if use_workspace else None
  f = (_FN_CACHE.get("os2s_conv1d_wgrad_ws") or _fn("os2s_conv1d_wgrad_ws",
          (c_void_p, c_void_p, c_ll, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
           c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t)))
  _lib.check(f(_stream(), c_void_p(x.data_ptr()), x.stride(1), _ptr(dy, torch.bfloat16),
               _ptr(out, torch.float32), _ptr(in_len, torch.int32, True), B, Tin,
               Cin, Cout, K, stride, dil, pad_left, Tout, int(accumulate), _ptr(ws, None, True),
               ws.numel() if ws is not None else 0),
             "os2s_conv1d_wgrad_ws")
  return out


class _CWgradGroup(_lib.ctypes.Structure):
  _fields_ = [("x", c_void_p), ("dy", c_void_p), ("dw", c_void_p), ("x_row_stride", c_ll)]


def conv1d_wgrad_grouped(items, K, *, stride=1, dil=1, pad_left=None, in_len=None, accumulate=True):
  """items: list (<= 8) of dict(x [B,Tin,Cin] bf16, dy [B,Tout,Cout] bf16, dw [K,Cout,Cin] fp32) of ONE shape over
  one batch: their weight gradients in one launch (os2s_conv1d_wgrad_grouped_ws)."""
  n = len(items)
  assert 1 <= n <= 8
  x0, dy0 = items[0]["x"], items[0]["dy"]
  B, Tin, Cin = x0.shape
  _, Tout, Cout = dy0.shape
  if pad_left is None:
    tout, pad_left = same_padding(Tin, K, stride, dil)
    assert tout == Tout
  arr = (_CWgradGroup * n)()
  for g, it in zip(arr, items):
    x, dy, dw = it["x"], it["dy"], it["dw"]
    assert tuple(x.shape) == (B, Tin, Cin) and tuple(dy.shape) == (B, Tout, Cout) and tuple(dw.shape) == (K, Cout, Cin)
    assert x.stride(2) == 1 and x.stride(0) == Tin * x.stride(1) and x.stride(1) == x0.stride(1)
    g.x, g.dy, g.dw = c_void_p(x.data_ptr()), _ptr(dy, torch.bfloat16), _ptr(dw, torch.float32)
    g.x_row_stride = x.stride(1)
  ws = conv1d_workspace(x0.device)
  f = (_FN_CACHE.get("os2s_conv1d_wgrad_grouped_ws") or _fn("os2s_conv1d_wgrad_grouped_ws",
          (c_void_p, _lib.ctypes.POINTER(_CWgradGroup), c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
           c_int, c_int, c_int, c_int, c_void_p, c_size_t)))
  _lib.check(f(_stream(), arr, n, _ptr(in_len, torch.int32, True), B, Tin, Cin, Cout, K, stride, dil, pad_left, Tout,
               int(bool(accumulate)), _ptr(ws), ws.numel()), "os2s_conv1d_wgrad_grouped_ws")


# --------------------------------------------------------------------------
# BatchNorm (+ residual sum + activation + dropout + mask)
# --------------------------------------------------------------------------
ACT_NONE, ACT_RELU, ACT_TANH = 0, 1, 2


class _WgradGroup(_lib.ctypes.Structure):
  _fields_ = [("x", c_void_p), ("dy", c_void_p), ("dw", c_void_p), ("x_row_stride", c_ll),
              ("Cin", c_int), ("Cout", c_int)]


def conv1x1_wgrad_grouped(items, in_len=None, pingpong=True):
  """items: list of dict(x [B,T,Cin] bf16 (may be a channel slice), dy [B,T,Cout] bf16,
  dw [1,Cout,Cin] fp32 accumulated into): the K = 1 weight gradients of up to 16 branches over the
  same (B, T, in_len) in one launch."""
  n = len(items)
  assert 1 <= n <= 16
  B, T, _ = items[0]["x"].shape
  arr = (_WgradGroup * n)()
  for i, it in enumerate(items):
    x, dy, dw = it["x"], it["dy"], it["dw"]
    assert x.shape[:2] == (B, T) and dy.shape[:2] == (B, T) and x.dtype == torch.bfloat16
    assert x.stride(2) == 1 and x.stride(0) == T * x.stride(1) and dy.is_contiguous()
    assert dw.dtype == torch.float32 and dw.is_contiguous() and dw.numel() == dy.shape[2] * x.shape[2]
    g = arr[i]
    g.x, g.dy, g.dw = c_void_p(x.data_ptr()), _ptr(dy, torch.bfloat16), _ptr(dw, torch.float32)
    g.x_row_stride, g.Cin, g.Cout = x.stride(1), x.shape[2], dy.shape[2]
  if not pingpong:
    f = (_FN_CACHE.get("os2s_conv1x1_wgrad_grouped") or _fn("os2s_conv1x1_wgrad_grouped",
            (c_void_p, _lib.ctypes.POINTER(_WgradGroup), c_int, c_void_p, c_int, c_int)))
    _lib.check(f(_stream(), arr, n, _ptr(in_len, torch.int32, True), B, T), "os2s_conv1x1_wgrad_grouped")
    return
  # wide branches of a big batch: the K = 1 ping-pong TN-GEMM kernel (deterministic), else the lockstep kernel
  ws = conv1d_workspace(items[0]["x"].device)
  f = (_FN_CACHE.get("os2s_conv1x1_wgrad_grouped_ws") or _fn("os2s_conv1x1_wgrad_grouped_ws",
          (c_void_p, _lib.ctypes.POINTER(_WgradGroup), c_int, c_void_p, c_int, c_int, c_void_p, c_size_t)))
  _lib.check(f(_stream(), arr, n, _ptr(in_len, torch.int32, True), B, T, _ptr(ws), ws.numel()),
             "os2s_conv1x1_wgrad_grouped_ws")


def gemm_wgrad_grouped(items, accumulate=True):
  """items: list of dict(x [M,Cin] bf16, dy [M,Cout] bf16, dw [Cout,Cin] fp32): the Dense weight
  gradients dw (+)= dy^T x of up to 16 layers over the same M rows in one launch
  (os2s_gemm_wgrad_grouped: ping-pong tiles, deterministic)."""
  n = len(items)
  assert 1 <= n <= 16
  M = items[0]["x"].shape[0]
  arr = (_WgradGroup * n)()
  keep = []
  for g, it in zip(arr, items):
    x, dy, dw = it["x"], it["dy"], it["dw"]
    if dy.stride(1) != 1 or dy.stride(0) != dy.shape[1]:
      dy = dy.contiguous()
    if x.stride(1) != 1 or x.stride(0) % 8:
      x = x.contiguous()
    keep += [x, dy]
    assert x.shape[0] == M and dy.shape[0] == M and x.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16
    assert dw.dtype == torch.float32 and dw.is_contiguous() and tuple(dw.shape[-2:]) == (dy.shape[1], x.shape[1])
    g.x, g.dy, g.dw = c_void_p(x.data_ptr()), c_void_p(dy.data_ptr()), _ptr(dw, torch.float32)
    g.x_row_stride, g.Cin, g.Cout = x.stride(0), x.shape[1], dy.shape[1]
  ws = conv1d_workspace(items[0]["x"].device)
  f = (_FN_CACHE.get("os2s_gemm_wgrad_grouped") or _fn("os2s_gemm_wgrad_grouped",
          (c_void_p, _lib.ctypes.POINTER(_WgradGroup), c_int, c_ll, c_int, c_void_p, c_size_t)))
  _lib.check(f(_stream(), arr, n, M, int(bool(accumulate)), _ptr(ws), ws.numel()), "os2s_gemm_wgrad_grouped")


def _ptr_array(tensors, dtype):
  arr = (c_void_p * len(tensors))()
  for i, t in enumerate(tensors):
    arr[i] = _ptr(t, dtype)
  return arr


def bn_finalize(partial, count, gamma, beta, eps, momentum, training, moving_mean,
                moving_var, mean_out, rstd_out, scale_out, shift_out):
  C = scale_out.numel()
  nparts = 0 if partial is None else partial.shape[0]
  f = (_FN_CACHE.get("os2s_bn_finalize") or _fn("os2s_bn_finalize",
          (c_void_p, c_void_p, c_int, c_int, c_ll, c_void_p, c_void_p, c_float,
           c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p)))
  _lib.check(f(_stream(), _ptr(partial, torch.float32, True), nparts, C, int(count),
               _ptr(gamma, torch.float32, True), _ptr(beta, torch.float32, True),
               float(eps), float(momentum), int(training),
               _ptr(moving_mean, torch.float32, True), _ptr(moving_var, torch.float32, True),
               _ptr(mean_out, torch.float32, True), _ptr(rstd_out, torch.float32, True),
               _ptr(scale_out, torch.float32), _ptr(shift_out, torch.float32)),
             "os2s_bn_finalize")


def bn_finalize_multi(items, count, eps, momentum, training):
  """items: list (<= 16) of dict(partial, gamma, beta, moving_mean, moving_var, mean_out, rstd_out,
  scale_out, shift_out) of one geometry — os2s_bn_finalize for all of them in one launch."""
  J = len(items)
  C = items[0]["scale_out"].numel()
  nparts = 0 if items[0]["partial"] is None else items[0]["partial"].shape[0]
  keys = ("partial", "gamma", "beta", "moving_mean", "moving_var", "mean_out", "rstd_out", "scale_out", "shift_out")
  arrs = {}
  for k in keys:
    a = (c_void_p * J)()
    for j, it in enumerate(items):
      t = it.get(k)
      assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
      if k == "partial" and t is not None:
        assert t.shape[0] == nparts and t.shape[2] == C
      a[j] = None if t is None else t.data_ptr()
    arrs[k] = a
  P = _lib.ctypes.POINTER(c_void_p)
  f = (_FN_CACHE.get("os2s_bn_finalize_multi") or _fn("os2s_bn_finalize_multi", (c_void_p, c_int, P, c_int, c_int, c_ll, P, P, c_float, c_float, c_int,
                                     P, P, P, P, P, P)))
  _lib.check(f(_stream(), J, arrs["partial"], nparts, C, int(count), arrs["gamma"], arrs["beta"], float(eps),
               float(momentum), int(training), arrs["moving_mean"], arrs["moving_var"], arrs["mean_out"],
               arrs["rstd_out"], arrs["scale_out"], arrs["shift_out"]), "os2s_bn_finalize_multi")


def bn_stats(y2d):
  rows, C = y2d.shape
  n = int((_FN_CACHE.get("os2s_bn_stats_num_parts") or _fn("os2s_bn_stats_num_parts", (c_ll,)))(rows))
  partial = torch.empty((n, 2, C), dtype=torch.float32, device=y2d.device)
  f = (_FN_CACHE.get("os2s_bn_stats") or _fn("os2s_bn_stats", (c_void_p, c_void_p, c_ll, c_int, c_void_p)))
  _lib.check(f(_stream(), _ptr(y2d, torch.bfloat16), rows, C, _ptr(partial)), "os2s_bn_stats")
  return partial


def sample_norm_fwd(x, gamma, beta, mode, eps):
  """Per-sample normalisation of a [B, T, C] bf16 tensor (csrc/sample_norm.hip): mode 0 = instance norm (per
  sample and channel over T), 1 = layer norm over (T, C). Returns (z bf16, mean [B, C], rstd [B, C])."""
  B, T, C = x.shape
  assert x.is_contiguous() and x.dtype == torch.bfloat16
  dev = x.device
  z = torch.empty_like(x)
  mean = torch.empty((B, C), dtype=torch.float32, device=dev)
  rstd = torch.empty((B, C), dtype=torch.float32, device=dev)
  nf = int((_FN_CACHE.get("os2s_sample_norm_partial_floats") or _fn("os2s_sample_norm_partial_floats", (c_int, c_int), c_size_t))(B, C))
  part = torch.empty(nf, dtype=torch.float32, device=dev)
  f = (_FN_CACHE.get("os2s_sample_norm_fwd") or _fn("os2s_sample_norm_fwd", (c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float,
                                  c_void_p, c_void_p, c_void_p, c_void_p)))
  _lib.check(f(_stream(), _ptr(x, torch.bfloat16), _ptr(gamma, torch.float32), _ptr(beta, torch.float32), B, T, C,
               int(mode), float(eps), _ptr(z, torch.bfloat16), _ptr(mean, torch.float32), _ptr(rstd, torch.float32),
               _ptr(part, torch.float32)), "os2s_sample_norm_fwd")
  return z, mean, rstd


def sample_norm_bwd(dz, x, gamma, mean, rstd, mode, dgamma, dbeta):
  """Gradient of sample_norm_fwd: returns dx (bf16); dgamma / dbeta (fp32 [C]) are accumulated into."""
  B, T, C = x.shape
  assert x.is_contiguous() and dz.is_contiguous()
  dev = x.device
  dx = torch.empty_like(x)
  nf = int((_FN_CACHE.get("os2s_sample_norm_partial_floats") or _fn("os2s_sample_norm_partial_floats", (c_int, c_int), c_size_t))(B, C))
  part = torch.empty(nf, dtype=torch.float32, device=dev)
  scratch = torch.empty(4 * B * C, dtype=torch.float32, device=dev)
  f = (_FN_CACHE.get("os2s_sample_norm_bwd") or _fn("os2s_sample_norm_bwd", (c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                  c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p)))
  _lib.check(f(_stream(), _ptr(dz, torch.bfloat16), _ptr(x, torch.bfloat16), _ptr(gamma, torch.float32),
               _ptr(mean, torch.float32), _ptr(rstd, torch.float32), B, T, C, int(mode), _ptr(dx, torch.bfloat16),
               _ptr(dgamma, torch.float32), _ptr(dbeta, torch.float32), _ptr(part, torch.float32),
               _ptr(scratch, torch.float32)), "os2s_sample_norm_bwd")
  return dx


def bn_act_fwd(ys, scales, shifts, out, out_len, act, keep_prob, seed):
  B, T, C = out.shape
  J = len(ys)
  f = (_FN_CACHE.get("os2s_bn_act_fwd") or _fn("os2s_bn_act_fwd",
          (c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
           c_int, c_int, c_int, c_float, c_uint64)))
  _lib.check(f(_stream(), J, _ptr_array(ys, torch.bfloat16),
               _ptr_array(scales, torch.float32), _ptr_array(shifts, torch.float32),
               _ptr(out, torch.bfloat16), _ptr(out_len, torch.int32, True), B, T, C,
               int(act), float(keep_prob), int(seed) & (2**64 - 1)), "os2s_bn_act_fwd")
  return out


def bn_act_bwd_num_parts(rows):
  return int((_FN_CACHE.get("os2s_bn_act_bwd_num_parts") or _fn("os2s_bn_act_bwd_num_parts", (c_ll,)))(rows))


def bn_act_bwd_reduce(dout, out, ys, means, rstds, dz, partial, out_len, act, keep_prob,
                      seed):
  B, T, C = out.shape
  J = len(ys)
  f = (_FN_CACHE.get("os2s_bn_act_bwd_reduce") or _fn("os2s_bn_act_bwd_reduce",
          (c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
           c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_uint64)))
  _lib.check(f(_stream(), J, _ptr(dout, torch.bfloat16), _ptr(out, torch.bfloat16),
               _ptr_array(ys, torch.bfloat16), _ptr_array(means, torch.float32),
               _ptr_array(rstds, torch.float32), _ptr(dz, torch.bfloat16),
               _ptr(partial, torch.float32), _ptr(out_len, torch.int32, True), B, T, C,
               int(act), float(keep_prob), int(seed) & (2**64 - 1)),
             "os2s_bn_act_bwd_reduce")


def bn_bwd_finalize(partial, q, count, dgamma, dbeta, accumulate, c1, c2):
  nparts, nq, C = partial.shape
  f = (_FN_CACHE.get("os2s_bn_bwd_finalize") or _fn("os2s_bn_bwd_finalize",
          (c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_ll, c_void_p, c_void_p,
           c_int, c_void_p, c_void_p)))
  _lib.check(f(_stream(), _ptr(partial, torch.float32), nparts, nq, q, C, int(count),
               _ptr(dgamma, torch.float32, True), _ptr(dbeta, torch.float32, True),
               int(accumulate), _ptr(c1, torch.float32), _ptr(c2, torch.float32)),
             "os2s_bn_bwd_finalize")


def bn_bwd_finalize_multi(partial, count, dgammas, dbetas, accumulate, c1, c2):
  """All J inputs of a block end in one launch: c1, c2 are [J, C] (row j for input j)."""
  import ctypes
  nparts, nq, C = partial.shape
  J = nq - 1
  assert len(dgammas) == J and len(dbetas) == J and tuple(c1.shape) == (J, C) == tuple(c2.shape)
  arr = ctypes.c_void_p * J
  dg = arr(*[t.data_ptr() if t is not None else None for t in dgammas])
  db = arr(*[t.data_ptr() if t is not None else None for t in dbetas])
  f = (_FN_CACHE.get("os2s_bn_bwd_finalize_multi") or _fn("os2s_bn_bwd_finalize_multi",
          (c_void_p, c_void_p, c_int, c_int, c_int, c_ll, c_void_p, c_void_p, c_int, c_void_p, c_void_p)))
  _lib.check(f(_stream(), _ptr(partial, torch.float32), nparts, J, C, int(count), dg, db,
               int(accumulate), _ptr(c1, torch.float32), _ptr(c2, torch.float32)),
             "os2s_bn_bwd_finalize_multi")


def bn_bwd_apply(dz, y, gamma, mean, rstd, c1, c2, dy, out_len=None, margin=0, dz_to_len=False):
  """out_len / margin (dz [B,T,C]): rows t >= out_len[b] + margin are written as zeros unread —
  margin = (K-1)*dilation of the convolution whose gradients consume dy."""
  C = dz.shape[-1]
  rows = dz.numel() // C
  if out_len is not None:
    B, T = dz.shape[0], dz.shape[1]
    f = _fn("os2s_bn_bwd_apply_ragged_dz" if dz_to_len else "os2s_bn_bwd_apply_ragged",
            (c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
             c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int))
    _lib.check(f(_stream(), _ptr(dz, torch.bfloat16), _ptr(y, torch.bfloat16),
                 _ptr(gamma, torch.float32, True), _ptr(mean, torch.float32),
                 _ptr(rstd, torch.float32), _ptr(c1, torch.float32), _ptr(c2, torch.float32),
                 _ptr(dy, torch.bfloat16), _ptr(out_len, torch.int32), int(margin), B, T, C),
               "os2s_bn_bwd_apply_ragged")
    return dy
  assert not dz_to_len
  f = (_FN_CACHE.get("os2s_bn_bwd_apply") or _fn("os2s_bn_bwd_apply",
          (c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
           c_void_p, c_void_p, c_ll, c_int)))
  _lib.check(f(_stream(), _ptr(dz, torch.bfloat16), _ptr(y, torch.bfloat16),
               _ptr(gamma, torch.float32, True), _ptr(mean, torch.float32),
               _ptr(rstd, torch.float32), _ptr(c1, torch.float32), _ptr(c2, torch.float32),
               _ptr(dy, torch.bfloat16), rows, C), "os2s_bn_bwd_apply")
  return dy


def dropout_mask(seed, n_elems, keep_prob, device):
  """Test hook: boolean keep mask [n_elems] the fused dropout uses."""
  assert n_elems % 8 == 0
  out = torch.empty((n_elems // 8,), dtype=torch.uint8, device=device)
  f = (_FN_CACHE.get("os2s_dropout_mask") or _fn("os2s_dropout_mask", (c_void_p, c_uint64, c_ll, c_float, c_void_p)))
  _lib.check(f(_stream(), int(seed) & (2**64 - 1), n_elems // 8, float(keep_prob),
               _ptr(out)), "os2s_dropout_mask")
  bits = (out[:, None].to(torch.int32) >> torch.arange(8, device=device)[None, :]) & 1
  return bits.reshape(-1).bool()


# --------------------------------------------------------------------------
# CTC loss
# --------------------------------------------------------------------------
def ctc_loss(logits, in_len, labels, label_len, blank=None, grad_scale=1.0,
             want_grad=True, want_grad_bf16=False, vpad=32, grad_scale_dev=None):
  """logits [T,B,V] fp32. Returns dict(loss_per_sample, loss_mean, dlogits, dlogits_bf16)."""
  T, B, V = logits.shape
  Lmax = labels.shape[1]
  if blank is None:
    blank = V - 1
  dev = logits.device
  nbytes = int((_FN_CACHE.get("os2s_ctc_loss_workspace_bytes") or _fn("os2s_ctc_loss_workspace_bytes", (c_int, c_int, c_int, c_int),
                   c_size_t))(T, B, V, Lmax))
  ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
  lps = torch.empty((B,), dtype=torch.float32, device=dev)
  lm = torch.empty((1,), dtype=torch.float32, device=dev)
  dl = torch.empty((T, B, V), dtype=torch.float32, device=dev) if want_grad else None
  dl16 = (torch.empty((B, T, vpad), dtype=torch.bfloat16, device=dev)
          if want_grad_bf16 else None)
  f = (_FN_CACHE.get("os2s_ctc_loss") or _fn("os2s_ctc_loss",
          (c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
           c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
           c_void_p, c_size_t)))
  _lib.check(f(_stream(), _ptr(logits, torch.float32), _ptr(in_len, torch.int32),
               _ptr(labels, torch.int32), _ptr(label_len, torch.int32), T, B, V, Lmax,
               int(blank), float(grad_scale), _ptr(grad_scale_dev, torch.float32, True),
               _ptr(lps), _ptr(lm),
               _ptr(dl, None, True), _ptr(dl16, None, True), int(vpad), _ptr(ws), nbytes),
             "os2s_ctc_loss")
  return {"loss_per_sample": lps, "loss_mean": lm, "dlogits": dl, "dlogits_bf16": dl16}


# --------------------------------------------------------------------------
# optimizer
# --------------------------------------------------------------------------
class OptConfig(_lib.ctypes.Structure):
  """ctypes mirror of os2s_opt_config_t (include/os2s.h)."""
  _fields_ = [
      ("optimizer", c_int), ("beta1", c_float), ("beta2", c_float), ("epsilon", c_float),
      ("weight_decay", c_float), ("grad_averaging", c_int), ("lr_policy", c_int),
      ("learning_rate", c_float), ("min_lr", c_float), ("power", c_float),
      ("decay_rate", c_float), ("max_lr", c_float), ("coefficient", c_float),
      ("decay_steps", c_ll), ("begin_decay_at", c_ll), ("warmup_steps", c_ll),
      ("use_staircase_decay", c_int), ("d_model", c_int), ("has_max_lr", c_int),
      ("use_larc", c_int), ("larc_eta", c_float), ("larc_min_update", c_float),
      ("larc_epsilon", c_float), ("larc_mode_scale", c_int), ("clip_global_norm", c_float),
      ("scaler", c_int), ("scale_min", c_float), ("scale_max", c_float),
      ("step_factor", c_float), ("step_window", c_ll), ("log_max", c_float),
      ("lm_beta1", c_float), ("lm_beta2", c_float), ("overflow_std_dev", c_float),
      ("world_size", c_int), ("pw_count", c_int), ("pw_boundaries", c_ll * 16), ("pw_rates", c_float * 17),
      ("novograd_ema", c_int),
  ]


OPT_STATE_FIELDS = [
    ("global_step", "q"), ("scaler_iteration", "q"), ("last_overflow_iteration", "q"),
    ("num_skipped", "q"), ("loss_scale", "f"), ("lr", "f"), ("global_grad_norm", "f"),
    ("grad_amax", "f"), ("has_nan", "i"), ("skip", "i"), ("x_hat", "f"),
    ("slow_x_hat", "f"), ("xsquared_hat", "f"), ("b1_correction", "f"),
    ("b2_correction", "f"), ("grad_scale_latched", "f"),
]


def opt_chunk_elems():
  return int((_FN_CACHE.get("os2s_opt_chunk_elems") or _fn("os2s_opt_chunk_elems", ()))())


def opt_state_bytes():
  n = int((_FN_CACHE.get("os2s_opt_state_bytes") or _fn("os2s_opt_state_bytes", (), c_size_t))())
  assert int((_FN_CACHE.get("os2s_opt_config_bytes") or _fn("os2s_opt_config_bytes", (), c_size_t))()) == _lib.ctypes.sizeof(OptConfig), \
      "OptConfig ctypes mirror out of sync with os2s_opt_config_t"
  return n


def opt_init_state(state, loss_scale):
  f = (_FN_CACHE.get("os2s_opt_init_state") or _fn("os2s_opt_init_state", (c_void_p, c_void_p, c_float)))
  _lib.check(f(_stream(), _ptr(state, torch.uint8), float(loss_scale)), "os2s_opt_init_state")


def opt_read_state(state):
  """Host copy of the device optimizer state (synchronises)."""
  import struct
  raw = bytes(state.cpu().numpy().tobytes())
  fmt = "<" + "".join(c for _, c in OPT_STATE_FIELDS)
  vals = struct.unpack(fmt, raw[:struct.calcsize(fmt)])
  gru_xcd_check()            # the step's sync point: a persistent GRU launch that gave up shows here
  return dict(zip([n for n, _ in OPT_STATE_FIELDS], vals))


def gru_xcd_status(clear=False):
  """The sticky abort word of the persistent GRU kernels (csrc/rnn_xcd.hip): 0, or the OR of 1 = poll timeout,
  2 = workgroup placement. Definite after a stream synchronisation. clear=True resets it (the caller has
  discarded or redone the step)."""
  return int((_FN_CACHE.get("os2s_gru_xcd_status") or _fn("os2s_gru_xcd_status", (c_int,)))(int(bool(clear))))


def gru_xcd_launch_count():
  f = (_FN_CACHE.get("os2s_gru_xcd_launch_count") or _fn("os2s_gru_xcd_launch_count", (), c_ll))
  return int(f())


def gru_xcd_set_mode(mode):
  """0: recurrent layers use the launch-per-step kernels, 1: persistent kernels where supported, -1: default."""
  (_FN_CACHE.get("os2s_gru_xcd_set_mode") or _fn("os2s_gru_xcd_set_mode", (c_int,), None))(int(mode))


def gru_xcd_check(clear=False):
  """Raises if a persistent GRU launch (forward or backward) has given up since the word was last cleared:
  its outputs — and every gradient computed from them — are invalid. Model.train_step recovers from this by
  itself (the step is redone on the launch-per-step kernels); this check is for callers that drive the
  kernels directly. With clear=True the word is reset and the code returned instead of raised."""
  st = gru_xcd_status(clear)
  if st and not clear:
    raise _lib.Os2sError("a persistent GRU launch gave up (code %d: 1 = poll timeout, 2 = workgroup placement); "
                         "the step's results are invalid. OS2S_GRU_XCD=0 selects the launch-per-step path" % st)
  return st


def opt_step(cfg, state, grads, weights, m1, m2, w16, chunk_tensor, tensor_chunk_begin,
             tensor_l2, tensor_wd_mask, partial, gnorm2, wnorm2, amax, mult, tensor_v):
  nchunks = chunk_tensor.numel()
  ntensors = tensor_chunk_begin.numel() - 1
  f = (_FN_CACHE.get("os2s_opt_step") or _fn("os2s_opt_step",
          (c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
           c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
           c_void_p, c_void_p, c_void_p, c_void_p)))
  _lib.check(f(_stream(), _lib.ctypes.byref(cfg), _ptr(state, torch.uint8),
               _ptr(grads, torch.float32), _ptr(weights, torch.float32),
               _ptr(m1, torch.float32, True), _ptr(m2, torch.float32, True),
               _ptr(w16, torch.bfloat16, True), nchunks, ntensors,
               _ptr(chunk_tensor, torch.int32), _ptr(tensor_chunk_begin, torch.int32),
               _ptr(tensor_l2, torch.float32, True), _ptr(tensor_wd_mask, torch.float32, True),
               _ptr(partial, torch.float32), _ptr(gnorm2, torch.float32),
               _ptr(wnorm2, torch.float32), _ptr(amax, torch.float32),
               _ptr(mult, torch.float32), _ptr(tensor_v, torch.float32, True)),
             "os2s_opt_step")


def opt_prepare(cfg, state, grads, weights, chunk_tensor, tensor_chunk_begin, tensor_l2, partial, gnorm2, wnorm2,
                amax, mult, tensor_v):
  """First half of opt_step (os2s_opt_prepare): everything that needs all gradients."""
  nchunks = chunk_tensor.numel()
  ntensors = tensor_chunk_begin.numel() - 1
  f = (_FN_CACHE.get("os2s_opt_prepare") or _fn("os2s_opt_prepare",
          (c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
           c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p)))
  _lib.check(f(_stream(), _lib.ctypes.byref(cfg), _ptr(state, torch.uint8), _ptr(grads, torch.float32),
               _ptr(weights, torch.float32), nchunks, ntensors, _ptr(chunk_tensor, torch.int32),
               _ptr(tensor_chunk_begin, torch.int32), _ptr(tensor_l2, torch.float32, True),
               _ptr(partial, torch.float32), _ptr(gnorm2, torch.float32), _ptr(wnorm2, torch.float32),
               _ptr(amax, torch.float32), _ptr(mult, torch.float32), _ptr(tensor_v, torch.float32, True)),
             "os2s_opt_prepare")


def opt_apply_range(cfg, state, grads, weights, m1, m2, w16, chunk_begin, chunk_end, chunk_tensor, tensor_l2, mult,
                    zero_grads=False):
  """Second half (os2s_opt_apply_range): the update of chunks [chunk_begin, chunk_end) of the flat buffers."""
  f = (_FN_CACHE.get("os2s_opt_apply_range") or _fn("os2s_opt_apply_range",
          (c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
           c_void_p, c_void_p, c_void_p, c_void_p, c_int)))
  _lib.check(f(_stream(), _lib.ctypes.byref(cfg), _ptr(state, torch.uint8), _ptr(grads, torch.float32),
               _ptr(weights, torch.float32), _ptr(m1, torch.float32, True), _ptr(m2, torch.float32, True),
               _ptr(w16, torch.bfloat16, True), int(chunk_begin), int(chunk_end), _ptr(chunk_tensor, torch.int32),
               _ptr(tensor_l2, torch.float32, True), _ptr(mult, torch.float32), c_void_p(0), int(bool(zero_grads))),
             "os2s_opt_apply_range")


def cast_f32_to_bf16(src, dst):
  n = src.numel()
  f = (_FN_CACHE.get("os2s_cast_f32_to_bf16") or _fn("os2s_cast_f32_to_bf16", (c_void_p, c_void_p, c_void_p, c_ll)))
  _lib.check(f(_stream(), _ptr(src, torch.float32), _ptr(dst, torch.bfloat16), n),
             "os2s_cast_f32_to_bf16")


def conv_weight_dgrad_copy(w16, wt16, descs, total_tiles):
  ndesc = descs.shape[0]
  f = (_FN_CACHE.get("os2s_conv_weight_dgrad_copy") or _fn("os2s_conv_weight_dgrad_copy",
          (c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int)))
  _lib.check(f(_stream(), _ptr(w16, torch.bfloat16), _ptr(wt16, torch.bfloat16),
               _ptr(descs, torch.uint8), ndesc, int(total_tiles)),
             "os2s_conv_weight_dgrad_copy")


# --------------------------------------------------------------------------
# log-mel front end
# --------------------------------------------------------------------------
def logmel(signal, n_samples, window, mel_start, mel_len, mel_wt, *, hop, n_mels, tmax, tpad,
           preemph=0.97, dither=0.0, seed=0, fixed_gain=-1.0, log_floor=1e-20,
           norm_per_feature=True, want_f32=False, n_fft=512):
  """signal [B,Nmax] float32|int16 -> (features bf16 [B,tpad,n_mels], frames int32 [B], f32|None)."""
  B, Nmax = signal.shape
  dev = signal.device
  is_i16 = signal.dtype == torch.int16
  assert is_i16 or signal.dtype == torch.float32
  nbytes = int((_FN_CACHE.get("os2s_logmel_workspace_bytes") or _fn("os2s_logmel_workspace_bytes", (c_int, c_int, c_int), c_size_t))(B, tmax, n_mels))
  ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
  out = torch.empty((B, tpad, n_mels), dtype=torch.bfloat16, device=dev)
  out32 = torch.empty((B, tpad, n_mels), dtype=torch.float32, device=dev) if want_f32 else None
  olen = torch.empty((B,), dtype=torch.int32, device=dev)
  f = (_FN_CACHE.get("os2s_logmel") or _fn("os2s_logmel",
          (c_void_p, c_void_p, c_void_p, c_int, c_int, c_ll, c_int, c_int, c_int, c_void_p,
           c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_uint64, c_float, c_float,
           c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t)))
  _lib.check(f(_stream(), _ptr(signal), _ptr(n_samples, torch.int32), int(is_i16), B, Nmax,
               n_fft, hop, n_mels, _ptr(window, torch.float32), _ptr(mel_start, torch.int32),
               _ptr(mel_len, torch.int32), _ptr(mel_wt, torch.float32), mel_wt.shape[0],
               float(preemph), float(dither), int(seed) & (2**64 - 1), float(fixed_gain),
               float(log_floor), int(bool(norm_per_feature)), tmax, tpad, _ptr(out),
               _ptr(out32, None, True), _ptr(olen), _ptr(ws), nbytes), "os2s_logmel")
  return out, olen, out32


def psf_spectrogram(signal, n_samples, *, n_win, n_step, pad_to, num_features, tpad, want_f32=False):
  """signal [B,Nmax] float32|int16 -> (features bf16 [B,tpad,F], frames int32 [B], f32|None): the psf
  backend's 'spectrogram' features (os2s_psf_spectrogram)."""
  B, Nmax = signal.shape
  dev = signal.device
  is_i16 = signal.dtype == torch.int16
  assert is_i16 or signal.dtype == torch.float32
  nbytes = int((_FN_CACHE.get("os2s_psf_spectrogram_workspace_bytes") or _fn("os2s_psf_spectrogram_workspace_bytes", (c_int, c_int, c_int), c_size_t))(B, tpad, num_features))
  ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
  out = torch.empty((B, tpad, num_features), dtype=torch.bfloat16, device=dev)
  out32 = torch.empty((B, tpad, num_features), dtype=torch.float32, device=dev) if want_f32 else None
  olen = torch.empty((B,), dtype=torch.int32, device=dev)
  f = (_FN_CACHE.get("os2s_psf_spectrogram") or _fn("os2s_psf_spectrogram", (c_void_p, c_void_p, c_int, c_void_p, c_int, c_ll, c_int, c_int, c_int, c_int,
                                  c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t)))
  _lib.check(f(_stream(), _ptr(signal), int(is_i16), _ptr(n_samples, torch.int32), B, Nmax, n_win, n_step,
               pad_to, num_features, tpad, _ptr(out), _ptr(out32, None, True), _ptr(olen), _ptr(ws), nbytes),
             "os2s_psf_spectrogram")
  return out, olen, out32


def psf_logfbank(signal, n_samples, fb, *, n_win, n_step, pad_to, nfft, tpad, want_f32=False):
  """signal [B,Nmax] float32|int16, fb [nfilt, nfft/2+1] fp32 -> (features bf16 [B,tpad,nfilt], frames int32 [B],
  f32|None): the psf backend's 'logfbank' features (os2s_psf_logfbank)."""
  B, Nmax = signal.shape
  dev = signal.device
  is_i16 = signal.dtype == torch.int16
  assert is_i16 or signal.dtype == torch.float32
  nfilt = fb.shape[0]
  assert fb.is_contiguous() and fb.shape[1] == nfft // 2 + 1
  nbytes = int((_FN_CACHE.get("os2s_psf_spectrogram_workspace_bytes") or _fn("os2s_psf_spectrogram_workspace_bytes", (c_int, c_int, c_int), c_size_t))(B, tpad, nfilt))
  ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
  out = torch.empty((B, tpad, nfilt), dtype=torch.bfloat16, device=dev)
  out32 = torch.empty((B, tpad, nfilt), dtype=torch.float32, device=dev) if want_f32 else None
  olen = torch.empty((B,), dtype=torch.int32, device=dev)
  f = (_FN_CACHE.get("os2s_psf_logfbank") or _fn("os2s_psf_logfbank", (c_void_p, c_void_p, c_int, c_void_p, c_int, c_ll, c_int, c_int, c_int, c_int,
                               c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t)))
  _lib.check(f(_stream(), _ptr(signal), int(is_i16), _ptr(n_samples, torch.int32), B, Nmax, n_win, n_step,
               pad_to, nfilt, nfft, _ptr(fb, torch.float32), tpad, _ptr(out), _ptr(out32, None, True),
               _ptr(olen), _ptr(ws), nbytes), "os2s_psf_logfbank")
  return out, olen, out32


def augment_signal(signal, n_in, n_out, ratio, noise_amp, interp_win, num_table, nout_max, seed=0,
                   fixed_gain=-1.0):
  """signal [B,Nmax] int16|float32 -> normalised, speed-perturbed, noised fp32 [B,nout_max]."""
  B, Nmax = signal.shape
  dev = signal.device
  is_i16 = signal.dtype == torch.int16
  assert is_i16 or signal.dtype == torch.float32
  out = torch.empty((B, int(nout_max)), dtype=torch.float32, device=dev)
  scratch = torch.empty(B, dtype=torch.int32, device=dev)
  f = (_FN_CACHE.get("os2s_augment_signal") or _fn("os2s_augment_signal", (c_void_p, c_void_p, c_int, c_int, c_ll, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_float, c_void_p, c_int, c_int, c_uint64, c_void_p, c_void_p,
                                  c_ll)))
  _lib.check(f(_stream(), _ptr(signal), int(is_i16), B, Nmax, _ptr(n_in, torch.int32),
               _ptr(n_out, torch.int32), _ptr(ratio, torch.float64), _ptr(noise_amp, torch.float32, True),
               float(fixed_gain), _ptr(interp_win, torch.float32), interp_win.numel(), int(num_table),
               int(seed) & (2**64 - 1), _ptr(scratch), _ptr(out), int(nout_max)), "os2s_augment_signal")
  return out


def spec_augment(feats, masks):
  """feats bf16 [B,T,F] (in place); masks int32 [B, n_masks, 4] = (t0, t1, f0, f1)."""
  B, T, F = feats.shape
  f = (_FN_CACHE.get("os2s_spec_augment") or _fn("os2s_spec_augment", (c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int)))
  _lib.check(f(_stream(), _ptr(feats, torch.bfloat16), B, T, F, _ptr(masks, torch.int32), masks.shape[1]),
             "os2s_spec_augment")
  return feats


# --------------------------------------------------------------------------
# Transformer kernels (packed token-major tensors)
# --------------------------------------------------------------------------
BIG_TILE_MIN_ROWS = 256     # plain matmuls with at least this many rows go to the 256 x 256 ping-pong tile


def gemm(x2d, w, **kw):
  """x2d [N,Cin] bf16, w [Cout,Cin] bf16 -> [N,Cout]. A matmul with at most an fp32 bias (no activation /
  dropout / residual / fp32 output) with enough rows runs on the 256 x 256 ping-pong tile
  (os2s_gemm_nt); everything else is the K=1 case of the in-tree conv1d_fwd kernel."""
  plain = (all(kw.get(k) is None for k in ("residual", "stats", "in_len", "out_len"))
           and not kw.get("act", 0) and not kw.get("out_f32", False) and not kw.get("time_major", False)
           and kw.get("keep_prob", 1.0) >= 1.0)
  bias = kw.get("bias", None)
  if bias is not None and bias.dtype != torch.float32:
    plain = False           # the ping-pong kernel adds an fp32 bias in its epilogue
  if plain and x2d.shape[0] >= BIG_TILE_MIN_ROWS and x2d.stride(1) == 1 and w.stride(1) == 1:
    out_t = kw.get("out", None)
    if (x2d.shape[1] % 64 == 0 and w.shape[0] % 8 == 0 and w.is_contiguous() and x2d.stride(0) % 8 == 0
          and (out_t is None or (out_t.stride(1) == 1 and out_t.stride(0) % 8 == 0))):
      return gemm_nt(x2d, w, out=out_t, bias=bias, accumulate=bool(kw.get("accumulate", False)))
  out = kw.pop("out", None)
  N, Cin = x2d.shape
  Cout = w.shape[0]
  if out is not None:
    out = out.view(1, N, Cout)
  res = kw.pop("residual", None)
  if res is not None:
    res = res.view(1, N, Cout)
  y = conv1d_fwd(x2d.view(1, N, Cin), w.view(1, Cout, Cin), pad_left=0, tout=N, out=out,
                 residual=res, **kw)
  return y.view(N, Cout)


def gemm_nt(a, w, out=None, bias=None, act=0, keep_prob=1.0, seed=0, residual=None,
            accumulate=False, out_f32=False):
  """out[M,N] (+)= a[M,K] @ w[N,K]^T with the fused epilogue residual + dropout(act(. + bias));
  the hand-written MFMA GEMM (os2s_gemm_nt). a: bf16, row stride free; w: bf16 contiguous."""
  M, K = a.shape
  N, K2 = w.shape
  assert K == K2 and a.stride(1) == 1 and w.is_contiguous()
  if out is None:
    assert not accumulate
    out = torch.empty((M, N), dtype=torch.float32 if out_f32 else torch.bfloat16, device=a.device)
  assert out.stride(1) == 1 and tuple(out.shape) == (M, N)
  assert residual is None or (residual.stride(1) == 1 and residual.stride(0) == out.stride(0))
  ws = conv1d_workspace(a.device)
  f = (_FN_CACHE.get("os2s_gemm_nt_ws") or _fn("os2s_gemm_nt_ws", (c_void_p, c_void_p, c_ll, c_void_p, c_void_p, c_ll, c_int, c_int, c_int,
                             c_void_p, c_int, c_float, c_uint64, c_void_p, c_int, c_int, c_void_p, c_size_t)))
  _lib.check(f(_stream(), c_void_p(a.data_ptr()), a.stride(0), c_void_p(w.data_ptr()),
               c_void_p(out.data_ptr()), out.stride(0), M, N, K, _ptr(bias, torch.float32, True),
               int(act), float(keep_prob), int(seed) & (2**64 - 1),
               c_void_p(residual.data_ptr()) if residual is not None else c_void_p(0),
               int(bool(accumulate)), int(out.dtype == torch.float32), _ptr(ws), ws.numel()), "os2s_gemm_nt_ws")
  return out


def gemm_nt_mask(a, w, mask_ref, mask_scale, out=None, want_colsum=False):
  """out[M,N] = (a[M,K] @ w[N,K]^T) * (mask_ref > 0 ? mask_scale : 0) (os2s_gemm_nt_mask_ws): a data
  gradient with the ReLU + dropout backward of the producing layer in its epilogue. Returns
  (out, partials [ceil(M/128), 2, N] or None) — partials[:, 0] are column sums of `out`."""
  M, K = a.shape
  N, K2 = w.shape
  assert K == K2 and a.stride(1) == 1 and w.is_contiguous() and K % 64 == 0 and N % 8 == 0
  if out is None:
    out = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
  assert out.stride(1) == 1 and tuple(out.shape) == (M, N) and tuple(mask_ref.shape) == (M, N)
  assert mask_ref.stride(1) == 1 and mask_ref.stride(0) == out.stride(0) and mask_ref.dtype == torch.bfloat16
  part = torch.empty(((M + 127) // 128, 2, N), dtype=torch.float32, device=a.device) if want_colsum else None
  ws = conv1d_workspace(a.device)
  f = (_FN_CACHE.get("os2s_gemm_nt_mask_ws") or _fn("os2s_gemm_nt_mask_ws", (c_void_p, c_void_p, c_ll, c_void_p, c_void_p, c_ll, c_int, c_int, c_int,
                                  c_void_p, c_float, c_void_p, c_void_p, c_size_t)))
  _lib.check(f(_stream(), c_void_p(a.data_ptr()), a.stride(0), c_void_p(w.data_ptr()),
               c_void_p(out.data_ptr()), out.stride(0), M, N, K, c_void_p(mask_ref.data_ptr()),
               float(mask_scale), _ptr(part, torch.float32, True), _ptr(ws), ws.numel()),
             "os2s_gemm_nt_mask_ws")
  return out, part


def dense_epilogue(y, bias=None, act=0, keep_prob=1.0, seed=0, residual=None):
  """In place: y = residual + dropout(act(y + bias)) on bf16 [rows, C]."""
  rows, C = y.shape
  f = (_FN_CACHE.get("os2s_dense_epilogue") or _fn("os2s_dense_epilogue", (c_void_p, c_void_p, c_void_p, c_ll, c_int, c_int, c_float, c_uint64, c_void_p)))
  _lib.check(f(_stream(), _ptr(y, torch.bfloat16), _ptr(bias, torch.float32, True), rows, C, int(act),
               float(keep_prob), int(seed) & (2**64 - 1), _ptr(residual, torch.bfloat16, True)),
             "os2s_dense_epilogue")
  return y


def gemm_skinny(x2d, w, bias=None, relu=False, residual=None):
  """x2d [M,K] bf16 (row stride free), w [N,K] bf16 -> [M,N]; for small M (decoding steps)."""
  M, K = x2d.shape
  N = w.shape[0]
  y = torch.empty((M, N), dtype=torch.bfloat16, device=x2d.device)
  assert x2d.stride(1) == 1 and w.stride(1) == 1 and (residual is None or residual.stride(1) == 1)
  f = (_FN_CACHE.get("os2s_gemm_skinny") or _fn("os2s_gemm_skinny", (c_void_p, c_void_p, c_ll, c_void_p, c_ll, c_void_p, c_void_p, c_ll,
                               c_int, c_int, c_int, c_int, c_void_p, c_ll)))
  _lib.check(f(_stream(), c_void_p(x2d.data_ptr()), x2d.stride(0), c_void_p(w.data_ptr()), w.stride(0),
               _ptr(bias, torch.float32, True),
               c_void_p(residual.data_ptr()) if residual is not None else c_void_p(0),
               residual.stride(0) if residual is not None else 0, M, N, K, int(bool(relu)), _ptr(y), N),
             "os2s_gemm_skinny")
  return y


def gemm_wgrad(x2d, dy2d, out, accumulate=True):
  """dW [Cout,Cin] (+)= dy^T x (fp32)."""
  N, Cin = x2d.shape
  Cout = dy2d.shape[1]
  if dy2d.stride(1) != 1 or dy2d.stride(0) != Cout:
    dy2d = dy2d.contiguous()
  if x2d.stride(1) != 1 or x2d.stride(0) % 8:
    x2d = x2d.contiguous()
  conv1d_wgrad(x2d.unsqueeze(0), dy2d.unsqueeze(0), 1, pad_left=0,
               out=out.view(1, Cout, Cin), accumulate=accumulate)


def embed_fwd(ids, pos, table, emb_scale, keep_prob, seed, plain=False):
  """pos=None with plain=True: tf.nn.embedding_lookup (+ dropout)."""
  N = ids.numel()
  V, D = table.shape
  out = torch.empty((N, D), dtype=torch.bfloat16, device=ids.device)
  f = (_FN_CACHE.get("os2s_embed_fwd") or _fn("os2s_embed_fwd", (c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_ll,
                             c_float, c_float, c_uint64, c_void_p, c_int)))
  _lib.check(f(_stream(), _ptr(ids, torch.int32), _ptr(pos, torch.int32, plain),
               _ptr(table, torch.bfloat16), V, D, N, float(emb_scale), float(keep_prob),
               int(seed) & (2**64 - 1), _ptr(out), int(plain)), "os2s_embed_fwd")
  return out


def embed_bwd(ids, dout, dtable, emb_scale, keep_prob, seed, plain=False):
  N = ids.numel()
  V, D = dtable.shape
  f = (_FN_CACHE.get("os2s_embed_bwd") or _fn("os2s_embed_bwd", (c_void_p, c_void_p, c_void_p, c_int, c_int, c_ll, c_float,
                             c_float, c_uint64, c_void_p, c_int)))
  _lib.check(f(_stream(), _ptr(ids, torch.int32), _ptr(dout, torch.bfloat16), V, D, N,
               float(emb_scale), float(keep_prob), int(seed) & (2**64 - 1),
               _ptr(dtable, torch.float32), int(plain)), "os2s_embed_bwd")


def layernorm_fwd(x, gamma, beta, eps=1e-6, save=True):
  N, D = x.shape
  y = torch.empty_like(x)
  mean = torch.empty(N, dtype=torch.float32, device=x.device) if save else None
  rstd = torch.empty(N, dtype=torch.float32, device=x.device) if save else None
  f = (_FN_CACHE.get("os2s_layernorm_fwd") or _fn("os2s_layernorm_fwd", (c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_ll, c_int,
                                 c_void_p, c_void_p, c_void_p)))
  _lib.check(f(_stream(), _ptr(x, torch.bfloat16), _ptr(gamma, torch.float32),
               _ptr(beta, torch.float32), float(eps), N, D, _ptr(y),
               _ptr(mean, None, True), _ptr(rstd, None, True)), "os2s_layernorm_fwd")
  return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dres, dgamma, dbeta, defer_param_grads=False):
  """Returns dx (= dres + LN'(dy)); accumulates dgamma/dbeta — or, with defer_param_grads, returns
  (dx, finish) where finish() reduces the partials into dgamma/dbeta (the caller may run it on
  another stream: nothing in the rest of backward reads a parameter gradient)."""
  N, D = x.shape
  dx = torch.empty_like(x)
  nparts = int((_FN_CACHE.get("os2s_layernorm_bwd_num_parts") or _fn("os2s_layernorm_bwd_num_parts", (c_ll,)))(N))
  partial = torch.empty((nparts, 2, D), dtype=torch.float32, device=x.device)
  f = (_FN_CACHE.get("os2s_layernorm_bwd") or _fn("os2s_layernorm_bwd", (c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_ll, c_int, c_void_p, c_void_p)))
  _lib.check(f(_stream(), _ptr(dy, torch.bfloat16), _ptr(x, torch.bfloat16),
               _ptr(gamma, torch.float32), _ptr(mean, torch.float32), _ptr(rstd, torch.float32),
               _ptr(dres, torch.bfloat16, True), N, D, _ptr(dx), _ptr(partial)),
             "os2s_layernorm_bwd")
  def finish():
    scratch = torch.empty((2, D), dtype=torch.float32, device=x.device)
    bn_bwd_finalize(partial, 1, 1, dgamma, dbeta, True, scratch[0], scratch[1])
  if defer_param_grads:
    return dx, finish, partial
  finish()
  return dx


def dropout_bwd(dout, keep_prob, seed=0, out=None, capped=False):
  """mode 0 (hash mask) when out is None, mode 1 (relu+dropout via saved output) otherwise; capped: mode 2,
  the saved output is dropout(min(relu(.), 20)) — no gradient at the cap either."""
  d = torch.empty_like(dout)
  f = (_FN_CACHE.get("os2s_dropout_bwd") or _fn("os2s_dropout_bwd", (c_void_p, c_void_p, c_void_p, c_int, c_float, c_uint64, c_ll,
                               c_void_p)))
  _lib.check(f(_stream(), _ptr(dout, torch.bfloat16), _ptr(out, torch.bfloat16, True),
               0 if out is None else (2 if capped else 1), float(keep_prob), int(seed) & (2**64 - 1),
               dout.numel(), _ptr(d)), "os2s_dropout_bwd")
  return d


def dropout_bwd_colsum(dout2d, keep_prob, seed=0, out=None, capped=False):
  """dropout_bwd on a [rows, C] matrix + partial column sums of the result ([nparts, 2, C] fp32,
  plane 0; reduce with bn_bwd_finalize(q=1)): returns (d, partial)."""
  rows, C = dout2d.shape
  assert dout2d.is_contiguous()
  d = torch.empty_like(dout2d)
  n = int((_FN_CACHE.get("os2s_dropout_bwd_colsum_num_parts") or _fn("os2s_dropout_bwd_colsum_num_parts", (c_ll,)))(rows))
  partial = torch.empty((n, 2, C), dtype=torch.float32, device=dout2d.device)
  f = (_FN_CACHE.get("os2s_dropout_bwd_colsum") or _fn("os2s_dropout_bwd_colsum", (c_void_p, c_void_p, c_void_p, c_int, c_float, c_uint64, c_ll, c_int,
                                      c_void_p, c_void_p)))
  _lib.check(f(_stream(), _ptr(dout2d, torch.bfloat16), _ptr(out, torch.bfloat16, True),
               0 if out is None else (2 if capped else 1), float(keep_prob), int(seed) & (2**64 - 1), rows, C,
               _ptr(d), _ptr(partial)), "os2s_dropout_bwd_colsum")
  return d, partial


def add_bf16(a, b, out=None):
  out = torch.empty_like(a) if out is None else out
  f = (_FN_CACHE.get("os2s_add_bf16") or _fn("os2s_add_bf16", (c_void_p, c_void_p, c_void_p, c_ll, c_void_p)))
  _lib.check(f(_stream(), _ptr(a, torch.bfloat16), _ptr(b, torch.bfloat16), a.numel(),
               _ptr(out)), "os2s_add_bf16")
  return out


def attention_fwd(q, k, v, cu_q, cu_k, H, max_len, causal, scale, keep_prob=1.0, seed=0):
  """q [Nq, >=H*64] / k, v [Nk, ...] bf16 row-major views (row stride = .stride(0));
  returns (o [Nq, H*64], lse [Nq, H])."""
  Nq = q.shape[0]
  dh = 64
  B = cu_q.numel() - 1
  o = torch.empty((Nq, H * dh), dtype=torch.bfloat16, device=q.device)
  lse = torch.empty((Nq, H), dtype=torch.float32, device=q.device)
  f = (_FN_CACHE.get("os2s_attention_fwd") or _fn("os2s_attention_fwd",
          (c_void_p,) * 8 + (c_int, c_int, c_int, c_int, c_ll, c_ll, c_ll, c_ll, c_int, c_float,
                             c_float, c_uint64)))
  _lib.check(f(_stream(), c_void_p(q.data_ptr()), c_void_p(k.data_ptr()), c_void_p(v.data_ptr()),
               _ptr(o), _ptr(lse), _ptr(cu_q, torch.int32), _ptr(cu_k, torch.int32), B, H, dh,
               int(max_len), q.stride(0), k.stride(0), v.stride(0), o.stride(0), int(causal),
               float(scale), float(keep_prob), int(seed) & (2**64 - 1)), "os2s_attention_fwd")
  return o, lse


def attention_bwd(q, k, v, d_o, lse, dq, dk, dv, cu_q, cu_k, H, max_len, causal, scale,
                  keep_prob=1.0, seed=0):
  B = cu_q.numel() - 1
  f = (_FN_CACHE.get("os2s_attention_bwd") or _fn("os2s_attention_bwd",
          (c_void_p,) * 11 + (c_int, c_int, c_int, c_int) + (c_ll,) * 7 + (c_int, c_float, c_float,
                                                                         c_uint64)))
  _lib.check(f(_stream(), c_void_p(q.data_ptr()), c_void_p(k.data_ptr()), c_void_p(v.data_ptr()),
               _ptr(d_o, torch.bfloat16), _ptr(lse, torch.float32), c_void_p(dq.data_ptr()),
               c_void_p(dk.data_ptr()), c_void_p(dv.data_ptr()), _ptr(cu_q, torch.int32),
               _ptr(cu_k, torch.int32), B, H, 64, int(max_len), q.stride(0), k.stride(0),
               v.stride(0), d_o.stride(0), dq.stride(0), dk.stride(0), dv.stride(0), int(causal),
               float(scale), float(keep_prob), int(seed) & (2**64 - 1)), "os2s_attention_bwd")


def xent_smooth(logits, labels, label_smoothing, grad_scale_dev=None, want_grad=True,
                v_valid=None, grad_scale=None):
  """logits [N,V] bf16, labels int32 [N] -> (row_loss [N], loss [1], dlogits|None).
  loss = grad_scale * sum(row_loss) (default grad_scale = 1/N: the token mean)."""
  N, V = logits.shape
  dev = logits.device
  row_loss = torch.empty(N, dtype=torch.float32, device=dev)
  mean = torch.empty(1, dtype=torch.float32, device=dev)
  dl = torch.empty_like(logits) if want_grad else None
  f = (_FN_CACHE.get("os2s_xent_smooth") or _fn("os2s_xent_smooth", (c_void_p, c_void_p, c_void_p, c_ll, c_int, c_int, c_ll, c_float,
                               c_float, c_void_p, c_void_p, c_void_p, c_void_p)))
  _lib.check(f(_stream(), _ptr(logits, torch.bfloat16), _ptr(labels, torch.int32), N, V,
               V if v_valid is None else int(v_valid), logits.stride(0), float(label_smoothing),
               1.0 / N if grad_scale is None else float(grad_scale),
               _ptr(grad_scale_dev, torch.float32, True), _ptr(row_loss), _ptr(mean),
               _ptr(dl, None, True)), "os2s_xent_smooth")
  return row_loss, mean, dl


def argmax_rows(x2d, v_valid=None):
  """bf16 [N, ld] -> int32 [N] argmax over the first v_valid columns."""
  N, V = x2d.shape
  out = torch.empty(N, dtype=torch.int32, device=x2d.device)
  f = (_FN_CACHE.get("os2s_argmax_rows") or _fn("os2s_argmax_rows", (c_void_p, c_void_p, c_ll, c_int, c_ll, c_void_p)))
  _lib.check(f(_stream(), _ptr(x2d, torch.bfloat16), N, V if v_valid is None else int(v_valid),
               x2d.stride(0), _ptr(out)), "os2s_argmax_rows")
  return out


# --------------------------------------------------------------------------
# Transformer beam search / incremental decoding
# --------------------------------------------------------------------------
class BeamState(object):
  """Device-resident loop state of SequenceBeamSearch (see include/os2s.h)."""

  def __init__(self, initial_ids, beam, vocab_size, max_decode_length, alpha, eos_id, debug=False):
    import numpy as np
    dev = initial_ids.device
    B = int(initial_ids.shape[0])
    self.B, self.beam, self.V, self.max_len, self.eos = B, int(beam), int(vocab_size), int(max_decode_length), int(eos_id)
    L1 = self.max_len + 1
    self.status = torch.zeros(4, dtype=torch.int32, device=dev)
    self.alive_seq = torch.empty((2, B, beam, L1), dtype=torch.int32, device=dev)
    self.fin_seq = torch.empty((2, B, beam, L1), dtype=torch.int32, device=dev)
    self.alive_lp = torch.empty((B, beam), dtype=torch.float32, device=dev)
    self.fin_scores = torch.empty((B, beam), dtype=torch.float32, device=dev)
    self.fin_flags = torch.empty((B, beam), dtype=torch.int32, device=dev)
    self.parent = torch.zeros(B * beam, dtype=torch.int32, device=dev)
    lengths = np.arange(L1, dtype=np.float32)
    lnorm = np.power(((np.float32(5.0) + lengths) / np.float32(6.0)).astype(np.float32), np.float32(alpha)).astype(np.float32)
    self.lnorm = torch.from_numpy(lnorm).to(dev)
    wsb = (_FN_CACHE.get("os2s_beam_workspace_bytes") or _fn("os2s_beam_workspace_bytes", (c_int, c_int, c_int), c_ll))
    self.ws = torch.empty(max(int(wsb(B, beam, self.V)), 16), dtype=torch.uint8, device=dev)
    self.topk_lp = torch.zeros((B, 2 * beam), dtype=torch.float32, device=dev) if debug else None
    self.topk_idx = torch.zeros((B, 2 * beam), dtype=torch.int32, device=dev) if debug else None
    self.last_ids = torch.empty(B * beam, dtype=torch.int32, device=dev)
    self.pos = torch.empty(B * beam, dtype=torch.int32, device=dev)
    self.initial_ids = initial_ids
    self.reset()

  def reset(self):
    """_create_initial_state."""
    f = (_FN_CACHE.get("os2s_beam_init") or _fn("os2s_beam_init", (c_void_p, c_int, c_int, c_int) + (c_void_p,) * 9))
    _lib.check(f(_stream(), self.B, self.beam, self.max_len, _ptr(self.initial_ids, torch.int32),
                 _ptr(self.status), _ptr(self.alive_seq), _ptr(self.fin_seq), _ptr(self.alive_lp),
                 _ptr(self.fin_scores), _ptr(self.fin_flags), _ptr(self.last_ids), _ptr(self.pos)),
               "os2s_beam_init")

  def alive_ids(self, i):
    """[B*beam, i+1] view of the alive sequences at loop index i."""
    return self.alive_seq[i & 1].view(self.B * self.beam, -1)[:, :i + 1]

  def step(self, logits):
    """One _search_step on logits [B*beam, >=V] (bf16 or fp32, row-contiguous)."""
    N = self.B * self.beam
    assert logits.shape[0] == N and logits.stride(1) == 1 and logits.shape[1] >= self.V
    if logits.dtype not in (torch.float32, torch.bfloat16):
      raise TypeError("logits must be fp32 or bf16")
    f = (_FN_CACHE.get("os2s_beam_step") or _fn("os2s_beam_step", (c_void_p, c_void_p, c_int, c_ll, c_int, c_int, c_int, c_int, c_int)
            + (c_void_p,) * 13))
    _lib.check(f(_stream(), c_void_p(logits.data_ptr()), int(logits.dtype == torch.float32),
                 logits.stride(0), self.B, self.beam, self.V, self.max_len, self.eos, _ptr(self.lnorm),
                 _ptr(self.status), _ptr(self.alive_seq), _ptr(self.fin_seq), _ptr(self.alive_lp),
                 _ptr(self.fin_scores), _ptr(self.fin_flags), _ptr(self.parent),
                 _ptr(self.topk_lp, allow_none=True), _ptr(self.topk_idx, allow_none=True),
                 _ptr(self.last_ids), _ptr(self.pos), _ptr(self.ws)), "os2s_beam_step")

  def read_status(self):
    s = self.status.cpu()
    return bool(s[0]), int(s[1])

  def finalize(self):
    out_seq = torch.empty((self.B, self.beam, self.max_len + 1), dtype=torch.int32, device=self.status.device)
    out_scores = torch.empty((self.B, self.beam), dtype=torch.float32, device=self.status.device)
    f = (_FN_CACHE.get("os2s_beam_finalize") or _fn("os2s_beam_finalize", (c_void_p, c_int, c_int, c_int) + (c_void_p,) * 8))
    _lib.check(f(_stream(), self.B, self.beam, self.max_len, _ptr(self.status), _ptr(self.alive_seq),
                 _ptr(self.fin_seq), _ptr(self.alive_lp), _ptr(self.fin_scores), _ptr(self.fin_flags),
                 _ptr(out_seq), _ptr(out_scores)), "os2s_beam_finalize")
    return out_seq, out_scores


class TfBeamState(object):
  """State of tf.contrib.seq2seq.BeamSearchDecoder on the device (os2s_tf_beam_step)."""

  def __init__(self, B, beam, vocab_size, eos_id, length_penalty_weight, device):
    self.B, self.beam, self.V, self.eos, self.lpw = int(B), int(beam), int(vocab_size), int(eos_id), float(length_penalty_weight)
    N = self.B * self.beam
    lp = torch.full((self.B, self.beam), float("-inf"), dtype=torch.float32)
    lp[:, 0] = 0.0
    self.log_probs = lp.reshape(-1).to(device)
    self.finished = torch.zeros(N, dtype=torch.int32, device=device)
    self.lengths = torch.zeros(N, dtype=torch.int32, device=device)
    self.word_ids = torch.zeros(N, dtype=torch.int32, device=device)
    self.parent = torch.zeros(N, dtype=torch.int32, device=device)
    self.scores = torch.zeros(N, dtype=torch.float32, device=device)
    wsb = (_FN_CACHE.get("os2s_tf_beam_workspace_bytes") or _fn("os2s_tf_beam_workspace_bytes", (c_int, c_int), c_ll))
    self.ws = torch.empty(int(wsb(self.B, self.beam)), dtype=torch.uint8, device=device)

  def step(self, logits, time):
    N = self.B * self.beam
    assert logits.shape[0] == N and logits.stride(1) == 1 and logits.shape[1] >= self.V
    f = (_FN_CACHE.get("os2s_tf_beam_step") or _fn("os2s_tf_beam_step", (c_void_p, c_void_p, c_int, c_ll, c_int, c_int, c_int, c_int, c_int, c_float)
            + (c_void_p,) * 7))
    _lib.check(f(_stream(), c_void_p(logits.data_ptr()), int(logits.dtype == torch.float32), logits.stride(0),
                 self.B, self.beam, self.V, self.eos, int(time), self.lpw, _ptr(self.log_probs),
                 _ptr(self.finished), _ptr(self.lengths), _ptr(self.word_ids), _ptr(self.parent),
                 _ptr(self.scores), _ptr(self.ws)), "os2s_tf_beam_step")


def gather_rows(src, idx, enable=None, out=None):
  """out[r] = src[idx[r]] along dim 0 (rows of >= 4 bytes, multiple of 4)."""
  assert src.is_contiguous() and idx.dtype == torch.int32
  rows = int(idx.shape[0])
  row_bytes = src.element_size() * (src.numel() // max(src.shape[0], 1))
  if out is None:
    out = torch.empty((rows,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
  f = (_FN_CACHE.get("os2s_gather_rows") or _fn("os2s_gather_rows", (c_void_p, c_void_p, c_void_p, c_ll, c_ll, c_void_p, c_void_p)))
  _lib.check(f(_stream(), _ptr(src), _ptr(idx), rows, row_bytes, _ptr(enable, allow_none=True),
               _ptr(out)), "os2s_gather_rows")
  return out


def decode_self_attention(q, knew, vnew, kcache, vcache, ancestry, H, step, scale, status=None):
  """q/knew/vnew: bf16 [N, D] column slices of one row-major buffer; caches [N, Tmax, D]."""
  N, D = q.shape
  Tmax = kcache.shape[1]
  o = torch.empty((N, D), dtype=torch.bfloat16, device=q.device)
  assert knew.stride(0) == vnew.stride(0) and q.stride(1) == 1 and knew.stride(1) == 1
  f = (_FN_CACHE.get("os2s_decode_self_attention") or _fn("os2s_decode_self_attention", (c_void_p, c_void_p, c_ll, c_void_p, c_void_p, c_ll, c_void_p,
                                         c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                         c_float, c_void_p, c_ll)))
  _lib.check(f(_stream(), c_void_p(q.data_ptr()), q.stride(0), c_void_p(knew.data_ptr()),
               c_void_p(vnew.data_ptr()), knew.stride(0), _ptr(kcache, torch.bfloat16),
               _ptr(vcache, torch.bfloat16), _ptr(ancestry, torch.int32), N, H, D // H, Tmax, int(step),
               _ptr(status, allow_none=True), float(scale), _ptr(o), D), "os2s_decode_self_attention")
  return o


def decode_cross_attention(q, k, v, cu_k, beam, H, max_len, scale):
  """q bf16 [N, D]; k, v: column slices [N_src, D] of the packed encoder projections."""
  N, D = q.shape
  o = torch.empty((N, D), dtype=torch.bfloat16, device=q.device)
  assert k.stride(0) == v.stride(0) and k.stride(1) == 1 and q.stride(1) == 1
  f = (_FN_CACHE.get("os2s_decode_cross_attention") or _fn("os2s_decode_cross_attention", (c_void_p, c_void_p, c_ll, c_void_p, c_void_p, c_ll, c_void_p,
                                          c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_ll)))
  _lib.check(f(_stream(), c_void_p(q.data_ptr()), q.stride(0), c_void_p(k.data_ptr()),
               c_void_p(v.data_ptr()), k.stride(0), _ptr(cu_k, torch.int32), int(beam), N, H, D // H,
               int(max_len), float(scale), _ptr(o), D), "os2s_decode_cross_attention")
  return o


# --------------------------------------------------------------------------
# recurrent layers
# --------------------------------------------------------------------------
CELL_GRU_CUDNN, CELL_LSTM_CUDNN, CELL_LSTM_TF = 0, 1, 2


class _RnnDirFwd(_lib.ctypes.Structure):
  _fields_ = [("gx", c_void_p), ("wh", c_void_p), ("bh", c_void_p), ("y", c_void_p),
              ("ldy", c_ll), ("gates", c_void_p), ("c_seq", c_void_p), ("reverse", c_int)]


class _RnnDirBwd(_lib.ctypes.Structure):
  _fields_ = [("whT", c_void_p), ("dy", c_void_p), ("lddy", c_ll), ("y", c_void_p), ("ldy", c_ll),
              ("gates", c_void_p), ("c_seq", c_void_p), ("dgx", c_void_p), ("dgr", c_void_p),
              ("reverse", c_int)]


def _addr(t):
  return None if t is None else t.data_ptr()


def rnn_layer_fwd_multi(cell, dirs, lens, H, forget_bias=1.0, save=True):
  """dirs: list (1 or 2) of dict(gx [B,T,G*H] bf16, wh [G*H,H] bf16, bh fp32|None, y view|None,
  reverse). All directions advance in one launch per time step. Returns a list of
  (y, gates|None, c_seq|None). A `y` may be a [B,T,H] channel-slice VIEW of a wider
  [B,T,ld] tensor (both directions of a layer share one buffer)."""
  B, T, GH = dirs[0]["gx"].shape
  dev = dirs[0]["gx"].device
  nd = len(dirs)
  arr = (_RnnDirFwd * nd)()
  outs = []
  for i, d in enumerate(dirs):
    gx = d["gx"]
    assert gx.dtype == torch.bfloat16 and gx.is_contiguous() and gx.shape == (B, T, GH)
    y = d.get("y")
    if y is None:
      y = (torch.zeros if lens is not None else torch.empty)((B, T, H), dtype=torch.bfloat16,
                                                             device=dev)
    assert y.stride(2) == 1 and y.stride(0) == T * y.stride(1)
    gates = torch.empty((B, T, 4 * H), dtype=torch.bfloat16, device=dev) if save else None
    c_seq = (torch.empty((B, T, H), dtype=torch.float32, device=dev)
             if (save and cell != CELL_GRU_CUDNN) else None)
    arr[i].gx, arr[i].wh = _ptr(gx, torch.bfloat16), _ptr(d["wh"], torch.bfloat16)
    arr[i].bh = _ptr(d.get("bh"), torch.float32, True)
    arr[i].y, arr[i].ldy = y.data_ptr(), y.stride(1)
    arr[i].gates, arr[i].c_seq = _addr(gates), _addr(c_seq)
    arr[i].reverse = int(bool(d["reverse"]))
    outs.append((y, gates, c_seq))
  n = nd * int((_FN_CACHE.get("os2s_rnn_fwd_workspace_bytes") or _fn("os2s_rnn_fwd_workspace_bytes", (c_int, c_int), c_size_t))(B, H))
  ws = torch.empty((n,), dtype=torch.uint8, device=dev)
  f = (_FN_CACHE.get("os2s_rnn_layer_fwd_multi") or _fn("os2s_rnn_layer_fwd_multi", (c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                                       c_int, c_float, c_void_p, c_size_t)))
  _lib.check(f(_stream(), int(cell), nd, _lib.ctypes.byref(arr), _ptr(lens, torch.int32, True), B, T,
               H, float(forget_bias), _ptr(ws), n), "os2s_rnn_layer_fwd_multi")
  return outs


def rnn_layer_fwd(cell, gx, wh, bh, lens, H, reverse, forget_bias=1.0, save=True, y=None):
  return rnn_layer_fwd_multi(cell, [dict(gx=gx, wh=wh, bh=bh, y=y, reverse=reverse)], lens, H,
                             forget_bias, save)[0]


def rnn_layer_bwd_multi(cell, dirs, lens, H, forget_bias=1.0):
  """dirs: list of dict(whT, dy, y, gates, c_seq, reverse); dy / y may be channel-slice views.
  Returns a list of (dgx [B,T,G*H], dgr (GRU) or dgx again (LSTM))."""
  B, T, _ = dirs[0]["dy"].shape
  G = 3 if cell == CELL_GRU_CUDNN else 4
  dev = dirs[0]["dy"].device
  nd = len(dirs)
  arr = (_RnnDirBwd * nd)()
  outs = []
  for i, d in enumerate(dirs):
    dy, y = d["dy"], d["y"]
    dgx = torch.empty((B, T, G * H), dtype=torch.bfloat16, device=dev)
    dgr = torch.empty_like(dgx) if cell == CELL_GRU_CUDNN else None
    arr[i].whT = _ptr(d["whT"], torch.bfloat16)
    arr[i].dy, arr[i].lddy = dy.data_ptr(), dy.stride(1)
    arr[i].y, arr[i].ldy = y.data_ptr(), y.stride(1)
    arr[i].gates = _ptr(d["gates"], torch.bfloat16)
    arr[i].c_seq = _ptr(d.get("c_seq"), torch.float32, True)
    arr[i].dgx, arr[i].dgr = dgx.data_ptr(), _addr(dgr)
    arr[i].reverse = int(bool(d["reverse"]))
    outs.append((dgx, dgr if dgr is not None else dgx))
  n = nd * int((_FN_CACHE.get("os2s_rnn_bwd_workspace_bytes") or _fn("os2s_rnn_bwd_workspace_bytes", (c_int, c_int), c_size_t))(B, H))
  ws = torch.empty((n,), dtype=torch.uint8, device=dev)
  f = (_FN_CACHE.get("os2s_rnn_layer_bwd_multi") or _fn("os2s_rnn_layer_bwd_multi", (c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                                       c_int, c_float, c_void_p, c_size_t)))
  _lib.check(f(_stream(), int(cell), nd, _lib.ctypes.byref(arr), _ptr(lens, torch.int32, True), B, T,
               H, float(forget_bias), _ptr(ws), n), "os2s_rnn_layer_bwd_multi")
  return outs


def rnn_layer_bwd(cell, whT, lens, dy, y, gates, c_seq, H, reverse, forget_bias=1.0):
  return rnn_layer_bwd_multi(cell, [dict(whT=whT, dy=dy, y=y, gates=gates, c_seq=c_seq,
                                         reverse=reverse)], lens, H, forget_bias)[0]


# --------------------------------------------------------------------------
# conv2d via banded channel mixing (DS2)
# --------------------------------------------------------------------------
def conv2d_toeplitz_expand(w, Fi, Fo, sF, padF, out):
  KT, KF, Cin, Cout = w.shape
  f = (_FN_CACHE.get("os2s_conv2d_toeplitz_expand") or _fn("os2s_conv2d_toeplitz_expand", (c_void_p, c_void_p) + (c_int,) * 8 + (c_void_p,)))
  _lib.check(f(_stream(), _ptr(w, torch.float32), KT, KF, Cin, Cout, Fi, Fo, sF, padF,
               _ptr(out, torch.bfloat16)), "os2s_conv2d_toeplitz_expand")
  return out


def conv2d_toeplitz_reduce(dwexp, Fi, Fo, sF, padF, dw):
  KT, KF, Cin, Cout = dw.shape
  f = (_FN_CACHE.get("os2s_conv2d_toeplitz_reduce") or _fn("os2s_conv2d_toeplitz_reduce", (c_void_p, c_void_p) + (c_int,) * 8 + (c_void_p,)))
  _lib.check(f(_stream(), _ptr(dwexp, torch.float32), KT, KF, Cin, Cout, Fi, Fo, sF, padF,
               _ptr(dw, torch.float32)), "os2s_conv2d_toeplitz_reduce")


# --------------------------------------------------------------------------
# attention-RNN decoder loop (NMT gnmt / Tacotron2)
# --------------------------------------------------------------------------
SCORE_BAHDANAU, SCORE_BAHDANAU_NORM, SCORE_LOCATION, SCORE_LUONG = 0, 1, 2, 3
c_ull = _lib.ctypes.c_ulonglong


class _AttnDecoder(_lib.ctypes.Structure):
  _fields_ = [(n, c_int) for n in ("B", "T", "S", "L", "H", "M", "U", "score_mode", "use_bias",
                                   "loc_k", "loc_f", "t_begin", "t_end")] + [
      ("forget_bias", c_float), ("attn_in_keep", c_float), ("attn_in_seed", c_ull),
      ("out_keep", c_float), ("out_seed", c_ull * 2),
      ("wcat", c_void_p * 2), ("bias", c_void_p * 2), ("wq", c_void_p),
      ("v", c_void_p), ("g", c_void_p), ("b", c_void_p),
      ("conv_w", c_void_p), ("conv_b", c_void_p), ("dense_w", c_void_p), ("loc_ws", c_void_p),
      ("gx0", c_void_p), ("keys", c_void_p), ("values", c_void_p),
      ("src_len", c_void_p), ("tgt_len", c_void_p),
      ("cat", c_void_p * 2), ("c_seq", c_void_p * 2), ("gates", c_void_p * 2),
      ("cum_seq", c_void_p), ("align_seq", c_void_p), ("q_seq", c_void_p),
      ("y_top", c_void_p), ("y_top_bs", c_ll), ("y_top_ts", c_ll),
      ("ctx", c_void_p), ("ctx_bs", c_ll), ("ctx_ts", c_ll),
      ("wcat8", c_void_p * 2), ("wcat8_scale", c_void_p * 2)]


def quantize_rows_e4m3(w2d, q=None, scale=None):
  """bf16 [rows, K] -> (uint8 e4m3 [rows, K], fp32 per-row scale [rows]); w ~= q * scale[:, None]."""
  rows, K = w2d.shape
  assert w2d.dtype == torch.bfloat16 and w2d.is_contiguous()
  if q is None:
    q = torch.empty((rows, K), dtype=torch.uint8, device=w2d.device)
    scale = torch.empty((rows,), dtype=torch.float32, device=w2d.device)
  f = (_FN_CACHE.get("os2s_quantize_rows_e4m3") or _fn("os2s_quantize_rows_e4m3", (c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p)))
  _lib.check(f(_stream(), _ptr(w2d, torch.bfloat16), rows, K, _ptr(q, torch.uint8), _ptr(scale, torch.float32)),
             "os2s_quantize_rows_e4m3")
  return q, scale


class _AttnDecoderGrads(_lib.ctypes.Structure):
  _fields_ = [("wcatT", c_void_p * 2), ("wqT", c_void_p),
              ("dy_top", c_void_p), ("dy_top_bs", c_ll), ("dy_top_ts", c_ll),
              ("dctx_ext", c_void_p), ("dctx_bs", c_ll), ("dctx_ts", c_ll),
              ("dg", c_void_p * 2), ("dq_seq", c_void_p), ("dctx_seq", c_void_p),
              ("dpre_seq", c_void_p), ("dkeys", c_void_p), ("dmem", c_void_p), ("dv", c_void_p), ("dg_scalar", c_void_p),
              ("dconv_w", c_void_p), ("dconv_b", c_void_p), ("ddense_w", c_void_p)]


class AttnDecoder(object):
  """Owns the sequence buffers of one os2s_attn_decoder_{fwd,bwd} call (see include/os2s.h).
  `y_top` / `ctx` may be channel-slice views of a wider [B,T,ld] tensor (Tacotron's
  concat(cell output, context) is then never materialised by a copy)."""

  def __init__(self, B, T, S, L, H, M, U, mode, device, use_bias=False, loc_k=0, loc_f=0,
               forget_bias=1.0, attn_in_keep=1.0, attn_in_seed=0, out_keep=1.0, out_seeds=(0, 0),
               save=True, y_top=None, ctx=None):
    self.dims = dict(B=B, T=T, S=S, L=L, H=H, M=M, U=U)
    self.mode, self.use_bias, self.loc_k, self.loc_f = mode, use_bias, loc_k, loc_f
    self.forget_bias, self.attn_in_keep, self.attn_in_seed = forget_bias, attn_in_keep, attn_in_seed
    self.out_keep, self.out_seeds = out_keep, tuple(out_seeds)
    bf, f32 = torch.bfloat16, torch.float32
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=device)
    kc = [M + H, 2 * H]
    self.cat = [z((B, T + 1, kc[l]), bf) for l in range(L)]
    self.c_seq = [z((B, T, H), f32) for l in range(L)]
    self.gates = [z((B, T, 4 * H), bf) if save else None for l in range(L)]
    self.cum_seq = z((B, T + 1, S), f32) if mode == SCORE_LOCATION else None
    self.loc_ws = z(((loc_k + 1) * U + B * 4 * S,), f32) if mode == SCORE_LOCATION else None   # os2s_attn_decoder_loc_ws_floats
    self.align_seq = z((B, T, S), f32)
    self.q_seq = z((B, T, U), f32)
    self.y_top = y_top if y_top is not None else z((B, T, H), bf)
    self.ctx = ctx if ctx is not None else z((B, T, M), bf)
    for t in (self.y_top, self.ctx):
      assert t.dtype == bf and t.stride(2) == 1
    self.params = None
    self.inputs = None

  def set_params(self, wcat, wq, v, bias=(None, None), g=None, b=None, conv_w=None, conv_b=None,
                 dense_w=None):
    self.params = dict(wcat=list(wcat), bias=list(bias) + [None] * (2 - len(bias)), wq=wq, v=v, g=g,
                       b=b, conv_w=conv_w, conv_b=conv_b, dense_w=dense_w, wcat8=None)

  def set_fp8_weights(self, wcat8):
    """wcat8: per layer (q uint8 [4H, Kc], scale fp32 [4H]) from quantize_rows_e4m3, or None."""
    self.params["wcat8"] = wcat8

  def _desc(self, t_begin, t_end):
    d, p, i = _AttnDecoder(), self.params, self.inputs
    for k, val in self.dims.items():
      setattr(d, k, val)
    L = self.dims["L"]
    d.score_mode, d.use_bias, d.loc_k, d.loc_f = self.mode, int(self.use_bias), self.loc_k, self.loc_f
    d.t_begin, d.t_end = t_begin, t_end
    d.forget_bias, d.attn_in_keep, d.attn_in_seed = self.forget_bias, self.attn_in_keep, self.attn_in_seed
    d.out_keep = self.out_keep
    for l in range(2):
      d.out_seed[l] = self.out_seeds[l]
      d.wcat[l] = _addr(p["wcat"][l]) if l < L else None
      w8 = p.get("wcat8")
      d.wcat8[l] = _addr(w8[l][0]) if (w8 is not None and l < L) else None
      d.wcat8_scale[l] = _addr(w8[l][1]) if (w8 is not None and l < L) else None
      d.bias[l] = _addr(p["bias"][l]) if l < L else None
      d.cat[l] = _addr(self.cat[l]) if l < L else None
      d.c_seq[l] = _addr(self.c_seq[l]) if l < L else None
      d.gates[l] = _addr(self.gates[l]) if l < L else None
    for k in ("wq", "v", "g", "b", "conv_w", "conv_b", "dense_w"):
      setattr(d, k, _addr(p[k]))
    for k in ("gx0", "keys", "values", "src_len", "tgt_len"):
      setattr(d, k, _addr(i[k]))
    d.cum_seq, d.align_seq, d.q_seq = _addr(self.cum_seq), _addr(self.align_seq), _addr(self.q_seq)
    d.loc_ws = _addr(self.loc_ws)
    d.y_top, d.y_top_bs, d.y_top_ts = self.y_top.data_ptr(), self.y_top.stride(0), self.y_top.stride(1)
    d.ctx, d.ctx_bs, d.ctx_ts = self.ctx.data_ptr(), self.ctx.stride(0), self.ctx.stride(1)
    return d

  def set_inputs(self, gx0, keys, values, src_len, tgt_len=None):
    for t in (gx0, keys, values):
      assert t.dtype == torch.bfloat16 and t.is_contiguous()
    assert src_len.dtype == torch.int32 and (tgt_len is None or tgt_len.dtype == torch.int32)
    self.inputs = dict(gx0=gx0, keys=keys, values=values, src_len=src_len, tgt_len=tgt_len)

  def forward(self, t_begin=0, t_end=None):
    t_end = self.dims["T"] if t_end is None else t_end
    d = self._desc(t_begin, t_end)
    f = (_FN_CACHE.get("os2s_attn_decoder_fwd") or _fn("os2s_attn_decoder_fwd", (c_void_p, c_void_p)))
    _lib.check(f(_stream(), _lib.ctypes.byref(d)), "os2s_attn_decoder_fwd")

  def backward(self, wcatT, wqT, dy_top=None, dctx_ext=None, dv=None, dg=None, dconv_w=None,
               dconv_b=None, ddense_w=None):
    """Returns dict(dg=[L x bf16 [B,T,4H]], dq_seq, dkeys fp32, dmem bf16)."""
    B, T, S, L, H, M, U = (self.dims[k] for k in "BTSLHMU")
    dev = self.align_seq.device
    bf = torch.bfloat16
    out = dict(dg=[torch.empty((B, T, 4 * H), dtype=bf, device=dev) for _ in range(L)],
               dq_seq=torch.empty((B, T, U), dtype=bf, device=dev),
               dkeys=torch.empty((B, S, U), dtype=torch.float32, device=dev),
               dmem=torch.empty((B, S, M), dtype=bf, device=dev))
    dctx_seq = torch.empty((B, T, M), dtype=bf, device=dev)
    dpre_seq = torch.empty((B, T, S, U), dtype=bf, device=dev)
    g = _AttnDecoderGrads()
    assert wqT.dtype == bf and wqT.is_contiguous() and wqT.shape == (H, U)
    g.wqT = wqT.data_ptr()
    for l in range(2):
      g.wcatT[l] = _addr(wcatT[l]) if l < L else None
      g.dg[l] = _addr(out["dg"][l]) if l < L else None
    if dy_top is not None:
      g.dy_top, g.dy_top_bs, g.dy_top_ts = dy_top.data_ptr(), dy_top.stride(0), dy_top.stride(1)
    if dctx_ext is not None:
      g.dctx_ext, g.dctx_bs, g.dctx_ts = dctx_ext.data_ptr(), dctx_ext.stride(0), dctx_ext.stride(1)
    g.dq_seq, g.dctx_seq, g.dkeys, g.dmem = (_addr(out["dq_seq"]), _addr(dctx_seq),
                                             _addr(out["dkeys"]), _addr(out["dmem"]))
    g.dpre_seq = _addr(dpre_seq)
    g.dv, g.dg_scalar = _addr(dv), _addr(dg)
    g.dconv_w, g.dconv_b, g.ddense_w = _addr(dconv_w), _addr(dconv_b), _addr(ddense_w)
    d = self._desc(0, T)
    n = int((_FN_CACHE.get("os2s_attn_decoder_bwd_workspace_bytes") or _fn("os2s_attn_decoder_bwd_workspace_bytes", (c_void_p,), c_size_t))(_lib.ctypes.byref(d)))
    ws = torch.empty((n,), dtype=torch.uint8, device=dev)
    f = (_FN_CACHE.get("os2s_attn_decoder_bwd") or _fn("os2s_attn_decoder_bwd", (c_void_p, c_void_p, c_void_p, c_void_p, c_size_t)))
    _lib.check(f(_stream(), _lib.ctypes.byref(d), _lib.ctypes.byref(g), _ptr(ws), n),
               "os2s_attn_decoder_bwd")
    return out


class _TacotronInfer(_lib.ctypes.Structure):
  """ctypes mirror of os2s_tacotron_infer_t (include/os2s.h)."""
  _fields_ = [("loop", c_void_p), ("P", c_int), ("n_mel", c_int), ("mask_decoder_sequence", c_int),
              ("prenet_keep", c_float), ("prenet_seed", c_ull * 2),
              ("w0x", c_void_p), ("w0x8", c_void_p), ("w0x8_scale", c_void_p), ("bias0", c_void_p),
              ("wp1", c_void_p), ("bp1", c_void_p), ("wp2", c_void_p), ("bp2", c_void_p),
              ("wout_h", c_void_p), ("pv_t", c_void_p), ("values_t", c_void_p), ("bout", c_void_p), ("wstop", c_void_p), ("bstop", c_void_p),
              ("mh", c_void_p), ("x_seq", c_void_p), ("mel", c_void_p), ("stop", c_void_p), ("state", c_void_p)]


class TacotronInfer(object):
  """Free-running Tacotron2 decoding on the device (os2s_tacotron_infer_steps): owns the frame / stop /
  pre-net / state buffers next to an AttnDecoder's sequence buffers. `w`: dict of the tensors of
  os2s_tacotron_infer_t (w0x bf16 [4H, P+M+H] or w0x8 = (uint8, scales), bias0, wp1, bp1, wp2, bp2,
  wout_h, pv_t, values_t, bout, wstop, bstop). Steps are enqueued without host interaction; `done_steps()` reads
  the device-resident stop decision (synchronises)."""

  def __init__(self, loop, P, n_mel, w, mask_decoder_sequence=True, prenet_keep=0.5, prenet_seeds=(0, 0)):
    self.loop, self.P, self.n_mel, self.w = loop, P, n_mel, w
    self.mask, self.keep, self.seeds = bool(mask_decoder_sequence), float(prenet_keep), tuple(prenet_seeds)
    B, T = loop.dims["B"], loop.dims["T"]
    dev = loop.align_seq.device
    self.x_seq = torch.zeros((B, T + 1, P), dtype=torch.bfloat16, device=dev)
    self.mel = torch.zeros((B, T, n_mel), dtype=torch.bfloat16, device=dev)
    self.stop = torch.zeros((B, T), dtype=torch.float32, device=dev)
    self.mh = torch.zeros((B, n_mel), dtype=torch.float32, device=dev)
    n = int((_FN_CACHE.get("os2s_tacotron_infer_state_ints") or _fn("os2s_tacotron_infer_state_ints", (c_int,), c_size_t))(B))
    self.state = torch.zeros((n,), dtype=torch.int32, device=dev)
    self._keep_alive = None

  def _desc(self):
    d = self.loop._desc(0, self.loop.dims["T"])
    x, w = _TacotronInfer(), self.w
    x.loop = _lib.ctypes.addressof(d)
    x.P, x.n_mel, x.mask_decoder_sequence = self.P, self.n_mel, int(self.mask)
    x.prenet_keep = self.keep
    x.prenet_seed[0], x.prenet_seed[1] = (int(v) & (2**64 - 1) for v in self.seeds)
    if w.get("w0x8") is not None:
      x.w0x, x.w0x8, x.w0x8_scale = None, _addr(w["w0x8"][0]), _addr(w["w0x8"][1])
    else:
      x.w0x, x.w0x8, x.w0x8_scale = _addr(w["w0x"]), None, None
    for k in ("bias0", "wp1", "bp1", "wp2", "bp2", "wout_h", "pv_t", "values_t", "bout", "wstop", "bstop"):
      setattr(x, k, _addr(w[k]))
    x.x_seq, x.mel, x.stop, x.state = _addr(self.x_seq), _addr(self.mel), _addr(self.stop), _addr(self.state)
    x.mh = _addr(self.mh)
    self._keep_alive = d
    return x

  def supported(self):
    return bool((_FN_CACHE.get("os2s_tacotron_infer_supported") or _fn("os2s_tacotron_infer_supported", (c_void_p,)))(_lib.ctypes.byref(self._desc())))

  def steps(self, t_begin, t_end):
    x = self._desc()
    f = (_FN_CACHE.get("os2s_tacotron_infer_steps") or _fn("os2s_tacotron_infer_steps", (c_void_p, c_void_p, c_int, c_int)))
    _lib.check(f(_stream(), _lib.ctypes.byref(x), int(t_begin), int(t_end)), "os2s_tacotron_infer_steps")

  def done_steps(self):
    """0 while samples are still running, else the number of steps after which all had finished."""
    return int(self.state[1].item())

  def poll_async(self):
    """Snapshot of the stop word AS OF the work enqueued so far, without draining the stream: an async copy
    of state[1] into pinned memory + an event. `poll_wait(handle)` waits for that event only — kernels enqueued
    after the call keep the GPU busy while the host looks."""
    if getattr(self, "_pin", None) is None:
      self._pin = torch.zeros(1, dtype=torch.int32).pin_memory()
    self._pin.copy_(self.state[1:2], non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    return ev

  def poll_wait(self, ev):
    ev.synchronize()
    return int(self._pin[0])

  @property
  def lengths(self):
    B = self.loop.dims["B"]
    return self.state[4 + B:4 + 2 * B]


# --------------------------------------------------------------------------
# text-to-speech loss terms and small element-wise ops
# --------------------------------------------------------------------------
LOSS_MSE, LOSS_L1, LOSS_SIGMOID_XENT = 0, 1, 2


def tts_loss(pred, target, lens, F, mode, weight, loss, grad_scale_dev=None, want_grad=True, target_pad=0.0):
  """pred bf16 [B,Tp,ld>=F], target fp32 [B,Tt,ld_t>=F] (both may be column-slice views);
  loss fp32 [1] is accumulated. Tp != Tt: both sides padded to max(Tp, Tt) as the reference does
  (predictions with zeros, targets with target_pad; os2s_tts_loss_padded). Returns dpred (same
  shape/strides as a fresh [B,Tp,ld] tensor; columns >= F untouched/zero) or None."""
  B, Tp, Tt = pred.shape[0], pred.shape[1], target.shape[1]
  assert target.shape[0] == B
  assert pred.stride(2) == 1 and pred.stride(0) == Tp * pred.stride(1)
  assert target.stride(2) == 1 and target.stride(0) == Tt * target.stride(1)
  dev = pred.device
  nparts = int((_FN_CACHE.get("os2s_tts_loss_num_parts") or _fn("os2s_tts_loss_num_parts", (c_int, c_int)))(B, max(Tp, Tt)))
  partial = torch.empty(nparts, dtype=torch.float32, device=dev)
  dpred = torch.zeros((B, Tp, pred.stride(1)), dtype=torch.bfloat16, device=dev) if want_grad else None
  f = (_FN_CACHE.get("os2s_tts_loss_padded") or _fn("os2s_tts_loss_padded", (c_void_p, c_void_p, c_ll, c_int, c_void_p, c_ll, c_int, c_float, c_void_p,
                                   c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p)))
  _lib.check(f(_stream(), c_void_p(pred.data_ptr()), pred.stride(1), Tp, c_void_p(target.data_ptr()),
               target.stride(1), Tt, float(target_pad), _ptr(lens, torch.int32, True), B, F, mode, float(weight),
               _ptr(grad_scale_dev, torch.float32, True), _ptr(partial), _ptr(loss, torch.float32),
               _ptr(dpred, None, True)), "os2s_tts_loss_padded")
  return dpred


def exp_fwd(x):
  y = torch.empty_like(x)
  _lib.check((_FN_CACHE.get("os2s_exp_fwd") or _fn("os2s_exp_fwd", (c_void_p, c_void_p, c_ll, c_void_p)))(
      _stream(), _ptr(x, torch.bfloat16), x.numel(), _ptr(y)), "os2s_exp_fwd")
  return y


def tanh_fwd(x):
  y = torch.empty_like(x)
  _lib.check((_FN_CACHE.get("os2s_tanh_fwd") or _fn("os2s_tanh_fwd", (c_void_p, c_void_p, c_ll, c_void_p)))(
      _stream(), _ptr(x, torch.bfloat16), x.numel(), _ptr(y)), "os2s_tanh_fwd")
  return y


def tanh_bwd(dy, y):
  dx = torch.empty_like(y)
  _lib.check((_FN_CACHE.get("os2s_tanh_bwd") or _fn("os2s_tanh_bwd", (c_void_p, c_void_p, c_void_p, c_ll, c_void_p)))(
      _stream(), _ptr(dy, torch.bfloat16), _ptr(y, torch.bfloat16), y.numel(), _ptr(dx)), "os2s_tanh_bwd")
  return dx


def mul_bf16(a, b):
  y = torch.empty_like(a)
  _lib.check((_FN_CACHE.get("os2s_mul_bf16") or _fn("os2s_mul_bf16", (c_void_p, c_void_p, c_void_p, c_ll, c_void_p)))(
      _stream(), _ptr(a, torch.bfloat16), _ptr(b, torch.bfloat16), a.numel(), _ptr(y)), "os2s_mul_bf16")
  return y


def sum_time(x, out, accumulate=False):
  """x bf16 [B,T,C] (may be a channel-slice view) -> out fp32 [B,C] (+)= sum over T."""
  B, T, C = x.shape
  assert x.stride(2) == 1 and x.stride(0) == T * x.stride(1)
  _lib.check((_FN_CACHE.get("os2s_sum_time") or _fn("os2s_sum_time", (c_void_p, c_void_p, c_ll, c_int, c_int, c_int, c_void_p, c_int)))(
      _stream(), c_void_p(x.data_ptr()), x.stride(1), B, T, C, _ptr(out, torch.float32),
      int(accumulate)), "os2s_sum_time")
  return out


# --------------------------------------------------------------------------
# global style tokens: TF GRUCell summary + token attention
# --------------------------------------------------------------------------
def gru_tf_fwd(gxg, gxc, wgh, wch, lens):
  """-> dict(h_final [B,H] fp32, saved tensors)."""
  B, T, H2 = gxg.shape
  H = H2 // 2
  dev = gxg.device
  f32, bf = torch.float32, torch.bfloat16
  sv = dict(h_seq=torch.empty((B, T + 1, H), dtype=f32, device=dev),
            r_seq=torch.empty((B, T, H), dtype=f32, device=dev),
            u_seq=torch.empty((B, T, H), dtype=f32, device=dev),
            c_seq=torch.empty((B, T, H), dtype=f32, device=dev),
            hprev16=torch.empty((B, T, H), dtype=bf, device=dev),
            rh16=torch.empty((B, T, H), dtype=bf, device=dev),
            h_final=torch.empty((B, H), dtype=f32, device=dev))
  f = (_FN_CACHE.get("os2s_gru_tf_fwd") or _fn("os2s_gru_tf_fwd", (c_void_p,) * 6 + (c_int,) * 3 + (c_void_p,) * 7))
  _lib.check(f(_stream(), _ptr(gxg, bf), _ptr(gxc, bf), _ptr(wgh, f32), _ptr(wch, f32),
               _ptr(lens, torch.int32, True), B, T, H, _ptr(sv["h_seq"]), _ptr(sv["r_seq"]),
               _ptr(sv["u_seq"]), _ptr(sv["c_seq"]), _ptr(sv["hprev16"]), _ptr(sv["rh16"]),
               _ptr(sv["h_final"])), "os2s_gru_tf_fwd")
  return sv


def gru_tf_bwd(dh_final, wghT, wchT, lens, sv):
  B, T, H = sv["r_seq"].shape
  dev = dh_final.device
  dgxg = torch.empty((B, T, 2 * H), dtype=torch.bfloat16, device=dev)
  dgxc = torch.empty((B, T, H), dtype=torch.bfloat16, device=dev)
  f = (_FN_CACHE.get("os2s_gru_tf_bwd") or _fn("os2s_gru_tf_bwd", (c_void_p,) * 5 + (c_int,) * 3 + (c_void_p,) * 6))
  _lib.check(f(_stream(), _ptr(dh_final, torch.float32), _ptr(wghT, torch.float32),
               _ptr(wchT, torch.float32), _ptr(lens, torch.int32, True), B, T, H, _ptr(sv["h_seq"]),
               _ptr(sv["r_seq"]), _ptr(sv["u_seq"]), _ptr(sv["c_seq"]), _ptr(dgxg), _ptr(dgxc)),
             "os2s_gru_tf_bwd")
  return dgxg, dgxc


def gst_attention_fwd(q, k, v, att_v, heads):
  B, D = q.shape
  N = k.shape[0]
  out = torch.empty_like(q)
  w = torch.empty((B, heads, N), dtype=torch.float32, device=q.device)
  f = (_FN_CACHE.get("os2s_gst_attention_fwd") or _fn("os2s_gst_attention_fwd", (c_void_p,) * 5 + (c_int,) * 3 + (c_void_p,) * 2))
  _lib.check(f(_stream(), _ptr(q, torch.bfloat16), _ptr(k, torch.bfloat16), _ptr(v, torch.bfloat16),
               _ptr(att_v, torch.float32), B, heads, N, _ptr(out), _ptr(w)), "os2s_gst_attention_fwd")
  return out, w


def gst_attention_bwd(dout, q, k, v, att_v, w, heads, dk, dv, datt_v):
  B, D = q.shape
  N = k.shape[0]
  dq = torch.empty_like(q)
  f = (_FN_CACHE.get("os2s_gst_attention_bwd") or _fn("os2s_gst_attention_bwd", (c_void_p,) * 7 + (c_int,) * 3 + (c_void_p,) * 4))
  _lib.check(f(_stream(), _ptr(dout, torch.bfloat16), _ptr(q, torch.bfloat16), _ptr(k, torch.bfloat16),
               _ptr(v, torch.bfloat16), _ptr(att_v, torch.float32), _ptr(w, torch.float32), B, heads,
               N, _ptr(dq), _ptr(dk, torch.float32), _ptr(dv, torch.float32),
               _ptr(datt_v, torch.float32)), "os2s_gst_attention_bwd")
  return dq


def tts_spectrogram(signal, n_samples, window, *, n_fft, hop, T, mag_power, data_min_mag, data_min_mel,
                    n_mag, n_mels, mel_start=None, mel_len=None, mel_wt=None, pad_mel=0.0, pad_mag=0.0):
  """signal fp32 [B, N] -> (mel fp32 [B,T,n_mels] | None, log-mag fp32 [B,T,n_mag] | None)."""
  B = signal.shape[0]
  dev = signal.device
  f32 = torch.float32
  mel = torch.empty((B, T, n_mels), dtype=f32, device=dev) if n_mels else None
  mag = torch.empty((B, T, n_mag), dtype=f32, device=dev) if n_mag else None
  f = (_FN_CACHE.get("os2s_tts_spectrogram") or _fn("os2s_tts_spectrogram", (c_void_p, c_void_p, c_ll, c_void_p, c_void_p, c_int, c_int, c_int,
                                   c_int, c_int, c_float, c_float, c_int, c_int, c_void_p, c_void_p,
                                   c_void_p, c_int, c_void_p, c_void_p, c_float, c_float)))
  _lib.check(f(_stream(), _ptr(signal, f32), signal.stride(0), _ptr(n_samples, torch.int32),
               _ptr(window, f32), B, n_fft, hop, T, mag_power, float(data_min_mag), float(data_min_mel),
               n_mag or 0, n_mels or 0, _ptr(mel_start, torch.int32, True), _ptr(mel_len, torch.int32, True),
               _ptr(mel_wt, f32, True), 0 if mel_wt is None else mel_wt.shape[0], _ptr(mel, None, True),
               _ptr(mag, None, True), float(pad_mel), float(pad_mag)), "os2s_tts_spectrogram")
  return mel, mag


# --------------------------------------------------------------------------
# depthwise conv1d (sep_conv1d)
# --------------------------------------------------------------------------
def depthwise_conv1d_fwd(x, w, *, stride=1, dil=1, pad_left=None, tout=None, in_len=None, out_len=None,
                         flip=False):
  """x bf16 [B,Tin,C], w fp32 [K,C] -> y bf16 [B,Tout,C]."""
  B, Tin, C = x.shape
  K = w.shape[0]
  if pad_left is None or tout is None:
    tout, pad_left = same_padding(Tin, K, stride, dil)
  y = (torch.zeros if out_len is not None else torch.empty)((B, tout, C), dtype=torch.bfloat16, device=x.device)
  f = (_FN_CACHE.get("os2s_depthwise_conv1d_fwd") or _fn("os2s_depthwise_conv1d_fwd", (c_void_p,) * 6 + (c_int,) * 9))
  _lib.check(f(_stream(), _ptr(x, torch.bfloat16), _ptr(w, torch.float32), _ptr(y),
               _ptr(in_len, torch.int32, True), _ptr(out_len, torch.int32, True), B, Tin, tout, C, K,
               stride, dil, pad_left, int(flip)), "os2s_depthwise_conv1d_fwd")
  return y


def depthwise_dgrad_bnact_supported(K, stride, dil):
  return stride == 1 and dil == 1 and 2 <= K <= 96


def depthwise_dgrad_bnact(dz, w, dx, *, pad_left, out_len, mask_ref, mask_scale, stat_ref, addend=None):
  """dx = mask(depthwise_conv(dz, flipped w) + addend), mask = (mask_ref > 0) * mask_scale; returns the
  BatchNorm-backward partials [nparts, 2, C] = (sum dx, sum dx * stat_ref) (os2s_depthwise_dgrad_bnact). dz [B,Tin,C],
  dx / addend / mask_ref / stat_ref [B,Tout,C] bf16 contiguous (addend may be dx itself); w fp32 [K,C]."""
  B, Tin, C = dz.shape
  K = w.shape[0]
  Tout = dx.shape[1]
  assert tuple(dx.shape) == (B, Tout, C) == tuple(mask_ref.shape) == tuple(stat_ref.shape)
  assert dz.is_contiguous() and dx.is_contiguous() and mask_ref.is_contiguous() and stat_ref.is_contiguous()
  assert addend is None or (tuple(addend.shape) == tuple(dx.shape) and addend.is_contiguous())
  n = int((_FN_CACHE.get("os2s_depthwise_dgrad_bnact_num_parts") or _fn("os2s_depthwise_dgrad_bnact_num_parts", (c_int, c_int, c_int)))(B, Tout, K))
  stats = _zero_arena.take((n, 2, C), dz.device)
  f = (_FN_CACHE.get("os2s_depthwise_dgrad_bnact") or _fn("os2s_depthwise_dgrad_bnact", (c_void_p,) * 7 + (c_int,) * 6 + (c_void_p, c_float, c_void_p)))
  _lib.check(f(_stream(), _ptr(dz, torch.bfloat16), _ptr(w, torch.float32), _ptr(dx, torch.bfloat16),
               _ptr(addend, torch.bfloat16, True), _ptr(stats, torch.float32), _ptr(out_len, torch.int32, True),
               B, Tin, Tout, C, K, int(pad_left), _ptr(mask_ref, torch.bfloat16), float(mask_scale),
               _ptr(stat_ref, torch.bfloat16)), "os2s_depthwise_dgrad_bnact")
  return stats


def pointwise_fold(w, d, out=None):
  """The 1x1-kernel operands of a one-tap separable layer (os2s_pointwise_fold): w fp32 [1, Cout, Cin], d fp32
  [1, Cin] -> (w_eff bf16 [1, Cout, Cin] = w * d, wt_eff bf16 [1, Cin, Cout] = its transpose), into `out` = a pair
  of such buffers if given."""
  _, cout, cin = w.shape
  assert d.numel() == cin
  if out is not None:
    w_eff, wt_eff = out
    assert tuple(w_eff.shape) == (1, cout, cin) and tuple(wt_eff.shape) == (1, cin, cout)
  else:
    w_eff = torch.empty((1, cout, cin), dtype=torch.bfloat16, device=w.device)
    wt_eff = torch.empty((1, cin, cout), dtype=torch.bfloat16, device=w.device)
  f = (_FN_CACHE.get("os2s_pointwise_fold") or _fn("os2s_pointwise_fold", (c_void_p,) * 5 + (c_int, c_int)))
  _lib.check(f(_stream(), _ptr(w, torch.float32), _ptr(d, torch.float32), _ptr(w_eff, torch.bfloat16),
               _ptr(wt_eff, torch.bfloat16), cout, cin), "os2s_pointwise_fold")
  return w_eff, wt_eff


def pointwise_fold_bwd(g, w, d, dw, dd):
  """dw += g * d (columns), dd += sum over rows of w * g (os2s_pointwise_fold_bwd): g, w, dw fp32 [1, Cout, Cin],
  d, dd fp32 [1, Cin]."""
  _, cout, cin = w.shape
  assert tuple(g.shape) == tuple(w.shape) == tuple(dw.shape) and d.numel() == cin == dd.numel()
  f = (_FN_CACHE.get("os2s_pointwise_fold_bwd") or _fn("os2s_pointwise_fold_bwd", (c_void_p,) * 6 + (c_int, c_int)))
  _lib.check(f(_stream(), _ptr(g, torch.float32), _ptr(w, torch.float32), _ptr(d, torch.float32),
               _ptr(dw, torch.float32), _ptr(dd, torch.float32), cout, cin), "os2s_pointwise_fold_bwd")


def depthwise_conv1d_wgrad(x, dy, dw, *, stride=1, dil=1, pad_left=None, in_len=None):
  """dw fp32 [K,C] += sum dy * shifted x."""
  B, Tin, C = x.shape
  K = dw.shape[0]
  tout = dy.shape[1]
  if pad_left is None:
    pad_left = same_padding(Tin, K, stride, dil)[1]
  f = (_FN_CACHE.get("os2s_depthwise_conv1d_wgrad") or _fn("os2s_depthwise_conv1d_wgrad", (c_void_p,) * 5 + (c_int,) * 8))
  _lib.check(f(_stream(), _ptr(x, torch.bfloat16), _ptr(dy, torch.bfloat16), _ptr(dw, torch.float32),
               _ptr(in_len, torch.int32, True), B, Tin, tout, C, K, stride, dil, pad_left),
             "os2s_depthwise_conv1d_wgrad")


# --------------------------------------------------------------------------
# CTC prefix beam search with an n-gram language model (host entry points)
# --------------------------------------------------------------------------
class CtcScorer(object):
  """Handle of os2s_ctc_scorer_create (language model + letter trie + alphabet)."""

  def __init__(self, lm_path, trie_path, alphabet_path, alpha, beta, trie_weight=0.1):
    import ctypes
    self._h = ctypes.c_void_p(0)
    self._destroy = (_FN_CACHE.get("os2s_ctc_scorer_destroy") or _fn("os2s_ctc_scorer_destroy", (c_void_p,), None))
    f = (_FN_CACHE.get("os2s_ctc_scorer_create") or _fn("os2s_ctc_scorer_create", (ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, c_float,
                                       c_float, c_float, ctypes.POINTER(ctypes.c_void_p))))
    _lib.check(f(str(lm_path).encode(), str(trie_path).encode(), str(alphabet_path).encode(),
                 float(alpha), float(beta), float(trie_weight), ctypes.byref(self._h)),
               "os2s_ctc_scorer_create(%s, %s, %s)" % (lm_path, trie_path, alphabet_path))

  @property
  def handle(self):
    return self._h

  def set_weights(self, alpha, beta, trie_weight=0.1):
    f = (_FN_CACHE.get("os2s_ctc_scorer_set_weights") or _fn("os2s_ctc_scorer_set_weights", (c_void_p, c_float, c_float, c_float)))
    _lib.check(f(self._h, float(alpha), float(beta), float(trie_weight)), "os2s_ctc_scorer_set_weights")

  def ngram_score(self, words):
    import ctypes
    arr = (ctypes.c_char_p * len(words))(*[w.encode() for w in words])
    out = c_float(0)
    f = (_FN_CACHE.get("os2s_ctc_scorer_ngram_score") or _fn("os2s_ctc_scorer_ngram_score", (c_void_p, ctypes.POINTER(ctypes.c_char_p), c_int,
                                            ctypes.POINTER(c_float))))
    _lib.check(f(self._h, arr, len(words), ctypes.byref(out)), "os2s_ctc_scorer_ngram_score")
    return out.value

  def __del__(self):
    h, self._h = getattr(self, "_h", None), None
    if h and getattr(self, "_destroy", None) is not None:
      self._destroy(h)


def ctc_beam_search(logits, seq_len, beam_width, scorer=None, top_paths=1, merge_repeated=False,
                    n_threads=0):
  """logits [T,B,C] float32 HOST tensor (time-major, raw logits), seq_len [B] int32 host tensor.
  Returns (ids [B,top_paths,T] int32 padded with -1, lens [B,top_paths], log_probs [B,top_paths]),
  host tensors. The reference's op is CPU-only as well (beam_search.cc:803)."""
  if logits.is_cuda or seq_len.is_cuda:
    raise ValueError("ctc_beam_search is a host entry point: pass CPU tensors")
  logits = logits.contiguous().float()
  seq_len = seq_len.contiguous().to(torch.int32)
  T, B, C = logits.shape
  ids = torch.empty((B, top_paths, max(T, 1)), dtype=torch.int32)
  lens = torch.empty((B, top_paths), dtype=torch.int32)
  lp = torch.empty((B, top_paths), dtype=torch.float32)
  f = (_FN_CACHE.get("os2s_ctc_beam_search") or _fn("os2s_ctc_beam_search", (c_void_p, c_int64, c_int64, c_void_p, c_int, c_int, c_int, c_int,
                                   c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p)))
  if T == 0:
    raise ValueError("empty logits")
  _lib.check(f(logits.data_ptr(), B * C, C, seq_len.data_ptr(), T, B, C, int(beam_width),
               int(top_paths), int(bool(merge_repeated)), scorer.handle if scorer else None,
               int(n_threads), ids.data_ptr(), lens.data_ptr(), lp.data_ptr()),
             "os2s_ctc_beam_search")
  return ids, lens, lp


def ctc_generate_trie(alphabet_path, lm_path, vocab_path, trie_path):
  """The reference's generate_trie tool (ctc_decoder_with_lm/generate_trie.cpp)."""
  import ctypes
  f = (_FN_CACHE.get("os2s_ctc_generate_trie") or _fn("os2s_ctc_generate_trie", (ctypes.c_char_p,) * 4))
  _lib.check(f(str(alphabet_path).encode(), str(lm_path).encode(), str(vocab_path).encode(),
               str(trie_path).encode()), "os2s_ctc_generate_trie")

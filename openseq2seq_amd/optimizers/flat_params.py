"""Flat parameter storage for the multi-tensor optimizer kernels.

All trainable tensors live in ONE fp32 master buffer, with matching flat
buffers for gradients, optimizer moments and the bf16 compute copy (the
reference's "FP32-master-copy" variables + fp16 variables,
open_seq2seq/optimizers/mp_wrapper.py:57-97). Every tensor starts at a multiple
of os2s_opt_chunk_elems() elements so one optimizer chunk belongs to exactly
one tensor; the flat gradient buffer is also what RCCL all-reduces in buckets.
Conv kernels additionally keep a tap-flipped transposed bf16 copy that the
data-gradient convolution consumes (see include/os2s.h, os2s_conv1d_fwd).
"""
import struct

import numpy as np
import os

import torch

from .. import capi


def _guarded(slot, wt=False):
  """A view into the flat buffers that the asynchronous optimizer (TrainOp.run_async) may still be writing: reading
  the attribute first makes the calling stream wait for the update range that holds the variable
  (FlatParams._ready). No update in flight: one list test."""
  def get(self):
    st = self.store
    if st is not None and (st._pending or st._wt_event is not None):
      st._ready(self, wt)
    return getattr(self, slot)

  def put(self, value):
    setattr(self, slot, value)
  return property(get, put)


class Param(object):
  __slots__ = ("name", "shape", "index", "offset", "numel", "l2", "kind", "store",
               "_master", "_grad", "_w16", "_wt16", "wt_offset", "trainable_mask", "logical_out")
  master = _guarded("_master")
  grad = _guarded("_grad")
  w16 = _guarded("_w16")
  wt16 = _guarded("_wt16", wt=True)

  def __init__(self, name, shape, kind, l2, logical_out=None):
    self.name, self.shape, self.kind, self.l2 = name, tuple(int(s) for s in shape), kind, l2
    self.numel = int(np.prod(self.shape))
    # output layers padded to an MFMA-friendly width: number of REAL output units (rows of a
    # [1, Vpad, H] kernel / entries of a [Vpad] bias); checkpoints carry the logical shape
    self.logical_out = logical_out
    self.store = None
    self._master = self._grad = self._w16 = self._wt16 = None


def _whole(slot):
  """A whole flat buffer: the reader waits for everything the asynchronous optimizer has in flight."""
  def get(self):
    if self._pending or self._wt_event is not None:
      self.wait_all()
    return getattr(self, slot)

  def put(self, value):
    setattr(self, slot, value)
  return property(get, put)


class FlatParams(object):
  """kind: 'conv' ([K,Cout,Cin], gets a dgrad copy), 'dense', 'vector' (BN/bias)."""

  master = _whole("_master")
  grads = _whole("_grads")
  m1 = _whole("_m1")
  m2 = _whole("_m2")
  w16 = _whole("_w16")
  wt16 = _whole("_wt16")
  t_v = _whole("_t_v")

  def __init__(self, device):
    self.device = device
    # asynchronous optimizer (TrainOp.run_async): [(end offset in elements, event)] of the update ranges still in
    # flight on the optimizer stream, ascending; the event behind the refresh of the transposed copies; the stream
    # the training step runs on; whether the update left the gradient buffer zeroed
    self._pending, self._wt_event, self._main_stream, self._opt_stream = [], None, None, None
    self._done_upto = 0               # elements of the flat buffers the current stream has already waited for
    self.grads_zeroed = False
    self._master = self._grads = self._m1 = self._m2 = self._w16 = self._wt16 = self._t_v = None
    self.params = []
    self.state = {}       # named non-trainable variables (BatchNorm moving statistics)
    self._inits = []
    self.finalized = False
    self.chunk = capi.opt_chunk_elems()

  def add(self, name, shape, init, kind="dense", l2=0.0, logical_out=None):
    assert not self.finalized
    for p in self.params:
      if p.name == name:
        raise ValueError("duplicate variable " + name)
    p = Param(name, shape, kind, float(l2), logical_out)
    p.index = len(self.params)
    p.store = self
    self.params.append(p)
    self._inits.append(init)
    return p

  def add_state(self, name, tensor):
    """Registers a non-trainable variable under its reference name (checkpoints)."""
    if name in self.state:
      raise ValueError("duplicate state variable " + name)
    self.state[name] = tensor
    return tensor

  def finalize(self, need_m2=False):
    dev, ch = self.device, self.chunk
    off = 0
    wt_off = 0
    begins = []
    for p in self.params:
      p.offset = off
      begins.append(off // ch)
      off += -(-p.numel // ch) * ch
      if p.kind == "conv":
        p.wt_offset = wt_off
        wt_off += -(-p.numel // 8) * 8
    self.total = off
    begins.append(off // ch)
    self.master = torch.zeros(off, dtype=torch.float32, device=dev)
    self.grads = torch.zeros(off, dtype=torch.float32, device=dev)
    self.m1 = torch.zeros(off, dtype=torch.float32, device=dev)
    self.m2 = torch.zeros(off, dtype=torch.float32, device=dev) if need_m2 else None
    self.w16 = torch.zeros(off, dtype=torch.bfloat16, device=dev)
    self.wt16 = torch.zeros(max(wt_off, 8), dtype=torch.bfloat16, device=dev)
    nt = len(self.params)
    chunk_tensor = np.zeros(off // ch, np.int32)
    for i, p in enumerate(self.params):
      chunk_tensor[begins[i]:begins[i + 1]] = i
    self.chunk_tensor = torch.from_numpy(chunk_tensor).to(dev)
    self.tensor_chunk_begin = torch.tensor(begins, dtype=torch.int32, device=dev)
    self.tensor_l2 = torch.tensor([p.l2 for p in self.params], dtype=torch.float32, device=dev)
    # no tensor has an l2 term: the optimizer's statistics pass does not read the weights at all (the host
    # knows; a caller that writes into tensor_l2 afterwards sets this flag too)
    self.l2_active = any(float(p.l2) != 0.0 for p in self.params)
    self.partial = torch.zeros(off // ch * 4, dtype=torch.float32, device=dev)
    self.t_gnorm2 = torch.zeros(nt, dtype=torch.float32, device=dev)
    self.t_wnorm2 = torch.zeros(nt, dtype=torch.float32, device=dev)
    self.t_amax = torch.zeros(nt, dtype=torch.float32, device=dev)
    self.t_mult = torch.zeros(nt, dtype=torch.float32, device=dev)
    self.t_v = torch.zeros(nt, dtype=torch.float32, device=dev)
    descs = b""
    tiles = 0
    for p, init in zip(self.params, self._inits):
      sl = slice(p.offset, p.offset + p.numel)
      p.master = self.master[sl].view(p.shape)
      p.grad = self.grads[sl].view(p.shape)
      p.w16 = self.w16[sl].view(p.shape)
      val = init(p.shape) if callable(init) else init
      p.master.copy_(torch.as_tensor(val, dtype=torch.float32).reshape(p.shape))
      if p.kind == "conv":
        K, Cout, Cin = p.shape
        p.wt16 = self.wt16[p.wt_offset:p.wt_offset + p.numel].view(K, Cin, Cout)
        descs += struct.pack("<qqiiii", p.offset, p.wt_offset, K, Cout, Cin, tiles)
        tiles += K * (-(-Cout // 64)) * (-(-Cin // 64))
    self._wt_tiles = tiles
    self._wt_descs = (torch.frombuffer(bytearray(descs), dtype=torch.uint8).to(dev)
                      if tiles else None)
    self._n_wt = len(descs) // 32
    self._inits = None
    self.finalized = True
    self.refresh_compute_copies()
    return self

  def refresh_compute_copies(self):
    """master fp32 -> bf16 compute copy (+ the conv dgrad copies)."""
    capi.cast_f32_to_bf16(self.master, self.w16)
    self.refresh_dgrad_copies()

  def refresh_dgrad_copies(self):
    # every path that changes the bf16 weights ends here (optimizer step, checkpoint load, weight
    # copy): consumers that keep derived copies (fp8 weight copies) compare this counter
    self.version = getattr(self, "version", 0) + 1
    self._refresh_dgrad_copies_raw()

  def _refresh_dgrad_copies_raw(self):
    """The launch itself, on the current stream, without the readiness waits of the guarded attributes (the
    asynchronous optimizer calls it on its own stream, behind its last update range)."""
    if self._wt_tiles:
      # (only backward reads the transposed copies, but refreshing the copies on the side stream, under the first convolutions of the next forward
      # pass, gains nothing: 39.3 / 40.0 vs 39.8 / 40.1 ms per Jasper step on one box — a ping-pong
      # convolution owns whole CUs, every workgroup of another kernel displaces one of its tiles)
      capi.conv_weight_dgrad_copy(self._w16, self._wt16, self._wt_descs.view(-1, 32),
                                  self._wt_tiles)

  def zero_grads(self):
    self.grads.zero_()

  # ---- readiness of the variables while an asynchronous update is in flight ---------------------------------
  def _streams(self):
    cur = torch.cuda.current_stream()
    return (cur,) if self._main_stream is None or cur == self._main_stream else (cur, self._main_stream)

  def _ready(self, p, wt=False):
    """The current stream (and the step's stream: a side-stream body forked before this call does not inherit a
    wait enqueued on the main stream afterwards, and the entry is dropped here) waits for the update range that
    holds the END of variable p — the ranges complete in order, so it covers the whole variable."""
    if wt:
      if self._wt_event is not None:
        for st in self._streams():
          st.wait_event(self._wt_event)
        self._wt_event = None
        self._pending = []            # the refresh of the transposed copies runs behind the last range
      return
    pend = self._pending
    end = p.offset + p.numel
    if not pend or end <= self._done_upto:       # nothing in flight, or the variable's range was waited for already
      return
    i, n = 0, len(pend)
    while i < n - 1 and pend[i][0] < end:
      i += 1
    for st in self._streams():
      st.wait_event(pend[i][1])
    self._done_upto = pend[i][0]
    del pend[:i + 1]

  def wait_all(self):
    """Everything the asynchronous optimizer has in flight (whole-buffer readers: checkpoints, broadcasts, the
    gradient fill of a redone step, tests)."""
    ev = self._wt_event if self._wt_event is not None else (self._pending[-1][1] if self._pending else None)
    if ev is not None:
      for st in self._streams():
        st.wait_event(ev)
    self._pending, self._wt_event = [], None
    self._done_upto = 0

  def opt_stream(self):
    if self._opt_stream is None:
      self._opt_stream = torch.cuda.Stream(device=self.device)
    return self._opt_stream

  def num_trainable(self):
    return sum(p.numel for p in self.params)

  def by_name(self, name):
    for p in self.params:
      if p.name == name:
        return p
    raise KeyError(name)

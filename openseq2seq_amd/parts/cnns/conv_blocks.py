"""conv_bn_actv / conv_bn_res_bn_actv — the conv blocks of
open_seq2seq/parts/cnns/conv_blocks.py:61-232, re-hosted on the HIP kernels.

There is no graph compiler here: a layer object owns its parameters, `forward`
enqueues the kernels and records one backward closure on a Tape; `Tape.backward`
replays them in reverse. Activations are bf16 [B,T,C] channels-last (the
reference's data_format="channels_last"), always stored ALREADY multiplied by
the sequence mask of their length vector — the reference multiplies the mask
onto every conv input (tdnn_encoder.py:185-186,204-205), we fold that multiply
into the producer's store.
"""
import math
import os

import torch

from ... import capi

ACT_IDS = {None: 0, "none": 0, "relu": 1, "tanh": 2, "relu20": 3}


class _Probe(object):
  """Stand-in argument for probing a config's activation lambda under the tensorflow token module
  (compat/tensorflow_shim.py): its functions return tokens that record their arguments."""
  __name__ = "x"


def _probe_activation(fn):
  """`lambda x: tf.minimum(tf.nn.relu(x), 20.0)` — the clipped ReLU of the reference's DeepSpeech2 /
  Wave2Letter configs (ds2_toy_config.py:79, test_speech_configs/*.py) — is recognised by calling it on a
  probe: the token module returns minimum(relu(x), 20.0) as a tree of tokens. Returns an id or None."""
  try:
    r = fn(_Probe())
  except Exception:
    return None
  if getattr(r, "__name__", "") != "minimum" or len(getattr(r, "args", ())) != 2:
    return None
  a, b = r.args
  if isinstance(a, (int, float)):
    a, b = b, a
  inner = getattr(a, "__name__", "")
  if inner == "relu" and isinstance(b, (int, float)) and float(b) == 20.0 and \
     len(getattr(a, "args", ())) == 1 and isinstance(a.args[0], _Probe):
    return ACT_IDS["relu20"]
  return None


def act_id(fn):
  """Maps config tokens (tf.nn.relu, tf.nn.tanh, None, names, the clipped-ReLU lambda) to kernel ids."""
  if fn is None:
    return 0
  name = fn if isinstance(fn, str) else getattr(fn, "__name__", str(fn))
  name = name.lower()
  if name not in ACT_IDS and callable(fn):
    pid = _probe_activation(fn)
    if pid is not None:
      return pid
  if name not in ACT_IDS:
    raise NotImplementedError("activation %r" % (fn,))
  return ACT_IDS[name]


_SIDE_STREAMS = {}


# A/B knob: 0 = every conv + BatchNorm + ReLU layer runs its own BatchNorm-backward reduction pass
# (round 2); default = the pass is folded into the data-gradient launch that finalises the layer's
# output gradient (capi.conv1d_dgrad_bnact)
FUSE_BN_BWD = os.environ.get("OS2S_FUSE_BN_BWD", "1") != "0"
# A/B knob: 0 = one K = 1 weight-gradient launch per residual branch (round 2), default = grouped
GROUP_WGRAD = os.environ.get("OS2S_GROUP_WGRAD", "1") != "0"
# A/B knobs: the pointwise (K = 1) kernel gradients of separable-conv layers collected N per launch of the ping-pong
# TN-GEMM kernel (0 = one lockstep launch each, round 1 - 4)
GROUP_POINTWISE_WGRAD = os.environ.get("OS2S_GROUP_POINTWISE_WGRAD", "1") != "0"
POINTWISE_WGRAD_GROUP = int(os.environ.get("OS2S_POINTWISE_WGRAD_GROUP", "5"))
# how many small Dense weight gradients (Transformer: the 1024 x 1024 projections, 16 tiles each) share one launch
SMALL_WGRAD_GROUP = int(os.environ.get("OS2S_SMALL_WGRAD_GROUP", "3"))
# Round 6: convolution weight gradients of same-shape layers CAN be collected until they cover this many units of the
# ping-pong kernel (128 co x 128 ci x 4 taps each; 256 CUs) and go out as one launch (Tape.defer_conv_wgrad,
# os2s_conv1d_wgrad_grouped_ws). Alone on the GPU the grouped launches save the reduction splits of the 256 - 640
# channel layers; in the Jasper step — where the weight gradients run next to the data-gradient chain — holding them
# back costs what it saves: 37.15 / 37.31 ms without, 37.1 - 37.3 at 64 - 128 units, 37.4 - 38.0 at 160 - 192
# (same box, interleaved). 0 (default) = every layer alone, as in rounds 2 - 5.
CONV_WGRAD_UNIT_BUDGET = int(os.environ.get("OS2S_CONV_WGRAD_UNIT_BUDGET", "0"))
# A/B knob: 0 = the grouped K = 1 weight gradients stay on the lockstep kernel with its atomics (round 2 - 4)
GROUP_WGRAD_PP = os.environ.get("OS2S_GROUP_WGRAD_PP", "1") != "0"
# A/B knob: the forward half of the dense-residual algebra (parts/cnns/dense_residual.py: source copy, Gram matrix,
# block end's residual GEMM) runs on the side stream next to the block's own layers (1) or in front of them (0)
DRES_FWD_SIDE = os.environ.get("OS2S_DRES_FWD_SIDE", "1") != "0"
# A/B knob: 0 = the dense-residual chains share the weight-gradient side stream (FIFO behind its backlog)
DRES_OWN_STREAM = os.environ.get("OS2S_DRES_OWN_STREAM", "1") != "0"
WGRAD_STREAMS = int(os.environ.get("OS2S_WGRAD_STREAMS", "1"))
_WGRAD_RR = 0
# A/B knob: 0 = block ends on the dense-residual algebra keep their own BatchNorm-backward reduction pass
DRES_FUSE_BN_BWD = os.environ.get("OS2S_DRES_FUSE_BN_BWD", "1") != "0"
# A/B knob: 0 = separable layers keep their own BatchNorm-backward reduction pass (rounds 4 - 5); default = it rides
# in the next separable layer's depthwise data gradient (capi.depthwise_dgrad_bnact)
SEP_FUSE_BN_BWD = os.environ.get("OS2S_SEP_FUSE_BN_BWD", "1") != "0"
# A/B knob: 0 = a one-tap separable layer (QuartzNet's residual branches) runs its depthwise scale as a launch of
# its own (rounds 4 - 6); default = one 1x1 convolution with the scale folded into the kernel (SepConvBN.folded)
FOLD_SEP_K1 = os.environ.get("OS2S_FOLD_SEP_K1", "1") != "0"


_SIDE_STREAM_ENABLED = True
# set while two half-batches of ONE step run on two main streams (backward_interleaved): both share the side streams
# of the step's own stream — parameter gradients accumulate into the same buffers, one FIFO keeps them in order
_SIDE_KEY_OVERRIDE = None


def set_side_stream_enabled(on):
  """Per-model switch (config key `os2s_side_stream`, set at the start of every train step): with False
  every `on_side_stream` body runs on the current stream. Returns the previous setting (the caller
  restores it when its step is over)."""
  global _SIDE_STREAM_ENABLED
  prev, _SIDE_STREAM_ENABLED = _SIDE_STREAM_ENABLED, bool(on)
  return prev


def _side_stream(device, which=0):
  """Side stream for work that may overlap the main stream inside one backward closure
  (OS2S_WGRAD_STREAM=0 or the model's `os2s_side_stream: False` keeps everything on one stream).
  which: 0 = the parameter-gradient stream (nothing on the main stream waits for it before the end of the pass),
  1 = the stream of side work the main stream DOES wait for (the dense-residual chains): a stream is a FIFO, a
  chain queued behind a backlog of weight-gradient kernels would stall the main stream until the backlog drained."""
  if not _SIDE_STREAM_ENABLED or os.environ.get("OS2S_WGRAD_STREAM", "1") == "0" or device.type != "cuda":
    return None
  if which == 1 and not DRES_OWN_STREAM:
    which = 0
  base = _SIDE_KEY_OVERRIDE if _SIDE_KEY_OVERRIDE is not None else capi._stream().value
  key = (device.index, base) if not which else (device.index, base, which)
  st = _SIDE_STREAMS.get(key)
  if st is None:
    # OS2S_SIDE_PRIO (experiment): stream priority of the side stream (HIP: lower number = higher
    # priority; the main stream has 0)
    prio = int(os.environ.get("OS2S_DRES_PRIO" if which else "OS2S_SIDE_PRIO", "0"))
    st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device, priority=prio)
  return st


_STREAM_OBJ = {}       # raw stream handle -> torch.cuda.Stream (torch.cuda.current_stream() builds a new object: 7 us)
_FORK_EVENT = {}       # side stream -> the event its forks are ordered by (re-recorded per use: a wait captures the
                       # record that precedes it)


def _current_stream_obj():
  raw = capi._stream().value
  st = _STREAM_OBJ.get(raw)
  if st is None:
    st = _STREAM_OBJ[raw] = torch.cuda.current_stream()
  return st


class on_side_stream(object):
  """`with on_side_stream(device, *operands):` enqueues the body on the side stream, ordered
  after everything the current stream has enqueued so far. For parameter-gradient kernels:
  nothing in the rest of backward reads their result, the main stream re-joins at the end of
  `Tape.backward` and the gradient reducer waits for the side stream itself. `operands` are the
  tensors the body reads that the main stream's closures release afterwards (their memory is kept
  until the side stream is done). With OS2S_WGRAD_STREAM=0 the body runs on the current stream.
  (Host cost matters here — QuartzNet's step is bound by the Python thread, and this context is entered ~250 times
  per step: cached stream objects, one re-recorded event per side stream and torch.cuda.set_stream instead of
  current_stream() / wait_stream() / the torch.cuda.stream context manager: ~35 -> ~10 us per use.)"""

  def __init__(self, device, *operands, which=0):
    self.side = _side_stream(device, which)
    self.operands = operands
    self.main = None

  def __enter__(self):
    if self.side is not None:
      self.main = _current_stream_obj()
      ev = _FORK_EVENT.get(self.side)
      if ev is None:
        ev = _FORK_EVENT[self.side] = torch.cuda.Event()
      ev.record(self.main)
      self.side.wait_event(ev)
      torch.cuda.set_stream(self.side)
    return self

  def __exit__(self, *exc):
    if self.side is not None:
      torch.cuda.set_stream(self.main)
      for t in self.operands:
        if t is not None:
          t.record_stream(self.side)
    return False

  def hand_over(self, *tensors):
    """Tensors ALLOCATED inside the body (side-stream allocations) that the main stream consumes
    after it has joined: their memory must not be recycled for later side-stream allocations while
    main-stream kernels still use them."""
    if self.side is not None:
      for t in tensors:
        if t is not None:
          t.record_stream(self.main)


def side_streams():
  return list(_SIDE_STREAMS.values())


_JOIN_EVENT = {}


def join_side_streams():
  """The current stream waits for everything enqueued on the side streams so far."""
  if _SIDE_STREAMS:
    cur = _current_stream_obj()
    for st in _SIDE_STREAMS.values():
      ev = _JOIN_EVENT.get(st)
      if ev is None:
        ev = _JOIN_EVENT[st] = torch.cuda.Event()
      ev.record(st)
      cur.wait_event(ev)


class Tape(object):
  """Reverse-mode tape. `record(fn, params)` also notes which parameters the closure writes
  gradients of. A parameter is FINAL once every closure that lists it has run; after each
  closure `on_done(w)` tells the data-parallel reducer the watermark w: every parameter at a flat
  offset >= w is final (parameters no closure lists receive no gradient), so complete gradient
  buckets above it can be all-reduced while the rest of backward runs. Variables are created in
  forward order, so the watermark normally falls with every closure; a variable used out of
  creation order (a tied embedding, say) only delays it."""

  def __init__(self, on_done=None):
    self.ops = []
    self.on_done = on_done
    # True: backward() leaves the closures (and the activations they hold) to the caller, who drops them AFTER it has
    # enqueued what follows the pass — releasing a step's few thousand tensors takes the host ~0.5 ms, during which
    # the GPU (which has caught up with the host by the end of backward) would wait for the optimizer launch
    self.defer_free = False

  def record(self, fn, params=()):
    self.ops.append((fn, params))

  def backward(self):
    # Re-entrant: a closure may run another tape's backward pass (a nested pass gets the zeroed scratch arena
    # of its own depth — capi.zero_arena_enter — and its own deferred-gradient list; `current_tape()` is the
    # innermost pass). Work parked on the side streams since the last join must have landed first.
    depth = len(_TAPE_STACK)
    join_side_streams()
    capi.zero_arena_enter(depth)   # the previous pass's statistic partials at this depth are dead: one fill
    _TAPE_STACK.append(self)
    self._deferred, self._pending = [], None
    self._cdeferred, self._ckey = [], None
    try:
      if self.on_done is None:
        for fn, _ in reversed(self.ops):
          fn()
        self.flush_deferred()
        self.flush_conv_wgrads()
      else:
        pending, by_id = {}, {}
        for _, params in self.ops:
          for p in params:
            pending[id(p)] = pending.get(id(p), 0) + 1
            by_id[id(p)] = p
        self._pending = pending
        order = sorted(by_id.values(), key=lambda p: -p.offset)
        ptr = 0

        def advance():
          nonlocal ptr
          moved = False
          while ptr < len(order) and pending[id(order[ptr])] == 0:
            ptr += 1
            moved = True
          if moved:
            self.on_done(order[ptr - 1].offset)

        for fn, params in reversed(self.ops):
          fn()
          if params:
            for p in params:
              pending[id(p)] -= 1
            advance()
        if self._deferred or self._cdeferred:
          self.flush_deferred()
          self.flush_conv_wgrads()
          advance()
    finally:
      _TAPE_STACK.pop()
      capi.zero_arena_leave(depth)
    if not self.defer_free:
      self.ops = []
    join_side_streams()

  # ---- deferred (grouped) weight gradients -----------------------------------------------------
  def defer_wgrad(self, param, item, group=None, unit_budget=None, side=True):
    """A Dense weight gradient too small to fill the chip alone is held back until `group` of them — or, with
    `unit_budget`, enough of them to cover that many 256 x 256 output tiles — can go out in one launch
    (capi.gemm_wgrad_grouped: at most 16 per launch). Until then `param` does not count as final for the
    gradient reducer. Optional keys of `item`: `after` — a callable run behind the grouped launch, on its stream (a
    folded one-tap separable layer splits its product into two variables' gradients there); `also` — further
    parameters that become final with that launch (the caller has raised their pending counts)."""
    # one grouped launch has ONE row count (os2s_gemm_wgrad_grouped takes a single M): a layer fed by
    # another number of packed rows (the enc-dec attention's k/v projection of the SOURCE tokens among
    # target-row layers) starts a new group
    if self._deferred and self._deferred[0][1]["x"].shape[0] != item["x"].shape[0]:
      self.flush_deferred()
    self._deferred.append((param, item))
    self._deferred_side = side        # False: the grouped launch stays on the current stream (serial profiles)
    if self._pending is not None and id(param) in self._pending:
      self._pending[id(param)] += 1
    if unit_budget is not None:
      units = sum(((it["dy"].shape[1] + 255) // 256) * ((it["x"].shape[1] + 255) // 256) for _, it in self._deferred)
      if units >= unit_budget or len(self._deferred) >= 16:
        self.flush_deferred()
    elif len(self._deferred) >= (group if group is not None else SMALL_WGRAD_GROUP):
      self.flush_deferred()

  def defer_conv_wgrad(self, param, key, item, units, launch_kw):
    """The weight gradient of a convolution layer is held back while layers of the SAME shape over the same batch
    follow (the `repeat` sub-blocks of a Jasper block: 12 - 150 units of work each for 256 CUs): they go out as one
    launch of the ping-pong kernel (capi.conv1d_wgrad_grouped, at most 8) once CONV_WGRAD_UNIT_BUDGET units are
    collected, when a layer of another shape arrives, or at the end of the pass. `param` is not final for the
    gradient reducer until then."""
    if self._cdeferred and self._ckey != key:
      self.flush_conv_wgrads()
    self._ckey, self._ckw = key, launch_kw
    self._cdeferred.append((param, item))
    if self._pending is not None and id(param) in self._pending:
      self._pending[id(param)] += 1
    if units * len(self._cdeferred) >= CONV_WGRAD_UNIT_BUDGET or len(self._cdeferred) >= 8:
      self.flush_conv_wgrads()

  def flush_conv_wgrads(self):
    if not self._cdeferred:
      return
    items = [it for _, it in self._cdeferred]
    with on_side_stream(items[0]["x"].device, *([it["x"] for it in items] + [it["dy"] for it in items])):
      capi.conv1d_wgrad_grouped(items, **self._ckw)
    if self._pending is not None:
      for p, _ in self._cdeferred:
        if id(p) in self._pending:
          self._pending[id(p)] -= 1
    self._cdeferred, self._ckey = [], None

  def flush_deferred(self):
    if not self._deferred:
      return
    items = [it for _, it in self._deferred]
    if getattr(self, "_deferred_side", True):
      with on_side_stream(items[0]["x"].device, *([it["x"] for it in items] + [it["dy"] for it in items] +
                                                  [it["dw"] for it in items if "after" in it])):
        capi.gemm_wgrad_grouped(items, accumulate=True)
        for it in items:
          if "after" in it:       # behind the launch, on its stream (a folded one-tap separable layer splits its dw)
            it["after"]()
    else:
      capi.gemm_wgrad_grouped(items, accumulate=True)
      for it in items:
        if "after" in it:
          it["after"]()
    if self._pending is not None:
      for p, it in self._deferred:
        for q in (p,) + tuple(it.get("also", ())):
          if id(q) in self._pending:
            self._pending[id(q)] -= 1
    self._deferred = []


_TAPE_STACK = []


def backward_interleaved(tapes, streams):
  """Tape.backward for several tapes of the SAME structure (the halves of one batch) on several streams: closure i
  of every tape is issued before closure i + 1 of any, each on its tape's stream — the kernels of one half that keep
  the matrix pipes idle (LayerNorm, attention, dropout, embedding) run under the other half's GEMMs. The caller has
  set _SIDE_KEY_OVERRIDE: parameter-gradient launches of all tapes queue on one side stream. Single-process only
  (no gradient reducer watermark)."""
  depth = len(_TAPE_STACK)
  join_side_streams()
  capi.zero_arena_enter(depth)
  for t in tapes:
    assert t.on_done is None
    t._deferred, t._pending = [], None
    t._cdeferred, t._ckey = [], None
  outer = torch.cuda.current_stream()
  try:
    n = max(len(t.ops) for t in tapes)
    for i in range(n):
      for t, st in zip(tapes, streams):
        if i < len(t.ops):
          fn = t.ops[len(t.ops) - 1 - i][0]
          _TAPE_STACK.append(t)
          torch.cuda.set_stream(st)
          try:
            fn()
          finally:
            _TAPE_STACK.pop()
    for t, st in zip(tapes, streams):
      _TAPE_STACK.append(t)
      torch.cuda.set_stream(st)
      try:
        t.flush_deferred()
        t.flush_conv_wgrads()
      finally:
        _TAPE_STACK.pop()
  finally:
    torch.cuda.set_stream(outer)
    capi.zero_arena_leave(depth)
  for t in tapes:
    t.ops = []
  for st in streams:
    outer.wait_stream(st)
  join_side_streams()


def current_tape():
  """The tape whose backward pass is running — the innermost one (None outside Tape.backward)."""
  return _TAPE_STACK[-1] if _TAPE_STACK else None


class Act(object):
  """An activation tensor + its valid lengths + (optionally) its gradient."""
  __slots__ = ("data", "lens", "grad", "grad_init", "requires_grad", "res_grad", "mask_scale",
               "grad_masked", "bias_part", "bn_y", "bn_scale", "grad_event", "bn_full_rows")

  def __init__(self, data, lens=None, requires_grad=True):
    self.data, self.lens = data, lens
    self.grad, self.grad_init = None, False
    self.requires_grad = requires_grad
    self.res_grad = None   # gradient arriving through a residual connection (pre-norm blocks)
    # set by a ReLU (+ dropout) Dense layer on its OUTPUT: the consumer's data-gradient GEMM may
    # apply (data > 0) * mask_scale in its epilogue (and leave the bias-gradient partials in
    # bias_part); it then sets grad_masked and the producer skips its own activation backward
    self.mask_scale = None
    self.grad_masked = False
    self.bias_part = None
    # set by conv_bn_actv on the output of a single-input conv + BatchNorm + ReLU (+ dropout) layer: the
    # convolution output y. The data gradient of the NEXT layer's main convolution — the last
    # contribution to this activation's gradient — then applies the ReLU / dropout backward and leaves
    # the BatchNorm-backward partials (sum dz, sum dz * y) in bias_part (capi.conv1d_dgrad_bnact)
    self.bn_y = None
    self.bn_scale = 1.0
    # the producer's backward reads EVERY row of the finalised gradient (separable layers: their BatchNorm-backward
    # apply pass is not ragged): only a consumer that defines all rows may finalise it
    self.bn_full_rows = False
    # set by a data-gradient contribution enqueued on the SIDE stream (dense_residual.backward_end): the next
    # writer or reader of the gradient on another stream waits for it first
    self.grad_event = None

  def wait_grad(self):
    if self.grad_event is not None:
      _current_stream_obj().wait_event(self.grad_event)
      self.grad_event = None

  def grad_buffer(self):
    # a gradient that already carries its producer's activation backward takes no more addends
    assert not self.grad_masked, "a second consumer wrote to an activation whose gradient was finalised"
    if self.grad_event is not None:
      self.wait_grad()
    if self.grad is None:
      self.grad = torch.empty_like(self.data)
      self.grad_init = False
    return self.grad


def xavier_normal_conv(shape_dev):
  """tf.contrib.layers.xavier_initializer(uniform=False) for a conv1d kernel:
  truncated normal, stddev = sqrt(1.3 * 2 / (fan_in + fan_out)) with
  fan_in = K*Cin, fan_out = K*Cout. Device layout [K, Cout, Cin]."""
  K, Cout, Cin = shape_dev
  std = math.sqrt(1.3 * 2.0 / (K * Cin + K * Cout))
  w = torch.empty(shape_dev)
  torch.nn.init.trunc_normal_(w, 0.0, std, -2 * std, 2 * std)
  return w


def glorot_uniform_conv(shape_dev):
  K, Cout, Cin = shape_dev
  lim = math.sqrt(6.0 / (K * Cin + K * Cout))
  return (torch.rand(shape_dev) * 2 - 1) * lim


class ConvBN(object):
  """tf.layers.conv1d(use_bias=False) + tf.layers.batch_normalization: the pair of
  variables '<name>/kernel' and '<name>/bn/{gamma,beta,moving_mean,moving_variance}'
  (conv_blocks.py:195-227; residual branches :78-99)."""

  def __init__(self, store, name, bn_name, cin, cout, k, stride=1, dilation=1,
               padding="SAME", bn_momentum=0.9, bn_epsilon=1e-3, l2=0.0,
               initializer=xavier_normal_conv):
    self.name, self.cin, self.cout, self.k = name, cin, cout, k
    self.stride, self.dil, self.padding = stride, dilation, padding
    self.momentum, self.eps = bn_momentum, bn_epsilon
    self.kernel = store.add(name + "/kernel", (k, cout, cin), initializer, kind="conv", l2=l2)
    self.gamma = store.add(bn_name + "/gamma", (cout,), torch.ones(cout), kind="vector", l2=l2)
    self.beta = store.add(bn_name + "/beta", (cout,), torch.zeros(cout), kind="vector")
    dev = store.device
    self.moving_mean = torch.zeros(cout, dtype=torch.float32, device=dev)
    self.moving_var = torch.ones(cout, dtype=torch.float32, device=dev)
    if hasattr(store, "add_state"):
      store.add_state(bn_name + "/moving_mean", self.moving_mean)
      store.add_state(bn_name + "/moving_variance", self.moving_var)

  def out_geometry(self, tin):
    if self.padding == "SAME":
      return capi.same_padding(tin, self.k, self.stride, self.dil)
    return capi.valid_padding(tin, self.k, self.stride, self.dil)

  def conv_bn_stats(self, x, training):
    """Returns dict(y, scale, shift, mean, rstd, tout, pad_left)."""
    B, Tin, _ = x.data.shape
    tout, pl = self.out_geometry(Tin)
    dev = x.data.device
    C = self.cout
    stats = None
    if training:
      stats = torch.empty((capi.conv1d_num_mtiles(B, tout), 2, C), dtype=torch.float32,
                          device=dev)
    y = capi.conv1d_fwd(x.data, self.kernel.w16, stride=self.stride, dil=self.dil,
                        pad_left=pl, tout=tout, in_len=x.lens, stats=stats)
    return self.bn_from_stats(y, stats, B, tout, training, pl)

  def bn_outputs(self, y, training, tout, pl=0):
    """The record conv_bn_stats returns, with freshly allocated scale / shift (/ mean / rstd) vectors."""
    sc, sh, mean, rstd = bn_vectors(self.cout, y.device, training)
    return dict(y=y, scale=sc, shift=sh, mean=mean, rstd=rstd, tout=tout, pad_left=pl)

  def bn_from_stats(self, y, stats, B, tout, training, pl=0):
    """BatchNorm scale / shift (and the moving-statistics update) from the conv's fused partials."""
    d = self.bn_outputs(y, training, tout, pl)
    capi.bn_finalize(stats, B * tout, self.gamma.master, self.beta.master, self.eps,
                     self.momentum, training, self.moving_mean, self.moving_var, d["mean"], d["rstd"],
                     d["scale"], d["shift"])
    return d

  def trainable(self):
    return [self.kernel, self.gamma, self.beta]

  def backward_weights(self, inp, dy, f):
    x = inp.data
    tape = current_tape()
    units = ((self.cout + 127) // 128) * ((self.cin + 127) // 128) * ((self.k + 3) // 4)
    if CONV_WGRAD_UNIT_BUDGET > 0 and tape is not None and self.stride == 1 and self.k >= 2 and \
       units < CONV_WGRAD_UNIT_BUDGET and x.dim() == 3 and x.shape[0] <= 64 and self.cout >= 128 and self.cin >= 64 and \
       dy.is_contiguous() and x.stride(2) == 1 and x.stride(0) == x.shape[1] * x.stride(1):
      # (the envelope of the ping-pong weight-gradient kernel; everything else goes out alone as before)
      key = (tuple(x.shape), tuple(dy.shape), x.stride(1), self.k, self.dil, f["pad_left"],
             None if inp.lens is None else inp.lens.data_ptr())
      tape.defer_conv_wgrad(self.kernel, key, dict(x=x, dy=dy, dw=self.kernel.grad), units,
                            dict(K=self.k, stride=1, dil=self.dil, pad_left=f["pad_left"], in_len=inp.lens))
      return
    global _WGRAD_RR
    _WGRAD_RR += 1
    # (experiment, OS2S_WGRAD_STREAMS=2: the layers' weight gradients alternate between two side streams)
    with on_side_stream(dy.device, inp.data, dy, which=0 if WGRAD_STREAMS < 2 else (0, 2)[_WGRAD_RR & 1]):
      capi.conv1d_wgrad(inp.data, dy, self.k, stride=self.stride, dil=self.dil,
                        pad_left=f["pad_left"], in_len=inp.lens, out=self.kernel.grad,
                        accumulate=True)

  def backward_branch(self, inp, dy, f, final=False):
    """Weight and data gradients of the convolution given dy = d(conv output). Nothing in the
    rest of backward depends on the weight gradient, so it runs on a side stream: its
    workgroups fill the CUs the data-gradient kernels leave idle in their last, partial round of
    tiles, and it overlaps the HBM-bound BatchNorm backward kernels of the layers below. The
    main stream re-joins at the end of `Tape.backward`; the gradient reducer waits for the side
    stream on its own stream. final: this is the LAST contribution to inp's gradient (the main
    branch of the layer that follows inp's producer: every other consumer of inp is a residual
    branch of a LATER block end, whose backward has already run)."""
    self.backward_weights(inp, dy, f)
    self.backward_data(inp, dy, f, final, self.kernel.wt16)

  def backward_data(self, inp, dy, f, final, wt16):
    """The data-gradient half of backward_branch with the transposed kernel wt16 [K, Cin, Cout]."""
    if inp.requires_grad:
      g = inp.grad_buffer()
      tin = inp.data.shape[1]
      pl = (self.k - 1) * self.dil - f["pad_left"]
      if self.stride != 1:
        # a strided layer past the first one: dx[t] = sum_k dy_up[t - k dil + padL] w[k] with the output gradient
        # zero-upsampled to the input's resolution — the stride-1 data gradient of dy_up (stride x the work of
        # a dedicated kernel; no configuration of the reference strides anywhere but in its first layer)
        dy = capi.upsample_rows(dy.contiguous(), self.stride, (dy.shape[1] - 1) * self.stride + 1)
        capi.conv1d_fwd(dy, wt16, dil=self.dil, pad_left=pl, tout=tin, out=g,
                        accumulate=inp.grad_init, out_len=inp.lens)
        inp.grad_init = True
        return
      if final and FUSE_BN_BWD and inp.bn_y is not None and not inp.bn_full_rows and dy.is_contiguous() and \
         g.is_contiguous():
        # the producer's ReLU / dropout backward and its BatchNorm-backward partial sums ride in this
        # launch's epilogue: its own reduction pass (bn_act_bwd_reduce) is skipped
        inp.bias_part = capi.conv1d_dgrad_bnact(dy, wt16, g, dil=self.dil, pad_left=pl,
                                                accumulate=inp.grad_init, out_len=inp.lens,
                                                mask_ref=inp.data, mask_scale=inp.bn_scale, stat_ref=inp.bn_y)
        inp.grad_init = True
        inp.grad_masked = True
        return
      capi.conv1d_fwd(dy, wt16, dil=self.dil, pad_left=pl, tout=tin, out=g,
                      accumulate=inp.grad_init, out_len=inp.lens)
      inp.grad_init = True


def bn_vectors(C, dev, training):
  """scale, shift (and mean, rstd in training) of one BatchNorm: rows of ONE allocation (an allocation is 1.4 us of
  the Python thread, four of them per layer were 0.4 ms of a QuartzNet step)."""
  if not training:
    sc, sh = torch.empty((2, C), dtype=torch.float32, device=dev).unbind(0)
    return sc, sh, None, None
  return torch.empty((4, C), dtype=torch.float32, device=dev).unbind(0)


class SepConvBN(ConvBN):
  """tf.layers.separable_conv1d(use_bias=False) + batch norm (layer type "sep_conv1d",
  conv_blocks.py:11-16): variables '<name>/depthwise_kernel' [K, Cin] (TF: [K, Cin, 1]),
  '<name>/pointwise_kernel' [1, Cout, Cin] (TF: [1, Cin, Cout]) and the BN pair. Residual
  branches of a separable block are separable too with k = 1 (:66,79-85)."""

  def __init__(self, store, name, bn_name, cin, cout, k, stride=1, dilation=1,
               padding="SAME", bn_momentum=0.9, bn_epsilon=1e-3, l2=0.0,
               initializer=xavier_normal_conv):
    self.name, self.cin, self.cout, self.k = name, cin, cout, k
    self.stride, self.dil, self.padding = stride, dilation, padding
    self.momentum, self.eps = bn_momentum, bn_epsilon

    def dw_init(shape):   # same initializer family over the TF shape [K, Cin, 1]
      return initializer((shape[0], 1, shape[1]))[:, 0, :] if shape[0] * shape[1] > 0 else torch.zeros(shape)

    self.depthwise = store.add(name + "/depthwise_kernel", (k, cin), dw_init, kind="vector", l2=l2)
    self.kernel = store.add(name + "/pointwise_kernel", (1, cout, cin), initializer, kind="conv", l2=l2)
    self.gamma = store.add(bn_name + "/gamma", (cout,), torch.ones(cout), kind="vector", l2=l2)
    self.beta = store.add(bn_name + "/beta", (cout,), torch.zeros(cout), kind="vector")
    dev = store.device
    self.moving_mean = torch.zeros(cout, dtype=torch.float32, device=dev)
    self.moving_var = torch.ones(cout, dtype=torch.float32, device=dev)
    if hasattr(store, "add_state"):
      store.add_state(bn_name + "/moving_mean", self.moving_mean)
      store.add_state(bn_name + "/moving_variance", self.moving_var)

  def conv_bn_stats(self, x, training):
    B, Tin, _ = x.data.shape
    tout, pl = self.out_geometry(Tin)
    dev = x.data.device
    C = self.cout
    stats = None
    if training:
      stats = torch.empty((capi.conv1d_num_mtiles(B, tout), 2, C), dtype=torch.float32, device=dev)
    wt_eff = None
    if self.folded():
      # one tap: the per-channel scale goes into the pointwise kernel (y = x (W diag d)^T), x is not copied
      if getattr(self, "_fold_bufs", None) is None:       # (the layer's own: rewritten by every forward)
        self._fold_bufs = (torch.empty((1, C, self.cin), dtype=torch.bfloat16, device=dev),
                           torch.empty((1, self.cin, C), dtype=torch.bfloat16, device=dev))
      w_eff, wt_eff = capi.pointwise_fold(self.kernel.master, self.depthwise.master, out=self._fold_bufs)
      z = None
      y = capi.conv1d_fwd(x.data, w_eff, pad_left=0, tout=tout, in_len=x.lens, stats=stats)
    else:
      z = capi.depthwise_conv1d_fwd(x.data, self.depthwise.master, stride=self.stride, dil=self.dil,
                                    pad_left=pl, tout=tout, in_len=x.lens)
      y = capi.conv1d_fwd(z, self.kernel.w16, pad_left=0, tout=tout, stats=stats)
    sc, sh, mean, rstd = bn_vectors(C, dev, training)
    capi.bn_finalize(stats, B * tout, self.gamma.master, self.beta.master, self.eps,
                     self.momentum, training, self.moving_mean, self.moving_var, mean, rstd,
                     sc, sh)
    return dict(y=y, scale=sc, shift=sh, mean=mean, rstd=rstd, tout=tout, pad_left=pl,
                z=z if training else None, wt_eff=wt_eff if training else None)

  def folded(self):
    """A one-tap separable layer (the residual branches of a separable block) runs as ONE 1x1 convolution with the
    depthwise scale folded into the pointwise kernel: no scaled copy of the input, no depthwise launches; the two
    variables' gradients come out of the 1x1 weight gradient of the layer's input (os2s_pointwise_fold_bwd)."""
    return FOLD_SEP_K1 and self.k == 1 and self.stride == 1 and self.cin % 8 == 0 and self.cout % 8 == 0

  def trainable(self):
    return [self.depthwise, self.kernel, self.gamma, self.beta]

  def backward_branch(self, inp, dy, f, final=False):
    """final: as ConvBN.backward_branch — this is the last contribution to inp's gradient; when inp is the output
    of a single-input conv + BatchNorm + ReLU layer its activation backward and BatchNorm-backward partial sums ride
    in the store phase of the depthwise data gradient (capi.depthwise_dgrad_bnact)."""
    if f.get("wt_eff") is not None:
      # G = dy^T x joins the block's grouped pointwise weight gradients (below); the launch's tail splits it into
      # the two variables' gradients
      x = inp.data
      rows = x.shape[0] * x.shape[1]
      gw = capi.zero_scratch(self.kernel.grad.shape, dy.device)

      def split(gw=gw):
        capi.pointwise_fold_bwd(gw, self.kernel.master, self.depthwise.master, self.kernel.grad, self.depthwise.grad)
      if GROUP_POINTWISE_WGRAD and current_tape() is not None and self.cin >= 128 and self.cout >= 128 and \
         rows >= 2048 and x.is_contiguous() and dy.is_contiguous():
        tape = current_tape()
        for p in (self.depthwise,):       # the scale's gradient is final with the same launch
          if tape._pending is not None and id(p) in tape._pending:
            tape._pending[id(p)] += 1
        tape.defer_wgrad(self.kernel, dict(x=x.view(rows, self.cin), dy=dy.view(rows, self.cout),
                                           dw=gw.view(self.cout, self.cin), after=split, also=(self.depthwise,)),
                         group=POINTWISE_WGRAD_GROUP)
      else:
        with on_side_stream(dy.device, x, dy, gw):
          capi.conv1d_wgrad(x, dy, 1, pad_left=0, in_len=inp.lens, out=gw, accumulate=True)
          split()
      wt_eff, f["wt_eff"] = f["wt_eff"], None
      self.backward_data(inp, dy, f, final, wt_eff)
      return
    z = f["z"]
    rows = z.shape[0] * z.shape[1]
    units = ((self.cout + 255) // 256) * ((self.cin + 255) // 256)
    if GROUP_POINTWISE_WGRAD and current_tape() is not None and units < 32 and self.cin >= 128 and \
       self.cout >= 128 and rows >= 2048 and self.cin % 8 == 0 and self.cout % 8 == 0 and \
       z.is_contiguous() and dy.is_contiguous():
      # the pointwise kernel gradient is a TN GEMM over the B * T rows with 1 ... 16 output tiles of 256 x 256:
      # alone it runs on the lockstep kernel with the batch split over fp32 atomics (42 us for 3.5 GFLOP);
      # several of them — consecutive layers see the same rows — go out as ONE launch of the ping-pong
      # TN-GEMM kernel (Tape.defer_wgrad, as the Transformer's 1024 x 1024 projections do): deterministic
      current_tape().defer_wgrad(self.kernel, dict(x=z.view(rows, self.cin), dy=dy.view(rows, self.cout),
                                                   dw=self.kernel.grad.view(self.cout, self.cin)),
                                 group=POINTWISE_WGRAD_GROUP)
    else:
      with on_side_stream(dy.device, z, dy):          # parameter gradients: see ConvBN.backward_branch
        capi.conv1d_wgrad(z, dy, 1, pad_left=0, out=self.kernel.grad, accumulate=True)
    dz = capi.conv1d_fwd(dy, self.kernel.wt16, pad_left=0, tout=z.shape[1])
    f["z"] = None
    with on_side_stream(dy.device, inp.data, dz):
      capi.depthwise_conv1d_wgrad(inp.data, dz, self.depthwise.grad, stride=self.stride, dil=self.dil,
                                  pad_left=f["pad_left"], in_len=inp.lens)
    if inp.requires_grad:
      if final and FUSE_BN_BWD and SEP_FUSE_BN_BWD and inp.bn_y is not None and dz.is_contiguous() and \
         capi.depthwise_dgrad_bnact_supported(self.k, self.stride, self.dil) and \
         (inp.grad is None or not inp.grad_init or inp.grad.is_contiguous()):
        inp.wait_grad()
        prev = inp.grad if inp.grad_init else None
        out = prev if prev is not None else torch.empty_like(inp.data)
        inp.bias_part = capi.depthwise_dgrad_bnact(dz, self.depthwise.master, out, pad_left=(self.k - 1) - f["pad_left"],
                                                   out_len=inp.lens, mask_ref=inp.data, mask_scale=inp.bn_scale,
                                                   stat_ref=inp.bn_y, addend=prev)
        inp.grad, inp.grad_init, inp.grad_masked = out, True, True
        return
      if self.stride != 1:      # as ConvBN.backward_branch: the stride-1 data gradient of the zero-upsampled dz
        dz = capi.upsample_rows(dz.contiguous(), self.stride, (dz.shape[1] - 1) * self.stride + 1)
      tin = inp.data.shape[1]
      dx = capi.depthwise_conv1d_fwd(dz, self.depthwise.master, dil=self.dil,
                                     pad_left=(self.k - 1) * self.dil - f["pad_left"], tout=tin,
                                     out_len=inp.lens, flip=True)
      accumulate_grad(inp, dx)


class DepthwiseBN(ConvBN):
  """Depthwise ("row" / in-plane) convolution over time + batch norm: the row_conv layer of
  DeepSpeech2 (encoders/ds2_encoder.py:38-83: tf.nn.depthwise_conv2d with a [width, 1, C, 1] filter,
  SAME padding, then tf.layers.batch_normalization and the activation). Variables '<name>/w'
  [K, C] (TF: [K, 1, C, 1]) and '<name>/bn/{gamma,beta}'. Same branch interface as ConvBN, so
  conv_bn_actv runs it."""

  def __init__(self, store, name, channels, k, bn_momentum=0.99, bn_epsilon=1e-3, l2=0.0):
    self.name, self.cin, self.cout, self.k = name, channels, channels, k
    self.stride, self.dil, self.padding = 1, 1, "SAME"
    self.momentum, self.eps = bn_momentum, bn_epsilon

    def init(shape):   # tf.get_variable default: glorot_uniform over [K, 1, C, 1]
      lim = math.sqrt(6.0 / (shape[0] * shape[1] + shape[0]))
      return (torch.rand(shape) * 2 - 1) * lim

    self.depthwise = store.add(name + "/w", (k, channels), init, kind="vector", l2=l2)
    self.gamma = store.add(name + "/bn/gamma", (channels,), torch.ones(channels), kind="vector", l2=l2)
    self.beta = store.add(name + "/bn/beta", (channels,), torch.zeros(channels), kind="vector")
    dev = store.device
    self.moving_mean = torch.zeros(channels, dtype=torch.float32, device=dev)
    self.moving_var = torch.ones(channels, dtype=torch.float32, device=dev)
    if hasattr(store, "add_state"):
      store.add_state(name + "/bn/moving_mean", self.moving_mean)
      store.add_state(name + "/bn/moving_variance", self.moving_var)

  def conv_bn_stats(self, x, training):
    B, Tin, C = x.data.shape
    tout, pl = self.out_geometry(Tin)
    dev = x.data.device
    y = capi.depthwise_conv1d_fwd(x.data, self.depthwise.master, stride=1, dil=1, pad_left=pl,
                                  tout=tout, in_len=x.lens)
    stats = capi.bn_stats(y.view(B * tout, C)) if training else None
    sc, sh, mean, rstd = bn_vectors(C, dev, training)
    capi.bn_finalize(stats, B * tout, self.gamma.master, self.beta.master, self.eps,
                     self.momentum, training, self.moving_mean, self.moving_var, mean, rstd, sc, sh)
    return dict(y=y, scale=sc, shift=sh, mean=mean, rstd=rstd, tout=tout, pad_left=pl)

  def trainable(self):
    return [self.depthwise, self.gamma, self.beta]

  def backward_branch(self, inp, dy, f):
    capi.depthwise_conv1d_wgrad(inp.data, dy, self.depthwise.grad, stride=1, dil=1,
                                pad_left=f["pad_left"], in_len=inp.lens)
    if inp.requires_grad:
      tin = inp.data.shape[1]
      dx = capi.depthwise_conv1d_fwd(dy, self.depthwise.master, dil=1,
                                     pad_left=(self.k - 1) - f["pad_left"], tout=tin,
                                     out_len=inp.lens, flip=True)
      accumulate_grad(inp, dx)


def conv_bn_res_bn_actv(main, res_branches, x, res_inputs, out_lens, activation_fn,
                        training, tape, keep_prob=1.0, seed=0, mask_output=True,
                        drop_block_prob=0.0, drop_block=False, res_fw=None):
  """act(BN(conv(x)) + sum_i BN_i(conv1x1_i(res_i))) -> dropout -> mask.

  main: ConvBN; res_branches: list of ConvBN (1x1) matching res_inputs (list of Act).
  With no residual branches this is conv_bn_actv (conv_blocks.py:170-232); with them
  conv_bn_res_bn_actv (:61-168). Dropout is the tf.nn.dropout the encoder applies
  to the block output (tdnn_encoder.py:255).

  Stochastic block dropping (:156-164): in training one uniform draw per block and step; below
  drop_block_prob the output is act(sum of the residual branches) — the main branch is still
  evaluated (its BatchNorm moving statistics update: `bn` is built outside the tf.cond) but takes
  no part in the output and gets no gradient. In eval, drop_block selects the same path."""
  act = act_id(activation_fn)
  branches = [main] + list(res_branches)
  inputs = [x] + list(res_inputs)
  fw_main = main.conv_bn_stats(x, training)
  if res_fw is not None:         # launched earlier on the side stream (launch_residual_early)
    join_side_streams()
    global _FWD_SIDE_BUSY
    _FWD_SIDE_BUSY = False
    fw = [fw_main] + res_fw
  else:
    fw = [fw_main] + grouped_conv1x1_bn_stats(res_branches, res_inputs, training)
  dropped = False
  if res_branches and drop_block_prob > 0:
    if training:
      import random
      dropped = random.Random(int(seed) * 2654435761 % (2 ** 31)).random() < drop_block_prob
    else:
      dropped = bool(drop_block)
  if dropped:
    tout0 = fw[0]["tout"]
    branches, inputs, fw = branches[1:], inputs[1:], fw[1:]
    fw[0]["tout"] = tout0
  B = x.data.shape[0]
  tout, C = fw[0]["tout"], main.cout
  out = torch.empty((B, tout, C), dtype=torch.bfloat16, device=x.data.device)
  lens = out_lens if mask_output else None
  capi.bn_act_fwd([f["y"] for f in fw], [f["scale"] for f in fw], [f["shift"] for f in fw],
                  out, lens, act, keep_prob if training else 1.0, seed)
  result = Act(out, out_lens if mask_output else None)
  if not (training and tape is not None):
    return result
  if FUSE_BN_BWD and len(fw) == 1 and act == 1 and (type(main) is ConvBN or (SEP_FUSE_BN_BWD and type(main) is SepConvBN)) \
     and not dropped and (lens is None or main.stride == 1):
    # single-input conv + BatchNorm + ReLU (+ dropout): out is zero exactly where the gradient is
    result.bn_y = fw[0]["y"]
    result.bn_scale = 1.0 / (keep_prob if training else 1.0)
    result.bn_full_rows = type(main) is not ConvBN

  def backward():
    result.wait_grad()
    dout = result.grad
    if dout is None and drop_block_prob > 0:
      return      # every consumer of this layer sat in a dropped block: its gradient is zero
    assert dout is not None, "no gradient reached " + main.name
    rows = B * tout
    J = len(branches)
    c1 = torch.empty((J, C), dtype=torch.float32, device=out.device)
    c2 = torch.empty((J, C), dtype=torch.float32, device=out.device)
    dz_to_len = False
    if result.grad_masked:
      # the data gradient that finalised `dout` already applied the ReLU / dropout backward and left
      # (sum dz, sum dz * y) per 128-row window: no reduction pass (ConvBN.backward_branch, final=True)
      dz, partial = dout, result.bias_part
      result.bias_part = None
      capi.bn_bwd_finalize_raw(partial, rows, fw[0]["mean"], fw[0]["rstd"], branches[0].gamma.grad,
                               branches[0].beta.grad, True, c1[0], c2[0])
      dz_to_len = lens is not None       # rows past the sequence ends were not written: zero, unread
    else:
      dz = torch.empty_like(out)
      partial = torch.empty((capi.bn_act_bwd_num_parts(rows), 1 + J, C), dtype=torch.float32,
                            device=out.device)
      capi.bn_act_bwd_reduce(dout, out, [f["y"] for f in fw], [f["mean"] for f in fw],
                             [f["rstd"] for f in fw], dz, partial, lens, act, keep_prob, seed)
      # dgamma / dbeta / the two means of every branch in ONE launch
      capi.bn_bwd_finalize_multi(partial, rows, [br.gamma.grad for br in branches],
                                 [br.beta.grad for br in branches], True, c1, c2)
    result.grad = None
    grouped = []        # plain 1x1 residual branches: their data gradients go out in one launch
    wgrouped = []       # ... and so do their weight gradients
    # (one launch = one (B, T) and one length vector: the predicate of the forward grouping)
    plain = [j for j in range(1, len(branches)) if _is_plain_1x1(branches[j])]
    can_group = len(branches) > 2 and len(plain) >= 2 and \
        len({tuple(inputs[j].data.shape[:2]) for j in plain}) == 1 and \
        len({id(inputs[j].lens) for j in plain}) == 1
    for j, (br, inp, f) in enumerate(zip(branches, inputs, fw)):
      dy = torch.empty_like(f["y"])
      # a plain convolution's gradients read dy at most (K-1)*dilation rows past the sequence end
      # (wgrad pairs it with the masked input, dgrad only produces live rows): the rest is not computed
      ragged = lens is not None and type(br) is ConvBN and br.stride == 1
      capi.bn_bwd_apply(dz, f["y"], br.gamma.master, f["mean"], f["rstd"], c1[j], c2[j], dy,
                        out_len=lens if ragged else None, margin=(br.k - 1) * br.dil,
                        dz_to_len=dz_to_len and ragged)
      f["y"] = None
      if can_group and j in plain:
        if GROUP_WGRAD:
          wgrouped.append((br, inp, dy))
        else:
          br.backward_weights(inp, dy, f)
        if inp.requires_grad:
          grouped.append((br, inp, dy))
      elif j == 0 and br is main and type(br) is ConvBN:
        # the last contribution to its input's gradient — unless the same activation also feeds a
        # residual branch of THIS call (a residual block with repeat = 1), which runs after it.
        # (With the block dropped, branches[0] is a RESIDUAL branch whose input still has
        # later-running consumers: never final.)
        br.backward_branch(inp, dy, f, final=all(r is not inp for r in inputs[1:]))
      elif j == 0 and br is main and type(br) is SepConvBN:
        br.backward_branch(inp, dy, f, final=all(r is not inp for r in inputs[1:]))
      else:
        br.backward_branch(inp, dy, f)
    if wgrouped:
      # the K = 1 weight gradients of all branches in one launch, on the side stream like every
      # parameter gradient (ConvBN.backward_branch)
      with on_side_stream(dz.device, *([w[1].data for w in wgrouped] + [w[2] for w in wgrouped])):
        for i0 in range(0, len(wgrouped), 16):
          capi.conv1x1_wgrad_grouped([dict(x=inp.data, dy=dy, dw=br.kernel.grad)
                                      for br, inp, dy in wgrouped[i0:i0 + 16]], in_len=wgrouped[0][1].lens,
                                     pingpong=GROUP_WGRAD_PP)
    if grouped:
      items = []
      for br, inp, dy in grouped:
        items.append(dict(x=dy, w=br.kernel.wt16, y=inp.grad_buffer(), accumulate=inp.grad_init))
        inp.grad_init = True
      capi.conv1x1_fwd_grouped(items, out_len=grouped[0][1].lens)

  tape.record(backward, [p for br in [main] + list(res_branches) for p in br.trainable()])
  return result


def launch_dense_residual(dpass, k, x):
  """Registers block k's input `x` as source k of the dense-residual pass and evaluates block end k's residual
  sum — it depends on the block INPUTS only — on the side stream, next to the block's own layers. Returns the
  record conv_bn_dres_actv consumes (it joins the side stream first)."""
  if not DRES_FWD_SIDE or _side_stream(x.data.device, 1) is None:
    dpass.add_source(x)
    return dpass.forward_end(k)
  global _FWD_SIDE_BUSY
  _FWD_SIDE_BUSY = True
  with on_side_stream(x.data.device, x.data, which=1) as ctx:
    dpass.add_source(x)
    fw = dpass.forward_end(k)
    ctx.hand_over(fw["y"])
  return fw


def conv_bn_dres_actv(main, x, dfw, out_lens, activation_fn, training, tape, keep_prob=1.0, seed=0,
                      mask_output=True):
  """conv_bn_res_bn_actv (conv_blocks.py:61-168) for a dense-residual block end whose residual branches come as
  ONE tensor: act(BN(conv(x)) + R + shift) -> dropout -> mask, R / shift = dense_residual.forward_end (the sum of
  the BatchNorm'd 1x1 branches, no branch tensor materialised). Backward hands the gradient at the sum to
  dense_residual.backward_end (side stream) and runs the main branch as conv_bn_res_bn_actv does."""
  act = act_id(activation_fn)
  fw = main.conv_bn_stats(x, training)
  join_side_streams()
  global _FWD_SIDE_BUSY
  _FWD_SIDE_BUSY = False
  B = x.data.shape[0]
  tout, C = fw["tout"], main.cout
  out = torch.empty((B, tout, C), dtype=torch.bfloat16, device=x.data.device)
  lens = out_lens if mask_output else None
  capi.bn_act_fwd([fw["y"], dfw["y"]], [fw["scale"], dfw["scale"]], [fw["shift"], dfw["shift"]],
                  out, lens, act, keep_prob if training else 1.0, seed)
  result = Act(out, lens)
  if not (training and tape is not None):
    return result
  dpass, k = dfw["dres"], dfw["k"]
  dfw = None
  if FUSE_BN_BWD and DRES_FUSE_BN_BWD and act == 1 and type(main) is ConvBN and (lens is None or main.stride == 1):
    # ONE BatchNorm'd tensor enters the sum (the residual sum R carries no statistics of its own): as for a
    # single-input layer, the data gradient that finalises this output's gradient may apply the ReLU / dropout
    # backward and leave (sum dz, sum dz * y_main) in its epilogue (ConvBN.backward_branch, final=True)
    result.bn_y = fw["y"]
    result.bn_scale = 1.0 / (keep_prob if training else 1.0)

  def backward():
    result.wait_grad()
    dout = result.grad
    assert dout is not None, "no gradient reached " + main.name
    rows = B * tout
    c1 = torch.empty((1, C), dtype=torch.float32, device=out.device)
    c2 = torch.empty((1, C), dtype=torch.float32, device=out.device)
    dz_to_len = False
    if result.grad_masked:
      dz, partial = dout, result.bias_part
      result.bias_part = None
      capi.bn_bwd_finalize_raw(partial, rows, fw["mean"], fw["rstd"], main.gamma.grad, main.beta.grad, True, c1[0], c2[0])
      dz_to_len = lens is not None       # rows past the sequence ends were not written: never read below
    else:
      dz = torch.empty_like(out)
      partial = torch.empty((capi.bn_act_bwd_num_parts(rows), 2, C), dtype=torch.float32, device=out.device)
      capi.bn_act_bwd_reduce(dout, out, [fw["y"]], [fw["mean"]], [fw["rstd"]], dz, partial, lens, act, keep_prob, seed)
      capi.bn_bwd_finalize_multi(partial, rows, [main.gamma.grad], [main.beta.grad], True, c1, c2)
    result.grad = None
    # every residual branch (kernel / gamma / beta gradients) and the finished data gradient of source k
    src = dpass.acts[k]
    with on_side_stream(dz.device, dz, c1, which=1) as ctx:
      dpass.backward_end(k, dz, c1[0], dz_lens=lens if dz_to_len else None)
      if src.requires_grad:
        ctx.hand_over(src.grad)
        if ctx.side is not None:
          src.grad_event = torch.cuda.Event()
          src.grad_event.record(ctx.side)
    dy = torch.empty_like(fw["y"])
    ragged = lens is not None and type(main) is ConvBN and main.stride == 1
    capi.bn_bwd_apply(dz, fw["y"], main.gamma.master, fw["mean"], fw["rstd"], c1[0], c2[0], dy,
                      out_len=lens if ragged else None, margin=(main.k - 1) * main.dil,
                      dz_to_len=dz_to_len and ragged)
    fw["y"] = None
    if type(main) is ConvBN:
      # the last contribution to its input's gradient unless that input is itself a source of this block end
      main.backward_branch(x, dy, fw, final=all(x is not a for a in dpass.acts[:k + 1]))
    else:
      main.backward_branch(x, dy, fw)

  tape.record(backward, [p for br in [main] + list(dpass.plan.ends[k].branches) for p in br.trainable()])
  return result


_FWD_SIDE_BUSY = False


def forward_side_busy():
  """True while residual branches launched early are (possibly) still running next to the main
  stream's forward kernels (bench.py does not time those launches: not alone on the GPU)."""
  return _FWD_SIDE_BUSY


def launch_residual_early(res_branches, res_inputs, training):
  """The 1x1 residual branches of a block END (conv + BatchNorm statistics of every dense-residual
  input, conv_blocks.py:78-100) depend only on the block INPUTS, so they can run while the block's
  own layers do: enqueued on the side stream, they use the CUs a 142-284-tile convolution leaves
  idle (an HBM-bound grouped launch next to MFMA-bound ones). Returns what
  grouped_conv1x1_bn_stats returns, for conv_bn_res_bn_actv(..., res_fw=...) which joins the side
  stream before its BatchNorm sum; None when there is no side stream (OS2S_WGRAD_STREAM=0)."""
  if not res_branches or _side_stream(res_inputs[0].data.device) is None:
    return None
  global _FWD_SIDE_BUSY
  _FWD_SIDE_BUSY = True
  with on_side_stream(res_inputs[0].data.device, *[r.data for r in res_inputs]) as ctx:
    fw = grouped_conv1x1_bn_stats(res_branches, res_inputs, training)
    for f in fw:
      ctx.hand_over(*[v for v in f.values() if torch.is_tensor(v)])
  return fw


def _is_plain_1x1(br):
  return type(br) is ConvBN and br.k == 1 and br.stride == 1


def grouped_conv1x1_bn_stats(branches, inputs, training):
  """conv_bn_stats of the 1x1 residual branches of a block end, the convolutions in ONE launch
  (os2s_conv1x1_fwd_grouped) when there are several plain 1x1 branches; anything else (separable
  branches, a single branch) goes through the branch's own conv_bn_stats."""
  plain = [i for i, br in enumerate(branches) if _is_plain_1x1(br)]
  if len(plain) < 2 or len({tuple(inputs[i].data.shape[:2]) for i in plain}) != 1 or \
     len({id(inputs[i].lens) for i in plain}) != 1:
    return [br.conv_bn_stats(inp, training) for br, inp in zip(branches, inputs)]
  out = [None] * len(branches)
  items = []
  for i in plain:
    br, x = branches[i], inputs[i]
    B, T, _ = x.data.shape
    dev = x.data.device
    stats = torch.empty((capi.conv1d_num_mtiles(B, T), 2, br.cout), dtype=torch.float32,
                        device=dev) if training else None
    y = torch.empty((B, T, br.cout), dtype=torch.bfloat16, device=dev)
    items.append(dict(x=x.data, w=br.kernel.w16, y=y, stats=stats))
    out[i] = (y, stats)
  capi.conv1x1_fwd_grouped(items, in_len=inputs[plain[0]].lens)
  # the BatchNorm finalizes of the grouped branches in ONE launch too (same Cout, same window count)
  B, T = inputs[plain[0]].data.shape[:2]
  same = len({branches[i].cout for i in plain}) == 1 and len(plain) <= 16 and \
      len({(branches[i].eps, branches[i].momentum) for i in plain}) == 1
  if same:
    fin = []
    for i in plain:
      br = branches[i]
      y, stats = out[i]
      d = br.bn_outputs(y, training, T)
      fin.append(dict(partial=stats, gamma=br.gamma.master, beta=br.beta.master, moving_mean=br.moving_mean,
                      moving_var=br.moving_var, mean_out=d["mean"], rstd_out=d["rstd"], scale_out=d["scale"],
                      shift_out=d["shift"]))
      out[i] = d
    br0 = branches[plain[0]]
    capi.bn_finalize_multi(fin, B * T, br0.eps, br0.momentum, training)
  for i, br in enumerate(branches):
    if out[i] is None:
      out[i] = br.conv_bn_stats(inputs[i], training)
    elif not same:
      y, stats = out[i]
      out[i] = br.bn_from_stats(y, stats, inputs[i].data.shape[0], inputs[i].data.shape[1], training)
  return out


class ConvOnly(ConvBN):
  """tf.layers.conv1d(use_bias=False) alone — the layer of conv_actv (conv_blocks.py:17-58), TDNNEncoder with
  normalization=None: one variable, '<name>/kernel'."""

  def __init__(self, store, name, cin, cout, k, stride=1, dilation=1, padding="SAME", l2=0.0,
               initializer=xavier_normal_conv):
    self.name, self.cin, self.cout, self.k = name, cin, cout, k
    self.stride, self.dil, self.padding = stride, dilation, padding
    self.kernel = store.add(name + "/kernel", (k, cout, cin), initializer, kind="conv", l2=l2)

  def trainable(self):
    return [self.kernel]


class ConvSampleNorm(ConvOnly):
  """The layer of conv_ln_actv / conv_in_actv (conv_blocks.py:234-309): tf.layers.conv1d(use_bias=False)
  followed by tf.contrib.layers.layer_norm (mode 1: per sample over all T x C values of the padded tensor,
  epsilon 1e-12, begin_norm_axis = 1 / begin_params_axis = -1 as the reference leaves them) or
  tf.contrib.layers.instance_norm (mode 0: per sample and channel over T, epsilon 1e-6). Variables:
  '<name>/kernel' and gamma / beta [C] under the scope tf.contrib gives them — 'LayerNorm' / 'InstanceNorm',
  uniquified '_1', '_2', ... in creation order, directly under the encoder's scope (the reference opens no
  per-layer scope around them)."""
  MODES = {"instance_norm": (0, 1e-6, "InstanceNorm"), "layer_norm": (1, 1e-12, "LayerNorm")}

  def __init__(self, store, name, norm_scope, kind, cin, cout, k, stride=1, dilation=1, padding="SAME", l2=0.0,
               initializer=xavier_normal_conv):
    super(ConvSampleNorm, self).__init__(store, name, cin, cout, k, stride, dilation, padding, l2, initializer)
    self.mode, self.eps, _ = self.MODES[kind]
    self.gamma = store.add(norm_scope + "/gamma", (cout,), torch.ones(cout), kind="vector")
    self.beta = store.add(norm_scope + "/beta", (cout,), torch.zeros(cout), kind="vector")

  def trainable(self):
    return [self.kernel, self.gamma, self.beta]


def conv_actv(layer, x, out_lens, activation_fn, training, tape, keep_prob=1.0, seed=0, mask_output=True):
  """act(conv(x)) -> dropout -> mask: conv_actv (conv_blocks.py:17-58) + the encoder's tf.nn.dropout and mask
  (tdnn_encoder.py:204-205, 255). Runs on the BatchNorm + activation kernels with the identity transform
  (scale 1, shift 0): one tested pass produces activation, dropout and the masked store, and its backward twin
  produces dz; the convolution's output takes the place of BatchNorm's input."""
  act = act_id(activation_fn)
  B, Tin, _ = x.data.shape
  tout, pl = layer.out_geometry(Tin)
  dev = x.data.device
  C = layer.cout
  y = capi.conv1d_fwd(x.data, layer.kernel.w16, stride=layer.stride, dil=layer.dil, pad_left=pl, tout=tout,
                      in_len=x.lens)
  # conv_ln_actv / conv_in_actv: the per-sample normalisation sits between the convolution and the activation
  norm = isinstance(layer, ConvSampleNorm)
  if norm:
    z, mean, rstd = capi.sample_norm_fwd(y, layer.gamma.master, layer.beta.master, layer.mode, layer.eps)
  else:
    z = y
  one, zero = torch.ones(C, dtype=torch.float32, device=dev), torch.zeros(C, dtype=torch.float32, device=dev)
  out = torch.empty((B, tout, C), dtype=torch.bfloat16, device=dev)
  lens = out_lens if mask_output else None
  keep = keep_prob if training else 1.0
  capi.bn_act_fwd([z], [one], [zero], out, lens, act, keep, seed)
  result = Act(out, lens)
  if not (training and tape is not None):
    return result

  def backward():
    dout = result.grad
    assert dout is not None, "no gradient reached " + layer.name
    dz = torch.empty_like(out)
    partial = torch.empty((capi.bn_act_bwd_num_parts(B * tout), 2, C), dtype=torch.float32, device=dev)
    # identity "BatchNorm": mean 0, rstd 1 — only dz = dout * act'(out) * dropout' is used, the partial sums
    # of the BatchNorm parameters are ignored
    capi.bn_act_bwd_reduce(dout, out, [z], [zero], [one], dz, partial, lens, act, keep, seed)
    result.grad = None
    if norm:
      # the statistics run over the PADDED tensor (as the reference's): the gradient reaches rows past the
      # sequence end too, and from there the kernel taps that still see live input rows
      dz = capi.sample_norm_bwd(dz, y, layer.gamma.master, mean, rstd, layer.mode, layer.gamma.grad,
                                layer.beta.grad)
    layer.backward_branch(x, dz, dict(pad_left=pl), final=False)

  tape.record(backward, layer.trainable())
  return result


def conv_bn_actv(layer, x, out_lens, activation_fn, training, tape, keep_prob=1.0, seed=0,
                 mask_output=True):
  return conv_bn_res_bn_actv(layer, [], x, [], out_lens, activation_fn, training, tape,
                             keep_prob, seed, mask_output)


def accumulate_grad(x, g):
  """x.grad (+)= g for an Act consumed by several ops."""
  if not x.requires_grad:
    return
  if x.grad_init and x.grad is not None:
    capi.add_bf16(x.grad, g, out=x.grad)
  else:
    x.grad, x.grad_init = g, True


def reshape_act(x, shape, tape, lens=None):
  """A view of an Act with another shape ([B,T,C] <-> [B*T,C]); gradients flow back."""
  v = Act(x.data.view(*shape), lens, requires_grad=x.requires_grad)
  if tape is not None and x.requires_grad:
    def backward():
      if v.grad is not None:
        accumulate_grad(x, v.grad.reshape(x.data.shape))
      v.grad = None
    tape.record(backward)
  return v

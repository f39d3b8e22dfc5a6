"""optimize_loss — the train-op factory of open_seq2seq/optimizers/optimizers.py:107-286,
re-hosted on the flat-buffer multi-tensor HIP kernels.

Reference order of one update (optimizers.py:208-270 + mp_wrapper.py:44-122):
  grads of (loss * loss_scale) -> fp32 -> + scale*d(reg) -> * 1/scale ->
  all-reduce average (Horovod) -> clip | LARC -> NaN/Inf + amax on the result ->
  loss-scaler update -> cond(skip) { optimizer apply on fp32 masters; cast to the
  half-precision copies }.
Here the backward pass already wrote fp32 gradients of the SCALED loss into the
flat gradient buffer (summed over ranks by RCCL); everything after that is
os2s_opt_step (no host sync: the skip decision, the loss scale and the global
step live on the device).
"""
import six

import torch

from .. import capi
from ..utils.utils import check_params
from . import lr_policies
from .automatic_loss_scaler import AutomaticLossScaler
from .novograd import NovoGrad

OPTIMIZER_CLS_NAMES = {"Adagrad": None, "Adam": 3, "Ftrl": None, "Momentum": 1,
                       "RMSProp": None, "SGD": 0, "AdamW": None, "NovoGrad": 2,
                       "LazyAdam": 3}

OPTIMIZER_SUMMARIES = ["learning_rate", "gradients", "gradient_norm", "global_gradient_norm",
                       "variables", "variable_norm", "larc_summaries", "loss_scale"]


def _optimizer_id(optimizer):
  if isinstance(optimizer, six.string_types):
    if optimizer not in OPTIMIZER_CLS_NAMES:
      raise ValueError("Optimizer name should be one of [{}], you provided {}.".format(
          ", ".join(OPTIMIZER_CLS_NAMES), optimizer))
    oid = OPTIMIZER_CLS_NAMES[optimizer]
  else:
    name = getattr(optimizer, "__name__", type(optimizer).__name__)
    name = {"LazyAdamOptimizer": "LazyAdam", "AdamOptimizer": "Adam",
            "MomentumOptimizer": "Momentum", "GradientDescentOptimizer": "SGD"}.get(name, name)
    oid = OPTIMIZER_CLS_NAMES.get(name)
  if oid is None:
    raise NotImplementedError("optimizer %r has no HIP implementation yet" % (optimizer,))
  return oid


def build_opt_config(optimizer, optimizer_params, learning_rate_decay_fn,
                     lr_policy_params, larc_params=None, loss_scaling=1.0,
                     loss_scaling_params=None, clip_gradients=None, dtype="mixed",
                     world_size=1):
  """Returns (capi.OptConfig, initial_loss_scale)."""
  if clip_gradients is not None and larc_params is not None:
    raise AttributeError("LARC and gradient norm clipping should not be used together")
  cfg = capi.OptConfig()
  oid = _optimizer_id(optimizer)
  op = dict(optimizer_params or {})
  cfg.optimizer = oid
  if oid == 2:
    d = dict(NovoGrad.DEFAULTS, **op)
    cfg.beta1, cfg.beta2, cfg.epsilon = d["beta1"], d["beta2"], d["epsilon"]
    cfg.weight_decay, cfg.grad_averaging = d["weight_decay"], int(bool(d["grad_averaging"]))
    cfg.novograd_ema = int(bool(d["ema_second_moment"]))
  elif oid == 1:
    cfg.beta1 = op.get("momentum", 0.9)
  elif oid == 3:
    cfg.beta1, cfg.beta2 = op.get("beta1", 0.9), op.get("beta2", 0.999)
    cfg.epsilon = op.get("epsilon", 1e-8)
  for k, v in lr_policies.device_policy(learning_rate_decay_fn, lr_policy_params).items():
    if isinstance(v, list):            # fixed-size arrays of the config struct (piecewise_constant)
      arr = getattr(cfg, k)
      for i, e in enumerate(v):
        arr[i] = e
    else:
      setattr(cfg, k, v)
  if larc_params is not None:
    check_params(larc_params, {'larc_eta': float},
                 {'larc_mode': ['clip', 'scale'], 'min_update': float, 'epsilon': float})
    cfg.use_larc = 1
    cfg.larc_eta = larc_params['larc_eta']
    cfg.larc_mode_scale = int(larc_params.get('larc_mode', 'clip') == 'scale')
    cfg.larc_min_update = larc_params.get('min_update', 1e-7)
    cfg.larc_epsilon = larc_params.get('epsilon', 1e-7)
  cfg.clip_global_norm = float(clip_gradients) if clip_gradients is not None else 0.0
  initial_scale = 1.0
  cfg.scaler = 0
  cfg.scale_min, cfg.scale_max, cfg.step_factor, cfg.step_window = 1.0, 2. ** 14, 2.0, 2000
  cfg.log_max, cfg.lm_beta1, cfg.lm_beta2, cfg.overflow_std_dev = 16., 0.99, 0.999, 3.09
  if dtype == "mixed":
    if isinstance(loss_scaling, six.string_types):
      als = AutomaticLossScaler(algorithm=loss_scaling, params=loss_scaling_params)
      cfg.scaler = als.scaler_id
      for k, v in als.cfg.items():
        setattr(cfg, k, v)
      initial_scale = als.initial_scale
    else:
      initial_scale = float(loss_scaling)
  cfg.world_size = int(world_size)
  return cfg, float(initial_scale)


class TrainOp(object):
  """The object optimize_loss returns: `run()` performs one optimisation step on
  the flat parameter store (after backward filled store.grads)."""

  def __init__(self, store, cfg, initial_scale):
    self.store, self.cfg = store, cfg
    self.state = torch.zeros(capi.opt_state_bytes(), dtype=torch.uint8, device=store.device)
    capi.opt_init_state(self.state, initial_scale)
    # float view of the loss scale (offset 32 in OptDeviceState) for device-side consumers
    self.loss_scale_view = self.state[32:36].view(torch.float32)
    self.lr_view = self.state[36:40].view(torch.float32)

  def run(self):
    s = self.store
    capi.opt_step(self.cfg, self.state, s.grads, s.master, s.m1, s.m2, s.w16,
                  s.chunk_tensor, s.tensor_chunk_begin, s.tensor_l2 if s.l2_active else None, None, s.partial,
                  s.t_gnorm2, s.t_wnorm2, s.t_amax, s.t_mult, s.t_v)
    s.refresh_dgrad_copies()

  def run_async(self, nranges=8):
    """The same step on the store's optimizer stream, NEXT TO whatever the calling stream does afterwards — the
    following forward pass: the update is bandwidth-bound, a forward pass of 132-tile GEMMs or 150-unit convolution
    launches leaves CUs and most of the HBM bandwidth idle. Order on the optimizer stream: wait for the caller's
    stream (the gradients are final there: backward, the weight-gradient stream and the all-reduce have been
    joined) -> os2s_opt_prepare -> `nranges` update ranges in flat-buffer order (= the order in which the forward
    pass needs the variables), an event behind each -> the transposed copies the data-gradient convolutions read.
    A variable's `w16` / `master` / `grad` attribute makes its reader wait for the range that holds it
    (FlatParams._ready). Each range zeroes its gradient chunks once they are read (also on a skipped step):
    `store.grads_zeroed` tells the next step that the fill of the gradient buffer is not needed."""
    s = self.store
    s.wait_all()                      # (a previous update still in flight: its ranges come first anyway)
    main = torch.cuda.current_stream()
    side = s.opt_stream()
    side.wait_stream(main)
    nchunks = s.chunk_tensor.numel()
    nranges = max(1, min(int(nranges), nchunks))
    bounds = [(nchunks * i) // nranges for i in range(nranges + 1)]
    pending = []
    with torch.cuda.stream(side):
      capi.opt_prepare(self.cfg, self.state, s._grads, s._master, s.chunk_tensor, s.tensor_chunk_begin,
                       s.tensor_l2 if s.l2_active else None, s.partial, s.t_gnorm2, s.t_wnorm2, s.t_amax,
                       s.t_mult, s._t_v)
      for c0, c1 in zip(bounds[:-1], bounds[1:]):
        capi.opt_apply_range(self.cfg, self.state, s._grads, s._master, s._m1, s._m2, s._w16, c0, c1,
                             s.chunk_tensor, s.tensor_l2 if s.l2_active else None, s.t_mult, zero_grads=True)
        ev = torch.cuda.Event()
        ev.record(side)
        pending.append((c1 * s.chunk, ev))
      s.version = getattr(s, "version", 0) + 1
      s._refresh_dgrad_copies_raw()
      evw = torch.cuda.Event()
      evw.record(side)
    s._pending, s._wt_event, s._main_stream, s._done_upto = pending, evw, main, 0
    s.grads_zeroed = True

  def read_state(self):
    self.store.wait_all()             # (the state block is written by the update's finalize pass)
    return capi.opt_read_state(self.state)


def optimize_loss(store, optimizer, optimizer_params, learning_rate_decay_fn,
                  lr_policy_params=None, dtype="mixed", clip_gradients=None, summaries=None,
                  larc_params=None, loss_scaling=1.0, loss_scaling_params=None,
                  on_horovod=False, iter_size=1, world_size=1):
  """Same argument meaning as optimizers.py:107-160 (loss/var_list are implicit in
  `store`). Returns a TrainOp."""
  if summaries is not None:
    for summ in summaries:
      if summ not in OPTIMIZER_SUMMARIES:
        raise ValueError("Summaries should be one of [{}], you provided {}.".format(
            ", ".join(OPTIMIZER_SUMMARIES), summ))
  cfg, scale = build_opt_config(optimizer, optimizer_params, learning_rate_decay_fn,
                                lr_policy_params or {}, larc_params, loss_scaling,
                                loss_scaling_params, clip_gradients, dtype, world_size)
  if iter_size > 1 and not on_horovod:
    raise ValueError("iter_size is only supported in Horovod mode")
  # optimizers.py:208-255: every micro-step adds grad / iter_size to the accumulator and the
  # update all-reduce-averages the accumulator. The flat gradient buffer holds the plain SUM
  # over micro-steps and ranks, so one division by world_size * iter_size is the same algebra.
  cfg.world_size = int(world_size) * int(iter_size)
  if cfg.optimizer == 3 and store.m2 is None:
    raise ValueError("Adam needs FlatParams.finalize(need_m2=True)")
  return TrainOp(store, cfg, scale)

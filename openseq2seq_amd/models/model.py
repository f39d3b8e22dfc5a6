"""Model shell — open_seq2seq/models/model.py:112-557 re-hosted on PyTorch-ROCm
device memory + the HIP kernels.

TF1 builds a graph once (`compile`) and runs it with `sess.run`; here `compile`
creates the data layer, the plugin objects and their variables (FlatParams), the
train op (optimize_loss) and the data-parallel gradient reducer, and
`train_step(batch)` enqueues one forward/backward/optimizer step (the body of the
reference's hot loop, utils/funcs.py:172-205). One process per GPU; `hvd`-style
data parallelism is RCCL through torch.distributed (parallel replaced by
openseq2seq_amd/utils/distributed.py).
"""
from __future__ import absolute_import, division, print_function

import abc
import os
import copy
import time

import six
import torch

from .. import capi
from ..optimizers.flat_params import FlatParams
from ..optimizers.optimizers import optimize_loss
from ..parts.cnns.conv_blocks import Tape, set_side_stream_enabled
from ..utils import distributed as dist_utils
from ..utils.utils import check_params

# OS2S_ASYNC_OPT=1 (or 'os2s_async_optimizer': True in the configuration): the optimizer update on its own stream
# next to the following forward pass (TrainOp.run_async) — bit-identical weights, and measured NEUTRAL on MI355X
# (Transformer-big 17.2 - 17.5 ms both ways, Jasper 38.4 both ways): the kernel trace shows the update ranges
# running under the forward GEMMs, each side 1.3 - 5 x slower while they share the memory system
# (profiles/r06_async_optimizer_overlap.log). Off by default: one stream, one fill of the gradient buffer.
ASYNC_OPTIMIZER = os.environ.get("OS2S_ASYNC_OPT", "0") == "1"
# A/B knob: 0 = Tape.backward releases the step's closures before the optimizer is enqueued (rounds 1 - 5)
DEFER_TAPE_FREE = os.environ.get("OS2S_DEFER_TAPE_FREE", "1") != "0"


def resolve_lr_params(lr_policy, lr_policy_params, last_step, steps_in_epoch, has_num_epochs):
  """The defaults Model.compile adds to `lr_policy_params` before the policy is built (models/model.py:479-495),
  by the policy function's signature: `decay_steps` = the last step of the run when the policy has the parameter
  and the config does not set it — then also `begin_decay_at` = max(begin_decay_at, warmup_steps) and
  `decay_steps` reduced by it (the decay starts no earlier than the warm-up ends and runs over what is left);
  `steps_per_epoch` = the epoch length when the run is configured in epochs (piecewise_constant boundaries are
  given in epochs then). `last_step` / `steps_in_epoch`: callables, evaluated only when needed. Returns a new
  dict."""
  import inspect
  lr_params = dict(lr_policy_params)
  try:
    func_params = inspect.signature(lr_policy).parameters
  except (TypeError, ValueError):
    func_params = {}
  if 'decay_steps' in func_params and 'decay_steps' not in lr_params:
    lr_params['decay_steps'] = last_step()
    if 'begin_decay_at' in func_params:
      if 'warmup_steps' in func_params:
        lr_params['begin_decay_at'] = max(lr_params.get('begin_decay_at', 0), lr_params.get('warmup_steps', 0))
      lr_params['decay_steps'] -= lr_params.get('begin_decay_at', 0)
  if 'steps_per_epoch' in func_params and 'steps_per_epoch' not in lr_params and has_num_epochs:
    spe = steps_in_epoch()
    if spe is not None:
      lr_params['steps_per_epoch'] = spe
  return lr_params


@six.add_metaclass(abc.ABCMeta)
class Model(object):
  @staticmethod
  def get_required_params():
    return {
        'use_horovod': bool,
        'batch_size_per_gpu': int,
        'data_layer': None,
    }

  @staticmethod
  def get_optional_params():
    return {
        'logdir': str, 'num_gpus': int, 'gpu_ids': list, 'load_model': str,
        'save_summaries_steps': None, 'print_loss_steps': None,
        'print_samples_steps': None, 'print_bench_info_steps': None,
        'save_checkpoint_steps': None, 'num_checkpoints': int,
        'restore_best_checkpoint': bool, 'eval_steps': int, 'finetune': bool,
        'eval_batch_size_per_gpu': int, 'hooks': list, 'random_seed': int,
        'num_epochs': int, 'max_steps': int, 'bench_start': int,
        'data_layer_params': dict, 'optimizer': None, 'optimizer_params': dict,
        'freeze_variables_regex': None, 'initializer': None, 'initializer_params': dict,
        'regularizer': None, 'regularizer_params': dict,
        'dtype': None,   # tf.float16 / tf.float32 tokens or 'mixed' (bf16 + fp32 masters here)
        'lr_policy': None, 'lr_policy_params': dict, 'max_grad_norm': float,
        'larc_params': dict, 'loss_scaling': None, 'loss_scaling_params': dict,
        'summaries': list, 'iter_size': int, 'lm_vocab_file': str,
        'processed_data_folder': str,
        'use_trt': bool, 'trt_precision_mode': str, 'trt_max_workspace_size_bytes': int,
        'trt_minimum_segment_size': int, 'trt_is_dynamic_op': bool,
        'trt_maximum_cached_engines': int, 'use_xla_jit': bool,
        # this engine only: parameter-gradient kernels on a second HIP stream (default True; the
        # Tacotron2 configuration turns it off — its many small layers lose more in stream hand-offs
        # than the overlap returns: 131.9 vs 128.9 ms/step)
        'os2s_side_stream': bool,
        'os2s_async_optimizer': bool,
    }

  def __init__(self, params, mode="train", hvd=None, device=None):
    """params/mode as in model.py:112-376. `hvd` is any object with rank()/size()
    (a torch.distributed adapter: utils/distributed.py) or None."""
    check_params(params, self.get_required_params(), self.get_optional_params())
    self._params = copy.deepcopy(params)
    if self._params.get('iter_size', 1) > 1 and not self._params['use_horovod']:
      raise ValueError("iter_size is only supported in Horovod mode")
    if mode not in ['train', 'infer', 'eval', 'interactive_infer']:
      raise ValueError("Mode has to be one of ['train', 'infer', 'eval', 'interactive_infer']")
    if 'max_steps' in params and 'num_epochs' in params and mode == "train":
      raise ValueError("You can't provide both max_steps and num_epochs. "
                       "Please, remove one of them from the config.")
    self._mode = mode
    self._hvd = hvd
    p = self._params
    # tower mode (models/model.py:293-303, 386-427): `use_horovod False, num_gpus N` means N
    # replicas and a global batch of N * batch_size_per_gpu. Replicas are ranks here, so the
    # process group has to have exactly N of them (run.py / bench.py launch them); training a
    # tower config on fewer GPUs than it names is refused, never silent.
    self._towers = 1
    if not p['use_horovod']:
      self._towers = dist_utils.configured_towers(p)
      world = hvd.size() if hvd is not None else 1
      if self._towers != world and mode in ("train", "eval"):
        raise ValueError(
            "config asks for %d replicas (use_horovod False, num_gpus/gpu_ids) but this process "
            "group has %d rank(s): launch through run.py (it starts the ranks itself) or "
            "`python -m torch.distributed.run --nproc-per-node %d`" % (self._towers, world,
                                                                      self._towers))
    p.setdefault('dtype', 'mixed')
    p.setdefault('iter_size', 1)
    p.setdefault('loss_scaling', 1.0)
    self._device = device or torch.device("cuda", torch.cuda.current_device())
    # seeds: random_seed + rank (model.py:309-313)
    rs = p.get('random_seed', int(time.time()))
    rank = hvd.rank() if hvd is not None else 0
    self._seed = rs + rank
    torch.manual_seed(rs)     # identical initial weights on every rank
    self._step_count = 0
    self._store = None
    self._train_op = None
    self._compiled = False
    self._data_layer = None

  # ---- accessors mirroring the reference ---------------------------------
  @property
  def params(self):
    return self._params

  @property
  def mode(self):
    return self._mode

  @property
  def hvd(self):
    return self._hvd

  @property
  def on_horovod(self):
    return self._hvd is not None

  def get_data_layer(self, worker_id=0):
    return self._data_layer

  @property
  def num_gpus(self):
    """Replicas the job trains on (model.py:293-303): the tower count of the config without
    Horovod — each one is a rank of this process group — and 1 per rank under Horovod."""
    return self._towers

  @property
  def store(self):
    return self._store

  @property
  def train_op(self):
    return self._train_op

  # ---- compile ---------------------------------------------------------------
  def compile(self, force_var_reuse=False, checkpoint=None):
    """model.py:378-557: build the forward pass objects + variables + train op."""
    self._store = FlatParams(self._device)
    self._build_forward_pass_objects(self._store)
    need_m2 = False
    if self._mode == "train":
      from ..optimizers.optimizers import _optimizer_id
      need_m2 = _optimizer_id(self._params['optimizer']) == 3
    self._store.finalize(need_m2=need_m2)
    world = self._hvd.size() if self._hvd is not None else 1
    if self._hvd is not None and world > 1:
      # BroadcastGlobalVariablesHook(0) (utils/hooks.py:15-55)
      dist_utils.broadcast_parameters(self._store, self._extra_state_tensors())
    if self._mode == "train":
      p = self._params
      if 'lr_policy' not in p:
        raise ValueError("lr_policy has to be specified for train mode")
      lr_params = resolve_lr_params(p['lr_policy'], p.get('lr_policy_params', {}), self._last_step,
                                    lambda: self.steps_in_epoch, 'num_epochs' in p)
      self._train_op = optimize_loss(
          self._store, p['optimizer'], p.get('optimizer_params', {}), p['lr_policy'],
          lr_params, dtype=p['dtype'], clip_gradients=p.get('max_grad_norm', None),
          summaries=p.get('summaries', None), larc_params=p.get('larc_params', None),
          loss_scaling=p.get('loss_scaling', 1.0),
          loss_scaling_params=p.get('loss_scaling_params', None),
          on_horovod=self.on_horovod, iter_size=p.get('iter_size', 1), world_size=world)
      # OS2S_FORCE_REDUCER: exercise the RCCL side-stream path with a one-rank group (tools/)
      force = self._hvd is not None and os.environ.get("OS2S_FORCE_REDUCER", "") == "1"
      self._reducer = dist_utils.GradientReducer(self._store, world) if (world > 1 or force) else None
    self._compiled = True
    return self

  def _last_step(self, default=100000):
    """models/model.py:346-365: `max_steps`, else num_epochs * steps_in_epoch with
    steps_in_epoch = dataset size // (batch_size_per_gpu * workers * iter_size). `default` when
    the data layer cannot tell its size (synthetic batches: the dataset files are absent)."""
    p = self._params
    if 'max_steps' in p:
      return p['max_steps']
    if 'num_epochs' in p and self._data_layer is not None:
      try:
        n = self._data_layer.get_size_in_samples()
        world = self._hvd.size() if self._hvd is not None else 1
        spe = n // (p['batch_size_per_gpu'] * world * p.get('iter_size', 1))
        return max(spe * p['num_epochs'], 1)
      except Exception:
        pass
    return default

  @property
  def steps_in_epoch(self):
    """models/model.py:340-344: dataset size // (batch_size_per_gpu * workers * iter_size); None when the data
    layer cannot tell its size (synthetic batches)."""
    p = self._params
    if self._data_layer is None:
      return None
    try:
      n = self._data_layer.get_size_in_samples()
    except Exception:
      return None
    world = self._hvd.size() if self._hvd is not None else 1
    return n // (p['batch_size_per_gpu'] * world * p.get('iter_size', 1))

  @property
  def last_step(self):
    """The step the training loop stops at and the final checkpoint is labelled with (the same
    number the lr schedule's decay_steps default to)."""
    return self._last_step()

  def _extra_state_tensors(self):
    """Non-trainable state that travels with the variables (rank-0 broadcast, train -> eval copy): every
    tensor registered with the store under its reference name (BatchNorm moving statistics of ALL layers —
    the row convolution of DeepSpeech2 included), in registration order (same build code => same order)."""
    return list(self._store.state.values()) if self._store is not None else []

  @abc.abstractmethod
  def _build_forward_pass_objects(self, store):
    """Create data layer + encoder/decoder/loss objects and their variables."""

  @abc.abstractmethod
  def _forward_backward(self, batch, tape):
    """Run forward (and record backward) for one batch; returns the loss tensor [1]."""

  @abc.abstractmethod
  def _get_num_objects_per_step(self, batch):
    pass

  # ---- one training step (funcs.py:184: sess.run([train_op, ...])) ----------
  def train_step(self, batch):
    assert self._compiled and self._mode == "train"
    p = self._params
    # the side-stream switch is this model's, for the duration of its step only (a second model in the
    # process — an eval twin, a test's bare Tape — keeps the setting it had)
    prev = set_side_stream_enabled(p.get('os2s_side_stream', True))
    try:
      iter_size = p.get('iter_size', 1)
      micro = self._step_count % iter_size
      if micro == 0:
        if self._store.grads_zeroed:
          self._store.grads_zeroed = False      # the asynchronous update zeroed each chunk behind its last read
        else:
          self._store.zero_grads()
      last_micro = (micro == iter_size - 1)
      overlap = self._reducer is not None and last_micro
      # what a persistent-GRU abort must be able to roll back (see _recover_gru_abort): the non-trainable state
      # always (a few KB: BatchNorm statistics a garbage forward pass would poison), the gradient buffer only
      # inside an accumulation window (iter_size > 1: earlier micro-steps already sit in it)
      snap = None
      if self._gru_guard:
        snap = ([t.clone() for t in self._extra_state_tensors()],
                self._store.grads.clone() if micro > 0 else None)
      launches0 = capi.gru_xcd_launch_count()
      tape = None
      if getattr(self, "_halves_enabled", None) is not None and self._halves_enabled():
        loss = self._forward_backward_halves(batch)          # experiment: two half-batches on two streams
      else:
        tape = Tape(on_done=self._reducer.mark_done if overlap else None)
        tape.defer_free = DEFER_TAPE_FREE
        loss = self._forward_backward(batch, tape)
        tape.backward()
      ran_persistent = capi.gru_xcd_launch_count() != launches0
      # Whether a rank ran persistent launches depends on ITS batch shape and environment (B <= 32, T >= 2,
      # OS2S_GRU_XCD), so under data parallelism the decision to enter the status all-reduce must not: every rank
      # of a model with recurrent layers enters it until the persistent path is known to be off everywhere.
      world = self._hvd.size() if self._hvd is not None else 1
      if self._gru_agree is None:          # from the CONFIGURATION (identical on every rank), not from what ran here
        self._gru_agree = world > 1 and self._has_gru_layers()
      if ran_persistent or (world > 1 and self._gru_agree):
        self._gru_guard = ran_persistent   # snapshots from the next step on — only while the persistent path runs
        loss = self._recover_gru_abort(batch, loss, snap, micro, overlap, ran_persistent)
      self._step_count += 1
      if last_micro:
        if self._reducer is not None:
          self._reducer.finish()
        # the update runs on its own stream next to the NEXT step's forward pass (TrainOp.run_async) unless
        # the configuration or OS2S_ASYNC_OPT=0 asks for the one-stream form
        if p.get('os2s_async_optimizer', ASYNC_OPTIMIZER):
          self._train_op.run_async()
        else:
          self._train_op.run()
      if tape is not None:
        tape.ops = []            # (Tape.defer_free: the step's closures are released behind the optimizer launch)
    finally:
      set_side_stream_enabled(prev)
    return loss

  def _has_gru_layers(self):
    """Does the configuration build cuDNN-form GRU layers (the only ones the persistent kernels take)?"""
    for part in (getattr(self, "_encoder", None), getattr(self, "_decoder", None)):
      params = getattr(part, "params", None) or {}
      if "gru" in str(params.get("rnn_type", "")).lower():
        return True
    return False

  _gru_guard = False
  _gru_agree = None      # world > 1: does this model take part in the per-step agreement (set on the first step)

  def _recover_gru_abort(self, batch, loss, snap, micro, overlap, ran_persistent=True):
    """The persistent GRU kernels (csrc/rnn_xcd.hip) need 32 co-resident workgroups per XCD and give up after a
    bounded wait when they do not get them (CUs held by RCCL's resident kernels, a second process on the
    device, a partitioned GPU): the launch sets a sticky word and its outputs are garbage. Instead of raising
    (round 4), the step is REDONE on the launch-per-step kernels: one synchronisation per step of a model that
    ran persistent launches, the answer agreed over the data-parallel ranks (every rank redoes — and re-reduces —
    or none does), state and gradient buffer rolled back, the persistent path switched off for the rest of the
    process. The first step of a model has no snapshot yet: its BatchNorm statistics are re-initialised by the
    redone forward pass only in so far as the moving average forgets (documented; the abort of a FIRST step was
    never observed — the placement check fails at launch, before any state is written)."""
    code = 0
    if ran_persistent:
      torch.cuda.current_stream().synchronize()
      code = capi.gru_xcd_status(clear=True)
    world = self._hvd.size() if self._hvd is not None else 1
    if world > 1:
      # [abort code, "I still run persistent launches"]: MAX over the ranks. Once no rank runs them the
      # agreement stops on every rank at the same step (the same collective sequence everywhere).
      t = torch.tensor([code, int(ran_persistent)], dtype=torch.int32, device=self._device)
      torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
      code, anyone = int(t[0].item()), int(t[1].item())
      self._gru_agree = bool(anyone) and code == 0
    if code == 0:
      return loss
    self._gru_guard = False             # the persistent path is off from here on: no more snapshots
    import warnings
    warnings.warn("a persistent GRU launch gave up (code %d: 1 = poll timeout, 2 = workgroup placement); the step "
                  "is redone on the launch-per-step kernels, which stay selected for this process" % code)
    capi.gru_xcd_set_mode(0)
    if self._reducer is not None:
      if overlap:
        self._reducer.finish()          # drain the buckets of the aborted pass (same collectives on every rank)
      self._reducer.reset()
    if snap is not None:
      for t, s0 in zip(self._extra_state_tensors(), snap[0]):
        t.copy_(s0)
    if micro == 0:
      self._store.zero_grads()
    elif snap is not None and snap[1] is not None:
      self._store.grads.copy_(snap[1])
    else:
      raise RuntimeError("persistent GRU abort inside an accumulation window before a snapshot existed")
    tape = Tape(on_done=self._reducer.mark_done if overlap else None)
    loss = self._forward_backward(batch, tape)
    tape.backward()
    torch.cuda.current_stream().synchronize()
    assert capi.gru_xcd_status(clear=True) == 0
    return loss

  def copy_weights_from(self, other):
    """Variable sharing between the train and the eval model (the reference builds the
    eval graph with reuse=True, utils.py:838-846): same build code => same flat layout."""
    assert self._store.total == other._store.total
    self._store.master.copy_(other._store.master)
    self._store.refresh_compute_copies()
    for mine, theirs in zip(self._extra_state_tensors(), other._extra_state_tensors()):
      mine.copy_(theirs)

  def global_step(self):
    return self._train_op.read_state()["global_step"]

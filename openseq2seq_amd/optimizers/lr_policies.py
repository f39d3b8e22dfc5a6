"""Learning-rate policies — same names and arguments as
open_seq2seq/optimizers/lr_policies.py:16-245. Each is a plain Python function
of the integer global step (host scalar math); `device_policy` maps a policy +
its params onto the enum the optimizer kernel evaluates at the device-resident
global step (so a skipped step needs no host round trip)."""
import math


def fixed_lr(global_step, learning_rate):
  return learning_rate


def piecewise_constant(global_step, learning_rate, boundaries, decay_rates,
                       steps_per_epoch=None):
  if steps_per_epoch is not None:
    boundaries = [steps_per_epoch * e for e in boundaries]
  vals = [learning_rate * d for d in [1.0] + list(decay_rates)]
  i = 0
  while i < len(boundaries) and global_step > boundaries[i]:
    i += 1
  return vals[i]


def exp_decay(global_step, learning_rate, decay_steps, decay_rate, use_staircase_decay,
              begin_decay_at=0, min_lr=0.0):
  lr = learning_rate
  if global_step >= begin_decay_at:
    p = (global_step - begin_decay_at) / float(decay_steps)
    if use_staircase_decay:
      p = math.floor(p)
    lr = learning_rate * decay_rate ** p
  return max(min_lr, lr)


def poly_decay(global_step, learning_rate, decay_steps, power=1.0, begin_decay_at=0,
               min_lr=0.0, warmup_steps=0):
  lr = learning_rate
  if warmup_steps > 0 and global_step < warmup_steps:
    lr = learning_rate * float(global_step) / float(warmup_steps)
  if global_step < begin_decay_at:
    return lr
  s = min(global_step - begin_decay_at, decay_steps)
  return (lr - min_lr) * (1.0 - s / float(decay_steps)) ** power + min_lr


def cosine_decay(global_step, learning_rate, decay_steps, power=1.0, begin_decay_at=0,
                 min_lr=0.0, warmup_steps=0):
  lr = learning_rate
  if warmup_steps > 0 and global_step < warmup_steps:
    lr = learning_rate * float(global_step) / float(warmup_steps)
  if global_step < begin_decay_at:
    return lr
  s = min(global_step - begin_decay_at, decay_steps)
  return lr * ((1 - min_lr) * 0.5 * (1 + math.cos(math.pi * s / float(decay_steps))) + min_lr)


def transformer_policy(global_step, learning_rate, d_model, warmup_steps, max_lr=None,
                       coefficient=1.0, dtype=None):
  step, ws = float(global_step), float(warmup_steps)
  decay = coefficient * d_model ** -0.5 * min((step + 1) * ws ** -1.5, (step + 1) ** -0.5)
  new_lr = decay * learning_rate
  return min(max_lr, new_lr) if max_lr is not None else new_lr


def inv_poly_decay(global_step, learning_rate, decay_steps, min_lr, power=1.0,
                   begin_decay_at=0, warmup_steps=0, name="learning_rate"):
  min_lr = min(max(min_lr, 1e-8), learning_rate)
  if power <= 0.:
    raise ValueError("Inv poly decay requires power >  0.")
  scale = (math.pow(learning_rate / min_lr, 1. / power) - 1.) / decay_steps
  return learning_rate / math.pow(1. + scale * global_step, power)


_DEVICE_IDS = {"fixed_lr": 0, "poly_decay": 1, "exp_decay": 2, "transformer_policy": 3,
               "cosine_decay": 4, "piecewise_constant": 5, "inv_poly_decay": 6}
MAX_BOUNDARIES = 16      # OS2S_LR_MAX_BOUNDARIES


def device_policy(policy_fn, params):
  """-> dict of os2s_opt_config_t lr fields for a policy the kernel implements."""
  name = getattr(policy_fn, "__name__", str(policy_fn))
  if name not in _DEVICE_IDS:
    raise NotImplementedError("lr policy %s has no device implementation" % name)
  p = dict(params)
  out = dict(lr_policy=_DEVICE_IDS[name], learning_rate=float(p.get("learning_rate", 0.0)),
             min_lr=float(p.get("min_lr", 0.0)), power=float(p.get("power", 1.0)),
             decay_rate=float(p.get("decay_rate", 1.0)),
             decay_steps=int(p.get("decay_steps", 1)),
             begin_decay_at=int(p.get("begin_decay_at", 0)),
             warmup_steps=int(p.get("warmup_steps", 0)),
             use_staircase_decay=int(bool(p.get("use_staircase_decay", False))),
             d_model=int(p.get("d_model", 1)), coefficient=float(p.get("coefficient", 1.0)),
             has_max_lr=int(p.get("max_lr") is not None),
             max_lr=float(p.get("max_lr") or 0.0))
  if name == "piecewise_constant":
    bounds = list(p["boundaries"])
    if p.get("steps_per_epoch") is not None:
      bounds = [p["steps_per_epoch"] * e for e in bounds]
    rates = [1.0] + list(p["decay_rates"])
    if len(bounds) > MAX_BOUNDARIES or len(rates) != len(bounds) + 1:
      raise ValueError("piecewise_constant: at most %d boundaries, one decay rate per boundary" % MAX_BOUNDARIES)
    out["pw_count"] = len(bounds)
    out["pw_boundaries"] = [int(b) for b in bounds]
    out["pw_rates"] = [float(r) for r in rates]
  if name == "inv_poly_decay" and float(p.get("power", 1.0)) <= 0.0:
    raise ValueError("Inv poly decay requires power >  0.")
  return out

"""ORACLE (test infrastructure only): NumPy restatement of the reference's
optimizer path (open_seq2seq/optimizers).

  lr policies        lr_policies.py:16-245 (+ tf.train.polynomial_decay /
                     exponential_decay / cosine_decay semantics, TF 1.13)
  MP wrapper         mp_wrapper.py:44-122 (loss*scale -> fp32 grads -> +scale*d(reg)
                     -> *1/scale; NaN/Inf skip; cast back)
  post-processing    optimizers.py:289-482 (global-norm clip, LARC)
  loss scalers       automatic_loss_scaler.py:50-203 (Backoff, LogMax)
  NovoGrad           novograd.py:93-126 (+ TF MomentumOptimizer: m = b1*m + g; w -= lr*m)
  Adam               tf.train.AdamOptimizer (lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
                     w -= lr_t*m/(sqrt(v)+eps))

PARITY STATUS: the reference pins only two relations here
(mp_wrapper_test.py:93-95 regulariser gradient 1e-8 under the MP wrapper;
optimizers_test.py:56-80 iter_size algebra); both are reproduced in
tests/test_oracle_optim.py. Round 5: the lr policies, both loss scalers and RefOptimizer
(mixed-precision wrapper + LARC / clipping + NovoGrad as written / Adam / Momentum) are pinned
to the reference's OWN CODE, executed from its files on the TF-primitive stand-in
oracle/ref_shim/tf1: tests/test_ref_exec_optim.py (policies 2e-6, Backoff trace exact),
tests/test_ref_exec_train_op.py (28 steps with fp16 overflow: loss scale, skipped steps,
master copies step by step).
"""
import math

import numpy as np


# ---------------------------------------------------------------------------
# lr policies (lr_policies.py)
# ---------------------------------------------------------------------------
def fixed_lr(global_step, learning_rate):
  return learning_rate


def poly_decay(global_step, learning_rate, decay_steps, power=1.0, begin_decay_at=0,
               min_lr=0.0, warmup_steps=0):
  lr = learning_rate
  if warmup_steps > 0 and global_step < warmup_steps:
    lr = learning_rate * float(global_step) / float(warmup_steps)
  if global_step < begin_decay_at:
    return lr
  s = min(global_step - begin_decay_at, decay_steps)
  return (lr - min_lr) * (1.0 - s / float(decay_steps)) ** power + min_lr


def exp_decay(global_step, learning_rate, decay_steps, decay_rate, use_staircase_decay,
              begin_decay_at=0, min_lr=0.0):
  lr = learning_rate
  if global_step >= begin_decay_at:
    p = (global_step - begin_decay_at) / float(decay_steps)
    if use_staircase_decay:
      p = math.floor(p)
    lr = learning_rate * decay_rate ** p
  return max(min_lr, lr)


def cosine_decay(global_step, learning_rate, decay_steps, power=1.0, begin_decay_at=0,
                 min_lr=0.0, warmup_steps=0):
  lr = learning_rate
  if warmup_steps > 0 and global_step < warmup_steps:
    lr = learning_rate * float(global_step) / float(warmup_steps)
  if global_step < begin_decay_at:
    return lr
  s = min(global_step - begin_decay_at, decay_steps)
  cosd = 0.5 * (1 + math.cos(math.pi * s / float(decay_steps)))
  return lr * ((1 - min_lr) * cosd + min_lr)


def transformer_policy(global_step, learning_rate, d_model, warmup_steps, max_lr=None,
                       coefficient=1.0):
  step = float(global_step)
  ws = float(warmup_steps)
  decay = coefficient * d_model ** -0.5 * min((step + 1) * ws ** -1.5, (step + 1) ** -0.5)
  new_lr = decay * learning_rate
  return min(max_lr, new_lr) if max_lr is not None else new_lr


def piecewise_constant(global_step, learning_rate, boundaries, decay_rates, steps_per_epoch=None):
  """lr_policies.py:30-57 over tf.train.piecewise_constant(x, boundaries, values): values[0] for
  x <= boundaries[0], values[i] for boundaries[i-1] < x <= boundaries[i], values[-1] beyond."""
  if steps_per_epoch is not None:
    boundaries = [steps_per_epoch * e for e in boundaries]
  vals = [learning_rate * d for d in [1.0] + list(decay_rates)]
  for i, b in enumerate(boundaries):
    if global_step <= b:
      return vals[i]
  return vals[-1]


def inv_poly_decay(global_step, learning_rate, decay_steps, min_lr, power=1.0, begin_decay_at=0,
                   warmup_steps=0):
  """lr_policies.py:203-245: lr / (1 + scale * t)^power with scale chosen so that lr(decay_steps) =
  min_lr (clamped to [1e-8, learning_rate]); begin_decay_at / warmup_steps are accepted and unused,
  as in the reference."""
  min_lr = min(max(min_lr, 1e-8), learning_rate)
  scale = (math.pow(learning_rate / min_lr, 1.0 / power) - 1.0) / decay_steps
  return learning_rate / math.pow(1.0 + scale * global_step, power)


# ---------------------------------------------------------------------------
# loss scalers (automatic_loss_scaler.py)
# ---------------------------------------------------------------------------
class BackoffScaler(object):
  def __init__(self, scale_min=1.0, scale_max=2.0 ** 14, step_factor=2.0, step_window=2000):
    self.scale_min, self.scale_max = scale_min, scale_max
    self.step_factor, self.step_window = step_factor, step_window
    self.iteration = 0
    self.last_overflow_iteration = -1
    self.scale = np.float32(scale_max)

  def update(self, has_nan, amax):
    overflow = bool(has_nan) or bool(np.isinf(amax))
    if overflow:
      self.scale = np.float32(np.clip(self.scale / self.step_factor, self.scale_min,
                                      self.scale_max))
      self.last_overflow_iteration = self.iteration
    else:
      since = self.iteration - self.last_overflow_iteration
      if since % self.step_window == 0:
        self.scale = np.float32(np.clip(self.scale * self.step_factor, self.scale_min,
                                        self.scale_max))
    self.iteration += 1
    return overflow


class LogMaxScaler(object):
  def __init__(self, scale_min=1.0, scale_max=2.0 ** 14, log_max=16., beta1=0.99,
               beta2=0.999, overflow_std_dev=3.09):
    self.scale_min, self.scale_max, self.log_max = scale_min, scale_max, log_max
    self.beta1, self.beta2, self.osd = beta1, beta2, overflow_std_dev
    self.iteration = 0
    self.scale = np.float32(1.0)
    self.x_hat = self.slow_x_hat = self.xsquared_hat = np.float32(0)
    self.b1c = self.b2c = np.float32(1)

  def update(self, has_nan, amax):
    f = np.float32
    nonfinite = bool(has_nan) or bool(np.isinf(amax))
    x = f(2.0 ** self.log_max) if nonfinite else f(np.log(f(amax)) / np.log(f(2.)))
    self.x_hat = f(self.beta1 * self.x_hat + (1 - self.beta1) * x)
    self.b1c = f(self.b1c * self.beta1)
    mu = self.x_hat / (1 - self.b1c)
    self.slow_x_hat = f(self.beta2 * self.slow_x_hat + (1 - self.beta2) * x)
    self.xsquared_hat = f(self.beta2 * self.xsquared_hat + (1 - self.beta2) * (x * x))
    self.b2c = f(self.b2c * self.beta2)
    e_x2 = self.xsquared_hat / (1 - self.b2c)
    slow_mu = self.slow_x_hat / (1 - self.b2c)
    sigma = np.sqrt(max(e_x2 - slow_mu * slow_mu, 0.))
    log_cutoff = sigma * self.osd + mu
    self.scale = f(np.clip(2.0 ** (16 - log_cutoff), self.scale_min, self.scale_max))
    self.iteration += 1
    return nonfinite


# ---------------------------------------------------------------------------
# one optimisation step on a list of fp32 master tensors
# ---------------------------------------------------------------------------
class RefOptimizer(object):
  """optimize_loss(...) apply path for dtype='mixed' (optimizers.py:194-286)."""

  def __init__(self, weights, optimizer="NovoGrad", opt_params=None, lr_fn=None,
               larc_params=None, clip_gradients=None, scaler=None, l2=None, world_size=1):
    self.w = [np.array(w, np.float32) for w in weights]
    self.opt = optimizer
    self.p = dict(opt_params or {})
    self.lr_fn = lr_fn or (lambda s: 0.01)
    self.larc = larc_params
    self.clip = clip_gradients
    self.scaler = scaler
    self.static_scale = np.float32(1.0)
    self.l2 = l2 or [0.0] * len(self.w)
    self.world = world_size
    self.global_step = 0
    self.m = [np.zeros_like(w) for w in self.w]
    self.v = [np.zeros_like(w) for w in self.w]
    self.ema = [np.float32(0)] * len(self.w)
    self.skipped = 0

  @property
  def loss_scale(self):
    return self.scaler.scale if self.scaler is not None else self.static_scale

  def step(self, scaled_grads_sum):
    """scaled_grads_sum: grads of (loss*loss_scale), summed over `world` ranks."""
    f = np.float32
    scale = f(self.loss_scale)
    lr = f(self.lr_fn(self.global_step))
    with np.errstate(all="ignore"):
      # mp_wrapper.py:79-95 (+ hvd.allreduce average, optimizers.py:96)
      g = [f(1) / (scale * f(self.world)) * np.asarray(x, np.float32) + f(l2) * w
           for x, w, l2 in zip(scaled_grads_sum, self.w, self.l2)]
      if self.clip is not None:   # optimizers.py:388-482
        gn = np.sqrt(sum(float(np.sum(np.square(x.astype(np.float64)))) for x in g))
        sc = f(self.clip * min(1.0 / gn, 1.0 / self.clip))
        g = [x * sc for x in g]
      if self.larc is not None:   # optimizers.py:333-377
        eta = self.larc["larc_eta"]
        mode = self.larc.get("larc_mode", "clip")
        mn = self.larc.get("min_update", 1e-7)
        eps = self.larc.get("epsilon", 1e-7)
        out = []
        for x, w in zip(g, self.w):
          vn = np.sqrt(np.sum(np.square(w.astype(np.float64))))
          gn = np.sqrt(np.sum(np.square(x.astype(np.float64))))
          if mode == "clip":
            u = min(max(eta * vn / (lr * (gn + eps)), mn), 1.0)
          else:
            u = max(eta * vn / (gn + eps), mn)
          out.append(x * f(u))
        g = out
      has_nan = any(bool(np.isnan(x).any()) for x in g)
      amax = max(float(np.max(np.abs(x))) if x.size else 0.0 for x in g)
    skip = False
    if self.scaler is not None:   # mp_wrapper.py:114-120
      skip = self.scaler.update(has_nan, amax)
    if skip:
      self.skipped += 1
      return True
    p = self.p
    if self.opt == "NovoGrad":    # novograd.py:100-126
      b1, b2 = p.get("beta1", 0.95), p.get("beta2", 0.98)
      eps, wd = p.get("epsilon", 1e-8), p.get("weight_decay", 0.0)
      # novograd.py:107-113 AS WRITTEN: `self._grads_ema[i] = tf.cond(tf.equal(var, 0.), g_2, ...)`
      # rebinds the Python list entry to the cond's output tensor; the tf variable nvgrad2_ema<i>
      # is never assigned, stays 0, and the cond takes the g_2 branch on every session.run:
      # v_t = |g_t|^2, beta2 is dead. `ema_second_moment` (not a reference parameter) selects the
      # moving average of the published algorithm instead.
      use_ema = bool(p.get("ema_second_moment", False))
      for i, (x, w) in enumerate(zip(g, self.w)):
        g2 = f(np.sum(np.square(x.astype(np.float64))))
        self.ema[i] = g2 if (self.ema[i] == 0 or not use_ema) else f(self.ema[i] * b2 + g2 * (1 - b2))
        x = x * f(1.0 / np.sqrt(self.ema[i] + eps))
        if wd > 0:
          x = x + f(wd) * w
        if p.get("grad_averaging", False):
          x = x * f(1 - b1)
        self.m[i] = f(b1) * self.m[i] + x
        self.w[i] = w - lr * self.m[i]
    elif self.opt == "Momentum":
      mom = p.get("momentum", 0.9)
      for i, (x, w) in enumerate(zip(g, self.w)):
        self.m[i] = f(mom) * self.m[i] + x
        self.w[i] = w - lr * self.m[i]
    elif self.opt == "Adam":
      b1, b2, eps = p.get("beta1", 0.9), p.get("beta2", 0.999), p.get("epsilon", 1e-8)
      t = self.global_step + 1
      lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
      for i, (x, w) in enumerate(zip(g, self.w)):
        self.m[i] = f(b1) * self.m[i] + f(1 - b1) * x
        self.v[i] = f(b2) * self.v[i] + f(1 - b2) * x * x
        self.w[i] = w - f(lr_t) * self.m[i] / (np.sqrt(self.v[i]) + f(eps))
    elif self.opt == "SGD":
      for i, (x, w) in enumerate(zip(g, self.w)):
        self.w[i] = w - lr * x
    else:
      raise ValueError(self.opt)
    self.global_step += 1
    return False

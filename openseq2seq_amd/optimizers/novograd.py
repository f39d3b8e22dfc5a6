"""NovoGrad marker class — config files name the optimizer by class
(`"optimizer": NovoGrad`, example_configs/speech2text/jasper10x5_LibriSpeech_nvgrad_masks.py:31).
The arithmetic (open_seq2seq/optimizers/novograd.py:93-126) runs in the
multi-tensor HIP kernels (csrc/optimizer.hip).

Second moment: the reference graph never assigns its `nvgrad2_ema*` variables (novograd.py:107-113,
the `tf.cond` result only replaces the Python list entry), so in the reference v_t = |g_t|^2 on
every step and `beta2` has no effect. That behaviour is the default here — the Jasper NovoGrad
hyper-parameters were tuned against it. `ema_second_moment=True` (an extension, not a reference
parameter) switches to the moving average v_t = beta2 v_{t-1} + (1 - beta2) |g_t|^2 of the
published algorithm."""


class NovoGrad(object):
  DEFAULTS = dict(beta1=0.95, beta2=0.98, epsilon=1e-8, weight_decay=0.0,
                  grad_averaging=False, ema_second_moment=False)

  def __init__(self, learning_rate=1.0, **kw):
    self.learning_rate = learning_rate
    self.params = dict(self.DEFAULTS, **kw)

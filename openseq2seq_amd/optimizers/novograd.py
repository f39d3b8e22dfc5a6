"""NovoGrad marker class — config files name the optimizer by class
(`"optimizer": NovoGrad`, example_configs/speech2text/jasper10x5_LibriSpeech_nvgrad_masks.py:31).
The arithmetic (open_seq2seq/optimizers/novograd.py:93-126) runs in the
multi-tensor HIP kernels (csrc/optimizer.hip)."""


class NovoGrad(object):
  DEFAULTS = dict(beta1=0.95, beta2=0.98, epsilon=1e-8, weight_decay=0.0,
                  grad_averaging=False)

  def __init__(self, learning_rate=1.0, **kw):
    self.learning_rate = learning_rate
    self.params = dict(self.DEFAULTS, **kw)

"""Loss-scaler parameter schemas (open_seq2seq/optimizers/automatic_loss_scaler.py:11-203).
State and update rules live on the device (csrc/optimizer.hip, mt_finalize_kernel)."""
from ..utils.utils import check_params


class AutomaticLossScaler(object):
  SUPPORTED_ALGOS = ['backoff', 'logmax']

  def __init__(self, algorithm='Backoff', params=None):
    algorithm = algorithm.lower().strip()
    params = params or {}
    if algorithm == 'backoff':
      check_params(params, {}, {'scale_min': float, 'scale_max': float,
                                'step_factor': float, 'step_window': int})
      self.scaler_id = 1
      self.cfg = dict(scale_min=params.get('scale_min', 1.0),
                      scale_max=params.get('scale_max', 2. ** 14),
                      step_factor=params.get('step_factor', 2.0),
                      step_window=params.get('step_window', 2000))
      self.initial_scale = self.cfg['scale_max']
    elif algorithm == 'logmax':
      check_params(params, {}, {'scale_min': float, 'scale_max': float, 'log_max': float,
                                'beta1': float, 'beta2': float, 'overflow_std_dev': float})
      self.scaler_id = 2
      self.cfg = dict(scale_min=params.get('scale_min', 1.0),
                      scale_max=params.get('scale_max', 2. ** 14),
                      log_max=params.get('log_max', 16.),
                      lm_beta1=params.get('beta1', 0.99), lm_beta2=params.get('beta2', 0.999),
                      overflow_std_dev=params.get('overflow_std_dev', 3.09))
      self.initial_scale = 1.0
    else:
      raise ValueError('Unknown scaling algorithm: {}'.format(algorithm))

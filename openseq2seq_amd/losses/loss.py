"""Abstract Loss — open_seq2seq/losses/loss.py:44-131."""
import abc
import copy

import six

from ..utils.utils import check_params


@six.add_metaclass(abc.ABCMeta)
class Loss(object):
  @staticmethod
  def get_required_params():
    return {}

  @staticmethod
  def get_optional_params():
    return {'dtype': None}

  def __init__(self, params, model, name="loss"):
    check_params(params, self.get_required_params(), self.get_optional_params())
    self._params = copy.deepcopy(params)
    self._model = model
    if 'dtype' not in self._params:
      self._params['dtype'] = model.params['dtype'] if model else "mixed"
    self._name = name

  def compute_loss(self, input_dict):
    return self._compute_loss(input_dict)

  @abc.abstractmethod
  def _compute_loss(self, input_dict):
    pass

  @property
  def params(self):
    return self._params

  @property
  def name(self):
    return self._name

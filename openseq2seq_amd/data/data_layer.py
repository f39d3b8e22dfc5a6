"""Abstract DataLayer — open_seq2seq/data/data_layer.py:16-157."""
import abc
import copy

import six

from ..utils.utils import check_params


@six.add_metaclass(abc.ABCMeta)
class DataLayer(object):
  @staticmethod
  def get_required_params():
    return {'mode': ['train', 'eval', 'infer', 'interactive_infer']}

  @staticmethod
  def get_optional_params():
    return {'batch_size': int, 'shuffle': bool, 'repeat': bool, 'dtype': None,
            'interactive': bool, 'cache_features': bool, 'cache_format': str,
            'cache_regenerate': bool}

  def __init__(self, params, model, num_workers, worker_id):
    check_params(params, self.get_required_params(), self.get_optional_params())
    self._params = copy.deepcopy(params)
    self._model = model
    if 'shuffle' not in self._params:
      self._params['shuffle'] = (self._params['mode'] == 'train')
    if 'repeat' not in self._params:
      self._params['repeat'] = (self._params['mode'] == 'train')
    self._num_workers = num_workers
    self._worker_id = worker_id

  @property
  def params(self):
    return self._params

  @abc.abstractmethod
  def build_graph(self):
    pass

  @property
  @abc.abstractmethod
  def input_tensors(self):
    pass

  def get_size_in_samples(self):
    return None

"""Abstract Decoder — the plugin API of open_seq2seq/decoders/decoder.py:49-192:
`Encoder(params, model, name, mode)`, static get_required_params/get_optional_params
validated with check_params (unknown key => ValueError), `encode(input_dict)` ->
`_encode`. Variables are created in `build(store, ...)` (the graph-construction
phase of the reference) on the model's FlatParams store."""
import abc
import copy

import six

from ..utils.utils import check_params


@six.add_metaclass(abc.ABCMeta)
class Decoder(object):
  @staticmethod
  def get_required_params():
    return {}

  @staticmethod
  def get_optional_params():
    return {
        'regularizer': None,
        'regularizer_params': dict,
        'initializer': None,
        'initializer_params': dict,
        'dtype': None,
    }

  def __init__(self, params, model, name="decoder", mode='train'):
    check_params(params, self.get_required_params(), self.get_optional_params())
    self._params = copy.deepcopy(params)
    self._model = model
    if 'dtype' not in self._params:
      self._params['dtype'] = model.params['dtype'] if model else "mixed"
    self._name = name
    self._mode = mode
    self._compiled = False

  def decode(self, input_dict):
    """decoder.py:95-153 (dtype casting / regularizer scoping is implicit here)."""
    return self._decode(input_dict)

  @abc.abstractmethod
  def _decode(self, input_dict):
    pass

  @property
  def params(self):
    return self._params

  @property
  def mode(self):
    return self._mode

  @property
  def name(self):
    return self._name

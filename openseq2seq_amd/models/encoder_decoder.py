"""EncoderDecoderModel — open_seq2seq/models/encoder_decoder.py:32-175: builds the
Encoder / Decoder / Loss plugin objects from the config and glues
encode -> decode -> compute_loss."""
from __future__ import absolute_import, division, print_function

from .model import Model


class EncoderDecoderModel(Model):
  @staticmethod
  def get_required_params():
    return dict(Model.get_required_params(), **{
        'encoder': None,
        'decoder': None,
    })

  @staticmethod
  def get_optional_params():
    return dict(Model.get_optional_params(), **{
        'encoder_params': dict,
        'decoder_params': dict,
        'loss': None,
        'loss_params': dict,
    })

  def __init__(self, params, mode="train", hvd=None, device=None):
    super(EncoderDecoderModel, self).__init__(params, mode=mode, hvd=hvd, device=device)
    if 'encoder_params' not in self.params:
      self.params['encoder_params'] = {}
    if 'decoder_params' not in self.params:
      self.params['decoder_params'] = {}
    if 'loss_params' not in self.params:
      self.params['loss_params'] = {}
    self._encoder = self._decoder = self._loss_computator = None

  def _create_encoder(self):
    params = self.params['encoder_params']
    return self.params['encoder'](params=params, mode=self.mode, model=self)

  def _create_decoder(self):
    params = self.params['decoder_params']
    return self.params['decoder'](params=params, mode=self.mode, model=self)

  def _create_loss(self):
    return self.params['loss'](params=self.params['loss_params'], model=self)

  def _create_data_layer(self):
    dl_params = dict(self.params.get('data_layer_params', {}))
    dl_params['batch_size'] = self.params['batch_size_per_gpu']
    dl_params['mode'] = self.mode
    world = self._hvd.size() if self._hvd is not None else 1
    rank = self._hvd.rank() if self._hvd is not None else 0
    return self.params['data_layer'](params=dl_params, model=self, num_workers=world,
                                     worker_id=rank)

  def get_encoder(self):
    return self._encoder

  def get_decoder(self):
    return self._decoder

  def get_loss_computator(self):
    return self._loss_computator

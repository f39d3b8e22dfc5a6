"""Text2Speech / Text2SpeechTacotron — open_seq2seq/models/text2speech.py:205-317,
text2speech_tacotron.py (model shell: feature sizes flow from the data layer to the
decoder and the loss; plotting / Griffin-Lim audio export are host-side and out of scope).
The reference defines no objects-per-step for TTS; benchmarks here count target mel frames."""
from __future__ import absolute_import, division, print_function

from .encoder_decoder import EncoderDecoderModel
from ..parts.transformer.layers import SeedSeq


class Text2Speech(EncoderDecoderModel):
  @staticmethod
  def get_required_params():
    return dict(EncoderDecoderModel.get_required_params(), **{})

  def _build_forward_pass_objects(self, store):
    self._data_layer = self._create_data_layer()
    self._encoder = self._create_encoder()
    self._decoder = self._create_decoder()
    if self.mode in ("train", "eval"):
      self._loss_computator = self._create_loss()
    self._encoder.build(store)
    self._decoder.build(store, memory_dim=self._encoder.output_dim)

  def _forward_backward(self, batch, tape):
    seeds = SeedSeq(self._seed * 7919 + self._step_count)
    enc = self._encoder.encode({'source_tensors': batch['source_tensors'], 'tape': tape,
                                'seeds': seeds})
    dec = self._decoder.decode({'encoder_output': enc, 'target_tensors': batch['target_tensors'],
                                'tape': tape})
    scale_dev = self._train_op.loss_scale_view if self._train_op is not None else None
    return self._loss_computator.compute_loss({
        'decoder_output': dec, 'target_tensors': batch['target_tensors'],
        'loss_scale_dev': scale_dev})

  def infer_batch(self, batch):
    enc = self._encoder.encode({'source_tensors': batch['source_tensors']})
    return self._decoder.decode({'encoder_output': enc})

  def _get_num_objects_per_step(self, batch):
    return batch['target_tensors'][2].sum()


class Text2SpeechTacotron(Text2Speech):
  pass

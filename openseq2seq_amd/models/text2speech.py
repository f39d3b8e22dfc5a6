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

  def infer_batch(self, batch, max_decoder_steps=None):
    """infer (text2speech.py:205-317 hands the decoder outputs to plotting / Griffin-Lim): free-running
    decode of one batch. max_decoder_steps overrides the reference's 10 x max(src_len) cap (benchmarks)."""
    enc = self._encoder.encode({'source_tensors': batch['source_tensors']})
    return self._decoder.decode({'encoder_output': enc, 'max_decoder_steps': max_decoder_steps})

  def evaluate_batch(self, batch):
    """Eval mode of the reference graph (utils/funcs.py:293-340 sums `eval_losses`): free-running decode,
    then Text2SpeechLoss with prediction and target padded to a common length
    (losses/text2speech_loss.py:80-131). Returns (loss, target frames)."""
    enc = self._encoder.encode({'source_tensors': batch['source_tensors']})
    dec = self._decoder.decode({'encoder_output': enc})
    loss = self._loss_computator.compute_loss({'decoder_output': dec, 'target_tensors': batch['target_tensors'],
                                               'want_grad': False})
    return float(loss.cpu()[0]), int(batch['target_tensors'][2].sum().item())

  def evaluate(self, device=None, max_batches=None):
    """One pass over the eval data layer (run.py --mode=eval / train_eval): mean eval loss per batch,
    'Validation loss' of utils/funcs.py:335-340."""
    dl = self.get_data_layer()
    losses, frames = [], 0
    for n, batch in enumerate(dl.iterate_batches(device or self._device, drop_remainder=False)):
      if max_batches is not None and n >= max_batches:
        break
      l, f = self.evaluate_batch(batch)
      losses.append(l)
      frames += f
    return {"eval_loss": sum(losses) / max(len(losses), 1), "batches": len(losses), "target_frames": frames}

  def _get_num_objects_per_step(self, batch):
    return batch['target_tensors'][2].sum()


class Text2SpeechTacotron(Text2Speech):
  pass

"""PaddedCrossEntropyLossWithSmoothing — open_seq2seq/losses/sequence_loss.py:233-309 on the
fused HIP kernel: soft targets (1-s at the label, s/(V-1) elsewhere), minus the smoothing
entropy constant, weights = (label != 0), sum / sum(weights). In the packed layout every
row is a non-pad target position, so sum(weights) = number of rows."""
from __future__ import absolute_import, division, print_function

from .loss import Loss
from .. import capi


class PaddedCrossEntropyLossWithSmoothing(Loss):
  @staticmethod
  def get_optional_params():
    return dict(Loss.get_optional_params(), **{
        'batch_size': int, 'tgt_vocab_size': int, 'label_smoothing': float,
        'pad_embeddings_2_eight': bool,
    })

  def __init__(self, params, model, name="padded_cross_entropy_with_smoothing"):
    super(PaddedCrossEntropyLossWithSmoothing, self).__init__(params, model, name)
    self._label_smoothing = self.params.get("label_smoothing", 0.0)

  def _compute_loss(self, input_dict):
    dec = input_dict["decoder_output"]
    logits = dec["logits"]                       # [N_tgt, V] bf16 (packed)
    labels = dec["packed_target"]["labels"]
    la = dec.get("logits_act")
    want_grad = la is not None and input_dict.get("want_grad", True)
    _, mean, dl = capi.xent_smooth(logits, labels, self._label_smoothing,
                                   grad_scale_dev=input_dict.get("loss_scale_dev"),
                                   want_grad=want_grad)
    if want_grad:
      la.grad = dl
      la.grad_init = True
    return mean

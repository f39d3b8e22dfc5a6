"""CTCLoss — open_seq2seq/losses/ctc_loss.py:19-88 on the HIP CTC kernel: the loss
is forced to fp32 (:42), labels are taken by LENGTH not by pad value (:12-16),
infeasible samples are ignored, non-finite losses masked to 0, batch mean."""
from __future__ import absolute_import, division, print_function

from .loss import Loss
from .. import capi


class CTCLoss(Loss):
  @staticmethod
  def get_optional_params():
    return dict(Loss.get_optional_params(), **{'mask_nan': bool})

  def __init__(self, params, model, name="ctc_loss"):
    super(CTCLoss, self).__init__(params, model, name)
    self._mask_nan = self.params.get("mask_nan", True)
    self.params['dtype'] = "float32"

  def _compute_loss(self, input_dict):
    """decoder_output: {'logits': [T,B,V] fp32, 'src_length': [B]};
    target_tensors: [tgt_sequence int32 [B,L], tgt_length int32 [B]].
    Returns the averaged CTC loss (device scalar, shape [1]). In training mode the
    gradient w.r.t. the logits (times loss_scale / B) is produced in the same call
    and handed to the decoder's backward."""
    dec = input_dict['decoder_output']
    logits, src_length = dec['logits'], dec['src_length']
    tgt_sequence, tgt_length = input_dict['target_tensors']
    sink = dec.get('_dlogits_sink')
    B = logits.shape[1]
    res = capi.ctc_loss(logits, src_length, tgt_sequence, tgt_length,
                        grad_scale=1.0 / B, grad_scale_dev=input_dict.get('loss_scale_dev'),
                        want_grad=False, want_grad_bf16=sink is not None,
                        vpad=input_dict.get('vpad', 32))
    if sink is not None:
      sink['dlogits_bf16'] = res['dlogits_bf16']
    self.last_loss_per_sample = res['loss_per_sample']
    return res['loss_mean']

"""BasicSequenceLoss (open_seq2seq/losses/sequence_loss.py:10-114) and
PaddedCrossEntropyLossWithSmoothing — open_seq2seq/losses/sequence_loss.py:233-309 on the
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


class BasicSequenceLoss(Loss):
  """Sparse softmax cross entropy of logits[:, t] against target[:, t+1]
  (offset_target_by_one), masked with sequence_mask(tgt_len - 1), summed and divided by the
  batch size — or averaged over all B x T entries with average_across_timestep
  (sequence_loss.py:53-114). One fused kernel produces the loss and d(logits); masked
  positions are rows with label -1."""

  @staticmethod
  def get_required_params():
    return dict(Loss.get_required_params(), **{'tgt_vocab_size': int, 'batch_size': int})

  @staticmethod
  def get_optional_params():
    return dict(Loss.get_optional_params(), **{
        'offset_target_by_one': bool, 'average_across_timestep': bool, 'do_mask': bool,
    })

  def __init__(self, params, model, name="basic_sequence_loss"):
    super(BasicSequenceLoss, self).__init__(params, model, name)
    self._tgt_vocab_size = self.params["tgt_vocab_size"]
    self._batch_size = self.params["batch_size"]
    self._offset_target_by_one = self.params.get("offset_target_by_one", True)
    self._average_across_timestep = self.params.get("average_across_timestep", False)
    self._do_mask = self.params.get("do_mask", True)

  def loss_labels(self, target, tgt_len, t_logits):
    """int32 [B, t_logits]: the label each logits row is scored against, -1 = masked."""
    import torch
    B, Lt = target.shape
    off = 1 if self._offset_target_by_one else 0
    cur = min(Lt, t_logits) - off
    labels = torch.full((B, t_logits), -1, dtype=torch.int32, device=target.device)
    if cur > 0:
      lab = target[:, off:off + cur].to(torch.int32)
      if self._do_mask:
        if tgt_len is None:
          raise ValueError("If you are masking loss, tgt_lengths can't be None")
        pos = torch.arange(cur, device=target.device)[None, :]
        lab = torch.where(pos < (tgt_len[:, None] - 1), lab, torch.full_like(lab, -1))
      labels[:, :cur] = lab
    return labels, cur

  def _compute_loss(self, input_dict):
    dec = input_dict["decoder_output"]
    logits = dec["logits"]                                  # [B, T, Vpad] bf16
    target, tgt_len = input_dict['target_tensors'][0], input_dict['target_tensors'][1]
    B, T, Vp = logits.shape
    labels, cur = self.loss_labels(target, tgt_len, T)
    scale = 1.0 / (B * max(cur, 1)) if self._average_across_timestep else 1.0 / self._batch_size
    la = dec.get("logits_act")
    want_grad = la is not None and input_dict.get("want_grad", True)
    _, loss, dl = capi.xent_smooth(logits.reshape(B * T, Vp), labels.reshape(-1).contiguous(), 0.0,
                                   grad_scale_dev=input_dict.get("loss_scale_dev"),
                                   want_grad=want_grad, v_valid=self._tgt_vocab_size,
                                   grad_scale=scale)
    if want_grad:
      la.grad = dl.view(B, T, Vp)
      la.grad_init = True
    return loss

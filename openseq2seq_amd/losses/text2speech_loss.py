"""Text2SpeechLoss — open_seq2seq/losses/text2speech_loss.py:12-209 on the fused loss kernel
(csrc/tts.hip): masked MSE (or L1) of the decoder mel, the post-net mel and (output_type
"both") the magnitude prediction against the target spectrogram, plus the masked sigmoid
cross entropy of the stop token; each term produces its gradient in the same pass. Predictions and
targets of different lengths (eval mode) are padded to the longer one inside the kernel (:80-131)."""
from __future__ import absolute_import, division, print_function

import torch

from .loss import Loss
from .. import capi
from ..parts.cnns.conv_blocks import accumulate_grad


class Text2SpeechLoss(Loss):
  @staticmethod
  def get_optional_params():
    return dict(Loss.get_optional_params(), **{
        'use_mask': bool, 'scale': float, 'stop_token_weight': float, 'mel_weight': float,
        'mag_weight': float, 'l1_norm': bool,
    })

  def __init__(self, params, model, name="tacotron_loss"):
    super(Text2SpeechLoss, self).__init__(params, model, name)

  def _compute_loss(self, input_dict):
    dec = input_dict["decoder_output"]
    acts = dec["acts"]
    n_mel, n_mag = dec["n_feats"]
    spec, stop_token, spec_len = input_dict["target_tensors"][:3]
    p = self.params
    B, T, _ = spec.shape
    # acts["mel"].data.shape[1] != T (eval / infer: the free-running decoder ran for its own number of
    # steps): the kernel pads both sides to the longer one, :80-131 — predictions with zeros, the
    # spectrogram with zeros and the stop token with ones
    lens = spec_len if p.get("use_mask", True) else None
    mode = capi.LOSS_L1 if p.get("l1_norm", False) else capi.LOSS_MSE
    scale = p.get("scale", None) or 1.0
    mel_w = p.get("mel_weight", 1.0) * scale
    gsd = input_dict.get("loss_scale_dev")
    want = input_dict.get("want_grad", True)
    loss = torch.zeros(1, dtype=torch.float32, device=spec.device)
    spec = spec.float()

    def term(act, target, F, mode_, w, pad=0.0):
      d = capi.tts_loss(act.data, target, lens, F, mode_, w, loss, grad_scale_dev=gsd, want_grad=want,
                        target_pad=pad)
      if want:
        accumulate_grad(act, d)

    term(acts["mel"], spec[:, :, :n_mel], n_mel, mode, mel_w)
    term(acts["post"], spec[:, :, :n_mel], n_mel, mode, mel_w)
    if acts.get("mag") is not None:
      term(acts["mag"], spec[:, :, n_mel:n_mel + n_mag], n_mag, mode, p.get("mag_weight", 1.0) * scale)
    st = stop_token.float().reshape(B, T, 1)
    term(acts["stop"], st, 1, capi.LOSS_SIGMOID_XENT, p.get("stop_token_weight", 1.0) * scale, pad=1.0)
    return loss

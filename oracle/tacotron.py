"""ORACLE (test infrastructure only): CPU fp32 restatement of the Tacotron2 training pass:
Tacotron2Encoder._encode (open_seq2seq/encoders/tacotron2_encoder.py:104-339),
Tacotron2Decoder._decode in train mode (decoders/tacotron2_decoder.py:257-567; Prenet
:22-63; TacotronTrainingHelper parts/tacotron/tacotron_helper.py:46-135;
LocationSensitiveAttention parts/rnns/attention_wrapper.py:641-878) and Text2SpeechLoss
(losses/text2speech_loss.py:35-209). Weights use the device layout ([out, in] / conv
[K, Cout, Cin]); dropout masks are passed explicitly (None = off).
PARITY STATUS: unpinned by the reference (no value tests for this model)."""
import torch
import torch.nn.functional as F

from . import attn_decoder as oad
from . import cnn


def conv_bn_act(x, w_dev, gamma, beta, eps, act, mask=None):
  """conv_bn_actv: conv1d SAME (no bias) -> fused BN (train statistics) -> activation
  [-> dropout mask]; w_dev [K, Cout, Cin]."""
  y = cnn.conv1d_tf(x, w_dev.permute(0, 2, 1))
  y = cnn.batch_norm_train(y, gamma, beta, eps)[0]
  y = cnn.act_fn(y, act)
  return y if mask is None else y * mask


def encoder(P, text, lstm, bn_eps=1e-5, style=None):
  """P: emb [V,E]; convs: list of (w, gamma, beta); lstm: torch.nn.LSTM(bidirectional,
  batch_first) holding the cuDNN-form weights (run over the padded batch, no lengths).
  style: optional [B, Es] global style embedding tiled over time."""
  x = P["emb"][text.long()]
  for w, g, b in P["convs"]:
    x = conv_bn_act(x, w, g, b, bn_eps, "relu")
  if lstm is not None:
    x, _ = lstm(x)
  if style is not None:
    x = torch.cat([x, style[:, None, :].expand(-1, x.shape[1], -1)], -1)
  return x


def decoder(P, enc_out, src_len, spec_mel, postnet_acts, bn_eps=1e-5, exp_mag=True,
            prenet_masks=None, out_masks=None, post_masks=None):
  """P: prenet [(w [out,in], b)], cell (dict for oad.attention_decoder + w_in, b0),
  out_w [n_mel, H+M], out_b, stop_w [1, n_mel], stop_b [1], postnet [(w,g,b)],
  mag: dict(c0=(w,g,b), c1=(w,g,b), proj [n_mag, 512]) or None.
  Returns dict(mel, post, stop (logits [B,T,1]), mag, align)."""
  B, T, nm = spec_mel.shape
  prev = torch.cat([torch.zeros_like(spec_mel[:, :1]), spec_mel[:, :-1]], 1)
  x = prev
  for i, (w, b) in enumerate(P["prenet"]):
    x = torch.relu(x @ w.t() + b)
    if prenet_masks is not None:
      x = x * prenet_masks[i]
  c = P["cell"]
  gx0 = x @ c["w_in"].t() + c["b0"]
  r = oad.attention_decoder(c, gx0, enc_out, src_len, None, None, out_masks, 1.0, "location")
  both = torch.cat([r["y"], r["ctx"]], -1)
  mel = both @ P["out_w"].t() + P["out_b"]
  stop = mel @ P["stop_w"].t() + P["stop_b"]
  top = mel
  for i, ((w, g, b), act) in enumerate(zip(P["postnet"], postnet_acts)):
    top = conv_bn_act(top, w, g, b, bn_eps, act, None if post_masks is None else post_masks[i])
  post = mel + top
  mag = None
  if P.get("mag") is not None:
    m = conv_bn_act(post, *P["mag"]["c0"], bn_eps, "relu")
    m = conv_bn_act(m, *P["mag"]["c1"], bn_eps, "relu")
    if exp_mag:
      m = torch.exp(m)
    mag = m @ P["mag"]["proj"].t()
  return dict(mel=mel, post=post, stop=stop, mag=mag, align=r["align"])


def text2speech_loss(out, spec, stop_token, spec_len, n_mel, n_mag, l1=False):
  """Masked MSE (SUM_BY_NONZERO_WEIGHTS over the broadcast mask) + masked stop xent."""
  B, T, _ = spec.shape
  mask = cnn.seq_mask(spec_len, T)

  def reg(pred, tgt):
    err = (pred - tgt).abs() if l1 else (pred - tgt) ** 2
    return (err * mask).sum() / (mask.sum() * pred.shape[-1])

  loss = reg(out["mel"], spec[..., :n_mel]) + reg(out["post"], spec[..., :n_mel])
  if out.get("mag") is not None:
    loss = loss + reg(out["mag"], spec[..., n_mel:n_mel + n_mag])
  xe = F.binary_cross_entropy_with_logits(out["stop"], stop_token[..., None], reduction="none")
  return loss + (xe * mask).sum() / mask.sum()

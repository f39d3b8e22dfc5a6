"""ORACLE (test infrastructure only): CPU fp32 restatement of the Tacotron2 training pass:
Tacotron2Encoder._encode (open_seq2seq/encoders/tacotron2_encoder.py:104-339),
Tacotron2Decoder._decode in train mode (decoders/tacotron2_decoder.py:257-567; Prenet
:22-63; TacotronTrainingHelper parts/tacotron/tacotron_helper.py:46-135;
LocationSensitiveAttention parts/rnns/attention_wrapper.py:641-878) and Text2SpeechLoss
(losses/text2speech_loss.py:35-209). Weights use the device layout ([out, in] / conv
[K, Cout, Cin]); dropout masks are passed explicitly (None = off).
PARITY STATUS (round 5): decoder, free-running decoder_infer and text2speech_loss are pinned to the
reference's OWN CODE — Tacotron2Decoder._decode (train and eval mode), TacotronDecoder / helpers,
LocationSensitiveAttention and Text2SpeechLoss executed from their files on the TF-primitive stand-in
oracle/ref_shim/tf1 with the pre-net's dropout masks recorded: outputs 1e-5, gradients 3e-6, 30
free-running steps 1e-4 with the same lengths; the encoder with global style tokens likewise (outputs
1e-5, gradients 7e-7; the cuDNN LSTM is torch.nn.LSTM on both sides: its wiring is what is pinned) —
tests/test_ref_exec_tacotron.py."""
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


def text2speech_loss(out, spec, stop_token, spec_len, n_mel, n_mag, l1=False, use_mask=True,
                     mel_weight=1.0, mag_weight=1.0, stop_token_weight=1.0, scale=None):
  """losses/text2speech_loss.py:35-209. Predictions (out["mel"], out["post"], out["stop"] logits
  [B,Tp,1], out["mag"]) and targets (spec [B,Tt,n_mel(+n_mag)], stop_token [B,Tt]) are first padded
  to max_length = max(Tp, Tt) (:80-117): predictions and the spectrogram with zeros, the stop-token
  TARGET with ones. Then masked (mask = sequence_mask(spec_len, max_length)) MSE / L1 with
  tf.losses' SUM_BY_NONZERO_WEIGHTS reduction (sum(w * err) / #non-zero broadcast weights) plus the
  masked stop-token sigmoid cross entropy / sum(mask) (:139-176), or their plain means without a
  mask (:177-195); weights and scale as :197-209."""
  B, Tt, _ = spec.shape
  Tp = out["mel"].shape[1]
  T = max(Tp, Tt)

  def pad_t(x, value=0.0):
    return x if x.shape[1] == T else F.pad(x, (0, 0, 0, T - x.shape[1]), value=value)

  mel, post, stop = pad_t(out["mel"]), pad_t(out["post"]), pad_t(out["stop"])
  spec = pad_t(spec)
  stop_t = pad_t(stop_token[..., None].float(), 1.0)
  mag = pad_t(out["mag"]) if out.get("mag") is not None else None
  mask = cnn.seq_mask(spec_len, T) if use_mask else torch.ones(B, T, 1)

  def reg(pred, tgt):
    err = (pred - tgt).abs() if l1 else (pred - tgt) ** 2
    if not use_mask:
      return err.mean()
    return (err * mask).sum() / (mask.sum() * pred.shape[-1])

  loss = mel_weight * (reg(mel, spec[..., :n_mel]) + reg(post, spec[..., :n_mel]))
  if mag is not None:
    loss = loss + mag_weight * reg(mag, spec[..., n_mel:n_mel + n_mag])
  xe = F.binary_cross_entropy_with_logits(stop, stop_t, reduction="none")
  stop_loss = (xe * mask).sum() / mask.sum() if use_mask else xe.mean()
  loss = loss + stop_token_weight * stop_loss
  return loss * scale if scale else loss


def decoder_infer(P, enc_out, src_len, max_steps=None, prenet_masks=None, mask_decoder_sequence=True,
                  round_frames_bf16=True):
  """Free-running decoding: Tacotron2Decoder._decode in eval / infer mode
  (decoders/tacotron2_decoder.py:378-428) = dynamic_decode(TacotronDecoder(helper=TacotronHelper),
  impute_finished=False, maximum_iterations=10 * max(src_len)); TacotronDecoder.step
  (parts/tacotron/tacotron_decoder.py:153-190): cell on prenet(inputs) -> spec projection -> stop
  projection -> helper.next_inputs (tacotron_helper.py:195-226): finished = round(sigmoid(stop)),
  next input = the projected frame. dynamic_decode: sequence_lengths[b] = first step count at which b
  was finished (the finishing step counts); the loop stops once every sample has finished.

  P as for `decoder` (prenet, cell incl. w_in / b0, out_w, out_b, stop_w, stop_b). prenet_masks:
  per pre-net layer a [T, B, units] keep mask already scaled by 1/keep (the always-on dropout), or None.
  round_frames_bf16: frames, stop logits and pre-net outputs are bf16 tensors on the device; rounding
  them here keeps a long free-running trajectory comparable.
  Returns dict(mel [B,steps,n_mel], stop [B,steps] logits, align [B,steps,S], lengths [B], steps)."""
  rb = (lambda x: x.to(torch.bfloat16).float()) if round_frames_bf16 else (lambda x: x)
  B, S, M = enc_out.shape
  c = P["cell"]
  L = len(c["wcat"])
  H = c["wq"].shape[1]
  T = int(max_steps) if max_steps else 10 * int(torch.as_tensor(src_len).max())
  values, mask = oad.prepare_memory(enc_out, src_len)
  keys = values @ c["wmem"].t()
  h = [enc_out.new_zeros(B, H) for _ in range(L)]
  cs = [enc_out.new_zeros(B, H) for _ in range(L)]
  attn = enc_out.new_zeros(B, M)
  cum = enc_out.new_zeros(B, S)
  frame = enc_out.new_zeros(B, P["out_w"].shape[0])
  finished = torch.zeros(B, dtype=torch.bool)
  lengths = torch.zeros(B, dtype=torch.int32)
  mels, stops, aligns = [], [], []
  steps = T
  for t in range(T):
    x = frame
    for i, (w, b) in enumerate(P["prenet"]):
      x = torch.relu(x @ w.t() + b)
      if prenet_masks is not None:
        x = x * prenet_masks[i][t]
      x = rb(x)
    gx0 = x @ c["w_in"].t() + c["b0"]
    xin = None
    for l in range(L):
      if l == 0:
        hn, cn = oad.lstm_cell(torch.cat([attn, h[0]], -1), cs[0], c["wcat"][0], gx0, 1.0)
      else:
        hn, cn = oad.lstm_cell(torch.cat([xin, h[l]], -1), cs[l], c["wcat"][l], c["bias"][l], 1.0)
      hn = rb(hn)                    # the state rows the next GEMMs read are bf16
      h[l], cs[l] = hn, cn
      xin = hn
    q = xin @ c["wq"].t()
    pre = keys + q[:, None, :] + oad.location_features(cum, c["conv_w"], c["conv_b"], c["dense_w"])
    if c.get("b") is not None:
      pre = pre + c["b"]
    al = oad.masked_softmax((c["v"] * torch.tanh(pre)).sum(-1), mask)
    attn_full = (al[:, :, None] * values).sum(1)
    cum = cum + al
    mel = rb(torch.cat([xin, attn_full], -1) @ P["out_w"].t() + P["out_b"])
    stop = rb(mel @ P["stop_w"].t() + P["stop_b"])[:, 0]
    attn = rb(attn_full)
    mels.append(mel)
    stops.append(stop)
    aligns.append(al)
    lengths = lengths + (~finished).to(torch.int32)
    if mask_decoder_sequence:
      finished = finished | (torch.sigmoid(stop) > 0.5)      # tf.round: half to even, 0.5 -> 0
    frame = mel
    if bool(finished.all()):
      steps = t + 1
      break
  return dict(mel=torch.stack(mels, 1), stop=torch.stack(stops, 1), align=torch.stack(aligns, 1),
              lengths=lengths, steps=steps)

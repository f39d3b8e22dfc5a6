"""ORACLE (test infrastructure only): CPU fp32 restatement of the RNN NMT model
(example_configs/text2text/en-de/en-de-nmt-small.py): BidirectionalRNNEncoderWithEmbedding
(encoders/rnn_encoders.py:221-305) -> RNNDecoderWithAttention, gnmt / gnmt_v2
(decoders/rnn_decoders.py:147-321, parts/rnns/gnmt.py:32-79) -> BasicSequenceLoss
(losses/sequence_loss.py:53-114). Dropout is off (keep probabilities 1.0); the dropout
plumbing of the kernels is pinned separately in tests/test_attn_decoder_gpu.py.
PARITY STATUS (round 5): pinned to the reference's OWN CODE — both encoders,
RNNDecoderWithAttention (gnmt / gnmt_v2 / skip connections) and BasicSequenceLoss executed from their
files on the TF-primitive stand-in oracle/ref_shim/tf1: outputs / logits / loss 1e-5, all gradients
6e-7 (tests/test_ref_exec_nmt.py); under oracle/rnn_beam_search.py the decoder returns what the reference's
BeamSearchRNNDecoderWithAttention returns (tests/test_ref_exec_nmt_beam.py). tf.nn.rnn_cell.LSTMCell, dynamic_rnn and dynamic_decode are
TensorFlow library code, restated in oracle/ref_shim/tf1/rnn.py."""
import torch
import torch.nn.functional as F

from . import attn_decoder as oad
from . import rnn as orn


def encoder(P, ids, lens):
  """P['emb'] [V,E]; P['fw'] / P['bw']: list of layers dict(wx [4H,In], wh [4H,H], b [4H])."""
  x = P["emb"][ids.long()]
  outs = []
  for key, rev in (("fw", False), ("bw", True)):
    if key not in P:
      continue
    h = x
    for lyr in P[key]:
      h = orn.lstm_tf(h, lens, lyr["wx"].t(), lyr["wh"].t(), lyr["b"], lyr.get("forget_bias", 1.0), rev)
    outs.append(h)
  return torch.cat(outs, -1)


def gnmt_like_encoder(P, ids, lens):
  """GNMTLikeEncoderWithEmbedding (encoders/rnn_encoders.py:320-470): one bidirectional layer
  (P['l1fw'], P['l1bw']), then unidirectional layers P['uni'] — each but the first with a residual
  connection (tf.contrib.rnn.ResidualWrapper)."""
  x = P["emb"][ids.long()]
  outs = []
  for key, rev in (("l1fw", False), ("l1bw", True)):
    lyr = P[key]
    outs.append(orn.lstm_tf(x, lens, lyr["wx"].t(), lyr["wh"].t(), lyr["b"], lyr.get("forget_bias", 1.0), rev))
  h = torch.cat(outs, -1)
  for l, lyr in enumerate(P["uni"]):
    y = orn.lstm_tf(h, lens, lyr["wx"].t(), lyr["wh"].t(), lyr["b"], lyr.get("forget_bias", 1.0), False)
    h = y + h if l > 0 else y
  return h


def decoder_logits(P, enc_out, src_len, tgt, tgt_len, attention_type="gnmt_v2", skip=False):
  """Teacher-forced logits [B,T,V]. P['demb'] [V,E]; P['cell']: dict for
  oracle.attn_decoder.attention_decoder + 'w_in' [4H,E], 'b0' [4H]; P['upper']: list of
  dict(wx_h [4H,H], wx_a [4H,M], wh [4H,H], b [4H]); P['proj'] [V,H]."""
  x = P["demb"][tgt.long()]
  c = P["cell"]
  gx0 = x @ c["w_in"].t() + c["b0"]
  r = oad.attention_decoder(c, gx0, enc_out, src_len, tgt_len, None, None, c.get("forget_bias", 1.0),
                            "bahdanau_norm")
  top, ctx = r["y"], r["ctx"]
  if attention_type == "gnmt":
    ctx = torch.cat([torch.zeros_like(ctx[:, :1]), ctx[:, :-1]], 1)
  for li, lyr in enumerate(P["upper"]):
    wx = torch.cat([lyr["wx_h"], lyr["wx_a"]], 1)
    y = orn.lstm_tf(torch.cat([top, ctx], -1), tgt_len, wx.t(), lyr["wh"].t(), lyr["b"],
                    lyr.get("forget_bias", 1.0), False)
    # decoder_use_skip_connections: ResidualWrapper(gnmt_residual_fn) on the upper cells from the
    # second one on (rnn_decoders.py:138-146): + the layer-input part of the cell inputs
    top = y + top if (skip and li >= 1) else y
  return top @ P["proj"].t()


def basic_sequence_loss(logits, tgt, tgt_len, batch_size, average_across_timestep=False):
  """offset_target_by_one + sequence_mask(len - 1) + sparse softmax xent, sum / batch."""
  B, T, V = logits.shape
  cur = min(tgt.shape[1], T) - 1
  lg = logits[:, :cur]
  lab = tgt[:, 1:1 + cur].long()
  mask = (torch.arange(cur)[None, :] < (torch.as_tensor(tgt_len)[:, None] - 1)).to(logits.dtype)
  xe = F.cross_entropy(lg.reshape(-1, V), lab.reshape(-1), reduction="none").view(B, cur)
  if average_across_timestep:
    return (xe * mask).mean()
  return (xe * mask).sum() / batch_size

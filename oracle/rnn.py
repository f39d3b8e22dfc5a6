"""ORACLE (test infrastructure only): CPU fp32 recurrent layers.

  * cuDNN-form GRU / LSTM: torch.nn.GRU / torch.nn.LSTM share cuDNN's gate formulation
    (r,z,n / i,f,g,o, two bias vectors) — what tf.contrib.cudnn_rnn.CudnnGRU/CudnnLSTM
    compute (encoders/ds2_encoder.py:294-328, tacotron2_encoder.py:254-263).
  * `lstm_tf`: restatement of tf.nn.rnn_cell.LSTMCell (gate order i, j, f, o; forget_bias
    added inside the sigmoid; one kernel over concat(x, h)) under dynamic_rnn with
    sequence_length (state copy-through and zero outputs past the end), and
    bidirectional_dynamic_rnn's per-sequence reversal (encoders/rnn_encoders.py:292-300).
PARITY STATUS: unpinned by the reference (no value tests). lstm_tf is
cross-checked against torch.nn.LSTM with permuted gates in tests/test_oracle_rnn.py."""
import torch


def _reverse_by_len(x, lens):
  """tf.reverse_sequence along time for [B,T,...]."""
  out = x.clone()
  for b in range(x.shape[0]):
    n = int(lens[b])
    out[b, :n] = x[b, :n].flip(0)
  return out


def lstm_tf(x, lens, wx, wh, bias, forget_bias=1.0, reverse=False):
  """x [B,T,In]; wx [In,4H], wh [H,4H] (TF kernel rows split), bias [4H]; gate order i,j,f,o.
  Returns outputs [B,T,H] (zero past lens)."""
  B, T, _ = x.shape
  H = wh.shape[0]
  if lens is None:
    lens = torch.full((B,), T)
  xin = _reverse_by_len(x, lens) if reverse else x
  h = x.new_zeros(B, H)
  c = x.new_zeros(B, H)
  outs = []
  for t in range(T):
    z = xin[:, t] @ wx + h @ wh + bias
    i, j, f, o = z.chunk(4, dim=-1)
    cn = c * torch.sigmoid(f + forget_bias) + torch.sigmoid(i) * torch.tanh(j)
    hn = torch.tanh(cn) * torch.sigmoid(o)
    live = (t < torch.as_tensor(lens)).float()[:, None]
    c = live * cn + (1 - live) * c
    h = live * hn + (1 - live) * h
    outs.append(hn * live)
  y = torch.stack(outs, 1)
  return _reverse_by_len(y, lens) if reverse else y


def cudnn_rnn(kind, x, lens, wx, wh, bx, bh, reverse=False):
  """kind 'gru' | 'lstm'. wx [G*H, In], wh [G*H, H] (torch/cuDNN layout), biases [G*H].
  cuDNN itself has no sequence lengths; when `lens` is given the sequence is reversed per
  sample and outputs past the end are zeroed (packed-sequence semantics)."""
  B, T, In = x.shape
  H = wh.shape[1]
  mod = (torch.nn.GRU if kind == "gru" else torch.nn.LSTM)(In, H, batch_first=True)
  if lens is None:
    lens = torch.full((B,), T)
  xin = _reverse_by_len(x, lens) if reverse else x
  packed = torch.nn.utils.rnn.pack_padded_sequence(xin, torch.as_tensor(lens).cpu(), batch_first=True,
                                                   enforce_sorted=False)
  out, _ = torch.func.functional_call(mod, {"weight_ih_l0": wx, "weight_hh_l0": wh,
                                            "bias_ih_l0": bx, "bias_hh_l0": bh}, (packed,))
  y, _ = torch.nn.utils.rnn.pad_packed_sequence(out, batch_first=True, total_length=T)
  return _reverse_by_len(y, lens) if reverse else y

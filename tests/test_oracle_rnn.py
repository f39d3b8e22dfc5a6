"""CPU: pin the TF-LSTMCell oracle against torch.nn.LSTM (gate permutation i,j,f,o ->
i,f,g,o, forget_bias folded into the bias) including sequence-length semantics."""
import torch

from oracle import rnn as orn


def test_lstm_tf_vs_torch():
  g = torch.Generator().manual_seed(0)
  B, T, In, H = 3, 9, 5, 4
  x = torch.randn(B, T, In, generator=g)
  lens = torch.tensor([9, 4, 6])
  wx = torch.randn(In, 4 * H, generator=g) * 0.5
  wh = torch.randn(H, 4 * H, generator=g) * 0.5
  b = torch.randn(4 * H, generator=g) * 0.1
  for reverse in (False, True):
    y = orn.lstm_tf(x, lens, wx, wh, b, forget_bias=1.0, reverse=reverse)
    # permute to torch order (i, f, g, o) and fold forget_bias
    def perm(m):
      i, j, f, o = m.chunk(4, dim=-1)
      return torch.cat([i, f, j, o], -1)
    bt = perm(b[None])[0].clone()
    bt[H:2 * H] += 1.0
    yt = orn.cudnn_rnn("lstm", x, lens, perm(wx).t().contiguous(), perm(wh).t().contiguous(), bt,
                       torch.zeros(4 * H), reverse=reverse)
    torch.testing.assert_close(y, yt, rtol=1e-5, atol=1e-5)
    assert float(y[1, 4:].abs().max()) == 0.0

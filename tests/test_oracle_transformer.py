"""CPU: Transformer oracle pins — the reference's own known-answer tests
(parts/transformer/utils_test.py:27-61) + cross-checks vs torch.nn.functional."""
import torch
import torch.nn.functional as F

from oracle import transformer as ot


def test_padding_kats():
  x = torch.tensor([[1, 0, 0, 0, 2], [3, 4, 0, 0, 0], [0, 5, 6, 0, 7]])
  assert ot.get_padding(x).tolist() == [[0, 1, 1, 1, 0], [0, 0, 1, 1, 1], [1, 0, 0, 1, 0]]
  b = ot.get_padding_bias(x)
  assert list(b.shape) == [3, 1, 1, 5]
  N = ot.NEG_INF
  assert b.reshape(3, 5).tolist() == [[0, N, N, N, 0], [0, 0, N, N, N], [N, 0, 0, N, 0]]
  c = ot.get_decoder_self_attention_bias(5)
  assert c.tolist() == [[[[0, N, N, N, N], [0, 0, N, N, N], [0, 0, 0, N, N], [0, 0, 0, 0, N],
                          [0, 0, 0, 0, 0]]]]


def test_layernorm_and_xent_vs_torch():
  g = torch.Generator().manual_seed(0)
  x = torch.randn(4, 7, 32, generator=g)
  s, b = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g)
  torch.testing.assert_close(ot.layer_norm(x, s, b), F.layer_norm(x, (32,), s, b, 1e-6),
                             rtol=1e-5, atol=1e-5)
  logits = torch.randn(3, 5, 16, generator=g)
  labels = torch.tensor([[3, 4, 1, 0, 0], [2, 1, 0, 0, 0], [5, 6, 7, 8, 1]])
  # smoothing 0 == plain masked cross entropy normalised by the non-pad count
  ref = F.cross_entropy(logits.reshape(-1, 16), labels.reshape(-1), ignore_index=0, reduction="sum") \
      / (labels != 0).sum()
  torch.testing.assert_close(ot.padded_xent_smoothing(logits, labels, 0.0), ref, rtol=1e-5, atol=1e-5)
  # with smoothing: a perfectly "smoothed-optimal" prediction gives loss ~0
  V = 16
  soft = torch.full((1, 1, V), 0.1 / (V - 1)); soft[0, 0, 3] = 0.9
  l0 = ot.padded_xent_smoothing(torch.log(soft), torch.tensor([[3]]), 0.1)
  assert abs(float(l0)) < 1e-5


def test_position_encoding_and_embedding():
  pe = ot.get_position_encoding(10, 8)
  assert pe.shape == (10, 8) and float(pe[0, :4].abs().max()) == 0.0 and float(pe[0, 4:].min()) == 1.0
  tab = torch.arange(24, dtype=torch.float32).reshape(6, 4) + 1
  e = ot.embedding(torch.tensor([[2, 0, 7]]), tab)
  assert float(e[0, 1].abs().max()) == 0.0 and float(e[0, 2].abs().max()) == 0.0   # pad / oob
  torch.testing.assert_close(e[0, 0], tab[2] * 2.0)

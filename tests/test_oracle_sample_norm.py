"""oracle/cnn.py: layer_norm_tf / instance_norm_tf against torch's own implementations (an independent statement of
the same published definitions) and the properties the definitions imply. TensorFlow (whose contrib layers the
reference calls, conv_blocks.py:262-264, 298-301) is not installable here: values "parity unpinned"."""
import torch
import torch.nn.functional as F

from oracle import cnn


def test_layer_norm_tf_is_layer_norm_over_time_and_channels():
  g = torch.Generator().manual_seed(0)
  y = torch.randn(3, 17, 12, generator=g) * 3 + 1.5
  gamma, beta = torch.rand(12, generator=g) + 0.5, torch.randn(12, generator=g)
  got = cnn.layer_norm_tf(y, gamma, beta)
  want = F.layer_norm(y, (17, 12), eps=1e-12) * gamma + beta
  torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
  # a sample's T x C values have mean 0 / variance 1 before the affine part, whatever the other samples hold
  z = cnn.layer_norm_tf(y, torch.ones(12), torch.zeros(12))
  torch.testing.assert_close(z.mean(dim=(1, 2)), torch.zeros(3), atol=1e-5, rtol=0)
  torch.testing.assert_close(z.var(dim=(1, 2), unbiased=False), torch.ones(3), atol=1e-4, rtol=0)
  y2 = y.clone()
  y2[1] += 100.0
  torch.testing.assert_close(cnn.layer_norm_tf(y2, gamma, beta)[0], got[0])


def test_instance_norm_tf_is_instance_norm_over_time():
  g = torch.Generator().manual_seed(1)
  y = torch.randn(4, 23, 10, generator=g) * 2 - 0.7
  gamma, beta = torch.rand(10, generator=g) + 0.5, torch.randn(10, generator=g)
  got = cnn.instance_norm_tf(y, gamma, beta)
  want = F.instance_norm(y.permute(0, 2, 1), eps=1e-6).permute(0, 2, 1) * gamma + beta      # torch wants [B, C, T]
  torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
  z = cnn.instance_norm_tf(y, torch.ones(10), torch.zeros(10))
  torch.testing.assert_close(z.mean(dim=1), torch.zeros(4, 10), atol=1e-5, rtol=0)
  # padded (zero) frames are part of the statistics, as in the reference: appending zeros changes the output
  yp = torch.cat([y, torch.zeros(4, 5, 10)], dim=1)
  assert not torch.allclose(cnn.instance_norm_tf(yp, gamma, beta)[:, :23], got, atol=1e-3)

"""CPU: pin the conv oracle (torch) against an independent direct NumPy loop and
the TF 'SAME' padding rule."""
import numpy as np
import pytest
import torch

from oracle import cnn


def test_same_padding_rule():
  # Jasper layer 1: K=11, s=2, T even -> 4 left / 5 right
  assert cnn.same_pad(1680, 11, 2, 1) == (840, 4, 5)
  assert cnn.same_pad(160, 41, 2, 1) == (80, 19, 20)  # DS2 conv1 freq axis
  assert cnn.same_pad(100, 29, 1, 2) == (100, 28, 28)
  assert cnn.same_pad(7, 3, 2, 1) == (4, 1, 1)


@pytest.mark.parametrize("T,K,s,d,pad", [(37, 11, 2, 1, "SAME"), (50, 5, 1, 2, "SAME"),
                                         (33, 7, 1, 1, "VALID"), (16, 1, 1, 1, "SAME"),
                                         (21, 4, 3, 1, "SAME")])
def test_conv_vs_direct(T, K, s, d, pad):
  rng = np.random.RandomState(0)
  x = rng.randn(2, T, 6).astype(np.float32)
  w = rng.randn(K, 6, 5).astype(np.float32)
  a = cnn.conv1d_tf(x, w, s, d, pad).numpy()
  b = cnn.conv1d_direct_numpy(x, w, s, d, pad)
  assert a.shape == b.shape
  np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-4)

"""CPU: the two CTC-loss oracles (direct NumPy recursion, torch) agree, incl. the
TF 'ignore_longer_outputs_than_inputs' rule with adjacent repeats."""
import numpy as np
import torch

from oracle import ctc


def test_numpy_vs_torch():
  rng = np.random.RandomState(0)
  T, B, V, L = 30, 6, 7, 9
  logits = rng.randn(T, B, V).astype(np.float32) * 2
  labels = rng.randint(0, V - 1, size=(B, L)).astype(np.int32)
  labels[1, :4] = [2, 2, 2, 3]           # adjacent repeats
  label_len = np.array([9, 4, 0, 5, 9, 1], np.int32)
  in_len = np.array([30, 6, 10, 5, 9, 1], np.int32)   # sample 1: 4+2 repeats = 6 ok
  a = ctc.ctc_loss_numpy(logits, in_len, labels, label_len)
  b, mean, _ = ctc.ctc_loss_torch(logits, in_len, labels, label_len)
  np.testing.assert_allclose(a, b.numpy(), rtol=1e-4, atol=1e-4)
  assert abs(float(mean) - a.mean()) < 1e-4


def test_infeasible_is_zero():
  rng = np.random.RandomState(1)
  logits = rng.randn(5, 2, 4).astype(np.float32)
  labels = np.array([[0, 0, 0, 1], [0, 1, 2, 0]], np.int32)
  # sample 0 needs 4 + 2 repeats = 6 > 5 frames -> ignored; sample 1 fits in 5? 4 <= 5 yes
  a = ctc.ctc_loss_numpy(logits, [5, 5], labels, [4, 4])
  b, _, g = ctc.ctc_loss_torch(logits, [5, 5], labels, [4, 4], want_grad=True)
  assert a[0] == 0.0 and b[0] == 0.0 and a[1] > 0
  assert float(g[:, 0].abs().max()) == 0.0

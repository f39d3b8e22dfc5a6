"""GPU parity: os2s_ctc_loss (loss + dlogits) vs the CPU oracle.
fp32 log-space recursions; tolerance rtol 1e-3 / atol 1e-3 on the loss (values
~1e2) and atol 2e-4 on the per-logit gradients (|g| <= 1)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ctc  # noqa: E402


def _case(seed, T, B, V, L, peaky=1.0):
  rng = np.random.RandomState(seed)
  logits = (rng.randn(T, B, V) * peaky).astype(np.float32)
  labels = rng.randint(0, V - 1, size=(B, max(L, 1))).astype(np.int32)
  label_len = rng.randint(0, L + 1, size=B).astype(np.int32)
  in_len = rng.randint(max(T // 2, 1), T + 1, size=B).astype(np.int32)
  in_len[0] = T
  if B > 2:
    labels[1, :3] = labels[1, 0]       # adjacent repeats
    label_len[2] = L; in_len[2] = max(L - 1, 1)   # infeasible -> ignored
  return logits, in_len, labels, label_len


@pytest.mark.parametrize("T,B,V,L,peaky", [(20, 4, 5, 6, 1.0), (64, 8, 29, 20, 2.0),
                                           (257, 5, 29, 100, 3.0), (840, 4, 29, 400, 1.0),
                                           (33, 3, 40, 1, 1.0), (50, 2, 29, 0, 1.0)])
def test_ctc_vs_oracle(cuda, T, B, V, L, peaky):
  from openseq2seq_amd import capi
  logits, in_len, labels, label_len = _case(T + B, T, B, V, L, peaky)
  r_loss, r_mean, r_grad = ctc.ctc_loss_torch(logits, in_len, labels, label_len, want_grad=True)
  d = cuda
  out = capi.ctc_loss(torch.from_numpy(logits).to(d), torch.from_numpy(in_len).to(d),
                      torch.from_numpy(labels).to(d), torch.from_numpy(label_len).to(d),
                      grad_scale=1.0, want_grad=True, want_grad_bf16=True,
                      vpad=((V + 7) // 8) * 8 if V > 32 else 32)
  torch.cuda.synchronize()
  torch.testing.assert_close(out["loss_per_sample"].cpu(), r_loss, rtol=1e-3, atol=1e-3)
  torch.testing.assert_close(out["loss_mean"].cpu()[0], r_mean, rtol=1e-3, atol=1e-3)
  torch.testing.assert_close(out["dlogits"].cpu(), r_grad, rtol=1e-3, atol=2e-4)
  g16 = out["dlogits_bf16"].float().cpu()      # [B,T,32]
  torch.testing.assert_close(g16[:, :, :V], r_grad.permute(1, 0, 2), rtol=1e-2, atol=4e-3)
  if g16.shape[2] > V:
    assert float(g16[:, :, V:].abs().max()) == 0.0


def test_ctc_small_numpy_pin(cuda):
  """Also pin directly against the float64 NumPy recursion on a small case."""
  from openseq2seq_amd import capi
  logits, in_len, labels, label_len = _case(7, 25, 6, 6, 8)
  ref = ctc.ctc_loss_numpy(logits, in_len, labels, label_len)
  out = capi.ctc_loss(torch.from_numpy(logits).to(cuda), torch.from_numpy(in_len).to(cuda),
                      torch.from_numpy(labels).to(cuda), torch.from_numpy(label_len).to(cuda),
                      want_grad=False)
  np.testing.assert_allclose(out["loss_per_sample"].cpu().numpy(), ref, rtol=1e-4, atol=1e-3)


def test_ctc_gradient_rows_sum_to_zero(cuda):
  """Property at BASELINE size: d/dlogits of a softmax-composed loss sums to 0
  over classes for every live frame; dead frames are exactly 0."""
  from openseq2seq_amd import capi
  T, B, V, L = 840, 32, 29, 200
  logits, in_len, labels, label_len = _case(1, T, B, V, L)
  label_len[:] = np.minimum(label_len, in_len // 3)
  out = capi.ctc_loss(torch.from_numpy(logits).to(cuda), torch.from_numpy(in_len).to(cuda),
                      torch.from_numpy(labels).to(cuda), torch.from_numpy(label_len).to(cuda))
  g = out["dlogits"].cpu()
  assert float(g.sum(-1).abs().max()) < 5e-4  # fp32 sum of 29 terms + fast exp
  for b in range(B):
    assert float(g[in_len[b]:, b].abs().max() if in_len[b] < T else 0.0) == 0.0
  assert torch.isfinite(out["loss_mean"]).all()

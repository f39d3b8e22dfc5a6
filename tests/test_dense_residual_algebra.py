"""The dense-residual algebra of csrc/dense_residual.hip / parts/cnns/dense_residual.py in float64 against autograd
(CPU; no device code): a BatchNorm'd 1x1 branch y = BN(r W^T) of a block end (conv_bn_res_bn_actv,
parts/cnns/conv_blocks.py:61-168 with residual_dense, encoders/tdnn_encoder.py:188-192) evaluated WITHOUT the branch
tensor z = r W^T — batch statistics from the column sums s = 1^T r and the Gram matrix G = r^T r of the block input,
the output as one product with the BN-scaled kernel, every gradient from P = dz^T r:

  m = s / N, C = G / N - m m^T, mu = W m, var = diag(W C W^T), rstd = (var + eps)^-1/2
  y = r (diag(gamma rstd) W)^T + (beta - gamma rstd mu)
  dbeta = 1^T dz, dgamma = rstd (rowsum(P * W) - mu dbeta)
  d1 = gamma rstd, d2 = d1 rstd dgamma / N
  dW = d1 P - (d1 dbeta / N) s^T - N d2 (W C)
  dr = dz (d1 W) - r (W^T diag(d2) W) + 1 (mu^T diag(d2) W - (d1 dbeta / N)^T W)

Several branches over sources of different widths sum into one block end, as in Jasper 10x5; the device path
concatenates them along the reduction dimension (one GEMM per block end), which is this sum."""
import torch


def _branch_autograd(r, W, gamma, beta, eps):
  z = r @ W.t()
  mu = z.mean(0)
  var = z.var(0, unbiased=False)
  return gamma * (z - mu) / torch.sqrt(var + eps) + beta


def _branch_algebra(r, W, gamma, beta, eps, dz):
  N = r.shape[0]
  s = r.sum(0)
  G = r.t() @ r
  m = s / N
  C = G / N - torch.outer(m, m)
  mu = W @ m
  T = W @ C
  var = (T * W).sum(1)
  rstd = 1.0 / torch.sqrt(var + eps)
  scale = gamma * rstd
  y = r @ (scale[:, None] * W).t() + (beta - scale * mu)
  P = dz.t() @ r
  dbeta = dz.sum(0)
  dgamma = rstd * ((P * W).sum(1) - mu * dbeta)
  d1 = scale
  d2 = d1 * rstd * dgamma / N
  dW = d1[:, None] * P - torch.outer(d1 * dbeta / N, s) - N * d2[:, None] * T
  dr = dz @ (d1[:, None] * W) - r @ (W.t() @ (d2[:, None] * W)) + ((mu * d2) @ W - (d1 * dbeta / N) @ W)[None, :]
  return y, dr, dW, dgamma, dbeta


def test_dense_residual_algebra_matches_autograd_fp64():
  g = torch.Generator().manual_seed(0)
  N, cout, eps = 700, 48, 1e-3
  widths = [24, 40, 64]                                    # three block inputs feed this block end
  rs = [torch.randn(N, c, generator=g, dtype=torch.float64).relu().requires_grad_(True) for c in widths]
  Ws = [(torch.randn(cout, c, generator=g, dtype=torch.float64) * 0.2).requires_grad_(True) for c in widths]
  gammas = [(torch.rand(cout, generator=g, dtype=torch.float64) + 0.5).requires_grad_(True) for _ in widths]
  betas = [torch.randn(cout, generator=g, dtype=torch.float64).requires_grad_(True) for _ in widths]
  dz = torch.randn(N, cout, generator=g, dtype=torch.float64)
  R = sum(_branch_autograd(r, W, ga, be, eps) for r, W, ga, be in zip(rs, Ws, gammas, betas))
  R.backward(dz)
  R_alg = torch.zeros_like(R)
  for r, W, ga, be in zip(rs, Ws, gammas, betas):
    y, dr, dW, dga, dbe = _branch_algebra(r.detach(), W.detach(), ga.detach(), be.detach(), eps, dz)
    R_alg = R_alg + y
    for name, got, ref in (("dr", dr, r.grad), ("dW", dW, W.grad), ("dgamma", dga, ga.grad), ("dbeta", dbe, be.grad)):
      err = float((got - ref).abs().max() / ref.abs().max())
      assert err < 1e-10, (name, err)
  assert float((R_alg - R.detach()).abs().max() / R.detach().abs().max()) < 1e-12

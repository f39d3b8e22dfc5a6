"""GPU parity of the Transformer kernels vs the CPU fp32 oracle (torch autograd for
gradients). bf16 I/O: rtol/atol 2e-2 relative to the tensor rms unless noted."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import transformer as ot  # noqa: E402


def _bf(x):
  return x.to(torch.bfloat16)


def _close(got, ref, tol=2e-2):
  got = got.float().cpu()
  scale = float(ref.detach().pow(2).mean().sqrt()) + 1e-8
  torch.testing.assert_close(got, ref.detach(), rtol=tol, atol=tol * scale)


def _pack(x_padded, lens):
  return torch.cat([x_padded[b, :lens[b]] for b in range(len(lens))], 0)


def _cu(lens, dev):
  return torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=dev)


@pytest.mark.parametrize("causal,cross", [(False, False), (True, False), (False, True)])
@pytest.mark.parametrize("keep", [1.0, 0.9])
@pytest.mark.parametrize("lens", [([64, 17, 1, 33, 56], [40, 64, 9, 2, 56]),
                                  ([32, 16, 48, 31, 49], [16, 32, 33, 64, 15])])
def test_attention_fwd_bwd(cuda, causal, cross, keep, lens):
  """Lengths on both sides of the 16- and 32-row block edges: the kernels skip whole 32 x 32 blocks
  (and 16-row reduction steps) past a sequence's length or above the causal diagonal; outputs are
  allocated with torch.empty so a skipped store shows."""
  from openseq2seq_amd import capi
  g = torch.Generator().manual_seed(7 + causal + 2 * cross)
  B, H, dh = 5, 3, 64
  D = H * dh
  lq = lens[0]
  lk = lens[1] if cross else lq
  q = _bf(torch.randn(sum(lq), D, generator=g))
  k = _bf(torch.randn(sum(lk), D, generator=g))
  v = _bf(torch.randn(sum(lk), D, generator=g))
  do = _bf(torch.randn(sum(lq), D, generator=g))
  scale = dh ** -0.5
  d = cuda
  cq, ck = _cu(lq, d), _cu(lk, d)
  seed = 99
  o, lse = capi.attention_fwd(q.to(d), k.to(d), v.to(d), cq, ck, H, 64, causal, scale, keep, seed)
  dq, dk, dv = (torch.empty(sum(lq), D, dtype=torch.bfloat16, device=d),
                torch.empty(sum(lk), D, dtype=torch.bfloat16, device=d),
                torch.empty(sum(lk), D, dtype=torch.bfloat16, device=d))
  capi.attention_bwd(q.to(d), k.to(d), v.to(d), do.to(d), lse, dq, dk, dv, cq, ck, H, 64, causal,
                     scale, keep, seed)
  torch.cuda.synchronize()
  # oracle per sequence (packed layout has no padding)
  qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
  outs = []
  oq = ok = 0
  for b in range(B):
    Q = qf[oq:oq + lq[b]].view(lq[b], H, dh).transpose(0, 1)
    K = kf[ok:ok + lk[b]].view(lk[b], H, dh).transpose(0, 1)
    V = vf[ok:ok + lk[b]].view(lk[b], H, dh).transpose(0, 1)
    S = (Q * scale) @ K.transpose(-1, -2)
    if causal:
      S = S + ot.get_decoder_self_attention_bias(lq[b])[0, 0][:, :lk[b]]
    P = torch.softmax(S, -1)
    lse_ref = torch.logsumexp(S.detach(), -1).transpose(0, 1)         # [lq, H]
    torch.testing.assert_close(lse[oq:oq + lq[b]].cpu(), lse_ref, rtol=2e-3, atol=2e-3)
    if keep < 1.0:
      # the device mask: element ((b*H+h)*64 + q)*64 + key
      n = B * H * 64 * 64
      m = capi.dropout_mask(seed, n, keep, d).cpu().view(B, H, 64, 64)[b, :, :lq[b], :lk[b]]
      P = P * m.float() / keep
    outs.append((P @ V).transpose(0, 1).reshape(lq[b], D))
    oq += lq[b]; ok += lk[b]
  ref = torch.cat(outs, 0)
  ref.backward(do.float())
  _close(o, ref)
  _close(dq, qf.grad, 3e-2)
  _close(dk, kf.grad, 3e-2)
  _close(dv, vf.grad, 3e-2)


def test_layernorm_fwd_bwd(cuda):
  from openseq2seq_amd import capi
  g = torch.Generator().manual_seed(1)
  for D in (1024, 512):
    N = 77
    x = _bf(torch.randn(N, D, generator=g) * 2 + 0.5)
    gam, bet = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g) * 0.1
    dy = _bf(torch.randn(N, D, generator=g))
    dres = _bf(torch.randn(N, D, generator=g))
    y, mean, rstd = capi.layernorm_fwd(x.to(cuda), gam.to(cuda), bet.to(cuda))
    dgam, dbet = torch.zeros(D, device=cuda), torch.zeros(D, device=cuda)
    dx = capi.layernorm_bwd(dy.to(cuda), x.to(cuda), gam.to(cuda), mean, rstd, dres.to(cuda), dgam, dbet)
    torch.cuda.synchronize()
    xf, gf, bf = x.float().requires_grad_(True), gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    ref = ot.layer_norm(xf, gf, bf)
    ref.backward(dy.float())
    _close(y, ref)
    _close(dx, xf.grad + dres.float())
    _close(dgam, gf.grad)
    _close(dbet, bf.grad)


def test_embedding_fwd_bwd(cuda):
  from openseq2seq_amd import capi
  g = torch.Generator().manual_seed(2)
  V, D, N = 200, 512, 150
  table = _bf(torch.randn(V, D, generator=g) * D ** -0.5)
  ids = torch.randint(0, V + 5, (N,), generator=g).to(torch.int32)   # some pad (0) and oob ids
  ids[:3] = 0
  pos = torch.randint(0, 56, (N,), generator=g).to(torch.int32)
  keep, seed = 0.7, 5
  out = capi.embed_fwd(ids.to(cuda), pos.to(cuda), table.to(cuda), D ** 0.5, keep, seed)
  mask = capi.dropout_mask(seed, N * D, keep, cuda).cpu().view(N, D)
  tf = table.float().requires_grad_(True)
  ref = ot.embedding(ids.long()[None], tf)[0] + ot.get_position_encoding(56, D)[pos.long()]
  ref = ref * mask.float() / keep
  _close(out, ref)
  dout = _bf(torch.randn(N, D, generator=g))
  ref.backward(dout.float())
  dt = torch.zeros(V, D, device=cuda)
  capi.embed_bwd(ids.to(cuda), dout.to(cuda), dt, D ** 0.5, keep, seed)
  torch.cuda.synchronize()
  torch.testing.assert_close(dt.cpu(), tf.grad, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("V", [32768, 1000, 8])
def test_xent_smooth(cuda, V):
  from openseq2seq_amd import capi
  g = torch.Generator().manual_seed(3)
  N = 37
  logits = _bf(torch.randn(N, V, generator=g) * 3)
  labels = torch.randint(1, V, (N,), generator=g).to(torch.int32)
  rl, mean, dl = capi.xent_smooth(logits.to(cuda), labels.to(cuda), 0.1)
  torch.cuda.synchronize()
  lf = logits.float().requires_grad_(True)
  ref = ot.padded_xent_smoothing(lf[None], labels[None], 0.1)
  ref.backward()
  torch.testing.assert_close(mean.cpu()[0], ref.detach(), rtol=2e-3, atol=2e-3)
  # dlogits are O(1/N * softmax): compare on that scale
  gref = lf.grad
  err = (dl.float().cpu() - gref).abs().max()
  assert float(err) < 2e-2 * float(gref.abs().max()) + 1e-6


def test_gemm_epilogue_relu_dropout_residual(cuda):
  from openseq2seq_amd import capi
  g = torch.Generator().manual_seed(4)
  N, Cin, Cout = 300, 256, 384
  x = _bf(torch.randn(N, Cin, generator=g))
  w = _bf(torch.randn(Cout, Cin, generator=g) * Cin ** -0.5)
  bias = torch.randn(Cout, generator=g)
  res = _bf(torch.randn(N, Cout, generator=g))
  keep, seed = 0.7, 21
  y = capi.gemm(x.to(cuda), w.to(cuda), bias=bias.to(cuda), act=1, keep_prob=keep, seed=seed,
                residual=res.to(cuda))
  mask = capi.dropout_mask(seed, N * Cout, keep, cuda).cpu().view(N, Cout)
  ref = res.float() + torch.relu(x.float() @ w.float().t() + bias) * mask.float() / keep
  _close(y, ref)
  # backward helpers
  h = capi.gemm(x.to(cuda), w.to(cuda), bias=bias.to(cuda), act=1, keep_prob=keep, seed=seed)
  dh = _bf(torch.randn(N, Cout, generator=g))
  d1 = capi.dropout_bwd(dh.to(cuda), keep, out=h)
  hr = torch.relu(x.float() @ w.float().t() + bias) * mask.float() / keep
  _close(d1, dh.float() * (hr > 0).float() / keep)
  d0 = capi.dropout_bwd(dh.to(cuda), keep, seed=seed)
  _close(d0, dh.float() * mask.float() / keep)


@pytest.mark.parametrize("causal,cross", [(False, False), (True, False), (False, True)])
def test_attention_fwd_long(cuda, causal, cross):
  """Multi-tile forward (sequences > 64 tokens, inference): same fp32 reference, atol 2e-2."""
  from openseq2seq_amd import capi
  g = torch.Generator().manual_seed(11 + causal + 2 * cross)
  B, H, dh = 5, 3, 64
  D = H * dh
  lq = [200, 17, 65, 128, 1]
  lk = [70, 300, 9, 64, 129] if cross else lq
  q = _bf(torch.randn(sum(lq), D, generator=g))
  k = _bf(torch.randn(sum(lk), D, generator=g))
  v = _bf(torch.randn(sum(lk), D, generator=g))
  scale = dh ** -0.5
  o, lse = capi.attention_fwd(q.to(cuda), k.to(cuda), v.to(cuda), _cu(lq, cuda), _cu(lk, cuda), H,
                              max(max(lq), max(lk)), causal, scale)
  oq = ok = 0
  for b in range(B):
    Q = q[oq:oq + lq[b]].float().view(lq[b], H, dh).transpose(0, 1)
    K = k[ok:ok + lk[b]].float().view(lk[b], H, dh).transpose(0, 1)
    V = v[ok:ok + lk[b]].float().view(lk[b], H, dh).transpose(0, 1)
    S = (Q * scale) @ K.transpose(-1, -2)
    if causal:
      S = S + torch.triu(torch.full((lq[b], lk[b]), -1e9), diagonal=1)
    ref = (torch.softmax(S, -1) @ V).transpose(0, 1).reshape(lq[b], D)
    torch.testing.assert_close(o[oq:oq + lq[b]].float().cpu(), ref, atol=2e-2, rtol=2e-2)
    ref_lse = torch.logsumexp(S, -1).transpose(0, 1)
    torch.testing.assert_close(lse[oq:oq + lq[b]].cpu(), ref_lse, atol=2e-3, rtol=1e-3)
    oq += lq[b]; ok += lk[b]


@pytest.mark.parametrize("M,N,K", [(300, 1024, 512), (4096, 3072, 1024), (8300, 1024, 4096), (777, 520, 192)])
def test_bare_matmuls_in_tree(cuda, M, N, K):
  """The three roles of a Dense layer as bare matmuls on the in-tree kernels (capi.gemm ->
  os2s_gemm_nt, capi.gemm_wgrad -> conv1d_wgrad: ping-pong 256 x 256 tiles, lockstep below
  their thresholds) vs fp32 references on the same bf16 inputs: forward x W^T, data gradient
  dz W (+ accumulate into an existing bf16 buffer), weight gradient dW += dy^T x (fp32 out), a
  column-slice view as input. bf16 outputs: rtol 1e-2; fp32 output: 2e-3 of the rms."""
  from openseq2seq_amd import capi
  g = torch.Generator().manual_seed(M + N + K)
  x = _bf(torch.randn(M, K, generator=g)).to(cuda)
  w = _bf(torch.randn(N, K, generator=g) / K ** 0.5).to(cuda)
  dz = _bf(torch.randn(M, N, generator=g)).to(cuda)
  y = capi.gemm(x, w)
  ref = x.float() @ w.float().t()
  torch.testing.assert_close(y.float(), ref, rtol=1e-2, atol=2e-2)
  wt = w.t().contiguous()
  dx = capi.gemm(dz, wt)
  ref = dz.float() @ w.float()
  torch.testing.assert_close(dx.float(), ref, rtol=1e-2, atol=2e-2 * float(ref.std()))
  dx2 = capi.gemm(dz, wt, out=dx.clone(), accumulate=True)
  torch.testing.assert_close(dx2.float(), 2 * ref, rtol=2e-2, atol=4e-2 * float(ref.std()))
  base = torch.randn(N, K, generator=g).to(cuda)
  dw = base.clone()
  capi.gemm_wgrad(x, dz, dw, accumulate=True)
  ref = base + dz.float().t() @ x.float()
  torch.testing.assert_close(dw, ref, rtol=2e-3, atol=2e-3 * float(ref.std()))
  wide = _bf(torch.randn(M, 2 * K, generator=g)).to(cuda)
  y2 = capi.gemm(wide[:, K:], w)
  torch.testing.assert_close(y2.float(), wide[:, K:].float() @ w.float().t(), rtol=1e-2, atol=2e-2)
  dw2 = torch.zeros(N, K, device=cuda)
  capi.gemm_wgrad(wide[:, K:], dz, dw2, accumulate=False)
  ref = dz.float().t() @ wide[:, K:].float()
  torch.testing.assert_close(dw2, ref, rtol=2e-3, atol=2e-3 * float(ref.std()))


def test_dense_epilogue_matches_fused_gemm(cuda):
  """bare os2s_gemm_nt + os2s_dense_epilogue == the fused in-tree GEMM epilogue (bias, ReLU, dropout
  with the same (seed, element) stream, residual): identical dropout pattern, values to bf16
  rounding of the intermediate (atol 3e-2)."""
  from openseq2seq_amd import capi
  g = torch.Generator().manual_seed(3)
  M, K, N = 500, 256, 1024
  x = _bf(torch.randn(M, K, generator=g)).to(cuda)
  w = _bf(torch.randn(N, K, generator=g) / K ** 0.5).to(cuda)
  b = torch.randn(N, generator=g).to(cuda)
  r = _bf(torch.randn(M, N, generator=g)).to(cuda)
  for act, keep, res in ((1, 0.7, None), (0, 0.9, r), (0, 1.0, r), (1, 1.0, None)):
    fused = capi.gemm(x, w, bias=b, act=act, keep_prob=keep, seed=11, residual=res)
    y = capi.gemm_nt(x, w)
    capi.dense_epilogue(y, bias=b, act=act, keep_prob=keep, seed=11, residual=res)
    torch.testing.assert_close(y.float(), fused.float(), atol=3e-2, rtol=2e-2)
    if keep < 1.0 and act == 0:      # dropped elements are exactly the residual (or 0)
      base = res.float() if res is not None else torch.zeros_like(y, dtype=torch.float32)
      # (a kept element can round onto the residual in one path only: allow 0.1 % disagreement)
      assert float(((y.float() == base) != (fused.float() == base)).float().mean()) < 1e-3
      frac = float((y.float() == base).float().mean())
      assert abs(frac - (1 - keep)) < 0.02


@pytest.mark.parametrize("rows,C", [(8300, 4096), (777, 1024), (130, 800), (64, 8)])
def test_dropout_bwd_colsum(cuda, rows, C):
  """os2s_dropout_bwd_colsum == os2s_dropout_bwd (bit-identical d, both modes) and its partial
  column sums reduce to the fp32 column sums of the ROUNDED d (what the separate os2s_bn_stats
  pass computed): rtol 1e-5 of the column's absolute sum."""
  from openseq2seq_amd import capi
  g = torch.Generator().manual_seed(rows + C)
  dy = _bf(torch.randn(rows, C, generator=g)).to(cuda)
  y = _bf(torch.relu(torch.randn(rows, C, generator=g))).to(cuda)
  for kw in (dict(seed=17), dict(out=y)):
    ref = capi.dropout_bwd(dy, 0.8, **kw)
    d, part = capi.dropout_bwd_colsum(dy, 0.8, **kw)
    assert torch.equal(d, ref)
    acc = torch.full((C,), 3.0, device=cuda)
    scratch = torch.empty(2, C, device=cuda)
    capi.bn_bwd_finalize(part, 1, 1, None, acc, True, scratch[0], scratch[1])
    want = 3.0 + ref.float().sum(0)
    tol = 1e-5 * ref.float().abs().sum(0) + 1e-4
    assert bool(((acc - want).abs() <= tol).all())

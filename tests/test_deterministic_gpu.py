"""Deterministic mode (os2s_set_deterministic / OS2S_DETERMINISTIC): every kernel that accumulates a
parameter gradient with fp32 atomics from several workgroups is run twice on the same inputs and must
give BIT-IDENTICAL results; its result must also agree with the default (atomic) launch geometry to
fp32 summation-order accuracy. Covered: narrow / K = 1 / stride-2 conv weight gradients (lockstep
kernel with a reduction split), the grouped K = 1 weight gradients, depthwise weight gradients
(register-window and generic kernels), the embedding gradient with repeated ids, the style-token
attention gradient, and the location-sensitive attention backward of the Tacotron2 decoder (per-stream
slabs instead of LDS float atomics — deterministic in either mode)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def det(cuda):
  from openseq2seq_amd import capi
  before = capi.deterministic()
  yield capi
  capi.set_deterministic(before)


def _twice(fn):
  a = fn()
  b = fn()
  torch.cuda.synchronize()
  return a, b


def _bits_equal(a, b):
  return torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a, b.view(torch.int32) if b.dtype == torch.float32 else b)


def test_conv_weight_gradients_bit_identical(det, cuda):
  capi = det
  g = torch.Generator().manual_seed(0)
  for (B, T, Cin, Cout, K, stride) in [(8, 700, 64, 128, 3, 1), (6, 900, 256, 256, 1, 1), (8, 640, 64, 256, 11, 2)]:
    x = torch.randn(B, T, Cin, generator=g).to(torch.bfloat16).to(cuda)
    tout = (T + stride - 1) // stride
    dy = torch.randn(B, tout, Cout, generator=g).to(torch.bfloat16).to(cuda)

    def run():
      dw = torch.zeros(K, Cout, Cin, dtype=torch.float32, device=cuda)
      capi.conv1d_wgrad(x, dy, K, stride=stride, out=dw, accumulate=True)
      return dw
    capi.set_deterministic(True)
    a, b = _twice(run)
    assert _bits_equal(a, b), (B, T, Cin, Cout, K, stride)
    capi.set_deterministic(False)
    c = run()
    torch.testing.assert_close(a, c, rtol=2e-4, atol=2e-3)


def test_depthwise_weight_gradients_bit_identical(det, cuda):
  capi = det
  g = torch.Generator().manual_seed(1)
  for (B, T, C, K, stride, dil) in [(8, 600, 256, 33, 1, 1), (4, 500, 128, 51, 1, 2), (6, 520, 64, 33, 2, 1)]:
    x = torch.randn(B, T, C, generator=g).to(torch.bfloat16).to(cuda)
    tout = (T + stride - 1) // stride
    dy = torch.randn(B, tout, C, generator=g).to(torch.bfloat16).to(cuda)
    lens = torch.tensor([T - 37 * i for i in range(B)], dtype=torch.int32, device=cuda)

    def run():
      dw = torch.zeros(K, C, dtype=torch.float32, device=cuda)
      capi.depthwise_conv1d_wgrad(x, dy, dw, stride=stride, dil=dil, in_len=lens)
      return dw
    capi.set_deterministic(True)
    a, b = _twice(run)
    assert _bits_equal(a, b), (B, T, C, K, stride, dil)
    capi.set_deterministic(False)
    c = run()
    torch.testing.assert_close(a, c, rtol=2e-4, atol=2e-3)


def test_embedding_gradient_with_repeated_ids_bit_identical(det, cuda):
  capi = det
  g = torch.Generator().manual_seed(2)
  N, V, D = 3000, 50, 512                        # 50 rows for 3000 tokens: every row is hit ~60 times
  ids = torch.randint(0, V, (N,), generator=g).to(torch.int32).to(cuda)
  dout = torch.randn(N, D, generator=g).to(torch.bfloat16).to(cuda)

  def run():
    dt = torch.zeros(V, D, dtype=torch.float32, device=cuda)
    capi.embed_bwd(ids, dout, dt, D ** 0.5, 0.9, 7)
    return dt
  capi.set_deterministic(True)
  a, b = _twice(run)
  assert _bits_equal(a, b)
  capi.set_deterministic(False)
  c = run()
  torch.testing.assert_close(a, c, rtol=1e-4, atol=1e-2)


def test_style_token_attention_gradient_bit_identical(det, cuda):
  capi = det
  g = torch.Generator().manual_seed(3)
  B, heads, N = 32, 8, 10
  D = heads * 64
  q = torch.randn(B, D, generator=g).to(torch.bfloat16).to(cuda)
  k = torch.randn(N, D, generator=g).to(torch.bfloat16).to(cuda)
  v = torch.randn(N, D, generator=g).to(torch.bfloat16).to(cuda)
  att_v = torch.randn(64, generator=g).to(cuda)
  dout = torch.randn(B, D, generator=g).to(torch.bfloat16).to(cuda)
  out, w = capi.gst_attention_fwd(q, k, v, att_v, heads)

  def run():
    dk = torch.zeros(N, D, dtype=torch.float32, device=cuda)
    dv = torch.zeros(N, D, dtype=torch.float32, device=cuda)
    da = torch.zeros(64, dtype=torch.float32, device=cuda)
    dq = capi.gst_attention_bwd(dout, q, k, v, att_v, w, heads, dk, dv, da)
    return torch.cat([dk.flatten(), dv.flatten(), da, dq.float().flatten()])
  capi.set_deterministic(True)
  a, b = _twice(run)
  assert _bits_equal(a, b)
  capi.set_deterministic(False)
  c = run()
  torch.testing.assert_close(a, c, rtol=1e-4, atol=1e-3)

"""fp8 (OCP e4m3) weight storage of the Tacotron2 decoder LSTM stack (BASELINE.json configs[4]:
"location-sensitive attention decode, fp8 weights"; the reference has no fp8, models/model.py:88, so
the storage / scaling policy is this library's: per-row scale = max|w| / 448, e4m3 round-to-nearest,
bf16 activations, fp32 accumulation, the scale applied to the finished dot product).

 1. os2s_quantize_rows_e4m3 is bit-exact against torch.float8_e4m3fn on the same scaled values.
 2. The decoder loop with e4m3 weights over 240 decoder steps (2 LSTM layers + location-sensitive
    attention with cumulative alignments) computes the quantised model: outputs / contexts atol
    3e-2, alignments atol 5e-3 against the fp32 oracle evaluated with the DEQUANTISED weights
    q * scale — the tolerance of the bf16-weight path (tests/test_attn_decoder_gpu.py).
 3. Against the fp32 oracle with the ORIGINAL weights — what the quantisation costs over the 240
    steps: decoder outputs (the frames the mel projection is applied to) rel-L2 <= 3e-2 (measured
    ~1e-2), alignments max error <= 4e-2.
The synthetic decoder is kept in a well-conditioned regime (recurrent weights 0.35/sqrt(K), soft
attention): with unit-scale random weights the recurrence + attention feedback is chaotic and ANY
rounding difference — bf16 activations included — decorrelates the trajectories within ~100 steps
(measured: the bf16-weight kernel ends 0.15 rel-L2 from its own oracle), which would test the
dynamics and not the kernel."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import attn_decoder as oad  # noqa: E402


def _bf(t):
  return t.to(torch.bfloat16)


def test_quantize_rows_e4m3_bit_exact(cuda):
  from openseq2seq_amd import capi
  g = torch.Generator().manual_seed(0)
  w = _bf(torch.randn(300, 520, generator=g) * torch.rand(300, 1, generator=g) * 3.0)
  w[7] = 0                                              # all-zero row: scale 1
  w[11, 5] = 1000.0                                     # one dominant value
  q, sc = capi.quantize_rows_e4m3(w.to(cuda))
  torch.cuda.synchronize()
  amax = w.float().abs().amax(1)
  ref_sc = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
  torch.testing.assert_close(sc.cpu(), ref_sc, rtol=1e-6, atol=0)
  ref_q = (w.float() * (1.0 / sc.cpu())[:, None]).clamp(-448, 448).to(torch.float8_e4m3fn)
  assert torch.equal(q.cpu().view(torch.float8_e4m3fn).float(), ref_q.float())
  # dequantisation error bound: half an e4m3 ulp = 2^-4 relative to the row maximum scale
  deq = q.cpu().view(torch.float8_e4m3fn).float() * sc.cpu()[:, None]
  assert float((deq - w.float()).abs().max() / w.float().abs().max()) < 2.0 ** -4


@pytest.mark.parametrize("size", ["h128_240_steps", "tacotron_gst_full_size"])
def test_decoder_loop_fp8_weights_240_steps(cuda, size):
  """h128_240_steps: H = M = 128 over 240 steps. tacotron_gst_full_size: the decoder cell of
  tacotron_gst.py:152-170 (H = M = 1024, U = 128, 32 x 32 location filters) over S = 200 source
  positions and 60 steps — the weight streams the fp8 path was built for (2 x 4096 x 2048 bytes)."""
  from openseq2seq_amd import capi
  B, T, S, L, H, M, U, K, F = (4, 240, 40, 2, 128, 128, 128, 32, 32) if size == "h128_240_steps" \
      else (4, 60, 200, 2, 1024, 1024, 128, 32, 32)
  g = torch.Generator().manual_seed(5)
  rn = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
  kc = [M + H, 2 * H]
  wcat = [_bf(rn(4 * H, kc[l], sc=0.35 / math.sqrt(kc[l]))) for l in range(L)]
  bias = [None, rn(4 * H, sc=0.1)]
  wq = _bf(rn(U, H, sc=1.0 / math.sqrt(H)))
  wmem = _bf(rn(U, M, sc=1.0 / math.sqrt(M)))
  v, bb = rn(U, sc=0.2), rn(U, sc=0.1)   # soft attention: a unit-scale v makes the softmax winner-take-all
  conv_w, conv_b, dense_w = rn(K, F, sc=0.5), rn(F, sc=0.1), rn(F, U, sc=0.3)
  gx0 = _bf(rn(B, T, 4 * H, sc=0.7))
  memory = _bf(rn(B, S, M))
  src_len = torch.tensor([S, (31 * S) // 40, (17 * S) // 40, (25 * S) // 40], dtype=torch.int32)
  values_h, _ = oad.prepare_memory(memory.float(), src_len)
  values = _bf(values_h)
  keys = _bf(values.float() @ wmem.float().t())
  dev = cuda
  dec = capi.AttnDecoder(B, T, S, L, H, M, U, capi.SCORE_LOCATION, dev, use_bias=True, loc_k=K, loc_f=F,
                         forget_bias=1.0, save=False)
  dec.set_params([w.to(dev) for w in wcat], wq.to(dev), v.to(dev), bias=[None, bias[1].to(dev)],
                 b=bb.to(dev), conv_w=conv_w.to(dev), conv_b=conv_b.to(dev), dense_w=dense_w.to(dev))
  w8 = [capi.quantize_rows_e4m3(w.to(dev)) for w in wcat]
  dec.set_fp8_weights(w8)
  dec.set_inputs(gx0.to(dev), keys.to(dev), values.to(dev), src_len.to(dev), None)
  dec.forward()
  torch.cuda.synchronize()
  deq = [q.cpu().view(torch.float8_e4m3fn).float() * sc.cpu()[:, None] for q, sc in w8]

  def oracle(ws):
    P = dict(wcat=ws, bias=bias, wq=wq.float(), wmem=wmem.float(), v=v, g=None, b=bb, conv_w=conv_w,
             conv_b=conv_b, dense_w=dense_w)
    with torch.no_grad():
      return oad.attention_decoder(P, gx0.float(), memory.float(), src_len, None, None, None, 1.0, "location",
                                   keys_override=keys.float(), values_override=values.float())

  y, ctx, al = dec.y_top.float().cpu(), dec.ctx.float().cpu(), dec.align_seq.cpu()
  rel = lambda a, b: float((a - b).norm() / b.norm())
  ref = oracle(deq)                                     # the quantised model, evaluated in fp32
  full = oracle([w.float() for w in wcat])              # the unquantised model
  # (a) the kernel computes the quantised model, all 240 steps, at the bf16 path's tolerance
  torch.testing.assert_close(y, ref["y"], atol=3e-2, rtol=3e-2)
  torch.testing.assert_close(ctx, ref["ctx"], atol=3e-2, rtol=3e-2)
  torch.testing.assert_close(al, ref["align"], atol=5e-3, rtol=3e-2)
  # (b) what the quantisation costs against the unquantised fp32 model over the 240 steps
  r_quant = rel(y, full["y"])
  a_quant = float((al - full["align"]).abs().max())
  print("fp8 weights, %s:" % size, " outputs rel-L2 vs quantised-model oracle %.3e, vs unquantised oracle %.3e "
        "(oracle vs oracle %.3e); max alignment error %.3e"
        % (rel(y, ref["y"]), r_quant, rel(ref["y"], full["y"]), a_quant))
  assert r_quant <= 3e-2, r_quant
  assert a_quant <= 4e-2, a_quant

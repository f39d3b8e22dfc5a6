"""GPU parity: os2s_conv1d_fwd (MFMA implicit GEMM) vs the CPU oracle.
Floating point: inputs are bf16-rounded identically on both sides, the oracle
accumulates in fp32; tolerance = bf16 output rounding (2^-8 relative) + fp32
accumulation-order noise: rtol 1e-2, atol 1e-2 * rms(y)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cnn  # noqa: E402


def _bf(x):
  return torch.as_tensor(x).to(torch.bfloat16)


def _check(y, ref, extra=1.0):
  y = y.float().cpu()
  scale = float(ref.pow(2).mean().sqrt()) + 1e-6
  torch.testing.assert_close(y, ref, rtol=1e-2 * extra, atol=1e-2 * scale * extra)


CASES = [
    # B, T, Cin, Cout, K, s, d
    (2, 200, 64, 256, 11, 2, 1),     # Jasper layer 1 (stride 2, asymmetric SAME)
    (3, 171, 256, 256, 11, 1, 1),    # Jasper B1 (ragged T vs BM=128)
    (2, 140, 768, 896, 29, 1, 2),    # dilated K=29
    (2, 130, 896, 1024, 1, 1, 1),    # 1x1
    (1, 64, 72, 200, 5, 1, 1),       # Cin tail chunk, Cout not multiple of 128
    (2, 300, 128, 128, 3, 1, 1),
]


@pytest.mark.parametrize("B,T,Cin,Cout,K,s,d", CASES)
def test_conv_fwd(cuda, B, T, Cin, Cout, K, s, d):
  from openseq2seq_amd import capi
  g = torch.Generator().manual_seed(B * 1000 + T + K)
  x = _bf(torch.randn(B, T, Cin, generator=g))
  w_tf = _bf(torch.randn(K, Cin, Cout, generator=g) * (1.0 / (K * Cin) ** 0.5))
  lens = torch.randint(T // 2, T + 1, (B,), generator=g).to(torch.int32)
  lens[0] = T
  ref = cnn.conv1d_tf(x.float(), w_tf.float(), s, d, "SAME", mask_len=lens)
  tout = ref.shape[1]
  nm = capi.conv1d_num_mtiles(B, tout)
  stats = torch.full((nm, 2, Cout), float("nan"), device=cuda)
  y = capi.conv1d_fwd(x.to(cuda), cnn.to_dev_layout(w_tf).to(cuda), stride=s, dil=d,
                      in_len=lens.to(cuda), stats=stats)
  torch.cuda.synchronize()
  assert tuple(y.shape) == (B, tout, Cout)
  _check(y, ref)
  # fused BN partial sums: computed from the bf16-ROUNDED outputs
  yr = y.float().cpu()
  s1 = stats[:, 0, :].sum(0).cpu()
  s2 = stats[:, 1, :].sum(0).cpu()
  torch.testing.assert_close(s1, yr.sum((0, 1)), rtol=1e-4, atol=1e-2)
  torch.testing.assert_close(s2, yr.pow(2).sum((0, 1)), rtol=1e-4, atol=1e-2)


def test_conv_valid_padding_and_accumulate(cuda):
  from openseq2seq_amd import capi
  g = torch.Generator().manual_seed(5)
  B, T, Cin, Cout, K = 2, 90, 64, 128, 7
  x = _bf(torch.randn(B, T, Cin, generator=g))
  w_tf = _bf(torch.randn(K, Cin, Cout, generator=g) * 0.05)
  ref = cnn.conv1d_tf(x.float(), w_tf.float(), 1, 1, "VALID")
  tout, pl = capi.valid_padding(T, K, 1, 1)
  y0 = _bf(torch.randn(B, tout, Cout, generator=g))
  y = y0.clone().to(cuda)
  capi.conv1d_fwd(x.to(cuda), cnn.to_dev_layout(w_tf).to(cuda), pad_left=pl, tout=tout,
                  out=y, accumulate=True)
  torch.cuda.synchronize()
  _check(y, ref + y0.float(), extra=2.0)


def test_fc_logits_time_major_fp32(cuda):
  """FullyConnectedTimeDecoder: dense(V=29, bias) -> time-major fp32 logits."""
  from openseq2seq_amd import capi
  g = torch.Generator().manual_seed(9)
  B, T, C, V = 3, 77, 1024, 29
  x = _bf(torch.randn(B, T, C, generator=g))
  w_tf = _bf(torch.randn(1, C, V, generator=g) * 0.03)
  bias = torch.randn(V, generator=g)
  ref = cnn.conv1d_tf(x.float(), w_tf.float()) + bias
  y = capi.conv1d_fwd(x.to(cuda), cnn.to_dev_layout(w_tf).to(cuda), bias=bias.to(cuda),
                      out_f32=True, time_major=True)
  torch.cuda.synchronize()
  assert tuple(y.shape) == (T, B, V) and y.dtype == torch.float32
  torch.testing.assert_close(y.cpu().permute(1, 0, 2), ref, rtol=1e-4, atol=1e-3)


def test_dgrad_via_flipped_weights(cuda):
  """dX of a stride-1 SAME conv == conv of dY with tap-flipped transposed weights."""
  from openseq2seq_amd import capi
  g = torch.Generator().manual_seed(11)
  B, T, Cin, Cout, K, d = 2, 150, 128, 256, 13, 1
  x = torch.randn(B, T, Cin, generator=g, requires_grad=True)
  w_tf = _bf(torch.randn(K, Cin, Cout, generator=g) * 0.02)
  dy = _bf(torch.randn(B, T, Cout, generator=g))
  y = cnn.conv1d_tf(x, w_tf.float(), 1, d, "SAME")
  y.backward(dy.float())
  _, pl = capi.same_padding(T, K, 1, d)
  # wT[k'][ci][co] = w_dev[K-1-k'][co][ci]  (w_dev = [K,Cout,Cin])
  w_dev = cnn.to_dev_layout(w_tf)
  wT = w_dev.flip(0).permute(0, 2, 1).contiguous()
  dx = capi.conv1d_fwd(dy.to(cuda), wT.to(cuda), pad_left=(K - 1) * d - pl, tout=T)
  torch.cuda.synchronize()
  _check(dx, x.grad)


WG_CASES = [
    (2, 200, 64, 256, 11, 2, 1),
    (3, 171, 256, 256, 11, 1, 1),
    (2, 140, 128, 384, 29, 1, 2),
    (2, 130, 384, 128, 1, 1, 1),
    (1, 64, 72, 200, 5, 1, 1),
    (4, 100, 1024, 32, 1, 1, 1),   # FC weight-grad shape (V padded to 32)
    (2, 150, 192, 512, 11, 1, 1),  # 256-wide output-channel tile variant
    (3, 100, 256, 768, 9, 1, 2),
]


@pytest.mark.parametrize("B,T,Cin,Cout,K,s,d", WG_CASES)
@pytest.mark.parametrize("accumulate", [False, True])
def test_conv_wgrad(cuda, B, T, Cin, Cout, K, s, d, accumulate):
  from openseq2seq_amd import capi
  g = torch.Generator().manual_seed(B * 77 + T + K)
  x = _bf(torch.randn(B, T, Cin, generator=g))
  w_tf = (torch.randn(K, Cin, Cout, generator=g) * 0.05).requires_grad_(True)
  lens = torch.randint(T // 2, T + 1, (B,), generator=g).to(torch.int32)
  lens[0] = T
  y = cnn.conv1d_tf(x.float(), w_tf, s, d, "SAME", mask_len=lens)
  dy = _bf(torch.randn(y.shape, generator=g))
  y.backward(dy.float())
  ref = cnn.to_dev_layout(w_tf.grad)  # [K,Cout,Cin]
  if accumulate:
    base = torch.randn(K, Cout, Cin, generator=g)
    out = base.clone().to(cuda)
    capi.conv1d_wgrad(x.to(cuda), dy.to(cuda), K, stride=s, dil=d, in_len=lens.to(cuda),
                      out=out, accumulate=True)
    ref = ref + base
  else:
    out = capi.conv1d_wgrad(x.to(cuda), dy.to(cuda), K, stride=s, dil=d,
                            in_len=lens.to(cuda))
  torch.cuda.synchronize()
  scale = float(ref.pow(2).mean().sqrt()) + 1e-6
  # fp32 accumulation of exact bf16 products: only summation-order noise
  torch.testing.assert_close(out.cpu(), ref, rtol=2e-3, atol=2e-3 * scale)


@pytest.mark.parametrize("variant", [0, 3, 5, 10])
@pytest.mark.parametrize("B,T,Cin,Cout,K,s,d", [
    (3, 300, 128, 768, 9, 1, 1), (2, 260, 192, 640, 5, 1, 2), (3, 200, 64, 200, 3, 1, 1),
    (2, 140, 128, 384, 7, 2, 1)])
def test_conv_fwd_tile_variants(cuda, variant, B, T, Cin, Cout, K, s, d):
  """Every tile (128x128 with double / single X buffer, 256x256 lockstep, ping-pong) FORCED on the same
  problems, incl. Cout that is not a multiple of the tile and ragged lengths; output, the fused
  ReLU/residual epilogue and the BN partial sums."""
  from openseq2seq_amd import capi, _lib
  g = torch.Generator().manual_seed(B * 1000 + T + K + Cout)
  x = _bf(torch.randn(B, T, Cin, generator=g))
  w_tf = _bf(torch.randn(K, Cin, Cout, generator=g) * (1.0 / (K * Cin) ** 0.5))
  lens = torch.randint(T // 2, T + 1, (B,), generator=g).to(torch.int32)
  lens[0] = T
  ref = cnn.conv1d_tf(x.float(), w_tf.float(), s, d, "SAME", mask_len=lens)
  tout = ref.shape[1]
  res = _bf(torch.randn(B, tout, Cout, generator=g))
  nm = capi.conv1d_num_mtiles(B, tout)
  stats = torch.full((nm, 2, Cout), float("nan"), device=cuda)
  _lib.set_option("conv1d.variant", variant)
  try:
    y = capi.conv1d_fwd(x.to(cuda), cnn.to_dev_layout(w_tf).to(cuda), stride=s, dil=d,
                        in_len=lens.to(cuda), stats=stats)
    y2 = capi.conv1d_fwd(x.to(cuda), cnn.to_dev_layout(w_tf).to(cuda), stride=s, dil=d,
                         in_len=lens.to(cuda), act=1, residual=res.to(cuda))
    torch.cuda.synchronize()
  finally:
    _lib.set_option("conv1d.variant", -1)
  _check(y, ref)
  _check(y2, torch.relu(ref) + res.float(), extra=2.0)
  yr = y.float().cpu()
  torch.testing.assert_close(stats[:, 0, :].sum(0).cpu(), yr.sum((0, 1)), rtol=1e-4, atol=1e-2)
  torch.testing.assert_close(stats[:, 1, :].sum(0).cpu(), yr.pow(2).sum((0, 1)), rtol=1e-4, atol=1e-2)


PP_CASES = [
    # B, T, Cin, Cout, K, d  — envelope of the ping-pong kernel (stride 1, Cin % 64 == 0)
    (3, 420, 256, 512, 17, 1),     # several windows per sample, odd window count per sample
    (2, 300, 320, 640, 21, 1),     # Cout = 2.5 tiles of 256, 5 chunks x 21 taps = odd step count
    (2, 700, 128, 768, 25, 1),     # long sequence
    (2, 333, 768, 896, 29, 2),     # dilated K = 29: largest X window (157 KB of LDS)
    (5, 129, 64, 256, 11, 1),      # single chunk, 2 windows per sample, second nearly empty
]


@pytest.mark.parametrize("pp", [10, 12, 13, 14])
@pytest.mark.parametrize("use_ws", [True, False])
@pytest.mark.parametrize("B,T,Cin,Cout,K,d", PP_CASES)
def test_conv_pingpong_kernel(cuda, pp, use_ws, B, T, Cin, Cout, K, d):
  """pp = option conv1d.variant: 10 = tile chosen on the device, 12 / 13 = the narrow tiles (2 / 3 live
  windows x 128 columns, conv1d_ppn_kernel; a layer they do not fit falls back to 14), 14 = 2 windows x 256
  columns.
  The ping-pong kernel (conv1d_pp_kernel) forced on Jasper-shaped layers with ragged lengths
  (dead windows, odd live-window counts), with a workspace (the tail of the launch is split over
  the input channels and reduced by the last arriver) and without: forward + BN partial sums
  (dead windows must read as zero rows / zero partials) vs the fp32 oracle, and the
  data-gradient call (tap-flipped weights, out_len skipping) vs autograd of the oracle.
  Tolerance: bf16 output rounding, rtol 1e-2 / atol 1e-2 rms."""
  from openseq2seq_amd import capi, _lib
  g = torch.Generator().manual_seed(B * 1000 + T + K + Cout)
  x = _bf(torch.randn(B, T, Cin, generator=g))
  w_tf = _bf(torch.randn(K, Cin, Cout, generator=g) * (1.0 / (K * Cin) ** 0.5))
  lens = torch.randint(T // 4, T + 1, (B,), generator=g).to(torch.int32)
  lens[0] = T
  lens[-1] = max(1, T // 5)            # whole windows of this sample are padding
  xr = x.float().clone().requires_grad_(True)
  ref = cnn.conv1d_tf(xr, w_tf.float(), 1, d, "SAME", mask_len=lens)
  dy = _bf(torch.randn(B, T, Cout, generator=g))
  mask = (torch.arange(T)[None, :] < lens[:, None]).float()[:, :, None]
  (ref * dy.float() * mask).sum().backward()     # dY is masked by the consumer (out_len rows)
  nm = capi.conv1d_num_mtiles(B, T)
  stats = torch.full((nm, 2, Cout), float("nan"), device=cuda)
  w_dev = cnn.to_dev_layout(w_tf)
  wT = w_dev.flip(0).permute(0, 2, 1).contiguous()
  _, pl = capi.same_padding(T, K, 1, d)
  dym = _bf(dy.float() * mask)
  _lib.set_option("conv1d.variant", pp)
  try:
    y = torch.full((B, T, Cout), 5.0, dtype=torch.bfloat16, device=cuda)
    capi.conv1d_fwd(x.to(cuda), w_dev.to(cuda), dil=d, in_len=lens.to(cuda), stats=stats, out=y,
                    use_workspace=use_ws)
    dx = torch.full((B, T, Cin), 7.0, dtype=torch.bfloat16, device=cuda)
    capi.conv1d_fwd(dym.to(cuda), wT.to(cuda), dil=d, pad_left=(K - 1) * d - pl, tout=T,
                    in_len=lens.to(cuda), out_len=lens.to(cuda), out=dx, use_workspace=use_ws)
    torch.cuda.synchronize()
  finally:
    _lib.set_option("conv1d.variant", -1)
  _check(y, ref.detach())
  yr = y.float().cpu()
  assert bool(torch.isfinite(stats).all())
  torch.testing.assert_close(stats[:, 0, :].sum(0).cpu(), yr.sum((0, 1)), rtol=1e-4, atol=1e-2)
  torch.testing.assert_close(stats[:, 1, :].sum(0).cpu(), yr.pow(2).sum((0, 1)), rtol=1e-4, atol=1e-2)
  if use_ws:   # the tickets are left zero
    assert int(capi.conv1d_workspace(cuda)[:4096].view(torch.int32).abs().sum()) == 0
  # data gradient: rows below out_len must match autograd (rows past out_len are don't-care:
  # their tiles may be skipped, the consumer masks them)
  dxc = dx.float().cpu()
  gx = xr.grad
  for b in range(B):
    n = int(lens[b])
    scale = float(gx[b, :n].pow(2).mean().sqrt()) + 1e-6
    torch.testing.assert_close(dxc[b, :n], gx[b, :n], rtol=1e-2, atol=1e-2 * scale)


@pytest.mark.parametrize("ragged", [False, True])
def test_conv_pingpong_full_size_vs_lockstep_tile(cuda, ragged):
  """BASELINE-size layer (Jasper B9/B10: B=32, T'=840, 768 -> 768, K=25; 336 units on 256 CUs, so
  the dense launch has one full round + a split tail, the ragged one a single partial round).
  Size-independent property instead of the (slow) CPU oracle: without a workspace the ping-pong
  kernel accumulates in the same (chunk, tap) order as the oracle-checked 128x128 tile ->
  outputs and BN partial sums are BIT-IDENTICAL; with the workspace the split tail only changes
  the fp32 summation order (bf16 outputs within 1 ulp = 2^-8 relative), and two runs agree
  bitwise (deterministic reduction)."""
  from openseq2seq_amd import capi, _lib
  g = torch.Generator().manual_seed(77)
  B, T, C, K = 32, 840, 768, 25
  x = _bf(torch.randn(B, T, C, generator=g)).to(cuda)
  w = _bf(torch.randn(K, C, C, generator=g) * (1.0 / (K * C) ** 0.5)).to(cuda)
  lens = None
  if ragged:
    lens = torch.randint(100, T + 1, (B,), generator=g).to(torch.int32)
    lens[3] = T
    lens = lens.to(cuda)
  nm = capi.conv1d_num_mtiles(B, T)
  outs = {}
  try:
    # 14 = the 256-column ping-pong tile (10 lets the device choose: a narrow tile reduces the
    # BatchNorm partials in another fp32 grouping, see test_conv_narrow_pingpong_tiles_full_size_bit_identical)
    for name, v, ws in (("tile", 3, False), ("pp", 14, False), ("pp_ws", 14, True), ("pp_ws2", 14, True)):
      _lib.set_option("conv1d.variant", v)
      st = torch.full((nm, 2, C), float("nan"), device=cuda)
      y = capi.conv1d_fwd(x, w, in_len=lens, stats=st, use_workspace=ws)
      torch.cuda.synchronize()
      outs[name] = (y, st)
  finally:
    _lib.set_option("conv1d.variant", -1)
  assert torch.equal(outs["pp"][0], outs["tile"][0])
  assert torch.equal(outs["pp"][1], outs["tile"][1])
  assert torch.equal(outs["pp_ws"][0], outs["pp_ws2"][0]) and torch.equal(outs["pp_ws"][1], outs["pp_ws2"][1])
  a, b = outs["pp_ws"][0].float(), outs["tile"][0].float()
  assert float((a - b).abs().max()) <= 2.0 ** -7 * float(b.abs().max())
  torch.testing.assert_close(outs["pp_ws"][1].sum(0), outs["tile"][1].sum(0), rtol=1e-3, atol=1e-1)


@pytest.mark.parametrize("C,K", [(384, 13), (512, 17), (640, 21), (768, 25)])
def test_conv_narrow_pingpong_tiles_full_size_bit_identical(cuda, C, K):
  """The narrow ping-pong tiles (2 / 3 live windows x 128 columns: conv1d_ppn_kernel) at the Jasper block
  shapes, B = 32, ragged: they own disjoint outputs and accumulate in the (chunk, tap) order of the
  oracle-checked 128x128 tile, so outputs are BIT-IDENTICAL to it (BatchNorm partial sums: equal up to the
  fp32 grouping of a window's rows) — for every
  live window count modulo 2 and 3 (the last window group of a launch is partly dead), for the
  148-row window image of K = 21 whose last DMA instruction is half masked, and for the layer the
  three-window tile does not fit (K = 25: falls back to the 256-column tile). Also the tile chosen on
  the device (variant 10) and the data-gradient call with out_len."""
  from openseq2seq_amd import capi, _lib
  g = torch.Generator().manual_seed(C + K)
  B, T = 32, 840
  x = _bf(torch.randn(B, T, C, generator=g)).to(cuda)
  w = _bf(torch.randn(K, C, C, generator=g) * (1.0 / (K * C) ** 0.5)).to(cuda)
  nm = capi.conv1d_num_mtiles(B, T)
  for trial in range(3):
    lens = torch.randint(100, T + 1, (B,), generator=g).to(torch.int32)
    lens[3] = T
    lens[5] = 1 + trial          # a sample of one window with a single live row
    lens = lens.to(cuda)
    outs = {}
    try:
      for name, v in (("tile", 3), ("n2", 12), ("n3", 13), ("auto", 10)):
        _lib.set_option("conv1d.variant", v)
        st = torch.full((nm, 2, C), float("nan"), device=cuda)
        y = torch.full((B, T, C), 3.0, dtype=torch.bfloat16, device=cuda)
        capi.conv1d_fwd(x, w, in_len=lens, stats=st, out=y)
        dx = torch.full((B, T, C), 7.0, dtype=torch.bfloat16, device=cuda)
        capi.conv1d_fwd(x, w, pad_left=(K - 1) // 2, tout=T, in_len=lens, out_len=lens, out=dx)
        torch.cuda.synchronize()
        outs[name] = (y, st, dx)
    finally:
      _lib.set_option("conv1d.variant", -1)
    live = (torch.arange(T, device=cuda)[None, :] < lens[:, None])[:, :, None]
    for name in ("n2", "n3", "auto"):
      if name == "auto" or (name == "n3" and K > 21):    # (K = 25: the three-window tile does not fit)
        # the device may pick the 256-column tile with a split tail: fp32 summation order of the split
        # units differs (bf16 outputs within 1 ulp = 2^-8 relative)
        a, b = outs[name][0].float(), outs["tile"][0].float()
        assert float((a - b).abs().max()) <= 2.0 ** -7 * float(b.abs().max()), (name, trial)
      else:
        assert torch.equal(outs[name][0], outs["tile"][0]), (name, trial)
      # BatchNorm partials: the narrow tiles sum a window's 128 rows in 8 row groups of 16 (512 threads
      # over 64 column pairs), the 128x128 tile in 4 groups of 32: same values, other fp32 grouping
      torch.testing.assert_close(outs[name][1], outs["tile"][1], rtol=2e-5, atol=1e-3)
      # data gradient: rows below out_len (rows past it are don't-care)
      if name == "auto" or (name == "n3" and K > 21):
        a, b = (outs[name][2] * live).float(), (outs["tile"][2] * live).float()
        assert float((a - b).abs().max()) <= 2.0 ** -7 * float(b.abs().max()), (name, trial)
      else:
        assert torch.equal(outs[name][2] * live, outs["tile"][2] * live), (name, trial)


@pytest.mark.parametrize("C,K", [(384, 13), (512, 17), (640, 21)])
def test_conv_host_length_hint_cannot_change_results(cuda, C, K):
  """os2s_conv1d_set_host_lens: with a host copy of the lengths the launcher evaluates the tile choice
  itself and enqueues one kernel. Whatever the hint says — the true lengths, lengths that are all wrong, a
  vector of the wrong size (ignored) — the output equals the oracle-checked 128x128 tile's within the 1 ulp
  a split 256-column tail may differ by, and the true hint gives the BITS of the device-side choice."""
  from openseq2seq_amd import capi, _lib
  g = torch.Generator().manual_seed(5 * C + K)
  B, T = 32, 840
  x = _bf(torch.randn(B, T, C, generator=g)).to(cuda)
  w = _bf(torch.randn(K, C, C, generator=g) * (1.0 / (K * C) ** 0.5)).to(cuda)
  lens_h = torch.randint(100, T + 1, (B,), generator=g).to(torch.int32)
  lens = lens_h.to(cuda)

  def run(variant, hint):
    _lib.set_option("conv1d.variant", variant)
    capi.conv1d_set_host_lens(hint)
    try:
      y = torch.full((B, T, C), 3.0, dtype=torch.bfloat16, device=cuda)
      capi.conv1d_fwd(x, w, in_len=lens, out=y)
      torch.cuda.synchronize()
      return y
    finally:
      capi.conv1d_set_host_lens(None)
      _lib.set_option("conv1d.variant", -1)

  ref = run(3, None)
  dev_choice = run(-1, None)
  hinted = run(-1, lens_h.tolist())
  assert torch.equal(hinted, dev_choice)
  for hint in ([T] * B, [1] * B, [T // 3] * B, lens_h.tolist()[:5]):
    y = run(-1, hint)
    assert float((y.float() - ref.float()).abs().max()) <= 2.0 ** -7 * float(ref.float().abs().max())


@pytest.mark.parametrize("B,T,Cin,Cout,K,d", [(3, 420, 256, 512, 17, 1), (2, 333, 320, 640, 21, 1),
                                              (2, 300, 768, 896, 29, 2), (4, 200, 128, 256, 2, 1),
                                              (2, 150, 192, 200, 5, 1)])
@pytest.mark.parametrize("split", [1, 3, -1])
def test_conv_wgrad_pingpong_kernel(cuda, B, T, Cin, Cout, K, d, split):
  """conv1d_wgrad_pp_kernel forced (odd tap counts, Cin / Cout tails, ragged lengths with dead
  chunks) with the reduction unsplit, cut 3 ways and by the cost model: vs autograd of the fp32
  oracle (only fp32 summation-order noise: rtol 2e-3), accumulate on top of a previous dW, and
  run-to-run bitwise reproducibility (no atomics)."""
  from openseq2seq_amd import capi, _lib
  g = torch.Generator().manual_seed(B * 77 + T + K + Cout)
  x = _bf(torch.randn(B, T, Cin, generator=g))
  w_tf = (torch.randn(K, Cin, Cout, generator=g) * 0.05).requires_grad_(True)
  lens = torch.randint(T // 4, T + 1, (B,), generator=g).to(torch.int32)
  lens[0] = T
  lens[-1] = max(1, T // 6)
  y = cnn.conv1d_tf(x.float(), w_tf, 1, d, "SAME", mask_len=lens)
  dy = _bf(torch.randn(y.shape, generator=g))
  y.backward(dy.float())
  ref = cnn.to_dev_layout(w_tf.grad)
  base = torch.randn(K, Cout, Cin, generator=g)
  L = _lib.lib()
  _lib.set_option("conv1d_wgrad.variant", 1); _lib.set_option("conv1d_wgrad.split", split)
  try:
    o1 = capi.conv1d_wgrad(x.to(cuda), dy.to(cuda), K, dil=d, in_len=lens.to(cuda))
    o2 = capi.conv1d_wgrad(x.to(cuda), dy.to(cuda), K, dil=d, in_len=lens.to(cuda))
    o3 = base.clone().to(cuda)
    capi.conv1d_wgrad(x.to(cuda), dy.to(cuda), K, dil=d, in_len=lens.to(cuda), out=o3, accumulate=True)
    torch.cuda.synchronize()
  finally:
    _lib.set_option("conv1d_wgrad.variant", -1); _lib.set_option("conv1d_wgrad.split", -1)
  scale = float(ref.pow(2).mean().sqrt()) + 1e-6
  torch.testing.assert_close(o1.cpu(), ref, rtol=2e-3, atol=2e-3 * scale)
  assert torch.equal(o1, o2)
  torch.testing.assert_close(o3.cpu(), ref + base, rtol=2e-3, atol=2e-3 * scale)
  assert int(capi.conv1d_workspace(cuda)[:4096].view(torch.int32).abs().sum()) == 0


def test_conv_wgrad_pingpong_full_size_vs_lockstep(cuda):
  """BASELINE-size layer (B=32, T'=840, 768 -> 768, K=25, ragged): unsplit, the ping-pong kernel
  adds the live 64-row chunks in the same order as the oracle-checked lockstep kernel -> dW is
  BIT-IDENTICAL; split by the cost model it differs by fp32 summation order only."""
  from openseq2seq_amd import capi, _lib
  g = torch.Generator().manual_seed(5)
  B, T, C, K = 32, 840, 768, 25
  x = _bf(torch.randn(B, T, C, generator=g)).to(cuda)
  dy = _bf(torch.randn(B, T, C, generator=g)).to(cuda)
  lens = torch.randint(100, T + 1, (B,), generator=g).to(torch.int32).to(cuda)
  L = _lib.lib()
  try:
    _lib.set_option("conv1d_wgrad.variant", 0); _lib.set_option("conv1d_wgrad.split", -1)
    ref = capi.conv1d_wgrad(x, dy, K, in_len=lens)
    _lib.set_option("conv1d_wgrad.variant", 1); _lib.set_option("conv1d_wgrad.split", 1)
    a = capi.conv1d_wgrad(x, dy, K, in_len=lens)
    _lib.set_option("conv1d_wgrad.variant", 1); _lib.set_option("conv1d_wgrad.split", 4)
    b = capi.conv1d_wgrad(x, dy, K, in_len=lens)
    torch.cuda.synchronize()
  finally:
    _lib.set_option("conv1d_wgrad.variant", -1); _lib.set_option("conv1d_wgrad.split", -1)
  assert torch.equal(a, ref)
  torch.testing.assert_close(b, ref, rtol=1e-4, atol=1e-4 * float(ref.abs().max()))


@pytest.mark.parametrize("B,T,Cin,Cout,K,d", [(3, 420, 256, 512, 17, 1), (2, 333, 384, 640, 21, 1),
                                              (2, 300, 768, 896, 29, 2), (4, 200, 128, 256, 2, 1),
                                              (2, 150, 256, 128, 5, 1), (3, 260, 128, 128, 3, 1),
                                              (2, 190, 256, 384, 13, 3), (2, 500, 128, 256, 11, 5),
                                              (1, 64, 128, 128, 4, 1), (5, 130, 256, 256, 8, 2)])
@pytest.mark.parametrize("split", [1, 3, -1])
def test_conv_wgrad_one_wave_per_simd_kernel(cuda, B, T, Cin, Cout, K, d, split):
  """conv1d_wgrad_sw_kernel forced (option conv1d_wgrad.variant 3: one wave per SIMD, 16 accumulator blocks per
  wave, hand-written instruction stream) at every class of tap count (4n: no dead tap; 4n + 1 ... 4n + 3: waves whose
  tap pair is half or wholly past K), dilation 1 ... 5 (the widest X window the 20 KB slot holds), one-step and
  many-step reductions, ragged lengths with dead chunks; reduction unsplit, cut 3 ways and by the cost model: vs
  autograd of the fp32 oracle (fp32 summation-order noise only: rtol 2e-3), accumulate on top of a previous dW,
  run-to-run bitwise reproducibility (no atomics), tickets left at zero."""
  from openseq2seq_amd import capi, _lib
  g = torch.Generator().manual_seed(B * 77 + T + K + Cout + 1)
  x = _bf(torch.randn(B, T, Cin, generator=g))
  w_tf = (torch.randn(K, Cin, Cout, generator=g) * 0.05).requires_grad_(True)
  lens = torch.randint(T // 4, T + 1, (B,), generator=g).to(torch.int32)
  lens[0] = T
  lens[-1] = max(1, T // 6)
  y = cnn.conv1d_tf(x.float(), w_tf, 1, d, "SAME", mask_len=lens)
  dy = _bf(torch.randn(y.shape, generator=g))
  y.backward(dy.float())
  ref = cnn.to_dev_layout(w_tf.grad)
  base = torch.randn(K, Cout, Cin, generator=g)
  _lib.set_option("conv1d_wgrad.variant", 3); _lib.set_option("conv1d_wgrad.split", split)
  try:
    o1 = capi.conv1d_wgrad(x.to(cuda), dy.to(cuda), K, dil=d, in_len=lens.to(cuda))
    o2 = capi.conv1d_wgrad(x.to(cuda), dy.to(cuda), K, dil=d, in_len=lens.to(cuda))
    o3 = base.clone().to(cuda)
    capi.conv1d_wgrad(x.to(cuda), dy.to(cuda), K, dil=d, in_len=lens.to(cuda), out=o3, accumulate=True)
    _lib.set_option("conv1d_wgrad.variant", 0); _lib.set_option("conv1d_wgrad.split", -1)
    lock = capi.conv1d_wgrad(x.to(cuda), dy.to(cuda), K, dil=d, in_len=lens.to(cuda))
    torch.cuda.synchronize()
  finally:
    _lib.set_option("conv1d_wgrad.variant", -1); _lib.set_option("conv1d_wgrad.split", -1)
  scale = float(ref.pow(2).mean().sqrt()) + 1e-6
  torch.testing.assert_close(o1.cpu(), ref, rtol=2e-3, atol=2e-3 * scale)
  assert torch.equal(o1, o2)
  torch.testing.assert_close(o3.cpu(), ref + base, rtol=2e-3, atol=2e-3 * scale)
  assert int(capi.conv1d_workspace(cuda)[:4096].view(torch.int32).abs().sum()) == 0
  if split == 1:      # unsplit: the live 64-row chunks are added in the order of the lockstep kernel
    assert torch.equal(o1, lock)


@pytest.mark.parametrize("C,K,d", [(768, 25, 1), (896, 29, 2), (512, 17, 1), (384, 13, 1), (1024, 1 + 4, 1)])
def test_conv_wgrad_one_wave_per_simd_full_size_vs_lockstep(cuda, C, K, d):
  """BASELINE-size layers (B = 32, T' = 840, ragged): the one-wave-per-SIMD kernel unsplit is BIT-IDENTICAL to
  the oracle-checked lockstep kernel and to the ping-pong kernel; cut 4 ways it differs by fp32 summation order
  only. Also the default launch (ping-pong kernel, cost-model split, XCD-ordered ranks): deterministic."""
  from openseq2seq_amd import capi, _lib
  g = torch.Generator().manual_seed(5 + C + K)
  B, T = 32, 840
  x = _bf(torch.randn(B, T, C, generator=g)).to(cuda)
  dy = _bf(torch.randn(B, T, C, generator=g)).to(cuda)
  lens = torch.randint(100, T + 1, (B,), generator=g).to(torch.int32).to(cuda)
  try:
    _lib.set_option("conv1d_wgrad.variant", 0); _lib.set_option("conv1d_wgrad.split", -1)
    ref = capi.conv1d_wgrad(x, dy, K, dil=d, in_len=lens)
    _lib.set_option("conv1d_wgrad.variant", 1); _lib.set_option("conv1d_wgrad.split", 1)
    pp = capi.conv1d_wgrad(x, dy, K, dil=d, in_len=lens)
    _lib.set_option("conv1d_wgrad.variant", 3); _lib.set_option("conv1d_wgrad.split", 1)
    a = capi.conv1d_wgrad(x, dy, K, dil=d, in_len=lens)
    # (a register read in the shadow of the last MFMAs showed up as ONE wrong 32 x 32 block in one launch of ten)
    again = [capi.conv1d_wgrad(x, dy, K, dil=d, in_len=lens) for _ in range(12)]
    _lib.set_option("conv1d_wgrad.variant", 3); _lib.set_option("conv1d_wgrad.split", 4)
    b = capi.conv1d_wgrad(x, dy, K, dil=d, in_len=lens)
    _lib.set_option("conv1d_wgrad.variant", -1); _lib.set_option("conv1d_wgrad.split", -1)
    c = capi.conv1d_wgrad(x, dy, K, dil=d, in_len=lens)
    c2 = capi.conv1d_wgrad(x, dy, K, dil=d, in_len=lens)
    torch.cuda.synchronize()
  finally:
    _lib.set_option("conv1d_wgrad.variant", -1); _lib.set_option("conv1d_wgrad.split", -1)
  assert torch.equal(pp, ref)
  assert torch.equal(a, ref), (int((a != ref).sum()), float((a - ref).abs().max()))
  for o in again:
    assert torch.equal(o, ref), (int((o != ref).sum()), float((o - ref).abs().max()))
  tol = 1e-4 * float(ref.abs().max())
  torch.testing.assert_close(b, ref, rtol=1e-4, atol=tol)
  torch.testing.assert_close(c, ref, rtol=1e-4, atol=tol)
  assert torch.equal(c, c2)


@pytest.mark.parametrize("B,T,Cin,Cout,K,d,n", [(4, 300, 256, 256, 11, 1, 5), (3, 420, 384, 384, 13, 1, 4),
                                                (2, 260, 128, 640, 5, 2, 8), (32, 840, 640, 640, 21, 1, 2),
                                                (2, 150, 192, 200, 5, 1, 3), (2, 100, 64, 64, 3, 1, 2)])
def test_conv_wgrad_grouped_equals_single_launches(cuda, B, T, Cin, Cout, K, d, n):
  """os2s_conv1d_wgrad_grouped_ws: n layers of ONE shape over one ragged batch in one launch of the ping-pong
  weight-gradient kernel (units ranked group-major; the tail of the launch cut along the reduction) against n single
  launches: the same fp32 sums up to the order in which split pieces are added (rtol 1e-4), accumulate semantics,
  run-to-run bitwise reproducibility, tickets left at zero; shapes outside the kernel's envelope (the last two cases:
  channel tails / fewer than 128 output channels) fall back to single launches and are bit-equal to them."""
  from openseq2seq_amd import capi
  g = torch.Generator().manual_seed(B * 31 + T + K + Cout + n)
  lens = torch.randint(T // 4, T + 1, (B,), generator=g).to(torch.int32)
  lens[0] = T
  lens = lens.to(cuda)
  xs = [_bf(torch.randn(B, T, Cin, generator=g)).to(cuda) for _ in range(n)]
  dys = [_bf(torch.randn(B, T, Cout, generator=g)).to(cuda) for _ in range(n)]
  base = [torch.randn(K, Cout, Cin, generator=g).to(cuda) for _ in range(n)]
  single = [b.clone() for b in base]
  for x, dy, dw in zip(xs, dys, single):
    capi.conv1d_wgrad(x, dy, K, dil=d, in_len=lens, out=dw, accumulate=True)
  outs = []
  for _ in range(2):
    grouped = [b.clone() for b in base]
    capi.conv1d_wgrad_grouped([dict(x=x, dy=dy, dw=dw) for x, dy, dw in zip(xs, dys, grouped)], K, dil=d,
                              in_len=lens, accumulate=True)
    outs.append(grouped)
  fresh = [torch.full_like(b, float("nan")) for b in base]
  capi.conv1d_wgrad_grouped([dict(x=x, dy=dy, dw=dw) for x, dy, dw in zip(xs, dys, fresh)], K, dil=d,
                            in_len=lens, accumulate=False)
  torch.cuda.synchronize()
  envelope = Cout >= 128 and Cin >= 64
  for i in range(n):
    scale = float((single[i] - base[i]).abs().max())
    if envelope:
      torch.testing.assert_close(outs[0][i], single[i], rtol=1e-4, atol=1e-4 * scale)
    else:
      assert torch.equal(outs[0][i], single[i])
    assert torch.equal(outs[0][i], outs[1][i])
    torch.testing.assert_close(fresh[i], single[i] - base[i], rtol=1e-4, atol=1e-4 * scale)
  assert int(capi.conv1d_workspace(cuda)[:4096].view(torch.int32).abs().sum()) == 0


@pytest.mark.parametrize("B,T,Cin,Cout", [(3, 420, 256, 512), (2, 333, 320, 640), (1, 1111, 1024, 1024),
                                         (4, 200, 128, 264), (5, 97, 520, 136)])
@pytest.mark.parametrize("split", [1, 3, -1])
def test_conv_wgrad1x1_pingpong_kernel(cuda, B, T, Cin, Cout, split):
  """conv1d_wgrad1x1_pp_kernel (K = 1: the Dense weight gradient, 256 x 256 tile) forced, with Cin /
  Cout tails, ragged lengths incl. an empty sample, a row-strided X view, the reduction unsplit,
  cut 3 ways and by the cost model: vs the fp64 product of the same bf16 values (fp32 summation
  order is the only noise: rtol 2e-3), accumulate on top of a previous dW, bitwise run-to-run
  reproducibility (no atomics), tickets back at zero."""
  from openseq2seq_amd import capi, _lib
  g = torch.Generator().manual_seed(B * 131 + T + Cout)
  xw = _bf(torch.randn(B, T, Cin + 64, generator=g))
  x = xw[:, :, 32:32 + Cin]                                 # channel slice of a wider tensor
  dy = _bf(torch.randn(B, T, Cout, generator=g))
  lens = torch.randint(T // 4, T + 1, (B,), generator=g).to(torch.int32)
  lens[0] = T
  if B > 2:
    lens[1] = 0
  mask = (torch.arange(T)[None, :] < lens[:, None]).double()[:, :, None]
  ref = torch.einsum("btc,bti->ci", dy.double() * mask, x.double()).float()[None]   # [1,Cout,Cin]
  base = torch.randn(1, Cout, Cin, generator=g)
  xd = xw.to(cuda)[:, :, 32:32 + Cin]
  L = _lib.lib()
  _lib.set_option("conv1d_wgrad.variant", 2); _lib.set_option("conv1d_wgrad.split", split)
  try:
    o1 = capi.conv1d_wgrad(xd, dy.to(cuda), 1, pad_left=0, in_len=lens.to(cuda))
    o2 = capi.conv1d_wgrad(xd, dy.to(cuda), 1, pad_left=0, in_len=lens.to(cuda))
    o3 = base.clone().to(cuda)
    capi.conv1d_wgrad(xd, dy.to(cuda), 1, pad_left=0, in_len=lens.to(cuda), out=o3, accumulate=True)
    torch.cuda.synchronize()
  finally:
    _lib.set_option("conv1d_wgrad.variant", -1); _lib.set_option("conv1d_wgrad.split", -1)
  scale = float(ref.pow(2).mean().sqrt()) + 1e-6
  torch.testing.assert_close(o1.cpu(), ref, rtol=2e-3, atol=2e-3 * scale)
  assert torch.equal(o1, o2)
  torch.testing.assert_close(o3.cpu(), ref + base, rtol=2e-3, atol=2e-3 * scale)
  assert int(capi.conv1d_workspace(cuda)[:4096].view(torch.int32).abs().sum()) == 0


def test_conv_wgrad1x1_pingpong_dense_size_vs_lockstep(cuda):
  """Transformer-big Dense weight gradients (8300 packed tokens; 1024 -> 1024 / 4096 / 3072 and the
  tied-softmax 1024 -> 32768) by the cost model's split: the ping-pong kernel vs the oracle-checked
  lockstep kernel without batch split (fp32 summation order only)."""
  from openseq2seq_amd import capi, _lib
  g = torch.Generator().manual_seed(11)
  N = 8300
  L = _lib.lib()
  for Cin, Cout in ((1024, 1024), (1024, 4096), (4096, 1024), (1024, 3072), (1024, 32768)):
    x = _bf(torch.randn(1, N, Cin, generator=g)).to(cuda)
    dy = _bf(torch.randn(1, N, Cout, generator=g)).to(cuda)
    try:
      _lib.set_option("conv1d_wgrad.variant", 0); _lib.set_option("conv1d_wgrad.split", -1)
      ref = capi.conv1d_wgrad(x, dy, 1, pad_left=0)
      _lib.set_option("conv1d_wgrad.variant", 2); _lib.set_option("conv1d_wgrad.split", -1)
      a = capi.conv1d_wgrad(x, dy, 1, pad_left=0)
      torch.cuda.synchronize()
    finally:
      _lib.set_option("conv1d_wgrad.variant", -1); _lib.set_option("conv1d_wgrad.split", -1)
    torch.testing.assert_close(a, ref, rtol=1e-4, atol=1e-4 * float(ref.abs().max()))


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_conv1x1_grouped_equals_single_launches(cuda, variant):
  """os2s_conv1x1_fwd_grouped (the dense-residual branches of a block end in one launch; their data
  gradients with accumulate / out_len) against single-layer launches of the lockstep 128x128 tile:
  outputs BIT-IDENTICAL group by group, ragged lengths and dead windows included (same k order in
  every tile); BatchNorm partials identical for the lockstep tile and within fp32 summation-order
  noise (1e-5 relative to the window's largest partial) for the ping-pong tile, whose waves cover
  the 128 rows of a window in a different order. variant = option conv1x1.variant."""
  from openseq2seq_amd import capi, _lib
  g = torch.Generator().manual_seed(21)
  B, T = 3, 300
  lens = torch.tensor([300, 170, 40], dtype=torch.int32, device=cuda)
  shapes = [(256, 768), (384, 768), (640, 768), (768, 200), (64, 128), (128, 520)]
  nm = capi.conv1d_num_mtiles(B, T)
  L = _lib.lib()
  items, ref = [], []
  items2, ref2 = [], []
  try:
    _lib.set_option("conv1x1.variant", 1)
    _lib.set_option("conv1d.variant", 0)
    for cin, cout in shapes:
      x = _bf(torch.randn(B, T, cin, generator=g)).to(cuda)
      w = _bf(torch.randn(1, cout, cin, generator=g) * 0.05).to(cuda)
      y = torch.full((B, T, cout), 3.0, dtype=torch.bfloat16, device=cuda)
      st = torch.full((nm, 2, cout), float("nan"), device=cuda)
      items.append(dict(x=x, w=w, y=y, stats=st))
      st2 = torch.full((nm, 2, cout), float("nan"), device=cuda)
      ref.append((capi.conv1d_fwd(x, w, pad_left=0, tout=T, in_len=lens, stats=st2), st2))
    for cin, cout in shapes[:3]:
      dy = _bf(torch.randn(B, T, cout, generator=g)).to(cuda)
      wt = _bf(torch.randn(1, cin, cout, generator=g) * 0.05).to(cuda)
      base = _bf(torch.randn(B, T, cin, generator=g)).to(cuda)
      a, b = base.clone(), base.clone()
      capi.conv1d_fwd(dy, wt, pad_left=0, tout=T, out=a, accumulate=True, out_len=lens)
      items2.append(dict(x=dy, w=wt, y=b, accumulate=True))
      ref2.append(a)
    _lib.set_option("conv1d.variant", -1)
    _lib.set_option("conv1x1.variant", variant)
    capi.conv1x1_fwd_grouped(items, in_len=lens)
    # data-gradient form: accumulate into existing buffers, rows past out_len untouched
    capi.conv1x1_fwd_grouped(items2, out_len=lens)
    # a wide K = 1 layer through the ordinary entry point (bias + ReLU epilogue)
    xs, ws = items[2]["x"], items[2]["w"]
    bias = torch.randn(768, generator=g).to(cuda)
    single = capi.conv1d_fwd(xs, ws, pad_left=0, tout=T, in_len=lens, bias=bias, act=1)
    _lib.set_option("conv1x1.variant", 1)
    _lib.set_option("conv1d.variant", 0)
    single_ref = capi.conv1d_fwd(xs, ws, pad_left=0, tout=T, in_len=lens, bias=bias, act=1)
    torch.cuda.synchronize()
  finally:
    _lib.set_option("conv1d.variant", -1)
    _lib.set_option("conv1x1.variant", -1)
  for it, (y, st) in zip(items, ref):
    assert torch.equal(it["y"], y)
    if variant == 1:
      assert torch.equal(it["stats"], st)
    else:
      torch.testing.assert_close(it["stats"], st, rtol=0, atol=1e-5 * float(st.abs().max()))
  for it, a in zip(items2, ref2):
    assert torch.equal(it["y"], a)
  live = (torch.arange(T, device=cuda)[None, :] < lens[:, None])[:, :, None]
  assert torch.equal(torch.where(live, single, torch.zeros((), dtype=single.dtype, device=cuda)),
                     torch.where(live, single_ref, torch.zeros((), dtype=single.dtype, device=cuda)))


@pytest.mark.parametrize("case", ["block_768", "narrow_split", "sliced_input", "block_768_big", "sliced_big"])
def test_conv1x1_wgrad_grouped_equals_single_launches(cuda, case):
  """os2s_conv1x1_wgrad_grouped (the K = 1 weight gradients of the dense-residual branches of a
  block end in one launch) against one os2s_conv1d_wgrad per branch and against the fp32 matmul:
  ragged lengths with dead 64-row chunks, channel tails, accumulation into non-zero dW, an input
  that is a channel slice of a wider tensor.
    block_768: the branches of the last Jasper block (+ 2 more) at a small T — 288 output tiles
      (>= 256), so the reduction is not split: every dW element receives ONE add -> bit-identical
      from run to run,
      and equal to the single launches up to fp32 summation order (rtol 1e-4);
    narrow_split: few small groups -> the reduction is split and combined with fp32 atomics;
    block_768_big / sliced_big (round 5): >= 2048 rows and every group >= 128 x 128 channels -> the launch runs
      on the K = 1 ping-pong TN-GEMM kernel (os2s_conv1x1_wgrad_grouped_ws): 57 / 3 tiles of 256 x 256 cut
      along the live 64-row chunks, one owner per dW element -> bit-identical from run to run, also with
      pingpong=False (the lockstep kernel) inside the same tolerances."""
  from openseq2seq_amd import capi
  g = torch.Generator().manual_seed(33)
  if case == "block_768":
    B, T, lens = 4, 330, [330, 201, 64, 17]
    shapes = [(256, 768)] * 3 + [(384, 768)] * 2 + [(512, 768)] * 2 + [(640, 768)] * 2 + [(768, 768)] * 3
  elif case == "narrow_split":
    B, T, lens = 5, 500, [500, 350, 129, 64, 1]
    shapes = [(256, 256), (72, 200), (128, 136)]
  elif case == "block_768_big":
    B, T, lens = 8, 400, [400, 330, 201, 64, 17, 400, 1, 129]
    shapes = [(256, 768)] * 3 + [(384, 768)] * 2 + [(512, 768)] * 2 + [(640, 768)] * 2 + [(768, 768)] * 3
  elif case == "sliced_big":
    B, T, lens = 6, 400, [400, 90, 33, 400, 257, 128]
    shapes = [(128, 256), (192, 256), (192, 384)]
  else:
    B, T, lens = 3, 200, [200, 90, 33]
    shapes = [(128, 256), (192, 256)]
  lens = torch.tensor(lens, dtype=torch.int32, device=cuda)
  mask = (torch.arange(T, device=cuda)[None, :] < lens[:, None])[:, :, None]
  items, refs, singles = [], [], []
  wide = _bf(torch.randn(B, T, 512, generator=g)).to(cuda) * mask if case.startswith("sliced") else None
  off = 0
  for cin, cout in shapes:
    if wide is not None:
      x = wide[:, :, off:off + cin]
      off += cin
    else:
      x = (_bf(torch.randn(B, T, cin, generator=g)).to(cuda) * mask).contiguous()   # masked conv inputs
    dy = _bf(torch.randn(B, T, cout, generator=g)).to(cuda)
    base = torch.randn(1, cout, cin, generator=g).to(cuda)
    items.append(dict(x=x, dy=dy, dw=base.clone()))
    ref = base.double() + torch.einsum("btc,bti->ci", dy.double(), x.double())[None]
    refs.append(ref.float())
    s = base.clone()
    capi.conv1d_wgrad(x, dy, 1, pad_left=0, in_len=lens, out=s, accumulate=True)
    singles.append(s)
  capi.conv1x1_wgrad_grouped(items, in_len=lens)
  torch.cuda.synchronize()
  for it, ref, s in zip(items, refs, singles):
    scale = float(ref.abs().max())
    torch.testing.assert_close(it["dw"], ref, rtol=1e-4, atol=1e-4 * scale)
    torch.testing.assert_close(it["dw"], s, rtol=1e-4, atol=1e-4 * scale)
  if case.endswith("_big"):     # the lockstep kernel on the same problem: same tolerances
    lock = [dict(it, dw=b.clone()) for it, b in zip(items, [r - torch.einsum("btc,bti->ci", it["dy"].double(),
            it["x"].double())[None].float().to(cuda) for it, r in zip(items, [r.to(cuda) for r in refs])])]
    capi.conv1x1_wgrad_grouped(lock, in_len=lens, pingpong=False)
    torch.cuda.synchronize()
    for it, ref in zip(lock, refs):
      torch.testing.assert_close(it["dw"], ref.to(cuda), rtol=2e-4, atol=2e-4 * float(ref.abs().max()))
  if case in ("block_768", "block_768_big", "sliced_big"):       # one owner per element: run-to-run bit-identical
    again = [dict(it, dw=torch.zeros_like(it["dw"])) for it in items]
    again2 = [dict(it, dw=torch.zeros_like(it["dw"])) for it in items]
    capi.conv1x1_wgrad_grouped(again, in_len=lens)
    capi.conv1x1_wgrad_grouped(again2, in_len=lens)
    torch.cuda.synchronize()
    for a, b in zip(again, again2):
      assert torch.equal(a["dw"], b["dw"])

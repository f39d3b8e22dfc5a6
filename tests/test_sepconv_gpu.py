"""GPU parity of the separable-convolution path (QuartzNet, layer type "sep_conv1d"):
 * the depthwise kernels (forward, flipped-tap data gradient, weight gradient) vs
   torch.nn.functional.conv1d(groups=C) with ragged lengths, stride and dilation;
 * a scaled-down QuartzNet (stride-2 first layer, residual separable blocks with K = 33 / 39,
   dilated K = 87 layer, 1x1 layer) + FC-CTC: logits, loss and every parameter gradient vs the
   fp32 oracle (bf16 storage noise through 8 separable conv+BN layers with tiny BN batches:
   logits rel-L2 <= 3e-2, gradients cosine >= 0.97, rel-L2 <= 0.25)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

LAYERS = [
    {"type": "sep_conv1d", "repeat": 1, "kernel_size": [33], "stride": [2], "num_channels": 128,
     "padding": "SAME", "dilation": [1]},
    {"type": "sep_conv1d", "repeat": 2, "kernel_size": [33], "stride": [1], "num_channels": 128,
     "padding": "SAME", "dilation": [1], "residual": True, "residual_dense": False},
    {"type": "sep_conv1d", "repeat": 2, "kernel_size": [39], "stride": [1], "num_channels": 256,
     "padding": "SAME", "dilation": [1], "residual": True, "residual_dense": False},
    {"type": "sep_conv1d", "repeat": 1, "kernel_size": [87], "stride": [1], "num_channels": 256,
     "padding": "SAME", "dilation": [2]},
    {"type": "conv1d", "repeat": 1, "kernel_size": [1], "stride": [1], "num_channels": 384,
     "padding": "SAME", "dilation": [1]},
]


@pytest.mark.parametrize("K,stride,dil", [(33, 1, 1), (11, 2, 1), (87, 1, 2), (1, 1, 1), (75, 1, 1), (16, 1, 1),
                                          (39, 1, 2), (20, 1, 4)])
@pytest.mark.parametrize("shape", [(3, 200, 72), (4, 700, 200)])
def test_depthwise_kernels(cuda, K, stride, dil, shape):
  """Depthwise forward / flipped-tap data gradient / weight gradient vs conv1d(groups=C) of the
  same bf16 inputs. Stride 1, dilation 1, 2 <= K <= 96 takes the matrix-core kernels of round 6 (K = 16, 33, 75:
  4 / 4 / 7 window steps, 2 / 3 / 6 window tiles of the weight gradient; C = 72 / 200: a partial 32- and
  16-channel block); stride 1 with dilation 2 / 4 and K = 1 the register-window kernels (dilation = residue
  classes of the rows; T = 700: three 256-step tiles per sample, one of them partial), stride 2 the generic
  kernels."""
  _check_depthwise(cuda, K, stride, dil, shape)


@pytest.mark.parametrize("variant", [-1, 1])
@pytest.mark.parametrize("K", [2, 33, 64, 96])
@pytest.mark.parametrize("shape", [(2, 2100, 96), (5, 1000, 40), (33, 130, 64)])
def test_depthwise_matrix_core_kernels_tile_edges(cuda, K, shape, variant):
  """The matrix-core kernels past one tile: T = 2100 = three 928 / 992-step tiles per sample (the last one partial,
  the weight gradient's accumulators carried over the tiles a workgroup walks), T = 1000 just past a tile, 33 samples
  of 130 steps (tiles with 4 - 5 of 29 - 31 segments), the smallest and the largest K the kernels take; variant 1 =
  the same shapes on the register-window kernels (the two families against the same reference)."""
  from openseq2seq_amd import _lib
  import ctypes
  _lib.lib().os2s_set_option(b"depthwise.variant", ctypes.c_double(variant))
  try:
    _check_depthwise(cuda, K, 1, 1, shape)
  finally:
    _lib.lib().os2s_set_option(b"depthwise.variant", ctypes.c_double(-1))


def _check_depthwise(cuda, K, stride, dil, shape):
  from openseq2seq_amd import capi
  g = torch.Generator().manual_seed(K)
  B, T, C = shape
  lens = torch.tensor(([T, (2 * T) // 3 + 1, T // 4 + 7] + [T // 2] * max(B - 3, 0))[:B], dtype=torch.int32)
  x = torch.randn(B, T, C, generator=g).to(torch.bfloat16)
  x = x * (torch.arange(T)[None, :, None] < lens[:, None, None])      # inputs are stored masked
  w = torch.randn(K, C, generator=g) * 0.3
  tout, pl = capi.same_padding(T, K, stride, dil)
  y = capi.depthwise_conv1d_fwd(x.to(cuda), w.to(cuda), stride=stride, dil=dil, in_len=lens.to(cuda))
  xr = x.float().requires_grad_(True)
  wr = w.clone().requires_grad_(True)
  tot = max((tout - 1) * stride + (K - 1) * dil + 1 - T, 0)
  xp = F.pad(xr.permute(0, 2, 1), (pl, tot - pl))
  ref = F.conv1d(xp, wr.t()[:, None, :], stride=stride, dilation=dil, groups=C).permute(0, 2, 1)
  assert y.shape == ref.shape
  torch.testing.assert_close(y.float().cpu(), ref.detach(), atol=3e-2, rtol=2e-2)
  dy = torch.randn(B, tout, C, generator=g).to(torch.bfloat16)
  (ref * dy.float()).sum().backward()
  dw = torch.zeros(K, C, device=cuda)
  capi.depthwise_conv1d_wgrad(x.to(cuda), dy.to(cuda), dw, stride=stride, dil=dil, in_len=lens.to(cuda))
  torch.testing.assert_close(dw.cpu(), wr.grad, atol=5e-2, rtol=2e-2)
  if stride == 1:
    dx = capi.depthwise_conv1d_fwd(dy.to(cuda), w.to(cuda), dil=dil, pad_left=(K - 1) * dil - pl, tout=T,
                                   out_len=lens.to(cuda), flip=True).float().cpu()
    m = (torch.arange(T)[None, :, None] < lens[:, None, None])
    torch.testing.assert_close(dx * m, xr.grad * m, atol=5e-2, rtol=2e-2)


def test_quartznet_small_fwd_bwd(cuda):
  from openseq2seq_amd.optimizers.flat_params import FlatParams
  from openseq2seq_amd.encoders.tdnn_encoder import TDNNEncoder
  from openseq2seq_amd.decoders.fc_decoders import FullyConnectedCTCDecoder
  from openseq2seq_amd.losses.ctc_loss import CTCLoss
  from openseq2seq_amd.parts.cnns.conv_blocks import Tape
  from oracle import tdnn
  torch.manual_seed(0)
  store = FlatParams(cuda)
  enc = TDNNEncoder({"convnet_layers": [dict(l, dropout_keep_prob=1.0) for l in LAYERS],
                     "dropout_keep_prob": 1.0, "activation_fn": "relu", "use_conv_mask": True,
                     "dtype": "mixed"}, None, mode="train").build(store, 64)
  dec = FullyConnectedCTCDecoder({"tgt_vocab_size": 29, "dtype": "mixed"}, None,
                                 mode="train").build(store, enc.output_dim)
  lossf = CTCLoss({"dtype": "mixed"}, None)
  store.finalize()
  g = torch.Generator().manual_seed(1)
  B, T = 3, 160
  x = torch.randn(B, T, 64, generator=g).to(torch.bfloat16)
  lens = torch.tensor([160, 118, 75], dtype=torch.int32)
  labels = torch.randint(0, 28, (B, 12), generator=g).to(torch.int32)
  label_len = torch.tensor([12, 7, 3], dtype=torch.int32)
  tape = Tape()
  e = enc.encode({"source_tensors": [x.to(cuda), lens.to(cuda)], "tape": tape, "seed": 3})
  d = dec.decode({"encoder_output": e, "tape": tape})
  L = lossf.compute_loss({"decoder_output": d, "target_tensors": [labels.to(cuda), label_len.to(cuda)]})
  store.zero_grads()
  tape.backward()
  torch.cuda.synchronize()
  pre = "ForwardPass/w2l_encoder/"
  w = {}
  for p in store.params:
    if p.name.startswith(pre):
      n = p.name[len(pre):]
      if p.kind == "conv":
        w[n] = p.w16.float().cpu().permute(0, 2, 1).contiguous().requires_grad_(True)
      else:
        w[n] = p.master.cpu().clone().requires_grad_(True)
  fcw = dec.kernel.w16.float().cpu()[0, :29, :].t().contiguous().requires_grad_(True)
  fcb = dec.bias.master.cpu()[:29].clone().requires_grad_(True)
  out, olen = tdnn.tdnn_encode(x.float(), lens, LAYERS, w)
  logits, loss = tdnn.fc_ctc(out, olen, fcw, fcb, labels, label_len)
  loss.backward()
  lg = d["logits"].cpu()
  rel = float((lg - logits.detach()).norm() / logits.detach().norm())
  assert rel < 3e-2, rel
  torch.testing.assert_close(L.cpu()[0], loss.detach(), rtol=2e-2, atol=1e-2)
  bad = []
  for p in store.params:
    if p.name.startswith(pre):
      ref = w[p.name[len(pre):]].grad
      if p.kind == "conv":
        ref = ref.permute(0, 2, 1)
      got = p.grad.cpu()
    elif p.name.endswith("fully_connected/kernel"):
      ref, got = fcw.grad.t(), p.grad.cpu()[0, :29, :]
    else:
      ref, got = fcb.grad, p.grad.cpu()[:29]
    cos = float(torch.nn.functional.cosine_similarity(got.flatten(), ref.flatten(), dim=0))
    relerr = float((got - ref).norm() / (ref.norm() + 1e-12))
    if not (cos > 0.97 and relerr < 0.25):
      bad.append((p.name, round(cos, 4), round(relerr, 4)))
  assert not bad, bad


@pytest.mark.parametrize("ragged", [True, False])
def test_separable_layers_fused_bn_backward(cuda, monkeypatch, ragged):
  """The BatchNorm-backward reduction of a separable conv + BN + ReLU layer rides in the store phase of the NEXT
  separable layer's depthwise data gradient (capi.depthwise_dgrad_bnact -> os2s_depthwise_dgrad_bnact, the
  depthwise twin of the convolution's fused data gradient): a QuartzNet-shaped stack (stride-2 first layer, two
  residual separable blocks — the block inputs also feed a residual branch, so the fused launch ADDS to an earlier
  contribution — K = 33 / 39, a dilated layer that keeps the unfused path) run with the fusion on and off. With
  keep = 1 both paths store the same dz values; the sums differ in order only (fp32), which moves a few bf16
  roundings of dy in the layers below."""
  from openseq2seq_amd import capi
  from openseq2seq_amd.optimizers.flat_params import FlatParams
  from openseq2seq_amd.encoders.tdnn_encoder import TDNNEncoder
  from openseq2seq_amd.parts.cnns import conv_blocks
  from openseq2seq_amd.parts.cnns.conv_blocks import Tape
  g = torch.Generator().manual_seed(9)
  B, T, F_ = 4, 1500, 64
  x0 = torch.randn(B, T, F_, generator=g).to(torch.bfloat16).to(cuda)
  lens = torch.tensor([1500, 1111, 700, 90] if ragged else [1500] * 4, dtype=torch.int32, device=cuda)
  calls = []
  orig = capi.depthwise_dgrad_bnact

  def counting(*a, **kw):
    calls.append(kw.get("addend") is not None)
    return orig(*a, **kw)
  monkeypatch.setattr(capi, "depthwise_dgrad_bnact", counting)
  res = {}
  for fused in (False, True):
    monkeypatch.setattr(conv_blocks, "SEP_FUSE_BN_BWD", fused)
    torch.manual_seed(3)
    store = FlatParams(cuda)
    # (a plain stride-1 separable layer in front of the first residual block: its output feeds the block's first
    # layer AND the block end's residual branch — the fused launch then adds to the branch's earlier contribution)
    layers = [dict(l, dropout_keep_prob=1.0) for l in
              [LAYERS[0], dict(LAYERS[1], repeat=1, residual=False)] + LAYERS[1:]]
    enc = TDNNEncoder({"convnet_layers": layers, "dropout_keep_prob": 1.0, "activation_fn": "relu",
                       "use_conv_mask": True, "dtype": "mixed"}, None, mode="train").build(store, F_)
    store.finalize()
    store.zero_grads()
    tape = Tape()
    out = enc._encode({"source_tensors": [x0, lens], "tape": tape, "seed": 1})["outputs_act"]
    out.grad = torch.randn(out.data.shape, generator=torch.Generator().manual_seed(5)).to(torch.bfloat16).to(cuda)
    tape.backward()
    torch.cuda.synchronize()
    res[fused] = (out.data.float().cpu(), {p.name: p.grad.float().cpu().clone() for p in store.params})
  assert len(calls) >= 3 and any(calls) and not all(calls), calls     # fused launches with and without an addend
  assert torch.equal(res[True][0], res[False][0])
  worst = (0.0, "")
  for n, gu in res[False][1].items():
    gf = res[True][1][n]
    r = float((gf - gu).norm() / (gu.norm() + 1e-20))
    worst = max(worst, (r, n))
    assert r <= 1e-2, (n, r)        # (measured 3e-3 at the first layer: the sums differ in order, a few bf16 roundings of dy behind them)
  print("separable fused BN backward (ragged=%s): %d fused launches, worst gradient difference %.2e (%s)"
        % (ragged, len(calls), worst[0], worst[1]))


def test_pointwise_fold_kernels(cuda):
  """os2s_pointwise_fold / os2s_pointwise_fold_bwd against their definitions (fp32; bf16 outputs round to nearest
  even), with channel counts that are not multiples of the kernels' 32 / 64-wide blocks."""
  from openseq2seq_amd import capi
  g = torch.Generator().manual_seed(2)
  for cout, cin in [(256, 256), (200, 136), (512, 264), (8, 8)]:
    w = torch.randn(1, cout, cin, generator=g).to(cuda)
    d = (torch.randn(1, cin, generator=g) + 1.0).to(cuda)
    we, wte = capi.pointwise_fold(w, d)
    ref = (w * d.view(1, 1, cin)).to(torch.bfloat16)
    assert torch.equal(we, ref)
    assert torch.equal(wte, ref.permute(0, 2, 1).contiguous())
    G = torch.randn(1, cout, cin, generator=g).to(cuda)
    dw0 = torch.randn(1, cout, cin, generator=g).to(cuda)
    dd0 = torch.randn(1, cin, generator=g).to(cuda)
    dw, dd = dw0.clone(), dd0.clone()
    capi.pointwise_fold_bwd(G, w, d, dw, dd)
    torch.testing.assert_close(dw, dw0 + G * d.view(1, 1, cin), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(dd, dd0 + (w.double() * G.double()).sum(1).float(), rtol=1e-5, atol=1e-4)
    dw2, dd2 = dw0.clone(), dd0.clone()
    capi.pointwise_fold_bwd(G, w, d, dw2, dd2)
    assert torch.equal(dw, dw2) and torch.equal(dd, dd2)          # one writer per element, fixed order


def test_one_tap_separable_layer_math(cuda):
  """y = (x . d) W^T and its three gradients through the folded form — the 1x1 kernels with W diag(d), G = dy^T x
  split by os2s_pointwise_fold_bwd — against fp32 on the same bf16 inputs, ragged rows included."""
  from openseq2seq_amd import capi
  g = torch.Generator().manual_seed(6)
  B, T, cin, cout = 3, 300, 256, 384
  lens = torch.tensor([300, 170, 40], dtype=torch.int32)
  m = (torch.arange(T)[None, :, None] < lens[:, None, None]).float()
  x = (torch.randn(B, T, cin, generator=g) * m).to(torch.bfloat16)
  dy = (torch.randn(B, T, cout, generator=g) * m).to(torch.bfloat16)
  w = torch.randn(1, cout, cin, generator=g) * 0.05
  d = torch.randn(1, cin, generator=g) * 0.5 + 1.0
  we, wte = capi.pointwise_fold(w.to(cuda), d.to(cuda))
  y = capi.conv1d_fwd(x.to(cuda), we, pad_left=0, tout=T, in_len=lens.to(cuda)).float().cpu()
  dx = capi.conv1d_fwd(dy.to(cuda), wte, pad_left=0, tout=T, out_len=lens.to(cuda)).float().cpu()
  G = capi.conv1d_wgrad(x.to(cuda), dy.to(cuda), 1, pad_left=0, in_len=lens.to(cuda))
  dw = torch.zeros(1, cout, cin, device=cuda)
  dd = torch.zeros(1, cin, device=cuda)
  capi.pointwise_fold_bwd(G, w.to(cuda), d.to(cuda), dw, dd)
  xf, dyf, W = x.double(), dy.double(), w[0].double()
  z = xf * d.double().view(1, 1, cin)
  y_ref = z @ W.t()
  dz_ref = dyf @ W
  dx_ref = dz_ref * d.double().view(1, 1, cin)
  dw_ref = dyf.reshape(-1, cout).t() @ z.reshape(-1, cin)
  dd_ref = (xf * dz_ref).sum((0, 1))

  def rel(a, b):
    return float((a.double() - b).norm() / b.norm())
  errs = dict(y=rel(y * m, y_ref * m), dx=rel(dx * m, dx_ref * m), dw=rel(dw.cpu()[0], dw_ref), dd=rel(dd.cpu()[0], dd_ref))
  print("one-tap separable layer, folded form vs fp64:", {k: "%.2e" % v for k, v in errs.items()})
  # bf16 storage of W diag(d) (2^-9 per element, rms 2^-10.3) and of y / dx; dw and dd carry no rounding of their own
  assert errs["y"] <= 4e-3 and errs["dx"] <= 4e-3 and errs["dw"] <= 1e-4 and errs["dd"] <= 1e-4, errs


@pytest.mark.parametrize("ragged", [True, False])
def test_one_tap_separable_layer_folded(cuda, monkeypatch, ragged):
  """The residual branches of a separable block are separable layers with ONE tap (conv_blocks.py:66,79-85); they run
  as one 1x1 convolution with the depthwise scale folded into the pointwise kernel (SepConvBN.folded) — compared
  here with the two-launch form (OS2S_FOLD_SEP_K1=0) on a QuartzNet-shaped stack: the outputs differ by the bf16
  rounding of x * d against that of W * d, the gradients of every variable — the branches' depthwise and pointwise
  kernels among them — agree to the same order."""
  from openseq2seq_amd import capi
  from openseq2seq_amd.optimizers.flat_params import FlatParams
  from openseq2seq_amd.encoders.tdnn_encoder import TDNNEncoder
  from openseq2seq_amd.parts.cnns import conv_blocks
  from openseq2seq_amd.parts.cnns.conv_blocks import Tape
  g = torch.Generator().manual_seed(11)
  B, T, F_ = 4, 600, 64
  x0 = torch.randn(B, T, F_, generator=g).to(torch.bfloat16).to(cuda)
  lens = torch.tensor([600, 411, 300, 90] if ragged else [600] * 4, dtype=torch.int32, device=cuda)
  calls = []
  orig = capi.pointwise_fold

  def counting(*a, **kw):
    calls.append(1)
    return orig(*a, **kw)
  monkeypatch.setattr(capi, "pointwise_fold", counting)
  res = {}
  for fold in (False, True):
    monkeypatch.setattr(conv_blocks, "FOLD_SEP_K1", fold)
    torch.manual_seed(3)
    store = FlatParams(cuda)
    enc = TDNNEncoder({"convnet_layers": [dict(l, dropout_keep_prob=1.0) for l in LAYERS], "dropout_keep_prob": 1.0,
                       "activation_fn": "relu", "use_conv_mask": True, "dtype": "mixed"}, None,
                      mode="train").build(store, F_)
    store.finalize()
    # the depthwise scales of the branches away from their initial values
    for p in store.params:
      if p.name.endswith("depthwise_kernel") and p.shape[0] == 1:
        p.master.mul_(torch.rand(p.master.shape, generator=torch.Generator().manual_seed(4)).to(cuda) + 0.5)
    store.zero_grads()
    tape = Tape()
    out = enc._encode({"source_tensors": [x0, lens], "tape": tape, "seed": 1})["outputs_act"]
    out.grad = torch.randn(out.data.shape, generator=torch.Generator().manual_seed(5)).to(torch.bfloat16).to(cuda)
    tape.backward()
    torch.cuda.synchronize()
    res[fold] = (out.data.float().cpu(), {p.name: p.grad.float().cpu().clone() for p in store.params})
  assert len(calls) >= 1, "no one-tap separable branch in the stack"
  ya, yb = res[True][0], res[False][0]
  rel = float((ya - yb).norm() / yb.norm())
  assert rel <= 2e-2, rel       # (measured 1.1e-2: one bf16 rounding moved per branch, five BatchNorm layers behind it)
  worst = (0.0, "")
  for n, gu in res[False][1].items():
    gf = res[True][1][n]
    assert gu.abs().sum() > 0 or gf.abs().sum() == 0, n
    r = float((gf - gu).norm() / (gu.norm() + 1e-20))
    cos = float(torch.nn.functional.cosine_similarity(gf.flatten(), gu.flatten(), dim=0))
    worst = max(worst, (r, n))
    # (the forward passes differ by 1e-2: ReLU masks and BatchNorm statistics of five layers move with it; the
    # layer itself is held to fp64 in test_one_tap_separable_layer_math)
    assert cos >= 0.97 and r <= 0.25, (n, cos, r)
  print("one-tap separable branches folded (ragged=%s): %d folds, output difference %.2e, worst gradient difference "
        "%.2e (%s)" % (ragged, len(calls), rel, worst[0], worst[1]))

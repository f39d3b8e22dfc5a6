"""Layer-by-layer, teacher-forced GPU parity of the FULL-WIDTH Jasper 10x5 Dense-Residual encoder
(openseq2seq_amd/configs/jasper.py = jasper10x5_LibriSpeech_nvgrad_masks.py:58-147).

tests/test_jasper_full_size_gpu.py compares whole-network outputs, where a randomly initialised
53-layer BatchNorm stack amplifies bf16 rounding to ~7e-2 and the tolerances have to be that
loose. Here every one of the 53 conv_bn_actv / conv_bn_res_bn_actv calls
(parts/cnns/conv_blocks.py:61-232) is checked ON ITS OWN at full channel width (256 ... 1024
channels, K = 11 ... 29, dilation 2, stride 2, up to 11 dense-residual 1x1 branches): the device
network runs forward once on a ragged batch; for each layer the tensors the device layer actually
consumed (main input + residual inputs, bf16, masked) and a random upstream gradient are fed to
BOTH the device layer and the oracle layer (oracle/tdnn.py:tdnn_layer, fp32 math on the same
bf16-rounded weights), so no error is inherited from earlier layers and a wrong tap, a wrong
window, a mis-masked row or a mis-reduced BatchNorm statistic in any shape-specialised kernel
path (ping-pong 256x256 tiles, lockstep 128x128 tiles, grouped 1x1 launches, K = 1 / stride-2
weight gradients, dead and half-dead 128-row windows) shows at full size.

Tolerances (relative L2 over the live rows; bf16 storage of the conv output, of dz and of the
results is the only difference between the two sides):
  * layer output                          <= 2e-3   (measured worst 8.3e-5)
  * d(main input), d(residual inputs)     <= 6e-3   (measured worst 2.9e-3; cosine >= 0.9999)
  * d(kernel) of every branch             <= 6e-3   (measured worst 2.3e-3; cosine >= 0.9999)
  * d(gamma), d(beta) of every branch     <= 1e-2   (measured worst 2.3e-3; sums of ~1e5
                                                     bf16-rounded products that cancel)
The oracle emulates the device's bf16 STORAGE points (conv output, dz), which is what makes these
bounds reachable; the distance to the plain fp32 oracle is reported next to them (outputs 2.6e-3)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
  return float((a - b).norm() / (b.norm() + 1e-20))


def _cos(a, b):
  return float(torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0))


def test_jasper10x5_every_layer_teacher_forced(cuda, monkeypatch):
  from openseq2seq_amd.configs.jasper import jasper_convnet_layers
  from openseq2seq_amd.optimizers.flat_params import FlatParams
  from openseq2seq_amd.encoders.tdnn_encoder import TDNNEncoder
  from openseq2seq_amd.parts.cnns.conv_blocks import Act, Tape, conv_bn_res_bn_actv
  from oracle import cnn, tdnn
  torch.manual_seed(0)
  cfg_layers = [dict(l, dropout_keep_prob=1.0) for l in jasper_convnet_layers()]
  store = FlatParams(cuda)
  enc = TDNNEncoder({"convnet_layers": cfg_layers, "dropout_keep_prob": 1.0, "activation_fn": "relu",
                     "use_conv_mask": True, "dtype": "mixed"}, None, mode="train").build(store, 64)
  store.finalize()
  g = torch.Generator().manual_seed(1)
  B, T = 4, 600
  lens0 = torch.tensor([600, 452, 300, 130], dtype=torch.int32)
  x0 = torch.randn(B, T, 64, generator=g).to(torch.bfloat16)
  prefix = "ForwardPass/w2l_encoder/"
  nl = len(enc._layers)
  assert nl == 53        # 1 + 10 blocks x 5 + 1 + 1

  def oracle_weights(names):
    w = {}
    for n in names:
      p = store.by_name(prefix + n)
      w[n] = (p.w16.float().cpu().permute(0, 2, 1).contiguous() if p.kind == "conv"
              else p.master.cpu().clone()).requires_grad_(True)
    return w

  # the device chain supplies realistic inputs for every layer (post-ReLU, masked, bf16)
  x = Act(x0.to(cuda), lens0.to(cuda), requires_grad=False)
  src_len = lens0.clone()
  res_agg, layer_res = [], []
  lens_dev, grouped_seen = None, []
  from openseq2seq_amd import capi
  orig_grouped = capi.conv1x1_fwd_grouped

  def counting_grouped(items, **kw):
    grouped_seen.append(len(items))
    return orig_grouped(items, **kw)
  monkeypatch.setattr(capi, "conv1x1_fwd_grouped", counting_grouped)
  worst = {"out": (0.0, ""), "dx": (0.0, ""), "dw": (0.0, ""), "dbn": (0.0, "")}
  worst_fp32 = 0.0
  for li, L in enumerate(enc._layers):
    blk, main = L["cfg"], L["main"]
    if L["rep"] == 0 and blk.get("residual", False):
      res_agg.append(x)
      layer_res = list(res_agg)
    s = blk["stride"][0]
    src_len = (src_len + s - 1) // s
    last = li == nl - 1
    name = "conv%d%d" % (L["block"] + 1, L["rep"] + 1)
    # ---- device layer on fresh activation wrappers (nothing accumulates across layers) ----------
    store.zero_grads()
    xin = Act(x.data, x.lens, requires_grad=li > 0)   # the feature tensor needs no gradient
    rin = [Act(r.data, r.lens, requires_grad=True) for r in layer_res] if L["res"] else []
    tape = Tape()
    if s > 1 or li == 0:        # ONE length tensor per resolution, as the encoder has: the grouped
      lens_dev = src_len.to(cuda)  # 1x1 launches require the branches to share their length vector
    out = conv_bn_res_bn_actv(main, L["res"], xin, rin, lens_dev, "relu", True, tape, keep_prob=1.0,
                              seed=li, mask_output=not last)
    dy = torch.randn(out.data.shape, generator=g).to(torch.bfloat16)
    out.grad = dy.to(cuda)
    tape.backward()
    torch.cuda.synchronize()
    # ---- oracle layer on the same tensors ---------------------------------------------------
    names = [name + "/kernel", name + "/bn/gamma", name + "/bn/beta"]
    for i in range(len(L["res"])):
      names += [name + "/res_%d/kernel" % i, name + "/res_bn_%d/gamma" % i, name + "/res_bn_%d/beta" % i]
    Tin, Tout = x.data.shape[1], out.data.shape[1]
    in_len = x.lens.cpu() if x.lens is not None else torch.full((B,), Tin)
    in_mask = cnn.seq_mask(in_len, Tin)
    out_mask = cnn.seq_mask(src_len, Tout)
    w = oracle_weights(names)
    xo = x.data.float().cpu().requires_grad_(True)
    ro = [r.data.float().cpu().requires_grad_(True) for r in layer_res] if L["res"] else []
    # the encoder masks every conv input (tdnn_encoder.py:185-186,204-205); the inputs already are
    yo = tdnn.tdnn_layer(xo * in_mask, [r * out_mask for r in ro], blk, name, w,
                         None if last else out_mask, "relu", 1e-3, None, 1.0, True)
    (yo * dy.float()).sum().backward()
    with torch.no_grad():     # the plain fp32 oracle (no storage emulation): reported only
      y32 = tdnn.tdnn_layer(xo * in_mask, [r * out_mask for r in ro], blk, name, w,
                            None if last else out_mask, "relu", 1e-3, None, 1.0, False)
    yo, dxo, dro = yo.detach(), xo.grad, [r.grad for r in ro]
    live_out = out_mask.bool().expand_as(yo) if not last else torch.ones_like(yo, dtype=torch.bool)
    got = out.data.float().cpu()
    tag = "%s (layer %d, %d->%d K=%d s=%d d=%d, %d residual branches)" % (
        name, li, main.cin, main.cout, main.k, main.stride, main.dil, len(L["res"]))
    r = _rel(got[live_out], yo[live_out])
    worst["out"] = max(worst["out"], (r, tag))
    worst_fp32 = max(worst_fp32, _rel(got[live_out], y32[live_out]))
    assert r <= 2e-3, ("output", tag, r)
    if not last:    # masked rows are exact zeros on both sides
      assert float(got[~live_out].abs().max() if (~live_out).any() else 0.0) == 0.0, tag
    if li > 0:      # the first layer's input is the feature tensor: no data gradient
      live_in = in_mask.bool().expand_as(dxo)
      gx = xin.grad.float().cpu()
      r, c = _rel(gx[live_in], dxo[live_in]), _cos(gx[live_in], dxo[live_in])
      worst["dx"] = max(worst["dx"], (r, tag))
      assert r <= 6e-3 and c >= 0.9999, ("d(main input)", tag, r, c)
    for i, (ra, rg) in enumerate(zip(rin, dro)):
      live_r = out_mask.bool().expand_as(rg)
      gr = ra.grad.float().cpu()
      r, c = _rel(gr[live_r], rg[live_r]), _cos(gr[live_r], rg[live_r])
      worst["dx"] = max(worst["dx"], (r, tag + " res_%d" % i))
      assert r <= 6e-3 and c >= 0.9999, ("d(residual input %d)" % i, tag, r, c)
    for n in names:
      p = store.by_name(prefix + n)
      ref = w[n].grad.permute(0, 2, 1) if p.kind == "conv" else w[n].grad
      gp = p.grad.float().cpu()
      r, c = _rel(gp, ref), _cos(gp, ref)
      if p.kind == "conv":
        worst["dw"] = max(worst["dw"], (r, tag + " " + n))
        assert r <= 6e-3 and c >= 0.9999, ("d(kernel)", tag, n, r, c)
      else:
        worst["dbn"] = max(worst["dbn"], (r, tag + " " + n))
        assert r <= 1e-2, ("d(gamma/beta)", tag, n, r, c)
    x = Act(out.data, None if last else lens_dev, requires_grad=False)
  # the grouped 1x1 launches really ran: forward + data gradient at every block end with >= 2
  # dense-residual inputs (blocks 3..11: 2..10 branches)
  assert sorted(grouped_seen) == sorted(list(range(2, 11)) * 2), grouped_seen
  print("jasper10x5 layer by layer (rel-L2, device vs bf16-storage oracle): worst", worst,
        "| worst output vs plain fp32 oracle %.3e" % worst_fp32)


def _oracle_weights(store, prefix, names):
  w = {}
  for n in names:
    p = store.by_name(prefix + n)
    w[n] = (p.w16.float().cpu().permute(0, 2, 1).contiguous() if p.kind == "conv"
            else p.master.cpu().clone()).requires_grad_(True)
  return w


def _layer_names(name, nres):
  names = [name + "/kernel", name + "/bn/gamma", name + "/bn/beta"]
  for i in range(nres):
    names += [name + "/res_%d/kernel" % i, name + "/res_bn_%d/gamma" % i, name + "/res_bn_%d/beta" % i]
  return names


def test_jasper10x5_layer_pairs_fused_bn_backward(cuda, monkeypatch):
  """The BatchNorm-backward reduction of a single-input conv + BN + ReLU layer rides in the epilogue of
  the NEXT layer's data gradient (capi.conv1d_dgrad_bnact -> os2s_conv1d_dgrad_bnact_ws, finished by
  os2s_bn_bwd_finalize_raw; parts/cnns/conv_blocks.py ConvBN.backward_branch(final=True)): 41 of the
  53 layers of Jasper 10x5. That path only exists when layer L+1 consumes layer L's REAL output Act
  (it carries bn_y), so the one-layer test above never takes it. Here every consecutive pair
  (L, L+1) with L a plain conv_bn_actv layer — repeats 1-4 of the ten blocks, i.e. 384 ... 896 channels on
  the 256 x 256 ping-pong tile, K = 13 ... 29, and the dilated 896-channel layer in front of the 1 x 1
  1024-channel one — runs as a two-layer device graph on a ragged batch against the composition of two
  oracle layers (oracle/tdnn.py:tdnn_layer with bf16-storage emulation), teacher-forced with the tensors
  the device network produced. Asserted per pair: the fused call happened; layer L's d(gamma), d(beta),
  d(kernel) and d(input) and layer L+1's parameter gradients within the one-layer test's bounds
  (6e-3 / 1e-2 rel-L2). The accumulate = True flavour of the fused epilogue (layer L's output has a second,
  earlier-finishing consumer) is test_fused_bn_backward_accumulates_into_a_gradient_with_two_consumers."""
  from openseq2seq_amd import capi
  from openseq2seq_amd.configs.jasper import jasper_convnet_layers
  from openseq2seq_amd.optimizers.flat_params import FlatParams
  from openseq2seq_amd.encoders.tdnn_encoder import TDNNEncoder
  from openseq2seq_amd.parts.cnns.conv_blocks import Act, Tape, conv_bn_res_bn_actv
  from openseq2seq_amd.parts.cnns import conv_blocks
  from oracle import cnn, tdnn
  torch.manual_seed(0)
  cfg_layers = [dict(l, dropout_keep_prob=1.0) for l in jasper_convnet_layers()]
  store = FlatParams(cuda)
  enc = TDNNEncoder({"convnet_layers": cfg_layers, "dropout_keep_prob": 1.0, "activation_fn": "relu",
                     "use_conv_mask": True, "dtype": "mixed"}, None, mode="train").build(store, 64)
  store.finalize()
  g = torch.Generator().manual_seed(5)
  B, T = 4, 600
  lens0 = torch.tensor([600, 452, 300, 130], dtype=torch.int32)
  x0 = torch.randn(B, T, 64, generator=g).to(torch.bfloat16)
  prefix = "ForwardPass/w2l_encoder/"
  nl = len(enc._layers)
  fused_calls = []
  orig = capi.conv1d_dgrad_bnact

  def counting(dy, wt, g_, **kw):
    fused_calls.append((tuple(dy.shape), tuple(g_.shape), bool(kw.get("accumulate", False))))
    return orig(dy, wt, g_, **kw)
  monkeypatch.setattr(capi, "conv1d_dgrad_bnact", counting)

  # ---- device chain: realistic inputs for every layer (post-ReLU, masked, bf16) ------------------
  x = Act(x0.to(cuda), lens0.to(cuda), requires_grad=False)
  src_len = lens0.clone()
  res_agg, layer_res, lens_dev = [], [], None
  layer_in = []          # per layer: (input Act data, input lens, residual datas, out lens (host), lens_dev)
  for li, L in enumerate(enc._layers):
    blk = L["cfg"]
    if L["rep"] == 0 and blk.get("residual", False):
      res_agg.append(x)
      layer_res = list(res_agg)
    s = blk["stride"][0]
    src_len = (src_len + s - 1) // s
    if s > 1 or li == 0:
      lens_dev = src_len.to(cuda)
    last = li == nl - 1
    rin = [Act(r.data, r.lens, requires_grad=False) for r in layer_res] if L["res"] else []
    layer_in.append(dict(x=x, res=list(layer_res) if L["res"] else [], src_len=src_len.clone(), lens_dev=lens_dev))
    out = conv_bn_res_bn_actv(L["main"], L["res"], Act(x.data, x.lens, requires_grad=False), rin, lens_dev,
                              "relu", True, None, keep_prob=1.0, seed=li, mask_output=not last)
    x = Act(out.data, None if last else lens_dev, requires_grad=False)
  torch.cuda.synchronize()

  pairs = [li for li, L in enumerate(enc._layers[:-1])
           if li > 0 and not L["res"] and L["main"].stride == 1 and enc._layers[li + 1]["main"].stride == 1]
  assert len(pairs) == 41, len(pairs)
  worst = {"dbn": (0.0, ""), "dw": (0.0, ""), "dx": (0.0, ""), "out": (0.0, "")}
  for li in pairs:
    La, Lb = enc._layers[li], enc._layers[li + 1]
    ia, ib = layer_in[li], layer_in[li + 1]
    last_b = li + 1 == nl - 1
    na = "conv%d%d" % (La["block"] + 1, La["rep"] + 1)
    nb = "conv%d%d" % (Lb["block"] + 1, Lb["rep"] + 1)
    # ---- device: two layers on ONE tape, layer b consumes layer a's real output -----------------
    store.zero_grads()
    n_before = len(fused_calls)
    xin = Act(ia["x"].data, ia["x"].lens, requires_grad=True)
    rin = [Act(r.data, r.lens, requires_grad=True) for r in ib["res"]]
    tape = Tape()
    ya = conv_bn_res_bn_actv(La["main"], [], xin, [], ia["lens_dev"], "relu", True, tape, keep_prob=1.0, seed=li)
    assert ya.bn_y is not None, na
    yb = conv_bn_res_bn_actv(Lb["main"], Lb["res"], ya, rin, ib["lens_dev"], "relu", True, tape, keep_prob=1.0,
                             seed=li + 1, mask_output=not last_b)
    dy = torch.randn(yb.data.shape, generator=g).to(torch.bfloat16)
    yb.grad = dy.to(cuda)
    tape.backward()
    torch.cuda.synchronize()
    if conv_blocks.FUSE_BN_BWD:      # (OS2S_FUSE_BN_BWD=0: the A/B run of the separate reduction pass)
      assert len(fused_calls) == n_before + 1, (na, nb, fused_calls[n_before:])
      assert fused_calls[-1][1] == tuple(ya.data.shape) and fused_calls[-1][2] is False
    # ---- oracle: the same two layers composed --------------------------------------------------
    names_a, names_b = _layer_names(na, 0), _layer_names(nb, len(Lb["res"]))
    w = _oracle_weights(store, prefix, names_a + names_b)
    Ta = ia["x"].data.shape[1]
    in_len = ia["x"].lens.cpu()
    in_mask = cnn.seq_mask(in_len, Ta)
    mask_a = cnn.seq_mask(ia["src_len"], ya.data.shape[1])
    mask_b = cnn.seq_mask(ib["src_len"], yb.data.shape[1])
    xo = ia["x"].data.float().cpu().requires_grad_(True)
    ro = [r.data.float().cpu().requires_grad_(True) for r in ib["res"]]
    oa = tdnn.tdnn_layer(xo * in_mask, [], La["cfg"], na, w, mask_a, "relu", 1e-3, None, 1.0, True)
    # teacher forcing at the layer boundary: layer L+1 of the oracle consumes the DEVICE's activation (the
    # gradient still flows into the oracle's layer L). Without it the two sides' second layers see inputs
    # that differ by single bf16 ulps, a few pre-activations within that noise of zero flip their ReLU mask,
    # and every flipped position carries a full-size gradient error (measured 0.9 - 1.9e-2 rel-L2 on EVERY
    # gradient, identical for the fused and the unfused path): that is sensitivity of the function, not of
    # the kernels under test
    dev_a = ya.data.float().cpu()
    assert _rel(dev_a, oa.detach()) <= 2e-3
    oa = oa + (dev_a - oa).detach()
    ob = tdnn.tdnn_layer(oa * mask_a, [r * mask_b for r in ro], Lb["cfg"], nb, w, None if last_b else mask_b,
                         "relu", 1e-3, None, 1.0, True)
    (ob * dy.float()).sum().backward()
    tag = "%s -> %s (layers %d, %d: %d->%d K=%d d=%d, then %d->%d K=%d d=%d, %d residual branches)" % (
        na, nb, li, li + 1, La["main"].cin, La["main"].cout, La["main"].k, La["main"].dil, Lb["main"].cin,
        Lb["main"].cout, Lb["main"].k, Lb["main"].dil, len(Lb["res"]))
    live_b = mask_b.bool().expand_as(ob) if not last_b else torch.ones_like(ob, dtype=torch.bool)
    r = _rel(yb.data.float().cpu()[live_b], ob.detach()[live_b])
    worst["out"] = max(worst["out"], (r, tag))
    assert r <= 2e-3, ("output", tag, r)
    live_in = in_mask.bool().expand_as(xo)
    gx = xin.grad.float().cpu()
    r, c = _rel(gx[live_in], xo.grad[live_in]), _cos(gx[live_in], xo.grad[live_in])
    worst["dx"] = max(worst["dx"], (r, tag))
    assert r <= 6e-3 and c >= 0.9999, ("d(input of layer L)", tag, r, c)
    for n in names_a + names_b:
      p = store.by_name(prefix + n)
      ref = w[n].grad.permute(0, 2, 1) if p.kind == "conv" else w[n].grad
      gp = p.grad.float().cpu()
      r, c = _rel(gp, ref), _cos(gp, ref)
      if p.kind == "conv":
        worst["dw"] = max(worst["dw"], (r, tag + " " + n))
        assert r <= 6e-3 and c >= 0.9999, ("d(kernel)", tag, n, r, c)
      else:
        worst["dbn"] = max(worst["dbn"], (r, tag + " " + n))
        assert r <= 1e-2, ("d(gamma/beta)", tag, n, r, c)
  print("jasper10x5 layer pairs through the fused dgrad + BatchNorm-backward epilogue (rel-L2 vs bf16-storage "
        "oracle): worst", worst)


# gradients that passed through up to three bf16-stored BatchNorm backward stages
GRAD_TOL_3LAYER = 8e-3


@pytest.mark.parametrize("pp", [-1, 12, 13])
@pytest.mark.parametrize("C,K,dil", [(640, 21, 1), (768, 13, 2), (384, 13, 1)])
def test_fused_bn_backward_accumulates_into_a_gradient_with_two_consumers(cuda, monkeypatch, C, K, dil, pp):
  """pp: option conv1d.variant for the whole pass (-1 = default, 12 / 13 = the narrow ping-pong tiles of
  2 / 3 windows x 128 columns wherever a layer fits them — the fused epilogue on the straddling wave layout).
  accumulate = True in the fused epilogue: layer A (plain conv + BN + ReLU, stride 1) feeds a residual
  block of two repeats — its first repeat's main convolution AND the block end's 1 x 1 residual branch.
  The residual branch's data gradient reaches A's output gradient first (the block end runs first in
  backward); the first repeat's data gradient is the LAST contribution and adds
  (dgrad) onto it before the ReLU mask and the BatchNorm partial sums are taken (`accumulate=inp.grad_init`).
  Checked against three composed oracle layers at ping-pong widths, ragged."""
  from openseq2seq_amd import capi
  from openseq2seq_amd.optimizers.flat_params import FlatParams
  from openseq2seq_amd.encoders.tdnn_encoder import TDNNEncoder
  from openseq2seq_amd.parts.cnns.conv_blocks import Tape
  from oracle import cnn, tdnn
  torch.manual_seed(3)
  layers = [
      {"type": "conv1d", "repeat": 1, "kernel_size": [K], "stride": [1], "num_channels": C, "padding": "SAME",
       "dilation": [dil], "dropout_keep_prob": 1.0},
      {"type": "conv1d", "repeat": 2, "kernel_size": [K], "stride": [1], "num_channels": C, "padding": "SAME",
       "dilation": [dil], "dropout_keep_prob": 1.0, "residual": True, "residual_dense": True},
  ]
  store = FlatParams(cuda)
  enc = TDNNEncoder({"convnet_layers": layers, "dropout_keep_prob": 1.0, "activation_fn": "relu",
                     "use_conv_mask": True, "dtype": "mixed"}, None, mode="train").build(store, C)
  store.finalize()
  g = torch.Generator().manual_seed(9)
  B, T = 5, 330
  lens = torch.tensor([330, 257, 256, 129, 40], dtype=torch.int32)
  x0 = (torch.randn(B, T, C, generator=g) * cnn.seq_mask(lens, T)).to(torch.bfloat16)
  calls = []
  orig = capi.conv1d_dgrad_bnact

  def counting(dy, wt, g_, **kw):
    calls.append(bool(kw.get("accumulate", False)))
    return orig(dy, wt, g_, **kw)
  monkeypatch.setattr(capi, "conv1d_dgrad_bnact", counting)
  dev_acts = []
  from openseq2seq_amd.encoders import tdnn_encoder as te
  orig_block = te.conv_bn_res_bn_actv

  def recording_block(*a, **kw):
    r = orig_block(*a, **kw)
    dev_acts.append(r.data.float().cpu())
    return r
  monkeypatch.setattr(te, "conv_bn_res_bn_actv", recording_block)
  store.zero_grads()
  tape = Tape()
  from openseq2seq_amd import _lib
  _lib.set_option("conv1d.variant", pp)
  try:
    e = enc.encode({"source_tensors": [x0.to(cuda), lens.to(cuda)], "tape": tape, "seed": 3})
    out = e["outputs_act"]
    dy = torch.randn(out.data.shape, generator=g).to(torch.bfloat16)
    out.grad = dy.to(cuda)
    tape.backward()
    torch.cuda.synchronize()
  finally:
    _lib.set_option("conv1d.variant", -1)
  # two fused calls: block end -> repeat 1's output (fresh), repeat 1 -> layer A's output (accumulating)
  from openseq2seq_amd.parts.cnns import conv_blocks
  assert calls == ([False, True] if conv_blocks.FUSE_BN_BWD else []), calls
  prefix = "ForwardPass/w2l_encoder/"
  names = _layer_names("conv11", 0) + _layer_names("conv21", 0) + _layer_names("conv22", 1)
  w = _oracle_weights(store, prefix, names)
  m = cnn.seq_mask(lens, T)
  xo = x0.float().requires_grad_(True)
  # teacher forcing at both layer boundaries (see the pair test): the oracle's later layers consume the
  # device's activations, gradients flow through the oracle's graph
  a = tdnn.tdnn_layer(xo * m, [], layers[0], "conv11", w, m, "relu", 1e-3, None, 1.0, True)
  assert _rel(dev_acts[0], a.detach()) <= 2e-3
  a = a + (dev_acts[0] - a).detach()
  r1 = tdnn.tdnn_layer(a * m, [], layers[1], "conv21", w, m, "relu", 1e-3, None, 1.0, True)
  assert _rel(dev_acts[1], r1.detach()) <= 2e-3
  r1 = r1 + (dev_acts[1] - r1).detach()
  r2 = tdnn.tdnn_layer(r1 * m, [a * m], layers[1], "conv22", w, None, "relu", 1e-3, None, 1.0, True)
  (r2 * dy.float()).sum().backward()
  assert _rel(out.data.float().cpu(), r2.detach()) <= 2e-3
  report = []
  for n in names:
    p = store.by_name(prefix + n)
    ref = w[n].grad.permute(0, 2, 1) if p.kind == "conv" else w[n].grad
    gp = p.grad.float().cpu()
    report.append((n, round(_rel(gp, ref), 5), round(_cos(gp, ref), 6), p.kind))
  print("fused accumulate (C=%d K=%d dil=%d):" % (C, K, dil), report)
  bad = [r for r in report if r[1] > (GRAD_TOL_3LAYER if r[3] == "conv" else 2 * GRAD_TOL_3LAYER) or r[2] < 0.9998]
  assert not bad, bad

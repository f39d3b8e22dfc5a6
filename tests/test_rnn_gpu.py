"""GPU parity of the recurrent layers (forward + BPTT) vs the CPU fp32 oracle:
cuDNN-form GRU (DS2), cuDNN-form LSTM (Tacotron2 encoder), TF LSTMCell (NMT / Tacotron2
decoder), each in both directions with ragged lengths. bf16 weights/activations through T
recurrent steps: outputs rtol/atol 3e-2; gradients cosine >= 0.99, rel-L2 <= 0.1."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import rnn as orn  # noqa: E402


def _cmp(got, ref, name, cos_min=0.99, rel_max=0.1):
  got, ref = got.float().cpu().flatten(), ref.detach().flatten()
  cos = float(torch.nn.functional.cosine_similarity(got, ref, dim=0))
  rel = float((got - ref).norm() / (ref.norm() + 1e-12))
  assert cos > cos_min and rel < rel_max, (name, cos, rel)


@pytest.mark.parametrize("cell", ["gru_cudnn", "lstm_cudnn", "lstm_tf"])
@pytest.mark.parametrize("reverse", [False, True])
@pytest.mark.parametrize("use_lens", [True, False])
@pytest.mark.parametrize("shape", [(5, 23, 64, 96), (130, 5, 64, 1024)])
def test_rnn_direction(cuda, cell, reverse, use_lens, shape):
  """One recurrent direction fwd + bwd vs the fp32 oracle. The second shape (1024 units, 130
  samples) takes the 32-rows-per-workgroup backward step kernel and a reduction longer than one
  round of loads; the first the 8-row one."""
  from openseq2seq_amd.optimizers.flat_params import FlatParams
  from openseq2seq_amd.parts.rnns.rnn_layers import RNNDirection
  from openseq2seq_amd.parts.cnns.conv_blocks import Act, Tape
  torch.manual_seed(0)
  B, T, In, H = shape
  store = FlatParams(cuda)
  layer = RNNDirection(store, "rnn", cell, [In], H, reverse=reverse, forget_bias=1.0)
  store.finalize()
  g = torch.Generator().manual_seed(3)
  for p in store.params:            # non-trivial biases
    if p.kind == "vector":
      p.master.copy_((torch.randn(p.shape, generator=g) * 0.1).to(cuda))
  x = (torch.randn(B, T, In, generator=g)).to(torch.bfloat16)
  lens = None
  if use_lens:
    lens = torch.tensor([23, 7, 15, 1, 20], dtype=torch.int32) if B == 5 else \
        torch.randint(1, T + 1, (B,), generator=g).to(torch.int32)
  dy = torch.randn(B, T, H, generator=g).to(torch.bfloat16)
  xa = Act(x.to(cuda), None)
  tape = Tape()
  store.zero_grads()
  out = layer.forward([xa], lens.to(cuda) if use_lens else None, tape)
  out.grad = dy.to(cuda)
  tape.backward()
  torch.cuda.synchronize()
  # ---- oracle ---------------------------------------------------------------
  G = layer.G
  wx = layer.wx[0].w16.float().cpu()[0].clone().requires_grad_(True)    # [GH, In]
  wh = layer.wh.w16.float().cpu()[0].clone().requires_grad_(True)       # [GH, H]
  bx = layer.bx.master.cpu().clone().requires_grad_(True)
  xf = x.float().requires_grad_(True)
  if cell == "lstm_tf":
    ref = orn.lstm_tf(xf, lens, wx.t(), wh.t(), bx, 1.0, reverse)
    bh = None
  else:
    bh = layer.bh.master.cpu().clone().requires_grad_(True)
    ref = orn.cudnn_rnn("gru" if cell == "gru_cudnn" else "lstm", xf, lens, wx, wh, bx, bh, reverse)
  dyf = dy.float()
  if use_lens:
    m = (torch.arange(T)[None, :] < lens[:, None]).float()[:, :, None]
    dyf = dyf * m
  ref.backward(dyf)
  y = out.data.float().cpu()
  scale = float(ref.detach().pow(2).mean().sqrt())
  torch.testing.assert_close(y, ref.detach(), rtol=3e-2, atol=3e-2 * scale)
  _cmp(xa.grad, xf.grad, "dx")
  _cmp(layer.wx[0].grad[0], wx.grad, "dwx")
  _cmp(layer.wh.grad[0], wh.grad, "dwh")
  _cmp(layer.bx.grad, bx.grad, "dbx")
  if bh is not None:
    _cmp(layer.bh.grad, bh.grad, "dbh")


def test_birnn_stack_vs_oracle(cuda):
  """2-layer bidirectional cuDNN-form GRU with one shared [B,T,2H] output tensor per layer
  (DS2 stacking): outputs and input gradient vs torch.nn.GRU(bidirectional, 2 layers)."""
  from openseq2seq_amd.optimizers.flat_params import FlatParams
  from openseq2seq_amd.parts.rnns.rnn_layers import BiRNNStack
  from openseq2seq_amd.parts.cnns.conv_blocks import Act, Tape
  torch.manual_seed(1)
  B, T, In, H = 4, 30, 128, 96
  store = FlatParams(cuda)
  stack = BiRNNStack(store, "rnn", "gru_cudnn", In, H, 2, bidirectional=True)
  store.finalize()
  g = torch.Generator().manual_seed(5)
  xh = torch.randn(B, T, In, generator=g).to(torch.bfloat16)
  x = Act(xh.to(cuda), None)
  tape = Tape()
  out = stack.forward(x, None, tape)
  assert tuple(out.data.shape) == (B, T, 2 * H)
  dy = torch.randn(B, T, 2 * H, generator=g).to(torch.bfloat16)
  out.grad = dy.to(cuda)
  store.zero_grads()
  tape.backward()
  torch.cuda.synchronize()
  ref = torch.nn.GRU(In, H, num_layers=2, batch_first=True, bidirectional=True)
  with torch.no_grad():
    for l, dirs in enumerate(stack.layers):
      for d, layer in enumerate(dirs):
        sfx = "_l%d%s" % (l, "_reverse" if d else "")
        getattr(ref, "weight_ih" + sfx).copy_(layer.wx[0].w16.float().cpu()[0])
        getattr(ref, "weight_hh" + sfx).copy_(layer.wh.w16.float().cpu()[0])
        getattr(ref, "bias_ih" + sfx).copy_(layer.bx.master.cpu())
        getattr(ref, "bias_hh" + sfx).copy_(layer.bh.master.cpu())
  xf = xh.float().requires_grad_(True)
  yr, _ = ref(xf)
  yr.backward(dy.float())
  scale = float(yr.detach().pow(2).mean().sqrt())
  torch.testing.assert_close(out.data.float().cpu(), yr.detach(), rtol=3e-2, atol=3e-2 * scale)
  _cmp(x.grad, xf.grad, "dx")
  l0 = stack.layers[0][1]
  _cmp(l0.wh.grad[0], ref.weight_hh_l0_reverse.grad, "dwh l0 reverse")
  _cmp(stack.layers[1][0].wx[0].grad[0], ref.weight_ih_l1.grad, "dwx l1")


@pytest.mark.parametrize("B,T,H", [(16, 120, 800), (4, 37, 800), (32, 30, 416), (7, 50, 96)])
def test_gru_persistent_xcd_kernel_equals_step_launches(cuda, B, T, H):
  """csrc/rnn_xcd.hip (one persistent launch per layer: a direction per XCD, weights in registers,
  the hidden state exchanged through that XCD's L2) against csrc/rnn.hip (one launch per time step),
  both directions, ragged lengths, at the DeepSpeech2 layer size (800 units, ds2_large_8gpus.py:63-68)
  and at sizes with padding in every dimension (B = 7, H = 96 / 416, odd T). The two paths sum the
  800-deep reduction in a different order and use a different tanh approximation: outputs and saved
  gates agree to bf16 rounding (max |diff| <= 2 bf16 ulp of 1.0 = 1.6e-2, rel-L2 <= 3e-3); the
  oracle comparison of both paths lives in test_rnn_direction / test_ds2_gpu."""
  from openseq2seq_amd import capi, _lib
  g = torch.Generator().manual_seed(B * 1000 + T + H)
  bf = lambda t: t.to(torch.bfloat16).to(cuda)
  lens = torch.randint(1, T + 1, (B,), generator=g, dtype=torch.int32)
  lens[0] = T
  dirs = [dict(gx=bf(torch.randn(B, T, 3 * H, generator=g) * 0.5),
               wh=bf(torch.randn(3 * H, H, generator=g) * H ** -0.5),
               bh=(torch.randn(3 * H, generator=g) * 0.1).to(cuda), reverse=bool(d)) for d in range(2)]
  L = _lib.lib()
  res = {}
  try:
    for mode in (0, 1):
      L.os2s_gru_xcd_set_mode(mode)
      res[mode] = capi.rnn_layer_fwd_multi(capi.CELL_GRU_CUDNN, [dict(d) for d in dirs], lens.to(cuda), H)
      torch.cuda.synchronize()
    # a second persistent launch right behind the first: the exchange buffers are reset per launch
    again = capi.rnn_layer_fwd_multi(capi.CELL_GRU_CUDNN, [dict(d) for d in dirs], lens.to(cuda), H)
    torch.cuda.synchronize()
  finally:
    L.os2s_gru_xcd_set_mode(-1)
  m = (torch.arange(T)[None, :] < lens[:, None])[:, :, None].to(cuda)
  for d in range(2):
    for i, name in ((0, "y"), (1, "gates")):
      live = lambda t: torch.where(m, t.float(), torch.zeros((), device=cuda))   # gates past the ends: never written
      a, b = live(res[0][d][i]), live(res[1][d][i])
      assert float((a - b).abs().max()) <= 1.6e-2, (d, name, float((a - b).abs().max()))
      assert float((a - b).norm() / (a.norm() + 1e-20)) <= 3e-3, (d, name)
      assert torch.equal(b, live(again[d][i])), (d, name)   # run-to-run bit-identical (live rows)
    # rows past the sequence ends stay zero
    assert float((res[1][d][0].float() * (~m)).abs().max()) == 0.0
  # ---- backward through time on the saved activations of the persistent forward (B <= 16: the
  #      persistent backward kernel; B = 32 compares the step path with itself) -----------------------
  dys = [bf(torch.randn(B, T, H, generator=g)) for _ in range(2)]
  bw = {}
  try:
    for mode in (0, 1):
      L.os2s_gru_xcd_set_mode(mode)
      bw[mode] = capi.rnn_layer_bwd_multi(
          capi.CELL_GRU_CUDNN,
          [dict(whT=dirs[d]["wh"].t().contiguous(), dy=dys[d], y=res[1][d][0], gates=res[1][d][1],
                reverse=bool(d)) for d in range(2)], lens.to(cuda), H)
      torch.cuda.synchronize()
  finally:
    L.os2s_gru_xcd_set_mode(-1)
  for d in range(2):
    for i, name in ((0, "dgx"), (1, "dgr")):
      a, b = bw[0][d][i].float(), bw[1][d][i].float()
      assert bool(torch.isfinite(b).all()), (d, name)
      assert float((a * (~m)).abs().max()) == 0.0 and float((b * (~m)).abs().max()) == 0.0
      rel = float((a - b).norm() / (a.norm() + 1e-20))
      assert rel <= 1e-2, (d, name, rel)       # bf16 gate gradients through T steps, two summation orders


@pytest.mark.parametrize("cell", ["gru_cudnn", "lstm_tf"])
@pytest.mark.parametrize("reverse", [False, True])
@pytest.mark.parametrize("use_lens", [True, False])
def test_recurrent_weight_gradient_gemm_path_equals_shifted_window_path(cuda, monkeypatch, cell, reverse, use_lens):
  """dWh = sum_t dgr_t^T h_{t-1} two ways: the shifted-window K = 1 convolution gradient (small
  batches) and the plain TN GEMM over a time-shifted copy of y (B x T >= 4096 rows: DeepSpeech2).
  Same products, different summation order: rel-L2 <= 1e-3; ragged lengths and both directions
  (the first processed step of a reversed sample must see h = 0, not the next sample's rows)."""
  from openseq2seq_amd.optimizers.flat_params import FlatParams
  from openseq2seq_amd.parts.rnns import rnn_layers
  from openseq2seq_amd.parts.cnns.conv_blocks import Act, Tape
  B, T, In, H = 6, 40, 64, 128
  g = torch.Generator().manual_seed(5)
  x = torch.randn(B, T, In, generator=g).to(torch.bfloat16).to(cuda)
  dy = torch.randn(B, T, H, generator=g).to(torch.bfloat16).to(cuda)
  lens = torch.tensor([40, 17, 33, 1, 40, 8], dtype=torch.int32).to(cuda) if use_lens else None
  grads = []
  for min_rows in (1, 10 ** 9):
    monkeypatch.setattr(rnn_layers, "SHIFTED_WH_MIN_ROWS", min_rows)
    torch.manual_seed(0)
    store = FlatParams(cuda)
    layer = rnn_layers.RNNDirection(store, "rnn", cell, [In], H, reverse=reverse, forget_bias=1.0)
    store.finalize()
    tape = Tape()
    store.zero_grads()
    out = layer.forward([Act(x, None)], lens, tape)
    out.grad = dy.clone()
    tape.backward()
    torch.cuda.synchronize()
    grads.append(layer.wh.grad.float().cpu().clone())
  rel = float((grads[0] - grads[1]).norm() / grads[1].norm())
  assert float(grads[1].abs().max()) > 0 and rel <= 1e-3, rel

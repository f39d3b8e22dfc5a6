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
def test_rnn_direction(cuda, cell, reverse, use_lens):
  from openseq2seq_amd.optimizers.flat_params import FlatParams
  from openseq2seq_amd.parts.rnns.rnn_layers import RNNDirection
  from openseq2seq_amd.parts.cnns.conv_blocks import Act, Tape
  torch.manual_seed(0)
  B, T, In, H = 5, 23, 64, 96
  store = FlatParams(cuda)
  layer = RNNDirection(store, "rnn", cell, [In], H, reverse=reverse, forget_bias=1.0)
  store.finalize()
  g = torch.Generator().manual_seed(3)
  for p in store.params:            # non-trivial biases
    if p.kind == "vector":
      p.master.copy_((torch.randn(p.shape, generator=g) * 0.1).to(cuda))
  x = (torch.randn(B, T, In, generator=g)).to(torch.bfloat16)
  lens = torch.tensor([23, 7, 15, 1, 20], dtype=torch.int32) if use_lens else None
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


def test_birnn_stack_runs_ds2_shape(cuda):
  """DS2-like stack (2 layers, bidirectional GRU): shapes + finite gradients."""
  from openseq2seq_amd.optimizers.flat_params import FlatParams
  from openseq2seq_amd.parts.rnns.rnn_layers import BiRNNStack
  from openseq2seq_amd.parts.cnns.conv_blocks import Act, Tape
  torch.manual_seed(1)
  store = FlatParams(cuda)
  stack = BiRNNStack(store, "rnn", "gru_cudnn", 128, 160, 2, bidirectional=True)
  store.finalize()
  x = Act(torch.randn(4, 50, 128).to(torch.bfloat16).to(cuda), None)
  tape = Tape()
  outs = stack.forward(x, None, tape)
  assert len(outs) == 2 and tuple(outs[0].data.shape) == (4, 50, 160)
  for o in outs:
    o.grad = torch.randn(4, 50, 160).to(torch.bfloat16).to(cuda)
  store.zero_grads()
  tape.backward()
  torch.cuda.synchronize()
  assert torch.isfinite(store.grads).all() and float(store.grads.abs().sum()) > 0
  assert torch.isfinite(x.grad.float()).all()

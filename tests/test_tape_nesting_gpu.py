"""Tape.backward re-entrancy with real layers: the backward pass of a second network started from INSIDE a closure of
the first one — between two of its layers, while the BatchNorm-backward partials of the fused data-gradient
epilogue are in flight in the zeroed scratch arena — leaves both networks' gradients exactly what two separate
passes give (bit for bit in deterministic mode)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _net(cuda, seed, channels):
  from openseq2seq_amd.optimizers.flat_params import FlatParams
  from openseq2seq_amd.parts.cnns.conv_blocks import ConvBN
  torch.manual_seed(seed)
  store = FlatParams(cuda)
  layers = []
  cin = 64
  for i, c in enumerate(channels):
    n = "net%d/conv%d" % (seed, i)
    layers.append(ConvBN(store, n, n + "/bn", cin, c, 11))
    cin = c
  store.finalize()
  return store, layers


def _forward(layers, x, lens, tape, hook=None):
  from openseq2seq_amd.parts.cnns.conv_blocks import conv_bn_actv
  for i, L in enumerate(layers):
    if hook is not None and i == len(layers) - 1:
      tape.record(hook)              # runs after the last layer's closure, before the ones below it
    x = conv_bn_actv(L, x, lens, "relu", True, tape, keep_prob=1.0, seed=i, mask_output=True)
  return x


def test_nested_backward_passes_do_not_disturb_each_other(cuda):
  from openseq2seq_amd import capi
  from openseq2seq_amd.parts.cnns.conv_blocks import Act, Tape
  capi.set_deterministic(True)
  try:
    g = torch.Generator().manual_seed(0)
    B, T = 4, 300
    lens = torch.tensor([300, 211, 150, 40], dtype=torch.int32).to(cuda)
    xa = torch.randn(B, T, 64, generator=g).to(torch.bfloat16).to(cuda)
    xb = torch.randn(B, T, 64, generator=g).to(torch.bfloat16).to(cuda)
    sa, la = _net(cuda, 1, (384, 384, 512))
    sb, lb = _net(cuda, 2, (512, 384))
    dya = torch.randn(B, T, 512, generator=g).to(torch.bfloat16).to(cuda)
    dyb = torch.randn(B, T, 384, generator=g).to(torch.bfloat16).to(cuda)

    def run_b():
      sb.zero_grads()
      tape = Tape()
      out = _forward(lb, Act(xb, lens, requires_grad=False), lens, tape)
      out.grad = dyb.clone()
      tape.backward()

    # ---- separately -------------------------------------------------------------------------------------------------
    for _ in range(2):               # the second round runs on arena slices (the first teaches the arena its demand)
      sa.zero_grads()
      tape = Tape()
      out = _forward(la, Act(xa, lens, requires_grad=False), lens, tape)
      out.grad = dya.clone()
      tape.backward()
      run_b()
    torch.cuda.synchronize()
    ga, gb = sa.grads.clone(), sb.grads.clone()
    assert float(ga.abs().sum()) > 0 and float(gb.abs().sum()) > 0
    # ---- nested: network B's whole step runs inside network A's backward pass ---------------------------------------
    for _ in range(2):
      sa.zero_grads()
      tape = Tape()
      out = _forward(la, Act(xa, lens, requires_grad=False), lens, tape, hook=run_b)
      out.grad = dya.clone()
      tape.backward()
    torch.cuda.synchronize()
    assert torch.equal(sa.grads, ga)
    assert torch.equal(sb.grads, gb)
    assert len(capi._zero_arenas) >= 2
  finally:
    capi.set_deterministic(False)

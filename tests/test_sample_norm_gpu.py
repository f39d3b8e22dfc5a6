"""csrc/sample_norm.hip (tf.contrib.layers.layer_norm / instance_norm of conv_ln_actv / conv_in_actv,
open_seq2seq/parts/cnns/conv_blocks.py:234-309) against oracle/cnn.py, and TDNNEncoder with
normalization = 'layer_norm' / 'instance_norm' against the composed oracle layers. Tolerances: the kernels read
bf16 and write bf16 — outputs within bf16 rounding of the fp32 oracle on the same bf16 inputs (2e-3 rel-L2),
gradients 5e-3 (fp32 sums in another order), encoder-level kernel gradients 1e-2."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bf(x):
  return x.to(torch.bfloat16).float()


def _rel(a, b):
  return float((a - b).norm() / (b.norm() + 1e-20))


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("B,T,C", [(3, 210, 256), (2, 5, 64), (4, 97, 1024), (1, 840, 640), (5, 8, 6)])
def test_sample_norm_kernels(cuda, mode, B, T, C):
  from openseq2seq_amd import capi
  from oracle import cnn
  g = torch.Generator().manual_seed(B * 1000 + T + C + mode)
  x = _bf(torch.randn(B, T, C, generator=g) * 2.0 + torch.randn(1, 1, C, generator=g))     # per-channel offsets
  gamma = torch.rand(C, generator=g) + 0.5
  beta = torch.randn(C, generator=g) * 0.3
  dz = _bf(torch.randn(B, T, C, generator=g))
  eps = 1e-6 if mode == 0 else 1e-12
  z, mean, rstd = capi.sample_norm_fwd(x.to(torch.bfloat16).to(cuda), gamma.to(cuda), beta.to(cuda), mode, eps)
  dgamma = torch.full((C,), 0.5, device=cuda)          # accumulated into: start from a non-zero value
  dbeta = torch.full((C,), -0.25, device=cuda)
  dx = capi.sample_norm_bwd(dz.to(torch.bfloat16).to(cuda), x.to(torch.bfloat16).to(cuda), gamma.to(cuda), mean, rstd,
                            mode, dgamma, dbeta)
  torch.cuda.synchronize()
  xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
  fn = cnn.instance_norm_tf if mode == 0 else cnn.layer_norm_tf
  ref = fn(xr, gr, br, eps)
  (ref * dz).sum().backward()
  assert _rel(z.float().cpu(), ref.detach()) <= 2e-3
  assert _rel(dx.float().cpu(), xr.grad) <= 5e-3
  assert _rel(dgamma.cpu() - 0.5, gr.grad) <= 5e-3
  assert _rel(dbeta.cpu() + 0.25, br.grad) <= 5e-3
  # the saved statistics are the oracle's
  if mode == 0:
    torch.testing.assert_close(mean.cpu(), x.mean(dim=1), rtol=1e-4, atol=1e-4)
  else:
    torch.testing.assert_close(mean.cpu(), x.mean(dim=(1, 2))[:, None].expand(B, C), rtol=1e-4, atol=1e-4)
  # deterministic: a second call gives the same bits
  z2, _, _ = capi.sample_norm_fwd(x.to(torch.bfloat16).to(cuda), gamma.to(cuda), beta.to(cuda), mode, eps)
  assert torch.equal(z, z2)


@pytest.mark.parametrize("norm", ["layer_norm", "instance_norm"])
def test_tdnn_with_per_sample_normalization(cuda, norm):
  from openseq2seq_amd.optimizers.flat_params import FlatParams
  from openseq2seq_amd.encoders.tdnn_encoder import TDNNEncoder
  from openseq2seq_amd.parts.cnns.conv_blocks import ConvSampleNorm, Tape
  from oracle import cnn
  torch.manual_seed(0)
  layers = [
      {"type": "conv1d", "repeat": 1, "kernel_size": [11], "stride": [2], "num_channels": 128, "padding": "SAME",
       "dilation": [1]},
      {"type": "conv1d", "repeat": 2, "kernel_size": [7], "stride": [1], "num_channels": 192, "padding": "SAME",
       "dilation": [1]},
  ]
  store = FlatParams(cuda)
  enc = TDNNEncoder({"convnet_layers": layers, "dropout_keep_prob": 1.0, "activation_fn": "relu",
                     "normalization": norm, "use_conv_mask": True, "dtype": "mixed"}, None, mode="train").build(store, 64)
  store.finalize()
  assert all(isinstance(L["main"], ConvSampleNorm) for L in enc._layers)
  scope = {"layer_norm": "LayerNorm", "instance_norm": "InstanceNorm"}[norm]
  names = [p.name for p in store.params]
  # the variable names tf.contrib gives them: uniquified scopes directly under the encoder's scope
  assert names == ["ForwardPass/w2l_encoder/conv11/kernel",
                   "ForwardPass/w2l_encoder/%s/gamma" % scope, "ForwardPass/w2l_encoder/%s/beta" % scope,
                   "ForwardPass/w2l_encoder/conv21/kernel",
                   "ForwardPass/w2l_encoder/%s_1/gamma" % scope, "ForwardPass/w2l_encoder/%s_1/beta" % scope,
                   "ForwardPass/w2l_encoder/conv22/kernel",
                   "ForwardPass/w2l_encoder/%s_2/gamma" % scope, "ForwardPass/w2l_encoder/%s_2/beta" % scope], names
  g = torch.Generator().manual_seed(2)
  # non-trivial gamma / beta
  with torch.no_grad():
    for p in store.params:
      if p.name.endswith("/gamma"):
        p.master.copy_((torch.rand(p.master.shape, generator=g) + 0.5).to(cuda))
      if p.name.endswith("/beta"):
        p.master.copy_((torch.randn(p.master.shape, generator=g) * 0.2).to(cuda))
  store.refresh_compute_copies()
  B, T = 3, 210
  lens0 = torch.tensor([210, 133, 64], dtype=torch.int32)
  x0 = torch.randn(B, T, 64, generator=g).to(torch.bfloat16)
  store.zero_grads()
  tape = Tape()
  e = enc.encode({"source_tensors": [x0.to(cuda), lens0.to(cuda)], "tape": tape, "seed": 3})
  out = e["outputs_act"]
  dy = torch.randn(out.data.shape, generator=g).to(torch.bfloat16)
  out.grad = dy.to(cuda)
  tape.backward()
  torch.cuda.synchronize()
  # ---- restatement: masked input -> conv -> norm over the PADDED tensor -> relu -> mask ----------------------------
  P = {p.name: p for p in store.params}
  leaf = {}

  def var(name, conv=False):
    m = P[name].master.float().cpu()
    if conv:
      m = m.to(torch.bfloat16).float().permute(0, 2, 1).contiguous()       # device [K, Cout, Cin] -> TF [K, Cin, Cout]
    leaf[name] = m.requires_grad_(True)
    return leaf[name]

  fn = cnn.instance_norm_tf if norm == "instance_norm" else cnn.layer_norm_tf
  eps = 1e-6 if norm == "instance_norm" else 1e-12
  x = x0.float()
  lens = lens0.clone()
  specs = [("conv11", 2, ""), ("conv21", 1, "_1"), ("conv22", 1, "_2")]
  for i, (cname, stride, suffix) in enumerate(specs):
    x = x * cnn.seq_mask(lens, x.shape[1])
    y = cnn.conv1d_tf(x, var("ForwardPass/w2l_encoder/%s/kernel" % cname, True), stride, 1, "SAME")
    y = y + (_bf(y) - y).detach()                                          # the conv output is stored in bf16
    if stride > 1:
      lens = (lens + stride - 1) // stride
    z = fn(y, var("ForwardPass/w2l_encoder/%s%s/gamma" % (scope, suffix)),
           var("ForwardPass/w2l_encoder/%s%s/beta" % (scope, suffix)), eps)
    z = z + (_bf(z) - z).detach()                                          # ... and so is the normalised tensor
    a = torch.relu(z)
    if i < len(specs) - 1:
      a = a * cnn.seq_mask(lens, a.shape[1])
    x = a + (_bf(a) - a).detach()
  assert _rel(out.data.float().cpu(), x.detach()) <= 3e-3
  (x * dy.float()).sum().backward()
  for name, ref in leaf.items():
    got = P[name].grad.float().cpu()
    if name.endswith("/kernel"):
      got = got.permute(0, 2, 1)
    assert _rel(got, ref.grad) <= 1e-2, (name, _rel(got, ref.grad))

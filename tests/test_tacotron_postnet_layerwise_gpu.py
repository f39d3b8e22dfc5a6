"""The Tacotron2 post-net at the configuration's widths, LAYER BY LAYER and teacher-forced (VERDICT round 4,
weak #2: the config-width end-to-end test bounds the post-net output at 6e-2 — five conv + BatchNorm(train) +
tanh layers amplify bf16 rounding — and a wrong BatchNorm momentum or a dropped tanh in one layer could hide in
that). example_configs/text2speech/tacotron_gst.py:105-131: conv1d 80 -> 512 -> 512 -> 512 -> 512 -> 80, kernel 5,
SAME, tanh on the first four and none on the last, postnet_bn_momentum 0.1, postnet_bn_epsilon 1e-5
(decoders/tacotron2_decoder.py:520-552 of the reference: conv_bn_actv per layer, then dropout).

Each layer is run alone through parts/cnns/conv_blocks.conv_bn_actv on the tensor the DEVICE's previous layer
produced and compared with oracle/tdnn.py:tdnn_layer (fp32 math on the same bf16 weights, the device's bf16
storage points emulated) on that same tensor — nothing is inherited from earlier layers. Bounds (relative L2):
output 2e-3, d(input) / d(kernel) 6e-3, d(gamma) / d(beta) 1e-2 — the Jasper layer-wise bounds; the moving
statistics after the step against the oracle's update rule with momentum 0.1 (rtol 2e-3): a momentum of 0.9, or
TensorFlow's rule read the other way round, is off by a factor of 9 there."""
import pytest
import torch

pytestmark = pytest.mark.gpu

POSTNET = [(80, 512, "tanh"), (512, 512, "tanh"), (512, 512, "tanh"), (512, 512, "tanh"), (512, 80, None)]


def _rel(a, b):
  return float((a - b).norm() / (b.norm() + 1e-20))


def test_postnet_every_layer_teacher_forced(cuda):
  from openseq2seq_amd.optimizers.flat_params import FlatParams
  from openseq2seq_amd.parts.cnns.conv_blocks import Act, ConvBN, Tape, conv_bn_actv, xavier_normal_conv
  from oracle import cnn, tdnn
  torch.manual_seed(0)
  mom, eps = 0.1, 1e-5
  store = FlatParams(cuda)
  layers = []
  for i, (cin, cout, _) in enumerate(POSTNET):
    n = "ForwardPass/tacotron_2_decoder/decoder/conv%d" % (i + 1)
    layers.append(ConvBN(store, n, n + "/bn", cin, cout, 5, stride=1, padding="SAME", bn_momentum=mom,
                         bn_epsilon=eps, l2=0.0, initializer=xavier_normal_conv))
  store.finalize()
  g = torch.Generator().manual_seed(3)
  with torch.no_grad():      # gamma / beta / moving statistics away from their initial 1 / 0 / 0 / 1
    for L in layers:
      L.gamma.master.copy_((torch.rand(L.cout, generator=g) + 0.5).to(cuda))
      L.beta.master.copy_((torch.randn(L.cout, generator=g) * 0.3).to(cuda))
      L.moving_mean.copy_((torch.randn(L.cout, generator=g) * 0.1).to(cuda))
      L.moving_var.copy_((torch.rand(L.cout, generator=g) + 0.5).to(cuda))
  store.refresh_compute_copies()
  B, T = 8, 203          # mel frames of a batch; no sequence mask in the post-net
  x = Act((torch.randn(B, T, 80, generator=g) * 1.5).to(torch.bfloat16).to(cuda), None, requires_grad=True)
  worst = {}
  for i, (L, (cin, cout, act)) in enumerate(zip(layers, POSTNET)):
    name = "conv%d" % (i + 1)
    mm0, mv0 = L.moving_mean.cpu().clone(), L.moving_var.cpu().clone()
    store.zero_grads()
    xin = Act(x.data, None, requires_grad=True)
    tape = Tape()
    out = conv_bn_actv(L, xin, None, act, True, tape, keep_prob=1.0, seed=i, mask_output=False)
    dy = torch.randn(out.data.shape, generator=g).to(torch.bfloat16)
    out.grad = dy.to(cuda)
    tape.backward()
    torch.cuda.synchronize()
    # ---- oracle layer on the same input ------------------------------------------------------------------------
    w = {name + "/kernel": L.kernel.w16.float().cpu().permute(0, 2, 1).contiguous().requires_grad_(True),
         name + "/bn/gamma": L.gamma.master.cpu().clone().requires_grad_(True),
         name + "/bn/beta": L.beta.master.cpu().clone().requires_grad_(True)}
    xo = x.data.float().cpu().requires_grad_(True)
    blk = {"kernel_size": [5], "stride": [1], "padding": "SAME", "type": "conv1d"}
    yo = tdnn.tdnn_layer(xo, [], blk, name, w, None, act if act else "none", eps, None, 1.0, True)
    (yo * dy.float()).sum().backward()
    r = dict(out=_rel(out.data.float().cpu(), yo.detach()), dx=_rel(xin.grad.float().cpu(), xo.grad),
             dw=_rel(L.kernel.grad.float().cpu().permute(0, 2, 1), w[name + "/kernel"].grad),
             dgamma=_rel(L.gamma.grad.cpu(), w[name + "/bn/gamma"].grad),
             dbeta=_rel(L.beta.grad.cpu(), w[name + "/bn/beta"].grad))
    for k, v in r.items():
      worst[k] = max(worst.get(k, 0.0), v)
    assert r["out"] <= 2e-3, (name, r)
    assert r["dx"] <= 6e-3 and r["dw"] <= 6e-3, (name, r)
    assert r["dgamma"] <= 1e-2 and r["dbeta"] <= 1e-2, (name, r)
    # the activation is the configured one: tanh output is bounded, the linear last layer's is not
    if act == "tanh":
      assert float(out.data.float().abs().max()) <= 1.0
    else:
      assert float(out.data.float().abs().max()) > 1.5
    # moving statistics: moving * momentum + batch * (1 - momentum), Bessel-corrected batch variance
    with torch.no_grad():
      yc = cnn.conv1d_tf(x.data.float().cpu(), w[name + "/kernel"].detach(), 1, 1, "SAME")
      _, _, _, mm, mv = cnn.batch_norm_train(yc, w[name + "/bn/gamma"].detach(), w[name + "/bn/beta"].detach(), eps,
                                             mom, mm0, mv0)
    torch.testing.assert_close(L.moving_mean.cpu(), mm, rtol=2e-3, atol=2e-4)
    torch.testing.assert_close(L.moving_var.cpu(), mv, rtol=2e-3, atol=2e-4)
    x = Act(out.data, None, requires_grad=True)      # the DEVICE's output feeds the next layer on both sides
  print("post-net layer-wise worst rel-L2:", {k: "%.2e" % v for k, v in worst.items()})

"""Single conv_bn_actv layers, teacher-forced (the device layer and oracle/tdnn.py:tdnn_layer see the same bf16 input,
a ragged batch, and the same upstream gradient; the oracle emulates the device's bf16 storage points), for the layer
kinds the layer-wise Jasper test does not reach:

  * a STRIDED convolution past the first layer (the reference allows `stride` in any convnet_layers entry,
    encoders/tdnn_encoder.py:170-262; its example configs stride only in layer 1): the data gradient is the stride-1
    data gradient of the zero-upsampled output gradient (os2s_upsample_rows_bf16), conv1d and sep_conv1d;
  * QuartzNet 15x5's time-channel separable layers at their real widths and kernel sizes (K 33 ... 87, dilation 2,
    256 ... 1024 channels): every shape class of the register-window depthwise kernels + the pointwise GEMM + the
    fused BatchNorm statistics.

Bounds (relative L2, as tests/test_jasper_layerwise_gpu.py): output 2e-3, d(input) / d(kernel) 6e-3, BatchNorm
parameters 1e-2; measured d(input) 1.6e-3 ... 2.0e-3 on every case. (A composition of such layers adds 1 - 2e-2 of
ReLU-mask-flip noise to every gradient whatever the kernels do, which is why the end-to-end QuartzNet test in
test_sepconv_gpu.py has the loose bounds it has.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
  return float((a - b).norm() / (b.norm() + 1e-20))


def test_upsample_rows(cuda):
  from openseq2seq_amd import capi
  g = torch.Generator().manual_seed(0)
  x = torch.randn(3, 37, 72, generator=g).to(torch.bfloat16).to(cuda)
  for stride, tup in ((1, 37), (2, 73), (3, 109), (2, 80)):
    y = capi.upsample_rows(x, stride, tup)
    ref = torch.zeros(3, tup, 72, dtype=torch.bfloat16, device=cuda)
    ref[:, :(37 - 1) * stride + 1:stride] = x
    assert torch.equal(y, ref)


@pytest.mark.parametrize("ltype", ["conv1d", "sep_conv1d"])
@pytest.mark.parametrize("stride,K,dil", [(1, 11, 1), (2, 11, 1), (3, 7, 1), (2, 5, 2)])
def test_data_gradient_of_a_strided_layer(cuda, ltype, stride, K, dil):
  _one_layer(cuda, ltype, 128, 192, K, stride, dil, B=3, T=203, lens=[203, 131, 64])


# QuartzNet 15x5's time-channel separable layers at their real widths and kernel sizes
# (example_configs/speech2text/quartznet15x5_LibriSpeech.py: K 33 ... 75 at 256 / 512 channels, the K 87 dilation-2
# layer, the 512 -> 1024 K 1 layer): every shape class of the register-window depthwise kernels, one layer at a time
@pytest.mark.parametrize("cin,cout,K,dil", [(256, 256, 33, 1), (256, 256, 39, 1), (256, 512, 51, 1),
                                             (512, 512, 63, 1), (512, 512, 75, 1), (512, 512, 87, 2),
                                             (512, 1024, 1, 1)])
def test_quartznet_layer_shapes_teacher_forced(cuda, cin, cout, K, dil):
  _one_layer(cuda, "sep_conv1d", cin, cout, K, 1, dil, B=6, T=420, lens=[420, 400, 311, 203, 131, 64])


def _one_layer(cuda, ltype, cin, cout, K, stride, dil, B, T, lens):
  from openseq2seq_amd.optimizers.flat_params import FlatParams
  from openseq2seq_amd.parts.cnns.conv_blocks import Act, ConvBN, SepConvBN, Tape, conv_bn_actv
  from oracle import cnn, tdnn
  torch.manual_seed(0)
  store = FlatParams(cuda)
  Layer = ConvBN if ltype == "conv1d" else SepConvBN
  L = Layer(store, "ForwardPass/w2l_encoder/conv21", "ForwardPass/w2l_encoder/conv21/bn", cin, cout, K, stride, dil, "SAME")
  store.finalize()
  g = torch.Generator().manual_seed(stride * 10 + K)
  lens0 = torch.tensor(lens, dtype=torch.int32)
  in_mask = cnn.seq_mask(lens0, T)
  x0 = (torch.relu(torch.randn(B, T, cin, generator=g)) * in_mask).to(torch.bfloat16)      # a masked post-ReLU input
  out_len = (lens0 + stride - 1) // stride
  store.zero_grads()
  tape = Tape()
  xin = Act(x0.to(cuda), lens0.to(cuda), requires_grad=True)
  out = conv_bn_actv(L, xin, out_len.to(cuda), "relu", True, tape, keep_prob=1.0, seed=1, mask_output=True)
  dy = torch.randn(out.data.shape, generator=g).to(torch.bfloat16)
  out.grad = dy.to(cuda)
  tape.backward()
  torch.cuda.synchronize()
  name = "conv21"
  w = {}
  folded = ltype != "conv1d" and L.folded()
  if ltype == "conv1d":
    w[name + "/kernel"] = L.kernel.w16.float().cpu().permute(0, 2, 1).contiguous().requires_grad_(True)
  elif folded:
    # a one-tap separable layer runs as ONE 1x1 convolution with W diag(d) rounded to bf16 once (SepConvBN.folded,
    # os2s_pointwise_fold): the oracle layer gets that kernel and a unit scale (x * 1 is exact, so its storage point
    # between the halves is the identity); the two variables' gradients follow from its kernel gradient G by the
    # chain rule, dW = G diag(d), dd = sum_co W * G — in fp32 here, by os2s_pointwise_fold_bwd on the device
    dvec = L.depthwise.master.float().cpu()[0]
    Wm = L.kernel.master.float().cpu()[0]
    w[name + "/depthwise_kernel"] = torch.ones(1, cin)
    w[name + "/pointwise_kernel"] = (Wm * dvec[None, :]).to(torch.bfloat16).float().t()[None].contiguous().requires_grad_(True)
  else:
    w[name + "/depthwise_kernel"] = L.depthwise.master.float().cpu().clone().requires_grad_(True)
    w[name + "/pointwise_kernel"] = L.kernel.w16.float().cpu().permute(0, 2, 1).contiguous().requires_grad_(True)
  w[name + "/bn/gamma"] = L.gamma.master.cpu().clone().requires_grad_(True)
  w[name + "/bn/beta"] = L.beta.master.cpu().clone().requires_grad_(True)
  xo = x0.float().requires_grad_(True)
  blk = {"type": ltype, "kernel_size": [K], "stride": [stride], "dilation": [dil], "padding": "SAME"}
  Tout = out.data.shape[1]
  yo = tdnn.tdnn_layer(xo * in_mask, [], blk, name, w, cnn.seq_mask(out_len, Tout), "relu", 1e-3, None, 1.0, True)
  assert tuple(yo.shape) == tuple(out.data.shape)
  (yo * dy.float()).sum().backward()
  # (the separable layer stores the depthwise output — and its gradient — in bf16 between its two halves; the oracle
  # layer emulates that storage point too, oracle/cnn.py:sep_conv1d_tf `between`: without it d(input) is off by
  # 3.5 - 4 %, ReLU-mask flips from the 3e-3 the outputs then differ by; with it 1.7e-3, as for conv1d)
  ob, gb = 2e-3, 6e-3
  if ltype != "conv1d" and stride == 1 and dil == 1:
    # round 6: these layers' depthwise halves run on the matrix cores with the taps as a bf16 hi + lo pair — 16
    # mantissa bits instead of the oracle's 24: 0.2 % of the depthwise outputs round to the neighbouring bf16 value
    # (tools/check_dw_precision.py: 99.77 % of the elements equal bf16(fp32-tap convolution), the register-window
    # kernel 99.99 %), and the ReLU masks behind the BatchNorm flip with them: d(input) 7.6e-3 where the emulated
    # storage points otherwise give 1.7e-3 (without any emulation: 3.5 - 4 %)
    gb = 1e-2
  assert _rel(out.data.float().cpu(), yo.detach()) <= ob
  # rows past the input length are never read downstream and the kernels do not write them
  live = in_mask.bool().expand(B, T, cin)
  dx = torch.where(live, xin.grad.float().cpu(), torch.zeros(()))
  dxo = torch.where(live, xo.grad, torch.zeros(()))
  print(ltype, stride, K, dil, "dx rel", _rel(dx, dxo))
  assert _rel(dx, dxo) <= gb
  if ltype == "conv1d":
    assert _rel(L.kernel.grad.float().cpu().permute(0, 2, 1), w[name + "/kernel"].grad) <= gb
  elif folded:
    G = w[name + "/pointwise_kernel"].grad[0].t()                      # [Cout, Cin]
    assert _rel(L.kernel.grad.float().cpu()[0], G * dvec[None, :]) <= gb
    assert _rel(L.depthwise.grad.float().cpu()[0], (Wm * G).sum(0)) <= gb
  else:
    assert _rel(L.kernel.grad.float().cpu().permute(0, 2, 1), w[name + "/pointwise_kernel"].grad) <= gb
    assert _rel(L.depthwise.grad.float().cpu(), w[name + "/depthwise_kernel"].grad) <= gb
  assert _rel(L.gamma.grad.cpu(), w[name + "/bn/gamma"].grad) <= 1e-2
  assert _rel(L.beta.grad.cpu(), w[name + "/bn/beta"].grad) <= 1e-2

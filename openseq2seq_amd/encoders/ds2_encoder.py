"""DeepSpeech2Encoder — open_seq2seq/encoders/ds2_encoder.py:85-401 on the HIP kernels:
N x [conv2d + BatchNorm + activation] over (time, frequency) -> (bi)directional cuDNN-form
GRU/LSTM stack -> dense + activation + dropout.

conv2d runs on the 1-D implicit-GEMM MFMA kernel: activations stay [B, T, F*C]; the
frequency convolution is a banded channel mixing built from the [KT,KF,Cin,Cout] master
kernel (csrc/conv2d_toeplitz.hip), the time convolution is the tap loop. BatchNorm is per
output channel over (B, T, F) — the reference's fused BN on NHWC, conv_blocks.py:216-224 —
i.e. our BN kernels on the [B*T*F, C] view. The RNN ignores sequence lengths exactly like
the reference's cuDNN path (ds2_encoder.py:294-328 passes no lengths)."""
from __future__ import absolute_import, division, print_function

import math
import struct

import torch

from .encoder import Encoder
from .. import capi
from ..parts.cnns.conv_blocks import Act, act_id
from ..parts.rnns.rnn_layers import BiRNNStack
from ..parts.transformer.layers import Dense, SeedSeq


class Conv2dBN(object):
  def __init__(self, store, name, f_in, c_in, c_out, kernel_size, stride, padding, momentum,
               eps, l2):
    self.KT, self.KF = kernel_size
    self.sT, self.sF = stride
    self.Fi, self.Cin, self.Cout = f_in, c_in, c_out
    if padding == "SAME":
      self.Fo, self.padF = capi.same_padding(f_in, self.KF, self.sF, 1)
    else:
      self.Fo, self.padF = (f_in - self.KF) // self.sF + 1, 0
    self.padding = padding
    self.momentum, self.eps = momentum, eps
    KT, KF = self.KT, self.KF

    def init(shape):   # xavier (uniform) over fan_in = KT*KF*Cin, fan_out = KT*KF*Cout
      lim = math.sqrt(6.0 / (KT * KF * (c_in + c_out)))
      return (torch.rand(shape) * 2 - 1) * lim

    self.kernel = store.add(name + "/kernel", (KT, KF, c_in, c_out), init, kind="dense", l2=l2)
    self.gamma = store.add(name + "/bn/gamma", (c_out,), torch.ones(c_out), kind="vector", l2=l2)
    self.beta = store.add(name + "/bn/beta", (c_out,), torch.zeros(c_out), kind="vector")
    dev = store.device
    self.moving_mean = torch.zeros(c_out, device=dev)
    self.moving_var = torch.ones(c_out, device=dev)
    if hasattr(store, "add_state"):
      store.add_state(name + "/bn/moving_mean", self.moving_mean)
      store.add_state(name + "/bn/moving_variance", self.moving_var)
    self.cin_f, self.cout_f = f_in * c_in, self.Fo * c_out
    if self.cin_f % 8 or self.cout_f % 8:
      raise NotImplementedError("F*C must be a multiple of 8")
    self._wexp = self._wexpT = None
    self._desc = torch.frombuffer(bytearray(struct.pack(
        "<qqiiii", 0, 0, KT, self.cout_f, self.cin_f, 0)), dtype=torch.uint8).to(dev).view(1, 32)
    self._tiles = KT * (-(-self.cout_f // 64)) * (-(-self.cin_f // 64))

  def _expand(self, need_dgrad):
    dev = self.kernel.master.device
    if self._wexp is None:
      self._wexp = torch.empty((self.KT, self.cout_f, self.cin_f), dtype=torch.bfloat16, device=dev)
    capi.conv2d_toeplitz_expand(self.kernel.master, self.Fi, self.Fo, self.sF, self.padF, self._wexp)
    if need_dgrad:
      if self._wexpT is None:
        self._wexpT = torch.empty((self.KT, self.cin_f, self.cout_f), dtype=torch.bfloat16,
                                  device=dev)
      capi.conv_weight_dgrad_copy(self._wexp.view(-1), self._wexpT.view(-1), self._desc,
                                  self._tiles)

  def forward(self, x, activation_fn, training, tape):
    """x: Act [B, T, Fi*Cin] -> Act [B, T', Fo*Cout]."""
    B, Tin, _ = x.data.shape
    self._expand(need_dgrad=training and x.requires_grad)
    if self.padding == "SAME":
      tout, pl = capi.same_padding(Tin, self.KT, self.sT, 1)
    else:
      tout, pl = capi.valid_padding(Tin, self.KT, self.sT, 1)
    y = capi.conv1d_fwd(x.data, self._wexp, stride=self.sT, pad_left=pl, tout=tout)
    C = self.Cout
    rows = B * tout * self.Fo
    dev = y.device
    sc, sh = torch.empty(C, device=dev), torch.empty(C, device=dev)
    mean = rstd = part = None
    if training:
      part = capi.bn_stats(y.view(rows, C))
      mean, rstd = torch.empty(C, device=dev), torch.empty(C, device=dev)
    capi.bn_finalize(part, rows, self.gamma.master, self.beta.master, self.eps, self.momentum,
                     training, self.moving_mean, self.moving_var, mean, rstd, sc, sh)
    out = torch.empty_like(y)
    act = act_id(activation_fn)
    capi.bn_act_fwd([y.view(1, rows, C)], [sc], [sh], out.view(1, rows, C), None, act, 1.0, 0)
    res = Act(out, None)
    if not (training and tape is not None):
      return res
    L = self

    def backward():
      dout = res.grad
      assert dout is not None
      dz = torch.empty_like(out)
      partial = torch.empty((capi.bn_act_bwd_num_parts(rows), 2, C), dtype=torch.float32, device=dev)
      capi.bn_act_bwd_reduce(dout.view(1, rows, C), out.view(1, rows, C), [y.view(1, rows, C)],
                             [mean], [rstd], dz.view(1, rows, C), partial, None, act, 1.0, 0)
      c1, c2 = torch.empty(C, device=dev), torch.empty(C, device=dev)
      capi.bn_bwd_finalize(partial, 1, rows, L.gamma.grad, L.beta.grad, True, c1, c2)
      dy = torch.empty_like(y)
      capi.bn_bwd_apply(dz.view(rows, C), y.view(rows, C), L.gamma.master, mean, rstd, c1, c2,
                        dy.view(rows, C))
      dwexp = capi.conv1d_wgrad(x.data, dy, L.KT, stride=L.sT, pad_left=pl)
      capi.conv2d_toeplitz_reduce(dwexp, L.Fi, L.Fo, L.sF, L.padF, L.kernel.grad)
      if x.requires_grad:
        src = dy
        if L.sT != 1:
          # transposed convolution: zero-stuff dy to the input rate, then the stride-1
          # data-gradient convolution  dx[t] = sum_k up[t + pl - k] w[k]
          To = dy.shape[1]
          src = torch.zeros((B, (To - 1) * L.sT + 1, dy.shape[2]), dtype=dy.dtype, device=dev)
          src[:, ::L.sT] = dy
        g = x.grad_buffer()
        capi.conv1d_fwd(src, L._wexpT, pad_left=(L.KT - 1) - pl, tout=Tin, out=g,
                        accumulate=x.grad_init)
        x.grad_init = True
      res.grad = None

    tape.record(backward, [L.kernel, L.gamma, L.beta])
    return res


class DeepSpeech2Encoder(Encoder):
  @staticmethod
  def get_required_params():
    return dict(Encoder.get_required_params(), **{
        'dropout_keep_prob': float, 'conv_layers': list, 'activation_fn': None,
        'num_rnn_layers': int, 'row_conv': bool, 'n_hidden': int, 'use_cudnn_rnn': bool,
        'rnn_cell_dim': int,
        'rnn_type': ['layernorm_lstm', 'lstm', 'gru', 'cudnn_gru', 'cudnn_lstm'],
        'rnn_unidirectional': bool,
    })

  @staticmethod
  def get_optional_params():
    return dict(Encoder.get_optional_params(), **{
        'row_conv_width': int,
        'data_format': ['channels_first', 'channels_last', 'BCTF', 'BTFC', 'BCFT', 'BFTC'],
        'bn_momentum': float, 'bn_epsilon': float,
    })

  def __init__(self, params, model, name="ds2_encoder", mode='train'):
    super(DeepSpeech2Encoder, self).__init__(params, model, name, mode)
    if self.params['rnn_type'] not in ('cudnn_gru', 'gru', 'cudnn_lstm', 'lstm'):
      raise NotImplementedError("rnn_type " + self.params['rnn_type'])

  def build(self, store, num_features):
    p = self.params
    l2 = 0.0
    if p.get('regularizer', None) is not None:
      l2 = float(p.get('regularizer_params', {}).get('scale', 0.0))
    mom, eps = p.get('bn_momentum', 0.99), p.get('bn_epsilon', 1e-3)
    scope = "ForwardPass/" + self._name
    self.convs = []
    f, c = num_features, 1
    for i, cl in enumerate(p['conv_layers']):
      layer = Conv2dBN(store, "%s/conv%d" % (scope, i + 1), f, c, cl['num_channels'],
                       cl['kernel_size'], cl['stride'], cl['padding'], mom, eps, l2)
      self.convs.append(layer)
      f, c = layer.Fo, cl['num_channels']
    self.rnn = None
    width = f * c
    if p['num_rnn_layers'] > 0:
      cell = "gru_cudnn" if "gru" in p['rnn_type'] else "lstm_cudnn"
      self.rnn = BiRNNStack(store, scope + "/" + ("cudnn_gru" if "gru" in cell else "cudnn_lstm"),
                            cell, width, p['rnn_cell_dim'], p['num_rnn_layers'],
                            bidirectional=not p['rnn_unidirectional'])
      width = self.rnn.output_dim
    self.row_conv = None
    if p['row_conv'] and p.get('row_conv_width', 8) >= 2:       # ds2_encoder.py:41-42, 361-375
      from ..parts.cnns.conv_blocks import DepthwiseBN
      self.row_conv = DepthwiseBN(store, scope + "/row_conv", width, p.get('row_conv_width', 8), mom, eps, l2)
    self.fc = Dense(store, scope + "/fully_connected", width, p['n_hidden'], True)
    self.fc.kernel.l2 = l2
    self.output_dim = p['n_hidden']
    return self

  def _encode(self, input_dict):
    source_sequence, src_length = input_dict['source_tensors']
    tape = input_dict.get('tape', None)
    training = (self._mode == "train")
    seeds = SeedSeq(input_dict.get('seed', 0))
    keep = self.params['dropout_keep_prob'] if training else 1.0
    x = Act(source_sequence, None, requires_grad=False)
    for cl, layer in zip(self.params['conv_layers'], self.convs):
      s = cl['stride'][0]
      if cl['padding'] == "VALID":
        src_length = torch.div(src_length - cl['kernel_size'][0] + s, s, rounding_mode='floor')
      else:
        src_length = torch.div(src_length + s - 1, s, rounding_mode='floor')
      x = layer.forward(x, self.params['activation_fn'], training, tape if training else None)
    if self.rnn is not None:
      x = self.rnn.forward(x, None, tape if training else None, keep_prob=keep, seeds=seeds)
    if self.row_conv is not None:
      from ..parts.cnns.conv_blocks import conv_bn_actv
      x = conv_bn_actv(self.row_conv, x, None, self.params['activation_fn'], training,
                       tape if training else None)
    B, T, W = x.data.shape
    x2 = Act(x.data.view(B * T, W), None)
    if training and tape is not None:
      src = x

      def backward(x2=x2, src=src):   # un-flatten the gradient
        if x2.grad is not None:
          src.grad, src.grad_init = x2.grad.view(B, T, W), True
        x2.grad = None

      tape.record(backward)
    # the hidden layer takes the encoder's activation (ds2_encoder.py:381-388: the clipped ReLU of the configs)
    fc_act = act_id(self.params['activation_fn'])
    if fc_act not in (0, 1, 3):
      raise NotImplementedError("activation of the DeepSpeech2 hidden layer")
    y = self.fc.forward(x2, tape if training else None, act=fc_act, keep=keep, seed=seeds.next())
    out = Act(y.data.view(B, T, -1), None)
    if training and tape is not None:
      def backward2(out=out, y=y):
        if out.grad is not None:
          y.grad = out.grad.reshape(B * T, -1)
        out.grad = None

      tape.record(backward2)
    return {'outputs': out.data, 'src_length': src_length, 'outputs_act': out}

"""Global style token encoder — Tacotron2Encoder._embed_style
(open_seq2seq/encoders/tacotron2_encoder.py:341-505): the reference encoder (6 x conv2d 3x3
stride 2 + BN + ReLU over the style mel-spectrogram, :370-397), a GRUCell summary of the
remaining time axis (:401-425), Dense(128, tanh) "reference_activation" (:448-456) and the
multi-head "bahdanau" attention of that vector over the tanh of num_tokens non-trainable
random style tokens (:475-505). conv2d runs on the 1-D implicit-GEMM kernel through the
block-Toeplitz expansion (encoders/ds2_encoder.py:Conv2dBN); the GRU and the token attention
are csrc/gst.hip."""
from __future__ import absolute_import, division, print_function

import math

import torch

from ... import capi
from ...encoders.ds2_encoder import Conv2dBN
from ..cnns.conv_blocks import Act, accumulate_grad
from ..transformer.layers import Dense, _colsum_into


class StyleEncoder(object):
  def __init__(self, store, scope, params, num_features, activation_fn, bn_momentum, bn_eps, l2):
    p = params
    rt = p['rnn_type'] if isinstance(p['rnn_type'], str) else getattr(p['rnn_type'], "__name__", "")
    if "GRU" not in rt or not p['rnn_unidirectional'] or p['num_rnn_layers'] != 1:
      raise NotImplementedError("style encoder RNN other than one unidirectional GRUCell layer")
    self.act = activation_fn
    self.conv_params = p.get('conv_layers', [])
    self.convs = []
    f, c = num_features, 1
    for i, cl in enumerate(self.conv_params):
      layer = Conv2dBN(store, "%s/conv%d" % (scope, i + 1), f, c, cl['num_channels'],
                       cl['kernel_size'], cl['stride'], cl['padding'], bn_momentum, bn_eps, l2)
      self.convs.append(layer)
      f, c = layer.Fo, cl['num_channels']
    self.in_dim = f * c
    H = p['rnn_cell_dim']
    self.H = H
    rnn = scope + "/rnn/multi_rnn_cell/cell_0/gru_cell"
    fan = self.in_dim + H

    def glorot(fo):
      def init(shape):
        lim = math.sqrt(6.0 / (fan + fo))
        return (torch.rand(shape) * 2 - 1) * lim
      return init

    # TF kernels [in + H, out] split into the input rows (GEMM, device layout [out, in]) and
    # the state rows (fp32 [H, out], used inside the recurrence)
    self.wg_x = store.add(rnn + "/gates/kernel_x", (1, 2 * H, self.in_dim), glorot(2 * H), kind="conv", l2=l2)
    self.wg_h = store.add(rnn + "/gates/kernel_h", (H, 2 * H), glorot(2 * H), kind="vector", l2=l2)
    self.bg = store.add(rnn + "/gates/bias", (2 * H,), torch.ones(2 * H), kind="vector")
    self.wc_x = store.add(rnn + "/candidate/kernel_x", (1, H, self.in_dim), glorot(H), kind="conv", l2=l2)
    self.wc_h = store.add(rnn + "/candidate/kernel_h", (H, H), glorot(H), kind="vector", l2=l2)
    self.bc = store.add(rnn + "/candidate/bias", (H,), torch.zeros(H), kind="vector")
    self.ref = Dense(store, scope + "/reference_activation", H, 128, True)
    self.ref.kernel.l2 = l2
    att = p['attention_layer_size']
    self.heads, self.N, E = p['num_heads'], p['num_tokens'], p['emb_size']
    if att % self.heads or att // self.heads != 64:
      raise NotImplementedError("token attention with head depth != 64")
    self.att = att
    a = scope + "/attention"
    self.q = Dense(store, a + "/q", 128, att, False)
    self.k = Dense(store, a + "/k", E, att, False)
    self.v = Dense(store, a + "/v", E, att, False)
    self.att_v = store.add(a + "/attention_v", (64,), lambda s: (torch.rand(s) * 2 - 1) * math.sqrt(3.0 / 64),
                           kind="vector")
    self.o = Dense(store, a + "/output_transform", att, att, False)
    # trainable=False random tokens (:475-486): not part of the parameter store
    g = torch.Generator().manual_seed(20190501)
    self.tokens = ((torch.rand(self.N, E, generator=g) * 2 - 1)).to(store.device)
    self.output_dim = att

  def forward(self, style_spec, style_len, training, tape):
    """style_spec fp32/bf16 [B, T, F]; style_len int32 [B]. Returns Act [B, att] (bf16)."""
    B, T, F = style_spec.shape
    dev = style_spec.device
    x = Act(style_spec.to(torch.bfloat16).contiguous(), None, requires_grad=False)
    lens = style_len
    for cl, layer in zip(self.conv_params, self.convs):
      s = cl['stride'][0]
      if cl['padding'] == "VALID":
        lens = torch.div(lens - cl['kernel_size'][0] + s, s, rounding_mode='floor')
      else:
        lens = torch.div(lens + s - 1, s, rounding_mode='floor')
      x = layer.forward(x, self.act, training, tape)
    lens = lens.to(torch.int32)
    Bx, Tr, W = x.data.shape
    H = self.H
    flat = x.data.reshape(Bx * Tr, W)
    gxg = capi.gemm(flat, self.wg_x.w16.view(2 * H, W), bias=self.bg.master).view(B, Tr, 2 * H)
    gxc = capi.gemm(flat, self.wc_x.w16.view(H, W), bias=self.bc.master).view(B, Tr, H)
    sv = capi.gru_tf_fwd(gxg, gxc, self.wg_h.master, self.wc_h.master, lens)
    hf = torch.empty((B, H), dtype=torch.bfloat16, device=dev)
    capi.cast_f32_to_bf16(sv["h_final"].view(-1), hf.view(-1))
    hfin = Act(hf)
    enc = self
    if tape is not None:
      def gru_bwd():
        dh = hfin.grad.float()
        dgxg, dgxc = capi.gru_tf_bwd(dh, enc.wg_h.master.t().contiguous(),
                                     enc.wc_h.master.t().contiguous(), lens, sv)
        dg2, dc2 = dgxg.view(B * Tr, 2 * H), dgxc.view(B * Tr, H)
        capi.gemm_wgrad(flat, dg2, enc.wg_x.grad.view(2 * H, W), accumulate=True)
        capi.gemm_wgrad(flat, dc2, enc.wc_x.grad.view(H, W), accumulate=True)
        _colsum_into(dg2, enc.bg)
        _colsum_into(dc2, enc.bc)
        # recurrent kernels: dWg_h [H,2H] = hprev^T dg, dWc_h [H,H] = (r*h)^T dc
        tg = torch.zeros((2 * H, H), dtype=torch.float32, device=dev)
        capi.gemm_wgrad(sv["hprev16"].view(B * Tr, H), dg2, tg, accumulate=False)
        enc.wg_h.grad.add_(tg.t())
        tc = torch.zeros((H, H), dtype=torch.float32, device=dev)
        capi.gemm_wgrad(sv["rh16"].view(B * Tr, H), dc2, tc, accumulate=False)
        enc.wc_h.grad.add_(tc.t())
        if x.requires_grad:
          g = x.grad_buffer()
          capi.gemm(dg2, enc.wg_x.wt16.view(W, 2 * H), out=g.view(B * Tr, W), accumulate=x.grad_init)
          capi.gemm(dc2, enc.wc_x.wt16.view(W, H), out=g.view(B * Tr, W), accumulate=True)
          x.grad_init = True
        hfin.grad = None
      tape.record(gru_bwd, [self.wg_x, self.wg_h, self.bg, self.wc_x, self.wc_h, self.bc])
    lin = self.ref.forward(hfin, tape)                                # Dense(128, tanh)
    ref = Act(capi.tanh_fwd(lin.data))
    if tape is not None:
      def tanh_bwd():
        accumulate_grad(lin, capi.tanh_bwd(ref.grad, ref.data))
        ref.grad = None
      tape.record(tanh_bwd)
    q = self.q.forward(ref, tape)                                     # [B, att]
    tok = Act(torch.tanh(self.tokens).to(torch.bfloat16), requires_grad=False)
    k = self.k.forward(tok, tape)                                     # [N, att]
    v = self.v.forward(tok, tape)
    out, w = capi.gst_attention_fwd(q.data, k.data, v.data, self.att_v.master, self.heads)
    ao = Act(out)
    if tape is not None:
      def att_bwd():
        dk = torch.zeros((enc.N, enc.att), dtype=torch.float32, device=dev)
        dv = torch.zeros((enc.N, enc.att), dtype=torch.float32, device=dev)
        dq = capi.gst_attention_bwd(ao.grad, q.data, k.data, v.data, enc.att_v.master, w, enc.heads,
                                    dk, dv, enc.att_v.grad)
        accumulate_grad(q, dq)
        for t, d in ((k, dk), (v, dv)):
          d16 = torch.empty(d.shape, dtype=torch.bfloat16, device=dev)
          capi.cast_f32_to_bf16(d.view(-1), d16.view(-1))
          accumulate_grad(t, d16)
        ao.grad = None
      tape.record(att_bwd, [self.att_v])
    return self.o.forward(ao, tape)

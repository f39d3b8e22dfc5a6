"""RNN decoders with attention — open_seq2seq/decoders/rnn_decoders.py:22-321 on the HIP
attention-decoder loop (csrc/attn_decoder.hip).

RNNDecoderWithAttention, attention_type "gnmt" / "gnmt_v2" (the en-de-nmt-small and GNMT
configs): the bottom LSTM layer is the attention cell (AttentionWrapper with
attention_layer_size=None, output_attention=False, normalised Bahdanau attention over the
encoder outputs); GNMTAttentionMultiCell (parts/rnns/gnmt.py:32-79) feeds the attention
vector of the current step (gnmt_v2) or of the previous step (gnmt) to every upper layer.
Only the attention cell is sequential in the training pass (TrainingHelper = teacher
forcing): its input projection, the memory keys, all upper layers, the output projection
and every weight gradient are whole-sequence GEMMs / RNN layers. "bahdanau" is the plain
AttentionWrapper over the full MultiRNNCell (<= 2 layers) whose output is the context.
Eval / infer (GreedyEmbeddingHelper, at most 2 x max source length steps, :303-306) drives
the same kernels one step at a time."""
from __future__ import absolute_import, division, print_function

import math

import torch

from .decoder import Decoder
from .. import capi
from ..encoders.rnn_encoders import Embedding, apply_scope_initializer, cell_spec, dropout_act, residual_add
from ..parts.cnns.conv_blocks import Act
from ..parts.rnns.rnn_layers import RNNDirection, rnn_directions_forward
from ..parts.transformer.layers import SeedSeq, _colsum_into


def _round8(n):
  return (n + 7) // 8 * 8


class AttentionCell(object):
  """Parameters + training/inference passes of one AttentionWrapper(LSTM stack) loop."""

  def __init__(self, store, scope, in_dim, H, M, U, L, mode, forget_bias, use_bias=False,
               loc_k=0, loc_f=0):
    self.in_dim, self.H, self.M, self.U, self.L, self.mode = in_dim, H, M, U, L, mode
    self.forget_bias, self.use_bias, self.loc_k, self.loc_f = forget_bias, use_bias, loc_k, loc_f
    GH = 4 * H
    fan0 = in_dim + M + H

    def glorot(fan_in, fan_out):
      def f(shape):
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        return (torch.rand(shape) * 2 - 1) * lim
      return f

    # layer 0 kernel rows split by source: inputs | attention + recurrent state
    self.w_in = store.add(scope + "/cell_0/kernel_inputs", (1, GH, in_dim), glorot(fan0, GH), kind="conv")
    self.wcat = [store.add(scope + "/cell_0/kernel_attention_state", (1, GH, M + H), glorot(fan0, GH), kind="conv")]
    self.bias = [store.add(scope + "/cell_0/bias", (GH,), torch.zeros(GH), kind="vector")]
    for l in range(1, L):
      self.wcat.append(store.add(scope + "/cell_%d/kernel" % l, (1, GH, 2 * H), glorot(2 * H, GH), kind="conv"))
      self.bias.append(store.add(scope + "/cell_%d/bias" % l, (GH,), torch.zeros(GH), kind="vector"))
    att = scope + "/attention"
    self.w_mem = store.add(att + "/memory_layer/kernel", (1, U, M), glorot(M, U), kind="conv")
    self.luong = mode == capi.SCORE_LUONG
    if self.luong:
      # LuongAttention has no query layer and no score vector: the query is the cell output
      if U != H:
        raise ValueError("luong attention: attention_layer_size must equal the cell size")
      self.w_q = self.v = None
      self._eye = torch.eye(U, dtype=torch.bfloat16, device=store.device)
      self._zero_v = torch.zeros(U, dtype=torch.float32, device=store.device)
    else:
      self.w_q = store.add(att + "/query_layer/kernel", (1, U, H), glorot(H, U), kind="conv")
      self.v = store.add(att + "/attention_v", (U,), lambda s: (torch.rand(s) * 2 - 1) * math.sqrt(3.0 / U),
                         kind="vector")
    self.g = self.b = self.conv_w = self.conv_b = self.dense_w = None
    if mode == capi.SCORE_BAHDANAU_NORM:
      self.g = store.add(att + "/attention_g", (1,), torch.full((1,), math.sqrt(1.0 / U)), kind="vector")
      self.b = store.add(att + "/attention_b", (U,), torch.zeros(U), kind="vector")
    if mode == capi.SCORE_LOCATION:
      if use_bias:
        self.b = store.add(att + "/attention_bias", (U,), lambda s: (torch.rand(s) * 2 - 1) * math.sqrt(3.0 / U),
                           kind="vector")
      self.conv_w = store.add(att + "/location_conv/kernel", (loc_k, loc_f), glorot(loc_k, loc_k * loc_f),
                              kind="vector")
      self.conv_b = store.add(att + "/location_conv/bias", (loc_f,), torch.zeros(loc_f), kind="vector")
      self.dense_w = store.add(att + "/location_dense/kernel", (loc_f, U), glorot(loc_f, U), kind="vector")

  def params(self):
    ps = [self.w_in, self.w_mem] + ([] if self.luong else [self.w_q, self.v]) + self.wcat + self.bias
    return ps + [p for p in (self.g, self.b, self.conv_w, self.conv_b, self.dense_w) if p is not None]

  def _new_loop(self, B, T, S, dev, training, attn_in_keep, out_keep, seeds, y_top=None, ctx=None):
    m = lambda p: None if p is None else p.master
    dec = capi.AttnDecoder(B, T, S, self.L, self.H, self.M, self.U, self.mode, dev,
                           use_bias=self.use_bias, loc_k=self.loc_k, loc_f=self.loc_f,
                           forget_bias=self.forget_bias, attn_in_keep=attn_in_keep,
                           attn_in_seed=seeds.next(), out_keep=out_keep,
                           out_seeds=(seeds.next(), seeds.next()), save=training, y_top=y_top, ctx=ctx)
    GH = 4 * self.H
    dec.set_params([w.w16.view(GH, -1) for w in self.wcat],
                   self._eye if self.luong else self.w_q.w16.view(self.U, self.H),
                   self._zero_v if self.luong else self.v.master,
                   bias=[None] + [b.master for b in self.bias[1:]], g=m(self.g),
                   b=m(self.b), conv_w=m(self.conv_w), conv_b=m(self.conv_b), dense_w=m(self.dense_w))
    if getattr(self, "fp8_weights", False):
      dec.set_fp8_weights(self.fp8_copies())
    return dec

  def fp8_copies(self):
    """e4m3 copies (+ per-row scales) of the recurrent weight matrices the time loop streams every
    step, re-quantised whenever the bf16 weights changed (optimizer step, checkpoint load): the
    forward cell kernels read these, the backward pass keeps the bf16 weights (gradients are taken
    straight through the quantisation)."""
    store = getattr(self.wcat[0], "store", None)
    ver = getattr(store, "version", None) if store is not None else None
    cache = getattr(self, "_fp8_cache", None)
    GH = 4 * self.H
    if cache is None:
      cache = self._fp8_cache = {"ver": object(), "w8": [capi.quantize_rows_e4m3(w.w16.view(GH, -1))
                                                        for w in self.wcat]}
      cache["ver"] = ver
    elif ver is None or cache["ver"] != ver:
      for (q, sc), w in zip(cache["w8"], self.wcat):
        capi.quantize_rows_e4m3(w.w16.view(GH, -1), q, sc)
      cache["ver"] = ver
    return cache["w8"]

  def memory(self, enc, tape):
    """values (already zero past the source lengths) -> keys = memory_layer(values)."""
    B, S, M = enc.data.shape
    keys = capi.gemm(enc.data.reshape(B * S, M), self.w_mem.w16.view(self.U, M)).view(B, S, self.U)
    return keys

  def forward_train(self, x, enc, src_len, tgt_len, tape, seeds, attn_in_keep=1.0, out_keep=1.0,
                    y_top=None, ctx=None):
    """x: Act [B,T,in_dim] (already input-dropped); enc: Act [B,S,M] encoder outputs.
    Returns (y Act [B,T,H], ctx Act [B,T,M], loop) — teacher-forced pass over all T steps."""
    B, T, _ = x.data.shape
    S = enc.data.shape[1]
    H, M, U, GH = self.H, self.M, self.U, 4 * self.H
    dev = x.data.device
    training = tape is not None
    gx0 = capi.gemm(x.data.reshape(B * T, -1), self.w_in.w16.view(GH, -1),
                    bias=self.bias[0].master).view(B, T, GH)
    keys = self.memory(enc, tape)
    dec = self._new_loop(B, T, S, dev, training, attn_in_keep, out_keep, seeds, y_top, ctx)
    dec.set_inputs(gx0, keys, enc.data, src_len, tgt_len)
    dec.forward()
    y, c = Act(dec.y_top, tgt_len), Act(dec.ctx, tgt_len)
    if not training:
      return y, c, dec
    cell = self

    def backward():
      out = dec.backward([w.wt16.view(-1, GH) for w in cell.wcat],
                         cell._eye if cell.luong else cell.w_q.wt16.view(H, U), dy_top=y.grad, dctx_ext=c.grad,
                         dv=torch.zeros_like(cell._zero_v) if cell.luong else cell.v.grad,
                         dg=cell.g.grad if cell.g is not None else None,
                         dconv_w=cell.conv_w.grad if cell.conv_w is not None else None,
                         dconv_b=cell.conv_b.grad if cell.conv_b is not None else None,
                         ddense_w=cell.dense_w.grad if cell.dense_w is not None else None)
      y.grad = c.grad = None
      # weight gradients: whole-sequence GEMMs over the saved step inputs
      for l in range(cell.L):
        dg2 = out["dg"][l].view(B * T, GH)
        _wgrad_rows(dec.cat[l], out["dg"][l], cell.wcat[l].grad)
        _colsum_into(dg2, cell.bias[l])
      dg0 = out["dg"][0].view(B * T, GH)
      capi.gemm_wgrad(x.data.reshape(B * T, -1), dg0, cell.w_in.grad.view(GH, -1), accumulate=True)
      if x.requires_grad:
        g = x.grad_buffer()
        capi.gemm(dg0, cell.w_in.wt16.view(-1, GH), out=g.view(B * T, -1), accumulate=x.grad_init)
        x.grad_init = True
      dq2 = out["dq_seq"].view(B * T, U)
      if not cell.luong:
        capi.gemm_wgrad(dec.y_top.reshape(B * T, H) if dec.y_top.is_contiguous() else
                        dec.y_top.contiguous().view(B * T, H), dq2, cell.w_q.grad.view(U, H), accumulate=True)
      if cell.b is not None:
        _colsum_into(dq2, cell.b)
      # memory layer + encoder outputs
      dk16 = torch.empty((B * S, U), dtype=torch.bfloat16, device=dev)
      capi.cast_f32_to_bf16(out["dkeys"].view(-1), dk16.view(-1))
      capi.gemm_wgrad(enc.data.reshape(B * S, M), dk16, cell.w_mem.grad.view(U, M), accumulate=True)
      if enc.requires_grad:
        dmem = out["dmem"].view(B * S, M)
        capi.gemm(dk16, cell.w_mem.wt16.view(M, U), out=dmem, accumulate=True)
        if enc.grad_init and enc.grad is not None:
          capi.add_bf16(enc.grad, dmem.view(B, S, M), out=enc.grad)
        else:
          enc.grad, enc.grad_init = dmem.view(B, S, M), True

    tape.record(backward, self.params())
    return y, c, dec


def _wgrad_rows(cat, dg, out_grad):
  """dW[4H, Kc] += sum_{b,t<T} dg[b,t,:]^T cat[b,t,:] (cat has T+1 rows per sample: the
  first T are the step inputs)."""
  B, T1, Kc = cat.shape
  T = T1 - 1
  GH = dg.shape[2]
  capi.gemm_wgrad(cat[:, :T].contiguous().view(B * T, Kc), dg.view(B * T, GH),
                  out_grad.view(GH, Kc), accumulate=True)


class RNNDecoderWithAttention(Decoder):
  @staticmethod
  def get_required_params():
    return dict(Decoder.get_required_params(), **{
        'GO_SYMBOL': int, 'END_SYMBOL': int, 'tgt_vocab_size': int, 'tgt_emb_size': int,
        'attention_layer_size': int, 'attention_type': ['bahdanau', 'luong', 'gnmt', 'gnmt_v2'],
        'core_cell': None, 'decoder_layers': int, 'decoder_use_skip_connections': bool,
        'batch_size': int,
    })

  @staticmethod
  def get_optional_params():
    return dict(Decoder.get_optional_params(), **{
        'core_cell_params': dict, 'bahdanau_normalize': bool, 'luong_scale': bool,
        'decoder_dp_input_keep_prob': float, 'decoder_dp_output_keep_prob': float,
        'time_major': bool, 'use_swap_memory': bool, 'proj_size': int, 'num_groups': int,
        'PAD_SYMBOL': int, 'weight_tied': bool,
    })

  def __init__(self, params, model, name='rnn_decoder_with_attention', mode='train'):
    super(RNNDecoderWithAttention, self).__init__(params, model, name, mode)
    p = self.params
    self._batch_size = p['batch_size']
    self.GO_SYMBOL, self.END_SYMBOL = p['GO_SYMBOL'], p['END_SYMBOL']
    self._tgt_vocab_size, self._tgt_emb_size = p['tgt_vocab_size'], p['tgt_emb_size']
    self._weight_tied = bool(p.get('weight_tied', False))
    if p['decoder_use_skip_connections'] and not p['attention_type'].startswith('gnmt'):
      raise NotImplementedError("decoder_use_skip_connections outside the GNMT attention cells")
    if p.get('decoder_dp_output_keep_prob', 1.0) != 1.0:
      raise NotImplementedError("decoder_dp_output_keep_prob != 1.0")
    if p['attention_type'] == 'luong' and p.get('luong_scale', False):
      raise NotImplementedError("luong_scale=True (the learned scalar of LuongAttention)")
    if p.get('time_major', False):
      raise NotImplementedError("time_major")
    self._skip = bool(p['decoder_use_skip_connections'])

  def build(self, store, memory_dim=None):
    p = self.params
    first_param = len(store.params)
    cell, H, fb = cell_spec(p['core_cell'], p.get('core_cell_params', {}))
    self.H, self.U = H, p['attention_layer_size']
    self.M = memory_dim if memory_dim is not None else p.get('_memory_dim', 2 * H)
    scope = "ForwardPass/" + self._name
    V, E = self._tgt_vocab_size, self._tgt_emb_size
    self.Vpad = _round8(V)
    if self._weight_tied and not (p['attention_type'].startswith('gnmt') and E == H):
      raise NotImplementedError("weight_tied needs a GNMT decoder whose top cell size equals tgt_emb_size")
    at = p['attention_type']
    self.gnmt = at.startswith('gnmt')
    out_in = H if self.gnmt else self.M

    def init(shape):
      lim = math.sqrt(6.0 / (out_in + V))
      w = (torch.rand(shape) * 2 - 1) * lim
      w[:, V:, :] = 0.0          # vocabulary padding rows
      return w

    self.out_in = out_in
    if self._weight_tied:
      # the embedding IS the (transposed) output projection. The shared variable sits where the
      # embedding matrix would (first decoder variable): its gradient is complete only after the
      # embedding backward, the last decoder closure to run, and the overlapped gradient reducer
      # relies on variables becoming final in reverse creation order.
      self.proj = store.add(scope + "/dense/kernel", (1, self.Vpad, out_in), init, kind="conv",
                            logical_out=V)
      self.embedding = Embedding(store, None, V, E, table=self.proj)
    else:
      self.embedding = Embedding(store, scope + "/DecoderEmbeddingMatrix", V, E)
    nl = p['decoder_layers']
    if self.gnmt:
      mode = capi.SCORE_BAHDANAU_NORM
      loop_layers = 1
    else:
      mode = capi.SCORE_BAHDANAU_NORM if p.get('bahdanau_normalize', False) else capi.SCORE_BAHDANAU
      if at == 'luong':
        mode = capi.SCORE_LUONG
      loop_layers = nl
      if nl > 2:
        raise NotImplementedError("more than 2 layers inside the attention loop")
    self.cell = AttentionCell(store, scope + "/attention_cell", E, H, self.M, self.U, loop_layers,
                              mode, fb)
    self.upper = []
    if self.gnmt:
      for l in range(1, nl):
        self.upper.append(RNNDirection(store, "%s/multi_rnn_cell/cell_%d/lstm_cell" % (scope, l),
                                       cell, [H, self.M], H, reverse=False, forget_bias=fb))
    if not self._weight_tied:
      self.proj = store.add(scope + "/dense/kernel", (1, self.Vpad, out_in), init, kind="conv",
                            logical_out=V)
    apply_scope_initializer(store, first_param, p)
    return self

  # ---------------------------------------------------------------- training / scoring pass
  def _decode(self, input_dict):
    enc = input_dict['encoder_output']
    enc_act = enc.get('outputs_act') or Act(enc['outputs'], enc['src_lengths'], requires_grad=False)
    src_len = enc['src_lengths']
    training = self._mode == "train"
    if not training and self._mode in ("eval", "infer"):
      return self._greedy(enc_act, src_len)
    tape = input_dict.get('tape')
    seeds = enc.get('seeds') or SeedSeq(23)
    tgt, tgt_len = input_dict['target_tensors'][0], input_dict['target_tensors'][1]
    B, T = tgt.shape
    keep = self.params.get('decoder_dp_input_keep_prob', 1.0)
    emb = self.embedding.lookup(tgt.reshape(-1).contiguous(), tape, keep, seeds.next())
    x = Act(emb.data.view(B, T, -1), tgt_len)
    if tape is not None:
      def view_bwd():
        emb.grad, emb.grad_init = x.grad.reshape(B * T, -1), True
        x.grad = None
      tape.record(view_bwd)
    y, ctx, loop = self.cell.forward_train(x, enc_act, src_len, tgt_len, tape, seeds,
                                           attn_in_keep=keep)
    top = y
    if self.gnmt:
      att_in = ctx
      if self.params['attention_type'] == 'gnmt':      # upper layers see the PREVIOUS attention
        att_in = _shift_time(ctx, tape)
      for li, layer in enumerate(self.upper):
        xs = [dropout_act(top, keep, seeds.next(), tape), dropout_act(att_in, keep, seeds.next(), tape)]
        y = rnn_directions_forward([layer], xs, tgt_len, tape)[0]
        # _add_residual_wrapper(cells, start_ind=1) with gnmt_residual_fn (rnn_decoders.py:138-146,
        # parts/rnns/gnmt.py): from the second upper layer on, out += the layer-input part of the
        # cell inputs (not the attention part, not dropped)
        top = residual_add(y, top, tape) if (self._skip and li >= 1) else y
    else:
      top = ctx      # AttentionWrapper default output_attention=True: the attention vector
    feat = top.data.reshape(B * T, self.out_in)
    logits2 = capi.gemm(feat, self.proj.w16.view(self.Vpad, self.out_in))
    logits = Act(logits2.view(B, T, self.Vpad), tgt_len)
    if tape is not None:
      dec = self

      def backward():
        dl = logits.grad.reshape(B * T, dec.Vpad)
        capi.gemm_wgrad(feat, dl, dec.proj.grad.view(dec.Vpad, dec.out_in), accumulate=True)
        g = top.grad_buffer()
        capi.gemm(dl, dec.proj.wt16.view(dec.out_in, dec.Vpad), out=g.view(B * T, dec.out_in),
                  accumulate=top.grad_init)
        top.grad_init = True
        logits.grad = None

      tape.record(backward, [] if self._weight_tied else [self.proj])
    return {'logits': logits.data, 'logits_act': logits, 'vocab_size': self._tgt_vocab_size,
            'outputs': None, 'final_state': None, 'final_sequence_lengths': tgt_len,
            'lazy_outputs': lambda: [capi.argmax_rows(logits2, self._tgt_vocab_size).view(B, T)]}

  # ---------------------------------------------------------------- greedy decoding
  def _greedy(self, enc_act, src_len):
    """GreedyEmbeddingHelper + dynamic_decode(impute_finished=True,
    maximum_iterations = 2 * max(src_len)) (rnn_decoders.py:283-315)."""
    B, S, M = enc_act.data.shape
    dev = enc_act.data.device
    T = 2 * int(src_len.max().item())
    H, GH, Vp, V = self.H, 4 * self.H, self.Vpad, self._tgt_vocab_size
    seeds = SeedSeq(29)
    cell = self.cell
    loop = cell._new_loop(B, T, S, dev, False, 1.0, 1.0, seeds)
    keys = cell.memory(enc_act, None)
    gx0 = torch.zeros((B, T, GH), dtype=torch.bfloat16, device=dev)
    loop.set_inputs(gx0, keys, enc_act.data, src_len, None)
    ids = torch.full((B,), self.GO_SYMBOL, dtype=torch.int32, device=dev)
    finished = torch.zeros((B,), dtype=torch.bool, device=dev)
    lengths = torch.zeros((B,), dtype=torch.int32, device=dev)
    out_ids = torch.zeros((B, T), dtype=torch.int32, device=dev)
    logits_all = torch.zeros((B, T, Vp), dtype=torch.bfloat16, device=dev)
    steps = 0
    for t in range(T):
      e = capi.embed_fwd(ids, None, self.embedding.table.w16, 1.0, 1.0, 0, plain=True)
      gx0[:, t] = capi.gemm(e, cell.w_in.w16.view(GH, -1), bias=cell.bias[0].master)
      loop.forward(t, t + 1)
      if self.gnmt:
        top = Act(loop.y_top[:, :t + 1].contiguous())
        att = loop.ctx[:, :t + 1]
        if self.params['attention_type'] == 'gnmt':
          att = torch.cat([torch.zeros_like(att[:, :1]), att[:, :-1]], 1)
        att = Act(att.contiguous())
        for li, layer in enumerate(self.upper):   # upper layers re-run over the prefix (state is implicit)
          y = rnn_directions_forward([layer], [top, att], None, None)[0]
          top = residual_add(y, top, None) if (self._skip and li >= 1) else y
        feat = top.data[:, t].contiguous()
      else:
        feat = loop.ctx[:, t].contiguous()
      lg = capi.gemm(feat, self.proj.w16.view(Vp, self.out_in))
      nxt = capi.argmax_rows(lg, V)
      # impute_finished: finished samples emit zeros and keep their state
      logits_all[:, t] = torch.where(finished[:, None], torch.zeros_like(lg), lg)
      out_ids[:, t] = torch.where(finished, torch.zeros_like(nxt), nxt)
      lengths += (~finished).to(torch.int32)
      finished = finished | (nxt == self.END_SYMBOL)
      ids = nxt
      steps = t + 1
      if bool(finished.all()):
        break
    return {'logits': logits_all[:, :steps], 'outputs': [out_ids[:, :steps]], 'final_state': None,
            'final_sequence_lengths': lengths, 'vocab_size': V}


class BeamSearchRNNDecoderWithAttention(RNNDecoderWithAttention):
  """open_seq2seq/decoders/rnn_decoders.py:324-532: the same variables as RNNDecoderWithAttention,
  decoded with tf.contrib.seq2seq.BeamSearchDecoder (beam_width, length_penalty =
  length_penalty_weight) for at most 2 x max source length steps; outputs =
  predicted_ids[:, :, 0]. The per-step scoring / top-k / state update runs on the device
  (os2s_tf_beam_step); decoder state rows are re-gathered by parent beam every step."""

  @staticmethod
  def get_optional_params():
    return dict(RNNDecoderWithAttention.get_optional_params(), **{
        'length_penalty': float, 'beam_width': int,
    })

  def __init__(self, params, model, name='rnn_decoder_with_attention', mode='train'):
    super(BeamSearchRNNDecoderWithAttention, self).__init__(params, model, name, mode)
    if self._mode != 'infer':
      raise ValueError("BeamSearch decoder only supports infer mode, but got {}".format(self._mode))
    self._length_penalty_weight = self.params.get('length_penalty', 0.0)
    self._beam_width = self.params.get('beam_width', 1)

  def _decode(self, input_dict):
    enc = input_dict['encoder_output']
    enc_act = enc.get('outputs_act') or Act(enc['outputs'], enc['src_lengths'], requires_grad=False)
    return self._beam_search(enc_act, enc['src_lengths'])

  @staticmethod
  def _reorder(view, parent):
    tmp = view.contiguous()
    view.copy_(capi.gather_rows(tmp, parent))

  def _beam_search(self, enc_act, src_len):
    import numpy as np
    B, S, M = enc_act.data.shape
    W = self._beam_width
    N = B * W
    dev = enc_act.data.device
    T = 2 * int(src_len.max().item())
    H, GH, Vp, V = self.H, 4 * self.H, self.Vpad, self._tgt_vocab_size
    # tile_batch: every sentence repeated beam_width times
    tile = (torch.arange(N, dtype=torch.int32, device=dev) // W).to(torch.int32)
    mem = Act(capi.gather_rows(enc_act.data.contiguous(), tile), None)
    slen = capi.gather_rows(src_len.to(torch.int32).view(B, 1).contiguous(), tile).view(N)
    seeds = SeedSeq(31)
    cell = self.cell
    loop = cell._new_loop(N, T, S, dev, False, 1.0, 1.0, seeds)
    keys = cell.memory(mem, None)
    gx0 = torch.zeros((N, T, GH), dtype=torch.bfloat16, device=dev)
    loop.set_inputs(gx0, keys, mem.data, slen, None)
    st = capi.TfBeamState(B, W, V, self.END_SYMBOL, self._length_penalty_weight, dev)
    ids = torch.full((N,), self.GO_SYMBOL, dtype=torch.int32, device=dev)
    hist_ids, hist_par = [], []
    for t in range(T):
      e = capi.embed_fwd(ids, None, self.embedding.table.w16, 1.0, 1.0, 0, plain=True)
      gx0[:, t] = capi.gemm(e, cell.w_in.w16.view(GH, -1), bias=cell.bias[0].master)
      loop.forward(t, t + 1)
      if self.gnmt:
        top = Act(loop.y_top[:, :t + 1].contiguous())
        att = loop.ctx[:, :t + 1]
        if self.params['attention_type'] == 'gnmt':
          att = torch.cat([torch.zeros_like(att[:, :1]), att[:, :-1]], 1)
        att = Act(att.contiguous())
        for li, layer in enumerate(self.upper):
          y = rnn_directions_forward([layer], [top, att], None, None)[0]
          top = residual_add(y, top, None) if (self._skip and li >= 1) else y
        feat = top.data[:, t].contiguous()
      else:
        feat = loop.ctx[:, t].contiguous()
      lg = capi.gemm(feat, self.proj.w16.view(Vp, self.out_in))
      st.step(lg, t)
      hist_ids.append(st.word_ids.clone())
      hist_par.append(st.parent.clone())
      ids = st.word_ids.clone()
      # the next step reads these state rows: they follow their beams
      for l in range(cell.L):
        self._reorder(loop.cat[l][:, t + 1], st.parent)
        self._reorder(loop.c_seq[l][:, t], st.parent)
      if loop.cum_seq is not None:
        self._reorder(loop.cum_seq[:, t + 1], st.parent)
      if self.gnmt:      # the upper layers re-run over the prefix: it is part of the beam state
        self._reorder(loop.y_top[:, :t + 1], st.parent)
        self._reorder(loop.ctx[:, :t + 1], st.parent)
      if bool(st.finished.all()):
        break
    steps = len(hist_ids)
    step_ids = torch.stack(hist_ids).view(steps, B, W).cpu().numpy()
    parents = (torch.stack(hist_par).view(steps, B, W).cpu().numpy() % W)
    lengths = st.lengths.view(B, W).cpu().numpy()
    pred = np.full((steps, B, W), self.END_SYMBOL, np.int32)     # gather_tree (finalize)
    for b in range(B):
      L = min(int(lengths[b].max()), steps)
      for w in range(W):
        parent = w
        for t in range(L - 1, -1, -1):
          pred[t, b, w] = step_ids[t, b, parent]
          parent = parents[t, b, parent]
        seen = False
        for t in range(L):
          if seen:
            pred[t, b, w] = self.END_SYMBOL
          elif pred[t, b, w] == self.END_SYMBOL:
            seen = True
    predicted_ids = torch.from_numpy(np.ascontiguousarray(np.transpose(pred, (1, 0, 2)))).to(dev)
    top = predicted_ids[:, :, 0].contiguous()
    return {'logits': top, 'outputs': [top], 'predicted_ids': predicted_ids,
            'scores': st.scores.view(B, W), 'final_state': None,
            'final_sequence_lengths': torch.from_numpy(lengths[:, 0].astype(np.int32)).to(dev),
            'beam_sequence_lengths': lengths, 'vocab_size': V}


def _shift_time(x, tape):
  """y[:, t] = x[:, t-1] (zeros at t = 0): the attention of the previous step."""
  y = torch.zeros_like(x.data)
  y[:, 1:] = x.data[:, :-1]
  out = Act(y, x.lens)
  if tape is not None:
    def backward():
      g = torch.zeros_like(out.grad)
      g[:, :-1] = out.grad[:, 1:]
      if x.grad_init and x.grad is not None:
        capi.add_bf16(x.grad, g, out=x.grad)
      else:
        x.grad, x.grad_init = g, True
      out.grad = None
    tape.record(backward)
  return out

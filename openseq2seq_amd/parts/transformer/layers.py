"""Building blocks of the Transformer NMT path on packed token-major tensors
(host layer over the HIP kernels; same arithmetic as
open_seq2seq/parts/transformer/{attention_layer,ffn_layer,common,embedding_layer}.py).

Every block enqueues its forward kernels and records ONE backward closure on the
Tape (see parts/cnns/conv_blocks.py). Activations are `Act` holders with 2-D bf16
data [N_tokens, hidden]. Padding never exists in this layout, so the reference's
FFN "remove_padding" (ffn_layer.py:56-70) is implicit and extends to every token-wise
op (LayerNorm, all projections), and padded keys need no -1e9 bias.
"""
import math

import torch

from ... import capi
from ..cnns.conv_blocks import Act, on_side_stream, current_tape


SKINNY_MAX_ROWS = 512
import os as _os
# Dense weight gradients on the side stream (as the conv families do): most Dense GEMMs of a
# Transformer-big step are 132 tiles on 256 CUs (8300 tokens x 1024 columns), the split weight
# gradient fills the other half of the chip. 22.1 -> 20.3 ms/step, sustained over 300 steps
# (OS2S_DENSE_WGRAD_STREAM=0 keeps them on the main stream)
DENSE_WGRAD_STREAM = _os.environ.get("OS2S_DENSE_WGRAD_STREAM", "1") == "1"
# the ReLU + dropout backward of a Dense layer fused into the data-gradient GEMM of its consumer
# (os2s_gemm_nt_mask_ws); OS2S_FUSE_RELU_BWD=0 = the separate dropout_bwd_colsum pass of round 2
FUSE_RELU_BWD = _os.environ.get("OS2S_FUSE_RELU_BWD", "1") == "1"
# Dense weight gradients with fewer than 32 output tiles of 256 x 256 (the 1024 x 1024 projections)
# are collected three at a time into one ping-pong launch (OS2S_GROUP_SMALL_WGRAD=0: one lockstep
# launch with fp32 atomics each, as in round 2)
GROUP_SMALL_WGRAD = _os.environ.get("OS2S_GROUP_SMALL_WGRAD", "1") == "1"


def _small_wgrad(lin, dz):
  units = ((lin.cout + 255) // 256) * ((lin.cin + 255) // 256)
  return units < 32 and lin.cout >= 128 and lin.cin >= 128 and dz.shape[0] >= 2048 and \
      lin.cout % 8 == 0 and lin.cin % 8 == 0


# Round 6: EVERY Dense weight gradient with fewer output tiles than the chip has CUs is held back until the
# collected ones cover WGRAD_UNIT_BUDGET tiles of 256 x 256 (Transformer-big: ffn 64 + 64, q k v 48, the 1024 x 1024
# projections 16 each — 192 / 224 per encoder / decoder layer), then go out as ONE launch of the ping-pong TN-GEMM
# kernel: ~200 tiles of 130 reduction steps each and no reduction split, where the single launches were 16 - 64 tiles
# cut 4 - 16 ways (fill, 256 KB slab per piece, one reducer per tile). Transformer-big, same box, interleaved
# (ms per step): round-5 policy 18.30, budget 128: 17.56, 192: 17.27 - 17.39, 256: 17.85 — a launch that leaves a
# quarter of the CUs to the data-gradient chain on the main stream beats one that takes them all.
# OS2S_WGRAD_UNIT_BUDGET=0: the round-5 policy (1024 x 1024 projections three at a time, the rest alone).
WGRAD_UNIT_BUDGET = int(_os.environ.get("OS2S_WGRAD_UNIT_BUDGET", "192"))


def _groupable_wgrad(lin, dz):
  units = ((lin.cout + 255) // 256) * ((lin.cin + 255) // 256)
  return WGRAD_UNIT_BUDGET > 0 and units < 256 and lin.cout >= 128 and lin.cin >= 128 and dz.shape[0] >= 2048 and \
      lin.cout % 8 == 0 and lin.cin % 8 == 0 and dz.stride(1) == 1


SKINNY_LOGITS = False    # [256 x 32768 x 1024]: the LDS-tiled kernel wins (60 vs 139 us)


class SeedSeq(object):
  """Distinct dropout streams per op per step."""

  def __init__(self, base):
    self.base, self.n = int(base), 0

  def next(self):
    self.n += 1
    return (self.base * 1000003 + self.n) & ((1 << 62) - 1)


def _colsum_into(dy2d, bias_param):
  """bias.grad += column sums of dy (bf16 [N, C])."""
  C = dy2d.shape[1]
  part = capi.bn_stats(dy2d)
  scratch = torch.empty(2, C, dtype=torch.float32, device=dy2d.device)
  capi.bn_bwd_finalize(part, 1, 1, None, bias_param.grad, True, scratch[0], scratch[1])


def _accumulate_grad(x, dx):
  if not x.requires_grad:
    return
  assert not x.grad_masked, "a second consumer wrote to an activation whose gradient was finalised"
  if x.grad_init and x.grad is not None:
    capi.add_bf16(x.grad, dx, out=x.grad)
  else:
    x.grad = dx
    x.grad_init = True


class Dense(object):
  """tf.layers.Dense on [N, Cin] rows; kernel stored [1, Cout, Cin] (device layout,
  = the transpose of TF's [Cin, Cout])."""

  def __init__(self, store, name, cin, cout, use_bias):
    self.cin, self.cout = cin, cout

    def init(shape):   # tf.layers.Dense default initializer: glorot_uniform
      lim = math.sqrt(6.0 / (cin + cout))
      return (torch.rand(shape) * 2 - 1) * lim

    self.kernel = store.add(name + "/kernel", (1, cout, cin), init, kind="conv")
    self.bias = store.add(name + "/bias", (cout,), torch.zeros(cout), kind="vector") \
        if use_bias else None

  @property
  def w(self):
    return self.kernel.w16.view(self.cout, self.cin)

  def forward(self, x, tape, act=0, keep=1.0, seed=0, residual=None):
    """y = residual + dropout(act(x W^T + b)); x, residual: Act; returns Act."""
    if tape is None and keep >= 1.0 and x.data.shape[0] <= SKINNY_MAX_ROWS:
      # decoding step: a few hundred rows — latency-bound kernel (csrc/gemm_skinny.hip)
      return Act(capi.gemm_skinny(x.data, self.w, bias=self.bias.master if self.bias is not None else None,
                                  relu=(act == 1), residual=residual.data if residual is not None else None))
    bias = self.bias.master if self.bias is not None else None
    res = residual.data if residual is not None else None
    if self.cin % 64 == 0 and act in (0, 1, 3) and self.cout % 8 == 0:   # what gemm_pp.hip accepts
      y = capi.gemm_nt(x.data, self.w, bias=bias, act=act, keep_prob=keep, seed=seed, residual=res)
    else:
      y = capi.gemm(x.data, self.w, bias=bias, act=act, keep_prob=keep, seed=seed, residual=res)
    out = Act(y)
    if tape is None:
      return out
    lin = self
    assert not (act in (1, 3) and residual is not None)
    if act == 1 and FUSE_RELU_BWD and y.is_contiguous():
      out.mask_scale = 1.0 / keep       # y = dropout(relu(.)): zero exactly where the gradient is

    def backward():
      dy = out.grad
      assert dy is not None, "no gradient reached " + lin.kernel.name
      bias_part = None          # partial column sums of dz when the same pass can produce them
      fuse = lin.bias is not None and dy.is_contiguous()
      if act == 1 and out.grad_masked:
        # the consumer's data-gradient GEMM applied (y > 0) / keep in its epilogue
        dz, bias_part = dy, out.bias_part
        out.bias_part = None
      elif act in (1, 3):
        # act 3 = min(relu(.), 20): no gradient where the stored output sits at the cap either
        if fuse:
          dz, bias_part = capi.dropout_bwd_colsum(dy, keep, out=y, capped=(act == 3))
        else:
          dz = capi.dropout_bwd(dy, keep, out=y, capped=(act == 3))     # (y > 0) / keep
      elif keep < 1.0:
        if fuse:
          dz, bias_part = capi.dropout_bwd_colsum(dy, keep, seed=seed)
        else:
          dz = capi.dropout_bwd(dy, keep, seed=seed)     # recomputed hash mask / keep
      else:
        dz = dy
      # dW += dz^T x (fp32) and dx (+)= dz W: plain GEMMs. The weight gradient goes to the side
      # stream by default (OS2S_DENSE_WGRAD_STREAM): with the in-tree kernels it fills the half of
      # the chip a 132-tile data-gradient GEMM leaves idle, 22.1 -> 20.3 ms/step over 20 AND over
      # 300 steps (with the round-1 vendor GEMMs the same move lost 11 % at the power limit)
      if GROUP_SMALL_WGRAD and _groupable_wgrad(lin, dz) and current_tape() is not None:
        # (OS2S_DENSE_WGRAD_STREAM=0 keeps the grouped launch on the main stream: the serial profiles)
        current_tape().defer_wgrad(lin.kernel, dict(x=x.data, dy=dz, dw=lin.kernel.grad.view(lin.cout, lin.cin)),
                                   unit_budget=WGRAD_UNIT_BUDGET, side=DENSE_WGRAD_STREAM)
      elif DENSE_WGRAD_STREAM and GROUP_SMALL_WGRAD and _small_wgrad(lin, dz) and current_tape() is not None:
        # 16 output tiles: three of these go out as ONE launch (Tape.defer_wgrad)
        current_tape().defer_wgrad(lin.kernel, dict(x=x.data, dy=dz, dw=lin.kernel.grad.view(lin.cout, lin.cin)))
      elif DENSE_WGRAD_STREAM:
        with on_side_stream(dz.device, x.data, dz):
          capi.gemm_wgrad(x.data, dz, lin.kernel.grad.view(lin.cout, lin.cin), accumulate=True)
      else:
        capi.gemm_wgrad(x.data, dz, lin.kernel.grad.view(lin.cout, lin.cin), accumulate=True)
      if bias_part is not None and lin.bias is not None:
        with on_side_stream(dz.device, bias_part):        # parameter gradient: off the main chain
          scratch = torch.empty(2, lin.cout, dtype=torch.float32, device=dz.device)
          capi.bn_bwd_finalize(bias_part, 1, 1, None, lin.bias.grad, True, scratch[0], scratch[1])
      elif lin.bias is not None:
        with on_side_stream(dz.device, dz):
          _colsum_into(dz, lin.bias)
      if x.requires_grad:
        g = x.grad_buffer()
        if x.mask_scale is not None and not x.grad_init and lin.cout % 64 == 0 and lin.cin % 8 == 0 and \
            dz.stride(1) == 1 and g.is_contiguous():
          # x = dropout(relu(.)) of the layer below, this is the only consumer: its activation
          # backward (and the bias-gradient partials) ride in this GEMM's epilogue
          _, x.bias_part = capi.gemm_nt_mask(dz, lin.kernel.wt16.view(lin.cin, lin.cout), x.data, x.mask_scale,
                                             out=g, want_colsum=True)
          x.grad_masked = True
        else:
          capi.gemm(dz, lin.kernel.wt16.view(lin.cin, lin.cout), out=g, accumulate=x.grad_init)
        x.grad_init = True
      if residual is not None:
        residual.res_grad = dy      # consumed by the pre-norm LayerNorm backward of `residual`
      out.grad = None

    tape.record(backward, [lin.kernel] + ([lin.bias] if lin.bias is not None else []))
    return out


class LayerNorm(object):
  """LayerNormalization 'layernorm_L2' (common.py:41-68): fp32 scale/bias, eps 1e-6."""

  def __init__(self, store, name, hidden, eps=1e-6):
    self.scale = store.add(name + "/layer_norm_scale", (hidden,), torch.ones(hidden), kind="vector")
    self.bias = store.add(name + "/layer_norm_bias", (hidden,), torch.zeros(hidden), kind="vector")
    self.eps = eps

  def forward(self, x, tape):
    training = tape is not None
    y, mean, rstd = capi.layernorm_fwd(x.data, self.scale.master, self.bias.master, self.eps,
                                       save=training)
    out = Act(y)
    if not training:
      return out
    ln = self

    def backward():
      dy = out.grad
      assert dy is not None
      dres, x.res_grad = x.res_grad, None
      dx, finish, partial = capi.layernorm_bwd(dy, x.data, ln.scale.master, mean, rstd, dres, ln.scale.grad,
                                               ln.bias.grad, defer_param_grads=True)
      with on_side_stream(dx.device, partial):      # scale / bias gradients: off the main chain
        finish()
      _accumulate_grad(x, dx)
      out.grad = None

    tape.record(backward, [ln.scale, ln.bias])
    return out


class MultiHeadAttention(object):
  """Attention / SelfAttention (attention_layer.py:23-227), 'loung' mode, no biases.
  Self-attention uses one fused [3D, D] projection for q,k,v; enc-dec attention a [D, D]
  query projection and a fused [2D, D] key/value projection. (The reference keeps q, k, v
  as three Dense kernels; fusing only changes how the same numbers are laid out.)"""

  def __init__(self, store, name, hidden, num_heads, self_attention, kv=None):
    self.D, self.H, self.self_att = hidden, num_heads, self_attention
    self.scale = (hidden // num_heads) ** -0.5
    if hidden // num_heads != 64:
      raise NotImplementedError("HIP attention kernel is built for head dim 64")
    if self_attention:
      self.qkv = Dense(store, name + "/qkv", hidden, 3 * hidden, False)
    else:
      self.q = Dense(store, name + "/q", hidden, hidden, False)
      # (the decoder creates the key / value projections of all its layers next to each other — FusedCrossKV)
      self.kv = kv if kv is not None else Dense(store, name + "/kv", hidden, 2 * hidden, False)
    self.out = Dense(store, name + "/output_transform", hidden, hidden, False)

  def forward(self, x, y, cu_q, cu_k, max_len, causal, tape, seeds, att_keep, post_keep, residual, kv_pre=None):
    """x: queries source (Act [Nq,D]); y: keys/values source (Act [Nk,D]) — y is x for
    self-attention. Returns residual + dropout(W_o attention). kv_pre = (Act [Nk, n * 2D], l): the key / value
    projections of n layers computed by ONE GEMM (FusedCrossKV), this layer's are columns [l * 2D, (l + 1) * 2D)."""
    D, H = self.D, self.H
    kv_all = None
    if self.self_att:
      qkv = self.qkv.forward(x, tape)
      qv, kv_, vv = qkv.data[:, :D], qkv.data[:, D:2 * D], qkv.data[:, 2 * D:]
    elif kv_pre is not None:
      q = self.q.forward(x, tape)
      kv_all, lidx = kv_pre
      kd = kv_all.data[:, lidx * 2 * D:(lidx + 1) * 2 * D]
      qv, kv_, vv = q.data, kd[:, :D], kd[:, D:]
    else:
      q = self.q.forward(x, tape)
      kv = self.kv.forward(y, tape)
      qv, kv_, vv = q.data, kv.data[:, :D], kv.data[:, D:]
    seed = seeds.next()
    o, lse = capi.attention_fwd(qv, kv_, vv, cu_q, cu_k, H, max_len, causal, self.scale,
                                att_keep, seed)
    oa = Act(o)
    if tape is not None:
      att = self

      def backward():
        d_o = oa.grad
        assert d_o is not None
        if att.self_att:
          g = torch.empty_like(qkv.data)
          capi.attention_bwd(qv, kv_, vv, d_o, lse, g[:, :D], g[:, D:2 * D], g[:, 2 * D:], cu_q,
                             cu_k, H, max_len, causal, att.scale, att_keep, seed)
          qkv.grad = g
        elif kv_all is not None:
          # this layer's columns of the fused gradient; FusedCrossKV's closure runs after every layer's
          gq = torch.empty_like(q.data)
          gkv = kv_all.grad_buffer()[:, lidx * 2 * D:(lidx + 1) * 2 * D]
          capi.attention_bwd(qv, kv_, vv, d_o, lse, gq, gkv[:, :D], gkv[:, D:], cu_q, cu_k, H,
                             max_len, causal, att.scale, att_keep, seed)
          q.grad = gq
        else:
          gq = torch.empty_like(q.data)
          gkv = torch.empty_like(kv.data)
          capi.attention_bwd(qv, kv_, vv, d_o, lse, gq, gkv[:, :D], gkv[:, D:], cu_q, cu_k, H,
                             max_len, causal, att.scale, att_keep, seed)
          q.grad, kv.grad = gq, gkv
        oa.grad = None

      tape.record(backward)
    return self.out.forward(oa, tape, keep=post_keep, seed=seeds.next(), residual=residual)


# A/B knob: 0 = one key / value GEMM per decoder layer (rounds 1 - 4)
FUSE_CROSS_KV = _os.environ.get("OS2S_FUSE_CROSS_KV", "1") == "1"
FUSE_CROSS_KV_SIDE = _os.environ.get("OS2S_FUSE_CROSS_KV_SIDE", "1") == "1"


class FusedCrossKV(object):
  """The key / value projections of the encoder output for ALL decoder layers' encoder-decoder attention
  (transformer_decoder.py:155-230: every layer projects the same encoder output with its own k, v kernels) as ONE
  GEMM with N = n_layers * 2D columns — 1584 tiles instead of six launches of 264 for Transformer-big — and ONE
  TN GEMM for the six kernel gradients. The kernels stay six variables under their reference names; they are created
  next to each other, so their bf16 copies (and their gradients) ARE the rows of one [n * 2D, D] matrix. The data
  gradient into the encoder output stays one GEMM per layer (the transposed weight copies are per kernel)."""

  def __init__(self, store, kvs):
    self.store, self.kvs = store, kvs
    self.D = kvs[0].cin

  def join(self):
    """The current stream waits for the fused projection (no-op when it ran on the current stream)."""
    if getattr(self, "_side", None) is not None:
      torch.cuda.current_stream().wait_stream(self._side)
      self._side = None

  def usable(self):
    ks = [d.kernel for d in self.kvs]
    return FUSE_CROSS_KV and len(ks) > 1 and all(b.offset == a.offset + a.numel and b.numel == a.numel
                                                 for a, b in zip(ks, ks[1:]))

  def _rows(self, flat):
    k0, n = self.kvs[0].kernel, len(self.kvs)
    return flat[k0.offset:k0.offset + n * k0.numel].view(n * 2 * self.D, self.D)

  def forward(self, enc_out, tape):
    # on the side stream: nothing of the decoder needs the result before its first encoder-decoder attention, and
    # the embedding + self-attention sublayer in front of it are 132-tile launches that leave half the chip idle
    # (the decoder calls join() there)
    self._side = None
    if FUSE_CROSS_KV_SIDE:
      with on_side_stream(enc_out.data.device, enc_out.data) as ctx:
        y = capi.gemm_nt(enc_out.data, self._rows(self.store.w16))
        ctx.hand_over(y)
        self._side = ctx.side
    else:
      y = capi.gemm_nt(enc_out.data, self._rows(self.store.w16))
    out = Act(y)
    if tape is None:
      return out
    fused, D = self, self.D

    def backward():
      g = out.grad
      assert g is not None, "no decoder layer wrote the fused key / value gradient"
      with on_side_stream(g.device, enc_out.data, g):
        capi.gemm_wgrad(enc_out.data, g, fused._rows(fused.store.grads), accumulate=True)
      if enc_out.requires_grad:
        ge = enc_out.grad_buffer()
        for l, d in enumerate(fused.kvs):
          gl = g[:, l * 2 * D:(l + 1) * 2 * D]
          if g.shape[0] < capi.BIG_TILE_MIN_ROWS:     # the small-batch kernel wants contiguous rows
            gl = gl.contiguous()
          capi.gemm(gl, d.kernel.wt16.view(d.cin, d.cout), out=ge, accumulate=enc_out.grad_init)
          enc_out.grad_init = True
      out.grad = None

    tape.record(backward, [d.kernel for d in self.kvs])
    return out


class FeedForward(object):
  """FeedFowardNetwork (ffn_layer.py:25-85): Dense(filter, relu) -> dropout -> Dense(hidden)."""

  def __init__(self, store, name, hidden, filter_size):
    self.filter_layer = Dense(store, name + "/filter_layer", hidden, filter_size, True)
    self.output_layer = Dense(store, name + "/output_layer", filter_size, hidden, True)

  def forward(self, x, tape, seeds, relu_keep, post_keep, residual):
    s1, s2 = (seeds.next(), seeds.next()) if seeds is not None else (0, 0)
    h = self.filter_layer.forward(x, tape, act=1, keep=relu_keep, seed=s1)
    return self.output_layer.forward(h, tape, keep=post_keep, seed=s2, residual=residual)


class SharedEmbedding(object):
  """EmbeddingSharedWeights (embedding_layer.py:26-105): one [V, D] matrix used for the
  input embeddings of both stacks and, transposed, for the pre-softmax projection."""

  def __init__(self, store, name, vocab_size, hidden, pad_vocab_to_eight=False):
    if pad_vocab_to_eight and vocab_size % 8:
      vocab_size += 8 - vocab_size % 8
    if vocab_size % 8:
      raise NotImplementedError("vocab size must be a multiple of 8 (use pad_vocab_to_eight)")
    self.V, self.D = vocab_size, hidden

    def init(shape):   # random_normal_initializer(0, hidden**-0.5)
      return torch.randn(shape) * hidden ** -0.5

    self.weights = store.add(name + "/embedding_and_softmax/weights", (1, vocab_size, hidden),
                             init, kind="conv")

  @property
  def table(self):
    return self.weights.w16.view(self.V, self.D)

  def embed(self, ids, pos, tape, keep, seed, final_use=False):
    """final_use: True for the FIRST use in forward order (= the last closure of the
    backward pass that touches the shared weights; only then are their gradients final)."""
    out = Act(capi.embed_fwd(ids, pos, self.table, self.D ** 0.5, keep, seed))
    if tape is not None:
      emb = self

      def backward():
        if out.grad is not None:
          # a parameter gradient: side stream, like every other (all three writers of the shared
          # table — the softmax weight gradient and both embedding scatters — sit on that ONE stream,
          # in order: the scatter's atomics must not interleave with the GEMM's read-modify-write)
          with on_side_stream(out.grad.device, ids, out.grad):
            capi.embed_bwd(ids, out.grad, emb.weights.grad.view(emb.V, emb.D), emb.D ** 0.5, keep,
                           seed)
        out.grad = None

      tape.record(backward, [emb.weights] if final_use else ())
    return out

  def linear(self, x, tape):
    """logits = x E^T  (bf16 [N, V])."""
    if tape is None and x.data.shape[0] <= SKINNY_MAX_ROWS and SKINNY_LOGITS:
      return Act(capi.gemm_skinny(x.data, self.table))
    out = Act(capi.gemm(x.data, self.table))
    if tape is not None:
      emb = self

      def backward():
        dy = out.grad
        assert dy is not None
        g = x.grad_buffer()
        with on_side_stream(dy.device, x.data, dy):
          capi.gemm_wgrad(x.data, dy, emb.weights.grad.view(emb.V, emb.D), accumulate=True)
        capi.gemm(dy, emb.weights.wt16.view(emb.D, emb.V), out=g, accumulate=x.grad_init)
        x.grad_init = True
        out.grad = None

      tape.record(backward)
    return out

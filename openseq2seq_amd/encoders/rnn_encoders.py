"""RNN encoders with embeddings — open_seq2seq/encoders/rnn_encoders.py on the HIP kernels.

BidirectionalRNNEncoderWithEmbedding (:160-317): tf.nn.embedding_lookup ->
tf.nn.bidirectional_dynamic_rnn over TWO independent MultiRNNCell stacks (all forward
layers, all backward layers; each cell = DropoutWrapper(LSTMCell) with input dropout,
parts/rnns/utils.py:17-89) -> concat of the two top outputs. Layer l of both stacks runs
in one kernel launch per time step (rnn_directions_forward); the two top layers write the
halves of one [B, S, 2H] tensor, so the concat is never materialised.
UnidirectionalRNNEncoderWithEmbedding (:20-157) is the forward stack alone."""
from __future__ import absolute_import, division, print_function

import math

import torch

from .encoder import Encoder
from .. import capi
from ..parts.cnns.conv_blocks import Act
from ..parts.rnns.rnn_layers import RNNDirection, rnn_directions_forward
from ..parts.transformer.layers import SeedSeq


def cell_spec(core_cell, core_cell_params):
  """Maps the config's cell class token to (kernel cell name, num_units, forget_bias)."""
  name = core_cell if isinstance(core_cell, str) else getattr(core_cell, "__name__", str(core_cell))
  if "LSTM" not in name:
    raise NotImplementedError("core_cell %s (LSTMCell is built)" % name)
  return "lstm_tf", int(core_cell_params["num_units"]), float(core_cell_params.get("forget_bias", 1.0))


class _SharedTable(object):
  """[V', E] views of a parameter stored as [1, V', E] (the decoder's output projection when
  weight_tied: decoders/rnn_decoders.py:189-194 uses transpose(dense/kernel) as the embedding)."""

  def __init__(self, param, dim):
    self.param, self.dim = param, dim
    self.name = param.name

  @property
  def w16(self):
    return self.param.w16.view(-1, self.dim)

  @property
  def grad(self):
    return self.param.grad.view(-1, self.dim)


class Embedding(object):
  """tf.get_variable [V, E] + tf.nn.embedding_lookup (+ the first DropoutWrapper's input
  dropout fused into the gather)."""

  def __init__(self, store, name, vocab, dim, table=None):
    """table: an existing [*, >=vocab, dim] parameter to share (weight tying) instead of a new one."""
    self.vocab, self.dim = vocab, dim
    if table is not None:
      self.table = _SharedTable(table, dim)
      return

    def init(shape):   # the model-level initializer (glorot_uniform in the NMT configs)
      lim = math.sqrt(6.0 / (vocab + dim))
      return (torch.rand(shape) * 2 - 1) * lim

    self.table = store.add(name, (vocab, dim), init, kind="dense")

  def lookup(self, ids_flat, tape, keep=1.0, seed=0):
    y = capi.embed_fwd(ids_flat, None, self.table.w16, 1.0, keep, seed, plain=True)
    out = Act(y)
    if tape is not None:
      emb = self

      def backward():
        capi.embed_bwd(ids_flat, out.grad, emb.table.grad, 1.0, keep, seed, plain=True)
        out.grad = None

      tape.record(backward, [emb.table.param if isinstance(emb.table, _SharedTable) else emb.table])
    return out


def dropout_act(x, keep, seed, tape):
  """tf.nn.dropout on an Act (DropoutWrapper input dropout of the upper cells)."""
  if keep >= 1.0:
    return x
  y = capi.dropout_bwd(x.data.reshape(-1, x.data.shape[-1]), keep, seed=seed).view_as(x.data)
  out = Act(y, x.lens)
  if tape is not None:
    def backward():
      g = capi.dropout_bwd(out.grad.reshape(-1, out.grad.shape[-1]), keep, seed=seed).view_as(out.grad)
      if x.requires_grad:
        if x.grad_init and x.grad is not None:
          capi.add_bf16(x.grad, g, out=x.grad)
        else:
          x.grad, x.grad_init = g, True
      out.grad = None

    tape.record(backward)
  return out


def apply_scope_initializer(store, first, params):
  """tf.variable_scope(initializer=...) of Encoder/Decoder.encode (encoder.py / decoder.py): the
  configured initializer replaces the default of every variable created without an explicit one
  — kernels, embedding matrices, attention_v (LSTM biases, attention_g / attention_b have their
  own). Only random_uniform_initializer (the GNMT configs) needs handling here: the built-in
  default of this package is already glorot_uniform."""
  tok = params.get('initializer', None)
  name = getattr(tok, "__name__", str(tok))
  if "random_uniform" not in name:
    return
  ip = params.get('initializer_params', {}) or {}
  lo, hi = float(ip.get('minval', 0.0)), float(ip.get('maxval', 1.0))
  for i in range(first, len(store.params)):
    p = store.params[i]
    if p.kind in ("conv", "dense") or p.name.endswith("attention_v"):
      keep_zero_pad = p.name.endswith("/dense/kernel")      # vocabulary padding rows stay zero
      old = store._inits[i]

      def init(shape, lo=lo, hi=hi, old=old, keep=keep_zero_pad):
        w = torch.rand(shape) * (hi - lo) + lo
        if keep:
          ref = old(shape) if callable(old) else old
          w = torch.where(torch.as_tensor(ref) == 0, torch.zeros_like(w), w)
        return w

      store._inits[i] = init


def residual_add(y, x, tape):
  """tf.contrib.rnn.ResidualWrapper: out = y + x (x = the layer's raw, undropped input)."""
  out = Act(capi.add_bf16(y.data, x.data), y.lens)
  if tape is not None:
    def backward():
      g = out.grad
      for t in (y, x):
        if t.requires_grad:
          if t.grad_init and t.grad is not None:
            capi.add_bf16(t.grad, g, out=t.grad)
          else:
            t.grad, t.grad_init = g.clone(), True
      out.grad = None

    tape.record(backward)
  return out


class BidirectionalRNNEncoderWithEmbedding(Encoder):
  @staticmethod
  def get_required_params():
    return dict(Encoder.get_required_params(), **{
        'src_vocab_size': int, 'src_emb_size': int, 'encoder_layers': int,
        'encoder_use_skip_connections': bool, 'core_cell': None, 'core_cell_params': dict,
    })

  @staticmethod
  def get_optional_params():
    return dict(Encoder.get_optional_params(), **{
        'encoder_dp_input_keep_prob': float, 'encoder_dp_output_keep_prob': float,
        'time_major': bool, 'use_swap_memory': bool, 'proj_size': int, 'num_groups': int,
    })

  _bidirectional = True

  def __init__(self, params, model, name="bidir_rnn_encoder_with_emb", mode='train'):
    super(BidirectionalRNNEncoderWithEmbedding, self).__init__(params, model, name, mode)
    self._src_vocab_size = self.params['src_vocab_size']
    self._src_emb_size = self.params['src_emb_size']
    if self.params['encoder_use_skip_connections']:
      raise NotImplementedError("encoder_use_skip_connections (ResidualWrapper)")
    if self.params.get('encoder_dp_output_keep_prob', 1.0) != 1.0:
      raise NotImplementedError("encoder_dp_output_keep_prob != 1.0")
    if self.params.get('time_major', False):
      raise NotImplementedError("time_major layouts (batch-major [B,S,...] only)")

  def build(self, store):
    p = self.params
    first = len(store.params)
    cell, H, fb = cell_spec(p['core_cell'], p['core_cell_params'])
    self.H = H
    self.output_dim = H * (2 if self._bidirectional else 1)
    scope = "ForwardPass/" + self._name
    self.embedding = Embedding(store, scope + "/EncoderEmbeddingMatrix", self._src_vocab_size,
                               self._src_emb_size)
    # variables are created in execution order (layer by layer, both directions of a layer run in
    # one kernel sequence): the overlapped gradient reducer needs them final in reverse creation order
    tags = ("FW", "BW")[:2 if self._bidirectional else 1]
    self.stacks = [[] for _ in tags]   # [direction][layer]
    cin = self._src_emb_size
    for l in range(p['encoder_layers']):
      for d, tag in enumerate(tags):
        self.stacks[d].append(RNNDirection(
            store, "%s/%s/multi_rnn_cell/cell_%d/lstm_cell" % (scope, tag, l), cell, [cin], H,
            reverse=(d == 1), forget_bias=fb))
      cin = H
    apply_scope_initializer(store, first, p)
    return self

  def _encode(self, input_dict):
    ids, lens = input_dict['source_tensors'][0], input_dict['source_tensors'][1]
    B, S = ids.shape
    training = self._mode == "train"
    tape = input_dict.get('tape') if training else None
    seeds = input_dict.get('seeds') or SeedSeq(17)
    keep = self.params.get('encoder_dp_input_keep_prob', 1.0) if training else 1.0
    ndir = len(self.stacks)
    H = self.H
    emb = self.embedding.lookup(ids.reshape(-1).contiguous(), tape)
    emb = Act(emb.data.view(B, S, -1), lens) if tape is None else self._view3(emb, B, S, lens, tape)
    cur = [emb] * ndir
    out = Act(torch.zeros((B, S, H * ndir), dtype=torch.bfloat16, device=ids.device), lens)
    nl = len(self.stacks[0])
    for l in range(nl):
      xs = [[dropout_act(cur[d], keep, seeds.next(), tape)] for d in range(ndir)]
      dirs = [self.stacks[d][l] for d in range(ndir)]
      if l == nl - 1:
        ys = rnn_directions_forward(
            dirs, xs, lens, tape, [out.data[:, :, d * H:(d + 1) * H] for d in range(ndir)],
            [(lambda o=out, d=d: o.grad[:, :, d * H:(d + 1) * H]) for d in range(ndir)])
      else:
        ys = rnn_directions_forward(dirs, xs, lens, tape)
      cur = ys
    return {'outputs': out.data, 'outputs_act': out, 'state': None, 'src_lengths': lens,
            'encoder_input': ids, 'seeds': seeds}

  @staticmethod
  def _view3(emb, B, S, lens, tape):
    """[B*S, E] Act -> [B, S, E] Act sharing storage and gradient."""
    v = Act(emb.data.view(B, S, -1), lens)

    def backward():
      g = v.grad.reshape(B * S, -1)
      if emb.grad_init and emb.grad is not None:
        capi.add_bf16(emb.grad, g, out=emb.grad)
      else:
        emb.grad, emb.grad_init = g, True
      v.grad = None

    tape.record(backward)
    return v

  @property
  def src_vocab_size(self):
    return self._src_vocab_size

  @property
  def src_emb_size(self):
    return self._src_emb_size


class UnidirectionalRNNEncoderWithEmbedding(BidirectionalRNNEncoderWithEmbedding):
  _bidirectional = False

  def __init__(self, params, model, name="unidir_rnn_encoder_with_emb", mode='train'):
    super(UnidirectionalRNNEncoderWithEmbedding, self).__init__(params, model, name, mode)


class GNMTLikeEncoderWithEmbedding(BidirectionalRNNEncoderWithEmbedding):
  """encoders/rnn_encoders.py:320-470: embedding -> ONE bidirectional LSTM layer (no dropout)
  -> encoder_layers - 1 unidirectional layers (DropoutWrapper input dropout), every
  unidirectional layer but the first wrapped in a ResidualWrapper. Output depth = num_units."""

  def __init__(self, params, model, name="gnmt_encoder_with_emb", mode='train'):
    Encoder.__init__(self, params, model, name, mode)
    self._src_vocab_size = self.params['src_vocab_size']
    self._src_emb_size = self.params['src_emb_size']
    if self.params['encoder_layers'] < 2:
      raise ValueError("GNMT encoder must have at least 2 layers")
    if self.params.get('encoder_dp_output_keep_prob', 1.0) != 1.0:
      raise NotImplementedError("encoder_dp_output_keep_prob != 1.0")
    if self.params.get('time_major', False):
      raise NotImplementedError("time_major layouts (batch-major [B,S,...] only)")

  def build(self, store):
    p = self.params
    first = len(store.params)
    cell, H, fb = cell_spec(p['core_cell'], p['core_cell_params'])
    self.H = H
    self.output_dim = H
    scope = "ForwardPass/" + self._name
    self.embedding = Embedding(store, scope + "/EncoderEmbeddingMatrix", self._src_vocab_size,
                               self._src_emb_size)
    self.l1 = [RNNDirection(store, "%s/Level1%s/lstm_cell" % (scope, tag), cell, [self._src_emb_size], H,
                            reverse=(d == 1), forget_bias=fb) for d, tag in enumerate(("FW", "BW"))]
    self.uni = []
    cin = 2 * H
    for l in range(p['encoder_layers'] - 1):
      self.uni.append(RNNDirection(store, "%s/UniDirLevel/multi_rnn_cell/cell_%d/lstm_cell" % (scope, l),
                                   cell, [cin], H, reverse=False, forget_bias=fb))
      cin = H
    apply_scope_initializer(store, first, p)
    return self

  def _encode(self, input_dict):
    ids, lens = input_dict['source_tensors'][0], input_dict['source_tensors'][1]
    B, S = ids.shape
    training = self._mode == "train"
    tape = input_dict.get('tape') if training else None
    seeds = input_dict.get('seeds') or SeedSeq(19)
    keep = self.params.get('encoder_dp_input_keep_prob', 1.0) if training else 1.0
    H = self.H
    emb = self.embedding.lookup(ids.reshape(-1).contiguous(), tape)
    emb = Act(emb.data.view(B, S, -1), lens) if tape is None else self._view3(emb, B, S, lens, tape)
    l1 = Act(torch.zeros((B, S, 2 * H), dtype=torch.bfloat16, device=ids.device), lens)
    rnn_directions_forward(self.l1, [[emb], [emb]], lens, tape,
                           [l1.data[:, :, d * H:(d + 1) * H] for d in range(2)],
                           [(lambda o=l1, d=d: o.grad[:, :, d * H:(d + 1) * H]) for d in range(2)])
    cur = l1
    for l, layer in enumerate(self.uni):
      y = rnn_directions_forward([layer], [[dropout_act(cur, keep, seeds.next(), tape)]], lens, tape)[0]
      cur = residual_add(y, cur, tape) if l > 0 else y
    return {'outputs': cur.data, 'outputs_act': cur, 'state': None, 'src_lengths': lens,
            'encoder_input': ids, 'seeds': seeds}

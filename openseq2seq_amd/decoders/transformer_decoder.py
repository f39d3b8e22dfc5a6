"""TransformerDecoder — training/eval pass (decode_pass) of
open_seq2seq/decoders/transformer_decoder.py:20-230 on the HIP kernels (packed token
layout): shifted shared embedding + position signal + dropout, N x [causal self-attention,
encoder-decoder attention, FFN] in pre-norm residual form, final LayerNorm, tied softmax
projection; and beam-search prediction (predict, :232-326): an incremental decoder step on one
new position per beam row (append-only K/V caches + an ancestry table, encoder K/V projected
once) driven by parts/transformer/beam_search.sequence_beam_search."""
from __future__ import absolute_import, division, print_function

import torch

from .decoder import Decoder
from .. import capi
from ..parts.cnns.conv_blocks import Act
from ..parts.transformer import beam_search
from ..parts.transformer import layers as L
from ..parts.transformer import packing


class TransformerDecoder(Decoder):
  @staticmethod
  def get_required_params():
    return dict(Decoder.get_required_params(), **{
        'EOS_ID': int, 'layer_postprocess_dropout': float, 'num_hidden_layers': int,
        'hidden_size': int, 'num_heads': int, 'attention_dropout': float, 'relu_dropout': float,
        'filter_size': int, 'batch_size': int, 'tgt_vocab_size': int, 'beam_size': int,
        'alpha': float, 'extra_decode_length': int,
    })

  @staticmethod
  def get_optional_params():
    return dict(Decoder.get_optional_params(), **{
        'regularizer': None, 'regularizer_params': dict, 'initializer': None,
        'initializer_params': dict, 'GO_SYMBOL': int, 'PAD_SYMBOL': int, 'END_SYMBOL': int,
        'norm_params': dict,
    })

  def __init__(self, params, model, name="transformer_decoder", mode='train'):
    super(TransformerDecoder, self).__init__(params, model, name, mode)
    self.params['shared_embed'] = True
    self.layers = []

  def build(self, store):
    p = self.params
    D = p["hidden_size"]
    scope = "ForwardPass/" + self._name
    # the encoder-decoder key / value kernels of all layers, created next to each other (L.FusedCrossKV)
    kvs = [L.Dense(store, "%s/layer_%d/encdec_attention/attention/kv" % (scope, n), D, 2 * D, False)
           for n in range(p["num_hidden_layers"])]
    self.cross_kv = L.FusedCrossKV(store, kvs)
    for n in range(p["num_hidden_layers"]):
      ls = "%s/layer_%d" % (scope, n)
      self.layers.append(dict(
          ln1=L.LayerNorm(store, ls + "/self_attention/layer_normalization", D),
          self_att=L.MultiHeadAttention(store, ls + "/self_attention/self_attention", D,
                                        p["num_heads"], True),
          ln2=L.LayerNorm(store, ls + "/encdec_attention/layer_normalization", D),
          cross=L.MultiHeadAttention(store, ls + "/encdec_attention/attention", D,
                                     p["num_heads"], False, kv=kvs[n]),
          ln3=L.LayerNorm(store, ls + "/ffn/layer_normalization", D),
          ffn=L.FeedForward(store, ls + "/ffn/feed_foward_network", D, p["filter_size"])))
    self.output_normalization = L.LayerNorm(store, scope + "/layer_normalization", D)
    return self

  def _decode(self, input_dict):
    if 'target_tensors' not in input_dict:
      return self.predict(input_dict)
    enc = input_dict['encoder_output']
    emb = enc['embedding_softmax_layer']
    training = (self.mode == "train")
    tape = input_dict.get('tape', None) if training else None
    seeds = enc.get('seeds') or L.SeedSeq(1)
    ps = enc['packed_source']
    pt = input_dict.get('packed_target')
    if pt is None:
      ids, lens = input_dict['target_tensors']
      pt = packing.to_device(packing.pack_ids(ids.cpu().numpy(), lens.cpu().numpy(),
                                              shift_right=True), ids.device)
    p = self.params
    post_keep = 1.0 - p["layer_postprocess_dropout"] if training else 1.0
    att_keep = 1.0 - p["attention_dropout"] if training else 1.0
    relu_keep = 1.0 - p["relu_dropout"] if training else 1.0
    enc_out = enc['outputs_act']
    x = emb.embed(pt["ids"], pt["pos"], tape, post_keep, seeds.next())
    max_cross = max(pt["max_len"], ps["max_len"])
    kv_all = self.cross_kv.forward(enc_out, tape) if self.cross_kv.usable() else None
    for li, lyr in enumerate(self.layers):
      y = lyr["ln1"].forward(x, tape)
      x = lyr["self_att"].forward(y, y, pt["cu"], pt["cu"], pt["max_len"], True, tape, seeds,
                                  att_keep, post_keep, residual=x)
      y = lyr["ln2"].forward(x, tape)
      if li == 0 and kv_all is not None:
        self.cross_kv.join()
      x = lyr["cross"].forward(y, enc_out, pt["cu"], ps["cu"], max_cross, False, tape, seeds,
                               att_keep, post_keep, residual=x,
                               kv_pre=(kv_all, li) if kv_all is not None else None)
      y = lyr["ln3"].forward(x, tape)
      x = lyr["ffn"].forward(y, tape, seeds, relu_keep, post_keep, residual=x)
    out = self.output_normalization.forward(x, tape)
    logits = emb.linear(out, tape)
    return {"logits": logits.data, "logits_act": logits, "packed_target": pt,
            "outputs": None, "final_state": None, "final_sequence_lengths": None}

  # ---- inference: beam search (transformer_decoder.py:232-326) --------------------------------
  def _get_symbols_to_logits_fn(self, enc, beam, N, Tmax):
    """The incremental decoder step. State outside the beam-search cache: per-layer K/V
    caches [N, Tmax, D] (append-only) and the encoder K/V projections; inside it: the
    int32 ancestry table [N, Tmax] (which cache row holds position j of this beam)."""
    p = self.params
    D, H = p["hidden_size"], p["num_heads"]
    emb = enc['embedding_softmax_layer']
    ps = enc['packed_source']
    enc_out = enc['outputs_act']
    dev = enc_out.data.device
    scale = (D // H) ** -0.5
    kc = [torch.empty((N, Tmax, D), dtype=torch.bfloat16, device=dev) for _ in self.layers]
    vc = [torch.empty((N, Tmax, D), dtype=torch.bfloat16, device=dev) for _ in self.layers]
    enc_kv = [lyr["cross"].kv.forward(enc_out, None).data for lyr in self.layers]
    pos = torch.arange(Tmax, dtype=torch.int32, device=dev)[:, None].expand(Tmax, N).contiguous()

    def step(last, positions, i, status, cache):
      anc = cache["ancestry"]
      x = emb.embed(last, positions, None, 1.0, 0)   # id 0 (the initial id) embeds to zeros
      for l, lyr in enumerate(self.layers):
        y = lyr["ln1"].forward(x, None)
        qkv = lyr["self_att"].qkv.forward(y, None).data
        o = capi.decode_self_attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], kc[l], vc[l], anc,
                                       H, i, scale, status=status)
        x = lyr["self_att"].out.forward(Act(o), None, residual=x)
        y = lyr["ln2"].forward(x, None)
        q = lyr["cross"].q.forward(y, None).data
        o = capi.decode_cross_attention(q, enc_kv[l][:, :D], enc_kv[l][:, D:], ps["cu"], beam, H,
                                        ps["max_len"], scale)
        x = lyr["cross"].out.forward(Act(o), None, residual=x)
        y = lyr["ln3"].forward(x, None)
        x = lyr["ffn"].forward(y, None, None, 1.0, 1.0, residual=x)
      out = self.output_normalization.forward(x, None)
      return emb.linear(out, None).data, cache

    def symbols_to_logits_fn(ids, i, cache):
      return step(ids[:, -1].contiguous(), pos[i], i, None, cache)

    def device_step_fn(state, cache):
      """Same step, loop index / last ids / positions read from the beam state on the device."""
      return step(state.last_ids, state.pos, 0, state.status, cache)

    symbols_to_logits_fn.device_step_fn = device_step_fn
    return symbols_to_logits_fn

  def predict(self, input_dict):
    enc = input_dict['encoder_output']
    p = self.params
    src = enc['encoder_input']
    B, input_length = int(src.shape[0]), int(src.shape[1])
    max_decode_length = input_length + p["extra_decode_length"]
    beam = p["beam_size"]
    dev = src.device
    fn = self._get_symbols_to_logits_fn(enc, beam, B * beam, max_decode_length)
    initial_ids = torch.zeros(B, dtype=torch.int32, device=dev)
    cache = {"ancestry": torch.zeros((B, max_decode_length), dtype=torch.int32, device=dev)}
    emb = enc['embedding_softmax_layer']
    decoded_ids, scores = beam_search.sequence_beam_search(
        fn, initial_ids, cache, emb.V, beam, p["alpha"], max_decode_length, p["EOS_ID"],
        device_step_fn=fn.device_step_fn if input_dict.get('use_graph', True) else None)
    top_decoded_ids = decoded_ids[:, 0, 1:].contiguous()
    # the reference re-runs decode_pass on the decoded ids only to fill "logits", which no
    # consumer of the infer/eval modes reads (models/text2text.py:84-225); ask for it explicitly
    logits = None
    if input_dict.get('return_logits', False):
      lens = self.sequence_lengths(top_decoded_ids)
      logits = self._decode(dict(input_dict, target_tensors=[top_decoded_ids, lens]))["logits"]
    return {"logits": logits, "outputs": [top_decoded_ids], "scores": scores,
            "final_state": None, "final_sequence_lengths": None}

  def sequence_lengths(self, ids):
    """Tokens up to and including the first EOS (whole row if none)."""
    eos = (ids == self.params["EOS_ID"])
    T = ids.shape[1]
    first = torch.where(eos.any(1), eos.float().argmax(1) + 1, torch.full((ids.shape[0],), T, device=ids.device))
    return first.to(torch.int32)

"""TransformerDecoder — training/eval pass (decode_pass) of
open_seq2seq/decoders/transformer_decoder.py:20-230 on the HIP kernels (packed token
layout): shifted shared embedding + position signal + dropout, N x [causal self-attention,
encoder-decoder attention, FFN] in pre-norm residual form, final LayerNorm, tied softmax
projection. Beam-search prediction (:232-326) is the next row of SURVEY §8f."""
from __future__ import absolute_import, division, print_function

from .decoder import Decoder
from .. import capi
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
    for n in range(p["num_hidden_layers"]):
      ls = "%s/layer_%d" % (scope, n)
      self.layers.append(dict(
          ln1=L.LayerNorm(store, ls + "/self_attention/layer_normalization", D),
          self_att=L.MultiHeadAttention(store, ls + "/self_attention/self_attention", D,
                                        p["num_heads"], True),
          ln2=L.LayerNorm(store, ls + "/encdec_attention/layer_normalization", D),
          cross=L.MultiHeadAttention(store, ls + "/encdec_attention/attention", D,
                                     p["num_heads"], False),
          ln3=L.LayerNorm(store, ls + "/ffn/layer_normalization", D),
          ffn=L.FeedForward(store, ls + "/ffn/feed_foward_network", D, p["filter_size"])))
    self.output_normalization = L.LayerNorm(store, scope + "/layer_normalization", D)
    return self

  def _decode(self, input_dict):
    if 'target_tensors' not in input_dict:
      raise NotImplementedError("beam-search predict() is not built yet (SURVEY §8f rank 1)")
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
    for lyr in self.layers:
      y = lyr["ln1"].forward(x, tape)
      x = lyr["self_att"].forward(y, y, pt["cu"], pt["cu"], pt["max_len"], True, tape, seeds,
                                  att_keep, post_keep, residual=x)
      y = lyr["ln2"].forward(x, tape)
      x = lyr["cross"].forward(y, enc_out, pt["cu"], ps["cu"], max_cross, False, tape, seeds,
                               att_keep, post_keep, residual=x)
      y = lyr["ln3"].forward(x, tape)
      x = lyr["ffn"].forward(y, tape, seeds, relu_keep, post_keep, residual=x)
    out = self.output_normalization.forward(x, tape)
    logits = emb.linear(out, tape)
    return {"logits": logits.data, "logits_act": logits, "packed_target": pt,
            "outputs": None, "final_state": None, "final_sequence_lengths": None}

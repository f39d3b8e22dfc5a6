"""TransformerEncoder — open_seq2seq/encoders/transformer_encoder.py:20-170 on the HIP
kernels (packed token layout). Same param schema; pre-norm residual layers
(parts/transformer/common.py:99-106), shared embedding + sinusoid position signal +
dropout, final LayerNorm."""
from __future__ import absolute_import, division, print_function

from .encoder import Encoder
from ..parts.transformer import layers as L
from ..parts.transformer import packing


class TransformerEncoder(Encoder):
  @staticmethod
  def get_required_params():
    return dict(Encoder.get_required_params(), **{
        "encoder_layers": int, "hidden_size": int, "num_heads": int,
        "attention_dropout": float, "filter_size": int, "src_vocab_size": int,
        "relu_dropout": float, "layer_postprocess_dropout": float, "remove_padding": bool,
    })

  @staticmethod
  def get_optional_params():
    return dict(Encoder.get_optional_params(), **{
        'regularizer': None, 'regularizer_params': dict, 'initializer': None,
        'initializer_params': dict, 'pad_embeddings_2_eight': bool, 'norm_params': dict,
    })

  def __init__(self, params, model, name="transformer_encoder", mode='train'):
    super(TransformerEncoder, self).__init__(params, model, name=name, mode=mode)
    if self.params.get("norm_params", {"type": "layernorm_L2"}).get("type") != "layernorm_L2":
      raise NotImplementedError("only layernorm_L2 has HIP kernels")
    self.layers = []
    self.embedding_softmax_layer = None

  def build(self, store):
    p = self.params
    D = p["hidden_size"]
    scope = "ForwardPass/" + self._name
    # tf.layers.Layer scoping: EmbeddingSharedWeights is first CALLED inside the encoder's variable scope, so its
    # variable is '<encoder scope>/embedding_shared_weights/embedding_and_softmax/weights' (embedding_layer.py:49-54;
    # the name the reference's own code produces when it is executed: tests/golden/ref_exec_transformer.npz)
    self.embedding_softmax_layer = L.SharedEmbedding(
        store, scope + "/embedding_shared_weights", p["src_vocab_size"], D,
        pad_vocab_to_eight=p.get('pad_embeddings_2_eight', False))
    for n in range(p['encoder_layers']):
      ls = "%s/layer_%d" % (scope, n)
      self.layers.append(dict(
          ln1=L.LayerNorm(store, ls + "/self_attention/layer_normalization", D),
          att=L.MultiHeadAttention(store, ls + "/self_attention/self_attention", D,
                                   p["num_heads"], True),
          ln2=L.LayerNorm(store, ls + "/ffn/layer_normalization", D),
          ffn=L.FeedForward(store, ls + "/ffn/feed_foward_network", D, p["filter_size"])))
    self.output_normalization = L.LayerNorm(store, scope + "/layer_normalization", D)
    return self

  def _encode(self, input_dict):
    """source_tensors = [ids [B,L] int32 (pad 0), lengths [B]]; optional 'packed_source'
    (parts/transformer/packing.pack_ids on the device). Returns the reference's dict
    (outputs are PACKED [N_src, D]; 'packed_source' carries cu/max_len)."""
    training = (self.mode == "train")
    tape = input_dict.get('tape', None) if training else None
    seeds = input_dict.get('seeds') or L.SeedSeq(input_dict.get('seed', 0))
    pk = input_dict.get('packed_source')
    if pk is None:
      ids, lens = input_dict['source_tensors']
      pk = packing.to_device(packing.pack_ids(ids.cpu().numpy(), lens.cpu().numpy()), ids.device)
    p = self.params
    post_keep = 1.0 - p["layer_postprocess_dropout"] if training else 1.0
    att_keep = 1.0 - p["attention_dropout"] if training else 1.0
    relu_keep = 1.0 - p["relu_dropout"] if training else 1.0
    x = self.embedding_softmax_layer.embed(pk["ids"], pk["pos"], tape, post_keep, seeds.next(),
                                           final_use=True)
    for lyr in self.layers:
      y = lyr["ln1"].forward(x, tape)
      x = lyr["att"].forward(y, y, pk["cu"], pk["cu"], pk["max_len"], False, tape, seeds,
                             att_keep, post_keep, residual=x)
      y = lyr["ln2"].forward(x, tape)
      x = lyr["ffn"].forward(y, tape, seeds, relu_keep, post_keep, residual=x)
    out = self.output_normalization.forward(x, tape)
    return {'outputs': out.data, 'outputs_act': out, 'packed_source': pk,
            'inputs_attention_bias': None, 'state': None,
            'src_lengths': input_dict['source_tensors'][1],
            'embedding_softmax_layer': self.embedding_softmax_layer,
            'encoder_input': input_dict['source_tensors'][0], 'seeds': seeds}

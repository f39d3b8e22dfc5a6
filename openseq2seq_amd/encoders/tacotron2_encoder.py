"""Tacotron2Encoder — open_seq2seq/encoders/tacotron2_encoder.py:104-339 on the HIP kernels:
character embedding -> conv_bn_actv (k=5, BN momentum 0.1 / eps 1e-5) + dropout per layer
-> bidirectional cuDNN-form LSTM (no sequence lengths in training, exactly as the cuDNN
path of the reference, :248-263) -> [+ rnn dropout] -> [concat of the global-style-token
embedding tiled over time, :155-172, 332-333]. The style encoder (_embed_style, :341-505)
lives in parts/tacotron/gst.py."""
from __future__ import absolute_import, division, print_function

import torch

from .encoder import Encoder
from .rnn_encoders import Embedding
from .. import capi
from ..parts.cnns.conv_blocks import Act, ConvBN, conv_bn_actv, reshape_act, xavier_normal_conv
from ..parts.rnns.rnn_layers import RNNDirection, rnn_directions_forward
from ..parts.transformer.layers import SeedSeq


class Tacotron2Encoder(Encoder):
  @staticmethod
  def get_required_params():
    return dict(Encoder.get_required_params(), **{
        'cnn_dropout_prob': float, 'rnn_dropout_prob': float, 'src_emb_size': int,
        'conv_layers': list, 'activation_fn': None, 'num_rnn_layers': int, 'rnn_cell_dim': int,
        'use_cudnn_rnn': bool, 'rnn_type': None, 'rnn_unidirectional': bool,
    })

  @staticmethod
  def get_optional_params():
    return dict(Encoder.get_optional_params(), **{
        'data_format': ['channels_first', 'channels_last'], 'bn_momentum': float,
        'bn_epsilon': float, 'zoneout_prob': float, 'style_embedding_enable': bool,
        'style_embedding_params': dict,
    })

  def __init__(self, params, model, name="tacotron2_encoder", mode='train'):
    super(Tacotron2Encoder, self).__init__(params, model, name, mode)
    p = self.params
    if p.get('zoneout_prob', 0.) != 0.:
      raise NotImplementedError("zoneout")
    if p['rnn_unidirectional']:
      raise NotImplementedError("unidirectional encoder RNN")
    rt = p['rnn_type'] if isinstance(p['rnn_type'], str) else getattr(p['rnn_type'], "__name__", "")
    if "LSTM" not in rt.upper():
      raise NotImplementedError("rnn_type %s" % rt)
    self._cudnn_cell = bool(p['use_cudnn_rnn'])

  def build(self, store, src_vocab_size=None, num_style_features=None):
    p = self.params
    if src_vocab_size is None:
      src_vocab_size = self._model.get_data_layer().params['src_vocab_size']
    scope = "ForwardPass/" + self._name
    l2 = 0.0
    if p.get('regularizer', None) is not None:
      l2 = float(p.get('regularizer_params', {}).get('scale', 0.0))
    E = p['src_emb_size']
    self.embedding = Embedding(store, scope + "/EncoderEmbeddingMatrix", src_vocab_size, E)
    self.embedding.table.l2 = l2
    mom, eps = p.get('bn_momentum', 0.1), p.get('bn_epsilon', 1e-5)
    self.convs = []
    cin = E
    for i, cl in enumerate(p['conv_layers']):
      n = "%s/conv%d" % (scope, i + 1)
      self.convs.append(ConvBN(store, n, n + "/bn", cin, cl['num_channels'], cl['kernel_size'][0],
                               stride=cl['stride'][0], padding=cl['padding'], bn_momentum=mom,
                               bn_epsilon=eps, l2=l2, initializer=xavier_normal_conv))
      cin = cl['num_channels']
    H = p['rnn_cell_dim']
    self.H = H
    self.rnn = []
    if p['num_rnn_layers'] > 0:
      cell = "lstm_cudnn" if self._cudnn_cell else "lstm_tf"
      rin = cin
      for l in range(p['num_rnn_layers']):
        self.rnn.append([RNNDirection(store, "%s/cudnn_rnn/layer_%d/%s" % (scope, l, tag), cell,
                                      [rin], H, reverse=(d == 1)) for d, tag in enumerate(("fw", "bw"))])
        for d in self.rnn[-1]:
          for w in d.wx + [d.wh]:
            w.l2 = l2
        rin = 2 * H
      cin = 2 * H
    self.text_dim = cin
    self.style = None
    if p.get('style_embedding_enable', False):
      if 'style_embedding_params' not in p:
        raise ValueError("style_embedding_params must be passed if style embedding is enabled")
      from ..parts.tacotron.gst import StyleEncoder
      if num_style_features is None:
        naf = self._model.get_data_layer().params['num_audio_features']
        num_style_features = naf['mel'] if isinstance(naf, dict) else naf
      self.style = StyleEncoder(store, scope + "/style_encoder", p['style_embedding_params'],
                                num_style_features, p['activation_fn'], mom, eps, l2)
      cin += self.style.output_dim
    self.output_dim = cin
    return self

  def _encode(self, input_dict):
    src = input_dict['source_tensors']
    text, text_len = src[0], src[1]
    B, S = text.shape
    p = self.params
    training = self._mode == "train"
    tape = input_dict.get('tape') if training else None
    seeds = input_dict.get('seeds') or SeedSeq(31)
    emb = self.embedding.lookup(text.reshape(-1).contiguous(), tape)
    x = reshape_act(emb, (B, S, -1), tape)
    keep_cnn = 1.0 - p['cnn_dropout_prob'] if training else 1.0
    for cl, layer in zip(p['conv_layers'], self.convs):
      s = cl['stride'][0]
      if cl['padding'] == "VALID":
        text_len = torch.div(text_len - cl['kernel_size'][0] + s, s, rounding_mode='floor')
      else:
        text_len = torch.div(text_len + s - 1, s, rounding_mode='floor')
      x = conv_bn_actv(layer, x, None, p['activation_fn'], training, tape, keep_prob=keep_cnn,
                       seed=seeds.next(), mask_output=False)
    S2 = x.data.shape[1]
    out = Act(torch.empty((B, S2, self.output_dim), dtype=torch.bfloat16, device=text.device), None)
    H = self.H
    # inference uses CudnnCompatibleLSTMCell under stack_bidirectional_dynamic_rnn WITH
    # sequence lengths (:225-236); training runs the cuDNN kernel over the padded batch
    rnn_lens = None if (training and self._cudnn_cell) else text_len
    for l, dirs in enumerate(self.rnn):
      last = l == len(self.rnn) - 1
      if last:
        views = [out.data[:, :, d * H:(d + 1) * H] for d in range(2)]
        fns = [(lambda o=out, d=d: o.grad[:, :, d * H:(d + 1) * H]) for d in range(2)]
        rnn_directions_forward(dirs, [x], rnn_lens, tape, views, fns)
      else:
        ybuf = Act(torch.empty((B, S2, 2 * H), dtype=torch.bfloat16, device=text.device), None)
        rnn_directions_forward(
            dirs, [x], rnn_lens, tape, [ybuf.data[:, :, d * H:(d + 1) * H] for d in range(2)],
            [(lambda o=ybuf, d=d: o.grad[:, :, d * H:(d + 1) * H]) for d in range(2)])
        x = ybuf
    if not self.rnn:
      out.data[:, :, :self.text_dim].copy_(x.data)
      if tape is not None:
        def conv_out_bwd(x=x):
          x.grad, x.grad_init = out.grad[:, :, :self.text_dim].contiguous(), True
        tape.record(conv_out_bwd)
    keep_rnn = 1.0 - p['rnn_dropout_prob'] if training else 1.0
    if keep_rnn < 1.0:
      raise NotImplementedError("rnn_dropout_prob > 0")
    if self.style is not None:
      style_spec, style_len = src[2], src[3]
      emb_style = self.style.forward(style_spec, style_len, training, tape)      # Act [B, Es]
      td = self.text_dim
      out.data[:, :, td:] = emb_style.data[:, None, :]
      if tape is not None:
        def style_bwd():
          g32 = torch.empty((B, self.style.output_dim), dtype=torch.float32, device=text.device)
          capi.sum_time(out.grad[:, :, td:], g32)
          g = torch.empty_like(emb_style.data)
          capi.cast_f32_to_bf16(g32.view(-1), g.view(-1))
          emb_style.grad, emb_style.grad_init = g, True
        tape.record(style_bwd)
    return {'outputs': out.data, 'outputs_act': out, 'src_length': text_len, 'seeds': seeds}

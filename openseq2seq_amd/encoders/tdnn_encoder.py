"""TDNNEncoder — Jasper / QuartzNet / Wave2Letter+ encoder
(open_seq2seq/encoders/tdnn_encoder.py:10-265) on the HIP conv/BN kernels.

Same config schema, same bookkeeping as the reference's `_encode` (:87-265):
mask before every conv input, lengths shrink as ceil(L/stride) for SAME and
(L-K)//s+1 for VALID, dense residual aggregation (block k sees k inputs),
dropout on every layer output, NO mask on the final output.
"""
from __future__ import absolute_import, division, print_function

import torch

from .encoder import Encoder
from .. import capi
import os

from ..parts.cnns.conv_blocks import (Act, ConvBN, ConvOnly, ConvSampleNorm, SepConvBN, conv_actv, conv_bn_res_bn_actv,
                                      xavier_normal_conv, glorot_uniform_conv, launch_residual_early,
                                      launch_dense_residual, conv_bn_dres_actv)
from ..parts.cnns.dense_residual import DenseResidualPlan

# which layer of a residual block starts the block end's residual branches on the side stream
# (-1 = never: they run in front of the block's last BatchNorm, as in round 2). 2 leaves the first
# two convolutions of every block alone on the GPU (bench.py times those)
RES_EARLY_REP = int(os.environ.get("OS2S_RES_EARLY", "-1"))


class TDNNEncoder(Encoder):
  """General time delay neural network (TDNN) encoder. Fully convolutional model."""

  @staticmethod
  def get_required_params():
    return dict(Encoder.get_required_params(), **{
        'dropout_keep_prob': float,
        'convnet_layers': list,
        'activation_fn': None,  # any valid callable
    })

  @staticmethod
  def get_optional_params():
    return dict(Encoder.get_optional_params(), **{
        'data_format': ['channels_first', 'channels_last'],
        'normalization': [None, 'batch_norm', 'layer_norm', 'instance_norm'],
        'bn_momentum': float,
        'bn_epsilon': float,
        'use_conv_mask': bool,
        'drop_block_prob': float,
        'drop_block_index': int,
    })

  def __init__(self, params, model, name="w2l_encoder", mode='train'):
    super(TDNNEncoder, self).__init__(params, model, name, mode)
    # normalization None = conv_actv (conv_blocks.py:17-58): convolution + activation, BatchNorm only at the
    # residual block ends, which go through conv_bn_res_bn_actv whatever the setting (tdnn_encoder.py:216-233).
    # 'layer_norm' / 'instance_norm' = conv_ln_actv / conv_in_actv (conv_blocks.py:234-309) on
    # csrc/sample_norm.hip. With those two the reference passes NO bn_momentum / bn_epsilon to a residual block
    # end (normalization_params stays empty, tdnn_encoder.py:144-156) and conv_bn_res_bn_actv fails on its
    # missing arguments: residual blocks are refused here as well
    if self.params.get('normalization', 'batch_norm') not in ('batch_norm', None, 'layer_norm', 'instance_norm'):
      raise ValueError("Incorrect normalization")
    if self.params.get('data_format', 'channels_last') != 'channels_last':
      raise NotImplementedError("HIP path is channels_last (the reference's default)")
    self._layers = None

  # ---- variable creation (graph-construction phase of the reference) -----
  def build(self, store, num_features):
    p = self.params
    init_tok = p.get('initializer', None)
    init_name = getattr(init_tok, "__name__", str(init_tok)).lower()
    uniform = p.get('initializer_params', {}).get('uniform', True)
    initializer = glorot_uniform_conv if ("xavier" in init_name and uniform) or \
        "glorot_uniform" in init_name else xavier_normal_conv
    l2 = 0.0
    if p.get('regularizer', None) is not None:
      l2 = float(p.get('regularizer_params', {}).get('scale', 0.0))
    mom, eps = p.get('bn_momentum', 0.90), p.get('bn_epsilon', 1e-3)
    scope = "ForwardPass/" + self._name
    cin = num_features
    res_channels = []   # channels of the dense-residual inputs accumulated so far
    layers = []
    for ib, blk in enumerate(p['convnet_layers']):
      if blk['type'] not in ('conv1d', 'sep_conv1d'):
        raise NotImplementedError("layer type %s has no HIP kernel yet" % blk['type'])
      Layer = SepConvBN if blk['type'] == 'sep_conv1d' else ConvBN
      residual = blk.get('residual', False)
      dense = blk.get('residual_dense', False)
      if residual:
        if dense:
          res_channels.append(cin)
          res_in = list(res_channels)
        else:
          res_in = [cin]
      for ir in range(blk['repeat']):
        lname = "%s/conv%d%d" % (scope, ib + 1, ir + 1)
        block_end = residual and ir == blk['repeat'] - 1
        norm = p.get('normalization', 'batch_norm')
        if norm in ('layer_norm', 'instance_norm'):
          if block_end:
            raise ValueError("normalization %r with residual blocks: the reference calls conv_bn_res_bn_actv "
                             "without bn_momentum / bn_epsilon there (tdnn_encoder.py:144-156, 216-233)" % norm)
          if Layer is not ConvBN:
            raise NotImplementedError("normalization=%r with sep_conv1d layers" % norm)
          scope_name = ConvSampleNorm.MODES[norm][2]
          n_norm = sum(isinstance(L['main'], ConvSampleNorm) for L in layers)
          main = ConvSampleNorm(store, lname, "%s/%s%s" % (scope, scope_name, "_%d" % n_norm if n_norm else ""),
                                norm, cin, blk['num_channels'], blk['kernel_size'][0], blk['stride'][0],
                                blk['dilation'][0] if 'dilation' in blk else 1, blk['padding'], l2, initializer)
        elif norm is None and not block_end:
          if Layer is not ConvBN:
            raise NotImplementedError("normalization=None with sep_conv1d layers")
          main = ConvOnly(store, lname, cin, blk['num_channels'], blk['kernel_size'][0], blk['stride'][0],
                          blk['dilation'][0] if 'dilation' in blk else 1, blk['padding'], l2, initializer)
        else:
          main = Layer(store, lname, lname + "/bn", cin, blk['num_channels'],
                       blk['kernel_size'][0], blk['stride'][0], blk['dilation'][0] if
                       'dilation' in blk else 1, blk['padding'], mom, eps, l2, initializer)
        res = []
        if residual and ir == blk['repeat'] - 1:
          for i, rc in enumerate(res_in):
            rn = (lname + "/res_%d" % i) if dense else (lname + "/res")
            bn = (lname + "/res_bn_%d" % i) if dense else (lname + "/res_bn")
            # tf.layers.conv1d(res, filters, 1, use_bias=False): default glorot_uniform
            # residual branches go through the block's own layer type (conv_blocks.py:66,79-85)
            res.append(Layer(store, rn, bn, rc, blk['num_channels'], 1, 1, 1, "SAME", mom,
                             eps, l2, glorot_uniform_conv))
        layers.append(dict(block=ib, rep=ir, main=main, res=res, cfg=blk))
        cin = blk['num_channels']
    self._layers = layers
    self.output_dim = cin
    # Dense-residual block ends without branch tensors (parts/cnns/dense_residual.py): every residual block is
    # dense, keeps the frame rate, and its branches are plain 1x1 conv + BatchNorm pairs over channel counts that
    # are multiples of 64; block dropping keeps the branch-by-branch path
    self._dres_plan = None
    res_idx = [i for i, blk in enumerate(p['convnet_layers']) if blk.get('residual', False)]
    res_blocks = [p['convnet_layers'][i] for i in res_idx]
    # (every block from the first to the last residual one keeps the frame rate: all block inputs are [B, T, .])
    span = p['convnet_layers'][res_idx[0]:res_idx[-1] + 1] if res_idx else []
    ends = [L['res'] for L in layers if L['res']]
    if res_blocks and all(blk.get('residual_dense', False) for blk in res_blocks) and \
       all(blk['stride'][0] == 1 and blk['padding'] == "SAME" for blk in span) and \
       (p.get('drop_block_prob', 0.0) == 0.0 or self._mode != 'train') and p.get('drop_block_index', -1) == -1 and \
       DenseResidualPlan.eligible(ends):
      self._dres_ends = ends
      self._dres_plan = False        # built at the first pass (the store's device buffers exist then)
    return self

  def _encode(self, input_dict):
    """input_dict['source_tensors'] = [features bf16 [B,T,F], src_length int32 [B]].
    Returns {'outputs': [B,T',C] bf16, 'src_length': int32 [B]} (+ 'outputs_act')."""
    source_sequence, src_length = input_dict['source_tensors']
    tape = input_dict.get('tape', None)
    seed0 = int(input_dict.get('seed', 0))
    training = (self._mode == "train")
    use_mask = self.params.get("use_conv_mask", False)
    act_fn = self.params['activation_fn']
    default_keep = self.params['dropout_keep_prob']

    lens = src_length if use_mask else None
    # the data layer hands over already-padded features; mask the first conv input
    x = Act(source_sequence, lens, requires_grad=False)
    # a host copy of the lengths, when the data layer kept one ('source_lengths_host'): the convolution
    # launcher then chooses its tile at launch time instead of enqueueing both candidates
    # (capi.conv1d_set_host_lens — a hint, it cannot change results); followed through the strides below
    host_len = input_dict.get('source_lengths_host') if use_mask else None
    if host_len is not None:
      host_len = [int(v) for v in host_len]
    try:
      return self._encode_layers(x, src_length, host_len, tape, seed0, training, use_mask, act_fn, default_keep)
    finally:
      if host_len is not None:
        capi.conv1d_set_host_lens(None)

  def _encode_layers(self, x, src_length, host_len, tape, seed0, training, use_mask, act_fn, default_keep):
    residual_aggregation = []
    layer_res = []
    pending_res = None
    dpass, pending_dres = None, None
    if self._dres_plan is not None and x.data.is_cuda and x.data.shape[0] <= 64:
      if self._dres_plan is False:
        self._dres_plan = DenseResidualPlan(self._dres_ends, x.data.device)
      dpass = self._dres_plan.begin(training)
    hint_stale = host_len is not None
    nl = len(self._layers)
    for li, L in enumerate(self._layers):
      if hint_stale:
        capi.conv1d_set_host_lens(host_len)       # = the in_len of this layer's convolution
        hint_stale = False
      blk, main = L['cfg'], L['main']
      if L['rep'] == 0 and blk.get('residual', False):
        if blk.get('residual_dense', False):
          residual_aggregation.append(x)
          layer_res = list(residual_aggregation)
          if dpass is not None:
            pending_dres = launch_dense_residual(dpass, len(residual_aggregation) - 1, x)
        else:
          layer_res = [x]
      s = blk['stride'][0]
      if blk['padding'] == "VALID":
        new_len = torch.div(src_length - blk['kernel_size'][0], s, rounding_mode='floor') + 1
        if host_len is not None:
          host_len, hint_stale = [(v - blk['kernel_size'][0]) // s + 1 for v in host_len], True
      elif s > 1:
        new_len = torch.div(src_length + s - 1, s, rounding_mode='floor')
        if host_len is not None:
          host_len, hint_stale = [(v + s - 1) // s for v in host_len], True
      else:
        new_len = src_length
      src_length = new_len
      keep = blk.get('dropout_keep_prob', default_keep) if training else 1.0
      res_in = layer_res if L['res'] else []
      last = (li == nl - 1)
      # the block end's residual branches read the block inputs only: start them RES_EARLY_REP
      # layers into the block, on the side stream (conv_blocks.launch_residual_early)
      if dpass is None and blk.get('residual', False) and blk['repeat'] > 1 and s == 1 and \
         L['rep'] == min(RES_EARLY_REP, blk['repeat'] - 2) and RES_EARLY_REP >= 0:
        end = self._layers[li + blk['repeat'] - 1 - L['rep']]
        if end['res'] and len(end['res']) >= 2:
          pending_res = launch_residual_early(end['res'], layer_res, training)
      res_fw = None
      if L['res']:
        res_fw, pending_res = pending_res, None
      if isinstance(main, ConvOnly):
        x = conv_actv(main, x, src_length if use_mask else None, act_fn, training, tape, keep_prob=keep,
                      seed=seed0 * 1000003 + li, mask_output=(use_mask and not last))
        continue
      if L['res'] and pending_dres is not None:
        dfw, pending_dres = pending_dres, None
        x = conv_bn_dres_actv(main, x, dfw, src_length if use_mask else None, act_fn, training, tape,
                              keep_prob=keep, seed=seed0 * 1000003 + li, mask_output=(use_mask and not last))
        continue
      x = conv_bn_res_bn_actv(main, L['res'], x, res_in, src_length if use_mask else None,
                              act_fn, training, tape, keep_prob=keep,
                              seed=seed0 * 1000003 + li, mask_output=(use_mask and not last),
                              drop_block_prob=self.params.get('drop_block_prob', 0.0),
                              drop_block=(self.params.get('drop_block_index', -1) == L['block']),
                              res_fw=res_fw)
    return {'outputs': x.data, 'src_length': src_length, 'outputs_act': x}

"""FullyConnectedTimeDecoder / FullyConnectedCTCDecoder
(open_seq2seq/decoders/fc_decoders.py:73-251) on the HIP kernels: the dense layer
is the K=1 case of the MFMA implicit-GEMM kernel writing time-major fp32 logits,
the greedy CTC decode is os2s_ctc_greedy_decode."""
from __future__ import absolute_import, division, print_function

import math

import torch

from .decoder import Decoder
from .. import capi
from ..parts.cnns.conv_blocks import Act


class FullyConnectedTimeDecoder(Decoder):
  @staticmethod
  def get_required_params():
    return dict(Decoder.get_required_params(), **{'tgt_vocab_size': int})

  @staticmethod
  def get_optional_params():
    return dict(Decoder.get_optional_params(), **{
        'logits_to_outputs_func': None,
        'infer_logits_to_pickle': bool,
    })

  def __init__(self, params, model, name="fully_connected_time_decoder", mode='train'):
    super(FullyConnectedTimeDecoder, self).__init__(params, model, name, mode)
    self.kernel = self.bias = None

  def build(self, store, n_hidden):
    V = self.params['tgt_vocab_size']
    self.V = V
    self.Vpad = ((V + 7) // 8) * 8 if V > 32 else 32
    self.n_hidden = n_hidden
    l2 = 0.0
    if self.params.get('regularizer', None) is not None:
      l2 = float(self.params.get('regularizer_params', {}).get('scale', 0.0))
    scope = "ForwardPass/" + self._name + "/fully_connected"

    def init_kernel(shape):
      # tf.layers.dense default: glorot_uniform over [n_hidden, V]; rows >= V are padding
      lim = math.sqrt(6.0 / (n_hidden + V))
      w = torch.zeros(shape)
      w[0, :V, :] = (torch.rand(V, n_hidden) * 2 - 1) * lim
      return w

    self.kernel = store.add(scope + "/kernel", (1, self.Vpad, n_hidden), init_kernel,
                            kind="conv", l2=l2, logical_out=V)
    self.bias = store.add(scope + "/bias", (self.Vpad,), torch.zeros(self.Vpad), kind="vector",
                          logical_out=V)
    return self

  def _decode(self, input_dict):
    """encoder_output['outputs'] [B,T,A] -> logits [T,B,V] fp32 (time-major,
    fc_decoders.py:129-148) (+ 'outputs' from logits_to_outputs_func)."""
    enc = input_dict['encoder_output']
    x = enc.get('outputs_act') or Act(enc['outputs'], None)
    tape = input_dict.get('tape', None)
    B, T, A = x.data.shape
    V = self.V
    w = self.kernel.w16.view(-1)[:V * A].view(1, V, A)
    logits = capi.conv1d_fwd(x.data, w, bias=self.bias.master[:V].contiguous(), out_f32=True,
                             time_major=True)
    out = {'logits': logits, 'src_length': enc['src_length']}
    # text generation is only fetched outside training (the reference's graph has the op in
    # every mode but evaluates it on demand); a training step asks through decode_outputs()
    if 'logits_to_outputs_func' in self.params and self._mode != "train":
      out['outputs'] = self.params['logits_to_outputs_func'](logits, input_dict)
    if self._mode == "train" and tape is not None:
      dec = self
      holder = {}
      out['_dlogits_sink'] = holder   # the loss deposits dlogits_bf16 [B,T,Vpad] here

      def backward():
        dl = holder.get('dlogits_bf16')
        assert dl is not None, "loss did not provide dlogits"
        # dW [1,Vpad,A] += dl^T x ; db += column sums ; dX = dl @ W
        capi.conv1d_wgrad(x.data, dl, 1, pad_left=0, out=dec.kernel.grad, accumulate=True)
        part = capi.bn_stats(dl.view(-1, dec.Vpad))
        scratch = torch.empty(2, dec.Vpad, dtype=torch.float32, device=dl.device)
        part2 = part.view(part.shape[0], 2, dec.Vpad)
        capi.bn_bwd_finalize(part2, 1, 1, None, dec.bias.grad, True, scratch[0], scratch[1])
        if x.requires_grad:
          g = x.grad_buffer()
          capi.conv1d_fwd(dl, dec.kernel.wt16, pad_left=0, tout=T, out=g,
                          accumulate=x.grad_init)
          x.grad_init = True

      tape.record(backward, [dec.kernel, dec.bias])
    return out


def decode_outputs(decoder, decoder_output, encoder_src_length=None):
  """'outputs' of a decoder output dict, computed now if the pass skipped it (train mode)."""
  if 'outputs' in decoder_output:
    return decoder_output['outputs']
  fn = decoder.params['logits_to_outputs_func']
  return fn(decoder_output['logits'], {'encoder_output': {'src_length': decoder_output['src_length']}})


class FullyConnectedCTCDecoder(FullyConnectedTimeDecoder):
  """FC over time + CTC text generation (fc_decoders.py:160-251): greedy decode on the GPU,
  or — `use_language_model` — the prefix beam search with a word n-gram language model
  (:197-240). The reference binds a CPU-only custom TF op through `decoder_library_path`; here
  it is the host entry point os2s_ctc_beam_search of the same C-ABI library, so
  `decoder_library_path` is accepted and ignored. `lm_path`: ARPA text or the KenLM binary
  layout the reference ships a sample of (see include/os2s.h)."""

  @staticmethod
  def get_required_params():
    return FullyConnectedTimeDecoder.get_required_params()

  @staticmethod
  def get_optional_params():
    return dict(FullyConnectedTimeDecoder.get_optional_params(), **{
        'use_language_model': bool, 'decoder_library_path': str, 'beam_width': int,
        'alpha': float, 'beta': float, 'trie_weight': float, 'lm_path': str,
        'trie_path': str, 'alphabet_config_path': str,
    })

  def __init__(self, params, model, name="fully_connected_ctc_decoder", mode='train'):
    super(FullyConnectedCTCDecoder, self).__init__(params, model, name, mode)
    self.params['use_language_model'] = self.params.get('use_language_model', False)
    if self.params['use_language_model']:
      p = self.params
      # one scorer (language model + letter trie) per decoder, as the op holds one (beam_search.cc:757)
      scorer = capi.CtcScorer(p['lm_path'], p['trie_path'], p['alphabet_config_path'],
                              p['alpha'], p['beta'], p.get('trie_weight', 0.1))

      def decode_with_lm(logits, decoder_input, beam_width=p['beam_width'], top_paths=1,
                         merge_repeated=False):
        # fc_decoders.py:206-235; the logits leave the GPU here, as they do for the CPU op
        seq_len = decoder_input['encoder_output']['src_length']
        ids, lens, _ = capi.ctc_beam_search(logits.float().cpu(), seq_len.cpu(), beam_width,
                                            scorer, top_paths=top_paths,
                                            merge_repeated=merge_repeated)
        return [(ids[:, 0].to(logits.device), lens[:, 0].to(logits.device))]

      self.params['logits_to_outputs_func'] = decode_with_lm
      return

    def decode_without_lm(logits, decoder_input, merge_repeated=True):
      # fc_decoders.py:244-251: greedy decode on fp32 logits, neg_sum_logits discarded
      ids, lens, _ = capi.ctc_greedy_decode(
          logits, decoder_input['encoder_output']['src_length'], merge_repeated=merge_repeated)
      return [(ids, lens)]   # dense form of the reference's [SparseTensor]

    self.params['logits_to_outputs_func'] = decode_without_lm

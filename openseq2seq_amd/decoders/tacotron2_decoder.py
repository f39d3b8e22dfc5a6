"""Tacotron2Decoder — open_seq2seq/decoders/tacotron2_decoder.py:22-567 on the HIP kernels.

Training (TacotronTrainingHelper, teacher forcing, parts/tacotron/tacotron_helper.py:46-135):
the decoder input of step t is the target mel frame t-1 (zeros at t = 0), so the pre-net
(2 x Dense+ReLU with dropout 0.5 that is ALWAYS on, :22-63), the input projection of the
first LSTM, the mel / stop-token projections (moved outside the loop exactly as the
reference does when not sampling, :431-437), the post-net and the magnitude branch are
whole-sequence GEMMs / convolutions; only AttentionWrapper(MultiRNNCell[LSTMCell x L],
LocationSensitiveAttention, output_attention="both") is sequential (csrc/attn_decoder.hip).
dynamic_decode runs with impute_finished=False until every sample passed its length, so no
step is skipped and nothing is zeroed; the loss masks by length.
Inference (TacotronHelper, :137-226): free-running, the projected mel frame is fed back
through the pre-net; stops when every sample's sigmoid(stop token) rounds to 1 or after
10 x max source length steps."""
from __future__ import absolute_import, division, print_function

import os

import torch

from .decoder import Decoder
from .rnn_decoders import AttentionCell
from .. import capi
from ..parts.cnns.conv_blocks import (Act, ConvBN, accumulate_grad, conv_bn_actv, reshape_act,
                                      xavier_normal_conv)
from ..parts.transformer.layers import Dense, SeedSeq


PRENET_KEEP = 0.5   # tf.layers.dropout(rate=0.5, training=True): on in every mode (:63)


def _round8(n):
  return (n + 7) // 8 * 8


class Tacotron2Decoder(Decoder):
  @staticmethod
  def get_required_params():
    return dict(Decoder.get_required_params(), **{
        'attention_layer_size': int, 'attention_type': ['bahdanau', 'location', None],
        'decoder_cell_units': int, 'decoder_cell_type': None, 'decoder_layers': int,
    })

  @staticmethod
  def get_optional_params():
    return dict(Decoder.get_optional_params(), **{
        'bahdanau_normalize': bool, 'time_major': bool, 'use_swap_memory': bool,
        'enable_prenet': bool, 'prenet_layers': int, 'prenet_units': int, 'prenet_activation': None,
        'enable_postnet': bool, 'postnet_conv_layers': list, 'postnet_bn_momentum': float,
        'postnet_bn_epsilon': float, 'postnet_data_format': ['channels_first', 'channels_last'],
        'postnet_keep_dropout_prob': float, 'mask_decoder_sequence': bool,
        'attention_bias': bool, 'zoneout_prob': float, 'dropout_prob': float,
        'parallel_iterations': int,
        # extension (BASELINE.json configs[4] "fp8 weights"; the reference has no fp8): the recurrent
        # weight matrices of the decoder LSTM stack are streamed as OCP e4m3 with per-row scales
        'fp8_weights': bool,
    })

  def __init__(self, params, model, name='tacotron_2_decoder', mode='train'):
    super(Tacotron2Decoder, self).__init__(params, model, name, mode)
    p = self.params
    if p['attention_type'] != 'location':
      raise NotImplementedError("attention_type %r (location-sensitive attention is built)" % (p['attention_type'],))
    if p.get('zoneout_prob', 0.) != 0.:
      raise NotImplementedError("zoneout")
    if not p.get('enable_prenet', True) or not p.get('enable_postnet', True):
      raise NotImplementedError("decoder without pre-net / post-net")
    if p['decoder_layers'] not in (1, 2):
      raise NotImplementedError("decoder_layers > 2")
    ct = p['decoder_cell_type']
    ct = ct if isinstance(ct, str) else getattr(ct, "__name__", "")
    if "LSTM" not in ct:
      raise NotImplementedError("decoder_cell_type %s" % ct)

  def build(self, store, memory_dim=None, num_audio_features=None, exp_mag=None):
    p = self.params
    if num_audio_features is None:
      dlp = self._model.get_data_layer().params
      num_audio_features = dlp['num_audio_features']
      exp_mag = dlp.get('exp_mag', False)
    self._both = isinstance(num_audio_features, dict)
    self.n_mel = num_audio_features['mel'] if self._both else num_audio_features
    self.n_mag = num_audio_features['magnitude'] if self._both else 0
    self.exp_mag = bool(exp_mag)
    self.M = memory_dim if memory_dim is not None else p.get('_memory_dim')
    H = p['decoder_cell_units']
    self.H = H
    scope = "ForwardPass/" + self._name
    l2 = 0.0
    if p.get('regularizer', None) is not None:
      l2 = float(p.get('regularizer_params', {}).get('scale', 0.0))
    pu = p.get('prenet_units', 256)
    self.prenet = []
    cin = self.n_mel
    for i in range(p.get('prenet_layers', 2)):
      d = Dense(store, "%s/prenet_%d" % (scope, i + 1), cin, pu, True)
      d.kernel.l2 = l2
      self.prenet.append(d)
      cin = pu
    self.cell = AttentionCell(store, scope + "/attention_wrapper", pu, H, self.M,
                              p['attention_layer_size'], p['decoder_layers'], capi.SCORE_LOCATION,
                              1.0, use_bias=p.get('attention_bias', False), loc_k=32, loc_f=32)
    self.cell.fp8_weights = bool(p.get('fp8_weights', False))
    for w in [self.cell.w_in, self.cell.w_mem, self.cell.w_q] + self.cell.wcat:
      w.l2 = l2
    self.out_proj = Dense(store, scope + "/output_proj", H + self.M, self.n_mel, True)
    self.stop_proj = Dense(store, scope + "/stop_token_proj", self.n_mel, 8, True)   # 1 unit, padded to 8
    self.out_proj.kernel.l2 = self.stop_proj.kernel.l2 = l2
    self.stop_proj.kernel.logical_out = self.stop_proj.bias.logical_out = 1      # checkpoints carry the one unit
    mom, eps = p.get('postnet_bn_momentum', 0.1), p.get('postnet_bn_epsilon', 1e-5)
    self.postnet = []
    cin = self.n_mel
    for i, cl in enumerate(p['postnet_conv_layers']):
      cout = cl['num_channels'] if cl['num_channels'] != -1 else self.n_mel
      n = "%s/conv%d" % (scope, i + 1)
      self.postnet.append(ConvBN(store, n, n + "/bn", cin, cout, cl['kernel_size'][0],
                                 stride=cl['stride'][0], padding=cl['padding'], bn_momentum=mom,
                                 bn_epsilon=eps, l2=l2, initializer=xavier_normal_conv))
      cin = cout
    self.mag = None
    if self._both:
      n0, n1 = scope + "/conv_0", scope + "/conv_1"
      self.mag = [ConvBN(store, n0, n0 + "/bn", self.n_mel, 256, 4, bn_momentum=mom, bn_epsilon=eps, l2=l2),
                  ConvBN(store, n1, n1 + "/bn", 256, 512, 4, bn_momentum=mom, bn_epsilon=eps, l2=l2)]
      self.n_mag_pad = _round8(self.n_mag)

      def init(shape):
        w = xavier_normal_conv(shape)
        w[:, self.n_mag:, :] = 0.0
        return w

      self.mag_proj = store.add(scope + "/post_net_proj/kernel", (1, self.n_mag_pad, 512), init, kind="conv",
                                logical_out=self.n_mag)
    return self

  # ---------------------------------------------------------------- teacher-forced pass
  def _decode(self, input_dict):
    enc = input_dict['encoder_output']
    enc_act = enc.get('outputs_act') or Act(enc['outputs'], None, requires_grad=False)
    src_len = enc['src_length']
    training = self._mode == "train"
    if not training:
      return self._free_running(enc_act, src_len, input_dict.get('max_decoder_steps'))
    p = self.params
    tape = input_dict.get('tape')
    seeds = enc.get('seeds') or SeedSeq(37)
    spec = input_dict['target_tensors'][0]              # fp32 [B, T, n_mel (+ n_mag)]
    B, T, _ = spec.shape
    dev = spec.device
    nm, H, M = self.n_mel, self.H, self.M
    # decoder inputs: previous target frame (TacotronTrainingHelper.next_inputs)
    prev = torch.zeros((B, T, nm), dtype=torch.bfloat16, device=dev)
    prev[:, 1:] = spec[:, :-1, :nm]
    x = Act(prev.view(B * T, nm), requires_grad=False)
    for d in self.prenet:
      x = d.forward(x, tape, act=1, keep=PRENET_KEEP, seed=seeds.next())
    x3 = reshape_act(x, (B, T, -1), tape)
    both = Act(torch.empty((B, T, H + M), dtype=torch.bfloat16, device=dev))
    out_keep = 1.0 - p.get('dropout_prob', 0.1)
    y, ctx, loop = self.cell.forward_train(x3, enc_act, src_len, None, tape, seeds, attn_in_keep=1.0,
                                           out_keep=out_keep, y_top=both.data[:, :, :H],
                                           ctx=both.data[:, :, H:])
    if tape is not None:
      def split_bwd():      # gradient of concat(cell output, attention) -> the loop's two inputs
        y.grad, c_grad = both.grad[:, :, :H], both.grad[:, :, H:]
        ctx.grad = c_grad
      tape.record(split_bwd)
      # forward_train recorded its backward BEFORE split_bwd, so it runs after it
    flat = reshape_act(both, (B * T, H + M), tape)
    mel2 = self.out_proj.forward(flat, tape)                      # [B*T, n_mel]
    stop2 = self.stop_proj.forward(mel2, tape)                    # [B*T, 8]; unit 0 is the logit
    mel = reshape_act(mel2, (B, T, nm), tape)
    # post-net
    keep_post = p.get('postnet_keep_dropout_prob', 0.5)
    top = mel
    for cl, layer in zip(p['postnet_conv_layers'], self.postnet):
      top = conv_bn_actv(layer, top, None, cl['activation_fn'], True, tape, keep_prob=keep_post,
                         seed=seeds.next(), mask_output=False)
    post = Act(capi.add_bf16(mel.data, top.data))
    if tape is not None:
      def add_bwd():
        accumulate_grad(mel, post.grad)
        accumulate_grad(top, post.grad)
        post.grad = None
      tape.record(add_bwd)
    mag = None
    if self._both:
      m = conv_bn_actv(self.mag[0], post, None, "relu", True, tape, mask_output=False)
      m = conv_bn_actv(self.mag[1], m, None, "relu", True, tape, mask_output=False)
      if self.exp_mag:
        e = Act(capi.exp_fwd(m.data))
        if tape is not None:
          def exp_bwd(m=m, e=e):
            accumulate_grad(m, capi.mul_bf16(e.grad, e.data))
            e.grad = None
          tape.record(exp_bwd)
        m = e
      mflat = m.data.view(B * T, 512)
      mag2 = capi.gemm(mflat, self.mag_proj.w16.view(self.n_mag_pad, 512))
      mag = Act(mag2.view(B, T, self.n_mag_pad))
      if tape is not None:
        dec = self

        def mag_bwd(m=m):
          dl = mag.grad.reshape(B * T, dec.n_mag_pad)
          capi.gemm_wgrad(mflat, dl, dec.mag_proj.grad.view(dec.n_mag_pad, 512), accumulate=True)
          g = m.grad_buffer()
          capi.gemm(dl, dec.mag_proj.wt16.view(512, dec.n_mag_pad), out=g.view(B * T, 512),
                    accumulate=m.grad_init)
          m.grad_init = True
          mag.grad = None
        tape.record(mag_bwd, [self.mag_proj])
    stop = Act(stop2.data.view(B, T, 8))
    if tape is not None:
      def stop_bwd():
        stop2.grad = stop.grad.reshape(B * T, 8)
        stop.grad = None
      tape.record(stop_bwd)
    return {
        'outputs': [mel.data, post.data, loop.align_seq, stop.data[:, :, :1], None,
                    mag.data[:, :, :self.n_mag] if mag is not None else None],
        'stop_token_prediction': stop.data[:, :, :1],
        'acts': {'mel': mel, 'post': post, 'stop': stop, 'mag': mag},
        'n_feats': (self.n_mel, self.n_mag),
    }

  # ---------------------------------------------------------------- inference
  POLL_STEPS = 32      # host looks at the device-resident stop decision every POLL_STEPS steps

  def _infer_weights(self, fp8):
    """Inference copies the fused step kernels read (csrc/tacotron_infer.hpp), rebuilt when the bf16
    weights changed: layer 0 with the pre-net columns in front of the attention / state columns
    ([kernel_inputs | kernel_attention_state], optionally e4m3 with per-row scales) and the output
    projection split into its cell-output and context column blocks."""
    cell = self.cell
    store = getattr(cell.wcat[0], "store", None)
    ver = getattr(store, "version", None) if store is not None else None
    c = getattr(self, "_infer_cache", None)
    if c is not None and ver is not None and c["ver"] == ver and c["fp8"] == fp8:
      return c["w"]
    H, GH = self.H, 4 * self.H
    w0x = torch.cat([cell.w_in.w16.view(GH, -1), cell.wcat[0].w16.view(GH, -1)], dim=1).contiguous()
    w = {"w0x": w0x, "w0x8": capi.quantize_rows_e4m3(w0x) if fp8 else None,
         "bias0": cell.bias[0].master,
         "wp1": self.prenet[0].w.contiguous(), "bp1": self.prenet[0].bias.master,
         "wp2": self.prenet[1].w.contiguous(), "bp2": self.prenet[1].bias.master,
         "wout_h": self.out_proj.w[:, :H].contiguous(), "wout_c": self.out_proj.w[:, H:].contiguous(),
         "bout": self.out_proj.bias.master,
         "wstop": self.stop_proj.w[0].contiguous(), "bstop": self.stop_proj.bias.master[:1].contiguous()}
    self._infer_cache = {"ver": ver, "fp8": fp8, "w": w}
    return w

  def _free_running(self, enc_act, src_len, max_steps=None):
    """TacotronHelper decoding (tacotron_helper.py:138-226, tacotron2_decoder.py:378-428): at most
    10 x max(src_len) steps (`max_steps` overrides: benchmarks), until every sample's stop token has
    fired. Returns the reference's outputs plus `acts` (what Text2SpeechLoss reads: eval-mode loss)."""
    p = self.params
    B, S, M = enc_act.data.shape
    dev = enc_act.data.device
    nm, H = self.n_mel, self.H
    T = int(max_steps) if max_steps else 10 * int(src_len.max().item())
    seeds = SeedSeq(41)
    both = torch.zeros((B, T, H + M), dtype=torch.bfloat16, device=dev)
    cell = self.cell
    loop = cell._new_loop(B, T, S, dev, False, 1.0, 1.0, seeds, y_top=both[:, :, :H], ctx=both[:, :, H:])
    keys = cell.memory(enc_act, None)
    mask_seq = p.get('mask_decoder_sequence', True)
    prenet_seeds = (seeds.next(), seeds.next())
    fused = None
    if len(self.prenet) == 2 and os.environ.get("OS2S_TACOTRON_FUSED_DECODE", "1") != "0":
      w = dict(self._infer_weights(bool(getattr(cell, "fp8_weights", False))))
      values2d = enc_act.data.reshape(B * S, M)
      # once per batch: PV = values W_out[:, H:]^T (the context half of the frame projection commutes with the
      # attention sum) and the transposed copies the step kernels' weighted sums read (positions contiguous,
      # rows zero-padded to a multiple of 32)
      Sp, nm16 = (S + 31) // 32 * 32, (nm + 15) // 16 * 16
      pv = capi.gemm(values2d, w["wout_c"], out_f32=True).view(B, S, nm)
      w["pv_t"] = torch.zeros((B, nm16, Sp), dtype=torch.bfloat16, device=dev)
      w["pv_t"][:, :nm, :S] = pv.transpose(1, 2)
      w["values_t"] = torch.zeros((B, M, Sp), dtype=torch.bfloat16, device=dev)
      w["values_t"][:, :, :S] = enc_act.data.transpose(1, 2)
      loop.set_inputs(torch.zeros(8, dtype=torch.bfloat16, device=dev), keys, enc_act.data, src_len, None)
      fused = capi.TacotronInfer(loop, self.prenet[0].cout, nm, w, mask_seq, PRENET_KEEP, prenet_seeds)
      if not fused.supported():
        fused = None
    if fused is not None:
      steps, mels, stops, lengths = self._decode_fused(fused, T)
    else:
      steps, mels, stops, lengths = self._decode_stepwise(loop, keys, enc_act, src_len, both, T, mask_seq,
                                                          prenet_seeds)
    mel = Act(mels[:, :steps].contiguous())
    top = mel
    for cl, layer in zip(p['postnet_conv_layers'], self.postnet):
      top = conv_bn_actv(layer, top, None, cl['activation_fn'], False, None, mask_output=False)
    post = Act(capi.add_bf16(mel.data, top.data))
    mag = None
    if self._both:
      m = conv_bn_actv(self.mag[0], post, None, "relu", False, None, mask_output=False)
      m = conv_bn_actv(self.mag[1], m, None, "relu", False, None, mask_output=False)
      md = capi.exp_fwd(m.data) if self.exp_mag else m.data
      mag = Act(capi.gemm(md.view(-1, 512), self.mag_proj.w16.view(self.n_mag_pad, 512))
                .view(B, steps, self.n_mag_pad))
    st = stops[:, :steps].contiguous().view(B, steps, 1)
    stop8 = torch.zeros((B, steps, 8), dtype=torch.bfloat16, device=dev)     # the training pass's padded layout
    stop8[:, :, :1] = st.to(torch.bfloat16)
    return {'outputs': [mel.data, post.data, loop.align_seq[:, :steps], torch.sigmoid(st.float()),
                        lengths, mag.data[:, :, :self.n_mag] if mag is not None else None],
            'stop_token_prediction': st, 'n_feats': (self.n_mel, self.n_mag),
            'acts': {'mel': mel, 'post': post, 'stop': Act(stop8), 'mag': mag},
            'decoder_steps': steps, 'fused': fused is not None}

  def _decode_fused(self, fused, T):
    """Four launches per step, enqueued POLL_STEPS at a time; the stop decision is taken on the device
    (kernels enqueued past it return at once), the host reads it one chunk behind the enqueue front so the
    GPU never waits for the host. The result does not depend on POLL_STEPS."""
    # chunk k is enqueued, then an async snapshot of the stop word is taken BEHIND it (pinned copy + event),
    # then chunk k + 1 goes out, and only then does the host wait for the snapshot of chunk k: the GPU always
    # has a chunk queued while the host looks (state[1].item() drained the whole stream, chunk k + 1 included)
    done, t0 = 0, 0
    pending = None
    while t0 < T and not done:
      t1 = min(T, t0 + self.POLL_STEPS)
      fused.steps(t0, t1)
      t0 = t1
      ev = fused.poll_async()
      if pending is not None:
        done = fused.poll_wait(pending)
      pending = ev
    if not done and pending is not None:
      done = fused.poll_wait(pending)
    steps = done if done else T
    return steps, fused.mel, fused.stop, fused.lengths.clone()

  def _decode_stepwise(self, loop, keys, enc_act, src_len, both, T, mask_seq, prenet_seeds):
    """Generic path (shapes the fused step kernels are not built for): one os2s_attn_decoder_fwd call and
    a few GEMM launches per step; the finished test stays on the device, the host polls it every
    POLL_STEPS steps."""
    B = enc_act.data.shape[0]
    dev = enc_act.data.device
    nm, GH = self.n_mel, 4 * self.H
    cell = self.cell
    gx0 = torch.zeros((B, T, GH), dtype=torch.bfloat16, device=dev)
    loop.set_inputs(gx0, keys, enc_act.data, src_len, None)
    frame = torch.zeros((B, nm), dtype=torch.bfloat16, device=dev)
    mels = torch.zeros((B, T, nm), dtype=torch.bfloat16, device=dev)
    stops = torch.zeros((B, T), dtype=torch.float32, device=dev)
    finished = torch.zeros((B,), dtype=torch.bool, device=dev)
    lengths = torch.zeros((B,), dtype=torch.int32, device=dev)
    all_done_at = torch.zeros((), dtype=torch.int32, device=dev)    # first step count with everyone finished
    steps = T
    for t in range(T):
      x = Act(frame, requires_grad=False)
      for li, d in enumerate(self.prenet):          # pre-net dropout stays on at inference (:63)
        x = d.forward(x, None, act=1, keep=PRENET_KEEP, seed=prenet_seeds[li % 2] * 1000003 + t)
      gx0[:, t] = capi.gemm(x.data, cell.w_in.w16.view(GH, -1), bias=cell.bias[0].master)
      loop.forward(t, t + 1)
      mel_t = capi.gemm(both[:, t].contiguous(), self.out_proj.w, bias=self.out_proj.bias.master)
      stop_t = capi.gemm(mel_t, self.stop_proj.w, bias=self.stop_proj.bias.master)[:, 0].float()
      mels[:, t], stops[:, t] = mel_t, stop_t
      running = ~finished
      lengths += (running & (all_done_at == 0)).to(torch.int32)
      if mask_seq:
        finished = finished | (stop_t > 0)          # round(sigmoid(s)) == 1
        all_done_at = torch.where((all_done_at == 0) & finished.all(),
                                  torch.full_like(all_done_at, t + 1), all_done_at)
      frame = mel_t
      if mask_seq and (t + 1) % self.POLL_STEPS == 0 and int(all_done_at.item()):
        break
    if mask_seq and int(all_done_at.item()):
      steps = int(all_done_at.item())
    return steps, mels, stops, lengths

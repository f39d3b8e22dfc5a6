"""Text2Text — open_seq2seq/models/text2text.py:60-241 (NMT model shell): vocabulary sizes
flow from the data layer into encoder/decoder/loss params (:60-80); the tokens-per-step
metric counts source + target tokens (:227-241)."""
from __future__ import absolute_import, division, print_function

from .encoder_decoder import EncoderDecoderModel
from ..parts.transformer.layers import SeedSeq


class Text2Text(EncoderDecoderModel):
  def _build_forward_pass_objects(self, store):
    self._data_layer = self._create_data_layer()
    dl = self._data_layer
    self.params['encoder_params']['src_vocab_size'] = dl.params['src_vocab_size']
    self.params['decoder_params']['batch_size'] = self.params['batch_size_per_gpu']
    self.params['decoder_params']['tgt_vocab_size'] = dl.params['tgt_vocab_size']
    self.params['loss_params']['batch_size'] = self.params['batch_size_per_gpu']
    self.params['loss_params']['tgt_vocab_size'] = dl.params['tgt_vocab_size']
    self._encoder = self._create_encoder()
    self._decoder = self._create_decoder()
    if self.mode in ("train", "eval"):
      self._loss_computator = self._create_loss()
    self._encoder.build(store)
    if hasattr(self._encoder, "output_dim"):    # attention memory depth for RNN decoders
      self._decoder.params['_memory_dim'] = self._encoder.output_dim
    self._decoder.build(store)

  def _forward_backward(self, batch, tape):
    seeds = SeedSeq(self._seed * 7919 + self._step_count)
    enc = self._encoder.encode({'source_tensors': batch['source_tensors'], 'tape': tape,
                                'seeds': seeds, 'packed_source': batch.get('packed_source')})
    dec = self._decoder.decode({'encoder_output': enc, 'target_tensors': batch['target_tensors'],
                                'tape': tape, 'packed_target': batch.get('packed_target')})
    scale_dev = self._train_op.loss_scale_view if self._train_op is not None else None
    return self._loss_computator.compute_loss({
        'decoder_output': dec, 'target_tensors': batch['target_tensors'],
        'loss_scale_dev': scale_dev})

  # ---- experiment (os2s_half_batches / OS2S_HALF_BATCHES=1): one step as two half-batches on two streams ------------
  def _halves_enabled(self):
    import os
    return bool(self.params.get('os2s_half_batches', os.environ.get("OS2S_HALF_BATCHES", "0") == "1")) and \
        self._reducer is None and self.params.get('iter_size', 1) == 1 and hasattr(self._encoder, "embedding_softmax_layer")

  def _split_batch(self, batch):
    """The two halves of a batch (by sentences), with their packed forms; cached on the batch (the bench batch is
    reused every step, a data layer would hand the halves over ready-made)."""
    h = batch.get('_halves')
    if h is not None:
      return h
    from ..parts.transformer import packing
    (src, slen), (tgt, tlen) = batch['source_tensors'], batch['target_tensors']
    B = src.shape[0]
    out = []
    for sl in (slice(0, B // 2), slice(B // 2, B)):
      s, sn, t, tn = (v[sl].contiguous() for v in (src, slen, tgt, tlen))
      ps = packing.pack_ids(s.cpu().numpy(), sn.cpu().numpy())
      pt = packing.pack_ids(t.cpu().numpy(), tn.cpu().numpy(), shift_right=True)
      out.append({'source_tensors': [s, sn], 'target_tensors': [t, tn], 'n_tgt': pt["n"],
                  'packed_source': packing.to_device(ps, s.device), 'packed_target': packing.to_device(pt, s.device)})
    batch['_halves'] = out
    return out

  def _forward_backward_halves(self, batch):
    """Transformer has no batch statistics: the batch's gradient is the token-weighted sum of its halves' gradients.
    Each half runs forward on its own stream, the two backward passes are issued closure by closure in turn
    (conv_blocks.backward_interleaved); parameter gradients of both halves queue on ONE side stream."""
    import torch
    from ..parts.cnns import conv_blocks
    from ..parts.cnns.conv_blocks import Tape
    from .. import capi
    halves = self._split_batch(batch)
    main = torch.cuda.current_stream()
    if getattr(self, "_half_streams", None) is None:
      self._half_streams = [torch.cuda.Stream(device=main.device) for _ in halves]
    n_tot = float(sum(h['n_tgt'] for h in halves))
    scale_dev = self._train_op.loss_scale_view if self._train_op is not None else None
    tapes, losses = [], []
    conv_blocks._SIDE_KEY_OVERRIDE = capi._stream().value
    try:
      for k, (h, st) in enumerate(zip(halves, self._half_streams)):
        st.wait_stream(main)
        torch.cuda.set_stream(st)
        tape = Tape()
        w = h['n_tgt'] / n_tot
        seeds = SeedSeq(self._seed * 7919 + self._step_count + 104729 * k)
        enc = self._encoder.encode({'source_tensors': h['source_tensors'], 'tape': tape, 'seeds': seeds,
                                    'packed_source': h['packed_source']})
        dec = self._decoder.decode({'encoder_output': enc, 'target_tensors': h['target_tensors'], 'tape': tape,
                                    'packed_target': h['packed_target']})
        sd = scale_dev * w if scale_dev is not None else torch.full((1,), w, dtype=torch.float32, device=main.device)
        loss = self._loss_computator.compute_loss({'decoder_output': dec, 'target_tensors': h['target_tensors'],
                                                   'loss_scale_dev': sd})
        tapes.append(tape)
        losses.append(loss * w)
      torch.cuda.set_stream(main)
      conv_blocks.backward_interleaved(tapes, self._half_streams)
    finally:
      torch.cuda.set_stream(main)
      conv_blocks._SIDE_KEY_OVERRIDE = None
    return losses[0] + losses[1]

  def infer_batch(self, batch):
    """eval / infer: encoder + greedy (RNN) or beam-search (Transformer) decoding
    (models/text2text.py:98-190 without the printing). Returns (ids int32 [B, steps], lengths [B])."""
    assert self.mode in ("eval", "infer")
    enc = self._encoder.encode({'source_tensors': batch['source_tensors'],
                                'packed_source': batch.get('packed_source')})
    dec = self._decoder.decode({'encoder_output': enc})
    ids, lens = dec['outputs'][0], dec['final_sequence_lengths']
    if lens is None:     # Transformer beam search: rows are zero-padded after the first EOS
      lens = self._decoder.sequence_lengths(ids)
    return ids, lens

  def finalize_inference(self, results_per_batch, output_file):
    """models/text2text.py:112-124: one decoded sentence per line (special tokens dropped)."""
    import codecs
    dl = self.get_data_layer()
    idx2seq = getattr(dl, "tgt_idx2seq", None) or {}
    delim = dl.params.get("delimiter", " ")
    with codecs.open(output_file, "w", "utf-8") as fout:
      for ids, lens in results_per_batch:
        ids, lens = ids.cpu().numpy(), lens.cpu().numpy()
        for b in range(ids.shape[0]):
          toks = [int(t) for t in ids[b, :lens[b]] if int(t) not in (0, 1, 2, 3)]
          fout.write(delim.join(idx2seq.get(t, str(t)) for t in toks) + "\n")

  def evaluate(self, device=None, max_batches=None):
    """Greedy-decodes the eval set and scores it against the targets: corpus BLEU-4 (the
    reference prints "Eval BLUE score", text2text.py:192-225) and exact-match rate."""
    from ..utils.metrics import corpus_bleu
    dl = self.get_data_layer()
    hyps, refs = [], []
    for n, batch in enumerate(dl.iterate_batches(device or self._device, drop_remainder=False)):
      if max_batches is not None and n >= max_batches:
        break
      ids, lens = self.infer_batch(batch)
      ids, lens = ids.cpu().numpy(), lens.cpu().numpy()
      tgt, tl = batch['target_tensors'][0].cpu().numpy(), batch['target_tensors'][1].cpu().numpy()
      for b in range(ids.shape[0]):
        h = [int(t) for t in ids[b, :lens[b]] if t not in (0, 1, 2)]
        r = [int(t) for t in tgt[b, :tl[b]] if t not in (0, 1, 2)]
        hyps.append(h)
        refs.append(r)
    exact = sum(1 for h, r in zip(hyps, refs) if h == r) / max(len(hyps), 1)
    return {"bleu": corpus_bleu(refs, hyps), "exact_match": exact, "samples": len(hyps)}

  def _get_num_objects_per_step(self, batch):
    """text2text.py:227-241: source tokens + target tokens."""
    return batch['source_tensors'][1].sum() + batch['target_tensors'][1].sum()

"""GPU parity of the RNN NMT model (en-de-nmt-small architecture, scaled down): 2-layer
bidirectional LSTM encoder with embedding -> GNMT attention decoder (gnmt_v2 and gnmt) ->
BasicSequenceLoss; loss and all parameter gradients vs the CPU fp32 oracle with the same
bf16-rounded weights, dropout off. Tolerances: loss rel 2e-2; gradients cosine >= 0.99 and
relative L2 <= 0.12 (bf16 storage of activations / gate gradients through two recurrences).
Plus the reference's own relational test: BasicSequenceLoss == CrossEntropyWithSmoothing(0)
on the same logits (losses/sequence_loss_test.py:58-60)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cmp(got, ref, name, cos_min=0.99, rel_max=0.12):
  got, ref = got.float().cpu().flatten(), ref.detach().float().flatten()
  if float(ref.norm()) < 1e-9 and float(got.norm()) < 1e-6:
    return
  cos = float(torch.nn.functional.cosine_similarity(got, ref, dim=0))
  rel = float((got - ref).norm() / (ref.norm() + 1e-12))
  assert cos > cos_min and rel < rel_max, (name, cos, rel)


def _build(cuda, attention_type, V=30, E=64, H=64, layers=2):
  from openseq2seq_amd.optimizers.flat_params import FlatParams
  from openseq2seq_amd.encoders import BidirectionalRNNEncoderWithEmbedding
  from openseq2seq_amd.decoders import RNNDecoderWithAttention
  from openseq2seq_amd.losses import BasicSequenceLoss
  store = FlatParams(cuda)
  cellp = {"num_units": H, "forget_bias": 1.0}
  enc = BidirectionalRNNEncoderWithEmbedding(
      {"src_vocab_size": V, "src_emb_size": E, "encoder_layers": layers,
       "encoder_use_skip_connections": False, "core_cell": "LSTMCell", "core_cell_params": cellp,
       "encoder_dp_input_keep_prob": 1.0, "dtype": "mixed"}, None, mode="train").build(store)
  dec = RNNDecoderWithAttention(
      {"GO_SYMBOL": 2, "END_SYMBOL": 1, "tgt_vocab_size": V, "tgt_emb_size": E,
       "attention_layer_size": 128, "attention_type": attention_type, "core_cell": "LSTMCell",
       "core_cell_params": cellp, "decoder_layers": layers, "decoder_use_skip_connections": False,
       "decoder_dp_input_keep_prob": 1.0, "batch_size": 4, "dtype": "mixed"}, None, mode="train")
  dec.build(store, memory_dim=2 * H)
  loss = BasicSequenceLoss({"tgt_vocab_size": V, "batch_size": 4, "offset_target_by_one": True,
                            "average_across_timestep": False, "do_mask": True, "dtype": "mixed"}, None)
  store.finalize()
  return store, enc, dec, loss


def _oracle_params(store, enc, dec, V):
  """fp32 leaves of the bf16 compute copies, keyed for oracle.nmt + name map to device params."""
  leaves = {}

  def leaf(p, view=None, bf16=True):
    t = (p.w16.float() if bf16 else p.master).cpu().clone()
    if view is not None:
      t = t.view(*view)
    t.requires_grad_(True)
    leaves[p.name] = t
    return t

  P = {"emb": leaf(enc.embedding.table)}
  for key, stack in zip(("fw", "bw"), enc.stacks):
    P[key] = [dict(wx=leaf(l.wx[0], (4 * l.H, -1)), wh=leaf(l.wh, (4 * l.H, l.H)), b=leaf(l.bx, None, False))
              for l in stack]
  c = dec.cell
  H, M, U = c.H, c.M, c.U
  cell = dict(wcat=[leaf(c.wcat[0], (4 * H, M + H))], bias=[None], wq=leaf(c.w_q, (U, H)),
              wmem=leaf(c.w_mem, (U, M)), v=leaf(c.v, None, False), g=leaf(c.g, None, False),
              b=leaf(c.b, None, False), w_in=leaf(c.w_in, (4 * H, -1)), b0=leaf(c.bias[0], None, False))
  D = {"demb": leaf(dec.embedding.table), "cell": cell,
       "upper": [dict(wx_h=leaf(l.wx[0], (4 * H, H)), wx_a=leaf(l.wx[1], (4 * H, M)),
                      wh=leaf(l.wh, (4 * H, H)), b=leaf(l.bx, None, False)) for l in dec.upper],
       "proj": leaf(dec.proj, (dec.Vpad, -1))}
  return P, D, leaves


@pytest.mark.parametrize("attention_type", ["gnmt_v2", "gnmt"])
def test_nmt_small_fwd_bwd(cuda, attention_type):
  from openseq2seq_amd.parts.cnns.conv_blocks import Tape
  from openseq2seq_amd.parts.transformer.layers import SeedSeq
  from oracle import nmt as onmt
  torch.manual_seed(0)
  V = 30
  store, enc, dec, lossf = _build(cuda, attention_type, V=V)
  g = torch.Generator().manual_seed(1)
  for p in store.params:      # non-trivial biases / attention vectors
    if p.kind == "vector" and p.numel > 1:
      p.master.add_((torch.randn(p.shape, generator=g) * 0.1).to(cuda))
  store.refresh_compute_copies()
  B, S, T = 4, 11, 9
  src_len = torch.tensor([11, 6, 9, 3], dtype=torch.int32)
  tgt_len = torch.tensor([9, 4, 7, 2], dtype=torch.int32)
  src = torch.randint(4, V, (B, S), generator=g).to(torch.int32)
  tgt = torch.randint(4, V, (B, T), generator=g).to(torch.int32)
  for b in range(B):
    src[b, src_len[b]:] = 0
    tgt[b, 0] = 2
    tgt[b, tgt_len[b] - 1] = 1
    tgt[b, tgt_len[b]:] = 0
  tape = Tape()
  store.zero_grads()
  e = enc.encode({"source_tensors": [src.to(cuda), src_len.to(cuda)], "tape": tape, "seeds": SeedSeq(3)})
  d = dec.decode({"encoder_output": e, "target_tensors": [tgt.to(cuda), tgt_len.to(cuda)], "tape": tape})
  L = lossf.compute_loss({"decoder_output": d, "target_tensors": [tgt.to(cuda), tgt_len.to(cuda)]})
  tape.backward()
  torch.cuda.synchronize()
  # ---- oracle -------------------------------------------------------------------
  P, D, leaves = _oracle_params(store, enc, dec, V)
  enc_out = onmt.encoder(P, src, src_len)
  logits = onmt.decoder_logits(D, enc_out, src_len, tgt, tgt_len, attention_type)[..., :V]
  ref = onmt.basic_sequence_loss(logits, tgt, tgt_len, 4)
  ref.backward()
  torch.testing.assert_close(e["outputs"].float().cpu(), enc_out.detach(), atol=3e-2, rtol=3e-2)
  live = (torch.arange(T)[None, :] < tgt_len[:, None])
  got_logits = d["logits"].float().cpu()[..., :V]
  torch.testing.assert_close(got_logits[live], logits.detach()[live], atol=5e-2, rtol=5e-2)
  assert abs(float(L.item()) - float(ref)) <= 2e-2 * abs(float(ref)), (float(L.item()), float(ref))
  bad = []
  for p in store.params:
    gref = leaves[p.name].grad
    if gref is None:
      gref = torch.zeros_like(leaves[p.name])
    try:
      _cmp(p.grad.reshape(-1), gref.reshape(-1), p.name)
    except AssertionError as ex:
      bad.append(ex.args[0])
  assert not bad, bad
  # outputs are available lazily (argmax of the teacher-forced logits)
  ids = d["lazy_outputs"]()[0].cpu()
  assert torch.equal(ids[live], got_logits.argmax(-1).to(torch.int32)[live])


def test_basic_sequence_loss_equals_smoothing0(cuda):
  """BasicSequenceLoss == padded cross entropy with label_smoothing 0, up to the different
  normalisation (sum/batch vs mean over target tokens) — losses/sequence_loss_test.py:58-60."""
  from openseq2seq_amd import capi
  from openseq2seq_amd.losses import BasicSequenceLoss
  from openseq2seq_amd.parts.cnns.conv_blocks import Act
  g = torch.Generator().manual_seed(0)
  B, T, V = 5, 7, 40
  logits = (torch.randn(B, T, V, generator=g) * 2).to(torch.bfloat16).to(cuda)
  tgt = torch.randint(1, V, (B, T), generator=g).to(torch.int32).to(cuda)
  tgt_len = torch.tensor([7, 3, 5, 2, 6], dtype=torch.int32, device=cuda)
  lf = BasicSequenceLoss({"tgt_vocab_size": V, "batch_size": B}, None)
  la = Act(logits)
  loss = lf.compute_loss({"decoder_output": {"logits": logits, "logits_act": la},
                          "target_tensors": [tgt, tgt_len]})
  labels, cur = lf.loss_labels(tgt, tgt_len, T)
  rows = labels.reshape(-1) >= 0
  packed = logits.reshape(B * T, V)[rows].contiguous()
  row_loss, mean, _ = capi.xent_smooth(packed, labels.reshape(-1)[rows].contiguous(), 0.0, want_grad=False)
  n = int(rows.sum())
  assert n == int((tgt_len - 1).sum())
  torch.testing.assert_close(loss.cpu(), mean.cpu() * n / B, rtol=1e-5, atol=1e-5)
  # gradient rows of masked positions are exactly zero
  assert float(la.grad.reshape(B * T, V)[~rows].abs().max()) == 0.0


@pytest.mark.parametrize("weight_tied", [False, True])
def test_gnmt_like_encoder_and_skip_connections(cuda, weight_tied):
  """(weight_tied: en-de-gnmt-like-weight-tied-2GPUs.py — the decoder embedding is the transposed
  output projection.) en-de-gnmt-like architecture scaled down: GNMTLikeEncoderWithEmbedding (1 bidirectional + 2
  unidirectional layers, residual on the last) -> 3-layer gnmt_v2 decoder with
  decoder_use_skip_connections; loss, logits and every gradient vs the fp32 oracle."""
  from openseq2seq_amd.optimizers.flat_params import FlatParams
  from openseq2seq_amd.encoders import GNMTLikeEncoderWithEmbedding
  from openseq2seq_amd.decoders import RNNDecoderWithAttention
  from openseq2seq_amd.losses import BasicSequenceLoss
  from openseq2seq_amd.parts.cnns.conv_blocks import Tape
  from openseq2seq_amd.parts.transformer.layers import SeedSeq
  from oracle import nmt as onmt
  torch.manual_seed(0)
  V, E, H = 30, 64, 64
  store = FlatParams(cuda)
  cellp = {"num_units": H, "forget_bias": 1.0}
  enc = GNMTLikeEncoderWithEmbedding(
      {"src_vocab_size": V, "src_emb_size": E, "encoder_layers": 3, "encoder_use_skip_connections": True,
       "core_cell": "LSTMCell", "core_cell_params": cellp, "encoder_dp_input_keep_prob": 1.0,
       "dtype": "mixed"}, None, mode="train").build(store)
  dec = RNNDecoderWithAttention(
      {"GO_SYMBOL": 2, "END_SYMBOL": 1, "tgt_vocab_size": V, "tgt_emb_size": E,
       "attention_layer_size": 128, "attention_type": "gnmt_v2", "core_cell": "LSTMCell",
       "core_cell_params": cellp, "decoder_layers": 3, "decoder_use_skip_connections": True,
       "decoder_dp_input_keep_prob": 1.0, "batch_size": 4, "dtype": "mixed", "weight_tied": weight_tied},
      None, mode="train")
  dec.build(store, memory_dim=enc.output_dim)
  assert any(p.name.endswith("DecoderEmbeddingMatrix") for p in store.params) == (not weight_tied)
  if weight_tied:      # shared variable at the embedding's position: final last in backward
    names = [q.name for q in store.params if q.name.startswith("ForwardPass/" + dec._name)]
    assert names[0] == dec.proj.name
  lossf = BasicSequenceLoss({"tgt_vocab_size": V, "batch_size": 4, "offset_target_by_one": True,
                             "average_across_timestep": False, "do_mask": True, "dtype": "mixed"}, None)
  store.finalize()
  g = torch.Generator().manual_seed(1)
  for p in store.params:
    if p.kind == "vector" and p.numel > 1:
      p.master.add_((torch.randn(p.shape, generator=g) * 0.1).to(cuda))
  store.refresh_compute_copies()
  B, S, T = 4, 11, 9
  src_len = torch.tensor([11, 6, 9, 3], dtype=torch.int32)
  tgt_len = torch.tensor([9, 4, 7, 2], dtype=torch.int32)
  src = torch.randint(4, V, (B, S), generator=g).to(torch.int32)
  tgt = torch.randint(4, V, (B, T), generator=g).to(torch.int32)
  for b in range(B):
    src[b, src_len[b]:] = 0
    tgt[b, 0] = 2
    tgt[b, tgt_len[b] - 1] = 1
    tgt[b, tgt_len[b]:] = 0
  tape = Tape()
  store.zero_grads()
  e = enc.encode({"source_tensors": [src.to(cuda), src_len.to(cuda)], "tape": tape, "seeds": SeedSeq(3)})
  d = dec.decode({"encoder_output": e, "target_tensors": [tgt.to(cuda), tgt_len.to(cuda)], "tape": tape})
  L = lossf.compute_loss({"decoder_output": d, "target_tensors": [tgt.to(cuda), tgt_len.to(cuda)]})
  tape.backward()
  torch.cuda.synchronize()
  leaves = {}

  def leaf(p, view=None, bf16=True):
    t = (p.w16.float() if bf16 else p.master).cpu().clone()
    if view is not None:
      t = t.view(*view)
    t.requires_grad_(True)
    leaves[p.name] = t
    return t

  lay = lambda l: dict(wx=leaf(l.wx[0], (4 * l.H, -1)), wh=leaf(l.wh, (4 * l.H, l.H)), b=leaf(l.bx, None, False))
  P = {"emb": leaf(enc.embedding.table), "l1fw": lay(enc.l1[0]), "l1bw": lay(enc.l1[1]),
       "uni": [lay(l) for l in enc.uni]}
  c = dec.cell
  M, U = c.M, c.U
  cell = dict(wcat=[leaf(c.wcat[0], (4 * H, M + H))], bias=[None], wq=leaf(c.w_q, (U, H)),
              wmem=leaf(c.w_mem, (U, M)), v=leaf(c.v, None, False), g=leaf(c.g, None, False),
              b=leaf(c.b, None, False), w_in=leaf(c.w_in, (4 * H, -1)), b0=leaf(c.bias[0], None, False))
  proj_leaf = leaf(dec.proj, (dec.Vpad, -1))
  D = {"demb": proj_leaf if weight_tied else leaf(dec.embedding.table), "cell": cell,
       "upper": [dict(wx_h=leaf(l.wx[0], (4 * H, H)), wx_a=leaf(l.wx[1], (4 * H, M)),
                      wh=leaf(l.wh, (4 * H, H)), b=leaf(l.bx, None, False)) for l in dec.upper],
       "proj": proj_leaf}
  enc_out = onmt.gnmt_like_encoder(P, src, src_len)
  logits = onmt.decoder_logits(D, enc_out, src_len, tgt, tgt_len, "gnmt_v2", skip=True)[..., :V]
  ref = onmt.basic_sequence_loss(logits, tgt, tgt_len, 4)
  ref.backward()
  torch.testing.assert_close(e["outputs"].float().cpu(), enc_out.detach(), atol=4e-2, rtol=4e-2)
  assert abs(float(L.item()) - float(ref)) <= 2e-2 * abs(float(ref)), (float(L.item()), float(ref))
  bad = []
  for p in store.params:
    gref = leaves[p.name].grad
    if gref is None:
      gref = torch.zeros_like(leaves[p.name])
    try:
      _cmp(p.grad.reshape(-1), gref.reshape(-1), p.name, cos_min=0.985, rel_max=0.17)
    except AssertionError as ex:
      bad.append(ex.args[0])
  assert not bad, bad

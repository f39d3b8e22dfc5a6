"""GPU parity of the Transformer beam-search path through the C ABI:

* os2s_beam_init/step/finalize + os2s_gather_rows (sequence_beam_search) vs the NumPy oracle
  (oracle/beam_search.py, pinned to the reference's known answers): decoded sequences are
  compared exactly, scores to rtol 1e-5 (logsumexp reduction order), with fp32 and bf16
  logits, several chunks per row, exact ties, early termination and the no-EOS corner case;
* os2s_decode_self_attention / os2s_decode_cross_attention vs a torch fp32 restatement
  (atol 2e-2 on bf16 outputs of O(1) magnitude);
* the incremental TransformerDecoder step vs the (oracle-checked) full decode_pass, and the
  beam-search scores vs log-probabilities recomputed by decode_pass."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import beam_search as obs  # noqa: E402


CASES = {
    # name: B, beam, V, T, eos bias, dtype, quantise, alpha
    "fp32_multichunk": (3, 4, 9000, 14, 2.0, torch.float32, None, 0.6),
    "bf16_small_vocab": (5, 3, 40, 10, 1.0, torch.bfloat16, None, 0.6),
    "bf16_ties": (4, 4, 4500, 12, 1.5, torch.bfloat16, 0.5, 0.6),
    "beam1_alpha0": (2, 1, 300, 9, 1.0, torch.float32, None, 0.0),
    "wide_beam": (2, 16, 5000, 8, 2.5, torch.float32, None, 1.0),
    "never_eos": (2, 2, 64, 6, -1e4, torch.float32, None, 0.6),
    "huge_vocab_chunked": (2, 2, 70000, 4, 3.0, torch.float32, None, 0.6),
    "bf16_32k": (3, 4, 32768, 6, 3.0, torch.bfloat16, None, 0.6),
    "bf16_48k_unaligned": (2, 3, 48004, 5, 3.0, torch.bfloat16, 0.25, 0.6),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_sequence_beam_search_vs_oracle(cuda, case):
  from openseq2seq_amd.parts.transformer.beam_search import sequence_beam_search
  B, beam, V, T, eos_bias, dtype, quant, alpha = CASES[case]
  rng = np.random.RandomState(sum(map(ord, case)))
  Vt = min(V, 64)                           # previous-token dependence through a small alphabet
  table = (rng.randn(T, Vt, V) * 2.0).astype(np.float32)
  table[:, :, 1] += eos_bias                # EOS = 1
  if quant:
    table = np.round(table / quant) * quant
  # rows of the table are indexed by (last id % Vt) to keep it small
  tab = torch.from_numpy(table).to(cuda).to(dtype)
  tab_host = tab.float().cpu().numpy()

  def host_fn(ids, i, cache):
    return tab_host[i][ids[:, -1] % Vt], cache

  init = torch.zeros(B, dtype=torch.int32, device=cuda)

  # the device fn records each step's chosen id in the cache AFTER the step via the ids of the
  # next call; here the cache row simply carries its own history for the gather check
  class Fn(object):
    def __init__(self):
      self.ok = []          # steps enqueued after the search stopped see frozen state: not checked

    def __call__(self, ids, i, cache):
      if i > 0:
        cache["hist"][:, i - 1] = ids[:, -1].float()
      self.ok.append(torch.equal(cache["hist"][:, :i], ids[:, 1:i + 1].float()))
      last = ids[:, -1].long() % Vt
      return tab[i][last].contiguous(), cache

  cache = {"hist": torch.zeros((B, T), dtype=torch.float32, device=cuda)}
  fn = Fn()
  seq, scores = sequence_beam_search(fn, init, cache, V, beam, alpha, T, 1, poll_every=3)
  rseq, rscores, rsteps = obs.sequence_beam_search(host_fn, np.zeros(B, np.int32), {}, V, beam, alpha,
                                                   T, 1, return_steps=True)
  assert seq.shape[2] == rsteps + 1
  assert all(fn.ok[:rsteps]), fn.ok
  assert np.array_equal(seq.cpu().numpy(), rseq), case
  np.testing.assert_allclose(scores.cpu().numpy(), rscores, rtol=1e-5, atol=1e-5)
  if case == "never_eos":
    assert rsteps == T
  # independence from the polling interval
  cache = {"hist": torch.zeros((B, T), dtype=torch.float32, device=cuda)}
  seq2, scores2 = sequence_beam_search(Fn(), init, cache, V, beam, alpha, T, 1, poll_every=1)
  assert torch.equal(seq, seq2) and torch.equal(scores, scores2)


def test_early_termination(cuda):
  """A peaked model finishes long before max_decode_length; the device loop condition must
  stop at the same step as the reference's _continue_search."""
  from openseq2seq_amd.parts.transformer.beam_search import sequence_beam_search
  V, T, B, beam = 50, 40, 3, 2
  table = np.full((T, V), -8.0, np.float32)
  table[:, 7] = 4.0
  table[3:, 1] = 12.0       # EOS dominates from step 3 on
  tab = torch.from_numpy(table).to(cuda)

  def fn(ids, i, cache):
    return tab[i][None, :].expand(ids.shape[0], V).contiguous(), cache

  def host_fn(ids, i, cache):
    return np.tile(table[i][None], (ids.shape[0], 1)), cache

  seq, scores = sequence_beam_search(fn, torch.zeros(B, dtype=torch.int32, device=cuda), {}, V, beam, 0.6,
                                     T, 1, poll_every=4)
  rseq, rscores, rsteps = obs.sequence_beam_search(host_fn, np.zeros(B, np.int32), {}, V, beam, 0.6, T, 1,
                                                   return_steps=True)
  assert rsteps < 10 and seq.shape[2] == rsteps + 1
  assert np.array_equal(seq.cpu().numpy(), rseq)
  # log-probs of near-certain tokens are differences of O(10) numbers: absolute tolerance
  np.testing.assert_allclose(scores.cpu().numpy(), rscores, rtol=1e-5, atol=1e-5)
  assert seq[0, 0, :5].tolist() == [0, 7, 7, 7, 1]


def test_gather_rows(cuda):
  from openseq2seq_amd import capi
  x = torch.arange(24, dtype=torch.int32, device=cuda).reshape(6, 4)
  idx = torch.tensor([1, 2, 3, 5, 0, 0], dtype=torch.int32, device=cuda)
  assert torch.equal(capi.gather_rows(x, idx), x[idx.long()])
  off = torch.zeros(4, dtype=torch.int32, device=cuda)
  assert torch.equal(capi.gather_rows(x, idx, enable=off), x)
  y = torch.randn(7, 3, 10, device=cuda).to(torch.bfloat16)
  idx = torch.tensor([6, 6, 0, 2, 1, 1, 3], dtype=torch.int32, device=cuda)
  assert torch.equal(capi.gather_rows(y, idx), y[idx.long()])


@pytest.mark.parametrize("H,step", [(4, 0), (4, 5), (8, 70), (2, 199), (16, 33)])
def test_decode_self_attention(cuda, H, step):
  from openseq2seq_amd import capi
  N, dh, Tmax = 6, 64, 200
  D = H * dh
  g = torch.Generator().manual_seed(step + H)
  qkv = (torch.randn(N, 3 * D, generator=g)).to(torch.bfloat16).to(cuda)
  kc = torch.randn(N, Tmax, D, generator=g).to(torch.bfloat16).to(cuda)
  vc = torch.randn(N, Tmax, D, generator=g).to(torch.bfloat16).to(cuda)
  anc = torch.randint(0, N, (N, Tmax), generator=g, dtype=torch.int32).to(cuda)
  kc0, vc0 = kc.clone(), vc.clone()
  scale = dh ** -0.5
  o = capi.decode_self_attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], kc, vc, anc, H, step, scale)
  torch.cuda.synchronize()
  # cache append + ancestry of the new slot
  assert torch.equal(kc[:, step], qkv[:, D:2 * D]) and torch.equal(vc[:, step], qkv[:, 2 * D:])
  assert torch.equal(anc[:, step].cpu(), torch.arange(N, dtype=torch.int32))
  mask = torch.ones(Tmax, dtype=torch.bool)
  mask[step] = False
  assert torch.equal(kc[:, mask], kc0[:, mask]) and torch.equal(vc[:, mask], vc0[:, mask])
  # reference
  a = anc.long().cpu()
  kf, vf, q = kc.float().cpu(), vc.float().cpu(), qkv[:, :D].float().cpu()
  ref = torch.zeros(N, D)
  for n in range(N):
    rows = a[n, :step + 1]
    pos = torch.arange(step + 1)
    K = kf[rows, pos].view(step + 1, H, dh)
    Vv = vf[rows, pos].view(step + 1, H, dh)
    s = torch.einsum("hd,thd->ht", q[n].view(H, dh), K) * scale
    w = torch.softmax(s, -1)
    ref[n] = torch.einsum("ht,thd->hd", w, Vv).reshape(D)
  torch.testing.assert_close(o.float().cpu(), ref, atol=2e-2, rtol=2e-2)


def test_decode_cross_attention(cuda):
  from openseq2seq_amd import capi
  B, beam, H, dh = 3, 4, 4, 64
  D = H * dh
  lens = [5, 130, 64]
  cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
  g = torch.Generator().manual_seed(0)
  kv = torch.randn(int(cu[-1]), 2 * D, generator=g).to(torch.bfloat16).to(cuda)
  q = torch.randn(B * beam, D, generator=g).to(torch.bfloat16).to(cuda)
  scale = dh ** -0.5
  o = capi.decode_cross_attention(q, kv[:, :D], kv[:, D:], cu.to(cuda), beam, H, max(lens), scale)
  kvf, qf = kv.float().cpu(), q.float().cpu()
  for n in range(B * beam):
    b = n // beam
    K = kvf[cu[b]:cu[b + 1], :D].view(-1, H, dh)
    Vv = kvf[cu[b]:cu[b + 1], D:].view(-1, H, dh)
    s = torch.einsum("hd,thd->ht", qf[n].view(H, dh), K) * scale
    ref = torch.einsum("ht,thd->hd", torch.softmax(s, -1), Vv).reshape(D)
    torch.testing.assert_close(o[n].float().cpu(), ref, atol=2e-2, rtol=2e-2)


# ---- Transformer decoder: incremental step vs decode_pass, beam search consistency ----------------
def _tiny_transformer(cuda, V=200, D=512, H=8, NL=2, beam=4, extra=6):
  from openseq2seq_amd.optimizers.flat_params import FlatParams
  from openseq2seq_amd.encoders.transformer_encoder import TransformerEncoder
  from openseq2seq_amd.decoders.transformer_decoder import TransformerDecoder
  torch.manual_seed(3)
  store = FlatParams(cuda)
  enc = TransformerEncoder({"encoder_layers": NL, "hidden_size": D, "num_heads": H,
                            "attention_dropout": 0.1, "filter_size": 512, "src_vocab_size": V,
                            "relu_dropout": 0.1, "layer_postprocess_dropout": 0.1,
                            "remove_padding": True, "dtype": "mixed"}, None, mode="infer").build(store)
  dec = TransformerDecoder({"EOS_ID": 1, "layer_postprocess_dropout": 0.1, "num_hidden_layers": NL,
                            "hidden_size": D, "num_heads": H, "attention_dropout": 0.1,
                            "relu_dropout": 0.1, "filter_size": 512, "batch_size": 4,
                            "tgt_vocab_size": V, "beam_size": beam, "alpha": 0.6,
                            "extra_decode_length": extra, "dtype": "mixed"}, None,
                           mode="infer").build(store)
  store.finalize()
  # sharpen the output distribution so that beams are well separated
  emb = store.by_name("ForwardPass/transformer_encoder/embedding_shared_weights/embedding_and_softmax/weights")
  emb.master.mul_(4.0)
  store.refresh_compute_copies()
  return store, enc, dec


def _src_batch(cuda, V, B=3, Lmax=9, seed=0):
  rng = np.random.RandomState(seed)
  lens = rng.randint(3, Lmax + 1, size=B).astype(np.int32)
  lens[0] = Lmax
  ids = np.zeros((B, Lmax), np.int32)
  for b in range(B):
    ids[b, :lens[b] - 1] = rng.randint(4, V, size=lens[b] - 1)
    ids[b, lens[b] - 1] = 1
  return torch.from_numpy(ids).to(cuda), torch.from_numpy(lens).to(cuda)


def test_incremental_step_matches_decode_pass(cuda):
  V = 200
  store, enc, dec = _tiny_transformer(cuda, V=V)
  src, sl = _src_batch(cuda, V)
  B = src.shape[0]
  e = enc.encode({'source_tensors': [src, sl]})
  T = 12
  rng = np.random.RandomState(1)
  tgt = rng.randint(4, V, size=(B, T)).astype(np.int32)
  tl = np.full(B, T, np.int32)
  full = dec.decode({'encoder_output': e, 'target_tensors': [torch.from_numpy(tgt).to(cuda),
                                                             torch.from_numpy(tl).to(cuda)]})["logits"]
  full = full.float().cpu().view(B, T, -1)
  fn = dec._get_symbols_to_logits_fn(e, 1, B, T)
  cache = {"ancestry": torch.zeros((B, T), dtype=torch.int32, device=cuda)}
  ids = torch.zeros((B, T + 1), dtype=torch.int32, device=cuda)
  ids[:, 1:] = torch.from_numpy(tgt).to(cuda)
  for i in range(T):
    logits, cache = fn(ids[:, :i + 1], i, cache)
    torch.testing.assert_close(logits.float().cpu(), full[:, i], atol=6e-2, rtol=3e-2)


@pytest.mark.parametrize("Lmax", [9, 90])
def test_transformer_beam_search_consistency(cuda, Lmax):
  """Lmax 90: sources / decoded targets longer than one attention tile (multi-tile forward)."""
  V = 200
  store, enc, dec = _tiny_transformer(cuda, V=V)
  src, sl = _src_batch(cuda, V, B=4, Lmax=Lmax, seed=2)
  e = enc.encode({'source_tensors': [src, sl]})
  out = dec.decode({'encoder_output': e})
  ids, scores = out["outputs"][0], out["scores"]
  # the hipGraph-replayed loop (default) and the eager host loop run the same kernels
  eager = dec.decode({'encoder_output': e, 'use_graph': False})
  assert torch.equal(eager["outputs"][0], ids) and torch.equal(eager["scores"], scores)
  B, L = ids.shape
  assert L <= src.shape[1] + 6
  lens = dec.sequence_lengths(ids)
  # recompute the log-probability of the returned sequences with the training-path kernels
  full = dec.decode({'encoder_output': e, 'target_tensors': [ids, lens]})["logits"].float().cpu()
  lp = torch.log_softmax(full, -1)
  cu = np.concatenate([[0], np.cumsum(lens.cpu().numpy())])
  idc = ids.cpu().numpy()
  for b in range(B):
    n = int(lens[b])
    tot = sum(float(lp[cu[b] + t, idc[b, t]]) for t in range(n))
    finished = idc[b, n - 1] == 1
    expect = tot / ((5.0 + n) / 6.0) ** 0.6 if finished else tot
    assert abs(float(scores[b, 0]) - expect) < 0.05 * max(1.0, abs(expect)), (b, float(scores[b, 0]), expect)
    assert np.all(idc[b, n:] == 0)
  # beam search can only improve on greedy search under the same scoring
  dec1 = dec
  dec1.params["beam_size"] = 1
  out1 = dec1.decode({'encoder_output': e})
  dec1.params["beam_size"] = 4
  assert torch.all(scores[:, 0] >= out1["scores"][:, 0] - 1e-3)


@pytest.mark.parametrize("M,N,K,bias,relu,res", [
    (256, 1024, 1024, False, False, True), (37, 3072, 512, False, False, False),
    (256, 1024, 4096, True, False, True), (200, 4096, 1024, True, True, False),
    (5, 40, 72, True, False, True), (64, 2048, 1032, False, False, False)])
@pytest.mark.parametrize("variant", ["l64", "l32", "reg", "wide"])
def test_gemm_skinny(cuda, M, N, K, bias, relu, res, variant, monkeypatch):
  """fp32 reference on the same bf16 inputs; bf16 output rounding: rtol 1e-2, atol 2e-2."""
  from openseq2seq_amd import capi
  monkeypatch.setenv("OS2S_SKINNY_VARIANT", variant)
  g = torch.Generator().manual_seed(M + N + K)
  x = (torch.randn(M, K, generator=g)).to(torch.bfloat16).to(cuda)
  w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(cuda)
  b = torch.randn(N, generator=g).to(cuda) if bias else None
  r = torch.randn(M, N, generator=g).to(torch.bfloat16).to(cuda) if res else None
  y = capi.gemm_skinny(x, w, bias=b, relu=relu, residual=r)
  ref = x.float() @ w.float().t()
  if bias:
    ref = ref + b
  if relu:
    ref = torch.relu(ref)
  if res:
    ref = ref + r.float()
  torch.testing.assert_close(y.float(), ref, rtol=1e-2, atol=2e-2)
  # strided input (a column slice of a wider buffer)
  wide = torch.zeros(M, K + 64, dtype=torch.bfloat16, device=cuda)
  wide[:, 64:] = x
  y2 = capi.gemm_skinny(wide[:, 64:], w, bias=b, relu=relu, residual=r)
  assert torch.equal(y, y2)

"""The RNN beam-search oracle against the REFERENCE'S OWN CODE.

tests/golden/ref_exec_nmt_beam.npz = open_seq2seq's BidirectionalRNNEncoderWithEmbedding (infer) ->
BeamSearchRNNDecoderWithAttention (decoders/rnn_decoders.py:324-532) over the reference's own BeamSearchDecoder
(parts/rnns/rnn_beam_search_decoder.py: beams 1.. start finished with log-probability -inf, _mask_probs,
_get_scores / _length_penalty, top_k over beam x vocabulary, the state re-gathered by parent beam, gather_tree in
finalize) under dynamic_decode(maximum_iterations = 2 * max source length), executed from the reference's files by
tests/golden/make_ref_exec.py. The oracle composition the device's beam search is tested against —
oracle/rnn_beam_search.py driving oracle/nmt.py's decoder on the growing prefix of every beam — must return the same
top-beam ids, and every beam's length / finished flag, exactly, and the beams' log-probabilities to 1e-5."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import ref_exec_util as rx  # noqa: E402
from oracle import nmt as onmt  # noqa: E402
from oracle import rnn_beam_search as orb  # noqa: E402

ENC = "ForwardPass/bidir_rnn_encoder_with_emb/"
SC = "ForwardPass/rnn_decoder_with_attention/"
ATT = SC + "decoder/multi_rnn_cell/cell_0_attention/gnmt_attention/"


def oracle_params(d, case):
  """reference variable names -> the oracle's parameter dicts; values from the generator's seeded function."""
  D = rx.gen.NMT_FULL
  E, H, NL = D["E"], D["H"], D["layers"]
  M = 2 * H
  seed = int(d["seed"])
  names = [str(n) for n in d[case + "/var_names"]]
  used = set()

  def v(n):
    used.add(n)
    return torch.from_numpy(rx.gen.nmt_beam_variable(n, tuple(int(x) for x in d["%s/shape/%s" % (case, n)]), seed,
                                                     rx.gen.NMT_BEAM_CASES[case]["gain"]))

  def lyr(prefix, cin):
    k = v(prefix + "/kernel").t()
    return {"wx": k[:, :cin], "wh": k[:, cin:], "b": v(prefix + "/bias")}
  PE = {"emb": v(ENC + "EncoderEmbeddingMatrix")}
  for key in ("fw", "bw"):
    PE[key] = [lyr(ENC + "bidirectional_rnn/%s/multi_rnn_cell/cell_%d/lstm_cell" % (key, i), E if i == 0 else H)
               for i in range(NL)]
  k0 = v(ATT + "lstm_cell/kernel").t()
  cell = {"w_in": k0[:, :E], "wcat": [k0[:, E:]], "b0": v(ATT + "lstm_cell/bias"), "bias": [None],
          "wq": v(ATT + "bahdanau_attention/query_layer/kernel").t(),
          "wmem": v(SC + "AttentionMechanism/memory_layer/kernel").t(),
          "v": v(ATT + "bahdanau_attention/attention_v"), "g": v(ATT + "bahdanau_attention/attention_g"),
          "b": v(ATT + "bahdanau_attention/attention_b")}
  upper = []
  for i in range(1, NL):
    k = v(SC + "decoder/multi_rnn_cell/cell_%d/lstm_cell/kernel" % i).t()
    upper.append({"wx_h": k[:, :H], "wx_a": k[:, H:H + M], "wh": k[:, H + M:],
                  "b": v(SC + "decoder/multi_rnn_cell/cell_%d/lstm_cell/bias" % i)})
  PD = {"demb": v(SC + "DecoderEmbeddingMatrix"), "cell": cell, "upper": upper, "proj": v(SC + "decoder/dense/kernel").t()}
  assert used == set(names), "every reference variable is consumed by the oracle, and nothing else"
  return PE, PD


def oracle_beam_search(d, case):
  cfg = rx.gen.NMT_BEAM_CASES[case]
  D = rx.gen.NMT_FULL
  B, V, W = D["B"], D["V"], cfg["beam"]
  PE, PD = oracle_params(d, case)
  src, src_len = torch.from_numpy(d["src"]), torch.from_numpy(d["src_len"])
  with torch.no_grad():
    enc = onmt.encoder(PE, src, src_len)
    enc_t = enc.repeat_interleave(W, 0)                       # tile_batch: b0, b0, ..., b1, b1, ...
    len_t = src_len.repeat_interleave(W, 0)
    prefix = {"ids": None}

    def logits_fn(ids, time, parents):
      # the decoder's recurrent state is a function of the tokens fed so far: re-decode every beam's prefix
      cur = torch.from_numpy(np.asarray(ids)).long()[:, None]
      prefix["ids"] = cur if time == 0 else torch.cat([prefix["ids"][torch.from_numpy(parents).long()], cur], 1)
      p = prefix["ids"]
      lg = onmt.decoder_logits(PD, enc_t, len_t, p, torch.full((B * W,), p.shape[1], dtype=torch.int32),
                               attention_type=cfg["att"], skip=False)
      return lg[:, -1].numpy()
    return orb.beam_search(logits_fn, B, W, V, 2, cfg["END"], cfg["lp"], 2 * int(src_len.max()), return_log_probs=True)


@pytest.mark.parametrize("case", sorted(rx.gen.NMT_BEAM_CASES))
def test_oracle_reproduces_the_reference_rnn_beam_search(case):
  d = dict(np.load(os.path.join(HERE, "golden", "ref_exec_nmt_beam.npz")))
  pred, lengths, scores, _, _ = oracle_beam_search(d, case)
  ref = d[case + "/top_ids"]
  assert pred.shape[:2] == ref.shape, (pred.shape, ref.shape)
  assert np.array_equal(pred[:, :, 0], ref), (pred[:, :, 0], ref)
  assert np.array_equal(lengths, d[case + "/lengths"])
  # dynamic_decode's own sequence lengths: steps taken while the beam SLOT was unfinished
  assert d[case + "/final_sequence_lengths"].shape == lengths.shape


@pytest.mark.parametrize("case", sorted(rx.gen.NMT_BEAM_CASES))
def test_beam_log_probabilities_and_finished_flags(case):
  """final_state.log_probs / .finished of every beam (not only the top one)."""
  d = dict(np.load(os.path.join(HERE, "golden", "ref_exec_nmt_beam.npz")))
  cfg = rx.gen.NMT_BEAM_CASES[case]
  pred, lengths, scores, log_probs, finished = oracle_beam_search(d, case)
  assert np.abs(log_probs - d[case + "/log_probs"]).max() < 1e-5 * np.abs(d[case + "/log_probs"]).max()
  assert np.array_equal(finished, d[case + "/finished"])
  # the scores of the last step are log_probs / penalty(lengths) for every beam that did not finish AT that step
  # (an END candidate is scored one shorter than the length it ends with)
  T = pred.shape[1]
  ended_last = (pred[:, T - 1, :] == cfg["END"]) & (lengths == T)
  want = log_probs / orb.length_penalty(lengths, cfg["lp"])
  assert np.abs(scores - want)[~ended_last].max() < 1e-5 * np.abs(want).max()


def test_fixture_has_the_cases_that_matter():
  d = dict(np.load(os.path.join(HERE, "golden", "ref_exec_nmt_beam.npz")))
  fin = d["lp0_beam4/finished"]
  assert fin.all(1).any() and (~fin).all(1).any() and (fin.any(1) & ~fin.all(1)).any(), \
      "rows whose beams all finished, rows that ran to the cap, and a row with both kinds"
  assert len({int(v) for v in d["lp0_beam4/lengths"].reshape(-1)}) >= 5
  stable = np.concatenate([d[c + "/stable"] for c in rx.gen.NMT_BEAM_CASES])
  fin_top = np.concatenate([d[c + "/finished"][:, 0] for c in rx.gen.NMT_BEAM_CASES])
  assert stable.sum() >= 6 and (stable & fin_top).sum() >= 4 and (stable & ~fin_top).sum() >= 1, \
      "rows whose winner survives bf16-size perturbations of the variables: finished ones and ones that ran to the cap"
  assert len({int(v) for c in rx.gen.NMT_BEAM_CASES for v in d[c + "/lengths"][:, 0]}) >= 6, "winners of six lengths"
  assert not stable.all(), "and rows whose winner is decided by the last bits (near-ties at the beam boundary)"


@pytest.mark.skipif(not os.path.isdir("/root/reference/open_seq2seq"), reason="reference checkout not present")
def test_generator_reproduces_the_committed_fixture():
  r = subprocess.run([sys.executable, os.path.join(HERE, "golden", "make_ref_exec.py"), "--check", "nmt_beam"],
                     capture_output=True, text=True, timeout=600)
  assert r.returncode == 0 and r.stdout.count("reproduced") == 1, r.stdout + r.stderr

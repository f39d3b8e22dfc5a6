"""GPU parity of free-running Tacotron2 decoding (Tacotron2Decoder in eval / infer mode,
decoders/tacotron2_decoder.py:378-428; TacotronHelper parts/tacotron/tacotron_helper.py:138-226) on the
fused step kernels (csrc/tacotron_infer.hpp: os2s_tacotron_infer_steps, four launches per step, the stop
decision on the device) against the CPU oracle (oracle/tacotron.py:decoder_infer), plus the eval-mode loss
(Text2SpeechLoss with prediction and target padded to a common length, losses/text2speech_loss.py:80-131).

Tolerances: frames / stop logits / alignments are compared over the first steps of the trajectory (a
free-running LSTM + attention loop amplifies bf16 rounding from step to step): relative L2 <= 3e-2 for bf16
weights at the scaled-down sizes and at the configuration's sizes (H = M = 1024, S = 200, B = 32), <= 6e-2 with
e4m3 weights against the oracle on the SAME dequantised weights (measured at the configuration's sizes: 1.8e-3
frames, 8e-4 alignments, 2.6e-3 stop logits, bf16 and e4m3 alike); integer outputs (sequence lengths, executed
steps) are bit-exact against the reference's rule applied to the device's own stop logits."""
import pytest
import torch

pytestmark = pytest.mark.gpu

POST = [{"kernel_size": [5], "stride": [1], "num_channels": 64, "padding": "SAME", "activation_fn": "tanh"},
        {"kernel_size": [5], "stride": [1], "num_channels": -1, "padding": "SAME", "activation_fn": None}]
POST_FULL = [{"kernel_size": [5], "stride": [1], "num_channels": 512, "padding": "SAME", "activation_fn": "tanh"}] * 4 + \
            [{"kernel_size": [5], "stride": [1], "num_channels": -1, "padding": "SAME", "activation_fn": None}]


def _rel(a, b):
  a, b = a.float().cpu(), b.float().cpu()
  return float((a - b).norm() / (b.norm() + 1e-20))


def _build(cuda, H, M, P, NM, NG, post, mode="eval", fp8=False, seed=0, mask_seq=True):
  from openseq2seq_amd.optimizers.flat_params import FlatParams
  from openseq2seq_amd.decoders import Tacotron2Decoder
  torch.manual_seed(seed)
  store = FlatParams(cuda)
  dec = Tacotron2Decoder({"attention_layer_size": 128, "attention_type": "location", "attention_bias": True,
                          "decoder_cell_units": H, "decoder_cell_type": "LSTMCell", "decoder_layers": 2,
                          "dropout_prob": 0.1, "enable_prenet": True, "prenet_layers": 2, "prenet_units": P,
                          "enable_postnet": True, "postnet_keep_dropout_prob": 0.5, "postnet_conv_layers": post,
                          "mask_decoder_sequence": mask_seq, "fp8_weights": fp8, "dtype": "mixed"}, None, mode=mode)
  nf = {"mel": NM, "magnitude": NG} if NG else NM
  dec.build(store, memory_dim=M, num_audio_features=nf, exp_mag=False)
  store.finalize()
  g = torch.Generator().manual_seed(seed + 1)
  for p in store.params:      # biases / score vectors away from their zero initial values
    if p.kind == "vector" and p.numel > 1 and "gamma" not in p.name:
      p.master.add_((torch.randn(p.shape, generator=g) * 0.1).to(cuda))
  store.refresh_compute_copies()
  return store, dec


def _oracle_params(dec, fp8=False):
  """fp32 copies of the decoder's weights (bf16-rounded where the device reads bf16) in the layout
  oracle.tacotron.decoder / decoder_infer take. fp8: the recurrent matrices as the fused kernels stream
  them — e4m3 values x per-row scale of the CONCATENATED layer-0 row / of the layer-1 row."""
  c = dec.cell
  H, M, U, GH = c.H, c.M, c.U, 4 * c.H
  f = lambda p, *view: (p.w16.float().cpu().view(*view) if view else p.w16.float().cpu())
  m = lambda p: p.master.cpu().clone()
  w_in, w0, w1 = f(c.w_in, GH, -1), f(c.wcat[0], GH, M + H), f(c.wcat[1], GH, 2 * H)
  if fp8:
    def q(w):
      sc = w.abs().amax(dim=1, keepdim=True) / 448.0
      sc = torch.where(sc > 0, sc, torch.ones_like(sc))
      return (w / sc).clamp(-448, 448).to(torch.float8_e4m3fn).float() * sc
    w0x = q(torch.cat([w_in, w0], 1))
    w_in, w0, w1 = w0x[:, :w_in.shape[1]].contiguous(), w0x[:, w_in.shape[1]:].contiguous(), q(w1)
  cell = dict(wcat=[w0, w1], bias=[None, m(c.bias[1])], wq=f(c.w_q, U, H), wmem=f(c.w_mem, U, M), v=m(c.v),
              b=m(c.b) if c.b is not None else None, conv_w=m(c.conv_w), conv_b=m(c.conv_b),
              dense_w=m(c.dense_w), w_in=w_in, b0=m(c.bias[0]))
  return {"prenet": [(f(d.kernel, d.cout, d.cin), m(d.bias)) for d in dec.prenet], "cell": cell,
          "out_w": f(dec.out_proj.kernel, dec.n_mel, H + M), "out_b": m(dec.out_proj.bias),
          "stop_w": f(dec.stop_proj.kernel, 8, dec.n_mel)[:1].contiguous(), "stop_b": m(dec.stop_proj.bias)[:1]}


def _memory(B, S, M, lens, gen):
  mem = torch.randn(B, S, M, generator=gen) * 0.5
  mask = (torch.arange(S)[None, :] < lens[:, None]).float()[:, :, None]
  return (mem * mask).to(torch.bfloat16)


def _prenet_masks(dec_out_steps, B, P, seeds, keep, dev):
  from openseq2seq_amd import capi
  T = dec_out_steps
  return [capi.dropout_mask(sd, T * B * P, keep, dev).view(T, B, P).float().cpu() / keep for sd in seeds]


def _decode(dec, mem, lens, cuda, max_steps):
  from openseq2seq_amd.parts.cnns.conv_blocks import Act
  enc = {"outputs": mem.to(cuda), "outputs_act": Act(mem.to(cuda), None, requires_grad=False),
         "src_length": lens.to(cuda)}
  out = dec.decode({"encoder_output": enc, "max_decoder_steps": max_steps})
  torch.cuda.synchronize()
  return out


def _fused_seeds():
  # Tacotron2Decoder._free_running: SeedSeq(41); _new_loop draws three seeds, then the two pre-net seeds
  from openseq2seq_amd.parts.transformer.layers import SeedSeq
  s = SeedSeq(41)
  for _ in range(3):
    s.next()
  return s.next(), s.next()


@pytest.mark.parametrize("B,S,keep", [(3, 12, 0.5), (20, 33, 1.0), (32, 40, 0.5)])
def test_fused_decode_small_vs_oracle(cuda, monkeypatch, B, S, keep):
  """Scaled-down decoder (H = M = 64, pre-net 64, 16 mel bins), pre-net dropout on (masks shared with the oracle
  through the library's counter hash) and off; B = 3 / 20 / 32 cover one and two 16-sample MFMA column tiles."""
  from openseq2seq_amd import capi
  from openseq2seq_amd.decoders import tacotron2_decoder as t2d
  from oracle import tacotron as otac
  monkeypatch.setattr(t2d, "PRENET_KEEP", keep)
  H, M, P, NM = 64, 64, 64, 16
  store, dec = _build(cuda, H, M, P, NM, 0, POST, mask_seq=False)
  g = torch.Generator().manual_seed(7)
  lens = torch.randint(S // 2, S + 1, (B,), generator=g).to(torch.int32)
  lens[0] = S
  mem = _memory(B, S, M, lens, g)
  fused_calls = []
  orig = capi.TacotronInfer.steps
  monkeypatch.setattr(capi.TacotronInfer, "steps", lambda self, a, b: (fused_calls.append((a, b)), orig(self, a, b))[1])
  T = 20
  out = _decode(dec, mem, lens, cuda, T)
  assert fused_calls and fused_calls[0][0] == 0, "the fused step kernels did not run"
  assert out["decoder_steps"] == T
  masks = _prenet_masks(T, B, P, _fused_seeds(), keep, cuda) if keep < 1.0 else None
  ref = otac.decoder_infer(_oracle_params(dec), mem.float(), lens, max_steps=T, prenet_masks=masks,
                           mask_decoder_sequence=False)
  n = 12     # the early trajectory: later steps inherit amplified rounding differences
  assert _rel(out["outputs"][0][:, :n], ref["mel"][:, :n]) <= 3e-2
  assert _rel(out["stop_token_prediction"][:, :n, 0], ref["stop"][:, :n]) <= 3e-2
  assert _rel(out["outputs"][2][:, :n], ref["align"][:, :n]) <= 3e-2
  assert _rel(out["outputs"][0], ref["mel"]) <= 1e-1
  # alignments are distributions over the live source positions
  al = out["outputs"][2].float().cpu()
  assert float((al.sum(-1) - 1).abs().max()) < 1e-3
  for b in range(B):
    assert float(al[b, :, int(lens[b]):].abs().max() if int(lens[b]) < S else 0.0) == 0.0
  assert torch.equal(out["outputs"][4].cpu(), torch.full((B,), T, dtype=torch.int32))


def test_fused_decode_stop_token_and_lengths(cuda, monkeypatch):
  """mask_decoder_sequence (tacotron_helper.py:195-226 + dynamic_decode): a sample finishes at the first step
  whose stop logit is positive (round(sigmoid)); sequence_lengths count that step; decoding ends after the step
  at which the last sample finished; frames of finished samples keep being produced until then
  (impute_finished = False). The bookkeeping is checked bit-exactly against that rule applied to the device's
  own stop logits (an independent trajectory would flip decisions whose logit sits inside the bf16 noise); the
  logits themselves are held to the oracle over the first steps, and the host's polling interval must not
  change anything."""
  from openseq2seq_amd.decoders import tacotron2_decoder as t2d
  from oracle import tacotron as otac
  monkeypatch.setattr(t2d, "PRENET_KEEP", 0.5)     # the always-on dropout makes the frames (and stop logits) fluctuate
  H, M, P, NM, B, S, T = 64, 64, 64, 16, 6, 14, 80
  store, dec = _build(cuda, H, M, P, NM, 0, POST, seed=3)
  with torch.no_grad():       # stop logits with a spread
    dec.stop_proj.kernel.master.mul_(40.0)
  store.refresh_compute_copies()
  g = torch.Generator().manual_seed(11)
  lens = torch.tensor([14, 9, 12, 7, 14, 10], dtype=torch.int32)
  mem = _memory(B, S, M, lens, g)
  # the frames do not depend on the stop projection: decode once without the mask, then shift the stop bias so
  # that every sample crosses zero somewhere in the first 40 steps (at different steps per sample)
  dec.params["mask_decoder_sequence"] = False
  probe = _decode(dec, mem, lens, cuda, 40)["stop_token_prediction"][:, :, 0].float().cpu()
  dec.params["mask_decoder_sequence"] = True
  def rule(stop):     # the reference's bookkeeping on a [B, T'] logit trajectory -> (steps or None, lengths)
    fin = torch.zeros(stop.shape[0], dtype=torch.bool)
    ln = torch.zeros(stop.shape[0], dtype=torch.int32)
    for t in range(stop.shape[1]):
      ln += (~fin).to(torch.int32)
      fin |= stop[:, t] > 0
      if bool(fin.all()):
        return t + 1, ln
    return None, ln
  # (random weights give each sample a smooth logit trajectory: a rising one crosses zero once, at a step that
  # differs between samples — the sign of the projection is chosen so that it rises)
  b0 = float(dec.stop_proj.bias.master[0])
  best = None
  for sign in (1.0, -1.0):
    traj = sign * (probe - b0) + b0
    for shift in torch.linspace(-float(traj.max()), -float(traj.min()), 400).tolist():
      st, ln = rule(traj + shift)
      if st is not None and 2 <= st <= 38 and len(set(ln.tolist())) >= 2:
        margin = float((traj[:, :st] + shift).abs().min())
        if best is None or margin > best[0]:
          best = (margin, shift, sign)
  assert best is not None, ("no stop projection gives a staggered finish", probe[:, :8])
  shift = best[1]
  if best[2] < 0:
    with torch.no_grad():
      dec.stop_proj.kernel.master.mul_(-1.0)
  with torch.no_grad():
    dec.stop_proj.bias.master.add_(shift)
  store.refresh_compute_copies()
  results = []
  for poll in (1, 4, 32):
    monkeypatch.setattr(t2d.Tacotron2Decoder, "POLL_STEPS", poll)
    out = _decode(dec, mem, lens, cuda, T)
    results.append((out["decoder_steps"], out["outputs"][4].cpu().clone(), out["outputs"][0].float().cpu().clone(),
                    out["stop_token_prediction"][:, :, 0].float().cpu().clone()))
  for r in results[1:]:
    assert r[0] == results[0][0] and all(torch.equal(x, y) for x, y in zip(r[1:], results[0][1:]))
  steps, lengths, mel, stop = results[0]
  assert mel.shape[1] == steps == stop.shape[1]
  # the rule, on the device's logits
  fin = torch.zeros(B, dtype=torch.bool)
  want_len = torch.zeros(B, dtype=torch.int32)
  want_steps = T
  for t in range(steps):
    want_len += (~fin).to(torch.int32)
    fin |= stop[:, t] > 0
    if bool(fin.all()):
      want_steps = t + 1
      break
  assert steps == want_steps and torch.equal(lengths, want_len), (steps, want_steps, lengths, want_len)
  assert 1 < steps < T, ("the case does not exercise the stop token", steps)
  assert len(set(lengths.tolist())) > 1, lengths
  masks = _prenet_masks(steps, B, P, _fused_seeds(), 0.5, cuda)
  ref = otac.decoder_infer(_oracle_params(dec), mem.float(), lens, max_steps=steps, prenet_masks=masks,
                           mask_decoder_sequence=False)
  n = min(steps, 8)
  assert _rel(stop[:, :n], ref["stop"][:, :n]) <= 3e-2 and _rel(mel[:, :n], ref["mel"][:, :n]) <= 3e-2


def test_fused_decode_matches_stepwise_path(cuda, monkeypatch):
  """The generic path (os2s_attn_decoder_fwd + GEMM launches per step, taken for shapes the fused kernels
  are not built for) and the fused path decode the same frames (pre-net dropout off: the two paths index the
  dropout hash differently)."""
  from openseq2seq_amd.decoders import tacotron2_decoder as t2d
  monkeypatch.setattr(t2d, "PRENET_KEEP", 1.0)
  H, M, P, NM, NG, B, S = 64, 128, 64, 16, 24, 5, 18
  store, dec = _build(cuda, H, M, P, NM, NG, POST, seed=5, mask_seq=False)
  g = torch.Generator().manual_seed(13)
  lens = torch.tensor([18, 11, 15, 9, 18], dtype=torch.int32)
  mem = _memory(B, S, M, lens, g)
  a = _decode(dec, mem, lens, cuda, 16)
  monkeypatch.setenv("OS2S_TACOTRON_FUSED_DECODE", "0")
  b = _decode(dec, mem, lens, cuda, 16)
  n = 10
  assert _rel(a["outputs"][0][:, :n], b["outputs"][0][:, :n]) <= 2e-2
  assert _rel(a["outputs"][2][:, :n], b["outputs"][2][:, :n]) <= 2e-2
  assert _rel(a["outputs"][1][:, :n], b["outputs"][1][:, :n]) <= 5e-2       # post-net (eval BatchNorm) on top
  assert a["outputs"][5].shape == b["outputs"][5].shape == (B, 16, NG)
  assert set(a["acts"]) == {"mel", "post", "stop", "mag"}


@pytest.mark.parametrize("fp8", [False, True])
def test_fused_decode_config_width(cuda, monkeypatch, fp8):
  """tacotron_gst.py's decoder (2 x LSTM-1024, memory 1024 = BiLSTM-256 x 2 + 512 style, attention 128 with a
  32-tap / 32-filter location layer, pre-net 2 x 256 with dropout 0.5, 80 mel bins) at the bench batch
  (B = 32, S = 200, ragged): bf16 and e4m3 recurrent weights, 10 free-running steps vs the oracle."""
  from openseq2seq_amd.decoders import tacotron2_decoder as t2d
  from oracle import tacotron as otac
  keep = 0.5
  monkeypatch.setattr(t2d, "PRENET_KEEP", keep)
  H, M, P, NM, B, S, T = 1024, 1024, 256, 80, 32, 200, 10
  store, dec = _build(cuda, H, M, P, NM, 401, POST_FULL, fp8=fp8, seed=9, mask_seq=False)
  g = torch.Generator().manual_seed(17)
  lens = torch.randint(20, S + 1, (B,), generator=g).to(torch.int32)
  lens[3] = S
  mem = _memory(B, S, M, lens, g)
  out = _decode(dec, mem, lens, cuda, T)
  assert out["decoder_steps"] == T
  masks = _prenet_masks(T, B, P, _fused_seeds(), keep, cuda)
  ref = otac.decoder_infer(_oracle_params(dec, fp8=fp8), mem.float(), lens, max_steps=T, prenet_masks=masks,
                           mask_decoder_sequence=False)
  tol = 6e-2 if fp8 else 3e-2
  r_mel, r_al = _rel(out["outputs"][0], ref["mel"]), _rel(out["outputs"][2], ref["align"])
  r_stop = _rel(out["stop_token_prediction"][:, :, 0], ref["stop"])
  print("config-width free-running decode (fp8=%s): rel-L2 mel %.2e  align %.2e  stop %.2e" % (fp8, r_mel, r_al, r_stop))
  assert r_mel <= tol and r_al <= tol and r_stop <= tol, (r_mel, r_al, r_stop)
  assert out["outputs"][5].shape == (B, T, 401) and out["outputs"][1].shape == (B, T, NM)


@pytest.mark.parametrize("t_pred,use_mask,l1", [(9, True, False), (30, True, False), (9, False, True), (30, False, False)])
def test_text2speech_loss_pads_to_common_length(cuda, t_pred, use_mask, l1):
  """losses/text2speech_loss.py:80-131: predictions shorter / longer than the target are padded with zeros,
  the spectrogram with zeros and the stop-token target with ones, then masked by sequence_mask(spec_len,
  max_length) (or averaged over everything without a mask). Loss and the gradients of all four predictions vs
  the oracle (autograd)."""
  from openseq2seq_amd.losses import Text2SpeechLoss
  from openseq2seq_amd.parts.cnns.conv_blocks import Act
  from oracle import tacotron as otac
  g = torch.Generator().manual_seed(t_pred)
  B, Tt, NM, NG = 4, 20, 16, 24
  spec = torch.randn(B, Tt, NM + NG, generator=g)
  spec_len = torch.tensor([20, 13, 7, 18], dtype=torch.int32)
  stop = (torch.arange(Tt)[None, :] >= (spec_len[:, None] - 1)).float()
  bf = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16)
  mel, post, mag, st = bf(B, t_pred, NM), bf(B, t_pred, NM), bf(B, t_pred, 24), bf(B, t_pred, 1)
  st8 = torch.zeros(B, t_pred, 8, dtype=torch.bfloat16)
  st8[:, :, :1] = st
  acts = {"mel": Act(mel.to(cuda)), "post": Act(post.to(cuda)), "mag": Act(mag.to(cuda)), "stop": Act(st8.to(cuda))}
  lossf = Text2SpeechLoss({"use_mask": use_mask, "l1_norm": l1, "mel_weight": 1.5, "mag_weight": 0.5,
                           "stop_token_weight": 2.0, "scale": 3.0, "dtype": "mixed"}, None)
  L = lossf.compute_loss({"decoder_output": {"acts": acts, "n_feats": (NM, NG)},
                          "target_tensors": [spec.to(cuda), stop.to(cuda), spec_len.to(cuda)]})
  torch.cuda.synchronize()
  leaves = {k: v.float().requires_grad_(True) for k, v in (("mel", mel), ("post", post), ("mag", mag), ("stop", st))}
  ref = otac.text2speech_loss(leaves, spec, stop, spec_len, NM, NG, l1=l1, use_mask=use_mask, mel_weight=1.5,
                              mag_weight=0.5, stop_token_weight=2.0, scale=3.0)
  ref.backward()
  assert abs(float(L.cpu()) - float(ref)) <= 2e-4 * abs(float(ref)), (float(L.cpu()), float(ref))
  for k, F in (("mel", NM), ("post", NM), ("mag", NG), ("stop", 1)):
    got = acts[k].grad.float().cpu()[:, :, :F]
    want = leaves[k].grad
    assert _rel(got, want) <= 6e-3, (k, _rel(got, want))       # gradients are stored in bf16


def test_tacotron_gst_eval_mode_loss(cuda):
  """`--mode=eval` of tacotron_gst.py: free-running decode (its own number of steps) followed by
  Text2SpeechLoss against targets of another length (losses/text2speech_loss.py:80-131). The model's eval
  loss equals the oracle loss of the same predictions, and evaluate() runs end to end."""
  from openseq2seq_amd.configs.tacotron import tacotron_gst_config
  from oracle import tacotron as otac
  model_cls, params = tacotron_gst_config(batch_size_per_gpu=4, fp8_weights=True)
  model = model_cls(params, mode="eval", hvd=None, device=cuda)
  model.compile()
  batch = model.get_data_layer().synthetic_batch(cuda, seed=5, fixed_text=24)
  enc = model.get_encoder().encode({"source_tensors": batch["source_tensors"]})
  dec = model.get_decoder().decode({"encoder_output": enc})
  L = model.get_loss_computator().compute_loss({"decoder_output": dec, "target_tensors": batch["target_tensors"],
                                                "want_grad": False})
  torch.cuda.synchronize()
  spec, stop, spec_len = [t.cpu() for t in batch["target_tensors"]]
  steps = dec["decoder_steps"]
  assert 1 <= steps <= 240 and dec["outputs"][0].shape[1] == steps
  assert steps != spec.shape[1], "the case must exercise the padding branch"
  out = {"mel": dec["outputs"][0].float().cpu(), "post": dec["outputs"][1].float().cpu(),
         "stop": dec["stop_token_prediction"].float().cpu(), "mag": dec["outputs"][5].float().cpu()}
  ref = otac.text2speech_loss(out, spec.float(), stop.float(), spec_len, 80, 401)
  assert torch.isfinite(L).all()
  assert abs(float(L.cpu()) - float(ref)) <= 2e-3 * abs(float(ref)), (float(L.cpu()), float(ref))
  res = model.evaluate(max_batches=1)
  assert res["batches"] == 1 and res["eval_loss"] == res["eval_loss"]

"""GPU parity of the global-style-token encoder (Tacotron2Encoder._embed_style) vs the CPU
fp32 oracle: conv2d stack (stride 2 in time AND frequency, incl. the transposed-convolution
data gradient), TF GRUCell summary with ragged lengths, Dense+tanh, token attention; output
and all parameter gradients. bf16 activations: output rel-L2 <= 4e-2, gradients cosine >=
0.98 and rel-L2 <= 0.2."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CONVS = [{"kernel_size": [3, 3], "stride": [2, 2], "num_channels": c, "padding": "SAME"}
         for c in (16, 16, 32)]


def _cmp(got, ref, name, cos_min=0.98, rel_max=0.2):
  got, ref = got.float().cpu().flatten(), ref.detach().float().flatten()
  cos = float(torch.nn.functional.cosine_similarity(got, ref, dim=0))
  rel = float((got - ref).norm() / (ref.norm() + 1e-12))
  return None if (cos > cos_min and rel < rel_max) else (name, round(cos, 4), round(rel, 4))


def style_oracle_params(enc, leaf):
  """fp32 leaves of a StyleEncoder's parameters in the layout oracle.gst.style_encoder takes."""
  bf = lambda t: t.to(torch.bfloat16).float()
  H, W = enc.H, enc.in_dim
  P = {"convs": [(leaf(c.kernel, bf(c.kernel.master.cpu())), leaf(c.gamma, c.gamma.master.cpu()),
                  leaf(c.beta, c.beta.master.cpu())) for c in enc.convs]}
  wgx = leaf(enc.wg_x, enc.wg_x.w16.float().cpu().view(2 * H, W))
  wgh = leaf(enc.wg_h, enc.wg_h.master.cpu())
  wcx = leaf(enc.wc_x, enc.wc_x.w16.float().cpu().view(H, W))
  wch = leaf(enc.wc_h, enc.wc_h.master.cpu())
  P["wg"] = torch.cat([wgx.t(), wgh], 0)
  P["wc"] = torch.cat([wcx.t(), wch], 0)
  P["bg"], P["bc"] = leaf(enc.bg, enc.bg.master.cpu()), leaf(enc.bc, enc.bc.master.cpu())
  d2 = lambda dn: leaf(dn.kernel, dn.kernel.w16.float().cpu().view(dn.cout, dn.cin))
  P["ref_w"], P["ref_b"] = d2(enc.ref).t(), leaf(enc.ref.bias, enc.ref.bias.master.cpu())
  P["wq"], P["wk"], P["wv"], P["wo"] = d2(enc.q).t(), d2(enc.k).t(), d2(enc.v).t(), d2(enc.o).t()
  P["att_v"] = leaf(enc.att_v, enc.att_v.master.cpu())
  P["tokens"] = enc.tokens.cpu()
  return P


def test_style_encoder_fwd_bwd(cuda):
  from openseq2seq_amd.optimizers.flat_params import FlatParams
  from openseq2seq_amd.parts.tacotron.gst import StyleEncoder
  from openseq2seq_amd.parts.cnns.conv_blocks import Tape
  from oracle import gst as ogst
  torch.manual_seed(0)
  store = FlatParams(cuda)
  params = {"conv_layers": CONVS, "num_rnn_layers": 1, "rnn_cell_dim": 32, "rnn_unidirectional": True,
            "rnn_type": "GRUCell", "emb_size": 64, "attention_layer_size": 128, "num_tokens": 10,
            "num_heads": 2}
  F = 16
  enc = StyleEncoder(store, "style", params, F, "relu", 0.1, 1e-5, 0.0)
  store.finalize()
  g = torch.Generator().manual_seed(3)
  for p in store.params:
    if p.kind == "vector" and "gamma" not in p.name and "kernel_h" not in p.name:
      p.master.add_((torch.randn(p.shape, generator=g) * 0.1).to(cuda))
  store.refresh_compute_copies()
  B, T = 4, 40
  spec = (torch.randn(B, T, F, generator=g)).to(torch.bfloat16)
  lens = torch.tensor([40, 17, 33, 9], dtype=torch.int32)
  dout = torch.randn(B, 128, generator=g).to(torch.bfloat16)
  tape = Tape()
  store.zero_grads()
  out = enc.forward(spec.to(cuda), lens.to(cuda), True, tape)
  out.grad = dout.to(cuda)
  tape.backward()
  torch.cuda.synchronize()
  # ---- oracle --------------------------------------------------------------------------
  leaves = {}

  def leaf(p, t):
    t = t.clone().requires_grad_(True)
    leaves[p.name] = t
    return t

  P = style_oracle_params(enc, leaf)
  ref = ogst.style_encoder(P, spec.float(), lens, CONVS, 2)
  (ref * dout.float()).sum().backward()
  rel = float((out.data.float().cpu() - ref.detach()).norm() / ref.detach().norm())
  assert rel < 4e-2, rel
  bad = []
  for p in store.params:
    t = leaves[p.name]
    r = _cmp(p.grad.reshape(-1), (t.grad if t.grad is not None else torch.zeros_like(t)).reshape(-1), p.name)
    if r:
      bad.append(r)
  assert not bad, bad

"""Persistent XCD-local GRU forward (csrc/rnn_xcd.hip) vs one launch per time step (csrc/rnn.hip) at the
DeepSpeech2 layer shape: equality of the outputs and time per step.
usage: bench_gru_xcd.py [B] [T] [H]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openseq2seq_amd import capi, _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 400
H = int(sys.argv[3]) if len(sys.argv) > 3 else 800
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
bf = lambda t: t.to(torch.bfloat16).to(dev)
lens = torch.randint(T // 2, T + 1, (B,), generator=g, dtype=torch.int32)
lens[0] = T
dirs = []
for d in range(2):
  dirs.append(dict(gx=bf(torch.randn(B, T, 3 * H, generator=g) * 0.5),
                   wh=bf(torch.randn(3 * H, H, generator=g) * H ** -0.5),
                   bh=(torch.randn(3 * H, generator=g) * 0.1).to(dev), reverse=bool(d)))
L = _lib.lib()
res = {}
for mode in (0, 1):
  L.os2s_gru_xcd_set_mode(mode)
  outs = capi.rnn_layer_fwd_multi(capi.CELL_GRU_CUDNN, [dict(d) for d in dirs], lens.to(dev), H)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(3):
    outs = capi.rnn_layer_fwd_multi(capi.CELL_GRU_CUDNN, [dict(d) for d in dirs], lens.to(dev), H)
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / 3
  res[mode] = outs
  print("mode %d (%s): %.3f ms per layer = %.2f us per step" % (
      mode, "persistent XCD-local" if mode else "launch per step", dt * 1e3, dt * 1e6 / T))
L.os2s_gru_xcd_set_mode(-1)
for d in range(2):
  for name, i in (("y", 0), ("gates", 1)):
    a, b = res[0][d][i].float(), res[1][d][i].float()
    print("dir %d %-5s max|diff| %.3e  rel-L2 %.3e" % (d, name, float((a - b).abs().max()),
                                                      float((a - b).norm() / (a.norm() + 1e-20))))

# backward through time
dys = [bf(torch.randn(B, T, H, generator=g)) for _ in range(2)]
bw = {}
for mode in (0, 1):
  L.os2s_gru_xcd_set_mode(mode)
  mk = lambda: [dict(whT=dirs[d]["wh"].t().contiguous(), dy=dys[d], y=res[1][d][0], gates=res[1][d][1], reverse=bool(d))
                for d in range(2)]
  args = mk()
  bw[mode] = capi.rnn_layer_bwd_multi(capi.CELL_GRU_CUDNN, args, lens.to(dev), H)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(3):
    bw[mode] = capi.rnn_layer_bwd_multi(capi.CELL_GRU_CUDNN, args, lens.to(dev), H)
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / 3
  print("backward mode %d: %.3f ms per layer = %.2f us per step" % (mode, dt * 1e3, dt * 1e6 / T))
L.os2s_gru_xcd_set_mode(-1)
for d in range(2):
  a, b = bw[0][d][0].float(), bw[1][d][0].float()
  print("dir %d dgx rel-L2 %.3e" % (d, float((a - b).norm() / (a.norm() + 1e-20))))

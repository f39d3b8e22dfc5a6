"""Debug aid: each (case, variant) of the narrow ping-pong tiles in its own subprocess (a memory fault
aborts the process), compared bit for bit with the oracle-checked 128x128 tile (variant 3)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CASES = [  # name, B, T, Cin, Cout, K, dil, ragged, mode
    ("small512", 3, 420, 256, 512, 17, 1, 1, "fwd"),
    ("small640", 2, 300, 320, 640, 21, 1, 1, "fwd"),
    ("small768", 2, 700, 128, 768, 25, 1, 1, "fwd"),
    ("c768in_small", 2, 300, 768, 256, 13, 1, 1, "fwd"),
    ("c768in_mid", 8, 500, 768, 768, 25, 1, 1, "fwd"),
    ("big768_dense", 32, 840, 768, 768, 25, 1, 0, "fwd"),
    ("big768_rag", 32, 840, 768, 768, 25, 1, 1, "fwd"),
    ("big640_rag", 32, 840, 640, 640, 21, 1, 1, "fwd"),
    ("big896_rag", 32, 840, 768, 896, 29, 2, 1, "fwd"),
    ("small512_dgrad", 3, 420, 256, 512, 17, 1, 1, "dgrad"),
    ("big640_dgrad", 32, 840, 640, 640, 21, 1, 1, "dgrad"),
    ("tile_variants_200", 3, 200, 64, 200, 3, 1, 1, "fwd"),
    ("tile_variants_768", 3, 300, 128, 768, 9, 1, 1, "fwd"),
]


def run_case(name, B, T, Cin, Cout, K, dil, ragged, mode, variant):
  import torch
  from openseq2seq_amd import capi, _lib
  dev = torch.device("cuda:0")
  g = torch.Generator().manual_seed(5)
  x = (torch.randn(B, T, Cin, generator=g)).to(torch.bfloat16).to(dev)
  w = (torch.randn(K, Cout, Cin, generator=g) * (1.0 / (K * Cin) ** 0.5)).to(torch.bfloat16).to(dev)
  lens = None
  if ragged:
    lens = torch.randint(max(1, T // 5), T + 1, (B,), generator=g).to(torch.int32)
    lens[0] = T
    lens = lens.to(dev)
  nm = capi.conv1d_num_mtiles(B, T)
  outs = {}
  for v in (3, variant):
    _lib.set_option("conv1d.variant", v)
    st = torch.full((nm, 2, Cout), float("nan"), device=dev)
    y = torch.full((B, T, Cout), 3.0, dtype=torch.bfloat16, device=dev)
    if mode == "fwd":
      capi.conv1d_fwd(x, w, dil=dil, in_len=lens, stats=st, out=y)
    else:
      capi.conv1d_fwd(x, w, dil=dil, pad_left=(K - 1) * dil // 2, tout=T, in_len=lens, out_len=lens, out=y)
    torch.cuda.synchronize()
    outs[v] = (y, st)
  _lib.set_option("conv1d.variant", -1)
  a, b = outs[3], outs[variant]
  if mode == "fwd":
    eq = bool(torch.equal(a[0], b[0])) and bool(torch.equal(a[1], b[1]))
    d = float((a[0].float() - b[0].float()).abs().max())
  else:
    live = (torch.arange(T, device=dev)[None, :] < lens[:, None])[:, :, None]
    eq = bool(torch.equal(a[0] * live, b[0] * live))
    d = float(((a[0].float() - b[0].float()) * live).abs().max())
  print("RESULT %s v%d %s maxdiff %.4g" % (name, variant, "EQUAL" if eq else "DIFF", d), flush=True)


if __name__ == "__main__":
  if len(sys.argv) > 1:
    i, v = int(sys.argv[1]), int(sys.argv[2])
    run_case(*CASES[i], v)
  else:
    for i, c in enumerate(CASES):
      for v in (12, 13):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), str(i), str(v)], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, timeout=300)
        res = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        if res:
          print(res[-1], flush=True)
        else:
          err = [l for l in r.stderr.splitlines() if "fault" in l.lower() or "error" in l.lower()]
          print("RESULT %s v%d CRASH rc=%d %s" % (c[0], v, r.returncode, err[:2]), flush=True)

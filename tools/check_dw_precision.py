import sys, torch, ctypes
import torch.nn.functional as F
sys.path.insert(0, ".")
from openseq2seq_amd import capi, _lib
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, T, C, K = 4, 700, 64, 51
x = torch.relu(torch.randn(B, T, C, generator=g)).to(torch.bfloat16)
w = torch.randn(K, C, generator=g) * 0.3
tout, pl = capi.same_padding(T, K, 1, 1)
def ref(wv):
  xp = F.pad(x.double().permute(0, 2, 1), (pl, K - 1 - pl))
  return F.conv1d(xp, wv.double().t()[:, None, :], groups=C).permute(0, 2, 1)
r32 = ref(w).float().to(torch.bfloat16)
r16 = ref(w.to(torch.bfloat16).float()).float().to(torch.bfloat16)
for variant in (1, -1):
  _lib.lib().os2s_set_option(b"depthwise.variant", ctypes.c_double(variant))
  y = capi.depthwise_conv1d_fwd(x.to(dev), w.to(dev)).cpu()
  print("variant", variant, "equal to bf16(conv with fp32 taps): %.5f   equal to bf16(conv with bf16 taps): %.5f   rel-L2 vs fp64 fp32-tap conv %.3e"
        % (float((y == r32).float().mean()), float((y == r16).float().mean()),
           float((y.double() - ref(w)).norm() / ref(w).norm())))
  # flipped taps (the data gradient's launch)
  yf = capi.depthwise_conv1d_fwd(x.to(dev), w.to(dev), pad_left=(K - 1) - pl, tout=T, flip=True).cpu()
  xp = F.pad(x.double().permute(0, 2, 1), ((K - 1) - pl, pl))
  rf = F.conv1d(xp, w.flip(0).double().t()[:, None, :], groups=C).permute(0, 2, 1)
  print("   flipped: equal to bf16(fp32-tap conv) %.5f, rel-L2 %.3e" % (float((yf == rf.float().to(torch.bfloat16)).float().mean()),
        float((yf.double() - rf).norm() / rf.norm())))

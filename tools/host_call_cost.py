import sys, time
sys.path.insert(0, ".")
import torch
from openseq2seq_amd import capi, _lib
dev = torch.device("cuda:0")
C = 256
partial = torch.zeros(224, 2, C, device=dev); gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
mm = torch.zeros(C, device=dev); mv = torch.ones(C, device=dev)
mean = torch.empty(C, device=dev); rstd = torch.empty(C, device=dev); sc = torch.empty(C, device=dev); sh = torch.empty(C, device=dev)
def loop(fn, n=3000):
  for _ in range(200): fn()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(n): fn()
  t1 = time.perf_counter()
  torch.cuda.synchronize()
  t2 = time.perf_counter()
  return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
print("bn_finalize via capi: host %.2f us/call, wall %.2f us/call" % loop(lambda: capi.bn_finalize(partial, 224 * 128, gamma, beta, 1e-3, 0.9, True, mm, mv, mean, rstd, sc, sh)))
from ctypes import c_void_p, c_int, c_float, c_longlong as c_ll
f = capi._fn("os2s_bn_finalize", None)
st = capi._stream()
args = (st, partial.data_ptr(), 224, C, 224 * 128, gamma.data_ptr(), beta.data_ptr(), 1e-3, 0.9, 1, mm.data_ptr(), mv.data_ptr(), mean.data_ptr(), rstd.data_ptr(), sc.data_ptr(), sh.data_ptr())
print("bn_finalize prebuilt ctypes args: host %.2f us/call, wall %.2f" % loop(lambda: f(*args)))
print("torch.empty(C): %.2f us" % loop(lambda: torch.empty(C, dtype=torch.float32, device=dev))[0])
print("torch.empty((32,836,256) bf16): %.2f us" % loop(lambda: torch.empty((32, 836, 256), dtype=torch.bfloat16, device=dev))[0])
x = torch.empty((4, C), device=dev)
print("unbind: %.2f us" % loop(lambda: x.unbind(0))[0])
print("_ptr: %.2f us" % loop(lambda: capi._ptr(x, torch.float32))[0])
print("_stream: %.2f us" % loop(lambda: capi._stream())[0])
print("noop lambda: %.2f us" % loop(lambda: None)[0])
abi = _lib.lib().os2s_abi_version
print("ctypes 0-arg call: %.2f us" % loop(lambda: abi())[0])

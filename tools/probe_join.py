"""How long does a stream that WAITS for another stream's event take to resume once the event fires?
(kernel trace of a Transformer-big step, round 5: the main stream finishes its backward kernels, waits for the side
stream at the end of Tape.backward, the side stream's last kernel ends — and the optimizer's first kernel starts
330 us later, with nothing running in between; 273 us in the Jasper step.)
side stream: a spin kernel of `busy_us`; main stream: wait_stream(side), then an empty kernel. Measured with events:
end of the side kernel -> end of the empty kernel, for several waiting times (how long the main stream has been
blocked before the event fires)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
main = torch.cuda.current_stream()
side = torch.cuda.Stream(device=dev)
x = torch.zeros(64, device=dev)
CLK = 100e6       # torch.cuda._sleep counts cycles of the device clock it reads; calibrated below


def spin(us, cyc_per_us):
  torch.cuda._sleep(int(us * cyc_per_us))


# calibrate _sleep
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(10_000_000); e1.record(); torch.cuda.synchronize()
cyc_per_us = 10_000_000 / (e0.elapsed_time(e1) * 1e3)
print("_sleep: %.1f cycles per us" % cyc_per_us)
for busy_us in (20, 100, 300, 1000, 3000):
  res = []
  for rep in range(12):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
      spin(busy_us, cyc_per_us)
      a.record()
    main.wait_stream(side)          # the main stream blocks here for ~busy_us
    x.add_(1.0)                     # a tiny kernel
    b.record()
    torch.cuda.synchronize()
    res.append(a.elapsed_time(b) * 1e3)
  res.sort()
  print("main stream blocked for ~%5d us: event fired -> tiny kernel done: median %.1f us (min %.1f, max %.1f)"
        % (busy_us, res[len(res) // 2], res[0], res[-1]))
# the reverse order of arrival: the event has fired long before the waiting stream reaches the wait
res = []
for rep in range(12):
  torch.cuda.synchronize()
  a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
  with torch.cuda.stream(side):
    spin(50, cyc_per_us)
    a.record()
  spin(1000, cyc_per_us)            # the main stream is busy while the side stream finishes
  c.record()
  main.wait_stream(side)
  x.add_(1.0)
  b.record()
  torch.cuda.synchronize()
  res.append(c.elapsed_time(b) * 1e3)
res.sort()
print("event fired BEFORE the wait is reached: end of previous main kernel -> tiny kernel done: median %.1f us" % res[len(res) // 2])

# ---- does the wake-up depend on what the side stream WROTE before the event, and on the event's release scope? ------
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
hipEventDisableTiming, hipEventReleaseToDevice, hipEventReleaseToSystem = 0x2, 0x40000000, 0x80000000


def make_event(flags):
  ev = ctypes.c_void_p()
  assert hip.hipEventCreateWithFlags(ctypes.byref(ev), ctypes.c_uint(flags)) == 0
  return ev


def join_with(ev, src, dst):
  assert hip.hipEventRecord(ev, ctypes.c_void_p(src.cuda_stream)) == 0
  assert hip.hipStreamWaitEvent(ctypes.c_void_p(dst.cuda_stream), ev, 0) == 0


big = torch.empty(512 * 1024 * 1024, dtype=torch.float32, device=dev)     # 2 GB
evs = {"torch wait_stream": None,
       "hip event, default flags": make_event(hipEventDisableTiming),
       "hip event, ReleaseToDevice": make_event(hipEventDisableTiming | hipEventReleaseToDevice),
       "hip event, ReleaseToSystem": make_event(hipEventDisableTiming | hipEventReleaseToSystem)}
for mb in (0, 64, 512, 2048):
  for label, ev in evs.items():
    res = []
    for rep in range(10):
      torch.cuda.synchronize()
      a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      with torch.cuda.stream(side):
        if mb:
          big[:mb * 262144].fill_(float(rep))
        spin(100, cyc_per_us)
        a.record()
      if ev is None:
        main.wait_stream(side)
      else:
        join_with(ev, side, main)
      x.add_(1.0)
      b.record()
      torch.cuda.synchronize()
      res.append(a.elapsed_time(b) * 1e3)
    res.sort()
    print("side stream wrote %4d MB, %-28s: event -> tiny kernel done: median %.1f us" % (mb, label, res[len(res) // 2]))

"""Experiment: one QuartzNet / Transformer / Jasper train step captured in a hipGraph (torch.cuda.CUDAGraph) and replayed."""
import sys, time, importlib
sys.path.insert(0, ".")
import torch
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
from openseq2seq_amd import capi
from openseq2seq_amd.parts.cnns import conv_blocks

# join only the side streams that belong to the current (capturing) stream
def join_side_streams():
  if conv_blocks._SIDE_STREAMS:
    cur = conv_blocks._current_stream_obj()
    base = capi._stream().value
    for key, st in conv_blocks._SIDE_STREAMS.items():
      if key[1] != base:
        continue
      ev = conv_blocks._JOIN_EVENT.get(st)
      if ev is None:
        ev = conv_blocks._JOIN_EVENT[st] = torch.cuda.Event()
      ev.record(st)
      cur.wait_event(ev)
conv_blocks.join_side_streams = join_side_streams

specs = {"jasper": ("openseq2seq_amd.configs.jasper", "jasper10x5_config", {"batch_size_per_gpu": 32, "use_horovod": True}),
         "quartznet": ("openseq2seq_amd.configs.quartznet", "quartznet15x5_config", {}),
         "transformer": ("openseq2seq_amd.configs.transformer", "transformer_config", {"batch_size_per_gpu": 256})}
name = sys.argv[1] if len(sys.argv) > 1 else "quartznet"
mod, fn, kw = specs[name]
cls, params = getattr(importlib.import_module(mod), fn)(**kw)
m = cls(params, mode="train", hvd=None, device=dev); m.compile()
batch = m.get_data_layer().synthetic_batch(dev, seed=1234)
for _ in range(5): m.train_step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): l0 = m.train_step(batch)
torch.cuda.synchronize()
print("%s eager: %.3f ms/step, loss %s" % (name, (time.perf_counter() - t0) / 20 * 1e3, l0.float().cpu().flatten()[:1].tolist()), flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
  for _ in range(3): m.train_step(batch)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
  loss = m.train_step(batch)
torch.cuda.synchronize()
print("captured", flush=True)
for _ in range(5): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): g.replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("%s graph replay: %.3f ms/step (host %.3f ms/step), loss %s" % (name, (t2 - t0) / 20 * 1e3, (t1 - t0) / 20 * 1e3, loss.float().cpu().flatten()[:1].tolist()), flush=True)

"""Is a train step bound by the Python thread? Per model: the host time of ONE step enqueued on an idle GPU (no queue
back-pressure), the latency of that single step, and the steady-state step time."""
import sys, time, importlib, statistics
sys.path.insert(0, ".")
import torch
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
specs = {"jasper": ("openseq2seq_amd.configs.jasper", "jasper10x5_config", {"batch_size_per_gpu": 32, "use_horovod": True}),
         "quartznet": ("openseq2seq_amd.configs.quartznet", "quartznet15x5_config", {}),
         "transformer": ("openseq2seq_amd.configs.transformer", "transformer_config", {"batch_size_per_gpu": 256}),
         "nmt": ("openseq2seq_amd.configs.nmt", "nmt_small_config", {}),
         "ds2": ("openseq2seq_amd.configs.ds2", "ds2_large_config", {}),
         "tacotron": ("openseq2seq_amd.configs.tacotron", "tacotron_gst_config", {"style": True, "fp8_weights": True})}
for name in sys.argv[1:] or list(specs):
  mod, fn, kw = specs[name]
  try:
    cls, params = getattr(importlib.import_module(mod), fn)(**kw)
    m = cls(params, mode="train", hvd=None, device=dev); m.compile()
    batch = m.get_data_layer().synthetic_batch(dev, seed=1234)
    for _ in range(6): m.train_step(batch)
    torch.cuda.synchronize()
    host, lat = [], []
    for _ in range(7):
      torch.cuda.synchronize()
      t0 = time.perf_counter(); m.train_step(batch); t1 = time.perf_counter()
      torch.cuda.synchronize(); t2 = time.perf_counter()
      host.append((t1 - t0) * 1e3); lat.append((t2 - t0) * 1e3)
    t0 = time.perf_counter()
    for _ in range(15): m.train_step(batch)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%-12s host %.2f ms to enqueue one step on an idle GPU, single-step latency %.2f ms, steady state %.2f ms/step"
          % (name, statistics.median(host), statistics.median(lat), (t2 - t0) / 15 * 1e3), flush=True)
    del m, batch
    torch.cuda.empty_cache()
  except Exception as e:
    print(name, "failed:", repr(e)[:300])

"""cProfile of the host side of the Jasper train step (where the Python / ctypes time goes)."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openseq2seq_amd.configs.jasper import jasper10x5_config
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
cls, params = jasper10x5_config(batch_size_per_gpu=32, use_horovod=False, max_steps=100000)
m = cls(params, mode="train", hvd=None, device=dev); m.compile()
batch = m.get_data_layer().synthetic_batch(dev, seed=1234)
for _ in range(5): m.train_step(batch)
torch.cuda.synchronize()
# host-only enqueue time: how long the Python thread needs per step when it never waits
t0 = time.perf_counter()
for _ in range(10): m.train_step(batch)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("enqueue %.2f ms/step, drained after %.2f ms more" % ((t1 - t0) * 100, (t2 - t1) * 1000))
pr = cProfile.Profile(); pr.enable()
for _ in range(10): m.train_step(batch)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)

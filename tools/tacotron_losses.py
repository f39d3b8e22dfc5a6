"""Per-step loss / loss-scale trace of the Tacotron2-GST bench configuration (same batch every step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openseq2seq_amd.configs.tacotron import tacotron_gst_config
dev = torch.device("cuda:0")
fp8 = os.environ.get("FP8", "1") == "1"
model_cls, params = tacotron_gst_config(style=True, fp8_weights=fp8)
model = model_cls(params, mode="train", hvd=None, device=dev)
model.compile()
batch = model.get_data_layer().synthetic_batch(dev, seed=1234)
out = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
  loss = model.train_step(batch)
  st = model.train_op.read_state()
  out.append("%.3f(s%g,k%d)" % (float(loss.cpu()[0]), st["loss_scale"], st["num_skipped"]))
print("split=%s fp8=%s:" % (os.environ.get("OS2S_ATTN_SPLIT", "1"), fp8), " ".join(out))

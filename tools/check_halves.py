"""Transformer step as two half-batches on two streams (OS2S_HALF_BATCHES): gradients against the one-batch step."""
import sys, torch
sys.path.insert(0, ".")
from openseq2seq_amd.configs.transformer import transformer_config
from openseq2seq_amd.parts.cnns.conv_blocks import Tape
dev = torch.device("cuda:0")
model_cls, params = transformer_config(batch_size_per_gpu=64)
for part in ("encoder_params", "decoder_params"):
  for k in ("attention_dropout", "relu_dropout", "layer_postprocess_dropout"):
    params[part][k] = 0.0
model = model_cls(params, mode="train", hvd=None, device=dev)
model.compile()
batch = model.get_data_layer().synthetic_batch(dev, seed=7)
store = model._store
def run(halves):
  store.zero_grads()
  if halves:
    loss = model._forward_backward_halves(batch)
  else:
    tape = Tape()
    loss = model._forward_backward(batch, tape)
    tape.backward()
  torch.cuda.synchronize()
  return float(loss.cpu()[0]), store.grads.clone()
l0, g0 = run(False)
l1, g1 = run(True)
l2, g2 = run(False)
rel = lambda a, b: float((a - b).norm() / b.norm())
print("loss full %.6f halves %.6f full again %.6f" % (l0, l1, l2))
print("grads: halves vs full rel-L2 %.3e, full vs full %.3e" % (rel(g1, g0), rel(g2, g0)))
worst = (0.0, "")
for p in store.params:
  a, b = g1[p.offset:p.offset + p.numel], g0[p.offset:p.offset + p.numel]
  worst = max(worst, (float((a - b).norm() / (b.norm() + 1e-20)), p.name))
print("worst per-parameter difference", worst)

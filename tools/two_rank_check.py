"""Two data-parallel ranks SHARING ONE GPU (gloo group, both processes on cuda:0) running the real models:
the checks the first multi-GPU run would otherwise be the first to make.
Launched by tests/test_distributed_gpu.py::test_two_ranks_share_one_gpu through torch.distributed.run.

For a small Jasper (dense-residual TDNN), a small DeepSpeech2 (conv2d + bidirectional GRU) and a small
Transformer, each rank feeding a DIFFERENT batch:
  * rank-0 broadcast (utils/hooks.py:15-55): weights AND BatchNorm moving statistics of rank 1, perturbed
    beforehand, come back equal to rank 0's;
  * 5 training steps: the fp32 master weights are BIT-equal across the ranks (same reduced gradients, same
    device-side optimizer), the losses differ (different data);
  * an Inf in rank 1's local gradient: BOTH ranks skip the step (weights untouched) and BOTH halve the loss scale
    (optimizers/mp_wrapper.py:93-95 + automatic_loss_scaler.py: the overflow is seen in the REDUCED gradient);
  * iter_size = 2 (optimizers.py:255-270): gradients accumulate locally, the all-reduce and the update run
    every second step only.
The wire is gloo because two ranks cannot open one device through RCCL; everything above the collective —
bucketing, watermark overlap on the side stream, the skip decision, the update — is the production path."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from openseq2seq_amd.utils import distributed as du


def digest(t):
  return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()


def same_on_all_ranks(tag, t):
  mine = digest(t)
  got = [None] * dist.get_world_size()
  dist.all_gather_object(got, mine)
  assert all(g == got[0] for g in got), "%s differs across ranks: %s" % (tag, got)


def small_models():
  from openseq2seq_amd.configs.jasper import jasper10x5_config
  from openseq2seq_amd.configs.ds2 import ds2_large_config
  from openseq2seq_amd.configs.transformer import transformer_config
  cls, p = jasper10x5_config(batch_size_per_gpu=4, use_horovod=True, max_steps=100)
  p["encoder_params"]["convnet_layers"] = [
      {"type": "conv1d", "repeat": 1, "kernel_size": [11], "stride": [2], "num_channels": 128, "padding": "SAME",
       "dilation": [1], "dropout_keep_prob": 0.9},
      {"type": "conv1d", "repeat": 2, "kernel_size": [11], "stride": [1], "num_channels": 128, "padding": "SAME",
       "dilation": [1], "dropout_keep_prob": 0.9, "residual": True, "residual_dense": True},
      {"type": "conv1d", "repeat": 2, "kernel_size": [13], "stride": [1], "num_channels": 384, "padding": "SAME",
       "dilation": [1], "dropout_keep_prob": 0.9, "residual": True, "residual_dense": True},
      {"type": "conv1d", "repeat": 1, "kernel_size": [1], "stride": [1], "num_channels": 256, "padding": "SAME",
       "dilation": [1], "dropout_keep_prob": 0.9},
  ]
  yield "jasper-small", cls, p
  cls, p = ds2_large_config(batch_size_per_gpu=4, max_steps=100)
  p["encoder_params"].update(num_rnn_layers=2, rnn_cell_dim=128, n_hidden=256)
  yield "ds2-small", cls, p
  cls, p = transformer_config(d_model=512, num_layers=2, num_heads=8, batch_size_per_gpu=16, vocab_size=1024,
                              max_length=24, max_steps=1000)
  yield "transformer-small", cls, p


def main():
  rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
  dev = torch.device("cuda:0")          # BOTH ranks: one shared MI355X
  torch.cuda.set_device(dev)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  hvd = du.HvdAdapter()
  for name, cls, params in small_models():
    m = cls(dict(params), mode="train", hvd=hvd, device=dev)
    m.compile()
    assert m._reducer is not None and m._reducer.world == world
    # ---- broadcast: rank 1 starts from other values, incl. the BatchNorm moving statistics ----------
    extra = m._extra_state_tensors()
    if rank != 0:
      m.store.master.add_(0.25)
      for t in extra:
        t.add_(1.5)
    du.broadcast_parameters(m.store, extra)
    same_on_all_ranks(name + ": weights after the broadcast", m.store.master)
    for i, t in enumerate(extra):
      same_on_all_ranks(name + ": state tensor %d after the broadcast" % i, t)
    # ---- 5 steps on different batches -------------------------------------------------------------------
    dl = m.get_data_layer()
    losses = []
    for step in range(5):
      batch = dl.synthetic_batch(dev, seed=100 * (rank + 1) + step)
      losses.append(float(m.train_step(batch).cpu()[0]))
    torch.cuda.synchronize()
    same_on_all_ranks(name + ": master weights after 5 steps", m.store.master)
    got = [None] * world
    dist.all_gather_object(got, losses)
    assert got[0] != got[1], "the ranks saw the same data"
    st = m.train_op.read_state()
    assert st["num_skipped"] == 0 and st["global_step"] == 5, st
    # ---- an overflow on ONE rank --------------------------------------------------------------------------
    before = m.store.master.clone()
    scale0 = float(st["loss_scale"])
    batch = dl.synthetic_batch(dev, seed=777 + rank)
    orig = m._reducer.mark_done
    if rank == 1:
      # rank 1's LOCAL gradient of the last variable becomes Inf before its bucket is reduced (an input Inf
      # would do it too, but the CTC loss masks non-finite samples, losses/ctc_loss.py:84-87)
      last = m.store.params[-1]

      def poisoned(offset, orig=orig, m=m, last=last):
        m.store.grads[last.offset] = float("inf")
        return orig(offset)
      m._reducer.mark_done = poisoned
    m.train_step(batch)
    m._reducer.mark_done = orig
    torch.cuda.synchronize()
    st = m.train_op.read_state()
    assert st["num_skipped"] == 1, (name, rank, st)
    assert float(st["loss_scale"]) == scale0 / 2, (name, rank, scale0, st["loss_scale"])
    assert torch.equal(m.store.master, before), name + ": a skipped step changed the weights"
    same_on_all_ranks(name + ": master weights after the skipped step", m.store.master)
    # ---- one more good step: the ranks move on together -------------------------------------------------
    m.train_step(dl.synthetic_batch(dev, seed=900 + rank))
    torch.cuda.synchronize()
    same_on_all_ranks(name + ": master weights after the step that follows the skip", m.store.master)
    assert not torch.equal(m.store.master, before)
    del m
    # ---- iter_size = 2 ---------------------------------------------------------------------------------------
    m = cls(dict(params, iter_size=2), mode="train", hvd=hvd, device=dev)
    m.compile()
    calls = []
    orig_ar = m._reducer._all_reduce
    m._reducer._all_reduce = lambda view, orig_ar=orig_ar: (calls.append(view.numel()), orig_ar(view))[1]
    dl = m.get_data_layer()
    w0 = m.store.master.clone()
    for step in range(4):
      m.train_step(dl.synthetic_batch(dev, seed=300 * (rank + 1) + step))
      torch.cuda.synchronize()
      if step % 2 == 0:
        assert not calls or len(calls) == (step // 2) * len(m._reducer.bounds), (step, calls)
        assert torch.equal(m.store.master, w0), name + ": an accumulation step updated the weights"
      else:
        assert len(calls) == (step // 2 + 1) * len(m._reducer.bounds), (step, len(calls))
        assert not torch.equal(m.store.master, w0)
        w0 = m.store.master.clone()
    same_on_all_ranks(name + ": master weights with iter_size 2", m.store.master)
    assert m.train_op.read_state()["global_step"] == 2
    del m
    if rank == 0:
      print("OK", name, flush=True)
  dist.barrier()
  if rank == 0:
    print("ALL OK: two ranks on one GPU", flush=True)
  dist.destroy_process_group()


if __name__ == "__main__":
  main()

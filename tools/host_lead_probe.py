"""Is the GPU waiting for the HOST at the optimizer of a train step? (profiles/r05_gpu_idle_gaps.txt: 0.27-0.33 ms of
idle GPU per step end exactly where opt_latch_scale_kernel, the optimizer's first kernel, starts — in Jasper AND in
Transformer-big.) For N steps: an event is recorded on the main stream after Tape.backward returned (everything of the
backward pass enqueued, side streams joined); right before the optimizer is enqueued the host asks whether that event
has already completed. "done" = the GPU had finished the backward pass before the host got to the optimizer: the gap is
host latency, not a dependency.
  python tools/host_lead_probe.py [jasper|transformer]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
  which = sys.argv[1] if len(sys.argv) > 1 else "jasper"
  dev = torch.device("cuda", 0)
  torch.cuda.set_device(dev)
  if which == "jasper":
    from openseq2seq_amd.configs.jasper import jasper10x5_config
    model_cls, params = jasper10x5_config(batch_size_per_gpu=32, use_horovod=True)
  else:
    from openseq2seq_amd.configs.transformer import transformer_config
    model_cls, params = transformer_config()
  model = model_cls(params, mode="train", hvd=None, device=dev)
  model.compile()
  batch = model.get_data_layer().synthetic_batch(dev, seed=1234)
  op = model.train_op
  orig_run = op.run
  stats = {"done": 0, "n": 0, "host_us_backward_to_opt": []}
  from openseq2seq_amd.parts.cnns import conv_blocks
  orig_backward = conv_blocks.Tape.backward
  marks = {}

  joins = []
  orig_join = conv_blocks.join_side_streams
  state = {"in_bwd": False, "n": 0}

  def join():
    # the join at the END of Tape.backward (the second call of a pass): an event at the tail of every side stream
    # and one on the main stream right after it has waited for them
    state["n"] += 1
    if state["in_bwd"] and state["n"] == 2:
      tails = []
      for st in conv_blocks.side_streams():
        e = torch.cuda.Event(enable_timing=True)
        e.record(st)
        tails.append(e)
      last_main = torch.cuda.Event(enable_timing=True)
      last_main.record()
      orig_join()
      after = torch.cuda.Event(enable_timing=True)
      after.record()
      joins.append((tails, last_main, after))
      return
    orig_join()
  conv_blocks.join_side_streams = join

  def backward(self):
    state["in_bwd"], state["n"] = True, 0
    r = orig_backward(self)
    state["in_bwd"] = False
    ev = torch.cuda.Event()
    ev.record()
    marks["ev"], marks["t"] = ev, time.perf_counter()
    return r

  def run():
    ev = marks.get("ev")
    if ev is not None:
      stats["n"] += 1
      stats["done"] += int(ev.query())
      stats["host_us_backward_to_opt"].append(1e6 * (time.perf_counter() - marks["t"]))
    return orig_run()

  conv_blocks.Tape.backward = backward
  op.run = run
  for _ in range(8):
    model.train_step(batch)
  torch.cuda.synchronize()
  stats.update(done=0, n=0, host_us_backward_to_opt=[])
  t0 = time.perf_counter()
  for _ in range(20):
    model.train_step(batch)
  t_host = time.perf_counter() - t0
  torch.cuda.synchronize()
  t_all = time.perf_counter() - t0
  lat = []
  for tails, last_main, after in joins[-20:]:
    # time from the LATER of (side stream tail, main stream tail) to the main stream being past the join
    cands = [t.elapsed_time(after) for t in tails] + [last_main.elapsed_time(after)]
    lat.append(1e3 * min(cands))
  lat.sort()
  if lat:
    print("join at the end of Tape.backward, last stream tail -> main stream past the wait: median %.1f us (min %.1f, max %.1f)"
          % (lat[len(lat) // 2], lat[0], lat[-1]))
  h = stats["host_us_backward_to_opt"]
  print("%s: GPU had already finished backward when the host reached the optimizer in %d of %d steps; host time "
        "between the end of Tape.backward and opt_step: %.0f us; host enqueue %.2f ms/step, wall %.2f ms/step"
        % (which, stats["done"], stats["n"], sum(h) / max(len(h), 1), 1e3 * t_host / 20, 1e3 * t_all / 20))


if __name__ == "__main__":
  main()

"""The committed bench line (profiles/*_bench_default.json = stdout of `python bench.py` on an
MI355X) carries every field of the measurement contract, with consistent arithmetic."""
import glob
import json
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest():
  files = sorted(glob.glob(os.path.join(REPO, "profiles", "*_bench_default.json")))
  assert files, "no committed bench line"
  with open(files[-1]) as f:
    return json.load(f), files[-1]


def test_bench_line_schema():
  d, path = _latest()
  for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
    assert k in d, (path, k)
  assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
  assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "bf16"
  assert "workload" in d["config"] and "model" not in d["config"]
  # value = frames of all steps / time
  frames = d["config"]["frames_per_step"]
  assert abs(d["value"] - frames / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
  r = d["roofline"]
  for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
    assert k in r
  assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
  assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] < 1.0
  assert r["traffic"] is None or r["traffic"] > 0
  c = d["cpu_baseline"]
  for k in ("value", "unit", "cores", "kind", "sample"):
    assert k in c
  assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0
  assert d["value"] > 1000 * c["value"]        # sanity: the GPU path is not the CPU path


def test_committed_pmc_traffic_is_what_bench_reports():
  d, _ = _latest()
  src = d["roofline"].get("traffic_source")
  if src is None:
    return
  with open(os.path.join(REPO, src)) as f:
    t = json.load(f)
  assert abs(t["hbm_bytes_per_launch"] - d["roofline"]["traffic"]) < 1.0


def test_bench_gpus_n_spawns_n_ranks():
  """`python bench.py --gpus 2` with no launcher around it must start 2 ranks itself (the driver's
  scaling command; the reference relies on `mpirun -np N`, run.py:43-49). Dry run of the launcher
  and of the timing collectives on CPU over gloo: the line reports n_gpus = 2, the SUM over ranks
  of the per-rank units and the MAX over ranks of the timed interval (rank 1 is the slower one)."""
  import subprocess
  import sys
  env = dict(os.environ)
  for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
    env.pop(k, None)
  r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--launcher-dry-run"],
                     stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)
  assert r.returncode == 0, r.stderr[-2000:]
  lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
  assert len(lines) == 1, r.stdout          # rank 0 only
  d = json.loads(lines[0])
  assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2"
  assert d["config"]["units_sum_over_ranks"] == 3000.0
  assert d["ms_per_step"] >= 19.0           # rank 1 sleeps 20 ms: MAX over ranks, not rank 0's 10 ms


def test_bench_refuses_a_world_size_mismatch():
  import subprocess
  import sys
  env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
  r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--launcher-dry-run"],
                     stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)
  assert r.returncode != 0 and "--gpus 2" in r.stderr


def _load_bench():
  import importlib.util
  spec = importlib.util.spec_from_file_location("os2s_bench_module", os.path.join(REPO, "bench.py"))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


def test_step_breakdown_wrappers_accept_every_keyword_of_the_entry_points_they_wrap():
  """Round 5 shipped a `rest_of_step` block that was {"error": TypeError(... 'pingpong')}: capi.conv1x1_wgrad_grouped
  had grown a keyword its bench wrapper did not pass on. Every wrapper StepBreakdown / ConvTimer install is called
  here with ALL parameters of the function it replaces (positional ones by position, keyword ones by name), on
  stand-ins with the real signatures — a signature drift fails on CPU, not in the driver's record."""
  import inspect
  import types
  import torch
  bench = _load_bench()
  from openseq2seq_amd import capi as real

  x = torch.zeros((2, 8, 4))
  lens = torch.tensor([8, 5], dtype=torch.int32)
  by_name = {"x": x, "dy": x, "dz": x, "y": x, "out": x, "dout": x, "w": torch.zeros((3, 4, 4)), "K": 3,
             "ys": [x], "scales": [x], "shifts": [x], "means": [x], "rstds": [x], "partial": x,
             "items": [{"x": x, "dy": x, "dw": torch.zeros((3, 4, 4)), "w": torch.zeros((1, 4, 4))}], "in_len": lens,
             "out_len": lens,
             "weights": torch.zeros(16), "grads": torch.zeros(16)}
  names = ("conv1d_wgrad", "conv1d_wgrad_grouped", "conv1x1_wgrad_grouped", "bn_act_fwd", "bn_act_bwd_reduce",
           "bn_bwd_apply", "opt_step", "conv1x1_fwd_grouped")
  seen = {}

  def stand_in(name):
    sig = inspect.signature(getattr(real, name))

    def f(*a, **kw):
      sig.bind(*a, **kw)                 # TypeError if the wrapper dropped or invented an argument
      seen[name] = seen.get(name, 0) + 1
      return x
    f.__signature__ = sig
    return f

  fake = types.SimpleNamespace(**{n: stand_in(n) for n in names})
  fake.same_padding = real.same_padding
  bd = bench.StepBreakdown(fake)
  bd._bracket = lambda family, work, call: call()
  bd.install()
  for n in names[:-1]:
    sig = inspect.signature(getattr(real, n))
    args, kwargs = [], {}
    for p in sig.parameters.values():
      if p.kind is p.VAR_POSITIONAL or p.kind is p.VAR_KEYWORD:
        continue
      v = by_name.get(p.name, p.default if p.default is not p.empty else 1)
      if p.kind is p.KEYWORD_ONLY or p.default is not p.empty:
        kwargs[p.name] = v
      else:
        args.append(v)
    getattr(fake, n)(*args, **kwargs)
    assert seen.get(n) == 1, n
  bd.remove()
  # ConvTimer's wrappers (round 6: conv1x1_fwd_grouped grew out_f32 and the bench died in its wrapper)
  tnames = ("conv1d_fwd", "conv1x1_fwd_grouped", "conv1x1_cat_fwd")
  seen.clear()
  fake2 = types.SimpleNamespace(**{n: stand_in(n) for n in tnames})
  fake2.same_padding = real.same_padding
  timer = bench.ConvTimer(fake2)
  timer.install()
  try:
    for n in tnames:
      sig = inspect.signature(getattr(real, n))
      args, kwargs = [], {}
      for p in sig.parameters.values():
        if p.kind is p.VAR_POSITIONAL or p.kind is p.VAR_KEYWORD:
          continue
        v = by_name.get(p.name, p.default if p.default is not p.empty else 1)
        if p.kind is p.KEYWORD_ONLY or p.default is not p.empty:
          kwargs[p.name] = v
        else:
          args.append(v)
      getattr(fake2, n)(*args, **kwargs)
      assert seen.get(n) == 1, n
  finally:
    from openseq2seq_amd.parts.cnns import conv_blocks
    if hasattr(timer, "_orig_backward"):
      conv_blocks.Tape.backward = timer._orig_backward


def test_committed_line_breakdown_has_no_error_and_headline_is_last():
  d, path = _latest()
  r = d["roofline"]
  if "rest_of_step_ok" in r:                  # lines written from round 6 on
    assert r["rest_of_step_ok"] is True, (path, r.get("rest_of_step"))
    assert isinstance(r["rest_of_step"], dict) and "error" not in r["rest_of_step"]
    assert list(d.keys())[-1] == "headline"
    h = d["headline"]
    assert h["value"] == d["value"] and h["ms_per_step"] == d["ms_per_step"]
    assert abs(h["roofline"]["frac"] - r["frac"]) < 1e-12
    assert h["secondary"]["ms_per_step"] == d["secondary"]["ms_per_step"]
    assert len(json.dumps(h)) < 2000          # fits the tail the driver keeps
    # (None since the dense-residual chains run next to the forward convolutions: no launch is alone to sample)
    assert (r["sampled_frac"] is None or r["sampled_frac"] >= r["frac"] * 0.8) and "frac_is" in r


def test_with_headline_puts_both_metrics_in_the_tail():
  bench = _load_bench()
  out = {"metric": "m", "value": 1.0, "unit": "frames/sec", "n_gpus": 1, "steps": 2, "warmup": 1, "ms_per_step": 3.0,
         "dtype": "bf16", "config": {"workload": "w"},
         "roofline": {"bound": "mfma", "achieved": 1.0, "peak": 2.0, "unit": "TFLOP/s", "frac": 0.5, "traffic": None,
                      "note": "x" * 5000},
         "secondary": {"value": 9.0, "unit": "tokens/sec", "ms_per_step": 4.0, "roofline": {"frac": 0.25}},
         "other_configs": {"tacotron_decode": {"us_per_step": 30.0}, "ds2": {"ms_per_step": 50.0},
                           "nmt": {"error": "boom"}},
         "cpu_baseline": {"value": 2.0, "unit": "frames/sec", "cores": 4, "kind": "port", "sample": "s" * 3000}}
  line = json.dumps(bench.with_headline(out))
  tail = line[-2000:]
  assert '"headline"' in tail
  h = json.loads(line)["headline"]
  assert h["secondary"] == {"value": 9.0, "unit": "tokens/sec", "ms_per_step": 4.0, "roofline_frac": 0.25}
  assert h["tacotron_decode_us_per_step"] == 30.0 and h["other_ms_per_step"] == {"ds2": 50.0}
  c = json.loads(line)["config"]
  assert c["secondary_value"] == 9.0 and c["secondary_ms_per_step"] == 4.0 and c["secondary_roofline_frac"] == 0.25
  assert c["tacotron_decode_us_per_step"] == 30.0 and c["ds2_ms_per_step"] == 50.0

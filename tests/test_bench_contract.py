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

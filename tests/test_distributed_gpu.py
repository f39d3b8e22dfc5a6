"""GPU: the data-parallel path (RCCL process group, bucketed all-reduce of the flat gradient
buffer on a side stream overlapped with backward) exercised with a one-rank group — the only
multi-process configuration a 1-GPU box allows. World size 2 is covered on CPU with gloo
(tests/test_distributed_cpu.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_single_rank_rccl_path_matches_plain_run(cuda):
  repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
  r = subprocess.run([sys.executable, os.path.join(repo, "tools", "dist_single_rank_check.py")],
                     stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=600)
  assert r.returncode == 0 and "OK: reducer active in dist run: True" in r.stdout, r.stdout[-2000:]


def test_gradients_are_final_when_their_bucket_is_reduced(cuda):
  """Every model family, overlapped reducer with 1 MB buckets on a one-rank RCCL group and
  OS2S_CHECK_REDUCER=1: no backward closure may write into a bucket after it was all-reduced
  (shared variables — tied embeddings — are the ones that could)."""
  repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29543", HSA_ENABLE_IPC_MODE_LEGACY="0")
  r = subprocess.run([sys.executable, os.path.join(repo, "tools", "reducer_finality_check.py")],
                     stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=900)
  assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-3000:]


def test_two_ranks_share_one_gpu(cuda):
  """Two gloo ranks on ONE device run real train steps of a small Jasper, DeepSpeech2 and Transformer
  (tools/two_rank_check.py): rank-0 broadcast incl. BatchNorm statistics, bit-equal master weights after 5
  steps on different batches, an Inf on rank 1 skips the step on both ranks and halves both loss scales,
  iter_size = 2 reduces and updates every second step."""
  repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
  env.pop("RANK", None)
  env.pop("WORLD_SIZE", None)
  r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                      "--master-addr", "127.0.0.1", "--master-port", "29547",
                      os.path.join(repo, "tools", "two_rank_check.py")],
                     stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=1500)
  assert r.returncode == 0 and "ALL OK: two ranks on one GPU" in r.stdout, r.stdout[-4000:]


def test_bench_gpus_2_end_to_end_on_one_device(cuda):
  """The driver's scaling command, rehearsed where only one GPU exists: `python bench.py
  --gpus 2` spawns two ranks itself; --one-device puts both on cuda:0 over gloo. Everything above the wire is what
  an 8-GPU run executes: rank-0 broadcast, bucketed all-reduce on the side stream with the watermark overlap,
  barrier + MAX-over-ranks timing, SUM of the per-rank frames, ONE JSON line from rank 0 with n_gpus 2, a `comm`
  block (per-bucket ms, exposed wait per step, bucket size and wire dtype as given on the command line) for BOTH
  headline models, and the compact headline as the last key."""
  import json
  repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
  for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "OS2S_BUCKET_MB", "OS2S_ALLREDUCE_DTYPE"):
    env.pop(k, None)
  r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--one-device", "--backend", "gloo",
                      "--steps", "3", "--warmup", "2", "--batch", "8", "--transformer-batch", "32",
                      "--no-cpu-baseline", "--bucket-mb", "64", "--allreduce-dtype", "bf16"],
                     stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=1800)
  assert r.returncode == 0, r.stderr[-4000:]
  lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
  assert len(lines) == 1, r.stdout[-2000:]
  d = json.loads(lines[0])
  assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 2 and d["scaling"] == "weak"
  assert d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 16
  assert "all-reduce" in d["config"]["workload"] and "gloo" in d["config"]["workload"]
  assert abs(d["value"] - d["config"]["frames_per_step"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
  for blk in (d["comm"], d["secondary"]["comm"]):
    assert blk["world_size"] == 2 and blk["backend"] == "gloo"
    assert blk["bucket_mb"] == 64.0 and blk["allreduce_dtype"] == "bf16"
    assert len(blk["bucket_ms"]) == blk["buckets_per_step"] >= 2 and all(t > 0 for t in blk["bucket_ms"])
    assert blk["exposed_ms_per_step"] >= 0.0 and blk["allreduce_ms_per_step"] > 0.0
  assert list(d.keys())[-1] == "headline"
  h = d["headline"]
  assert h["n_gpus"] == 2 and h["comm"]["bucket_mb"] == 64.0 and h["secondary"]["comm"]["allreduce_dtype"] == "bf16"
  assert d["config"]["comm_bucket_mb"] == 64.0 and d["config"]["secondary_ms_per_step"] == d["secondary"]["ms_per_step"]
  assert d["config"]["skipped_steps"] == 0 and d["secondary"]["skipped_steps"] == 0

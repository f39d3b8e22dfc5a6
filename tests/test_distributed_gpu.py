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

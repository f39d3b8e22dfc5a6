import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
  sys.path.insert(0, REPO)


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")


@pytest.fixture(scope="session", autouse=True)
def _built():
  """Make sure the HIP library / oracle are built (hipcc cross-compiles on CPU)."""
  import __graft_entry__ as g
  so = os.path.join(REPO, "openseq2seq_amd", "csrc", "libos2s_hip.so")
  if not os.path.exists(so) or os.path.exists("/opt/rocm/bin/hipcc"):
    g.build()
  yield


@pytest.fixture(scope="session")
def cuda():
  import torch
  if not torch.cuda.is_available():
    pytest.skip("no GPU")
  return torch.device("cuda:0")

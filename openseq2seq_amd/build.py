"""In-tree build of libos2s_hip.so (hipcc, gfx950 only) and of the CPU oracle.

No JIT cache: objects and the shared library live next to the sources so the
built `.so` travels with the repository snapshot to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libos2s_hip.so")
ORACLE_DIR = os.path.join(REPO, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "liboracle.so")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")
REFERENCE_ROOT = "/root/reference"

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HIP_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    "-Wno-unused-result",
] + os.environ.get("OS2S_EXTRA_HIPFLAGS", "").split()   # experiments (e.g. -DOS2S_PP_PRIO=1)


def _newer(target: str, deps) -> bool:
  if not os.path.exists(target):
    return True
  t = os.path.getmtime(target)
  return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd, **kw):
  r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                     text=True, **kw)
  if r.returncode != 0:
    raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout))
  return r.stdout


def hip_sources():
  return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)
                if f.endswith(".hip") or f.endswith(".cpp"))


def build_hip(force: bool = False, verbose: bool = False, only=None) -> str:
  """Compile every csrc/*.hip for gfx950 (and the host-only csrc/*.cpp) and link libos2s_hip.so."""
  if shutil.which(HIPCC) is None and not os.path.exists(HIPCC):
    raise RuntimeError("hipcc not found (looked for %s)" % HIPCC)
  headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC)
             if f.endswith(".hpp")]
  headers.append(os.path.join(REPO, "include", "os2s.h"))
  srcs = hip_sources()
  objs = []
  jobs = []
  for s in srcs:
    o = os.path.splitext(s)[0] + ".o"
    objs.append(o)
    if (force and (only is None or os.path.basename(s) in only)) or _newer(o, [s] + headers):
      jobs.append((s, o))

  def cc(job):
    s, o = job
    if verbose:
      print("[os2s build] hipcc", os.path.basename(s), flush=True)
    _run([HIPCC] + HIP_FLAGS + ["-c", s, "-o", o])

  if jobs:
    with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
      list(ex.map(cc, jobs))
  if jobs or _newer(LIB_PATH, objs):
    _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH]
         + objs + ["-Wl,-rpath,/opt/rocm/lib"])
  return LIB_PATH


def build_oracle(verbose: bool = False) -> str:
  """gcc build of the plain-C restatements under oracle/ (test infrastructure)."""
  srcs = sorted(os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR)
                if f.endswith(".c"))
  if not srcs:
    return ""
  if _newer(ORACLE_LIB, srcs):
    if verbose:
      print("[os2s build] gcc oracle", flush=True)
    _run(["gcc", "-O2", "-std=c99", "-shared", "-fPIC", "-o", ORACLE_LIB]
         + srcs + ["-lm"])
  return ORACLE_LIB


def build_reference_ctc_greedy(verbose: bool = False) -> str:
  """Compile the reference's own decoders/ctc_greedy_decoder.cpp (where it lies
  under /root/reference) into oracle/_ref/. Only possible in the build
  container; the GPU box uses the prebuilt file."""
  out = os.path.join(REF_DIR, "libref_ctc_greedy.so")
  src = os.path.join(REFERENCE_ROOT, "decoders", "ctc_greedy_decoder.cpp")
  if not os.path.exists(src):
    return out if os.path.exists(out) else ""
  shim_dir = os.path.join(ORACLE_DIR, "ref_shim")
  wrapper = os.path.join(shim_dir, "ref_ctc_greedy_capi.cpp")
  stub = os.path.join(shim_dir, "ref_decoder_utils_stub.h")
  deps = [src, wrapper, stub]
  if _newer(out, deps):
    os.makedirs(REF_DIR, exist_ok=True)
    if verbose:
      print("[os2s build] g++ reference ctc_greedy_decoder.cpp", flush=True)
    # The reference file includes "decoder_utils.h", which needs OpenFST; only
    # VALID_CHECK_EQ is used from it. Pre-define its include guard so the real
    # header is skipped and pre-include a stub that supplies the macro.
    _run(["g++", "-O2", "-std=c++11", "-shared", "-fPIC",
          "-DDECODER_UTILS_H_", "-include", stub,
          "-I", os.path.join(REFERENCE_ROOT, "decoders"), src, wrapper,
          "-o", out])
  return out


def build_all(verbose: bool = False):
  lib = build_hip(verbose=verbose)
  build_oracle(verbose=verbose)
  build_reference_ctc_greedy(verbose=verbose)
  return lib


if __name__ == "__main__":
  print(build_all(verbose=True))

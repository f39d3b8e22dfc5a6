"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every
symbol include/os2s.h declares; the product package never imports the oracle."""
import ctypes
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
  txt = open(os.path.join(REPO, "include", "os2s.h")).read()
  txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
  return sorted(set(re.findall(r"\b(os2s_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
  from openseq2seq_amd import _lib
  lib = _lib.lib()
  syms = _declared_symbols()
  assert len(syms) >= 3
  missing = [s for s in syms if not hasattr(lib, s)]
  assert not missing, missing
  assert lib.os2s_abi_version() >= 1
  assert _lib.lib().os2s_strerror(-1).decode() == "invalid argument"


def test_no_torch_types_in_header():
  txt = open(os.path.join(REPO, "include", "os2s.h")).read()
  txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)  # code only, comments stripped
  assert "torch" not in txt.lower() and "at::" not in txt and "#include <hip" not in txt


def test_product_never_imports_oracle():
  pkg = os.path.join(REPO, "openseq2seq_amd")
  bad = []
  for root, _, files in os.walk(pkg):
    for f in files:
      if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
        src = open(os.path.join(root, f), errors="ignore").read()
        if f == "build.py":
          continue  # builds the checker; does not use it
        if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or \
           "liboracle" in src or "oracle/" in src:
          bad.append(os.path.join(root, f))
  assert not bad, bad


def test_missing_library_fails_loudly(monkeypatch):
  from openseq2seq_amd import _lib
  import pytest
  monkeypatch.setattr(_lib, "_lib", None)
  monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libos2s_hip.so")
  with pytest.raises(_lib.Os2sError):
    _lib.lib()


def test_named_options_are_the_only_knobs():
  """One entry point for every test / measurement knob (os2s_set_option): each name the tests, tools and
  bench.py use is registered, each registered name is documented in the header, an unknown name is refused,
  and the library exports no per-knob setter beside it (the product controls os2s_set_deterministic and
  os2s_gru_xcd_set_mode, and the host-length hint, are entry points of their own)."""
  import glob
  import subprocess
  from openseq2seq_amd import _lib
  names = _lib.option_names()
  assert len(names) == len(set(names)) >= 16
  header = open(os.path.join(REPO, "include", "os2s.h")).read()
  used = set()
  for f in glob.glob(os.path.join(REPO, "tests", "*.py")) + glob.glob(os.path.join(REPO, "tools", "*.py")) + \
      [os.path.join(REPO, "bench.py")]:
    if os.path.basename(f) == "test_boundary.py":
      continue
    src = open(f).read()
    used |= set(re.findall(r"set_option\(b?\"([a-z0-9_.]+)\"", src))
    used |= set(re.findall(r"\"((?:conv1d|conv1x1|conv1d_wgrad|gemm_nt|depthwise)\.[a-z0-9_]+)\"", src))
  assert used and not (used - set(names)), sorted(used - set(names))
  for n in names:
    assert n in header, n
  f = _lib.bind("os2s_set_option", [ctypes.c_char_p, ctypes.c_double])
  assert f(b"no.such.option", 1.0) == -1 and f(None, 1.0) == -1
  g = _lib.bind("os2s_set_debug_stamps", [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int])
  assert g(b"conv1d", None, 0) == 0 and g(b"conv1d_wgrad", None, 0) == 0 and g(b"nope", None, 0) == -1
  out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
  setters = sorted(set(re.findall(r"\b(os2s_[a-z0-9_]*_set_(?:variant|split|debug|tiling))\b", out)))
  assert setters == [], setters

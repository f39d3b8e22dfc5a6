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

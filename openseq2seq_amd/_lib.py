"""ctypes binding of libos2s_hip.so (the C ABI declared in include/os2s.h).

There is no CPU fallback: if the HIP library is missing this raises, loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libos2s_hip.so")

_lib = None

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_int64 = ctypes.c_int64
c_size_t = ctypes.c_size_t
c_float = ctypes.c_float
c_uint64 = ctypes.c_uint64


class Os2sError(RuntimeError):
  pass


def lib():
  """Returns the loaded shared library (loads it on first use)."""
  global _lib
  if _lib is None:
    if not os.path.exists(LIB_PATH):
      raise Os2sError(
          "libos2s_hip.so not found at %s. Build it with "
          "`python -c 'import __graft_entry__ as g; g.build()'` "
          "(there is no CPU fallback for the HIP path)." % LIB_PATH)
    # PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so). Device
    # pointers and streams are only valid inside ONE runtime instance, so make
    # sure torch's copy is the one already mapped (same soname) before our
    # library, which links libamdhip64.so.7, is loaded.
    import torch  # noqa: F401
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(tlib):
      ctypes.CDLL(tlib, mode=ctypes.RTLD_GLOBAL)
    _lib = ctypes.CDLL(LIB_PATH)
    _lib.os2s_strerror.restype = ctypes.c_char_p
    _lib.os2s_strerror.argtypes = [c_int]
    _lib.os2s_abi_version.restype = c_int
    _lib.os2s_last_error_detail.restype = ctypes.c_char_p
  return _lib


def check(code, what=""):
  if code != 0:
    msg = lib().os2s_strerror(int(code)).decode()
    detail = lib().os2s_last_error_detail().decode()
    if detail:
      msg += " [" + detail + "]"
    raise Os2sError("%s failed: %s (code %d)" % (what or "os2s call", msg, code))


def bind(name, argtypes, restype=c_int):
  """Returns the C function `name` with its signature attached."""
  f = getattr(lib(), name)
  f.argtypes = argtypes
  f.restype = restype
  return f


def set_option(name, value):
  """os2s_set_option: the library's named test / measurement options (include/os2s.h lists them)."""
  f = bind("os2s_set_option", [ctypes.c_char_p, ctypes.c_double])
  if f(name.encode(), float(value)) != 0:
    raise Os2sError("os2s_set_option: unknown option %r" % (name,))


def option_names():
  f = bind("os2s_option_name", [c_int], ctypes.c_char_p)
  out, i = [], 0
  while True:
    n = f(i)
    if n is None:
      return out
    out.append(n.decode())
    i += 1


def set_debug_stamps(kernel, ptr, mode=0):
  """os2s_set_debug_stamps: device buffer (address or None) an instrumented kernel writes time stamps into."""
  f = bind("os2s_set_debug_stamps", [ctypes.c_char_p, c_void_p, c_int])
  if f(kernel.encode(), c_void_p(int(ptr) if ptr else 0), int(mode)) != 0:
    raise Os2sError("os2s_set_debug_stamps: unknown kernel %r" % (kernel,))

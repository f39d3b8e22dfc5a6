"""Shared by the tests that hold the oracle (CPU) and the HIP path (GPU) against fixtures produced by executing the
reference's own source (tests/golden/make_ref_exec.py): fixture loading, the seeded variables of the fixtures that
do not store theirs, gradient projections."""
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("make_ref_exec", os.path.join(HERE, "golden", "make_ref_exec.py"))
gen = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gen)          # module level is NumPy only; the reference is touched only by its generators


def load(name):
  d = dict(np.load(os.path.join(HERE, "golden", "ref_exec_%s.npz" % name)))
  names = [str(n) for n in d["var_names"]]
  return d, names


def variables(d, names):
  """{reference variable name: fp32 array}: stored in the fixture, or regenerated from its seed."""
  if "var/" + names[0] in d:
    return {n: d["var/" + n] for n in names}
  seed = int(d["seed"])
  return {n: gen.seeded_array(n, tuple(int(v) for v in d["shape/" + n]), seed) for n in names}


def rel(a, b):
  a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
  return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def check_gradient(d, name, g, tol):
  """g (reference layout) against the fixture: the stored tensor (rel-L2 <= tol), or its (norm, projection on a
  seeded N(0,1) direction r): |norm ratio - 1| <= tol and |<g - g_ref, r>| <= 4 tol |g_ref| (<e, r> ~ N(0, |e|^2))."""
  if "grad/" + name in d:
    r = rel(g, d["grad/" + name])
    assert r <= tol, (name, r)
    return r
  norm, proj = [float(v) for v in d["gproj/" + name]]
  gn, gp = [float(v) for v in gen.projection(name, g, int(d["seed"]))]
  assert abs(gn / norm - 1.0) <= tol, (name, gn, norm)
  assert abs(gp - proj) <= 4.0 * tol * norm, (name, gp, proj, norm)
  return abs(gp - proj) / norm

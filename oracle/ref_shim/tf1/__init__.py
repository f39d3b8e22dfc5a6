"""TEST INFRASTRUCTURE ONLY — a stand-in for the TensorFlow 1.x graph API, so that the REFERENCE'S OWN PYTHON
SOURCE (the graph-building code under /root/reference/open_seq2seq) can be executed in a container that has no
TensorFlow. Same idea as ref_decoder_utils_stub.h next to it (a stub for an absent dependency of a reference
file that is otherwise run where it lies); no reference source is copied.

What is restated here is the TensorFlow LIBRARY (tf.matmul, tf.nn.softmax, tf.layers.conv1d "SAME" padding,
tf.layers.batch_normalization, variable scopes, ...) on top of torch CPU tensors — documented primitive
semantics of TF 1.13, each small enough to read. What is NOT restated is any line of the reference: the
encoders / decoders / losses / lr policies / loss scalers run from their files. tests/golden/make_ref_exec.py
drives them on seeded inputs and commits inputs, weights and outputs as fixtures; the oracle restatements
(oracle/*.py) and the HIP path are then compared with those fixtures. That turns "parity unpinned" rows into
"pinned to the reference's own code, modulo the TF primitives restated in this file".

Execution model: every op computes its value eagerly when it is created (so static shapes exist and the
reference's Python control flow works) AND records a node (function + inputs + control dependencies), so that
Session.run can re-evaluate a fetched tensor against the CURRENT variable values and feeds — the TF1 contract
"the graph is built once and run many times" that the optimizer-side code (loss scalers, lr policies, NovoGrad)
relies on. Stateful ops (assign, assign_add) do not fire at build time, only inside Session.run.
tf.gradients is torch.autograd over the re-evaluated forward values.

Never imported by the product; installed as the module name `tensorflow` only inside the fixture generator
(tf1.install()).
"""
import contextlib
import math
import re
import sys
import types

import numpy as np
import torch

__version__ = "1.13.1-shim"
_MISSING = object()
_range, _slice, _tuple, _abs, _round, _pow = range, slice, tuple, abs, round, pow

# ----------------------------------------------------------------------------------------------- dtypes


class DType(object):
  def __init__(self, name, tdt):
    self.name, self.torch = name, tdt

  @property
  def base_dtype(self):
    return self

  @property
  def as_numpy_dtype(self):
    return {"float16": np.float16, "float32": np.float32, "float64": np.float64, "int32": np.int32,
            "int64": np.int64, "bool": np.bool_, "uint8": np.uint8}[self.name]

  @property
  def is_floating(self):
    return self.name.startswith("float")

  @property
  def is_integer(self):
    return self.name.startswith("int") or self.name == "uint8"

  @property
  def min(self):
    return torch.finfo(self.torch).min if self.is_floating else torch.iinfo(self.torch).min

  @property
  def max(self):
    return torch.finfo(self.torch).max if self.is_floating else torch.iinfo(self.torch).max

  def __eq__(self, other):
    if isinstance(other, DType):
      return self.name == other.name
    if isinstance(other, str):
      return self.name == other
    return NotImplemented

  def __ne__(self, other):
    r = self.__eq__(other)
    return r if r is NotImplemented else not r

  def __hash__(self):
    return hash(self.name)

  def __repr__(self):
    return "tf." + self.name


float16 = half = DType("float16", torch.float16)
float32 = DType("float32", torch.float32)
float64 = double = DType("float64", torch.float64)
int32 = DType("int32", torch.int32)
int64 = DType("int64", torch.int64)
uint8 = DType("uint8", torch.uint8)
bool = DType("bool", torch.bool)          # noqa: A001  (the TF name)
string = DType("string", None)
_BY_TORCH = {d.torch: d for d in (float16, float32, float64, int32, int64, uint8, bool)}
_PYBOOL = type(True)


def as_dtype(d):
  if isinstance(d, DType):
    return d
  if isinstance(d, torch.dtype):
    return _BY_TORCH[d]
  if isinstance(d, str):
    return {"float16": float16, "float32": float32, "float64": float64, "int32": int32, "int64": int64,
            "bool": bool}[d]
  return _BY_TORCH[torch.from_numpy(np.zeros(1, dtype=d)).dtype]


# ----------------------------------------------------------------------------------------------- shapes


class Dimension(int):
  @property
  def value(self):
    return int(self)


class TensorShape(object):
  def __init__(self, dims):
    if isinstance(dims, TensorShape):
      dims = dims._dims
    elif dims is not None and not hasattr(dims, "__iter__"):
      dims = [dims]                               # TensorShape(5) is the shape [5]
    self._dims = None if dims is None else [None if d is None else Dimension(d) for d in dims]

  @property
  def dims(self):
    return self._dims

  @property
  def ndims(self):
    return None if self._dims is None else len(self._dims)

  def as_list(self):
    return [None if d is None else int(d) for d in self._dims]

  def is_fully_defined(self):
    return self._dims is not None and all(d is not None for d in self._dims)

  def is_compatible_with(self, other):
    other = other if isinstance(other, TensorShape) else TensorShape(other)
    if self._dims is None or other._dims is None:
      return True
    return len(self._dims) == len(other._dims) and all(a is None or b is None or int(a) == int(b)
                                                       for a, b in zip(self._dims, other._dims))

  def assert_is_compatible_with(self, other):
    if not self.is_compatible_with(other):
      raise ValueError("Shapes %s and %s are incompatible" % (self, other))

  def merge_with(self, other):
    self.assert_is_compatible_with(other)
    return self

  def concatenate(self, other):
    other = other if isinstance(other, TensorShape) else TensorShape(other)
    return TensorShape(list(self._dims) + list(other._dims))

  def with_rank_at_least(self, rank):
    return self

  def __len__(self):
    return len(self._dims)

  def __iter__(self):
    return iter(self._dims)

  def __getitem__(self, i):
    if isinstance(i, _slice):
      return TensorShape(self._dims[i])
    return self._dims[i]

  def __bool__(self):
    return self._dims is not None

  def __eq__(self, other):
    return list(self.as_list()) == list(TensorShape(other).as_list() if not isinstance(other, TensorShape)
                                         else other.as_list())

  def __repr__(self):
    return "TensorShape(%r)" % (self._dims,)


# ----------------------------------------------------------------------------------------------- graph

_CTRL = []          # stack of control-dependency lists
_NAME = []          # name-scope stack (names only; nothing depends on them)


def _walk(obj, fn):
  """obj with every Tensor replaced by fn(tensor) (lists / tuples / dicts are walked)."""
  if isinstance(obj, Tensor):
    return fn(obj)
  if isinstance(obj, (list, _tuple)):
    r = [_walk(o, fn) for o in obj]
    return type(obj)(r) if type(obj) in (list, _tuple) else r
  if isinstance(obj, dict):
    return {k: _walk(v, fn) for k, v in obj.items()}
  if isinstance(obj, _slice):
    return _slice(_walk(obj.start, fn), _walk(obj.stop, fn), _walk(obj.step, fn))
  return obj


class Tensor(object):
  """A graph node with its build-time value."""
  _count = 0

  def __init__(self, fn, args=(), kwargs=None, name=None, stateful=False, build_value=_MISSING):
    self._fn, self._args, self._kwargs = fn, args, (kwargs or {})
    self._ctrl = [c for lst in _CTRL for c in lst]
    self._stateful = stateful
    Tensor._count += 1
    self._id = Tensor._count
    self.name = ("/".join(_NAME + [name or "op"])) + "_%d:0" % self._id
    self._alt = None
    if build_value is not _MISSING:
      self._value = build_value
    else:
      self._value = self._compute(lambda t: t._value)
      self._shadow()

  def _shadow(self):
    """Static-shape inference for loop bodies: a loop variable whose shape invariant leaves a dimension open carries
    a second build value that is one longer there (`_alt`); every op evaluated on such inputs is evaluated on the
    shadow values too, and a dimension in which the two results differ is reported as unknown (None) by .shape — as
    TensorFlow reports it inside tf.while_loop, which is what makes the reference fall back to tf.shape()."""
    found = []
    _walk((self._args, self._kwargs), lambda t: found.append(t) or t)
    if not any(t._alt is not None for t in found):
      return
    try:
      alt = self._compute(lambda t: t._alt if t._alt is not None else t._value)
    except Exception:      # the shadow is advisory: an op that cannot take the longer operand keeps a static shape
      return
    if isinstance(alt, torch.Tensor) and isinstance(self._value, torch.Tensor):
      self._alt = alt

  def _compute(self, get):
    return self._fn(*_walk(self._args, get), **_walk(self._kwargs, get))

  def _eval(self, ev):                   # ev: Tensor -> value inside one Session.run
    for c in self._ctrl:
      ev(c)
    return self._compute(ev)

  # --- tf.Tensor surface
  @property
  def dtype(self):
    return _BY_TORCH[self._value.dtype]

  @property
  def shape(self):
    st = getattr(self, "_static_shape", None)
    if st is not None:
      return st
    dims = list(self._value.shape)
    alt = getattr(self, "_alt", None)
    if alt is not None:
      if alt.dim() != len(dims):
        return TensorShape(None)
      dims = [d if d == a else None for d, a in zip(dims, alt.shape)]
    return TensorShape(dims)

  def get_shape(self):
    return self.shape

  def set_shape(self, shape):
    pass

  @property
  def op(self):
    return self

  @property
  def graph(self):
    return _GRAPH

  @property
  def device(self):
    return ""

  def eval(self, feed_dict=None, session=None):
    return Session().run(self, feed_dict)

  def numpy(self):
    return self._value.detach().cpu().numpy()

  def __hash__(self):
    return id(self)

  def __eq__(self, other):          # TF1: identity
    return self is other

  def __ne__(self, other):
    return self is not other

  def __bool__(self):
    raise TypeError("Using a tf.Tensor as a Python bool is not allowed (graph mode).")

  def __len__(self):
    return int(self._value.shape[0])

  def __iter__(self):
    return (self[i] for i in _range(int(self._value.shape[0])))

  def __repr__(self):
    return "<tf1.Tensor %s shape=%s dtype=%s>" % (self.name, _tuple(self._value.shape), self.dtype.name)

  def __getitem__(self, idx):
    def f(x, i):
      def one(j):
        if isinstance(j, torch.Tensor):
          return int(j) if j.dim() == 0 else j.long()
        return j
      if isinstance(i, _tuple):
        i = _tuple(_slice(one(j.start), one(j.stop), one(j.step)) if isinstance(j, _slice) else one(j) for j in i)
      elif isinstance(i, _slice):
        i = _slice(one(i.start), one(i.stop), one(i.step))
      else:
        i = one(i)
      return x[i]
    return Tensor(f, (self, idx), name="strided_slice")

  __array_priority__ = 100
  __array_ufunc__ = None          # numpy scalars / arrays defer to the reflected operators above


def _binop(name, f):
  def op(a, b):
    return Tensor(lambda x, y: f(_t(x, like=y), _t(y, like=x)), (a, b), name=name)
  return op


def _t(x, like=None, dtype=None):
  """torch value of a Python / numpy / torch operand (Python scalars adopt `like`'s dtype, as in TF)."""
  if isinstance(x, torch.Tensor):
    return x if dtype is None else x.to(dtype)
  if isinstance(x, DType):
    raise TypeError("dtype passed where a tensor was expected")
  if dtype is None and isinstance(like, torch.Tensor) and isinstance(x, (int, float, _PYBOOL)):
    if isinstance(x, float) and not like.dtype.is_floating_point:
      dtype = torch.float32
    else:
      dtype = like.dtype
  if dtype is None:
    if isinstance(x, _PYBOOL):
      dtype = torch.bool
    elif isinstance(x, int):
      dtype = torch.int32
    elif isinstance(x, float):
      dtype = torch.float32
    elif isinstance(x, np.ndarray):
      return torch.from_numpy(np.ascontiguousarray(x))
    elif isinstance(x, (list, _tuple)):
      flat = _flatten(x)
      if any(isinstance(v, torch.Tensor) for v in flat):
        return torch.stack([_t(v) for v in x])
      if all(isinstance(v, (int, _PYBOOL, np.integer)) for v in flat):
        dtype = torch.int32 if not all(isinstance(v, _PYBOOL) for v in flat) or not flat else torch.bool
      else:
        dtype = torch.float32
    elif isinstance(x, np.generic):
      return torch.from_numpy(np.asarray(x))
  if isinstance(x, (list, _tuple)) and any(isinstance(v, torch.Tensor) for v in _flatten(x)):
    return torch.stack([_t(v, dtype=dtype) for v in x])
  return torch.tensor(x, dtype=dtype)


def _flatten(x):
  if isinstance(x, (list, _tuple)):
    out = []
    for v in x:
      out.extend(_flatten(v))
    return out
  return [x]


def _tdiv(x, y):
  if x.dtype.is_floating_point or y.dtype.is_floating_point:
    return x / y
  return (x.double() / y.double())


def _pydiv(x, y):        # tf.div / Python-2 style: floor for integers, true division for floats
  if x.dtype.is_floating_point or y.dtype.is_floating_point:
    return x / y
  return torch.div(x, y, rounding_mode="floor")


for _n, _f in (("add", lambda x, y: x + y), ("sub", lambda x, y: x - y), ("mul", lambda x, y: x * y),
               ("truediv", _tdiv), ("floordiv", lambda x, y: torch.div(x, y, rounding_mode="floor")),
               ("mod", lambda x, y: torch.remainder(x, y)), ("pow", lambda x, y: torch.pow(x, y)),
               ("lt", lambda x, y: x < y), ("le", lambda x, y: x <= y), ("gt", lambda x, y: x > y),
               ("ge", lambda x, y: x >= y), ("and", lambda x, y: x & y), ("or", lambda x, y: x | y),
               ("matmul", lambda x, y: x @ y)):
  _o = _binop(_n, _f)
  setattr(Tensor, "__%s__" % _n, _o)
  if _n not in ("lt", "le", "gt", "ge"):
    setattr(Tensor, "__r%s__" % _n, (lambda o: (lambda a, b: o(b, a)))(_o))
Tensor.__div__ = Tensor.__truediv__
Tensor.__rdiv__ = Tensor.__rtruediv__
Tensor.__neg__ = lambda a: Tensor(lambda x: -x, (a,), name="neg")
Tensor.__abs__ = lambda a: Tensor(lambda x: x.abs(), (a,), name="abs")
Tensor.__invert__ = lambda a: Tensor(lambda x: ~x, (a,), name="not")


def _op(fn, *args, **kwargs):
  name = kwargs.pop("_name", None)
  return Tensor(fn, args, kwargs, name=name)


def convert_to_tensor(value, dtype=None, name=None, preferred_dtype=None):
  if isinstance(value, Tensor) and dtype is None:
    return value
  td = as_dtype(dtype).torch if dtype is not None else None
  return Tensor(lambda v: _t(v, dtype=td), (value,), name=name or "const")


_OP_NAMES = {}


def _unique_op_name(base):
  """TensorFlow's op naming inside the current name scope: base, base_1, base_2 ... (mp_wrapper_test.py:91 asserts
  the name of a constant; only constants — and the reduction-axes constants reduce_* create — are named this way
  here, every other node keeps its internal name)."""
  key = "/".join(_NAME + [base])
  n = _OP_NAMES.get(key, 0)
  _OP_NAMES[key] = n + 1
  return key if n == 0 else "%s_%d" % (key, n)


def constant(value, dtype=None, shape=None, name="Const", verify_shape=False):
  if shape is None:
    t = convert_to_tensor(value, dtype, name)
  else:
    t = Tensor(lambda v: _t(v, dtype=as_dtype(dtype).torch if dtype else None).expand(*_ishape(shape)).clone(),
               (value,), name=name)
  t.name = _unique_op_name(name or "Const") + ":0"
  return t


def _ishape(shape):
  """Python ints of a shape argument whose entries may be ints, 0-d torch values or one 1-d torch value."""
  if isinstance(shape, torch.Tensor):
    return [int(v) for v in shape.reshape(-1).tolist()]
  if isinstance(shape, TensorShape):
    return shape.as_list()
  if isinstance(shape, (int, np.integer)):
    return [int(shape)]
  return [int(v) for v in shape]


class Graph(object):
  def __init__(self):
    self.collections = {}
    self.seed = 0

  @contextlib.contextmanager
  def as_default(self):
    """A fresh tf.Graph() made the default: the stand-in keeps ONE graph's state, so entering starts it over."""
    reset_default_graph()
    yield self

  def get_collection(self, name, scope=None):
    return get_collection(name, scope)


_GRAPH = Graph()


def get_default_graph():
  return _GRAPH


class GraphKeys(object):
  GLOBAL_VARIABLES = "variables"
  TRAINABLE_VARIABLES = "trainable_variables"
  UPDATE_OPS = "update_ops"
  REGULARIZATION_LOSSES = "regularization_losses"
  LOSSES = "losses"
  GLOBAL_STEP = "global_step"
  LOCAL_VARIABLES = "local_variables"
  SUMMARIES = "summaries"


def add_to_collection(name, value):
  _GRAPH.collections.setdefault(name, []).append(value)


def add_to_collections(names, value):
  for n in ([names] if isinstance(names, str) else names):
    add_to_collection(n, value)


def get_collection(name, scope=None):
  items = list(_GRAPH.collections.get(name, []))
  if scope:
    items = [v for v in items if getattr(v, "name", "").startswith(scope)]
  return items


def get_collection_ref(name):
  return _GRAPH.collections.setdefault(name, [])


# ----------------------------------------------------------------------------------------------- variables

_RNG = torch.Generator().manual_seed(0)


def set_random_seed(seed):
  _GRAPH.seed = int(seed)
  _RNG.manual_seed(int(seed))


class Variable(Tensor):
  def __init__(self, initial_value=None, trainable=True, collections=None, validate_shape=True, name=None,
               dtype=None, expected_shape=None, **unused):
    if callable(initial_value) and not isinstance(initial_value, Tensor):
      initial_value = initial_value()
    if isinstance(initial_value, Tensor):
      v = initial_value._value.detach().clone()
    else:
      v = _t(initial_value)
    if dtype is not None:
      v = v.to(as_dtype(dtype).torch)
    self._var = v
    self.trainable = _PYBOOL(trainable)
    if v.dtype.is_floating_point:          # tf.gradients may be taken w.r.t. any variable (mp_wrapper.py:79-88 does,
      self._var.requires_grad_(True)       # w.r.t. its non-trainable fp32 master copies)
    super(Variable, self).__init__(lambda: self._var, (), name=name, build_value=self._var)
    self._ctrl = []
    full = "/".join([s for s in (_scope().name, name or "Variable") if s])
    self.name = _unique_var_name(full) + ":0"
    add_to_collection(GraphKeys.GLOBAL_VARIABLES, self)
    if self.trainable:
      add_to_collection(GraphKeys.TRAINABLE_VARIABLES, self)
    for c in (collections or []):
      if c not in (GraphKeys.GLOBAL_VARIABLES, GraphKeys.TRAINABLE_VARIABLES):
        add_to_collection(c, self)

  # the build-time value of a variable IS its current value
  @property
  def _value(self):
    return self._var

  @_value.setter
  def _value(self, v):
    pass

  def _eval(self, ev):
    return self._var

  def _set(self, v):
    v = v.detach().to(self._var.dtype).clone()
    if _tuple(v.shape) != _tuple(self._var.shape):
      v = v.reshape(self._var.shape)
    if v.dtype.is_floating_point:
      v.requires_grad_(True)
    self._var = v
    return v

  def assign(self, value, use_locking=None, name=None, read_value=True):
    return assign(self, value)

  def assign_add(self, delta, use_locking=None, name=None, read_value=True):
    return assign_add(self, delta)

  def assign_sub(self, delta, use_locking=None, name=None, read_value=True):
    return assign_sub(self, delta)

  def initialized_value(self):
    return self

  def read_value(self):
    return identity(self)

  def value(self):
    return self

  @property
  def initializer(self):
    return no_op()

  def load(self, value, session=None):
    self._set(_t(value))

  __hash__ = Tensor.__hash__


_VAR_NAMES = {}


def _unique_var_name(full):
  n = _VAR_NAMES.get(full, 0)
  _VAR_NAMES[full] = n + 1
  return full if n == 0 else "%s_%d" % (full, n)


def assign(ref, value, validate_shape=None, use_locking=None, name=None):
  return Tensor(lambda v: ref._set(_t(v)), (value,), name="assign", stateful=True,
                build_value=_t(value._value if isinstance(value, Tensor) else value).to(ref._var.dtype))


def assign_add(ref, value, use_locking=None, name=None):
  return Tensor(lambda v: ref._set(ref._var.detach() + _t(v, like=ref._var)), (value,), name="assign_add",
                stateful=True, build_value=ref._var.detach())


def assign_sub(ref, value, use_locking=None, name=None):
  return Tensor(lambda v: ref._set(ref._var.detach() - _t(v, like=ref._var)), (value,), name="assign_sub",
                stateful=True, build_value=ref._var.detach())


AUTO_REUSE = "AUTO_REUSE"


class VariableScope(object):
  def __init__(self, name, parent=None, reuse=None, initializer=None, dtype=None, regularizer=None,
               custom_getter=None):
    self.name = name
    self.reuse = reuse if reuse is not None else (parent.reuse if parent else None)
    self.initializer = initializer if initializer is not None else (parent.initializer if parent else None)
    self.dtype = dtype if dtype is not None else (parent.dtype if parent else float32)
    self.regularizer = regularizer if regularizer is not None else (parent.regularizer if parent else None)
    self.custom_getter = custom_getter if custom_getter is not None else (parent.custom_getter if parent else None)
    self.original_name_scope = name + "/" if name else ""

  def reuse_variables(self):
    self.reuse = True

  def set_initializer(self, i):
    self.initializer = i

  def set_regularizer(self, r):
    self.regularizer = r

  def set_custom_getter(self, g):
    self.custom_getter = g


_SCOPES = [VariableScope("")]
_DEFAULT_NAMES = {}
_VARS = {}


def _scope():
  return _SCOPES[-1]


def get_variable_scope():
  return _scope()


def _unique_scope_name(parent, default_name):
  key = (parent, default_name)
  n = _DEFAULT_NAMES.get(key, 0)
  _DEFAULT_NAMES[key] = n + 1
  return default_name if n == 0 else "%s_%d" % (default_name, n)


@contextlib.contextmanager
def variable_scope(name_or_scope=None, default_name=None, values=None, initializer=None, regularizer=None,
                   caching_device=None, partitioner=None, custom_getter=None, reuse=None, dtype=None,
                   use_resource=None, constraint=None, auxiliary_name_scope=True):
  parent = _scope()
  if isinstance(name_or_scope, VariableScope):
    sc = VariableScope(name_or_scope.name, name_or_scope, reuse, initializer, dtype, regularizer, custom_getter)
  else:
    if name_or_scope is None:
      name_or_scope = _unique_scope_name(parent.name, default_name)
    full = "/".join([s for s in (parent.name, name_or_scope) if s])
    sc = VariableScope(full, parent, reuse, initializer, dtype, regularizer, custom_getter)
  _SCOPES.append(sc)
  _NAME.append(sc.name.split("/")[-1] if sc.name else "")
  try:
    yield sc
  finally:
    _NAME.pop()
    _SCOPES.pop()


@contextlib.contextmanager
def name_scope(name=None, default_name=None, values=None):
  _NAME.append(name or default_name or "scope")
  try:
    yield "/".join(_NAME) + "/"
  finally:
    _NAME.pop()


def _default_initializer(dtype):
  return glorot_uniform_initializer() if dtype.is_floating else zeros_initializer()


def get_variable(name, shape=None, dtype=None, initializer=None, regularizer=None, trainable=True,
                 collections=None, caching_device=None, partitioner=None, validate_shape=True,
                 use_resource=None, custom_getter=None, constraint=None, **unused):
  sc = _scope()
  getter = custom_getter or sc.custom_getter
  kw = dict(shape=shape, dtype=dtype, initializer=initializer, regularizer=regularizer, trainable=trainable,
            collections=collections)
  if getter is not None:
    return getter(_true_get_variable, name, **kw)
  return _true_get_variable(name, **kw)


def _true_get_variable(name, shape=None, dtype=None, initializer=None, regularizer=None, trainable=True,
                       collections=None, **unused):
  sc = _scope()
  full = "/".join([s for s in (sc.name, name) if s])
  if full in _VARS:
    if sc.reuse is None or sc.reuse is False:
      # TF would raise; the reference relies on AUTO_REUSE / explicit reuse only where it sets it
      raise ValueError("Variable %s already exists, disallowed (reuse not set)" % full)
    return _VARS[full]
  if sc.reuse is True:
    raise ValueError("Variable %s does not exist (reuse=True)" % full)
  dt = as_dtype(dtype) if dtype is not None else sc.dtype
  init = initializer if initializer is not None else sc.initializer
  if init is None:
    init = _default_initializer(dt)
  if isinstance(init, Tensor):
    val = init._value.detach().clone().to(dt.torch)
  elif isinstance(init, (np.ndarray, list, float, int)):
    val = _t(init, dtype=dt.torch)
  else:
    if isinstance(init, type):
      init = init()
    val = init(_ishape(shape) if shape is not None else [], dtype=dt)
    val = val._value if isinstance(val, Tensor) else _t(val)
    val = val.detach().to(dt.torch)
  saved = list(_SCOPES)
  _SCOPES[:] = [VariableScope("")]          # Variable() prefixes the scope name itself: give it the full name
  try:
    v = Variable(val, trainable=trainable, name=full, collections=collections)
  finally:
    _SCOPES[:] = saved
  _VARS[full] = v
  reg = regularizer if regularizer is not None else sc.regularizer
  if reg is not None:
    with name_scope(name + "/Regularizer"):
      loss = reg(v)
    if loss is not None:
      add_to_collection(GraphKeys.REGULARIZATION_LOSSES, loss)
  return v


def trainable_variables(scope=None):
  return get_collection(GraphKeys.TRAINABLE_VARIABLES, scope)


def global_variables(scope=None):
  return get_collection(GraphKeys.GLOBAL_VARIABLES, scope)


def global_variables_initializer():
  return no_op()


def local_variables_initializer():
  return no_op()


def variables_initializer(var_list, name=None):
  return no_op()


# --- initializers (callables: (shape, dtype, partition_info) -> value)
def _fans(shape):
  shape = list(shape)
  if len(shape) < 1:
    return 1.0, 1.0
  if len(shape) == 1:
    return float(shape[0]), float(shape[0])
  rf = 1.0
  for d in shape[:-2]:
    rf *= d
  return float(shape[-2] * rf), float(shape[-1] * rf)


class _Init(object):
  def __call__(self, shape, dtype=None, partition_info=None):
    raise NotImplementedError

  def get_config(self):
    return {}


class zeros_initializer(_Init):
  def __init__(self, dtype=None):
    pass

  def __call__(self, shape, dtype=None, partition_info=None):
    return torch.zeros(_ishape(shape), dtype=as_dtype(dtype or float32).torch)


class ones_initializer(_Init):
  def __init__(self, dtype=None):
    pass

  def __call__(self, shape, dtype=None, partition_info=None):
    return torch.ones(_ishape(shape), dtype=as_dtype(dtype or float32).torch)


class constant_initializer(_Init):
  def __init__(self, value=0, dtype=None, verify_shape=False):
    self.value = value

  def __call__(self, shape, dtype=None, partition_info=None):
    v = torch.as_tensor(np.asarray(self.value, dtype=np.float64))
    return (v.expand(*_ishape(shape)) if v.dim() == 0 else v.reshape(_ishape(shape))).clone().to(
        as_dtype(dtype or float32).torch)


class random_normal_initializer(_Init):
  def __init__(self, mean=0.0, stddev=1.0, seed=None, dtype=None):
    self.mean, self.stddev = mean, stddev

  def __call__(self, shape, dtype=None, partition_info=None):
    return (torch.randn(_ishape(shape), generator=_RNG, dtype=torch.float64) * self.stddev + self.mean).to(
        as_dtype(dtype or float32).torch)


truncated_normal_initializer = random_normal_initializer      # distribution detail: irrelevant to parity runs


class random_uniform_initializer(_Init):
  def __init__(self, minval=0.0, maxval=None, seed=None, dtype=None):
    self.lo, self.hi = minval, (1.0 if maxval is None else maxval)

  def __call__(self, shape, dtype=None, partition_info=None):
    return (torch.rand(_ishape(shape), generator=_RNG, dtype=torch.float64) * (self.hi - self.lo) + self.lo).to(
        as_dtype(dtype or float32).torch)


class variance_scaling_initializer(_Init):
  def __init__(self, scale=1.0, mode="fan_in", distribution="truncated_normal", seed=None, dtype=None):
    self.scale, self.mode, self.dist = scale, mode.lower(), distribution

  def __call__(self, shape, dtype=None, partition_info=None):
    fi, fo = _fans(_ishape(shape))
    n = {"fan_in": fi, "fan_out": fo, "fan_avg": (fi + fo) / 2.0}[self.mode]
    s = self.scale / max(1.0, n)
    if self.dist == "uniform":
      lim = math.sqrt(3.0 * s)
      r = (torch.rand(_ishape(shape), generator=_RNG, dtype=torch.float64) * 2 - 1) * lim
    else:
      r = torch.randn(_ishape(shape), generator=_RNG, dtype=torch.float64) * math.sqrt(s)
    return r.to(as_dtype(dtype or float32).torch)


def glorot_uniform_initializer(seed=None, dtype=None):
  return variance_scaling_initializer(1.0, "fan_avg", "uniform")


def glorot_normal_initializer(seed=None, dtype=None):
  return variance_scaling_initializer(1.0, "fan_avg", "normal")


# ----------------------------------------------------------------------------------------------- session


class Session(object):
  def __init__(self, target="", graph=None, config=None):
    pass

  def __enter__(self):
    return self

  def __exit__(self, *a):
    return False

  def close(self):
    pass

  def run(self, fetches, feed_dict=None, options=None, run_metadata=None):
    memo = {}
    for k, v in (feed_dict or {}).items():
      memo[id(k)] = _t(v).to(k._value.dtype) if isinstance(k, Tensor) else v

    def ev(t):
      if isinstance(t, Variable):       # a ref variable is read where it is used: after an assign that the
        return t._var                   # reader's control dependencies ran first, it shows the new value
      key = id(t)
      if key not in memo:
        memo[key] = t._eval(ev)
      return memo[key]

    def out(t):
      if t is None:
        return None
      if isinstance(t, Tensor):
        v = ev(t)
        return v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v
      if isinstance(t, IndexedSlices):
        return out(t.values)
      return t

    old = sys.getrecursionlimit()
    sys.setrecursionlimit(max(old, 100000))
    try:
      return _walk_fetch(fetches, out)
    finally:
      sys.setrecursionlimit(old)


def _walk_fetch(obj, fn):
  if isinstance(obj, (list, _tuple)):
    return type(obj)(_walk_fetch(o, fn) for o in obj) if type(obj) in (list, _tuple) else [_walk_fetch(o, fn) for o in obj]
  if isinstance(obj, dict):
    return {k: _walk_fetch(v, fn) for k, v in obj.items()}
  return fn(obj)


InteractiveSession = Session


class ConfigProto(object):
  def __init__(self, **kw):
    self.gpu_options = types.SimpleNamespace(allow_growth=False, visible_device_list="")


def placeholder(dtype, shape=None, name=None):
  shp = [1 if d is None else int(d) for d in (shape or [])]
  t = Tensor(lambda: torch.zeros(shp, dtype=as_dtype(dtype).torch), (), name=name or "placeholder")
  t._fed = True
  return t


def constant_value(tensor, partial=False):
  """tensor_util.constant_value: the NumPy value of a node that depends on no variable, placeholder, loop variable or
  stateful op; None otherwise."""
  if not isinstance(tensor, Tensor):
    return np.asarray(tensor)
  seen = {}

  def const(t):
    if id(t) in seen:
      return seen[id(t)]
    seen[id(t)] = False
    ok = not (isinstance(t, Variable) or t._fn is None or t._stateful or getattr(t, "_fed", False)
              or hasattr(t, "_deps"))
    if ok:
      found = []
      _walk((t._args, t._kwargs), lambda a: found.append(a) or a)
      ok = all(const(a) for a in found)
    seen[id(t)] = ok
    return ok
  if not const(tensor):
    return None
  v = tensor._value
  return v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else None


def tile_batch(t, multiplier, name=None):
  """tf.contrib.seq2seq.tile_batch: every [B, ...] entry of the structure -> [B * multiplier, ...], each row repeated
  `multiplier` times in place (b0, b0, ..., b1, b1, ...)."""
  return _rnn.map_structure(
      lambda x: Tensor(lambda v: torch.repeat_interleave(_t(v), int(multiplier), dim=0), (x,), name="tile_batch"), t)


def gather_tree(step_ids, parent_ids, max_sequence_lengths, end_token, name=None):
  """tf.contrib.seq2seq.gather_tree (the beam_search_ops kernel): [T, B, W] step ids and parent beams -> the full
  sequence behind every final beam, read backwards from step min(T, max_sequence_lengths[b]) - 1; positions past that
  length, and every position after a sequence's first end_token, hold end_token."""
  def f(ids, par, lens, end):
    ids, par, lens, end = _t(ids).long(), _t(par).long(), _t(lens).long(), int(_t(end))
    T, B, W = ids.shape
    out = torch.full((T, B, W), end, dtype=torch.long)
    L = torch.clamp(lens, max=T)
    cur = torch.arange(W).unsqueeze(0).expand(B, W).clone()      # the beam whose entry at step t is on the path
    for t in _range(T - 1, -1, -1):
      on = (t < L).unsqueeze(1).expand(B, W)
      out[t] = torch.where(on, torch.gather(ids[t], 1, cur), out[t])
      cur = torch.where(on, torch.gather(par[t], 1, cur), cur)
    ended = torch.cumsum((out == end).long(), 0) - (out == end).long() > 0      # an end_token strictly earlier
    out = torch.where(ended, torch.full_like(out, end), out)
    return out.to(torch.int32)
  return Tensor(f, (step_ids, parent_ids, max_sequence_lengths, end_token), name="gather_tree")


def reset_default_graph():
  global _GRAPH
  _GRAPH = Graph()
  _VARS.clear()
  _OP_NAMES.clear()
  _VAR_NAMES.clear()
  _DEFAULT_NAMES.clear()
  _SCOPES[:] = [VariableScope("")]
  del _CTRL[:]
  del _NAME[:]
  _LAYER_UIDS.clear()
  _RNG.manual_seed(0)


@contextlib.contextmanager
def control_dependencies(control_inputs):
  deps = []
  for c in (control_inputs or []):
    deps.extend([t for t in _flatten(c) if isinstance(t, Tensor)])
  _CTRL.append(deps)
  try:
    yield
  finally:
    _CTRL.pop()


def group(*inputs, **kwargs):
  deps = [t for t in _flatten(list(inputs)) if isinstance(t, Tensor)]
  t = Tensor(lambda: torch.zeros((), dtype=torch.bool), (), name=kwargs.get("name", "group"))
  t._ctrl = t._ctrl + deps
  return t


def no_op(name=None):
  return Tensor(lambda: torch.zeros((), dtype=torch.bool), (), name=name or "no_op")


def tuple_(tensors, name=None, control_inputs=None):
  return [identity(t) for t in tensors]


@contextlib.contextmanager
def device(name):
  yield


@contextlib.contextmanager
def colocate_with(op, ignore_existing=False):
  yield


def identity(x, name=None):
  return Tensor(lambda v: _t(v), (x,), name=name or "identity")


def stop_gradient(x, name=None):
  return Tensor(lambda v: _t(v).detach(), (x,), name="stop_gradient")


class _CondOut(Tensor):
  def __init__(self, pred, a, b):
    self._pred, self._a, self._b = pred, a, b
    pv = _PYBOOL(_t(pred._value if isinstance(pred, Tensor) else pred))
    chosen = a if pv else b
    super(_CondOut, self).__init__(None, (), name="cond", build_value=_t(chosen._value if isinstance(chosen, Tensor)
                                                                       else chosen))

  def _eval(self, ev):
    for c in self._ctrl:
      ev(c)
    p = ev(self._pred) if isinstance(self._pred, Tensor) else self._pred
    chosen = self._a if _PYBOOL(p) else self._b
    return ev(chosen) if isinstance(chosen, Tensor) else _t(chosen)


def cond(pred, true_fn=None, false_fn=None, strict=False, name=None, fn1=None, fn2=None):
  true_fn = true_fn or fn1
  false_fn = false_fn or fn2
  a, b = true_fn(), false_fn()          # TF1 builds BOTH branches; only the chosen one runs

  def pair(x, y):
    if isinstance(x, (list, _tuple)):
      return type(x)(pair(u, v) for u, v in zip(x, y))
    if x is None and y is None:
      return None
    return _CondOut(pred, x, y)
  return pair(a, b)


class _Grad(Tensor):
  """One entry of tf.gradients: torch.autograd over the values of THIS run."""

  def __init__(self, shared, index):
    self._shared, self._index = shared, index
    super(_Grad, self).__init__(None, (), name="gradients", build_value=shared.values(lambda t: t._value)[index])

  def _eval(self, ev):
    for c in self._ctrl:
      ev(c)
    return self._shared.values(ev)[self._index]


class _GradShared(object):
  def __init__(self, ys, xs, grad_ys):
    self.ys, self.xs, self.grad_ys = ys, xs, grad_ys
    self._memo = (None, None)

  def values(self, ev):
    if self._memo[0] is ev:
      return self._memo[1]
    ys = [ev(y) for y in self.ys]
    xs = [ev(x) for x in self.xs]
    gys = [torch.ones_like(y) if g is None else _t(ev(g) if isinstance(g, Tensor) else g).to(y.dtype)
           for y, g in zip(ys, self.grad_ys)]
    live = [i for i, x in enumerate(xs) if x.requires_grad]
    res = [torch.zeros_like(x) for x in xs]
    if live and any(y.requires_grad for y in ys):
      keep = [(y, g) for y, g in zip(ys, gys) if y.requires_grad]
      gs = torch.autograd.grad([k[0] for k in keep], [xs[i] for i in live], [k[1] for k in keep],
                               allow_unused=True, retain_graph=True)
      for i, g in zip(live, gs):
        if g is not None:
          res[i] = g
    self._memo = (ev, res)
    return res


def gradients(ys, xs, grad_ys=None, name="gradients", colocate_gradients_with_ops=False, gate_gradients=False,
              aggregation_method=None, stop_gradients=None):
  ys = list(ys) if isinstance(ys, (list, _tuple)) else [ys]
  single = not isinstance(xs, (list, _tuple))
  xs = [xs] if single else list(xs)
  grad_ys = list(grad_ys) if grad_ys is not None else [None] * len(ys)
  sh = _GradShared(ys, xs, grad_ys)
  out = [_Grad(sh, i) for i in _range(len(xs))]
  return out


class IndexedSlices(object):
  def __init__(self, values, indices, dense_shape=None):
    self.values, self.indices, self.dense_shape = values, indices, dense_shape

  @property
  def dtype(self):
    return self.values.dtype


class SparseTensor(object):
  def __init__(self, indices, values, dense_shape):
    self.indices, self.values, self.dense_shape = indices, values, dense_shape


# ----------------------------------------------------------------------------------------------- ops


def _axis(a):
  if a is None:
    return None
  if isinstance(a, torch.Tensor):
    a = a.tolist()
  if isinstance(a, (list, _tuple)):
    return _tuple(int(v) for v in a)
  return int(a)


def cast(x, dtype, name=None):
  td = as_dtype(dtype).torch
  return Tensor(lambda v: _t(v).to(td), (x,), name="cast")


def saturate_cast(value, dtype, name=None):
  d = as_dtype(dtype)

  def f(v):
    v = _t(v)
    if d.is_floating and v.dtype.is_floating_point and torch.finfo(d.torch).max < torch.finfo(v.dtype).max:
      v = v.clamp(torch.finfo(d.torch).min, torch.finfo(d.torch).max)
    return v.to(d.torch)
  return Tensor(f, (value,), name="saturate_cast")


to_float = lambda x, name=None: cast(x, float32)          # noqa: E731
to_int32 = lambda x, name=None: cast(x, int32)            # noqa: E731
to_int64 = lambda x, name=None: cast(x, int64)            # noqa: E731


def shape(input, name=None, out_type=None):               # noqa: A002
  td = as_dtype(out_type or int32).torch
  return Tensor(lambda v: torch.tensor(list(_t(v).shape), dtype=td), (input,), name="shape")


def size(input, name=None, out_type=None):                # noqa: A002
  return Tensor(lambda v: torch.tensor(_t(v).numel(), dtype=torch.int32), (input,), name="size")


def rank(input, name=None):                               # noqa: A002
  return Tensor(lambda v: torch.tensor(_t(v).dim(), dtype=torch.int32), (input,), name="rank")


def reshape(tensor, shape, name=None):                    # noqa: A002
  return Tensor(lambda v, s: _t(v).reshape(_ishape(s)), (tensor, shape), name="reshape")


def transpose(a, perm=None, name=None, conjugate=False):
  return Tensor(lambda v, p: _t(v).permute(*(_ishape(p) if p is not None else reversed(_range(_t(v).dim())))),
                (a, perm), name="transpose")


def expand_dims(input, axis=None, name=None, dim=None):   # noqa: A002
  ax = axis if axis is not None else dim

  def f(v, a):
    v = _t(v)
    a = int(a)
    return v.unsqueeze(a if a >= 0 else v.dim() + 1 + a)
  return Tensor(f, (input, ax), name="expand_dims")


def squeeze(input, axis=None, name=None, squeeze_dims=None):   # noqa: A002
  ax = axis if axis is not None else squeeze_dims

  def f(v, a):
    v = _t(v)
    if a is None:
      return v.squeeze()
    a = _axis(a)
    for d in sorted([a] if isinstance(a, int) else list(a), key=lambda z: z % v.dim(), reverse=True):
      v = v.squeeze(d)
    return v
  return Tensor(f, (input, ax), name="squeeze")


def concat(values, axis, name="concat"):
  return Tensor(lambda vs, a: torch.cat([_t(v) for v in vs], dim=int(a)), (list(values), axis), name="concat")


def stack(values, axis=0, name="stack"):
  return Tensor(lambda vs: torch.stack([_t(v) for v in vs], dim=int(axis)), (list(values),), name="stack")


def unstack(value, num=None, axis=0, name="unstack"):
  n = int(value._value.shape[axis])
  return [Tensor(lambda v, i=i: _t(v).select(axis, i), (value,), name="unstack") for i in _range(n)]


def split(value, num_or_size_splits, axis=0, num=None, name="split"):
  n = num_or_size_splits
  total = int(value._value.shape[axis])
  sizes = [total // n] * n if isinstance(n, int) else _ishape(n)
  offs = np.cumsum([0] + sizes)
  return [Tensor(lambda v, o=int(offs[i]), s=sizes[i]: _t(v).narrow(axis, o, s), (value,), name="split")
          for i in _range(len(sizes))]


def tile(input, multiples, name=None):                    # noqa: A002
  return Tensor(lambda v, m: _t(v).repeat(*_ishape(m)), (input, multiples), name="tile")


def pad(tensor, paddings, mode="CONSTANT", name=None, constant_values=0):
  def f(v, p):
    v = _t(v)
    p = _t(p).tolist() if not isinstance(p, list) else [[int(_t(a)) for a in row] for row in p]
    flat = []
    for lo, hi in reversed(p):
      flat += [int(lo), int(hi)]
    return torch.nn.functional.pad(v, flat, value=constant_values)
  return Tensor(f, (tensor, paddings), name="pad")


def slice_(input_, begin, size, name=None):               # noqa: A002
  def f(v, b, s):
    v = _t(v)
    b, s = _ishape(b), _ishape(s)
    idx = _tuple(_slice(bi, None if si == -1 else bi + si) for bi, si in zip(b, s))
    return v[idx]
  return Tensor(f, (input_, begin, size), name="slice")


def gather(params, indices, validate_indices=None, name=None, axis=0):
  return Tensor(lambda p, i: torch.index_select(_t(p), int(axis), _t(i).long().reshape(-1)).reshape(
      list(_t(p).shape[:int(axis)]) + list(_t(i).shape) + list(_t(p).shape[int(axis) + 1:])),
      (params, indices), name="gather")


def gather_nd(params, indices, name=None):
  def f(p, i):
    p, i = _t(p), _t(i).long()
    return p[_tuple(i[..., k] for k in _range(i.shape[-1]))]
  return Tensor(f, (params, indices), name="gather_nd")


def scatter_nd(indices, updates, shape, name=None):       # noqa: A002
  def f(i, u, s):
    i, u = _t(i).long(), _t(u)
    out = torch.zeros(_ishape(s), dtype=u.dtype)
    return out.index_put(_tuple(i[..., k] for k in _range(i.shape[-1])), u, accumulate=True)
  return Tensor(f, (indices, updates, shape), name="scatter_nd")


def where(condition, x=None, y=None, name=None):
  if x is None and y is None:
    return Tensor(lambda c: torch.nonzero(_t(c)), (condition,), name="where")
  def sel(c, a, b):
    c = _t(c)
    a = _t(a, like=_t(b) if isinstance(b, torch.Tensor) else None)
    b = _t(b, like=a)
    if c.dim() == 1 and a.dim() > 1:          # tf.where with a vector condition selects whole ROWS of x / y
      c = c.reshape([-1] + [1] * (a.dim() - 1))
    return torch.where(c, a, b)
  return Tensor(sel, (condition, x, y), name="select")


def sequence_mask(lengths, maxlen=None, dtype=bool, name=None):
  td = as_dtype(dtype).torch

  def f(l, m):
    l = _t(l)
    m = int(l.max()) if m is None else int(_t(m))
    return (torch.arange(m, dtype=torch.int64) < l.long().unsqueeze(-1)).to(td)
  return Tensor(f, (lengths, maxlen), name="sequence_mask")


def one_hot(indices, depth, on_value=None, off_value=None, axis=None, dtype=None, name=None):
  if dtype is None:
    dtype = next((v.dtype for v in (on_value, off_value) if isinstance(v, Tensor)), float32)
  td = as_dtype(dtype).torch

  def f(i, d, on, off):
    i = _t(i).long()
    hit = torch.nn.functional.one_hot(i.clamp(min=0), int(_t(d))).bool() & (i >= 0).unsqueeze(-1)
    on = torch.ones((), dtype=td) if on is None else _t(on).to(td)
    off = torch.zeros((), dtype=td) if off is None else _t(off).to(td)
    return torch.where(hit, on, off)           # a select: an infinite off_value stays out of the on positions
  return Tensor(f, (indices, depth, on_value, off_value), name="one_hot")


def range_(start, limit=None, delta=1, dtype=None, name="range"):
  def f(s, l, d):
    s, d = _t(s), _t(d)
    if l is None:
      s, l = torch.zeros_like(s), s
    l = _t(l)
    isf = any(v.dtype.is_floating_point for v in (s, l, d))
    td = as_dtype(dtype).torch if dtype is not None else (torch.float32 if isf else torch.int32)
    return torch.arange(s.item(), l.item(), d.item(), dtype=td)
  return Tensor(f, (start, limit, delta), name="range")


def zeros(shape, dtype=float32, name=None):               # noqa: A002
  return Tensor(lambda s: torch.zeros(_ishape(s), dtype=as_dtype(dtype).torch), (shape,), name="zeros")


def ones(shape, dtype=float32, name=None):                # noqa: A002
  return Tensor(lambda s: torch.ones(_ishape(s), dtype=as_dtype(dtype).torch), (shape,), name="ones")


def fill(dims, value, name=None):
  return Tensor(lambda s, v: torch.full(_ishape(s), _t(v).item(), dtype=_t(v).dtype), (dims, value), name="fill")


def zeros_like(tensor, dtype=None, name=None, optimize=True):
  return Tensor(lambda v: torch.zeros_like(_t(v), dtype=as_dtype(dtype).torch if dtype else None), (tensor,))


def ones_like(tensor, dtype=None, name=None, optimize=True):
  return Tensor(lambda v: torch.ones_like(_t(v), dtype=as_dtype(dtype).torch if dtype else None), (tensor,))


def matmul(a, b, transpose_a=False, transpose_b=False, adjoint_a=False, adjoint_b=False, a_is_sparse=False,
           b_is_sparse=False, name=None):
  def f(x, y):
    x, y = _t(x), _t(y)
    if transpose_a or adjoint_a:
      x = x.transpose(-1, -2)
    if transpose_b or adjoint_b:
      y = y.transpose(-1, -2)
    return x @ y
  return Tensor(f, (a, b), name="matmul")


def tensordot(a, b, axes, name=None):
  return Tensor(lambda x, y: torch.tensordot(_t(x), _t(y), dims=axes), (a, b), name="tensordot")


def _reduce(name, fn):
  def op(input_tensor, axis=None, keepdims=None, name=None, reduction_indices=None, keep_dims=None):
    ax = axis if axis is not None else reduction_indices
    kd = _PYBOOL(keepdims if keepdims is not None else (keep_dims or False))
    if ax is None:
      _unique_op_name("Const")          # TensorFlow materialises range(rank) as a Const op for a full reduction

    def f(v, a):
      v = _t(v)
      a = _axis(a)
      if a is None:
        a = _tuple(_range(v.dim()))
      if isinstance(a, int):
        a = (a,)
      if len(a) == 0:
        return v
      return fn(v, a, kd)
    return Tensor(f, (input_tensor, ax), name=name)
  return op


def _amax(v, a, kd):
  for d in sorted([x % v.dim() for x in a], reverse=True):
    v = v.max(dim=d, keepdim=kd).values
  return v


def _amin(v, a, kd):
  for d in sorted([x % v.dim() for x in a], reverse=True):
    v = v.min(dim=d, keepdim=kd).values
  return v


reduce_sum = _reduce("reduce_sum", lambda v, a, kd: v.sum(dim=a, keepdim=kd))
reduce_mean = _reduce("reduce_mean", lambda v, a, kd: v.mean(dim=a, keepdim=kd))
reduce_max = _reduce("reduce_max", _amax)
reduce_min = _reduce("reduce_min", _amin)
reduce_any = _reduce("reduce_any", lambda v, a, kd: v.to(torch.int32).sum(dim=a, keepdim=kd) > 0)
reduce_all = _reduce("reduce_all", lambda v, a, kd: (~v).to(torch.int32).sum(dim=a, keepdim=kd) == 0)
reduce_prod = _reduce("reduce_prod", lambda v, a, kd: _prod(v, a, kd))
reduce_logsumexp = _reduce("reduce_logsumexp", lambda v, a, kd: torch.logsumexp(v, dim=a, keepdim=kd))


def _prod(v, a, kd):
  for d in sorted([x % v.dim() for x in a], reverse=True):
    v = v.prod(dim=d, keepdim=kd)
  return v


def _unary(name, fn):
  return lambda x, name=None: Tensor(lambda v: fn(_t(v)), (x,), name=name)


abs = _unary("abs", torch.abs)                  # noqa: A001
square = _unary("square", lambda v: v * v)
sqrt = _unary("sqrt", torch.sqrt)
rsqrt = _unary("rsqrt", torch.rsqrt)
exp = _unary("exp", torch.exp)
log = _unary("log", torch.log)
sin = _unary("sin", torch.sin)
cos = _unary("cos", torch.cos)
tanh = _unary("tanh", torch.tanh)
sigmoid = _unary("sigmoid", torch.sigmoid)
floor = _unary("floor", torch.floor)
ceil = _unary("ceil", torch.ceil)
round = _unary("round", torch.round)            # noqa: A001
sign = _unary("sign", torch.sign)
negative = _unary("neg", torch.neg)
logical_not = _unary("logical_not", torch.logical_not)
is_nan = _unary("is_nan", torch.isnan)
is_inf = _unary("is_inf", torch.isinf)
is_finite = _unary("is_finite", torch.isfinite)
log1p = _unary("log1p", torch.log1p)
reciprocal = _unary("reciprocal", torch.reciprocal)


def _binary(name, fn):
  o = _binop(name, fn)
  return lambda x, y, name=None: o(x, y)


add = _binary("add", lambda x, y: x + y)
subtract = _binary("sub", lambda x, y: x - y)
multiply = _binary("mul", lambda x, y: x * y)
div = _binary("div", _pydiv)
divide = truediv = _binary("truediv", _tdiv)
realdiv = _binary("realdiv", lambda x, y: x / y)
floordiv = _binary("floordiv", lambda x, y: torch.div(x, y, rounding_mode="floor"))
mod = floormod = _binary("mod", torch.remainder)
maximum = _binary("maximum", torch.maximum)
minimum = _binary("minimum", torch.minimum)
pow = _binary("pow", torch.pow)                 # noqa: A001
equal = _binary("equal", lambda x, y: x == y)
not_equal = _binary("not_equal", lambda x, y: x != y)
less = _binary("less", lambda x, y: x < y)
less_equal = _binary("less_equal", lambda x, y: x <= y)
greater = _binary("greater", lambda x, y: x > y)
greater_equal = _binary("greater_equal", lambda x, y: x >= y)
logical_and = _binary("logical_and", torch.logical_and)
logical_or = _binary("logical_or", torch.logical_or)
squared_difference = _binary("squared_difference", lambda x, y: (x - y) * (x - y))


def add_n(inputs, name=None):
  return Tensor(lambda vs: sum(_t(v) for v in vs[1:]) + _t(vs[0]) if len(vs) > 1 else _t(vs[0]), (list(inputs),))


def argmax(input, axis=None, name=None, dimension=None, output_type=int64):     # noqa: A002
  ax = axis if axis is not None else (dimension if dimension is not None else 0)
  return Tensor(lambda v: _t(v).argmax(dim=int(ax)).to(as_dtype(output_type).torch), (input,), name="argmax")


def argmin(input, axis=None, name=None, dimension=None, output_type=int64):     # noqa: A002
  ax = axis if axis is not None else (dimension if dimension is not None else 0)
  return Tensor(lambda v: _t(v).argmin(dim=int(ax)).to(as_dtype(output_type).torch), (input,), name="argmin")


def clip_by_value(t, clip_value_min, clip_value_max, name=None):
  return Tensor(lambda v, lo, hi: torch.minimum(torch.maximum(_t(v), _t(lo, like=_t(v))), _t(hi, like=_t(v))),
                (t, clip_value_min, clip_value_max), name="clip_by_value")


def norm(tensor, ord="euclidean", axis=None, keepdims=None, name=None, keep_dims=None):   # noqa: A002
  kd = _PYBOOL(keepdims if keepdims is not None else (keep_dims or False))

  def f(v):
    v = _t(v)
    a = _axis(axis)
    p = 2 if ord in ("euclidean", 2, "fro") else (1 if ord == 1 else float(ord))
    if a is None:
      r = v.reshape(-1).norm(p=p)
      return r.reshape([1] * v.dim()) if kd else r
    return v.norm(p=p, dim=a, keepdim=kd)
  return Tensor(f, (tensor,), name="norm")


def global_norm(t_list, name=None):
  def f(vs):
    tot = None
    for v in vs:
      if v is None:
        continue
      s = (_t(v).to(torch.float32) ** 2).sum()
      tot = s if tot is None else tot + s
    return torch.sqrt(tot)
  return Tensor(f, ([v.values if isinstance(v, IndexedSlices) else v for v in t_list],), name="global_norm")


def clip_by_global_norm(t_list, clip_norm, use_norm=None, name=None):
  gn = use_norm if use_norm is not None else global_norm(t_list)
  scale = Tensor(lambda g, c: _t(c, like=g) / torch.maximum(g, _t(c, like=g)), (gn, clip_norm))
  return [None if t is None else t * scale for t in t_list], gn


def matrix_band_part(input, num_lower, num_upper, name=None):     # noqa: A002
  def f(v, lo, hi):
    v = _t(v)
    lo, hi = int(_t(lo)), int(_t(hi))
    m, n = v.shape[-2], v.shape[-1]
    i = torch.arange(m).unsqueeze(1)
    j = torch.arange(n).unsqueeze(0)
    keep = ((lo < 0) | ((i - j) <= lo)) & ((hi < 0) | ((j - i) <= hi))
    return v * keep.to(v.dtype)
  return Tensor(f, (input, num_lower, num_upper), name="matrix_band_part")


def random_uniform(shape, minval=0, maxval=None, dtype=float32, seed=None, name=None):    # noqa: A002
  d = as_dtype(dtype)

  def f(s, lo, hi):
    lo = _t(lo).item()
    hi = 1.0 if hi is None else _t(hi).item()
    if d.is_integer:
      return torch.randint(int(lo), int(hi), _ishape(s), generator=_RNG).to(d.torch)
    return (torch.rand(_ishape(s), generator=_RNG, dtype=torch.float64) * (hi - lo) + lo).to(d.torch)
  return Tensor(f, (shape, minval, maxval), name="random_uniform")


def random_normal(shape, mean=0.0, stddev=1.0, dtype=float32, seed=None, name=None):      # noqa: A002
  return Tensor(lambda s: (torch.randn(_ishape(s), generator=_RNG, dtype=torch.float64) * stddev + mean).to(
      as_dtype(dtype).torch), (shape,), name="random_normal")


def cumsum(x, axis=0, exclusive=False, reverse=False, name=None):
  def f(v):
    v = _t(v)
    if reverse:
      v = v.flip(axis)
    r = v.cumsum(axis)
    if exclusive:
      r = r - v
    return r.flip(axis) if reverse else r
  return Tensor(f, (x,), name="cumsum")


def reverse(tensor, axis, name=None):
  return Tensor(lambda v: _t(v).flip(list(_axis(axis)) if isinstance(_axis(axis), _tuple) else [_axis(axis)]), (tensor,))


def py_func(func, inp, Tout, stateful=True, name=None):
  raise NotImplementedError("tf.py_func is outside the paths this shim executes")


def load_op_library(path):
  raise NotImplementedError("custom op libraries are not available in the shim")


def Print(input_, data, message=None, first_n=None, summarize=None, name=None):   # noqa: N802
  return identity(input_)


def check_numerics(tensor, message, name=None):
  return identity(tensor)


def verify_tensor_all_finite(t, msg, name=None):
  return identity(t)


class _Noop(object):
  def __getattr__(self, n):
    return lambda *a, **k: None


summary = _Noop()
logging = types.SimpleNamespace(info=lambda *a, **k: None, warning=lambda *a, **k: None, warn=lambda *a, **k: None,
                                error=lambda *a, **k: None, set_verbosity=lambda *a, **k: None, INFO=20, WARN=30)


# ----------------------------------------------------------------------------------------------- tf.nn


def _softmax(logits, axis=None, name=None, dim=None):
  ax = axis if axis is not None else (dim if dim is not None else -1)
  return Tensor(lambda v: torch.softmax(_t(v), dim=int(ax)), (logits,), name="softmax")


def _log_softmax(logits, axis=None, name=None, dim=None):
  ax = axis if axis is not None else (dim if dim is not None else -1)
  return Tensor(lambda v: torch.log_softmax(_t(v), dim=int(ax)), (logits,), name="log_softmax")


DROPOUT_TAP = None      # fixture generator's tap: when a list, every evaluated dropout appends its scaled keep mask
DROPOUT_OFF = False     # fixture generator's switch: dropout passes its input through (the deterministic part of a
                        # graph whose dropout cannot be configured away: Tacotron's pre-net, rate 0.5 in every mode)


def _dropout(x, keep_prob=None, noise_shape=None, seed=None, name=None, rate=None):
  kp = keep_prob if keep_prob is not None else (None if rate is None else 1.0 - rate)
  if DROPOUT_OFF or (not isinstance(kp, Tensor) and float(kp) == 1.0):
    return identity(x)                # TF: keep_prob == 1 returns x itself

  def f(v, k):
    v = _t(v)
    k = float(_t(k))
    shp = list(v.shape) if noise_shape is None else _ishape(noise_shape)
    mask = (torch.rand(shp, generator=_RNG, dtype=torch.float32) + k).floor().to(v.dtype)
    if DROPOUT_TAP is not None:
      DROPOUT_TAP.append(mask / k)
    return v / k * mask
  return Tensor(f, (x, kp), name="dropout")


def _sparse_xent(_sentinel=None, labels=None, logits=None, name=None):
  def f(l, z):
    z = _t(z)
    l = _t(l).long()
    lp = torch.log_softmax(z, dim=-1)
    return -lp.gather(-1, l.unsqueeze(-1)).squeeze(-1)
  return Tensor(f, (labels, logits), name="sparse_softmax_cross_entropy")


def _soft_xent_v2(_sentinel=None, labels=None, logits=None, dim=-1, name=None, axis=None):
  ax = axis if axis is not None else dim
  return Tensor(lambda l, z: -(_t(l) * torch.log_softmax(_t(z), dim=int(ax))).sum(dim=int(ax)), (labels, logits),
                name="softmax_cross_entropy_with_logits")


def _soft_xent(_sentinel=None, labels=None, logits=None, dim=-1, name=None):
  return _soft_xent_v2(labels=stop_gradient(labels), logits=logits, dim=dim)


def _moments(x, axes, shift=None, name=None, keep_dims=False, keepdims=None):
  kd = _PYBOOL(keepdims if keepdims is not None else keep_dims)
  m = reduce_mean(x, axis=axes, keepdims=True)
  v = reduce_mean(square(x - stop_gradient(m)), axis=axes, keepdims=True)
  if not kd:
    m, v = squeeze(m, axes), squeeze(v, axes)
  return m, v


def _bias_add(value, bias, data_format=None, name=None):
  if data_format == "NCHW":
    return Tensor(lambda v, b: _t(v) + _t(b).reshape([1, -1] + [1] * (_t(v).dim() - 2)), (value, bias))
  return value + bias


def _same_pad(n, k, s, d):
  """TensorFlow "SAME": out = ceil(n / s); total = max((out - 1) s + (k - 1) d + 1 - n, 0); the extra one goes
  to the END (tensorflow/core/framework/common_shape_fns.cc GetWindowedOutputSizeVerboseV2)."""
  out = -(-n // s)
  total = max((out - 1) * s + (k - 1) * d + 1 - n, 0)
  return total // 2, total - total // 2


def _conv_nd(x, w, strides, padding, dilations, nd, groups=1):
  """x channels-first torch value, w torch layout [out, in / groups, *k]."""
  padding = padding.upper()
  if padding == "SAME":
    pads = []
    for i in _range(nd):
      lo, hi = _same_pad(x.shape[2 + i], w.shape[2 + i], strides[i], dilations[i])
      pads = [lo, hi] + pads
    x = torch.nn.functional.pad(x, pads)
  fn = {1: torch.nn.functional.conv1d, 2: torch.nn.functional.conv2d}[nd]
  return fn(x, w, None, stride=_tuple(strides), padding=0, dilation=_tuple(dilations), groups=groups)


def _ctc_greedy_decoder(inputs, sequence_length, merge_repeated=True):
  """tf.nn.ctc_greedy_decoder: inputs [T, B, V] logits, blank = V - 1; returns ([SparseTensor], neg_sum_logits)."""
  def f(z, sl):
    z, sl = _t(z), _t(sl).long()
    T, B, V = z.shape
    idx, vals, negsum = [], [], torch.zeros(B, 1, dtype=z.dtype)
    maxlen = 0
    for b in _range(B):
      prev, out = -1, []
      for t in _range(int(sl[b])):
        m = int(z[t, b].argmax())
        negsum[b, 0] -= z[t, b, m]
        if m != V - 1 and not (merge_repeated and m == prev):
          out.append(m)
        prev = m
      for j, c in enumerate(out):
        idx.append([b, j])
        vals.append(c)
      maxlen = max(maxlen, len(out))
    return (torch.tensor(idx, dtype=torch.int64).reshape(-1, 2), torch.tensor(vals, dtype=torch.int64),
            torch.tensor([B, maxlen], dtype=torch.int64), negsum)
  parts = [Tensor(lambda z, sl, i=i: f(z, sl)[i], (inputs, sequence_length), name="ctc_greedy") for i in _range(4)]
  return [SparseTensor(parts[0], parts[1], parts[2])], parts[3]


def _ctc_loss(labels, inputs, sequence_length, preprocess_collapse_repeated=False, ctc_merge_repeated=True,
              ignore_longer_outputs_than_inputs=False, time_major=True):
  """tf.nn.ctc_loss: labels SparseTensor [B, Lmax], inputs logits [T, B, V] (time major), blank = V - 1; per-sample
  negative log likelihood [B]; with ignore_longer_outputs_than_inputs a sample whose label sequence cannot be
  emitted in its input length gets loss 0 and no gradient (otherwise TensorFlow raises). Restated on
  torch.nn.functional.ctc_loss (float64 log-softmax)."""
  def f(idx, vals, shp, z, sl):
    z = _t(z)
    if not time_major:
      z = z.transpose(0, 1)
    T, B, V = z.shape
    sl = _t(sl).long()
    idx, vals = _t(idx).long(), _t(vals).long()
    lab = [[] for _ in _range(B)]
    for (b, j), c in zip(idx.tolist(), vals.tolist()):
      lab[b].append(c)
    lens = torch.tensor([len(l) for l in lab], dtype=torch.long)
    L = int(max(1, lens.max()))
    dense = torch.zeros((B, L), dtype=torch.long)
    for b, l in enumerate(lab):
      dense[b, :len(l)] = torch.tensor(l, dtype=torch.long)
    need = torch.tensor([len(l) + sum(1 for a, c in zip(l, l[1:]) if a == c) for l in lab], dtype=torch.long)
    feasible = need <= sl
    if not ignore_longer_outputs_than_inputs and not _PYBOOL(feasible.all()):
      raise ValueError("Not enough time for target transition sequence")
    lp = torch.log_softmax(z.double(), dim=-1)
    loss = torch.nn.functional.ctc_loss(lp, dense, sl, lens, blank=V - 1, reduction="none", zero_infinity=True)
    return torch.where(feasible, loss, torch.zeros_like(loss)).to(z.dtype)
  return Tensor(f, (labels.indices, labels.values, labels.dense_shape, inputs, sequence_length), name="ctc_loss")


def _depthwise_conv2d(input, filter, strides, padding, rate=None, name=None, data_format=None):   # noqa: A002
  """tf.nn.depthwise_conv2d: filter [KH, KW, Cin, mult], output channel = c * mult + m."""
  nhwc = data_format in (None, "NHWC")

  def f(x, w):
    x, w = _t(x), _t(w)
    w = w.to(x.dtype)
    if nhwc:
      x = x.permute(0, 3, 1, 2)
    kh, kw, cin, mult = w.shape
    wt = w.permute(2, 3, 0, 1).reshape(cin * mult, 1, kh, kw)
    st = strides[1:3] if nhwc else strides[2:4]
    y = _conv_nd(x, wt, [int(v) for v in st], padding, [1, 1] if rate is None else list(rate), 2, groups=cin)
    return y.permute(0, 2, 3, 1) if nhwc else y
  return Tensor(f, (input, filter), name="depthwise_conv2d")


def _top_k(input, k=1, sorted=True, name=None):          # noqa: A002
  """tf.nn.top_k: descending, the LOWER index first among equal values (a stable sort)."""
  def f(v, kk, which):
    r = torch.sort(_t(v), dim=-1, descending=True, stable=True)
    return r.values[..., :int(_t(kk))] if which == 0 else r.indices[..., :int(_t(kk))].to(torch.int32)
  return Tensor(f, (input, k, 0), name="top_k"), Tensor(f, (input, k, 1), name="top_k")


top_k = _top_k


def sparse_tensor_to_dense(sp_input, default_value=0, validate_indices=True, name=None):
  def f(i, v, s):
    out = torch.full(_ishape(s), default_value, dtype=_t(v).dtype)
    if _t(i).numel():
      out[_tuple(_t(i).long().t())] = _t(v)
    return out
  return Tensor(f, (sp_input.indices, sp_input.values, sp_input.dense_shape), name="sparse_to_dense")


nn = types.SimpleNamespace(
    softmax=_softmax, log_softmax=_log_softmax, relu=_unary("relu", torch.relu), tanh=tanh, sigmoid=sigmoid,
    relu6=_unary("relu6", lambda v: v.clamp(0, 6)), elu=_unary("elu", torch.nn.functional.elu),
    dropout=_dropout, sparse_softmax_cross_entropy_with_logits=_sparse_xent,
    softmax_cross_entropy_with_logits_v2=_soft_xent_v2, softmax_cross_entropy_with_logits=_soft_xent,
    moments=_moments, bias_add=_bias_add, l2_loss=lambda t, name=None: reduce_sum(square(t)) / 2.0,
    ctc_greedy_decoder=_ctc_greedy_decoder, ctc_loss=_ctc_loss, top_k=_top_k, depthwise_conv2d=None, l2_normalize=lambda x, axis=None, epsilon=1e-12, name=None, dim=None:
    x * rsqrt(maximum(reduce_sum(square(x), axis if axis is not None else dim, keepdims=True), epsilon)))


# ----------------------------------------------------------------------------------------------- tf.layers

_LAYER_UIDS = {}


def _snake(name):
  s = re.sub("(.)([A-Z][a-z0-9]+)", r"\1_\2", name)
  s = re.sub("([a-z])([A-Z])", r"\1_\2", s).lower()
  return ("private" + s) if s.startswith("_") else s


class Layer(object):
  """tf.layers.Layer: variables live under a variable scope named after the layer (explicit `name`, else the
  snake-cased class name made unique among its siblings), opened on every call and created on the first."""

  def __init__(self, trainable=True, name=None, dtype=None, **kwargs):
    self.trainable, self.built, self._given_name, self._scope_obj = trainable, False, name, None
    self.dtype = dtype

  @property
  def name(self):
    return self._given_name or _snake(self.__class__.__name__)

  @property
  def _base_name(self):
    return self.name

  def build(self, input_shape):
    self.built = True

  def call(self, inputs, *args, **kwargs):
    return inputs

  def add_variable(self, name, shape, dtype=None, initializer=None, regularizer=None, trainable=True, **kw):
    if dtype is None:       # tf.layers: a layer's variables take the layer's dtype = the dtype of its first input
      dtype = self.dtype if self.dtype is not None else getattr(self, "_input_dtype", None)
    return get_variable(name, shape, dtype=dtype, initializer=initializer, regularizer=regularizer,
                        trainable=trainable and self.trainable)

  add_weight = add_variable

  def __call__(self, inputs, *args, **kwargs):
    if self._scope_obj is None:
      cm = variable_scope(self._given_name) if self._given_name else variable_scope(None, default_name=self.name)
      with cm as sc:
        self._scope_obj = sc
        return self._run(inputs, args, kwargs)
    with variable_scope(self._scope_obj, reuse=True if self.built else None):
      return self._run(inputs, args, kwargs)

  def _run(self, inputs, args, kwargs):
    if not self.built:
      first = inputs[0] if isinstance(inputs, (list, _tuple)) else inputs
      if isinstance(first, Tensor) and first.dtype.is_floating:
        self._input_dtype = first.dtype
      with variable_scope(_scope(), reuse=AUTO_REUSE):
        self.build(first.get_shape() if isinstance(first, Tensor) else None)
      self.built = True
    return self.call(inputs, *args, **kwargs)

  apply = __call__

  @property
  def scope_name(self):
    return self._scope_obj.name if self._scope_obj is not None else None

  @property
  def variables(self):
    if self.scope_name is None:
      return []
    p = self.scope_name + "/"
    return [v for v in global_variables() if v.name.startswith(p)]

  weights = variables

  @property
  def trainable_variables(self):
    if self.scope_name is None:           # not called yet: TensorFlow returns the (empty) list of built weights
      return []
    p = self.scope_name + "/"
    return [v for v in trainable_variables() if v.name.startswith(p)]

  trainable_weights = trainable_variables


class Dense(Layer):
  def __init__(self, units, activation=None, use_bias=True, kernel_initializer=None, bias_initializer=None,
               kernel_regularizer=None, bias_regularizer=None, activity_regularizer=None, kernel_constraint=None,
               bias_constraint=None, trainable=True, name=None, **kwargs):
    super(Dense, self).__init__(trainable=trainable, name=name, **kwargs)
    self.units, self.activation, self.use_bias = int(units), activation, use_bias
    self.kernel_initializer, self.bias_initializer = kernel_initializer, bias_initializer or zeros_initializer()
    self.kernel_regularizer, self.bias_regularizer = kernel_regularizer, bias_regularizer

  @property
  def name(self):
    return self._given_name or "dense"

  def build(self, input_shape):
    cin = int(input_shape[-1])
    self.kernel = self.add_variable("kernel", [cin, self.units], initializer=self.kernel_initializer,
                                    regularizer=self.kernel_regularizer)
    self.bias = self.add_variable("bias", [self.units], initializer=self.bias_initializer,
                                  regularizer=self.bias_regularizer) if self.use_bias else None
    self.built = True

  def call(self, inputs):
    y = Tensor(lambda x, w: _t(x) @ _t(w).to(_t(x).dtype), (inputs, self.kernel), name="dense")
    if self.use_bias:
      y = y + cast(self.bias, y.dtype)
    return self.activation(y) if self.activation is not None else y


def _dense(inputs, units, activation=None, use_bias=True, kernel_initializer=None, bias_initializer=None,
           kernel_regularizer=None, bias_regularizer=None, activity_regularizer=None, kernel_constraint=None,
           bias_constraint=None, trainable=True, name=None, reuse=None):
  return Dense(units, activation, use_bias, kernel_initializer, bias_initializer, kernel_regularizer,
               bias_regularizer, trainable=trainable, name=name)(inputs)


class _Conv(Layer):
  ND = 1

  def __init__(self, filters, kernel_size, strides=1, padding="valid", data_format="channels_last",
               dilation_rate=1, activation=None, use_bias=True, kernel_initializer=None, bias_initializer=None,
               kernel_regularizer=None, bias_regularizer=None, activity_regularizer=None, kernel_constraint=None,
               bias_constraint=None, trainable=True, name=None, **kwargs):
    super(_Conv, self).__init__(trainable=trainable, name=name, dtype=kwargs.get("dtype"))
    tup = lambda v: _tuple(int(a) for a in (v if isinstance(v, (list, _tuple)) else [v] * self.ND))   # noqa: E731
    self.filters, self.k, self.s, self.d = int(filters), tup(kernel_size), tup(strides), tup(dilation_rate)
    self.padding, self.data_format, self.activation, self.use_bias = padding, data_format, activation, use_bias
    self.kernel_initializer, self.bias_initializer = kernel_initializer, bias_initializer or zeros_initializer()
    self.kernel_regularizer, self.bias_regularizer = kernel_regularizer, bias_regularizer

  @property
  def name(self):
    return self._given_name or ("conv%dd" % self.ND)

  def _cin(self, input_shape):
    return int(input_shape[-1] if self.data_format == "channels_last" else input_shape[1])

  def build(self, input_shape):
    self.kernel = self.add_variable("kernel", list(self.k) + [self._cin(input_shape), self.filters],
                                    initializer=self.kernel_initializer, regularizer=self.kernel_regularizer)
    self.bias = self.add_variable("bias", [self.filters], initializer=self.bias_initializer,
                                  regularizer=self.bias_regularizer) if self.use_bias else None
    self.built = True

  def _to_cf(self, x):
    return x.permute(0, x.dim() - 1, *_range(1, x.dim() - 1)) if self.data_format == "channels_last" else x

  def _from_cf(self, y):
    return y.permute(0, *_range(2, y.dim()), 1) if self.data_format == "channels_last" else y

  def call(self, inputs):
    nd = self.ND

    def f(x, w):
      x, w = _t(x), _t(w).to(_t(x).dtype)
      wt = w.permute(nd + 1, nd, *_range(nd))              # [*k, in, out] -> [out, in, *k]
      return self._from_cf(_conv_nd(self._to_cf(x), wt, self.s, self.padding, self.d, nd))
    y = Tensor(f, (inputs, self.kernel), name=self.name)
    if self.use_bias:
      y = _bias_add(y, cast(self.bias, y.dtype), "NHWC" if self.data_format == "channels_last" else "NCHW")
    return self.activation(y) if self.activation is not None else y


class Conv1D(_Conv):
  ND = 1


class Conv2D(_Conv):
  ND = 2


class SeparableConv1D(_Conv):
  ND = 1

  def __init__(self, filters, kernel_size, depth_multiplier=1, depthwise_initializer=None,
               pointwise_initializer=None, depthwise_regularizer=None, pointwise_regularizer=None, **kwargs):
    super(SeparableConv1D, self).__init__(filters, kernel_size, **kwargs)
    self.dm = int(depth_multiplier)
    self.dw_init, self.pw_init = depthwise_initializer, pointwise_initializer
    self.dw_reg, self.pw_reg = depthwise_regularizer, pointwise_regularizer

  @property
  def name(self):
    return self._given_name or "separable_conv1d"

  def build(self, input_shape):
    cin = self._cin(input_shape)
    self.depthwise_kernel = self.add_variable("depthwise_kernel", list(self.k) + [cin, self.dm],
                                              initializer=self.dw_init, regularizer=self.dw_reg)
    self.pointwise_kernel = self.add_variable("pointwise_kernel", [1, cin * self.dm, self.filters],
                                              initializer=self.pw_init, regularizer=self.pw_reg)
    self.bias = self.add_variable("bias", [self.filters], initializer=self.bias_initializer,
                                  regularizer=self.bias_regularizer) if self.use_bias else None
    self.built = True

  def call(self, inputs):
    def f(x, dw, pw):
      x = _t(x)
      dw, pw = _t(dw).to(x.dtype), _t(pw).to(x.dtype)
      cin = dw.shape[1]
      # depthwise: [k, in, mult] -> [in * mult, 1, k] (output channel = in * mult + m, as TF)
      wd = dw.permute(1, 2, 0).reshape(cin * self.dm, 1, dw.shape[0])
      h = _conv_nd(self._to_cf(x), wd, self.s, self.padding, self.d, 1, groups=cin)
      return self._from_cf(torch.nn.functional.conv1d(h, pw.permute(2, 1, 0)))
    y = Tensor(f, (inputs, self.depthwise_kernel, self.pointwise_kernel), name="separable_conv1d")
    if self.use_bias:
      y = _bias_add(y, cast(self.bias, y.dtype), "NHWC" if self.data_format == "channels_last" else "NCHW")
    return self.activation(y) if self.activation is not None else y


def _fn_layer(cls):
  def fn(inputs, *args, **kwargs):
    kwargs.pop("reuse", None)
    return cls(*args, **kwargs)(inputs)
  return fn


class BatchNormalization(Layer):
  """tf.layers.BatchNormalization (TF 1.13, virtual_batch_size None, renorm False). Training: batch mean and
  BIASED variance normalise; the moving averages move by `momentum` inside UPDATE_OPS assign ops — with the
  biased variance on the non-fused path (inputs that are not 4-D) and with Bessel's correction on the fused path
  (4-D inputs: fused_batch_norm hands back the unbiased variance for the running average)."""

  def __init__(self, axis=-1, momentum=0.99, epsilon=1e-3, center=True, scale=True, beta_initializer=None,
               gamma_initializer=None, moving_mean_initializer=None, moving_variance_initializer=None,
               beta_regularizer=None, gamma_regularizer=None, beta_constraint=None, gamma_constraint=None,
               renorm=False, renorm_clipping=None, renorm_momentum=0.99, fused=None, trainable=True,
               virtual_batch_size=None, adjustment=None, name=None, **kwargs):
    super(BatchNormalization, self).__init__(trainable=trainable, name=name)
    self.axis, self.momentum, self.epsilon, self.center, self.scale = axis, momentum, epsilon, center, scale
    self.beta_regularizer, self.gamma_regularizer, self.fused = beta_regularizer, gamma_regularizer, fused

  @property
  def name(self):
    return self._given_name or "batch_normalization"

  def build(self, input_shape):
    nd = len(input_shape)
    self._ax = self.axis % nd
    c = int(input_shape[self._ax])
    if self.fused is None:
      self.fused = (nd == 4)
    self.gamma = self.add_variable("gamma", [c], dtype=float32, initializer=ones_initializer(),
                                   regularizer=self.gamma_regularizer) if self.scale else None
    self.beta = self.add_variable("beta", [c], dtype=float32, initializer=zeros_initializer(),
                                  regularizer=self.beta_regularizer) if self.center else None
    self.moving_mean = self.add_variable("moving_mean", [c], dtype=float32, initializer=zeros_initializer(),
                                         trainable=False)
    self.moving_variance = self.add_variable("moving_variance", [c], dtype=float32,
                                             initializer=ones_initializer(), trainable=False)
    self.built = True

  def call(self, inputs, training=False):
    nd = len(inputs.get_shape())
    red = [i for i in _range(nd) if i != self._ax]
    bshape = [1] * nd
    bshape[self._ax] = -1
    if isinstance(training, Tensor):
      training = _PYBOOL(training._value)

    def norm_(x, mean, var, g, b):
      x = _t(x)
      xf = x.to(torch.float32)
      y = (xf - mean.reshape(bshape)) * torch.rsqrt(var.reshape(bshape) + self.epsilon)
      if g is not None:
        y = y * _t(g).reshape(bshape)
      if b is not None:
        y = y + _t(b).reshape(bshape)
      return y.to(x.dtype)
    if not training:
      return Tensor(lambda x, m, v, g, b: norm_(x, _t(m), _t(v), g, b),
                    (inputs, self.moving_mean, self.moving_variance, self.gamma, self.beta), name="batchnorm")
    mean = Tensor(lambda x: _t(x).to(torch.float32).mean(dim=red), (inputs,), name="bn_mean")
    var = Tensor(lambda x, m: ((_t(x).to(torch.float32) - m.reshape(bshape)) ** 2).mean(dim=red), (inputs, mean),
                 name="bn_var")
    n = 1
    for i in red:
      n *= int(inputs._value.shape[i])
    var_avg = var * (float(n) / max(n - 1, 1)) if self.fused else var
    d = 1.0 - self.momentum
    add_to_collection(GraphKeys.UPDATE_OPS, assign_sub(self.moving_mean, (self.moving_mean - stop_gradient(mean)) * d))
    add_to_collection(GraphKeys.UPDATE_OPS,
                      assign_sub(self.moving_variance, (self.moving_variance - stop_gradient(var_avg)) * d))
    return Tensor(lambda x, m, v, g, b: norm_(x, m, v, g, b), (inputs, mean, var, self.gamma, self.beta),
                  name="batchnorm")


def _batch_normalization(inputs, axis=-1, momentum=0.99, epsilon=1e-3, center=True, scale=True, training=False,
                         name=None, reuse=None, **kwargs):
  kwargs.pop("trainable", None)
  layer = BatchNormalization(axis=axis, momentum=momentum, epsilon=epsilon, center=center, scale=scale, name=name,
                             **kwargs)
  return layer(inputs, training=training)


def _flatten_layer(inputs, name=None):
  return Tensor(lambda x: _t(x).reshape(_t(x).shape[0], -1), (inputs,), name="flatten")


layers = types.SimpleNamespace(
    Layer=Layer, Dense=Dense, dense=_dense, Conv1D=Conv1D, Conv2D=Conv2D, SeparableConv1D=SeparableConv1D,
    conv1d=_fn_layer(Conv1D), conv2d=_fn_layer(Conv2D), separable_conv1d=_fn_layer(SeparableConv1D),
    BatchNormalization=BatchNormalization, batch_normalization=_batch_normalization, flatten=_flatten_layer,
    dropout=lambda inputs, rate=0.5, noise_shape=None, seed=None, training=False, name=None:
    (_dropout(inputs, keep_prob=1.0 - rate, noise_shape=noise_shape) if training else identity(inputs)))

class _CudnnRNN(Layer):
  """tf.contrib.cudnn_rnn.CudnnGRU / CudnnLSTM: time-major input [T, B, I], NO sequence lengths (the whole padded
  length is run), num_layers stacked, optionally bidirectional with the two directions concatenated after every
  layer. Built on torch.nn.GRU / LSTM, which implement the same cuDNN cell equations; the parameters are exposed as
  variables under torch's names (TensorFlow keeps one opaque buffer)."""
  KIND = None

  def __init__(self, num_layers, num_units, input_mode="linear_input", direction="unidirectional", dropout=0.0,
               seed=None, dtype=None, kernel_initializer=None, bias_initializer=None, name=None):
    super(_CudnnRNN, self).__init__(name=name)
    self.num_layers, self.num_units = int(num_layers), int(num_units)
    self.bidir = direction == "bidirectional"
    self.dropout = float(dropout)

  @property
  def name(self):
    return self._given_name or ("cudnn_" + self.KIND.lower())

  def build(self, input_shape):
    cls = {"GRU": torch.nn.GRU, "LSTM": torch.nn.LSTM}[self.KIND]
    self._mod = cls(int(input_shape[-1]), self.num_units, num_layers=self.num_layers, bidirectional=self.bidir)
    self._pnames = [n for n, _ in self._mod.named_parameters()]
    k = 1.0 / math.sqrt(self.num_units)
    self._pvars = [self.add_variable(n, list(p.shape), dtype=float32,
                                     initializer=random_uniform_initializer(-k, k))
                   for n, p in self._mod.named_parameters()]
    self.built = True

  def call(self, inputs, initial_state=None, training=True):
    if self.dropout != 0.0:
      raise NotImplementedError("inter-layer dropout of the cuDNN RNN is not restated (use dropout 0)")
    names, mod = self._pnames, self._mod

    def f(x, ps):
      out, state = torch.func.functional_call(mod, {n: _t(p) for n, p in zip(names, ps)}, (_t(x),))
      return out
    out = Tensor(f, (inputs, list(self._pvars)), name=self.name)
    return out, None


class _CudnnGRU(_CudnnRNN):
  KIND = "GRU"


class _CudnnLSTM(_CudnnRNN):
  KIND = "LSTM"


keras = types.SimpleNamespace(
    initializers=types.SimpleNamespace(Zeros=zeros_initializer, Ones=ones_initializer,
                                       glorot_uniform=glorot_uniform_initializer,
                                       RandomNormal=random_normal_initializer),
    layers=types.SimpleNamespace(Layer=Layer, Dense=Dense))
initializers = types.SimpleNamespace(zeros=zeros_initializer, ones=ones_initializer,
                                     random_normal=random_normal_initializer,
                                     random_uniform=random_uniform_initializer,
                                     glorot_uniform=glorot_uniform_initializer,
                                     variance_scaling=variance_scaling_initializer,
                                     constant=constant_initializer)

# ----------------------------------------------------------------------------------------------- tf.contrib


def _l2_regularizer(scale, scope=None):
  if float(scale) == 0.0:
    return lambda _: None
  return lambda w: Tensor(lambda v: (_t(v) ** 2).sum() / 2.0 * _t(scale, like=_t(v)), (w,), name="l2_regularizer")


def _layer_norm(inputs, center=True, scale=True, activation_fn=None, reuse=None, variables_collections=None,
                outputs_collections=None, trainable=True, begin_norm_axis=1, begin_params_axis=-1, scope=None):
  """tf.contrib.layers.layer_norm: moments over axes [begin_norm_axis, rank), beta / gamma over the last axis,
  variance epsilon 1e-12."""
  with variable_scope(scope, default_name="LayerNorm"):
    nd = len(inputs.get_shape())
    c = int(inputs.get_shape()[-1])
    beta = get_variable("beta", [c], dtype=float32, initializer=zeros_initializer()) if center else None
    gamma = get_variable("gamma", [c], dtype=float32, initializer=ones_initializer()) if scale else None
    axes = list(_range(begin_norm_axis % nd, nd))

    def f(x, g, b):
      x = _t(x)
      m = x.mean(dim=axes, keepdim=True)
      v = ((x - m) ** 2).mean(dim=axes, keepdim=True)
      y = (x - m) * torch.rsqrt(v + 1e-12)
      if g is not None:
        y = y * _t(g).to(x.dtype)
      if b is not None:
        y = y + _t(b).to(x.dtype)
      return y
    y = Tensor(f, (inputs, gamma, beta), name="layer_norm")
    return activation_fn(y) if activation_fn else y


def _instance_norm(inputs, center=True, scale=True, epsilon=1e-6, activation_fn=None, param_initializers=None,
                   reuse=None, variables_collections=None, outputs_collections=None, trainable=True,
                   data_format="NHWC", scope=None):
  """tf.contrib.layers.instance_norm: moments over every axis but batch and channels."""
  with variable_scope(scope, default_name="InstanceNorm"):
    nd = len(inputs.get_shape())
    cax = nd - 1 if data_format == "NHWC" else 1
    c = int(inputs.get_shape()[cax])
    beta = get_variable("beta", [c], dtype=float32, initializer=zeros_initializer()) if center else None
    gamma = get_variable("gamma", [c], dtype=float32, initializer=ones_initializer()) if scale else None
    axes = [i for i in _range(1, nd) if i != cax]
    bs = [1] * nd
    bs[cax] = -1

    def f(x, g, b):
      x = _t(x)
      m = x.mean(dim=axes, keepdim=True)
      v = ((x - m) ** 2).mean(dim=axes, keepdim=True)
      y = (x - m) * torch.rsqrt(v + epsilon)
      if g is not None:
        y = y * _t(g).to(x.dtype).reshape(bs)
      if b is not None:
        y = y + _t(b).to(x.dtype).reshape(bs)
      return y
    y = Tensor(f, (inputs, gamma, beta), name="instance_norm")
    return activation_fn(y) if activation_fn else y


def _apply_regularization(regularizer, weights_list=None):
  ws = weights_list if weights_list is not None else trainable_variables()
  ls = [regularizer(w) for w in ws]
  ls = [l for l in ls if l is not None]
  return add_n(ls) if ls else constant(0.0)


contrib = types.SimpleNamespace(
    layers=types.SimpleNamespace(l2_regularizer=_l2_regularizer, layer_norm=_layer_norm,
                                 instance_norm=_instance_norm, apply_regularization=_apply_regularization,
                                 xavier_initializer=lambda uniform=True, seed=None, dtype=None:
                                 glorot_uniform_initializer() if uniform else glorot_normal_initializer(),
                                 variance_scaling_initializer=lambda factor=2.0, mode="FAN_IN", uniform=False,
                                 seed=None, dtype=None: variance_scaling_initializer(
                                     factor, mode.lower(), "uniform" if uniform else "normal")),
    opt=types.SimpleNamespace(), seq2seq=types.SimpleNamespace(), rnn=types.SimpleNamespace(),
    cudnn_rnn=types.SimpleNamespace(), framework=types.SimpleNamespace(nest=None))

contrib.cudnn_rnn = types.SimpleNamespace(CudnnGRU=_CudnnGRU, CudnnLSTM=_CudnnLSTM)
nn.depthwise_conv2d = _depthwise_conv2d

losses = types.SimpleNamespace(
    get_regularization_losses=lambda scope=None: get_collection(GraphKeys.REGULARIZATION_LOSSES, scope),
    get_regularization_loss=lambda scope=None, name=None: (
        add_n(get_collection(GraphKeys.REGULARIZATION_LOSSES, scope))
        if get_collection(GraphKeys.REGULARIZATION_LOSSES, scope) else constant(0.0)),
    Reduction=types.SimpleNamespace(NONE="none", SUM="weighted_sum", MEAN="weighted_mean"))


def _losses_softmax_xent(onehot_labels, logits, weights=1.0, label_smoothing=0, scope=None, loss_collection=None,
                         reduction="weighted_sum_by_nonzero_weights"):
  """tf.losses.softmax_cross_entropy: smoothed labels = onehot (1 - s) + s / classes; Reduction.NONE returns the
  weighted per-example losses."""
  n = int(onehot_labels.get_shape()[-1])
  lab = cast(onehot_labels, logits.dtype)
  if label_smoothing:
    lab = lab * (1.0 - label_smoothing) + label_smoothing / n
  per = _soft_xent_v2(labels=lab, logits=logits)
  per = per * weights
  if reduction == "none":
    return per
  if reduction == "weighted_sum":
    return reduce_sum(per)
  nz = reduce_sum(cast(not_equal(weights, 0.0), per.dtype)) if isinstance(weights, Tensor) else \
      cast(size(per), per.dtype)
  return reduce_sum(per) / nz


losses.softmax_cross_entropy = _losses_softmax_xent


def _weighted_loss(per_element, weights, reduction):
  """tf.losses.compute_weighted_loss, default reduction SUM_BY_NONZERO_WEIGHTS: sum(losses * weights) divided by
  the number of elements whose (broadcast) weight is non-zero; 0 when there is none."""
  def f(l, w):
    l = _t(l).to(torch.float32)
    w = _t(w, like=l).to(torch.float32)
    wl = l * w
    if reduction == "none":
      return wl
    if reduction == "weighted_sum":
      return wl.sum()
    present = (torch.broadcast_to(w, l.shape) != 0).to(torch.float32).sum()
    return torch.where(present > 0, wl.sum() / torch.clamp(present, min=1.0), torch.zeros(()))
  return Tensor(f, (per_element, weights), name="weighted_loss")


def _mean_squared_error(labels, predictions, weights=1.0, scope=None, loss_collection=None,
                        reduction="weighted_sum_by_nonzero_weights"):
  return _weighted_loss(squared_difference(cast(predictions, float32), cast(labels, float32)), weights, reduction)


def _absolute_difference(labels, predictions, weights=1.0, scope=None, loss_collection=None,
                         reduction="weighted_sum_by_nonzero_weights"):
  return _weighted_loss(abs(cast(predictions, float32) - cast(labels, float32)), weights, reduction)


losses.mean_squared_error, losses.absolute_difference = _mean_squared_error, _absolute_difference


def _sigmoid_xent(_sentinel=None, labels=None, logits=None, name=None):
  """max(x, 0) - x z + log(1 + exp(-|x|))"""
  return Tensor(lambda z, x: torch.clamp(_t(x), min=0) - _t(x) * _t(z).to(_t(x).dtype) +
                torch.log1p(torch.exp(-_t(x).abs())), (labels, logits), name="sigmoid_cross_entropy_with_logits")


nn.sigmoid_cross_entropy_with_logits = _sigmoid_xent

# ----------------------------------------------------------------------------------------------- tf.train

_GLOBAL_STEP = []


def _get_or_create_global_step(graph=None):
  gs = get_collection(GraphKeys.GLOBAL_STEP)
  if gs:
    return gs[0]
  saved = list(_SCOPES)
  _SCOPES[:] = [VariableScope("")]
  try:
    v = Variable(torch.zeros((), dtype=torch.int64), trainable=False, name="global_step")
  finally:
    _SCOPES[:] = saved
  add_to_collection(GraphKeys.GLOBAL_STEP, v)
  return v


def _exponential_decay(learning_rate, global_step, decay_steps, decay_rate, staircase=False, name=None):
  """lr * rate ^ (step / decay_steps) (floor of the exponent when staircase)."""
  def f(lr, gs, ds, dr):
    lr = _t(lr).to(torch.float32)
    p = _t(gs).to(torch.float32) / _t(ds).to(torch.float32)
    if staircase:
      p = torch.floor(p)
    return lr * torch.pow(_t(dr).to(torch.float32), p)
  return Tensor(f, (learning_rate, global_step, decay_steps, decay_rate), name="exponential_decay")


def _polynomial_decay(learning_rate, global_step, decay_steps, end_learning_rate=0.0001, power=1.0, cycle=False,
                      name=None):
  """(lr - end) (1 - min(step, decay_steps) / decay_steps) ^ power + end; cycle: decay_steps grows to the next
  multiple that holds the step."""
  def f(lr, gs, ds, end, pw):
    lr, end, pw = (_t(v).to(torch.float32) for v in (lr, end, pw))
    gs, ds = _t(gs).to(torch.float32), _t(ds).to(torch.float32)
    if cycle:
      mult = torch.where(gs == 0, torch.ones_like(gs), torch.ceil(gs / ds))
      ds = ds * mult
    else:
      gs = torch.minimum(gs, ds)
    return (lr - end) * torch.pow(1.0 - gs / ds, pw) + end
  return Tensor(f, (learning_rate, global_step, decay_steps, end_learning_rate, power), name="polynomial_decay")


def _piecewise_constant(x, boundaries, values, name=None):
  """values[0] for x <= boundaries[0], values[i] for boundaries[i-1] < x <= boundaries[i], values[-1] after."""
  def f(xv, bs, vs):
    xv = _t(xv)
    out = _t(vs[-1])
    for i in _range(len(bs) - 1, -1, -1):
      out = torch.where(xv <= _t(bs[i], like=xv), _t(vs[i], like=out), out)
    return out
  return Tensor(f, (x, list(boundaries), list(values)), name="piecewise_constant")


def _cosine_decay(learning_rate, global_step, decay_steps, alpha=0.0, name=None):
  def f(lr, gs, ds):
    lr = _t(lr).to(torch.float32)
    gs, ds = _t(gs).to(torch.float32), _t(ds).to(torch.float32)
    frac = torch.minimum(gs, ds) / ds
    cd = 0.5 * (1.0 + torch.cos(math.pi * frac))
    return lr * ((1 - alpha) * cd + alpha)
  return Tensor(f, (learning_rate, global_step, decay_steps), name="cosine_decay")


class Optimizer(object):
  GATE_NONE, GATE_OP, GATE_GRAPH = 0, 1, 2

  def __init__(self, use_locking=False, name="Optimizer"):
    self._name = name
    self._use_locking = use_locking
    self._slots = {}

  def get_name(self):
    return self._name

  LAST_COMPUTED = None      # (fixture generator's tap) the grads_and_vars of the latest compute_gradients call

  def compute_gradients(self, loss, var_list=None, gate_gradients=1, aggregation_method=None,
                        colocate_gradients_with_ops=False, grad_loss=None):
    vs = list(var_list) if var_list is not None else trainable_variables()
    gs = gradients(loss, vs, grad_ys=None if grad_loss is None else [grad_loss])
    Optimizer.LAST_COMPUTED = list(zip(gs, vs))
    return list(zip(gs, vs))

  def _zeros_slot(self, var, slot_name, op_name=None):
    key = (id(var), slot_name)
    if key not in self._slots:
      saved = list(_SCOPES)
      _SCOPES[:] = [VariableScope("")]
      try:
        self._slots[key] = Variable(torch.zeros_like(var._var.detach()), trainable=False,
                                    name=var.name.split(":")[0] + "/" + (op_name or self._name))
      finally:
        _SCOPES[:] = saved
    return self._slots[key]

  def get_slot(self, var, name):
    return self._slots.get((id(var), name))

  def _lr_t(self):
    lr = self._lr() if callable(self._lr) else self._lr
    return lr

  def _apply_dense(self, grad, var):
    raise NotImplementedError

  def apply_gradients(self, grads_and_vars, global_step=None, name=None):
    ups = []
    for g, v in grads_and_vars:
      if g is None:
        continue
      if isinstance(g, IndexedSlices):
        g = Tensor(lambda vals, idx, ref=v: torch.zeros_like(ref._var.detach()).index_add(
            0, _t(idx).long().reshape(-1), _t(vals).reshape(-1, *ref._var.shape[1:]).to(ref._var.dtype)),
            (g.values, g.indices), name="densify")
      ups.append(self._apply_dense(g, v))
    with control_dependencies(ups):
      fin = assign_add(global_step, 1) if global_step is not None else no_op()
    return group(fin, *ups)

  def minimize(self, loss, global_step=None, var_list=None, **kw):
    return self.apply_gradients(self.compute_gradients(loss, var_list), global_step)


class GradientDescentOptimizer(Optimizer):
  def __init__(self, learning_rate, use_locking=False, name="GradientDescent"):
    super(GradientDescentOptimizer, self).__init__(use_locking, name)
    self._lr = learning_rate

  def _apply_dense(self, grad, var):
    return assign_sub(var, cast(self._lr_t(), var.dtype) * cast(grad, var.dtype))


class MomentumOptimizer(Optimizer):
  """accum = momentum * accum + grad; var -= lr * accum (Nesterov: var -= lr * (grad + momentum * accum))."""

  def __init__(self, learning_rate, momentum, use_locking=False, name="Momentum", use_nesterov=False):
    super(MomentumOptimizer, self).__init__(use_locking, name)
    self._lr, self._momentum, self._nesterov = learning_rate, momentum, use_nesterov

  def _apply_dense(self, grad, var):
    acc = self._zeros_slot(var, "momentum")
    new_acc = assign(acc, cast(self._momentum, var.dtype) * acc + cast(grad, var.dtype))
    lr = cast(self._lr_t(), var.dtype)
    if self._nesterov:
      return assign_sub(var, lr * (cast(grad, var.dtype) + cast(self._momentum, var.dtype) * new_acc))
    return assign_sub(var, lr * new_acc)


class AdamOptimizer(Optimizer):
  """lr_t = lr sqrt(1 - b2^t) / (1 - b1^t); m, v moving averages; var -= lr_t m / (sqrt(v) + eps)."""

  def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, use_locking=False, name="Adam"):
    super(AdamOptimizer, self).__init__(use_locking, name)
    self._lr, self._b1, self._b2, self._eps = learning_rate, beta1, beta2, epsilon
    self._t = None

  def apply_gradients(self, grads_and_vars, global_step=None, name=None):
    if self._t is None:
      saved = list(_SCOPES)
      _SCOPES[:] = [VariableScope("")]
      try:
        self._t = Variable(torch.zeros((), dtype=torch.float32), trainable=False, name=self._name + "/t")
      finally:
        _SCOPES[:] = saved
    self._t_next = self._t + 1.0
    up = super(AdamOptimizer, self).apply_gradients(grads_and_vars, global_step, name)
    with control_dependencies([up]):
      return group(assign(self._t, self._t_next))

  def _apply_dense(self, grad, var):
    m, v = self._zeros_slot(var, "m", self._name), self._zeros_slot(var, "v", self._name + "_1")
    t = self._t_next
    g = cast(grad, var.dtype)
    lr_t = cast(self._lr_t(), var.dtype) * sqrt(1.0 - pow(self._b2, t)) / (1.0 - pow(self._b1, t))
    m_t = assign(m, m * self._b1 + g * (1.0 - self._b1))
    v_t = assign(v, v * self._b2 + g * g * (1.0 - self._b2))
    return assign_sub(var, lr_t * m_t / (sqrt(v_t) + self._eps))


def _unsupported_optimizer(name):
  class _U(Optimizer):
    def __init__(self, *a, **k):
      raise NotImplementedError("tf.train.%s is not restated in the shim" % name)
  _U.__name__ = name
  return _U


train = types.ModuleType("tensorflow.train")
train.__dict__.update(dict(
    get_or_create_global_step=_get_or_create_global_step, get_global_step=lambda graph=None: (
        get_collection(GraphKeys.GLOBAL_STEP) or [None])[0],
    create_global_step=_get_or_create_global_step,
    exponential_decay=_exponential_decay, polynomial_decay=_polynomial_decay,
    piecewise_constant=_piecewise_constant, cosine_decay=_cosine_decay,
    Optimizer=Optimizer, GradientDescentOptimizer=GradientDescentOptimizer, MomentumOptimizer=MomentumOptimizer,
    AdamOptimizer=AdamOptimizer, RMSPropOptimizer=_unsupported_optimizer("RMSPropOptimizer"),
    FtrlOptimizer=_unsupported_optimizer("FtrlOptimizer"), AdagradOptimizer=_unsupported_optimizer("AdagradOptimizer"),
    AdadeltaOptimizer=_unsupported_optimizer("AdadeltaOptimizer"),
    SessionRunHook=object, SessionRunArgs=lambda *a, **k: None, Saver=object))
contrib.opt.AdamWOptimizer = _unsupported_optimizer("AdamWOptimizer")
contrib.opt.LazyAdamOptimizer = AdamOptimizer       # dense gradients: LazyAdam == Adam

learn = types.SimpleNamespace()
app = types.SimpleNamespace(flags=types.SimpleNamespace(FLAGS=types.SimpleNamespace()), run=lambda *a, **k: None)
import unittest as _unittest


class _TestCase(_unittest.TestCase):
  """tf.test.TestCase: unittest + test_session + the array assertions (what the reference's *_test.py files use)."""

  def setUp(self):
    reset_default_graph()

  def test_session(self, graph=None, config=None, use_gpu=False, force_gpu=False):
    return Session()

  session = test_session
  cached_session = test_session

  def assertAllEqual(self, a, b, msg=None):             # noqa: N802
    np.testing.assert_array_equal(np.asarray(a), np.asarray(b), err_msg=msg or "")

  def assertAllClose(self, a, b, rtol=1e-6, atol=1e-6, msg=None):      # noqa: N802
    np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=rtol, atol=atol, err_msg=msg or "")

  def assertAllCloseAccordingToType(self, a, b, **kw):    # noqa: N802
    self.assertAllClose(a, b)


test = types.SimpleNamespace(TestCase=_TestCase, main=lambda *a, **k: _unittest.main(*a, **k),
                             is_gpu_available=lambda *a, **k: False)
gfile = types.SimpleNamespace()

softmax, log_softmax, relu, dropout = _softmax, _log_softmax, nn.relu, _dropout
assert_equal = lambda *a, **k: no_op()                   # noqa: E731  (tf.python.ops.check_ops)
assert_positive = assert_greater = assert_less_equal = assert_rank = assert_equal
with_same_shape = lambda old, new: new                   # noqa: E731  (contrib.framework tensor_util)

from . import rnn as _rnn                                # noqa: E402  (the recurrent part of the stand-in)
nn.rnn_cell = types.SimpleNamespace(
    RNNCell=_rnn.RNNCell, LSTMCell=_rnn.LSTMCell, BasicLSTMCell=_rnn.BasicLSTMCell, MultiRNNCell=_rnn.MultiRNNCell,
    GRUCell=_rnn.GRUCell, LSTMStateTuple=_rnn.LSTMStateTuple, ResidualWrapper=_rnn.ResidualWrapper, DropoutWrapper=_rnn.DropoutWrapper)
nn.dynamic_rnn, nn.bidirectional_dynamic_rnn = _rnn.dynamic_rnn, _rnn.bidirectional_dynamic_rnn
nn.embedding_lookup = embedding_lookup = _rnn.embedding_lookup
while_loop = _rnn.while_loop
contrib.rnn = types.SimpleNamespace(MultiRNNCell=_rnn.MultiRNNCell, ResidualWrapper=_rnn.ResidualWrapper,
                                    LSTMStateTuple=_rnn.LSTMStateTuple, DropoutWrapper=_rnn.DropoutWrapper,
                                    LSTMCell=_rnn.LSTMCell, BasicLSTMCell=_rnn.BasicLSTMCell, RNNCell=_rnn.RNNCell)
contrib.seq2seq = types.SimpleNamespace(
    Decoder=_rnn.Decoder, Helper=_rnn.Helper, TrainingHelper=_rnn.TrainingHelper, BasicDecoder=_rnn.BasicDecoder,
    BasicDecoderOutput=_rnn.BasicDecoderOutput, dynamic_decode=_rnn.dynamic_decode, tile_batch=tile_batch,
    gather_tree=gather_tree)

slice = slice_          # noqa: A001  (the TF names; Python's own slice / range are not used below this line)
range = range_          # noqa: A001
tuple = tuple_          # noqa: A001


# ----------------------------------------------------------------------------------------------- install


def install():
  """Registers this module as `tensorflow` (+ the handful of `tensorflow.python...` sub-module names the
  reference imports) in sys.modules. Only tests/golden/make_ref_exec.py and the tests of the shim call this."""
  me = sys.modules[__name__]
  mods = {"tensorflow": me, "tensorflow.train": train}

  def sub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    mods[name] = m
    return m
  sub("tensorflow.python")
  sub("tensorflow.python.framework")
  sub("tensorflow.python.framework.ops", convert_to_tensor=convert_to_tensor, Tensor=Tensor,
      IndexedSlices=IndexedSlices, GraphKeys=GraphKeys, colocate_with=colocate_with,
      control_dependencies=control_dependencies, name_scope=name_scope, get_default_graph=get_default_graph)
  sub("tensorflow.python.framework.dtypes", float32=float32, float16=float16, int32=int32, int64=int64, bool=bool,
      as_dtype=as_dtype, DType=DType)
  sub("tensorflow.python.ops")
  sub("tensorflow.python.training")
  sub("tensorflow.python.training.optimizer", Optimizer=Optimizer)
  sub("tensorflow.python.training.training_ops")
  sub("tensorflow.python.client")
  sub("tensorflow.python.client.device_lib", list_local_devices=lambda: [])
  sub("tensorflow.python.util")
  sub("tensorflow.python.util.nest", flatten=_rnn.flatten, map_structure=_rnn.map_structure,
      is_sequence=_rnn.is_sequence, pack_sequence_as=_rnn.pack_sequence_as,
      assert_same_structure=_rnn.assert_same_structure)
  for opsmod in ("array_ops", "check_ops", "clip_ops", "functional_ops", "init_ops", "math_ops", "nn_ops",
                 "random_ops", "control_flow_ops", "state_ops"):
    mods["tensorflow.python.ops." + opsmod] = me         # one namespace: every op lives in this module
  me.with_dependencies = lambda deps, out, name=None: _with_deps(deps, out)
  me.Zeros, me.Ones = zeros_initializer, ones_initializer
  sub("tensorflow.python.ops.variable_scope", get_variable=get_variable, variable_scope=variable_scope,
      get_variable_scope=get_variable_scope, AUTO_REUSE=AUTO_REUSE, VariableScope=VariableScope)
  sub("tensorflow.python.ops.rnn_cell_impl", RNNCell=_rnn.RNNCell, LayerRNNCell=_rnn.RNNCell,
      assert_like_rnncell=_rnn.assert_like_rnncell, _zero_state_tensors=_rnn._zero_state_tensors,
      LSTMStateTuple=_rnn.LSTMStateTuple, LSTMCell=_rnn.LSTMCell, MultiRNNCell=_rnn.MultiRNNCell)
  sub("tensorflow.python.ops.rnn_cell", ResidualWrapper=_rnn.ResidualWrapper, DropoutWrapper=_rnn.DropoutWrapper,
      LSTMCell=_rnn.LSTMCell, MultiRNNCell=_rnn.MultiRNNCell, RNNCell=_rnn.RNNCell)
  sub("tensorflow.python.ops.tensor_array_ops", TensorArray=_rnn.TensorArray)
  sub("tensorflow.python.framework.tensor_shape", TensorShape=TensorShape, Dimension=Dimension,
      as_shape=lambda s: s if isinstance(s, TensorShape) else TensorShape(s))
  sub("tensorflow.python.layers.convolutional", Conv1D=Conv1D, Conv2D=Conv2D)
  sub("tensorflow.contrib.framework")
  sub("tensorflow.contrib.framework.python")
  sub("tensorflow.contrib.framework.python.framework")
  sub("tensorflow.contrib.framework.python.framework.tensor_util", with_same_shape=with_same_shape)
  sub("tensorflow.contrib.rnn", **vars(contrib.rnn))
  sub("tensorflow.contrib.seq2seq", **vars(contrib.seq2seq))
  sub("tensorflow.contrib.seq2seq.python")
  sub("tensorflow.contrib.seq2seq.python.ops")
  sub("tensorflow.contrib.seq2seq.python.ops.decoder", Decoder=_rnn.Decoder, dynamic_decode=_rnn.dynamic_decode,
      _transpose_batch_time=_rnn._transpose_batch_time)
  sub("tensorflow.contrib.seq2seq.python.ops.beam_search_ops", gather_tree=gather_tree)
  sub("tensorflow.python.framework.tensor_util", constant_value=constant_value)
  mods["tensorflow.python.ops.embedding_ops"] = me
  sub("tensorflow.contrib.seq2seq.python.ops.helper", Helper=_rnn.Helper, TrainingHelper=_rnn.TrainingHelper)
  sub("tensorflow.contrib")
  sub("tensorflow.contrib.cudnn_rnn", CudnnGRU=_CudnnGRU, CudnnLSTM=_CudnnLSTM)
  sub("tensorflow.contrib.cudnn_rnn.python")
  sub("tensorflow.contrib.cudnn_rnn.python.ops")
  sub("tensorflow.contrib.cudnn_rnn.python.ops.cudnn_rnn_ops", CUDNN_RNN_UNIDIRECTION="unidirectional",
      CUDNN_RNN_BIDIRECTION="bidirectional")
  sub("tensorflow.python.layers")
  sub("tensorflow.python.layers.base", Layer=Layer)
  sub("tensorflow.python.layers.core", Dense=Dense)
  for k, m in mods.items():
    sys.modules[k] = m
  for k, m in mods.items():
    if "." in k:
      parent, _, leaf = k.rpartition(".")
      if parent in mods and parent != "tensorflow":
        setattr(mods[parent], leaf, m)
  me.python = mods["tensorflow.python"]
  return me


def _with_deps(deps, out):
  with control_dependencies(deps):
    return identity(out)


def _map_structure(f, *structs):
  s0 = structs[0]
  if isinstance(s0, (list, _tuple)):
    return type(s0)(_map_structure(f, *[s[i] for s in structs]) for i in _range(len(s0)))
  if isinstance(s0, dict):
    return {k: _map_structure(f, *[s[k] for s in structs]) for k in s0}
  return f(*structs)

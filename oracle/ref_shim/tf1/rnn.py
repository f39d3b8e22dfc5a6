"""TEST INFRASTRUCTURE ONLY — second part of the TensorFlow-1 stand-in (see __init__.py): the recurrent pieces of the
TF LIBRARY that the reference's RNN / attention code is written against, restated on the same node machinery:

  * tf.python.util.nest (flatten / pack_sequence_as / map_structure over tuples, namedtuples, lists, dicts)
  * tf.nn.rnn_cell: RNNCell, LSTMStateTuple, LSTMCell / BasicLSTMCell (gate order i, j, f, o; forget_bias inside the
    sigmoid), MultiRNNCell ('cell_%d' scopes), ResidualWrapper, DropoutWrapper (keep probability 1 only)
  * tf.nn.dynamic_rnn / bidirectional_dynamic_rnn: the cell is CALLED ONCE (as inside tf.while_loop) and the recorded
    step graph is replayed per time step; past a sample's sequence_length the output is zero and the state is copied
  * tf.contrib.seq2seq: Decoder / Helper bases, TrainingHelper, BasicDecoder(+Output), dynamic_decode(impute_finished,
    maximum_iterations) — again one traced step, replayed until every sample has finished

None of the reference's logic lives here: its attention mechanisms, AttentionWrapper, GNMT multi-cell, Tacotron decoder
and helpers run from their own files (parts/rnns/attention_wrapper.py, parts/rnns/gnmt.py, parts/tacotron/*.py).
"""
import collections

import torch

from . import (Tensor, Variable, Layer, TensorShape, variable_scope, get_variable, zeros_initializer, _t, _PYBOOL,
               _range, _tuple, as_dtype, float32, int32, concat, matmul, sigmoid, tanh, split, identity, no_op, cast,
               convert_to_tensor)

# ------------------------------------------------------------------------------------------------------- nest


def is_sequence(x):
  return isinstance(x, (list, _tuple, dict)) and not isinstance(x, str)


def flatten(x):
  if isinstance(x, dict):
    out = []
    for k in sorted(x):
      out.extend(flatten(x[k]))
    return out
  if isinstance(x, (list, _tuple)):
    out = []
    for v in x:
      out.extend(flatten(v))
    return out
  return [x]


def _pack(structure, it):
  if isinstance(structure, dict):
    return {k: _pack(structure[k], it) for k in sorted(structure)}
  if isinstance(structure, _tuple) and hasattr(structure, "_fields"):
    return type(structure)(*[_pack(v, it) for v in structure])
  if isinstance(structure, (list, _tuple)):
    return type(structure)(_pack(v, it) for v in structure)
  return next(it)


def pack_sequence_as(structure, flat_sequence):
  return _pack(structure, iter(list(flat_sequence)))


def map_structure(fn, *structures, **kwargs):
  flats = [flatten(s) for s in structures]
  return pack_sequence_as(structures[0], [fn(*vals) for vals in zip(*flats)])


def assert_same_structure(a, b, check_types=True):
  if len(flatten(a)) != len(flatten(b)):
    raise ValueError("The two structures don't have the same number of elements")


# ------------------------------------------------------------------------------------------------------- tracing


class _Slot(Tensor):
  """A loop variable of a traced step: bound to a value by the replay, never evaluated on its own."""

  def __init__(self, value, name="loop_var"):
    super(_Slot, self).__init__(None, (), name=name, build_value=value)
    self._ctrl = []

  def _eval(self, ev):
    raise RuntimeError("loop variable %s evaluated outside its loop" % self.name)


def _children(t):
  """Every tensor the evaluation of `t` may ask for."""
  if isinstance(t, Variable):
    return []
  if hasattr(t, "_deps"):
    return list(t._deps()) + list(t._ctrl)
  from . import _CondOut, _Grad
  if isinstance(t, _CondOut):
    return [x for x in (t._pred, t._a, t._b) if isinstance(x, Tensor)] + list(t._ctrl)
  if isinstance(t, _Grad):
    sh = t._shared
    return list(sh.ys) + list(sh.xs) + [g for g in sh.grad_ys if isinstance(g, Tensor)] + list(t._ctrl)
  found = []

  def walk(o):
    if isinstance(o, Tensor):
      found.append(o)
    elif isinstance(o, (list, _tuple)):
      for v in o:
        walk(v)
    elif isinstance(o, dict):
      for v in o.values():
        walk(v)
    elif isinstance(o, slice):
      walk(o.start), walk(o.stop), walk(o.step)
  walk(t._args)
  walk(t._kwargs)
  return found + list(t._ctrl)


class Traced(object):
  """slots -> outputs, recorded once; run(ev_outer, values) replays the recorded nodes with the slots bound to
  `values`. Nodes that do not depend on a slot (weights, the prepared attention memory ...) are taken from the
  enclosing evaluation, so they are computed once per Session.run and shared by every step."""

  def __init__(self, slots, outputs):
    self.slots, self.outputs = list(slots), list(outputs)
    self._dep = {id(s): True for s in self.slots}

  def _depends(self, t):
    k = id(t)
    if k not in self._dep:
      self._dep[k] = False                                   # (cycles do not occur; the entry guards re-entry)
      self._dep[k] = any(self._depends(c) for c in _children(t))
    return self._dep[k]

  def run(self, ev_outer, values):
    local = {id(s): v for s, v in zip(self.slots, values)}

    def ev(t):
      if isinstance(t, Variable):
        return t._var
      k = id(t)
      if k in local:
        return local[k]
      if not self._depends(t):
        return ev_outer(t)
      local[k] = t._eval(ev)
      return local[k]
    return [ev(o) if isinstance(o, Tensor) else o for o in self.outputs]


class _LoopOut(Tensor):
  """One output of a replayed loop; all outputs share one computation per evaluation."""

  def __init__(self, shared, index, build_value):
    self._shared, self._index = shared, index
    super(_LoopOut, self).__init__(None, (), name="loop_out", build_value=build_value)

  def _deps(self):
    return self._shared.deps()

  def _eval(self, ev):
    for c in self._ctrl:
      ev(c)
    return self._shared.values(ev)[self._index]


class _LoopShared(object):
  def __init__(self, compute, deps):
    self._compute, self._deps, self._memo = compute, list(deps), (None, None)

  def deps(self):
    return self._deps

  def values(self, ev):
    if self._memo[0] is not ev:
      self._memo = (ev, self._compute(ev))
    return self._memo[1]


def _outputs_of(compute, deps, n, like=None):
  """n tensors backed by one `compute(ev) -> list of n torch values` (evaluated eagerly once for the build values);
  an output whose counterpart in `like` is a TensorArray is one too."""
  shared = _LoopShared(compute, deps)
  build = shared.values(_BUILD_EV)
  like = like or [None] * n
  return [(_TALoopOut if isinstance(like[i], _TAMethods) else _LoopOut)(shared, i, build[i]) for i in _range(n)]


def _BUILD_EV(t):
  return t._value


# ------------------------------------------------------------------------------------------------------- cells

LSTMStateTuple = collections.namedtuple("LSTMStateTuple", ("c", "h"))


def _zero_state_tensors(state_size, batch_size, dtype):
  td = as_dtype(dtype).torch

  def one(s):
    dims = s.as_list() if isinstance(s, TensorShape) else ([int(s)] if not isinstance(s, (list, _tuple)) else
                                                           [int(v) for v in s])
    return Tensor(lambda b: torch.zeros([int(_t(b))] + dims, dtype=td), (batch_size,), name="zeros")
  return map_structure(one, state_size)


class RNNCell(Layer):
  def __call__(self, inputs, state, scope=None):
    return Layer.__call__(self, inputs, state)

  @property
  def state_size(self):
    raise NotImplementedError

  @property
  def output_size(self):
    raise NotImplementedError

  def zero_state(self, batch_size, dtype):
    return _zero_state_tensors(self.state_size, batch_size, dtype)

  def build(self, _):
    self.built = True


LayerRNNCell = RNNCell


def assert_like_rnncell(cell_name, cell):
  for a in ("output_size", "state_size", "zero_state"):
    if not hasattr(cell, a):
      raise TypeError("The argument %r is not an RNNCell: %r missing" % (cell_name, a))


class LSTMCell(RNNCell):
  """tf.nn.rnn_cell.LSTMCell / BasicLSTMCell without peepholes / projection: kernel [in + H, 4H] applied to
  concat(inputs, h), bias zeros [4H], gates i, j, f, o; c = sigmoid(f + forget_bias) c + sigmoid(i) tanh(j),
  h = sigmoid(o) tanh(c)."""

  def __init__(self, num_units, use_peepholes=False, cell_clip=None, initializer=None, num_proj=None,
               proj_clip=None, num_unit_shards=None, num_proj_shards=None, forget_bias=1.0, state_is_tuple=True,
               activation=None, reuse=None, name=None, dtype=None, **kwargs):
    super(LSTMCell, self).__init__(name=name, dtype=dtype)
    if use_peepholes or num_proj or cell_clip or activation is not None or not state_is_tuple:
      raise NotImplementedError("only the plain LSTM cell is restated")
    self._num_units, self._forget_bias, self._initializer = int(num_units), float(forget_bias), initializer

  @property
  def name(self):
    return self._given_name or "lstm_cell"

  @property
  def state_size(self):
    return LSTMStateTuple(self._num_units, self._num_units)

  @property
  def output_size(self):
    return self._num_units

  def build(self, input_shape):
    cin = int(input_shape[-1])
    self._kernel = self.add_variable("kernel", [cin + self._num_units, 4 * self._num_units],
                                     initializer=self._initializer)
    self._bias = self.add_variable("bias", [4 * self._num_units], initializer=zeros_initializer())
    self.built = True

  def call(self, inputs, state):
    c_prev, h_prev = state
    z = matmul(concat([inputs, h_prev], 1), cast(self._kernel, inputs.dtype)) + cast(self._bias, inputs.dtype)
    i, j, f, o = split(z, 4, axis=1)
    c = sigmoid(f + self._forget_bias) * c_prev + sigmoid(i) * tanh(j)
    h = sigmoid(o) * tanh(c)
    return h, LSTMStateTuple(c, h)


class BasicLSTMCell(LSTMCell):
  def __init__(self, num_units, forget_bias=1.0, state_is_tuple=True, activation=None, reuse=None, name=None,
               dtype=None, **kwargs):
    super(BasicLSTMCell, self).__init__(num_units, forget_bias=forget_bias, state_is_tuple=state_is_tuple,
                                        activation=activation, name=name, dtype=dtype)

  @property
  def name(self):
    return self._given_name or "basic_lstm_cell"


class GRUCell(RNNCell):
  """tf.nn.rnn_cell.GRUCell: gates/kernel [in + H, 2H] (r | u, bias initialised to ones), candidate/kernel [in + H, H];
  r, u = sigmoid([x, h] Wg + bg); c = tanh([x, r * h] Wc + bc); h' = u h + (1 - u) c (the reset gate is applied BEFORE
  the candidate product — unlike the cuDNN form)."""

  def __init__(self, num_units, activation=None, reuse=None, kernel_initializer=None, bias_initializer=None,
               name=None, dtype=None, **kwargs):
    super(GRUCell, self).__init__(name=name, dtype=dtype)
    self._num_units = int(num_units)

  @property
  def name(self):
    return self._given_name or "gru_cell"

  state_size = property(lambda self: self._num_units)
  output_size = property(lambda self: self._num_units)

  def build(self, input_shape):
    from . import ones_initializer
    cin, H = int(input_shape[-1]), self._num_units
    self._gk = self.add_variable("gates/kernel", [cin + H, 2 * H])
    self._gb = self.add_variable("gates/bias", [2 * H], initializer=ones_initializer())
    self._ck = self.add_variable("candidate/kernel", [cin + H, H])
    self._cb = self.add_variable("candidate/bias", [H], initializer=zeros_initializer())
    self.built = True

  def call(self, inputs, state):
    ru = sigmoid(matmul(concat([inputs, state], 1), self._gk) + self._gb)
    r, u = split(ru, 2, axis=1)
    c = tanh(matmul(concat([inputs, r * state], 1), self._ck) + self._cb)
    new_h = u * state + (1 - u) * c
    return new_h, new_h


class MultiRNNCell(RNNCell):
  def __init__(self, cells, state_is_tuple=True):
    super(MultiRNNCell, self).__init__()
    self._cells = list(cells)

  @property
  def name(self):
    return self._given_name or "multi_rnn_cell"

  @property
  def state_size(self):
    return _tuple(c.state_size for c in self._cells)

  @property
  def output_size(self):
    return self._cells[-1].output_size

  def zero_state(self, batch_size, dtype):
    return _tuple(c.zero_state(batch_size, dtype) for c in self._cells)

  def call(self, inputs, state):
    cur, new_states = inputs, []
    for i, cell in enumerate(self._cells):
      with variable_scope("cell_%d" % i):
        cur, ns = cell(cur, state[i])
        new_states.append(ns)
    return cur, _tuple(new_states)


class ResidualWrapper(RNNCell):
  def __init__(self, cell, residual_fn=None):
    super(ResidualWrapper, self).__init__()
    self._cell, self._residual_fn = cell, residual_fn

  state_size = property(lambda self: self._cell.state_size)
  output_size = property(lambda self: self._cell.output_size)

  def zero_state(self, batch_size, dtype):
    return self._cell.zero_state(batch_size, dtype)

  def __call__(self, inputs, state, scope=None):
    outputs, new_state = self._cell(inputs, state)
    fn = self._residual_fn or (lambda i, o: map_structure(lambda a, b: a + b, i, o))
    return fn(inputs, outputs), new_state


class DropoutWrapper(RNNCell):
  def __init__(self, cell, input_keep_prob=1.0, output_keep_prob=1.0, state_keep_prob=1.0, **kwargs):
    super(DropoutWrapper, self).__init__()
    if not (float(input_keep_prob) == float(output_keep_prob) == float(state_keep_prob) == 1.0):
      raise NotImplementedError("DropoutWrapper with keep probabilities below 1 is not restated (parity runs use 1)")
    self._cell = cell

  state_size = property(lambda self: self._cell.state_size)
  output_size = property(lambda self: self._cell.output_size)

  def zero_state(self, batch_size, dtype):
    return self._cell.zero_state(batch_size, dtype)

  def __call__(self, inputs, state, scope=None):
    return self._cell(inputs, state)


# ------------------------------------------------------------------------------------------------------- dynamic_rnn


def _slots_like(structure):
  flat = flatten(structure)
  slots = [(_TASlot if isinstance(v, _TAMethods) else _Slot)(_t(v._value if isinstance(v, Tensor) else v))
           for v in flat]
  return pack_sequence_as(structure, slots), slots


def dynamic_rnn(cell, inputs, sequence_length=None, initial_state=None, dtype=None, parallel_iterations=None,
                swap_memory=False, time_major=False, scope=None):
  """outputs [B, T, size] (or [T, B, size]), final state. Steps at or past sequence_length[b]: zero output, state
  copied through (tf.nn.dynamic_rnn's _rnn_step)."""
  x = inputs if not time_major else Tensor(lambda v: _t(v).transpose(0, 1), (inputs,), name="to_batch_major")
  B, T = int(x._value.shape[0]), int(x._value.shape[1])
  with variable_scope(scope or "rnn"):
    if initial_state is None:
      initial_state = cell.zero_state(B, dtype or x.dtype)
    x_slot = _Slot(x._value[:, 0])
    state_slots, flat_slots = _slots_like(initial_state)
    out, new_state = cell(x_slot, state_slots)
  flat_out, flat_new = flatten(out), flatten(new_state)
  traced = Traced([x_slot] + flat_slots, flat_out + flat_new)
  init_flat = flatten(initial_state)
  n_out, n_state = len(flat_out), len(flat_new)

  def compute(ev):
    xv = ev(x)
    st = [ev(s) if isinstance(s, Tensor) else _t(s) for s in init_flat]
    lens = None if sequence_length is None else _t(ev(sequence_length) if isinstance(sequence_length, Tensor)
                                                   else sequence_length).long()
    outs = [[] for _ in _range(n_out)]
    for t in _range(T):
      res = traced.run(ev, [xv[:, t]] + st)
      o, ns = res[:n_out], res[n_out:]
      if lens is not None:
        live = (t < lens)
        o = [torch.where(live.reshape([-1] + [1] * (v.dim() - 1)), v, torch.zeros_like(v)) for v in o]
        ns = [torch.where(live.reshape([-1] + [1] * (n.dim() - 1)), n, s) if n.dim() > 0 else n
              for n, s in zip(ns, st)]
      for k in _range(n_out):
        outs[k].append(o[k])
      st = ns
    return [torch.stack(v, 1 if not time_major else 0) for v in outs] + st
  deps = [x] + [s for s in init_flat if isinstance(s, Tensor)] + \
      ([sequence_length] if isinstance(sequence_length, Tensor) else []) + flat_out + flat_new
  res = _outputs_of(compute, deps, n_out + n_state, like=flat_out + flat_new)
  return pack_sequence_as(out, res[:n_out]), pack_sequence_as(new_state, res[n_out:])


def _reverse_sequence(x, lengths):
  def f(v, l):
    v, l = _t(v), _t(l).long()
    T = v.shape[1]
    idx = torch.arange(T).unsqueeze(0).expand(v.shape[0], T)
    rev = torch.where(idx < l.unsqueeze(1), l.unsqueeze(1) - 1 - idx, idx)
    return torch.gather(v, 1, rev.reshape(list(rev.shape) + [1] * (v.dim() - 2)).expand_as(v))
  return Tensor(f, (x, lengths), name="reverse_sequence")


def bidirectional_dynamic_rnn(cell_fw, cell_bw, inputs, sequence_length=None, initial_state_fw=None,
                              initial_state_bw=None, dtype=None, parallel_iterations=None, swap_memory=False,
                              time_major=False, scope=None):
  if time_major:
    raise NotImplementedError("time_major bidirectional_dynamic_rnn")
  with variable_scope(scope or "bidirectional_rnn"):
    with variable_scope("fw") as fw_scope:
      out_fw, st_fw = dynamic_rnn(cell_fw, inputs, sequence_length, initial_state_fw, dtype, scope=fw_scope)
    B, T = int(inputs._value.shape[0]), int(inputs._value.shape[1])
    lens = sequence_length if sequence_length is not None else convert_to_tensor([T] * B, dtype=int32)
    with variable_scope("bw") as bw_scope:
      rev_in = _reverse_sequence(inputs, lens)
      tmp, st_bw = dynamic_rnn(cell_bw, rev_in, sequence_length, initial_state_bw, dtype, scope=bw_scope)
    out_bw = _reverse_sequence(tmp, lens)
  return (out_fw, out_bw), (st_fw, st_bw)


# ------------------------------------------------------------------------------------------------------- seq2seq

BasicDecoderOutput = collections.namedtuple("BasicDecoderOutput", ("rnn_output", "sample_id"))


class Decoder(object):
  @property
  def batch_size(self):
    raise NotImplementedError

  @property
  def tracks_own_finished(self):
    return False

  def finalize(self, outputs, final_state, sequence_lengths):
    return outputs, final_state


class Helper(object):
  pass


class TrainingHelper(Helper):
  """next input = inputs[:, time + 1]; finished once time + 1 >= sequence_length; sample ids = argmax (unused)."""

  def __init__(self, inputs, sequence_length, time_major=False, name=None):
    self._inputs = inputs if not time_major else Tensor(lambda v: _t(v).transpose(0, 1), (inputs,))
    self._sequence_length = convert_to_tensor(sequence_length)
    self._batch_size = int(self._inputs._value.shape[0])

  batch_size = property(lambda self: self._batch_size)

  def initialize(self, name=None):
    finished = Tensor(lambda l: _t(l).long() <= 0, (self._sequence_length,), name="finished0")
    return finished, self._inputs[:, 0]

  def sample(self, time, outputs, state, name=None):
    return Tensor(lambda o: _t(o).argmax(-1).to(torch.int32), (outputs,), name="sample")

  def next_inputs(self, time, outputs, state, sample_ids, name=None):
    nt = time + 1
    finished = Tensor(lambda t, l: int(_t(t)) >= _t(l).long(), (nt, self._sequence_length), name="finished")

    def nxt(x, t):
      x, t = _t(x), int(_t(t))
      return x[:, t] if t < x.shape[1] else torch.zeros_like(x[:, 0])
    return finished, Tensor(nxt, (self._inputs, nt), name="next_inputs"), state


class BasicDecoder(Decoder):
  def __init__(self, cell, helper, initial_state, output_layer=None):
    self._cell, self._helper, self._initial_state, self._output_layer = cell, helper, initial_state, output_layer

  batch_size = property(lambda self: self._helper.batch_size)

  def initialize(self, name=None):
    return self._helper.initialize() + (self._initial_state,)

  def step(self, time, inputs, state, name=None):
    cell_outputs, cell_state = self._cell(inputs, state)
    if self._output_layer is not None:
      cell_outputs = self._output_layer(cell_outputs)
    sample_ids = self._helper.sample(time=time, outputs=cell_outputs, state=cell_state)
    finished, next_inputs, next_state = self._helper.next_inputs(time=time, outputs=cell_outputs, state=cell_state,
                                                                 sample_ids=sample_ids)
    return BasicDecoderOutput(cell_outputs, sample_ids), next_state, next_inputs, finished


def dynamic_decode(decoder, output_time_major=False, impute_finished=False, maximum_iterations=None,
                   parallel_iterations=32, swap_memory=False, scope=None):
  """tf.contrib.seq2seq.dynamic_decode: initialize, then step until every sample has finished (or
  maximum_iterations). A sample's sequence length is the number of steps taken while it was not finished. With
  impute_finished the outputs of finished samples are zeros and their state is copied through. Returns
  (final_outputs [B, T, ...] per output field, final_state, final_sequence_lengths), after decoder.finalize."""
  with variable_scope(scope or "decoder"):
    init_finished, init_inputs, init_state = decoder.initialize()
    time_slot = _Slot(torch.zeros((), dtype=torch.int32))
    in_slots, in_flat = _slots_like(init_inputs)
    st_slots, st_flat = _slots_like(init_state)
    outputs, next_state, next_inputs, finished = decoder.step(time_slot, in_slots, st_slots)
  out_flat, ns_flat, ni_flat = flatten(outputs), flatten(next_state), flatten(next_inputs)
  traced = Traced([time_slot] + in_flat + st_flat, out_flat + ns_flat + ni_flat + [finished])
  n_o, n_s, n_i = len(out_flat), len(ns_flat), len(ni_flat)
  i_fin, i_in, i_st = init_finished, flatten(init_inputs), flatten(init_state)

  def compute(ev):
    val = lambda z: ev(z) if isinstance(z, Tensor) else _t(z)      # noqa: E731
    fin = val(i_fin).bool()
    inp, st = [val(z) for z in i_in], [val(z) for z in i_st]
    max_it = None if maximum_iterations is None else int(val(maximum_iterations))
    lens = torch.zeros(fin.shape, dtype=torch.int32)       # [B], or [B, beam_width] under a beam-search decoder
    outs = [[] for _ in _range(n_o)]
    t = 0
    while not _PYBOOL(fin.all()) and (max_it is None or t < max_it):
      res = traced.run(ev, [torch.tensor(t, dtype=torch.int32)] + inp + st)
      o, ns, ni, f = res[:n_o], res[n_o:n_o + n_s], res[n_o + n_s:n_o + n_s + n_i], res[-1].bool()
      if decoder.tracks_own_finished:
        next_fin = f
      else:
        next_fin = f | fin
      if impute_finished:
        o = [torch.where(fin.reshape([-1] + [1] * (v.dim() - 1)), torch.zeros_like(v), v) for v in o]
        ns = [torch.where(fin.reshape([-1] + [1] * (n.dim() - 1)), s, n) if n.dim() > 0 and n.shape[0] == fin.shape[0]
              else n for n, s in zip(ns, st)]
      lens = torch.where(fin, lens, torch.full_like(lens, t + 1))
      for k in _range(n_o):
        outs[k].append(o[k])
      fin, inp, st = next_fin, ni, ns
      t += 1
    stacked = [torch.stack(v, 0) if v else torch.zeros(0) for v in outs]            # time-major, as TF stacks them
    return stacked + st + [lens]
  deps = [z for z in [i_fin] + i_in + i_st + [maximum_iterations] if isinstance(z, Tensor)] + \
      out_flat + ns_flat + ni_flat + [finished]
  res = _outputs_of(compute, deps, n_o + n_s + 1, like=out_flat + ns_flat + [None])
  final_outputs = pack_sequence_as(outputs, res[:n_o])
  final_state = pack_sequence_as(next_state, res[n_o:n_o + n_s])
  lengths = res[-1]
  final_outputs, final_state = decoder.finalize(final_outputs, final_state, lengths)
  if not output_time_major:                    # after finalize, as in TensorFlow (gather_tree reads [T, B, W])
    final_outputs = map_structure(_transpose_batch_time, final_outputs)
  return final_outputs, final_state, lengths


class _TAMethods(object):
  """tf.TensorArray over the node machinery: the array IS a tensor node holding the stacked elements [n, ...]; write
  returns a new array (functional, as TensorFlow's flow semantics are), so it can be a loop variable."""

  def write(self, index, value, name=None):
    def f(cur, i, v):
      cur, i, v = _t(cur), int(_t(i)), _t(v)
      if cur.numel() == 0 and cur.dim() == 1:
        if i != 0:
          raise IndexError("TensorArray.write past the end of an empty array")
        return v.unsqueeze(0)
      if i == cur.shape[0]:
        return torch.cat([cur, v.unsqueeze(0).to(cur.dtype)], 0)
      out = cur.clone()
      out[i] = v
      return out
    return TATensor(f, (self, index, value), name="ta_write")

  def read(self, index, name=None):
    return Tensor(lambda cur, i: _t(cur)[int(_t(i))], (self, index), name="ta_read")

  def stack(self, name=None):
    return identity(self)

  def unstack(self, value, name=None):
    return TATensor(lambda v: _t(v), (value,), name="ta_unstack")

  def size(self, name=None):
    return Tensor(lambda cur: torch.tensor(0 if (_t(cur).dim() == 1 and _t(cur).numel() == 0) else _t(cur).shape[0],
                                           dtype=torch.int32), (self,), name="ta_size")


class TATensor(_TAMethods, Tensor):
  pass


class _TASlot(_TAMethods, _Slot):
  pass


class _TALoopOut(_TAMethods, _LoopOut):
  pass


class _TAMeta(type):
  def __instancecheck__(cls, obj):          # isinstance(x, tf.TensorArray) (rnn_beam_search_decoder.py:128)
    return isinstance(obj, _TAMethods)


class TensorArray(object, metaclass=_TAMeta):   # noqa: N801
  def __new__(cls, dtype, size=0, dynamic_size=False, clear_after_read=None, element_shape=None, **kwargs):
    td = as_dtype(dtype).torch
    return TATensor(lambda: torch.zeros(0, dtype=td), (), name="tensor_array")


def _transpose_batch_time(x):
  return Tensor(lambda v: _t(v).transpose(0, 1), (x,), name="transpose_batch_time")


def while_loop(cond, body, loop_vars, shape_invariants=None, parallel_iterations=10, back_prop=True,
               swap_memory=False, name=None, maximum_iterations=None, return_same_structure=False):
  """tf.while_loop: cond and body are CALLED ONCE on loop-variable slots (that is how TensorFlow builds the loop) and
  the two recorded graphs are replayed while cond holds; loop variables may change shape between iterations (the
  shape_invariants of the caller say the same)."""
  single = not isinstance(loop_vars, (list, _tuple))
  lv = [loop_vars] if single else list(loop_vars)
  slots_struct, slots = _slots_like(lv)
  if shape_invariants is not None:
    inv = flatten([shape_invariants] if single else list(shape_invariants))
    for sl, sh in zip(slots, inv):
      if isinstance(sh, TensorShape) and sh.dims is not None and any(d is None for d in sh.dims):
        v = sl._value
        if len(sh.dims) == v.dim():
          # one longer in the LAST open dimension (never the leading one): the dimension that grows from one
          # iteration to the next in the loops the reference writes (decoded length, cache length). Open leading
          # dimensions (batch, beam) are tied across loop variables and do not change inside a loop: lengthening
          # them independently per variable would make the shadow values inconsistent with each other
          open_dims = [i for i, d in enumerate(sh.dims) if d is None and i > 0]
          if open_dims:
            pad = []
            for i in reversed(_range(len(sh.dims))):
              pad += [0, 1 if i == open_dims[-1] else 0]
            sl._alt = torch.nn.functional.pad(v, pad)
  c = cond(*slots_struct)
  out = body(*slots_struct)
  if not isinstance(out, (list, _tuple)):
    out = [out]
  out_flat = flatten(list(out))
  if len(out_flat) != len(slots):
    raise ValueError("while_loop: body returned %d tensors for %d loop variables" % (len(out_flat), len(slots)))
  tc, tb = Traced(slots, [c]), Traced(slots, out_flat)
  init_flat = flatten(lv)

  def compute(ev):
    vals = [ev(z) if isinstance(z, Tensor) else _t(z) for z in init_flat]
    it = 0
    while _PYBOOL(tc.run(ev, vals)[0]) and (maximum_iterations is None or it < int(maximum_iterations)):
      vals = tb.run(ev, vals)
      it += 1
    return vals
  deps = [z for z in init_flat if isinstance(z, Tensor)] + [c] + [o for o in out_flat if isinstance(o, Tensor)]
  res = _outputs_of(compute, deps, len(slots), like=out_flat)
  packed = pack_sequence_as(lv, res)
  return packed[0] if single else packed


def embedding_lookup(params, ids, partition_strategy="mod", name=None, validate_indices=True, max_norm=None):
  return Tensor(lambda p, i: _t(p)[_t(i).long()], (params, ids), name="embedding_lookup")

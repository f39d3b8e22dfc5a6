"""capi._ZeroArena (the per-backward-pass zeroed scratch for BatchNorm-backward statistic partials):
every slice handed out reads as zeros, slices of one pass never overlap, and reset() re-zeroes what the
previous pass dirtied. Host logic only (CPU tensors)."""
import torch


def test_zero_arena_hands_out_disjoint_zeroed_slices():
  from openseq2seq_amd.capi import _ZeroArena
  dev = torch.device("cpu")
  arena = _ZeroArena()
  arena.reset()                                   # nothing known yet: a no-op
  shapes = [(7, 2, 24), (3, 2, 1024), (1, 2, 8)]
  first = [arena.take(s, dev) for s in shapes]    # before the first reset: plain zero tensors
  assert arena.buf is None and all(float(t.abs().sum()) == 0.0 and tuple(t.shape) == s for t, s in zip(first, shapes))
  arena.reset()                                   # sized to the demand of that pass
  assert arena.buf is not None and arena.buf.numel() >= sum(t.numel() for t in first)
  for rounds in range(2):
    got = [arena.take(s, dev) for s in shapes]
    for t, s in zip(got, shapes):
      assert tuple(t.shape) == s and float(t.abs().sum()) == 0.0
      assert t.data_ptr() >= arena.buf.data_ptr() and t.data_ptr() < arena.buf.data_ptr() + arena.buf.numel() * 4
    spans = sorted((t.data_ptr(), t.data_ptr() + t.numel() * 4) for t in got)
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))     # disjoint
    for t in got:
      t.fill_(3.0)                                                  # a pass writes its partials
    arena.reset()
  # a pass that needs more than the buffer holds falls back to fresh zeros and grows the buffer next time
  big = arena.take((64, 2, 4096), dev)
  assert float(big.abs().sum()) == 0.0
  arena.reset()
  assert arena.buf.numel() >= big.numel()


def test_tape_backward_refuses_to_nest():
  """The zeroed scratch is per process: a backward pass started from inside another one must fail loudly
  instead of re-zeroing the outer pass's partials."""
  import pytest
  from openseq2seq_amd.parts.cnns import conv_blocks as cb
  from openseq2seq_amd import capi
  calls = []
  saved = (cb.join_side_streams, capi.zero_arena_reset)
  cb.join_side_streams = lambda: None
  capi.zero_arena_reset = lambda: calls.append("reset")
  try:
    inner = cb.Tape()
    inner.record(lambda: calls.append("inner"))
    outer = cb.Tape()
    outer.record(lambda: inner.backward())
    with pytest.raises(RuntimeError, match="another backward pass"):
      outer.backward()
    assert cb.current_tape() is None and calls == ["reset"]      # the outer pass cleaned up, the inner never ran
    inner.backward()                                             # sequential passes are fine
    assert calls == ["reset", "reset", "inner"]
  finally:
    cb.join_side_streams, capi.zero_arena_reset = saved

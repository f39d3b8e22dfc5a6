"""sequence_beam_search — open_seq2seq/parts/transformer/beam_search.py:386-418 with the same
signature and results, on device-resident state (csrc/beam_search.hip).

symbols_to_logits_fn(ids [B*beam, i+1] int32, i, cache) -> (logits [B*beam, vocab] fp32|bf16,
cache) exactly as in the reference; `cache` is a (nested) dict of device tensors whose leading
dimension is the beam row — every leaf is re-gathered by parent beam after each step
(_gather_beams, :505-537). Callers that keep large per-beam state should store only an index
table in `cache` (see decoders/transformer_decoder.py: the K/V caches never move).

The loop condition lives on the device; the host polls it every `poll_every` steps — steps
enqueued after the search has finished are no-ops, so the result does not depend on the
polling interval."""
from __future__ import absolute_import, division, print_function

import torch

from ... import capi

INF = 32768.0   # beam_search.py:26


def _map(fn, nested):
  if isinstance(nested, dict):
    return {k: _map(fn, v) for k, v in nested.items()}
  if isinstance(nested, (list, tuple)):
    return type(nested)(_map(fn, v) for v in nested)
  return fn(nested)


def _expand_to_beam_size(tensor, beam_size):
  """[B, ...] -> [B*beam, ...] (already flattened: _expand_to_beam_size + _flatten_beam_dim)."""
  B = tensor.shape[0]
  idx = (torch.arange(B * beam_size, dtype=torch.int32, device=tensor.device) // beam_size).to(torch.int32)
  return capi.gather_rows(tensor.contiguous(), idx)


def _leaves(nested, out=None):
  out = [] if out is None else out
  if isinstance(nested, dict):
    for k in sorted(nested):
      _leaves(nested[k], out)
  elif isinstance(nested, (list, tuple)):
    for v in nested:
      _leaves(v, out)
  else:
    out.append(nested)
  return out


def _graph_search(state, device_step_fn, cache, max_decode_length, poll_every):
  """The loop as replays of ONE captured hipGraph: `device_step_fn(state, cache)` reads the
  loop index, last tokens and positions from device state (state.status / last_ids / pos), so
  a step has no host-side arguments. The step is launch-bound when run eagerly (~90 small
  kernels); a graph replay removes the per-launch host cost."""
  leaves = _leaves(cache)
  tmp = [torch.empty_like(t) for t in leaves]

  def one_step():
    logits, _ = device_step_fn(state, cache)
    state.step(logits)
    for t, buf in zip(leaves, tmp):       # in-place permutation through a scratch copy
      capi.gather_rows(t, state.parent, enable=state.status, out=buf)
      t.copy_(buf)

  one_step()            # warm-up outside the capture (one-time kernel attribute / autotune calls)
  state.reset()
  torch.cuda.synchronize()
  graph = torch.cuda.CUDAGraph()
  with torch.cuda.graph(graph):
    one_step()
  state.reset()         # the capture itself does not execute, but keep the contract explicit
  i = 0
  while i < max_decode_length:
    graph.replay()
    i += 1
    if i % poll_every == 0 or i == max_decode_length:
      running, _ = state.read_status()
      if not running:
        break


def sequence_beam_search(symbols_to_logits_fn, initial_ids, initial_cache, vocab_size, beam_size,
                         alpha, max_decode_length, eos_id, poll_every=4, debug_state=None,
                         device_step_fn=None):
  """Returns (top sequences int32 [B, beam, steps + 1], scores fp32 [B, beam]).
  device_step_fn(state, cache) -> (logits, cache): optional variant of symbols_to_logits_fn
  that takes the loop index / last ids from device state; the loop then runs as hipGraph
  replays (same kernels, same results)."""
  initial_ids = initial_ids.to(torch.int32).contiguous()
  B = int(initial_ids.shape[0])
  max_decode_length = int(max_decode_length)
  state = capi.BeamState(initial_ids, beam_size, vocab_size, max_decode_length, alpha, eos_id,
                         debug=debug_state is not None)
  cache = _map(lambda t: _expand_to_beam_size(t, beam_size), initial_cache)
  if device_step_fn is not None and max_decode_length > 0:
    _graph_search(state, device_step_fn, cache, max_decode_length, max(poll_every, 8))
    _, steps = state.read_status()
    out_seq, out_scores = state.finalize()
    return out_seq[:, :, :steps + 1], out_scores
  i = 0
  while i < max_decode_length:
    logits, cache = symbols_to_logits_fn(state.alive_ids(i), i, cache)
    state.step(logits)
    if debug_state is not None:
      debug_state(i, state)
    cache = _map(lambda t: capi.gather_rows(t, state.parent, enable=state.status), cache)
    i += 1
    if i % poll_every == 0 or i == max_decode_length:
      running, _ = state.read_status()
      if not running:
        break
  _, steps = state.read_status()
  out_seq, out_scores = state.finalize()
  return out_seq[:, :, :steps + 1], out_scores

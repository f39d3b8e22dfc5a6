"""Host-side index preparation for the packed (padding-free) token layout.

The reference feeds padded [B, L] id matrices (pad id 0) and removes / masks padding
inside the graph (parts/transformer/utils.py:82-129, ffn_layer.py:56-70). Here the data
layer (which has the lengths on the host anyway) emits, next to the padded matrices, the
flat index vectors the kernels consume; a padded batch without them is packed with one
device->host copy of the lengths."""
import numpy as np
import torch


def pack_ids(ids_padded_host, lens_host, shift_right=False):
  """ids [B, L] (numpy), lens [B] -> dict(ids [N], pos [N], cu [B+1], labels [N]|None, max_len).
  shift_right: decoder inputs = previous target token (0 = pad -> zero embedding for the
  first position, decoders/transformer_decoder.py:197-202); labels = the tokens themselves."""
  B = len(lens_host)
  lens = np.asarray(lens_host, np.int64)
  cu = np.zeros(B + 1, np.int32)
  cu[1:] = np.cumsum(lens)
  N = int(cu[-1])
  ids = np.zeros(N, np.int32)
  pos = np.zeros(N, np.int32)
  labels = np.zeros(N, np.int32) if shift_right else None
  for b in range(B):
    n = int(lens[b])
    row = np.asarray(ids_padded_host[b][:n], np.int32)
    s = cu[b]
    pos[s:s + n] = np.arange(n)
    if shift_right:
      labels[s:s + n] = row
      ids[s + 1:s + n] = row[:n - 1]
    else:
      ids[s:s + n] = row
  return dict(ids=ids, pos=pos, cu=cu, labels=labels, max_len=int(lens.max()) if B else 0, n=N)


def to_device(p, device):
  out = dict(p)
  for k in ("ids", "pos", "cu", "labels"):
    if p.get(k) is not None:
      out[k] = torch.from_numpy(p[k]).to(device, non_blocking=True)
  return out


def unpack_rows(x_packed, cu_host, max_len):
  """[N, D] packed -> [B, max_len, D] zero padded (host/debug helper)."""
  B = len(cu_host) - 1
  out = x_packed.new_zeros((B, max_len) + tuple(x_packed.shape[1:]))
  for b in range(B):
    n = cu_host[b + 1] - cu_host[b]
    out[b, :n] = x_packed[cu_host[b]:cu_host[b + 1]]
  return out

"""ORACLE (test infrastructure only): CTC loss on the CPU.

Restates tf.nn.ctc_loss(labels, logits[T,B,V], seq_len,
ignore_longer_outputs_than_inputs=True) + mask_nans + reduce_mean as used by
open_seq2seq/losses/ctc_loss.py:77-88 (blank = V-1, ctc_merge_repeated=True).

PARITY STATUS: the reference pins no CTC *loss* values ("parity
unpinned"); two independent implementations cross-pin each other here:
  * `ctc_loss_numpy`  — direct alpha recursion in float64 (Graves 2006, eq. 6-8);
  * `ctc_loss_torch`  — torch.nn.functional.ctc_loss (an unrelated codebase),
    which also supplies the gradient w.r.t. the logits via autograd.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _feasible(lab, T):
  """TF's check (ctc_loss_calculator.h): the label needs len + #adjacent-repeats frames."""
  rep = sum(1 for i in range(1, len(lab)) if lab[i] == lab[i - 1])
  return T > 0 and len(lab) + rep <= T


def ctc_loss_numpy(logits, in_len, labels, label_len, blank=None):
  """Per-sample -log p(l|x), 0 for infeasible samples. logits [T,B,V]."""
  logits = np.asarray(logits, np.float64)
  T, B, V = logits.shape
  if blank is None:
    blank = V - 1
  out = np.zeros(B)
  for b in range(B):
    Tb = int(in_len[b]); L = int(label_len[b])
    lab = [int(x) for x in labels[b][:L]]
    if not _feasible(lab, Tb):
      continue
    x = logits[:Tb, b, :]
    lp = x - np.log(np.exp(x - x.max(-1, keepdims=True)).sum(-1, keepdims=True)) - x.max(-1, keepdims=True)
    ext = [blank]
    for c in lab:
      ext += [c, blank]
    S = len(ext)
    a = np.full(S, -np.inf)
    a[0] = lp[0, blank]
    if S > 1:
      a[1] = lp[0, ext[1]]
    for t in range(1, Tb):
      n = np.full(S, -np.inf)
      for s in range(S):
        c = [a[s]]
        if s >= 1:
          c.append(a[s - 1])
        if s >= 2 and ext[s] != blank and ext[s] != ext[s - 2]:
          c.append(a[s - 2])
        m = max(c)
        if m > -np.inf:
          n[s] = m + np.log(sum(np.exp(v - m) for v in c)) + lp[t, ext[s]]
      a = n
    c = [a[S - 1]] + ([a[S - 2]] if S > 1 else [])
    m = max(c)
    ll = m + np.log(sum(np.exp(v - m) for v in c)) if m > -np.inf else -np.inf
    out[b] = -ll if np.isfinite(ll) else 0.0
  return out


def ctc_loss_torch(logits, in_len, labels, label_len, blank=None, want_grad=False):
  """Returns (loss_per_sample [B], mean over batch, dlogits [T,B,V] of sum_b loss_b)."""
  # float64 internally: the fp32 alpha/beta recursions of any implementation
  # (TF's included) carry ~1e-3 absolute noise on peaky inputs; the oracle should
  # not contribute its own.
  logits = torch.as_tensor(logits).double().clone().requires_grad_(want_grad)
  T, B, V = logits.shape
  if blank is None:
    blank = V - 1
  in_len = torch.as_tensor(in_len, dtype=torch.long).clamp(0, T)
  label_len = torch.as_tensor(label_len, dtype=torch.long)
  labels = torch.as_tensor(labels, dtype=torch.long)
  feas = torch.tensor([_feasible([int(v) for v in labels[b][:label_len[b]]], int(in_len[b]))
                       for b in range(B)])
  lp = F.log_softmax(logits, dim=-1)
  # torch requires in_len >= 1; infeasible samples are zeroed afterwards anyway
  loss = F.ctc_loss(lp, labels, in_len.clamp(min=1), label_len, blank=blank,
                    reduction="none", zero_infinity=True)
  loss = torch.where(feas, loss, torch.zeros_like(loss))
  loss = torch.where(torch.isfinite(loss), loss, torch.zeros_like(loss))  # mask_nans
  grad = None
  if want_grad:
    loss.sum().backward()
    # frames past in_len get no gradient
    tmask = (torch.arange(T)[:, None] < in_len[None, :]).double()[:, :, None]
    grad = (logits.grad.detach() * tmask).float()
  return loss.detach().float(), loss.detach().mean().float(), grad

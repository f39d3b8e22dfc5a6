"""Dense-residual block ends without branch tensors.

conv_bn_res_bn_actv (open_seq2seq/parts/cnns/conv_blocks.py:61-168) adds, at the end of block k, one
1x1 convolution + BatchNorm per dense-residual input (encoders/tdnn_encoder.py:188-192: the inputs of blocks
0 .. k) to the main branch: 55 branch tensors per pass of Jasper 10x5, each written, normalised, re-read twice
in backward and paired with a gradient tensor of its own. Every branch is a linear image of a block input
("source"), so this module evaluates the same sum and the same gradients from

  * the channel-concatenated sources Xcat [B, T, sum c_i] (one masked copy per source),
  * per source: column sums and the Gram matrix r_i^T r_i (one TN GEMM)  -> the branch statistics,
  * per block end: ONE GEMM Xcat[:, :, :K_k] . (BN-scaled stacked kernels)^T      (the sum of the branches),
    and in backward ONE TN GEMM P_k = dz_k^T Xcat[:, :, :K_k]                      (all kernel gradients),
  * per source: ONE GEMM [dz_i | .. | dz_n] . (stacked kernels x gamma rstd) and one with the source itself
    against a [c_i, c_i] matrix                                                    (its data gradient).

The algebra and the small kernels between the GEMMs: csrc/dense_residual.hip. Parameters, their names, the
moving statistics and every gradient are those of the branch-by-branch path (conv_blocks.conv_bn_res_bn_actv),
which stays the path for everything this plan does not take (separable branches, block dropping, channel counts
that are not multiples of 64, blocks that change the frame rate).
"""
from __future__ import absolute_import, division, print_function

import os

import torch

from ... import capi

# A/B knob: 0 = branch-by-branch residual path everywhere (rounds 1 - 5)
ENABLED = os.environ.get("OS2S_DENSE_RES_ALGEBRA", "1") != "0"
# A/B knob: 0 = the Gram matrices on the lockstep TN kernel (fp32 atomics over a wide reduction split) instead of the
# ping-pong kernel (one owner per tile sums up to 17 slabs)
GRAM_PINGPONG = os.environ.get("OS2S_DRES_GRAM_PP", "1") != "0"
P_PINGPONG = os.environ.get("OS2S_DRES_P_PP", "1") != "0"     # the same choice for the P_k products


class _End(object):
  __slots__ = ("k", "branches", "cout", "kk", "doff", "eps", "momentum", "tt", "mean", "rstd", "wp", "shift", "P",
               "coef", "table", "ptrs")


class _Source(object):
  __slots__ = ("i", "c", "koff", "wd", "gram", "s", "m", "chl", "wd1", "wd2", "wt", "mf", "mb")


class DenseResidualPlan(object):
  """Static part: which branch reads which source, the persistent weight-sized buffers and the device tables of
  os2s_dres_seg_t. `ends[k]` = the k + 1 plain 1x1 ConvBN branches of block end k (branch i reads source i)."""

  @staticmethod
  def eligible(ends):
    from .conv_blocks import ConvBN
    if not ENABLED or len(ends) < 2 or len(ends) > 16:
      return False
    for k, brs in enumerate(ends):
      if len(brs) != k + 1:
        return False
      for i, br in enumerate(brs):
        if type(br) is not ConvBN or br.k != 1 or br.stride != 1 or br.cin % 64 or br.cout % 64:
          return False
        if br.cout != brs[0].cout or br.eps != brs[0].eps or br.momentum != brs[0].momentum:
          return False
        if br.cin != ends[-1][i].cin:
          return False
    return True

  def __init__(self, ends, device):
    self.device = device
    self.generation = 0
    n = len(ends)
    self.n = n
    f32 = dict(dtype=torch.float32, device=device)
    b16 = dict(dtype=torch.bfloat16, device=device)
    chans = [ends[-1][i].cin for i in range(n)]
    couts = [ends[k][0].cout for k in range(n)]
    self.ktot = sum(chans)
    self.dtot = sum(couts)
    doff = [sum(couts[:k]) for k in range(n)]
    self.sources = []
    for i in range(n):
      S = _Source()
      S.i, S.c, S.koff = i, chans[i], sum(chans[:i])
      S.wd = self.dtot - doff[i]               # columns of the stacks: every block end k >= i
      S.gram = torch.zeros((S.c, S.c), **f32)
      S.s = torch.zeros(S.c, **f32)
      S.m = torch.zeros(S.c, **f32)
      S.chl = torch.zeros((2 * S.c, S.c), **b16)
      S.wd1 = torch.zeros((S.c, S.wd), **b16)
      S.wd2 = torch.zeros((S.c + 8, S.wd), **b16)      # row c: the constant-row coefficients, rows c+1.. stay zero
      S.wt = torch.zeros((S.c, S.wd), **b16)
      S.mf = torch.zeros((S.c + 8, S.c), **f32)
      S.mb = torch.zeros((S.c, S.c), **b16)
      self.sources.append(S)
    self.ends = []
    for k in range(n):
      E = _End()
      E.k, E.branches, E.cout = k, list(ends[k]), couts[k]
      E.kk = sum(chans[:k + 1])
      E.doff = doff[k]
      E.eps, E.momentum = ends[k][0].eps, ends[k][0].momentum
      E.tt = [torch.zeros((E.cout, 2 * chans[i]), **f32) for i in range(k + 1)]
      E.mean = [torch.zeros(E.cout, **f32) for _ in range(k + 1)]
      E.rstd = [torch.ones(E.cout, **f32) for _ in range(k + 1)]
      E.wp = torch.zeros((E.cout, E.kk), **b16)
      E.shift = torch.zeros(E.cout, **f32)
      E.P = torch.zeros((E.cout, E.kk), **f32)
      E.coef = torch.zeros((k + 1) * 4 * E.cout, **f32)
      E.table, E.ptrs = None, None
      self.ends.append(E)
    self._ones = {}

  def ones(self, c):
    t = self._ones.get(c)
    if t is None:
      t = self._ones[c] = torch.ones(c, dtype=torch.float32, device=self.device)
    return t

  def table(self, E):
    """The device table of block end E (rebuilt when a parameter view moved: checkpoint loads copy in place, so in
    practice it is built once)."""
    ptrs = tuple(p.data_ptr() for br in E.branches
                 for p in (br.kernel._w16, br.kernel._grad, br.gamma._master, br.beta._master, br.gamma._grad,
                           br.beta._grad, br.moving_mean, br.moving_var))
    if E.table is not None and ptrs == E.ptrs:
      return E.table
    segs = []
    for i, br in enumerate(E.branches):
      S = self.sources[i]
      col = E.doff - (self.dtot - S.wd)         # this block end's first column in the source's stacks
      segs.append(dict(w=br.kernel._w16, tt=E.tt[i], m=S.m, s=S.s, gamma=br.gamma._master, beta=br.beta._master,
                       moving_mean=br.moving_mean, moving_var=br.moving_var, mean=E.mean[i], rstd=E.rstd[i],
                       dgamma=br.gamma._grad, dbeta=br.beta._grad, dw=br.kernel._grad, wd1=S.wd1[:, col:],
                       wd2=S.wd2[:, col:], wt=S.wt[:, col:], ld=S.wd, c=S.c, koff=S.koff))
    E.table, E.ptrs = capi.dres_seg_table(segs, self.device), ptrs
    return E.table

  def begin(self, training):
    if training:
      self.generation += 1
    return DenseResidualPass(self, training)


class DenseResidualPass(object):
  """One forward (and, in training, backward) pass over the plan: the concatenated activations and gradients."""

  def __init__(self, plan, training):
    self.plan, self.training = plan, training
    self.generation = plan.generation
    self.xcat = self.dzcat = None
    self.acts = []
    self.lens = None
    self.B = self.T = 0

  # ---- forward -----------------------------------------------------------------------------------------------
  def add_source(self, x):
    """x: Act, the input of the next dense-residual block. Copies it (masked) into the concatenated buffer; in
    training also its column sums, Gram matrix and the covariance pair."""
    plan = self.plan
    i = len(self.acts)
    S = plan.sources[i]
    B, T, C = x.data.shape
    assert C == S.c
    if i == 0:
      self.B, self.T, self.lens = B, T, x.lens
      self.xcat = torch.empty((B, T, plan.ktot), dtype=torch.bfloat16, device=x.data.device)
    elif (B, T) != (self.B, self.T) or x.lens is not self.lens:
      raise ValueError("dense-residual plan: block inputs of different geometry")
    self.acts.append(x)
    xs = self.xcat[:, :, S.koff:S.koff + C]
    part = capi.dres_copy_cols(x.data, xs, self.lens, want_colsum=self.training)
    if self.training:
      S.gram.zero_()
      capi.conv1x1_wgrad_grouped([dict(x=xs, dy=x.data, dw=S.gram.view(1, C, C))], in_len=self.lens,
                                 pingpong=GRAM_PINGPONG or capi.deterministic())
      capi.dres_cov(part, S.gram, B * T, S.s, S.m, S.chl)

  def forward_end(self, k):
    """The sum of the k + 1 BatchNorm'd residual branches of block end k as one tensor R and one shift vector:
    returns the record conv_bn_dres_actv consumes."""
    plan = self.plan
    E = plan.ends[k]
    assert len(self.acts) >= k + 1
    table = plan.table(E)
    for br in E.branches:         # readiness of the variables under an asynchronous optimizer update
      br.kernel.w16, br.gamma.master, br.beta.master
    if self.training:
      items = []
      for i, br in enumerate(E.branches):
        S = plan.sources[i]
        items.append(dict(x=br.kernel._w16.view(1, E.cout, S.c), w=S.chl.view(1, 2 * S.c, S.c),
                          y=E.tt[i].view(1, E.cout, 2 * S.c)))
      capi.conv1x1_fwd_grouped(items, out_f32=True)
    capi.dres_bn_fwd(table, k + 1, E.cout, E.kk, E.wp, E.shift, self.B * self.T, E.eps, E.momentum, self.training)
    R = torch.empty((self.B, self.T, E.cout), dtype=torch.bfloat16, device=self.xcat.device)
    capi.conv1x1_cat_fwd(self.xcat[:, :, :E.kk], E.wp, R, in_len=self.lens)
    return dict(y=R, scale=plan.ones(E.cout), shift=E.shift, dres=self, k=k)

  # ---- backward ----------------------------------------------------------------------------------------------
  def backward_end(self, k, dz, mean_dz, dz_lens=None):
    """dz [B, T, Cout_k] = the gradient at the block end's sum (activation / dropout / mask backward applied),
    mean_dz [Cout_k] = its column means: every branch's kernel / gamma / beta gradient, and — block end k is the
    last reader of source k in backward order — the complete residual data gradient of source k."""
    plan = self.plan
    if self.generation != plan.generation:
      raise RuntimeError("dense-residual plan: a later training forward pass overwrote this pass's saved statistics")
    E = plan.ends[k]
    N = self.B * self.T
    if self.dzcat is None:
      self.dzcat = torch.empty((self.B, self.T, plan.dtot), dtype=torch.bfloat16, device=dz.device)
    for br in E.branches:
      br.kernel.grad, br.gamma.grad, br.beta.grad
    # (dz_lens: dz came out of a data-gradient epilogue that writes the live windows only — rows past a sequence
    # end are unwritten memory: masked here; the TN GEMM below visits live 64-row chunks only)
    capi.dres_copy_cols(dz, self.dzcat[:, :, E.doff:E.doff + E.cout], dz_lens)
    E.P.zero_()
    capi.conv1x1_wgrad_grouped([dict(x=self.xcat[:, :, :E.kk], dy=dz, dw=E.P.view(1, E.cout, E.kk))],
                               in_len=self.lens, pingpong=P_PINGPONG or capi.deterministic())
    capi.dres_bn_bwd(plan.table(E), k + 1, E.cout, E.kk, E.P, mean_dz, N, E.coef)
    inp = self.acts[k]
    if not inp.requires_grad:
      return
    S = plan.sources[k]
    g = inp.grad_buffer()
    capi.conv1x1_cat_fwd(self.dzcat[:, :, E.doff:], S.wd1, g, out_len=self.lens, accumulate=inp.grad_init)
    capi.gemm_nt(S.wd2, S.wt, out=S.mf, out_f32=True)
    capi.cast_f32_to_bf16(S.mf[:S.c], S.mb)
    capi.conv1x1_cat_fwd(self.xcat[:, :, S.koff:S.koff + S.c], S.mb, g, out_len=self.lens, bias=S.mf[S.c],
                         accumulate=True)
    inp.grad_init = True

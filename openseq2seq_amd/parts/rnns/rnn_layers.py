"""Recurrent layers on the HIP RNN kernels (csrc/rnn.hip): cuDNN-form GRU / LSTM
(tf.contrib.cudnn_rnn.CudnnGRU/CudnnLSTM, encoders/ds2_encoder.py:294-328,
encoders/tacotron2_encoder.py:254-263) and tf.nn.rnn_cell.LSTMCell under
(bidirectional_)dynamic_rnn (encoders/rnn_encoders.py:221-305, parts/rnns/utils.py:17-89).

One `RNNDirection` = one direction of one layer: the input projection of all time steps is
one MFMA GEMM per input tensor (a bidirectional lower layer is consumed as two tensors, so
no concat is materialised), the recurrence is os2s_rnn_layer_fwd/bwd, and the weight
gradients are GEMMs over the saved gate gradients (h_{t-1} enters as a time-shifted
operand of the wgrad kernel)."""
import math
import os

import torch

from ... import capi
from ..cnns.conv_blocks import Act, on_side_stream
from ..transformer.layers import _colsum_into

CELLS = {"gru_cudnn": capi.CELL_GRU_CUDNN, "lstm_cudnn": capi.CELL_LSTM_CUDNN,
         "lstm_tf": capi.CELL_LSTM_TF}


# rows (B x T) from which the recurrent weight gradient goes through the ping-pong TN GEMM
SHIFTED_WH_MIN_ROWS = int(os.environ.get("OS2S_RNN_WH_GEMM_ROWS", "4096"))


class RNNDirection(object):
  def __init__(self, store, name, cell, input_sizes, hidden, reverse=False, forget_bias=1.0):
    self.cell_name, self.cell = cell, CELLS[cell]
    self.H, self.G = hidden, 3 if cell == "gru_cudnn" else 4
    self.reverse, self.forget_bias = reverse, (forget_bias if cell == "lstm_tf" else 0.0)
    GH = self.G * hidden
    tot_in = sum(input_sizes)

    def init_w(fan_in):
      def f(shape):   # glorot-uniform over the full [in + H, G*H] matrix (TF LSTMCell default)
        lim = math.sqrt(6.0 / (fan_in + hidden + GH))
        return (torch.rand(shape) * 2 - 1) * lim
      return f

    self.wx = [store.add("%s/wx_%d" % (name, i), (1, GH, n), init_w(tot_in), kind="conv")
               for i, n in enumerate(input_sizes)]
    self.wh = store.add(name + "/wh", (1, GH, hidden), init_w(tot_in), kind="conv")
    self.bx = store.add(name + "/bias", (GH,), torch.zeros(GH), kind="vector")
    self.bh = store.add(name + "/bias_h", (GH,), torch.zeros(GH), kind="vector") \
        if cell != "lstm_tf" else None

  def input_projection(self, xs):
    B, T, _ = xs[0].data.shape
    GH = self.G * self.H
    gx = None
    for i, (x, w) in enumerate(zip(xs, self.wx)):
      gx = capi.gemm(x.data.reshape(B * T, -1), w.w16.view(GH, -1),
                     bias=self.bx.master if i == 0 else None, out=gx, accumulate=i > 0)
    return gx.view(B, T, GH)

  def params(self):
    return [self.wh, self.bx] + self.wx + ([self.bh] if self.bh is not None else [])

  def weight_backward(self, xs, lens, y, dgx, dgr):
    """dX, dWx, dWh and the bias gradients from the saved gate gradients."""
    B, T, _ = xs[0].data.shape
    H, GH = self.H, self.G * self.H
    d2 = dgx.view(B * T, GH)
    # the data gradients first, on the main stream (the layer below waits for them) ...
    for x, w in zip(xs, self.wx):
      if x.requires_grad:
        g = x.grad_buffer()
        capi.gemm(d2, w.wt16.view(-1, GH), out=g.view(B * T, -1), accumulate=x.grad_init)
        x.grad_init = True
    # ... the parameter gradients on the side stream: nothing in the rest of backward reads them, and
    # during the next layer's recurrence (ONE persistent launch on two XCDs for a DeepSpeech2 GRU layer)
    # six XCDs have nothing else to do
    with on_side_stream(dgx.device, dgx, dgr, y, *[x.data for x in xs]):
      for x, w in zip(xs, self.wx):
        capi.gemm_wgrad(x.data.reshape(B * T, -1), d2, w.grad.view(GH, -1), accumulate=True)
      _colsum_into(d2, self.bx)
      if self.bh is not None:
        _colsum_into(dgr.view(B * T, GH), self.bh)
      # dWh += dgr^T . h_{t-1}: h_{t-1} is y shifted by one step in processing order
      if B * T >= SHIFTED_WH_MIN_ROWS and H % 8 == 0 and T > 1:
        # as a plain TN GEMM over a shifted copy of y (the 256 x 256 ping-pong weight-gradient kernel:
        # 165 -> ~90 us per DeepSpeech2 layer and direction) instead of the shifted-window K = 1 convolution
        # gradient on the lockstep tile. Rows at or past a sample's length carry zero gate gradients, and
        # y is zero there (the step kernels never write a finished sample), so the copy needs no mask.
        ysh = torch.empty((B, T, H), dtype=y.dtype, device=y.device)
        if self.reverse:
          ysh[:, :-1] = y[:, 1:]
          ysh[:, -1] = 0
        else:
          ysh[:, 1:] = y[:, :-1]
          ysh[:, 0] = 0
        capi.gemm_wgrad(ysh.view(B * T, H), dgr.view(B * T, GH), self.wh.grad.view(GH, H), accumulate=True)
      else:
        capi.conv1d_wgrad(y, dgr, 1, pad_left=(-1 if self.reverse else 1), in_len=lens,
                          out=self.wh.grad, accumulate=True)

  def forward(self, xs, lens, tape, y_view=None, dy_view_fn=None):
    """xs: list of Act [B,T,In_i]; lens int32 [B] or None. Returns Act [B,T,H].
    y_view: optional [B,T,H] channel-slice view to write the outputs into;
    dy_view_fn() then returns the matching slice of the shared gradient."""
    return rnn_directions_forward([self], xs, lens, tape, [y_view],
                                  [dy_view_fn] if dy_view_fn is not None else None)[0]


def rnn_directions_forward(dirs, xs, lens, tape, y_views=None, dy_view_fns=None):
  """Runs 1 or 2 `RNNDirection`s of the same cell/size with ONE kernel launch per time step
  (os2s_rnn_layer_{fwd,bwd}_multi). `xs` is a list of Act shared by the directions, or one
  such list per direction. Returns one Act per direction."""
  d0 = dirs[0]
  H = d0.H
  training = tape is not None
  y_views = y_views or [None] * len(dirs)
  xs_per = xs if isinstance(xs[0], (list, tuple)) else [xs] * len(dirs)
  gxs = [d.input_projection(x) for d, x in zip(dirs, xs_per)]
  res = capi.rnn_layer_fwd_multi(
      d0.cell, [dict(gx=gx, wh=d.wh.w16.view(d.G * H, H),
                     bh=d.bh.master if d.bh is not None else None, y=yv, reverse=d.reverse)
                for d, gx, yv in zip(dirs, gxs, y_views)],
      lens, H, d0.forget_bias, save=training)
  outs = [Act(r[0], lens) for r in res]
  if not training:
    return outs

  def backward():
    dys = [f() for f in dy_view_fns] if dy_view_fns is not None else [o.grad for o in outs]
    assert all(g is not None for g in dys)
    grads = capi.rnn_layer_bwd_multi(
        d0.cell, [dict(whT=d.wh.wt16.view(H, d.G * H), dy=dy, y=r[0], gates=r[1], c_seq=r[2],
                       reverse=d.reverse) for d, dy, r in zip(dirs, dys, res)],
        lens, H, d0.forget_bias)
    for d, x, r, (dgx, dgr) in zip(dirs, xs_per, res, grads):
      d.weight_backward(x, lens, r[0], dgx, dgr)
    for o in outs:
      o.grad = None

  tape.record(backward, [p for d in dirs for p in d.params()])
  return outs


class BiRNNStack(object):
  """num_layers x (forward + backward direction). The two directions of a layer write the two
  halves of ONE [B,T,2H] tensor (no concat), which the next layer / the following dense
  layer consumes as a single input (cuDNN 'bidirectional' stacking;
  tf.bidirectional_dynamic_rnn + tf.concat, ds2_encoder.py:294-380). Optional dropout between
  layers (cuDNN RNN `dropout`)."""

  def __init__(self, store, name, cell, input_size, hidden, num_layers, bidirectional=True,
               forget_bias=1.0):
    self.layers = []
    self.H, self.ndir = hidden, 2 if bidirectional else 1
    in_size = input_size
    for l in range(num_layers):
      dirs = [RNNDirection(store, "%s/layer_%d/fw" % (name, l), cell, [in_size], hidden, False,
                           forget_bias)]
      if bidirectional:
        dirs.append(RNNDirection(store, "%s/layer_%d/bw" % (name, l), cell, [in_size], hidden,
                                 True, forget_bias))
      self.layers.append(dirs)
      in_size = hidden * self.ndir
    self.output_dim = in_size

  def forward(self, x, lens, tape, keep_prob=1.0, seeds=None):
    """x: Act [B,T,In] -> Act [B,T,ndir*H]."""
    H = self.H
    cur = x
    for li, dirs in enumerate(self.layers):
      B, T, _ = cur.data.shape
      ybuf = (torch.zeros if lens is not None else torch.empty)(
          (B, T, H * self.ndir), dtype=torch.bfloat16, device=cur.data.device)
      out = Act(ybuf, lens)
      rnn_directions_forward(
          dirs, [cur], lens, tape, [ybuf[:, :, d * H:(d + 1) * H] for d in range(len(dirs))],
          [(lambda o=out, d=d: o.grad[:, :, d * H:(d + 1) * H]) for d in range(len(dirs))])
      cur = out
      if keep_prob < 1.0 and li < len(self.layers) - 1:
        seed = seeds.next() if seeds is not None else li + 1
        dropped = Act(capi.dropout_bwd(cur.data.view(-1, cur.data.shape[-1]), keep_prob,
                                       seed=seed).view_as(cur.data), lens)
        if tape is not None:
          src = cur

          def backward(dropped=dropped, src=src, seed=seed):
            g = capi.dropout_bwd(dropped.grad.view(-1, dropped.grad.shape[-1]), keep_prob,
                                 seed=seed).view_as(dropped.grad)
            src.grad, src.grad_init = g, True
            dropped.grad = None

          tape.record(backward)
        cur = dropped
    return cur

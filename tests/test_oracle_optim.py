"""CPU: optimizer oracle pins. Reproduces the reference's own relations:
 * mp_wrapper_test.py:54-56,93-95 — an L2 regulariser gradient of 1e-8 vanishes
   in half precision without the wrapper and equals 1e-8 (atol 1e-11) when the
   regulariser acts on the fp32 master copy;
 * lr policies against closed forms; Backoff scaler trajectory
   (automatic_loss_scaler.py:78-106)."""
import numpy as np

from oracle import optim


def test_mp_regulariser_gradient_1e8():
  # fp16(1e-8 * w) underflows; on the fp32 master it survives: grad = scale_reg * w
  w = np.float32(1.0)
  reg = np.float32(1e-8)
  assert np.float16(reg * np.float16(w)) == 0.0
  o = optim.RefOptimizer([np.array([w])], optimizer="SGD", lr_fn=lambda s: 1.0, l2=[reg])
  o.step([np.zeros(1, np.float32)])
  g = (1.0 - o.w[0][0])  # lr = 1 -> applied gradient
  assert abs(g - 1e-8) < 1e-11 or abs(float(np.float32(w) - np.float32(w - reg)) - g) < 1e-11


def test_lr_policies():
  assert optim.poly_decay(0, 0.02, 1000, power=2.0, min_lr=1e-5) == 0.02
  assert abs(optim.poly_decay(500, 0.02, 1000, power=2.0, min_lr=1e-5) -
             ((0.02 - 1e-5) * 0.25 + 1e-5)) < 1e-12
  assert optim.poly_decay(5000, 0.02, 1000, power=2.0, min_lr=1e-5) == 1e-5
  # transformer policy: peak at step = warmup-1
  lr = [optim.transformer_policy(s, 2.0, 1024, 8000) for s in (0, 7999, 8000, 100000)]
  assert lr[0] < lr[1] and lr[1] >= lr[2] > lr[3]
  assert abs(lr[1] - 2.0 * 1024 ** -0.5 * 8000 ** -0.5) < 1e-9
  assert optim.exp_decay(10, 1.0, 5, 0.5, True) == 0.25


def test_backoff_trajectory():
  s = optim.BackoffScaler(step_window=4)
  assert s.scale == 2.0 ** 14
  assert s.update(True, 1.0) is True and s.scale == 2.0 ** 13
  scales = []
  for _ in range(9):
    s.update(False, 1.0)
    scales.append(float(s.scale))
  # doubles when (iter - last_overflow) % 4 == 0: iterations 4 and 8 (last_overflow=0)
  assert scales == [2.0 ** 13] * 3 + [2.0 ** 14] * 6


def test_novograd_second_moment_follows_the_reference_graph_as_written():
  """novograd.py:107-113: `self._grads_ema[i] = tf.cond(tf.equal(self._grads_ema[i], 0.), lambda:
  g_2, lambda: ema*beta2 + g_2*(1-beta2))` rebinds the Python list entry; the variable
  nvgrad2_ema<i> is never assigned and stays 0, so each session.run takes the g_2 branch:
  v_t = |g_t|^2 and beta2 has no effect. The oracle's default must reproduce exactly that; the
  `ema_second_moment` switch (not a reference parameter) gives the published moving average."""
  import numpy as np
  from oracle import optim as oopt
  w0 = [np.full((4,), 0.5, np.float32)]
  g1 = [np.array([3.0, 0.0, 0.0, 4.0], np.float32)]      # |g|^2 = 25
  g2 = [np.array([0.0, 1.0, 0.0, 0.0], np.float32)]      # |g|^2 = 1
  for b2 in (0.5, 0.98):
    ref = oopt.RefOptimizer(w0, optimizer="NovoGrad", opt_params=dict(beta1=0.0, beta2=b2, epsilon=0.0),
                            lr_fn=lambda s: 1.0)
    ref.step(g1); ref.step(g2)
    assert float(ref.ema[0]) == 1.0                       # not 0.5*25 + 0.5*1
    # beta1 = 0, lr = 1: the second update is exactly g2 / sqrt(|g2|^2)
    np.testing.assert_allclose(ref.w[0], np.array([0.5 - 0.6, 0.5 - 1.0, 0.5, 0.5 - 0.8], np.float32), rtol=1e-6)
  ema = oopt.RefOptimizer(w0, optimizer="NovoGrad",
                          opt_params=dict(beta1=0.0, beta2=0.5, epsilon=0.0, ema_second_moment=True),
                          lr_fn=lambda s: 1.0)
  ema.step(g1); ema.step(g2)
  assert float(ema.ema[0]) == 13.0


def test_piecewise_constant_and_inv_poly_closed_forms():
  """lr_policies.py:30-57 / :203-245. tf.train.piecewise_constant switches AFTER a boundary step
  (x <= b uses the earlier value); inv_poly_decay hits learning_rate at step 0 and min_lr at decay_steps."""
  from oracle import optim as o
  pw = dict(learning_rate=0.1, boundaries=[3, 5], decay_rates=[0.5, 0.25])
  got = [o.piecewise_constant(s, **pw) for s in range(8)]
  assert got == [0.1, 0.1, 0.1, 0.1, 0.05, 0.05, 0.025, 0.025], got
  ep = dict(learning_rate=1.0, boundaries=[1, 2], decay_rates=[0.1, 0.01], steps_per_epoch=10)
  assert o.piecewise_constant(10, **ep) == 1.0 and o.piecewise_constant(11, **ep) == 0.1
  assert o.piecewise_constant(20, **ep) == 0.1 and o.piecewise_constant(21, **ep) == 0.01
  ip = dict(learning_rate=0.03, decay_steps=100, min_lr=1e-4, power=0.5)
  assert abs(o.inv_poly_decay(0, **ip) - 0.03) < 1e-12
  assert abs(o.inv_poly_decay(100, **ip) - 1e-4) < 1e-12
  assert o.inv_poly_decay(50, **ip) < o.inv_poly_decay(49, **ip)
  # min_lr is clamped into [1e-8, learning_rate]
  assert abs(o.inv_poly_decay(100, 0.03, 100, 0.0, 1.0) - 1e-8) < 1e-15
  assert abs(o.inv_poly_decay(100, 0.03, 100, 1.0, 1.0) - 0.03) < 1e-12
  # the product's host-side policies are the same functions of the step
  from openseq2seq_amd.optimizers import lr_policies as lp
  for s in range(0, 130, 7):
    assert abs(lp.inv_poly_decay(s, **ip) - o.inv_poly_decay(s, **ip)) < 1e-15
    assert lp.piecewise_constant(s, **ep) == o.piecewise_constant(s, **ep)

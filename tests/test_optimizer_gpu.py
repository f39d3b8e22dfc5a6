"""GPU parity: os2s_opt_step (multi-tensor MP optimizer) vs the NumPy oracle
(oracle/optim.py) over several steps, including an overflow/skip step.
fp32 elementwise math; per-tensor norms are reduced in a different order =>
rtol 2e-4."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import optim as oopt  # noqa: E402

SHAPES = [(11, 40, 24), (40,), (40,), (5000,), (3, 16, 8), (1, 29, 64), (29,)]
_LAST = {}


def _setup(cuda, need_m2=False):
  from openseq2seq_amd.optimizers.flat_params import FlatParams
  rng = np.random.RandomState(0)
  store = FlatParams(cuda)
  ws = []
  for i, s in enumerate(SHAPES):
    w = (rng.randn(*s) * 0.1).astype(np.float32)
    ws.append(w)
    store.add("v%d" % i, s, w, kind="conv" if len(s) == 3 else "vector")
  store.finalize(need_m2=need_m2)
  return store, ws, rng


def _run(cuda, optimizer, opt_params, lr_fn, lr_params, larc=None, clip=None,
         loss_scaling="Backoff", need_m2=False, nsteps=6, inf_step=2, world=1, l2=None):
  from openseq2seq_amd.optimizers import lr_policies
  from openseq2seq_amd.optimizers.optimizers import optimize_loss
  store, ws, rng = _setup(cuda, need_m2)
  _LAST["store"] = store
  if l2:
    store.tensor_l2.copy_(torch.tensor(l2))
    store.l2_active = True
  op = optimize_loss(store, optimizer, opt_params, getattr(lr_policies, lr_fn), lr_params,
                     larc_params=larc, clip_gradients=clip, loss_scaling=loss_scaling,
                     world_size=world)
  scaler = None
  if loss_scaling == "Backoff":
    scaler = oopt.BackoffScaler()
  elif loss_scaling == "LogMax":
    scaler = oopt.LogMaxScaler()
  name = optimizer if isinstance(optimizer, str) else optimizer.__name__
  ref = oopt.RefOptimizer(ws, optimizer=name, opt_params=opt_params,
                          lr_fn=lambda s: getattr(oopt, lr_fn)(s, **lr_params),
                          larc_params=larc, clip_gradients=clip, scaler=scaler, l2=l2,
                          world_size=world)
  if scaler is None:
    ref.static_scale = np.float32(loss_scaling)
  for step in range(nsteps):
    scale = float(ref.loss_scale)
    st = op.read_state()
    assert abs(st["loss_scale"] - scale) <= 1e-6 * scale, (step, st["loss_scale"], scale)
    grads = [(rng.randn(*s) * 0.01).astype(np.float32) * scale * world for s in SHAPES]
    if step == inf_step:
      grads[3][17] = np.inf
    for p, g in zip(store.params, grads):
      p.grad.copy_(torch.from_numpy(g))
    op.run()
    skipped = ref.step(grads)
    st = op.read_state()
    assert bool(st["skip"]) == skipped, step
    assert st["global_step"] == ref.global_step
    for p, w in zip(store.params, ref.w):
      torch.testing.assert_close(p.master.cpu(), torch.from_numpy(w), rtol=2e-4, atol=1e-6)
      torch.testing.assert_close(p.w16.float().cpu(),
                                 torch.from_numpy(w).to(torch.bfloat16).float(),
                                 rtol=1e-2, atol=1e-3)
  # dgrad copies follow the bf16 weights: wT[k'][ci][co] = w[K-1-k'][co][ci]
  p0 = store.params[0]
  torch.testing.assert_close(p0.wt16.float().cpu(),
                             p0.w16.float().cpu().flip(0).permute(0, 2, 1), rtol=0, atol=0)
  return op.read_state()


def test_novograd_larc_backoff_jasper_cfg(cuda):
  from openseq2seq_amd.optimizers.novograd import NovoGrad
  st = _run(cuda, NovoGrad,
            dict(beta1=0.95, beta2=0.98, epsilon=1e-8, weight_decay=0.001, grad_averaging=False),
            "poly_decay", dict(learning_rate=0.02, min_lr=1e-5, power=2.0, decay_steps=1000),
            larc=dict(larc_eta=0.001))
  assert st["num_skipped"] == 1 and st["loss_scale"] == 2.0 ** 13


def test_novograd_second_moment_modes(cuda):
  """Default = the reference graph as written (v_t = |g_t|^2 every step, beta2 dead:
  novograd.py:107-113 never assigns nvgrad2_ema*); ema_second_moment=True = the moving average of
  the published algorithm. Both against the oracle; the two trajectories must differ."""
  from openseq2seq_amd.optimizers.novograd import NovoGrad
  outs = []
  for ema in (False, True):
    _run(cuda, NovoGrad, dict(beta1=0.95, beta2=0.5, epsilon=1e-8, weight_decay=0.001,
                              ema_second_moment=ema),
         "poly_decay", dict(learning_rate=0.02, min_lr=1e-5, power=2.0, decay_steps=1000),
         larc=dict(larc_eta=0.001), inf_step=-1)
    outs.append(_LAST["store"].master.clone())
  assert not torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("scaling", ["Backoff", "LogMax"])
def test_inf_gradient_under_global_norm_clip_is_an_overflow(cuda, scaling):
  """An Inf gradient with max_grad_norm set (tacotron_mixed: clip 1 + Backoff): the global norm is
  inf, the clip factor 0, the clipped gradient inf * 0 = NaN -> the reference's check_grads
  reports has_nan and the step is skipped (optimizers.py:388-482, mp_wrapper.py:114-120). The
  weights must stay finite and the scaler must see an overflow."""
  st = _run(cuda, "Momentum", dict(momentum=0.9), "fixed_lr", dict(learning_rate=0.01),
            clip=1.0, loss_scaling=scaling, inf_step=1, nsteps=4)
  assert st["num_skipped"] == 1
  assert bool(torch.isfinite(_LAST["store"].master).all())


def test_adam_transformer_policy(cuda):
  _run(cuda, "Adam", dict(beta1=0.9, beta2=0.997, epsilon=1e-9), "transformer_policy",
       dict(learning_rate=2.0, warmup_steps=8000, d_model=1024), need_m2=True, world=4)


def test_momentum_clip_static_scale(cuda):
  _run(cuda, "Momentum", dict(momentum=0.9), "exp_decay",
       dict(learning_rate=0.01, decay_steps=2, decay_rate=0.5, use_staircase_decay=True,
            min_lr=1e-4), clip=0.5, loss_scaling=128.0, inf_step=-1,
       l2=[5e-4, 0, 0, 5e-4, 0, 0, 0])


def test_sgd_logmax(cuda):
  _run(cuda, "SGD", {}, "fixed_lr", dict(learning_rate=0.1), loss_scaling="LogMax",
       inf_step=1)


def test_iter_size_algebra_kat(cuda):
  """The reference's iter_size known answer (optimizers/optimizers_test.py:27-80): a linear
  least-squares model, SGD lr 0.1, iter_size 8; after accumulating 4 micro-steps and then
  applying, var == v - 0.1 * 4 * (grad / 8), the accumulator is zero again, and nothing
  moves during the accumulating micro-steps (fp32 masters: atol 1e-6; the reference states 1e-7 on its fp64-free TF graph)."""
  from openseq2seq_amd.optimizers import lr_policies
  from openseq2seq_amd.optimizers.flat_params import FlatParams
  from openseq2seq_amd.optimizers.optimizers import optimize_loss
  n_samples = n_hid = 10
  iter_size = 8
  np.random.seed(0)
  X = np.random.rand(n_samples, n_hid)
  y = np.random.rand(n_samples, 1)
  store = FlatParams(cuda)
  w0 = (np.random.rand(n_hid, 1).astype(np.float32) - 0.5)
  p = store.add("dense/kernel", (n_hid, 1), w0, kind="vector")
  store.finalize()
  op = optimize_loss(store, "SGD", {}, lr_policies.fixed_lr, dict(learning_rate=0.1), dtype="float32",
                     loss_scaling=1.0, iter_size=iter_size, on_horovod=True, world_size=1)
  for _ in range(3):
    v = p.master.cpu().numpy().astype(np.float64)
    store.zero_grads()
    assert float(p.grad.abs().max()) == 0.0
    g = 2 * (X.T.dot(X).dot(v) - X.T.dot(y)) / X.shape[0]     # d(mse)/dw of one micro-step
    true_g = g / iter_size
    gt = torch.from_numpy(g.astype(np.float32)).to(cuda)
    for k in range(1, 5):                                       # four accumulating micro-steps
      p.grad.add_(gt)
      np.testing.assert_allclose(p.grad.cpu().numpy() / iter_size, true_g * k, atol=1e-6)
      np.testing.assert_allclose(p.master.cpu().numpy(), v)
    op.run()
    np.testing.assert_allclose(p.master.cpu().numpy(), v - 0.1 * true_g * 4, atol=1e-6)
  with pytest.raises(ValueError):
    optimize_loss(store, "SGD", {}, lr_policies.fixed_lr, dict(learning_rate=0.1), iter_size=2)


def test_conv_weight_dgrad_copies_every_shape_class(cuda):
  """wT[k'][ci][co] = w[K-1-k'][co][ci], bit-exact, for the 16-byte-piece path (channel counts that are
  multiples of 8: whole 64 x 64 tiles, ragged tiles, a single tile) and for the element path (a channel
  count that is not a multiple of 8), in one batched launch as the optimizer step issues it."""
  from openseq2seq_amd.optimizers.flat_params import FlatParams
  rng = np.random.RandomState(3)
  shapes = [(3, 128, 64), (5, 136, 72), (1, 8, 1024), (2, 29, 64), (11, 40, 24), (1, 1024, 8), (4, 200, 264)]
  store = FlatParams(cuda)
  for i, s in enumerate(shapes):
    store.add("w%d" % i, s, (rng.randn(*s)).astype(np.float32), kind="conv")
  store.finalize()
  torch.cuda.synchronize()
  for p in store.params:
    want = p.w16.float().cpu().flip(0).permute(0, 2, 1)
    torch.testing.assert_close(p.wt16.float().cpu(), want, rtol=0, atol=0)
  # and again after the weights change (the refresh every optimizer step ends with)
  store.master.mul_(-0.5)
  store.refresh_compute_copies()
  torch.cuda.synchronize()
  for p in store.params:
    want = p.w16.float().cpu().flip(0).permute(0, 2, 1)
    torch.testing.assert_close(p.wt16.float().cpu(), want, rtol=0, atol=0)


def test_lr_policies_cosine_piecewise_inv_poly_on_device(cuda):
  """The three lr policies the earlier tests never ran on the GPU (device ids 4, 5, 6 of
  os2s_opt_config_t.lr_policy: lr_policies.py:134-170, :30-57, :203-245): weights after 12 Momentum
  steps follow the oracle, and the lr the kernel latched at every step equals the host formula
  (warm-up, begin_decay_at, boundary steps themselves, the far end of the schedule)."""
  cases = [
      ("cosine_decay", dict(learning_rate=0.05, decay_steps=9, min_lr=0.1, begin_decay_at=2, warmup_steps=2)),
      ("cosine_decay", dict(learning_rate=0.05, decay_steps=5)),
      ("piecewise_constant", dict(learning_rate=0.04, boundaries=[2, 3, 7], decay_rates=[0.5, 0.1, 0.02])),
      ("piecewise_constant", dict(learning_rate=0.04, boundaries=[1, 2], decay_rates=[0.3, 0.05], steps_per_epoch=3)),
      ("inv_poly_decay", dict(learning_rate=0.03, decay_steps=10, min_lr=1e-4, power=0.75)),
      ("inv_poly_decay", dict(learning_rate=0.03, decay_steps=4, min_lr=0.0, power=2.0)),
  ]
  from openseq2seq_amd.optimizers import lr_policies
  for lr_fn, lr_params in cases:
    st = _run(cuda, "Momentum", dict(momentum=0.9), lr_fn, lr_params, loss_scaling=1.0, nsteps=12, inf_step=-1)
    assert st["global_step"] == 12
    # the lr of the LAST applied step (global step 11 at the time of the update)
    want = getattr(oopt, lr_fn)(11, **lr_params)
    assert abs(st["lr"] - want) <= 2e-6 * max(abs(want), 1e-6) + 1e-12, (lr_fn, lr_params, st["lr"], want)
    # product-side host function == oracle at every step of the schedule
    for step in range(0, 40):
      a, b = getattr(lr_policies, lr_fn)(step, **lr_params), getattr(oopt, lr_fn)(step, **lr_params)
      assert abs(a - b) <= 1e-12 + 1e-9 * abs(b), (lr_fn, step, a, b)


def test_piecewise_constant_rejects_too_many_boundaries(cuda):
  from openseq2seq_amd.optimizers import lr_policies
  with pytest.raises(ValueError):
    lr_policies.device_policy(lr_policies.piecewise_constant,
                              dict(learning_rate=0.1, boundaries=list(range(1, 19)), decay_rates=[0.5] * 18))


def test_asynchronous_update_is_bit_identical_to_the_one_stream_step(cuda):
  """TrainOp.run_async ('os2s_async_optimizer': True / OS2S_ASYNC_OPT=1): os2s_opt_prepare + eight os2s_opt_apply_range
  launches on the optimizer stream, an event behind each, the next forward pass waiting per variable for the range
  that holds it, every gradient chunk zeroed behind its last read. Same kernels, same order per element: after 5
  steps (one of them skipped on an injected Inf) master weights, bf16 copies, moments, transposed data-gradient
  copies, loss scale and losses are BIT-equal to the one-stream step, and the gradient buffer is zero."""
  from openseq2seq_amd import capi
  from openseq2seq_amd.configs.jasper import jasper10x5_config
  from openseq2seq_amd.configs.transformer import transformer_config

  def models():
    cls, p = jasper10x5_config(batch_size_per_gpu=4, use_horovod=False, max_steps=100)
    p["encoder_params"]["convnet_layers"] = [
        {"type": "conv1d", "repeat": 1, "kernel_size": [11], "stride": [2], "num_channels": 128, "padding": "SAME",
         "dilation": [1], "dropout_keep_prob": 0.9},
        {"type": "conv1d", "repeat": 2, "kernel_size": [11], "stride": [1], "num_channels": 128, "padding": "SAME",
         "dilation": [1], "dropout_keep_prob": 0.9, "residual": True, "residual_dense": True},
        {"type": "conv1d", "repeat": 2, "kernel_size": [13], "stride": [1], "num_channels": 384, "padding": "SAME",
         "dilation": [1], "dropout_keep_prob": 0.9, "residual": True, "residual_dense": True},
        {"type": "conv1d", "repeat": 1, "kernel_size": [1], "stride": [1], "num_channels": 256, "padding": "SAME",
         "dilation": [1], "dropout_keep_prob": 0.9},
    ]
    yield "jasper", cls, p
    cls, p = transformer_config(d_model=512, num_layers=2, num_heads=8, batch_size_per_gpu=16, vocab_size=1024,
                                max_length=24, max_steps=1000)
    yield "transformer", cls, p

  def run(cls, p, async_opt):
    import copy
    import random
    import numpy as np
    torch.manual_seed(5); np.random.seed(5); random.seed(5)
    p = copy.deepcopy(p)
    p["os2s_async_optimizer"] = async_opt
    p["random_seed"] = 5                # (unset, the model seeds weights and dropout from the wall clock: model.py:309-313)
    m = cls(p, mode="train", hvd=None, device=cuda)
    m.compile()
    dl = m.get_data_layer()
    losses = []
    for s in range(5):
      if s == 3:        # an overflow: the step is skipped, the loss scale halves, the Inf must not survive in the buffer
        orig = m._forward_backward

        def poisoned(batch, tape, orig=orig, m=m):
          out = orig(batch, tape)
          tape.record(lambda: m.store.params[0].grad.view(-1)[:1].fill_(float("inf")))
          return out
        m._forward_backward = poisoned
      losses.append(float(m.train_step(dl.synthetic_batch(cuda, seed=60 + s)).cpu()[0]))
      if s == 3:
        m._forward_backward = orig
    torch.cuda.synchronize()
    st = m.train_op.read_state()
    s_ = m.store
    out = dict(losses=losses, master=s_.master.clone(), w16=s_.w16.clone(), m1=s_.m1.clone(), wt16=s_.wt16.clone(),
               scale=st["loss_scale"], skipped=st["num_skipped"], step=st["global_step"],
               grads_abs=float(s_.grads.abs().sum()) if async_opt else 0.0)
    del m
    return out

  det = capi.deterministic()
  capi.set_deterministic(True)
  try:
    for name, cls, p in models():
      a, b = run(cls, p, False), run(cls, p, True)
      assert a["losses"] == b["losses"], (name, a["losses"], b["losses"])
      for k in ("master", "w16", "m1", "wt16"):
        assert torch.equal(a[k], b[k]), (name, k)
      assert a["scale"] == b["scale"] and a["skipped"] == b["skipped"] == 1 and a["step"] == b["step"], (name, a, b)
      assert b["grads_abs"] == 0.0, (name, b["grads_abs"])
  finally:
    capi.set_deterministic(det)

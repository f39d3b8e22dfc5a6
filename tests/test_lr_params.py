"""Model.compile's defaults for lr_policy_params (models/model.py:479-495 of the reference): the schedule a
config written in EPOCHS gets. Host logic only; the policies themselves are checked against the oracle in
test_oracle_optim.py and on the device in test_optimizer_gpu.py."""
import numpy as np

from openseq2seq_amd.models.model import resolve_lr_params
from openseq2seq_amd.optimizers import lr_policies
from oracle import optim as ooptim


def _never():
  raise AssertionError("not needed for this policy")


def test_piecewise_constant_boundaries_in_epochs_get_the_epoch_length():
  # the way the reference uses the policy: boundaries in epochs + num_epochs in the config
  cfg = dict(learning_rate=0.1, boundaries=[30, 60, 80], decay_rates=[0.1, 0.01, 0.001])
  lp = resolve_lr_params(lr_policies.piecewise_constant, cfg, _never, lambda: 250, True)
  assert lp['steps_per_epoch'] == 250 and 'decay_steps' not in lp
  assert cfg == dict(learning_rate=0.1, boundaries=[30, 60, 80], decay_rates=[0.1, 0.01, 0.001])   # not modified
  # the learning rate then drops after epoch 30, not after step 30 (tf.train.piecewise_constant: x <= boundary)
  for step, want in ((31, 0.1), (30 * 250, 0.1), (30 * 250 + 1, 0.01), (60 * 250 + 1, 0.001), (80 * 250 + 7, 0.0001)):
    got = ooptim.piecewise_constant(step, **lp)
    np.testing.assert_allclose(got, want, rtol=1e-6)
  # a run configured in steps (no num_epochs), or a data layer that cannot tell its size: boundaries stay steps
  assert 'steps_per_epoch' not in resolve_lr_params(lr_policies.piecewise_constant, cfg, _never, lambda: 250, False)
  assert 'steps_per_epoch' not in resolve_lr_params(lr_policies.piecewise_constant, cfg, _never, lambda: None, True)
  # an explicit value wins
  lp = resolve_lr_params(lr_policies.piecewise_constant, dict(cfg, steps_per_epoch=7), _never, lambda: 250, True)
  assert lp['steps_per_epoch'] == 7


def test_decay_starts_after_the_warm_up_and_runs_over_what_is_left():
  lp = resolve_lr_params(lr_policies.poly_decay, dict(learning_rate=0.02, power=2.0, warmup_steps=800,
                                                      begin_decay_at=300), lambda: 10000, _never, False)
  assert lp['begin_decay_at'] == 800 and lp['decay_steps'] == 9200
  lp = resolve_lr_params(lr_policies.poly_decay, dict(learning_rate=0.02, power=2.0, begin_decay_at=1500,
                                                      warmup_steps=800), lambda: 10000, _never, False)
  assert lp['begin_decay_at'] == 1500 and lp['decay_steps'] == 8500
  # decay_steps given: nothing is touched
  lp = resolve_lr_params(lr_policies.poly_decay, dict(learning_rate=0.02, decay_steps=400, warmup_steps=800),
                         _never, _never, False)
  assert lp == dict(learning_rate=0.02, decay_steps=400, warmup_steps=800)
  # a policy without decay_steps
  lp = resolve_lr_params(lr_policies.fixed_lr, dict(learning_rate=0.3), _never, _never, True)
  assert lp == dict(learning_rate=0.3)

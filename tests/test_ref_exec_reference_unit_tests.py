"""The reference's OWN unit tests, run where they lie on the TensorFlow-1 stand-in (oracle/ref_shim/tf1).

A check of the stand-in, not of this repository's product: the primitives it restates have to satisfy what the
reference's authors assert about their own code. tests/golden/run_reference_unit_tests.py loads
parts/transformer/utils_test.py (padding / bias known answers), parts/transformer/beam_search_test.py (beam tensor
plumbing incl. tf.nn.top_k + gather_nd), losses/sequence_loss_test.py (sparse vs smoothed cross entropy, 36 parameter
combinations), optimizers/mp_wrapper_test.py (regulariser gradient 1e-8 under the mixed-precision wrapper incl. the
'Const_1:0' op-name assertion, 2 x 6000 steps of least-squares convergence through optimize_loss in fp32 and mixed
precision) and optimizers/optimizers_test.py (the iter_size accumulate / apply algebra of the Horovod branch, with a
one-rank stand-in for horovod.tensorflow) and — on the stand-ins for librosa / python_speech_features / resampy
(oracle/ref_shim/audio_libs) — data/speech2text/speech_utils_test.py (108 feature extractions from the reference's toy
wav files: shapes padded to a multiple of 8, mean 0 / std 1 to six places, the num_features assertion; 200 random speed
perturbations within their length bounds; augmentation changes the features), and runs them with unittest; NumPy's
generator is seeded before each test (two of the reference's augmentation assertions hold for most random draws, not
for all). All 18 must pass."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(not os.path.isdir("/root/reference/open_seq2seq"), reason="reference checkout not present")
def test_reference_unit_tests_pass_on_the_stand_in():
  r = subprocess.run([sys.executable, os.path.join(HERE, "golden", "run_reference_unit_tests.py")],
                     capture_output=True, text=True, timeout=1200)
  lines = [l.split() for l in r.stdout.strip().splitlines() if "::" in l and "::test_session" not in l]
  assert r.returncode == 0 and len(lines) == 18, r.stdout + r.stderr[-2000:]
  bad = [l for l in lines if l[1] != "PASS"]
  assert not bad, bad
  names = {l[0] for l in lines}
  assert {"optimizers.mp_wrapper_test::test_regularization_mixed", "optimizers.optimizers_test::test_updates",
          "optimizers.mp_wrapper_test::test_convergence", "parts.transformer.beam_search_test::test_gather_topk_beams",
          "losses.sequence_loss_test::test_compute_loss",
          "data.speech2text.speech_utils_test::test_get_speech_features_from_file"} <= names

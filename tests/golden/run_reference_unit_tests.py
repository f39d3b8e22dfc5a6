"""Runs the REFERENCE'S OWN unit-test files (/root/reference/open_seq2seq/**/*_test.py, where they lie) on the
TensorFlow-1 stand-in oracle/ref_shim/tf1 — a check of the STAND-IN: the primitives it restates must satisfy what the
reference's authors assert about their own code (known-answer matrices of the Transformer utilities, beam-search
tensor plumbing, the sparse / smoothed cross-entropy relation, the regulariser gradient under the mixed-precision
wrapper, least-squares convergence of optimize_loss in fp32 and mixed precision, the iter_size accumulate / apply
algebra of the Horovod branch with a one-rank stand-in for horovod.tensorflow).

    python tests/golden/run_reference_unit_tests.py            # prints one line per test: module::test PASS|FAIL|SKIP

Not part of the product and not an oracle: tests/test_ref_exec_reference_unit_tests.py calls it when the reference
checkout is present."""
import importlib
import io
import os

import numpy as np
import sys
import types
import unittest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_ref_exec as gen  # noqa: E402

MODULES = ["open_seq2seq.parts.transformer.utils_test", "open_seq2seq.parts.transformer.beam_search_test",
           "open_seq2seq.losses.sequence_loss_test", "open_seq2seq.optimizers.mp_wrapper_test",
           "open_seq2seq.optimizers.optimizers_test",
           # the feature front end, on the stand-ins for librosa / python_speech_features / resampy
           # (oracle/ref_shim/audio_libs); reads the reference's toy wav files by paths relative to its root
           "open_seq2seq.data.speech2text.speech_utils_test"]


def main():
  tf, imp = gen._install()
  # one rank of Horovod (optimizers.py:77-104 imports allreduce / size inside reduce_gradients)
  hvd = types.ModuleType("horovod.tensorflow")
  hvd.init, hvd.size, hvd.rank, hvd.allreduce = (lambda: None), (lambda: 1), (lambda: 0), (lambda g, **k: g)
  sys.modules["horovod"] = types.ModuleType("horovod")
  sys.modules["horovod.tensorflow"] = hvd
  sys.modules["horovod"].tensorflow = hvd
  opt = importlib.import_module("open_seq2seq.optimizers.optimizers")
  sys.modules["open_seq2seq.optimizers"].optimize_loss = opt.optimize_loss
  sys.path.insert(0, os.path.join(gen.REPO, "oracle", "ref_shim"))
  import audio_libs
  audio_libs.install()
  os.chdir(os.path.dirname(gen.PKG))
  results = []
  for name in MODULES:
    mod = importlib.import_module(name)
    suite = unittest.defaultTestLoader.loadTestsFromModule(mod)
    for case in suite:
      for test in case:
        res = unittest.TestResult()
        buf, old = io.StringIO(), sys.stdout
        sys.stdout = buf
        np.random.seed(7)             # the augmentation tests draw unseeded random stretches (one asserts that a random
        try:                          # stretch changes the padded length: true for most draws, not for all)
          test.run(res)
        finally:
          sys.stdout = old
        tag = "PASS" if res.wasSuccessful() and not res.skipped else ("SKIP" if res.skipped else "FAIL")
        results.append((name.replace("open_seq2seq.", "") + "::" + test._testMethodName, tag,
                        (res.failures + res.errors)[0][1].strip().splitlines()[-1] if tag == "FAIL" else ""))
  for r in results:
    print("%s %s %s" % (r[0], r[1], r[2]))
  return 0


if __name__ == "__main__":
  sys.exit(main())

"""Generates the committed golden fixtures under tests/golden/ from the
reference checkout (only runnable where /root/reference exists).

  ctc_test_logits.npy  <- ctc_decoder_with_lm/ctc-test.pickle  (float32 [184,1,29])
  ctc_test_meta.json   <- the expectations asserted by the reference's own test
                          ctc_decoder_with_lm/ctc-test.py:60-78 and the vocabulary
                          open_seq2seq/test_utils/toy_speech_data/vocab.txt
  ctc_test_lm.binary   <- ctc_decoder_with_lm/ctc-test-lm.binary (1.4 KB KenLM bigram model, data)
  ctc_test_lm.trie     <- ctc_decoder_with_lm/ctc-test-lm.trie   (1.1 KB letter trie, data)
  toy_data_lm.binary   <- open_seq2seq/test_utils/toy_speech_data/toy_data-lm.binary (7.6 KB KenLM
                          trigram model in the probing layout, data)
  ../../open_seq2seq/test_utils/toy_speech_data/{toy_data.csv, wav_files/*.wav}
                       <- the reference's toy speech corpus (8 WSJ utterances, 1.6 MB, data): the input of
                          its ASR acceptance tests (models/speech2text_test.py, scripts/run_all_tests.sh:71-83).
                          Kept at the path the reference's configs name, relative to the repository root, so
                          that `python run.py --config_file=open_seq2seq/test_utils/test_speech_configs/...`
                          runs as in the reference (tests/test_speech_acceptance_gpu.py).
"""
import json
import os
import pickle
import shutil

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
  with open(os.path.join(REF, "ctc_decoder_with_lm", "ctc-test.pickle"), "rb") as f:
    seq, label = pickle.load(f, encoding="latin1")
  seq = np.asarray(seq, dtype=np.float32)
  assert seq.shape == (184, 1, 29)
  np.save(os.path.join(OUT, "ctc_test_logits.npy"), seq)
  with open(os.path.join(REF, "open_seq2seq", "test_utils", "toy_speech_data",
                         "vocab.txt")) as f:
    vocab = [line[0] for line in f.read().split("\n") if len(line) > 0]
  meta = {
      "source": "ctc_decoder_with_lm/ctc-test.pickle",
      "label": str(label),
      "vocab": vocab,
      # ctc_decoder_with_lm/ctc-test.py:66-67
      "greedy_text": "then seconds",
      "greedy_neg_sum_logits": -7079.117,
      # ctc_decoder_with_lm/ctc-test.py:36-78: beam width 16, merge_repeated False
      "beam_width": 16,
      "beam_text": "then seconds",
      "beam_log_prob": -1.1842575,
      "lm_text": "ten seconds",
      "lm_log_prob": -4.619581,
      "lm_alpha": 2.0, "lm_beta": 0.5, "lm_trie_weight": 0.1,
      "tol": 1e-3,
  }
  for src, dst in (("ctc-test-lm.binary", "ctc_test_lm.binary"), ("ctc-test-lm.trie", "ctc_test_lm.trie")):
    shutil.copyfile(os.path.join(REF, "ctc_decoder_with_lm", src), os.path.join(OUT, dst))
    os.chmod(os.path.join(OUT, dst), 0o644)
  shutil.copyfile(os.path.join(REF, "open_seq2seq", "test_utils", "toy_speech_data", "toy_data-lm.binary"),
                  os.path.join(OUT, "toy_data_lm.binary"))
  os.chmod(os.path.join(OUT, "toy_data_lm.binary"), 0o644)
  with open(os.path.join(OUT, "ctc_test_meta.json"), "w") as f:
    json.dump(meta, f, indent=1)
  toy_src = os.path.join(REF, "open_seq2seq", "test_utils", "toy_speech_data")
  toy_dst = os.path.join(os.path.dirname(os.path.dirname(OUT)), "open_seq2seq", "test_utils", "toy_speech_data")
  os.makedirs(os.path.join(toy_dst, "wav_files"), exist_ok=True)
  shutil.copyfile(os.path.join(toy_src, "toy_data.csv"), os.path.join(toy_dst, "toy_data.csv"))
  os.chmod(os.path.join(toy_dst, "toy_data.csv"), 0o644)
  for w in sorted(os.listdir(os.path.join(toy_src, "wav_files"))):
    shutil.copyfile(os.path.join(toy_src, "wav_files", w), os.path.join(toy_dst, "wav_files", w))
    os.chmod(os.path.join(toy_dst, "wav_files", w), 0o644)
  print("wrote", OUT)


if __name__ == "__main__":
  main()

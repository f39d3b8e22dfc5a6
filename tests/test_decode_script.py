"""scripts/decode.py (offline CTC decoding + LM weight tuning on dumped logits) on the
reference's golden utterance: the greedy transcript has one wrong word ('then seconds'), the
language model fixes it ('ten seconds') — ctc_decoder_with_lm/ctc-test.py:29-78. Host-only."""
import csv
import json
import os
import pickle
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")


def _inputs(tmp_path):
  with open(os.path.join(GOLD, "ctc_test_meta.json")) as f:
    meta = json.load(f)
  seq = np.load(os.path.join(GOLD, "ctc_test_logits.npy"))[:, 0, :]
  dump = {"logits": {"a.wav": seq, "b.wav": seq[:150]}, "step_size": 0.02,
          "vocab": {i: c for i, c in enumerate(meta["vocab"])}}
  p = str(tmp_path / "logits.pkl")
  with open(p, "wb") as f:
    pickle.dump(dump, f)
  labels = str(tmp_path / "labels.csv")
  with open(labels, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["wav_filename", "wav_filesize", "transcript"])
    w.writerow(["a.wav", "1", "ten seconds"])
    w.writerow(["b.wav", "1", "ten seconds"])
  vocab = str(tmp_path / "vocab.txt")
  with open(vocab, "w") as f:
    f.write("\n".join(meta["vocab"]) + "\n")
  return meta, p, labels, vocab


def _main():
  sys.path.insert(0, os.path.join(REPO, "scripts"))
  import decode
  return decode.main


def test_eval_grid_search_and_infer(tmp_path, capsys):
  meta, logits, labels, vocab = _inputs(tmp_path)
  main = _main()
  common = ["--logits", logits, "--labels", labels, "--vocab", vocab, "--decoder", "ctc_decoder_with_lm",
            "--lm", os.path.join(GOLD, "ctc_test_lm.binary"), "--trie", os.path.join(GOLD, "ctc_test_lm.trie"),
            "--beam_width", "16"]
  beams = str(tmp_path / "beams.txt")
  best = main(common + ["--mode", "eval", "--alpha", "0.0", "--alpha_max", "2.0", "--alpha_step", "1.0",
                        "--beta", "0.0", "--trie_weight", "0.0", "--dump_all_beams_to", beams])
  out = capsys.readouterr().out
  assert "Greedy WER = 0.5000" in out                    # 'then' != 'ten', twice
  assert "alpha=0.00, beta=0.00: WER=0.5000" in out      # all weights zero: the plain beam search
  assert "alpha=2.00, beta=0.00: WER=0.0000" in out
  assert "BEST: alpha=" in out and best["wer"] == 0.0 and best["alpha"] >= 1.0
  assert open(beams).read().count("B=>>>>>>>>") == 2
  res = str(tmp_path / "out.csv")
  preds = main(common + ["--mode", "infer", "--alpha", "2.0", "--beta", "0.5", "--infer_output_file", res])
  rows = list(csv.reader(open(res)))
  assert rows == [["wav_filename", "transcript"], ["a.wav", "ten seconds"], ["b.wav", preds[1]]]


def test_trie_built_from_arpa_unigrams(tmp_path, capsys):
  meta, logits, labels, vocab = _inputs(tmp_path)
  arpa = str(tmp_path / "lm.arpa")
  with open(arpa, "w") as f:
    f.write("\\data\\\nngram 1=5\nngram 2=3\n\n\\1-grams:\n-0.90309\t<unk>\n0\t<s>\t-0.30103\n"
            "-0.5351132\tten\t-0.30103\n-0.5351132\tseconds\t-0.30103\n-0.5351132\t</s>\n\n"
            "\\2-grams:\n-0.1898795\t<s> ten\n-0.1898795\tten seconds\n-0.1898795\tseconds </s>\n\n\\end\\\n")
  main = _main()
  best = main(["--logits", logits, "--labels", labels, "--vocab", vocab, "--lm", arpa, "--beam_width", "16",
               "--decoder", "ctc_decoder_with_lm", "--mode", "eval", "--alpha", "2.0", "--beta", "0.5"])
  assert best["wer"] == 0.0


def test_default_decoder_is_the_reference_scripts_decoder(tmp_path, capsys):
  """--decoder ctc_decoders (default): the module scripts/decode.py of the reference calls."""
  meta, logits, labels, vocab = _inputs(tmp_path)
  main = _main()
  common = ["--logits", logits, "--labels", labels, "--vocab", vocab, "--beam_width", "16",
            "--lm", os.path.join(GOLD, "ctc_test_lm.binary")]
  beams = str(tmp_path / "beams.txt")
  best = main(common + ["--mode", "eval", "--alpha", "2.0", "--beta", "0.5", "--dump_all_beams_to", beams])
  out = capsys.readouterr().out
  assert "Greedy WER = 0.5000" in out and "alpha=2.00, beta=0.50: WER=0.0000" in out
  assert abs(best["beams"][0][0][0] + 4.0845) < 1e-3         # scripts/ctc_decoders_test.py:79
  assert open(beams).read().count("E=>>>>>>>>") == 2
  res = str(tmp_path / "out.csv")
  main(common + ["--mode", "infer", "--alpha", "2.0", "--beta", "0.5", "--infer_output_file", res])
  assert list(csv.reader(open(res)))[1] == ["a.wav", "ten seconds"]

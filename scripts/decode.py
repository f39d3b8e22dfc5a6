"""CTC decoding of dumped logits with language-model rescoring, and tuning of the LM weights.

Same workflow and command line as the reference's scripts/decode.py (argument names and
meanings, :20-77; `eval` = grid search of alpha x beta reporting the word error rate, `infer` =
one decoding pass written as CSV): the input is the pickle `run.py --mode=infer` writes with
decoder_params['infer_logits_to_pickle'] = True (models/speech2text.py:327-346) and the csv of
file names (+ transcripts for `eval`).

Two decoders of the C-ABI library can do the rescoring (--decoder):
  ctc_decoders         (default) the decoder the reference script uses: the `ctc_decoders` module of
                       decoders/ (softmax probabilities, dictionary built from the LM vocabulary,
                       score = alpha * log10 P(word | history) + beta per word) — os2s_ctc_dict_*
  ctc_decoder_with_lm  the reference's in-graph decoder (raw logits, letter trie):
      --trie          the text trie of generate_trie; built on the fly from the unigrams of an
                      ARPA model if omitted
      --trie_weight   weight of the letter-prefix score
The language model is an ARPA file or a KenLM binary of the supported layout (include/os2s.h).
"""
from __future__ import absolute_import, division, print_function

import argparse
import csv
import os
import pickle
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
  sys.path.insert(0, REPO)


def parse(argv=None):
  ap = argparse.ArgumentParser(description="CTC decoding and tuning with LM rescoring")
  ap.add_argument("--mode", default="eval", help="either 'eval' (default) or 'infer'")
  ap.add_argument("--infer_output_file", help="output CSV file for 'infer' mode")
  ap.add_argument("--logits", required=True, help="pickle file with CTC logits")
  ap.add_argument("--labels", required=True,
                  help="CSV file with audio filenames (and ground truth transcriptions for 'eval' mode)")
  ap.add_argument("--lm", required=True, help="language model: ARPA text or KenLM binary")
  ap.add_argument("--vocab", required=True, help="vocab file with characters (alphabet)")
  ap.add_argument("--decoder", default="ctc_decoders", choices=["ctc_decoders", "ctc_decoder_with_lm"])
  ap.add_argument("--trie", help="letter trie (generate_trie format); default: built from the ARPA unigrams")
  ap.add_argument("--trie_weight", type=float, default=0.1)
  ap.add_argument("--alpha", type=float, required=True, help="value of LM weight")
  ap.add_argument("--alpha_max", type=float, help="maximum value of LM weight (grid search in 'eval' mode)")
  ap.add_argument("--alpha_step", type=float, default=0.1)
  ap.add_argument("--beta", type=float, required=True, help="value of word count weight")
  ap.add_argument("--beta_max", type=float, help="maximum value of word count weight (grid search)")
  ap.add_argument("--beta_step", type=float, default=0.1)
  ap.add_argument("--beam_width", type=int, default=128)
  ap.add_argument("--dump_all_beams_to", default="",
                  help="filename to dump all beams of the best setting in eval mode")
  return ap.parse_args(argv)


def load_labels(path):
  with open(path, newline="", encoding="utf-8") as f:
    rows = list(csv.reader(f))
  return rows[1:]


def load_alphabet(path):
  with open(path, encoding="utf-8") as f:
    return [line[0] for line in f.read().split("\n") if len(line) > 0]


def get_logits(data, labels):
  """The dump is either a raw array (one row of logits per label line) or the dict written by
  Speech2Text.finalize_inference (scripts/decode.py:119-134)."""
  if isinstance(data, np.ndarray):
    return {line[0]: data[i] for i, line in enumerate(labels)}
  return data["logits"] if "logits" in data else data[b"logits"]


def arpa_unigrams(path):
  words, on = [], False
  with open(path, encoding="utf-8") as f:
    for line in f:
      line = line.strip()
      if line == "\\1-grams:":
        on = True
        continue
      if on:
        if line.startswith("\\"):
          break
        parts = line.split()
        if len(parts) >= 2 and not (parts[1].startswith("<") and parts[1].endswith(">")):
          words.append(parts[1])
  return words


def build_trie(args, alphabet, workdir):
  from openseq2seq_amd import capi
  with open(args.lm, "rb") as f:
    if f.read(7) == b"mmap lm":
      raise SystemExit("--trie is required with a KenLM binary (its word list is not plain text)")
  letters = set(alphabet)
  words = [w for w in arpa_unigrams(args.lm) if all(c in letters for c in w)]
  vocab_path = os.path.join(workdir, "lm_words.txt")
  with open(vocab_path, "w", encoding="utf-8") as f:
    f.write("\n".join(words) + "\n")
  trie_path = os.path.join(workdir, "lm.trie")
  capi.ctc_generate_trie(args.vocab, args.lm, vocab_path, trie_path)
  return trie_path


def texts_from_ids(ids, lens, alphabet, top=0):
  return ["".join(alphabet[c] for c in ids[b, top, :int(lens[b, top])].tolist()) for b in range(ids.shape[0])]


def batch_logits(logits, names):
  """Pad the per-utterance [T, C] logits into one time-major [Tmax, B, C] batch."""
  import torch
  T = max(int(logits[n].shape[0]) for n in names)
  C = int(logits[names[0]].shape[-1])
  out = np.zeros((T, len(names), C), dtype=np.float32)
  lens = np.zeros(len(names), dtype=np.int32)
  for b, n in enumerate(names):
    l = np.asarray(logits[n], dtype=np.float32).reshape(-1, C)
    out[:l.shape[0], b] = l
    lens[b] = l.shape[0]
  return torch.from_numpy(out), torch.from_numpy(lens)


def word_error_rate(labels, preds):
  from openseq2seq_amd.models.speech2text import levenshtein
  dist = count = 0
  for line, pred in zip(labels, preds):
    ref = line[-1].lower().split()
    dist += levenshtein(ref, pred.lower().split())
    count += len(ref)
  return dist / max(count, 1)


def main(argv=None):
  args = parse(argv)
  from openseq2seq_amd import capi
  if args.alpha_max is None:
    args.alpha_max = args.alpha
  if args.beta_max is None:
    args.beta_max = args.beta
  with open(args.logits, "rb") as f:
    data = pickle.load(f, encoding="bytes")
  labels = load_labels(args.labels)
  names = [line[0] for line in labels]
  logits = get_logits(data, labels)
  alphabet = load_alphabet(args.vocab)
  batch, lens = batch_logits(logits, names)
  if batch.shape[2] != len(alphabet) + 1:
    raise SystemExit("logits have %d classes, the alphabet %d labels (+ blank)" % (batch.shape[2], len(alphabet)))
  def greedy_texts():
    out = []
    for b in range(len(names)):
      best = batch[:int(lens[b]), b].argmax(-1).tolist()
      s_, prev = [], -1
      for c in best:
        if c != prev and c != len(alphabet):
          s_.append(alphabet[c])
        prev = c
      out.append("".join(s_))
    return out

  with tempfile.TemporaryDirectory() as workdir:
    if args.decoder == "ctc_decoders":
      from openseq2seq_amd import ctc_decoders
      scorer = ctc_decoders.Scorer(args.alpha, args.beta, args.lm, alphabet)
      x = batch.numpy()
      x = np.exp(x - x.max(-1, keepdims=True))
      probs = x / x.sum(-1, keepdims=True)
      split = [probs[:int(lens[b]), b] for b in range(len(names))]

      def decode(alpha, beta, n_best):
        scorer.reset_params(alpha, beta)
        res = ctc_decoders.ctc_beam_search_decoder_batch(split, alphabet, args.beam_width, os.cpu_count() or 1,
                                                         ext_scoring_func=scorer)
        return [r[:n_best] for r in res]
    else:
      trie = args.trie or build_trie(args, alphabet, workdir)
      scorer = capi.CtcScorer(args.lm, trie, args.vocab, args.alpha, args.beta, args.trie_weight)

      def decode(alpha, beta, n_best):
        scorer.set_weights(alpha, beta, args.trie_weight)
        ids, ln, lp = capi.ctc_beam_search(batch, lens, args.beam_width, scorer, top_paths=n_best)
        return [[(float(lp[b, k]), texts_from_ids(ids, ln, alphabet, k)[b]) for k in range(n_best)]
                for b in range(len(names))]

    if args.mode == "eval":
      print("Greedy WER = {:.4f}".format(word_error_rate(labels, greedy_texts())))
      best = {"wer": 1e6, "alpha": 0.0, "beta": 0.0, "beams": None}
      n_best = min(args.beam_width, 8) if args.dump_all_beams_to else 1
      for alpha in np.arange(args.alpha, args.alpha_max + args.alpha_step / 10.0, args.alpha_step):
        for beta in np.arange(args.beta, args.beta_max + args.beta_step / 10.0, args.beta_step):
          res = decode(alpha, beta, n_best)
          wer = word_error_rate(labels, [r[0][1] for r in res])
          if wer < best["wer"]:
            best.update(wer=wer, alpha=alpha, beta=beta, beams=res)
          print("alpha={:.2f}, beta={:.2f}: WER={:.4f}".format(alpha, beta, wer))
      print("BEST: alpha={:.2f}, beta={:.2f}, WER={:.4f}".format(best["alpha"], best["beta"], best["wer"]))
      if args.dump_all_beams_to:
        with open(args.dump_all_beams_to, "w", encoding="utf-8") as f:
          for beam in best["beams"]:
            f.write("B=>>>>>>>>\n")
            for score, text in beam:
              f.write("{} 0.0 0.0 {}\n".format(score, text))
            f.write("E=>>>>>>>>\n")
      return best
    if args.mode == "infer":
      if not args.infer_output_file:
        raise SystemExit("--infer_output_file is required in 'infer' mode")
      preds = [r[0][1] for r in decode(args.alpha, args.beta, 1)]
      with open(args.infer_output_file, "w", newline="", encoding="utf-8") as f:
        w = csv.writer(f)
        w.writerow(["wav_filename", "transcript"])
        for n, p in zip(names, preds):
          w.writerow([n, p])
      return preds
    raise SystemExit("unknown mode %r" % args.mode)


if __name__ == "__main__":
  main()

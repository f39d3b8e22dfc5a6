"""Toy sequence-reversal corpus for the RNN NMT configs — the task that
scripts/create_toy_data.sh builds in the reference (test_utils/create_reversed_examples.py:
10 Greek letters, sentence lengths 5..50, target = reversed source; 10 000 / 1 000 / 2 000
train / dev / test lines; vocab files "token<TAB>count"). Same directory layout, so the
reference's toy-reversal configs point at it unchanged:
  <data_path>/{train,dev,test}/{source,target}.txt, <data_path>/vocab/{source,target}.txt"""
from __future__ import absolute_import, division, print_function

import io
import os
import shutil

import numpy as np

LETTERS = [chr(0x03B1 + i) for i in range(10)]   # alpha .. kappa


def _write_lines(path, rows):
  with io.open(path, "w", encoding="utf-8") as f:
    for row in rows:
      f.write(u" ".join(row) + u"\n")


def create_data(train_corpus_size=10000, dev_corpus_size=1000, test_corpus_size=2000,
                data_path="./toy_text_data", seed=None):
  rng = np.random.RandomState(seed)
  counts = {}
  for split, n in (("train", train_corpus_size), ("dev", dev_corpus_size), ("test", test_corpus_size)):
    d = os.path.join(data_path, split)
    os.makedirs(d, exist_ok=True)
    src = []
    for _ in range(n):
      ids = rng.randint(0, len(LETTERS), size=rng.randint(5, 51))
      for i in ids:
        counts[int(i)] = counts.get(int(i), 0) + 1
      src.append([LETTERS[i] for i in ids])
    _write_lines(os.path.join(d, "source.txt"), src)
    _write_lines(os.path.join(d, "target.txt"), [list(reversed(r)) for r in src])
  vd = os.path.join(data_path, "vocab")
  os.makedirs(vd, exist_ok=True)
  for name in ("source.txt", "target.txt"):   # source and target vocabularies are the same
    with io.open(os.path.join(vd, name), "w", encoding="utf-8") as f:
      for i in sorted(counts):
        f.write(u"%s\t%d\n" % (LETTERS[i], counts[i]))


def remove_data(data_path="./toy_text_data"):
  shutil.rmtree(data_path)


if __name__ == "__main__":
  create_data(data_path="toy_text_data")

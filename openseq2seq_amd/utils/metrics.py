"""Host-side evaluation metrics (callers of the hot path).

corpus_bleu: BLEU-4 with brevity penalty over token-id sequences, one reference per
hypothesis — what nltk.translate.bleu_score.corpus_bleu computes for the reference's
"Eval BLUE score" (models/text2text.py:214-222) with default uniform weights and no
smoothing. levenshtein / WER are in models/speech2text.py."""
from __future__ import division

import collections
import math


def _ngrams(seq, n):
  return collections.Counter(tuple(seq[i:i + n]) for i in range(len(seq) - n + 1))


def corpus_bleu(references, hypotheses, max_n=4):
  match = [0] * max_n
  total = [0] * max_n
  ref_len = hyp_len = 0
  for ref, hyp in zip(references, hypotheses):
    ref_len += len(ref)
    hyp_len += len(hyp)
    for n in range(1, max_n + 1):
      h, r = _ngrams(hyp, n), _ngrams(ref, n)
      total[n - 1] += max(len(hyp) - n + 1, 0)
      match[n - 1] += sum(min(c, r[g]) for g, c in h.items())
  if hyp_len == 0 or min(total) == 0 or min(match) == 0:
    return 0.0
  logp = sum(math.log(m / t) for m, t in zip(match, total)) / max_n
  bp = 1.0 if hyp_len > ref_len else math.exp(1.0 - ref_len / hyp_len)
  return bp * math.exp(logp)

"""Host-side metric helpers pinned to the reference's own known-answer tests:
levenshtein values (open_seq2seq/models/speech2text_test.py:229-256), the aggregate
'Eval WER' 37/40 and the 'Sample WER' 0.4 of its evaluate / finalize_evaluation /
maybe_print_logs test (:262-351), and corpus BLEU sanity (the reversal acceptance metric)."""
from openseq2seq_amd.models.speech2text import finalize_wer, levenshtein, sample_wer, wer_counts
from openseq2seq_amd.utils.metrics import corpus_bleu


def test_levenshtein_kats():
  cases = [('this is a great day', 'this is great day', 1),
           ('this is a great day', 'this great day', 2),
           ('this is a great day', 'this day is a great', 2),
           ('this is a great day', 'this day is great', 3),
           ('london is the capital of great britain', 'london capital gret britain', 4)]
  for a, b, d in cases:
    assert levenshtein(a.split(), b.split()) == d
    assert levenshtein(b.split(), a.split()) == d
  a, b = 'london is the capital of great britain', 'london capital gret britain'
  assert levenshtein(a, b) == 11 and levenshtein(b, a) == 11     # character level


INPUTS = [
    ['this is a great day', 'london is the capital of great britain'],
    ['ooo', 'lll'],
    ['a b c\' asdf', 'blah blah bblah'],
    ['this is great day', 'london capital gret britain'],
    ['aaaaaaaasdfdasdf', 'df d sdf asd fd f sdf df blah\' blah'],
]
OUTPUTS = [
    ['this is great a day', 'london capital gret britain'],
    ['ooo', 'lll'],
    ['aaaaaaaasdfdasdf', 'df d sdf asd fd f sdf df blah blah'],
    ['this is a great day', 'london is the capital of great britain'],
    ['a b c\' asdf', 'blah blah\' bblah'],
]


def test_eval_wer_kat():
  # the reference evaluates every batch twice before finalize_evaluation; the ratio is 37/40
  results = [wer_counts(i, o) for i, o in zip(INPUTS, OUTPUTS)] * 2
  assert finalize_wer(results)["Eval WER"] == 37 / 40.0
  assert sample_wer(INPUTS[0][0], OUTPUTS[0][0]) == 0.4


def test_corpus_bleu():
  ref = [[1, 2, 3, 4, 5, 6], [7, 8, 9, 10, 11]]
  assert abs(corpus_bleu(ref, ref) - 1.0) < 1e-12
  assert corpus_bleu(ref, [[1, 2, 3, 4, 5, 6], [11, 10, 9, 8, 7]]) < 0.7
  assert corpus_bleu(ref, [[], []]) == 0.0

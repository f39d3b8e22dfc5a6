"""CPU restatement of the reference's second CTC decoder, the `ctc_decoders` module under
decoders/ (prefix beam search over softmax probabilities with an external scorer and a
dictionary constraint; used offline by scripts/decode.py).  TEST INFRASTRUCTURE ONLY: nothing
outside tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Restated from
  * decoders/ctc_beam_search_decoder.cpp:18-178 — the time loop: vocabulary pruning
    (decoder_utils.cpp:7-37), the beam cut-off (:62-72,83-86), blank / repeated / new-character
    updates (:88-110), language-model scoring when a word ends (:112-131), the end-of-utterance
    scoring of an unfinished word (:148-160), result extraction (decoder_utils.cpp:40-95);
  * decoders/path_trie.cpp:37-158 — the prefix tree with the dictionary automaton: a character
    that does not continue a vocabulary word is refused; at the end of a word (+ space) the
    automaton is re-armed by the FIRST refused character and only the next attempt succeeds
    (:53-63 — kept, it shifts word starts by one frame);
  * decoders/scorer.cpp:71-200 — `get_log_cond_prob` (log10 probability of the last word of an
    n-gram scored from the null context, OOV = -1000), `make_ngram` (last `order` words of a
    prefix, padded with <s>), the dictionary = every LM vocabulary word spellable with the
    alphabet, followed by a space (:203-230 builds it as a minimised FST with OpenFST; any
    deterministic automaton of that language behaves identically, a character trie is used here).
The language model is oracle/ctc_beam_search.NGramLM (KenLM restated).

Parity pin (tests/test_oracle_ctc_decoders.py): scripts/ctc_decoders_test.py:73-80 — the golden
utterance with Scorer(alpha=2.0, beta=0.5, ctc-test-lm.binary), beam 16: 'ten seconds', score
-4.0845 +- 1e-3.
"""
from __future__ import absolute_import, division, print_function

import math

import numpy as np

FLT_MAX = float(np.finfo(np.float32).max)
FLT_MIN = float(np.finfo(np.float32).tiny)
NEG = -FLT_MAX
OOV_SCORE = -1000.0


def log_sum_exp(x, y):
  if x <= NEG:
    return y
  if y <= NEG:
    return x
  m = max(x, y)
  return math.log(math.exp(x - m) + math.exp(y - m)) + m


class Dictionary(object):
  """Automaton of {word + ' '}: children[state][label] -> state; final = after the space."""

  def __init__(self, words, alphabet):
    self.children = [{}]
    self.final = [False]
    lab = {c: i for i, c in enumerate(alphabet)}
    space = lab.get(" ")
    self.size = 0
    for w in words:
      if any(ch not in lab for ch in w):
        continue
      seq = [lab[ch] for ch in w] + ([space] if space is not None else [])
      s = 0
      for l in seq:
        nxt = self.children[s].get(l)
        if nxt is None:
          nxt = len(self.children)
          self.children.append({})
          self.final.append(False)
          self.children[s][l] = nxt
        s = nxt
      self.final[s] = True
      self.size += 1


class Scorer(object):
  """decoders/scorer.h: alpha, beta, the language model, the derived dictionary."""

  def __init__(self, alpha, beta, lm, alphabet):
    self.alpha, self.beta, self.lm, self.alphabet = alpha, beta, lm, list(alphabet)
    self.max_order = lm.order
    special = ("<unk>", "<s>", "</s>")
    self.is_character_based = all(len(w) <= 1 for w in lm.vocab if w not in special)
    self.space_id = self.alphabet.index(" ") if " " in self.alphabet else -1
    self.dictionary = None if self.is_character_based else Dictionary(lm.vocab, self.alphabet)

  def get_log_cond_prob(self, words):
    hist, p = [], 0.0
    for w in words:
      wid = self.lm.index(w)
      if wid == 0:
        return OOV_SCORE
      p = self.lm.score(hist, wid)
      hist.append(wid)
    return p

  def make_ngram(self, node):
    ngram, cur = [], node
    for order in range(self.max_order):
      if self.is_character_based:
        vec, new = cur.path_vec(self.space_id, 1)
        cur = new
      else:
        vec, new = cur.path_vec(self.space_id)
        cur = new.parent
      ngram.append("".join(self.alphabet[c] for c in vec))
      if new.character == -1:
        ngram.extend(["<s>"] * (self.max_order - order - 1))
        break
    return ngram[::-1]


class PathTrie(object):
  __slots__ = ("b_prev", "nb_prev", "b_cur", "nb_cur", "score", "character", "parent", "exists",
               "children", "dict", "dict_state")

  def __init__(self):
    self.b_prev = self.nb_prev = self.b_cur = self.nb_cur = self.score = NEG
    self.character, self.parent, self.exists, self.children = -1, None, True, []
    self.dict, self.dict_state = None, 0

  def get_path_trie(self, c):
    for ch, node in self.children:
      if ch == c:
        if not node.exists:
          node.exists = True
          node.b_prev = node.nb_prev = node.b_cur = node.nb_cur = NEG
        return node
    state = 0
    if self.dict is not None:
      nxt = self.dict.children[self.dict_state].get(c)
      if nxt is None:
        if self.dict.final[self.dict_state]:
          self.dict_state = 0            # re-armed; THIS attempt is still refused
        return None
      state = nxt
    node = PathTrie()
    node.character, node.parent, node.dict, node.dict_state = c, self, self.dict, state
    self.children.append((c, node))
    return node

  def path_vec(self, stop=-1, max_steps=None):
    out, n = [], self
    while not (n.character == stop or n.character == -1 or (max_steps is not None and len(out) == max_steps)):
      out.append(n.character)
      n = n.parent
    return out[::-1], n

  def iterate_to_vec(self, out):
    if self.exists:
      self.b_prev, self.nb_prev = self.b_cur, self.nb_cur
      self.b_cur = self.nb_cur = NEG
      self.score = log_sum_exp(self.b_prev, self.nb_prev)
      out.append(self)
    for _, ch in list(self.children):
      ch.iterate_to_vec(out)

  def remove(self):
    self.exists = False
    if not self.children:
      p = self.parent
      p.children = [(c, n) for c, n in p.children if n is not self]
      if not p.children and not p.exists:
        p.remove()


def _sort_key(n):
  return (-n.score, n.character)


def pruned_log_probs(prob, cutoff_prob, cutoff_top_n):
  idx = list(range(len(prob)))
  cutoff_len = len(prob)
  if cutoff_prob < 1.0 or cutoff_top_n < cutoff_len:
    idx.sort(key=lambda i: -prob[i])
    if cutoff_prob < 1.0:
      cum, cutoff_len = 0.0, 0
      for i in idx:
        cum += prob[i]
        cutoff_len += 1
        if cum >= cutoff_prob or cutoff_len >= cutoff_top_n:
          break
    idx = idx[:cutoff_len]
  return [(i, math.log(prob[i] + FLT_MIN)) for i in idx]


def ctc_beam_search_decoder(probs_seq, alphabet, beam_size, cutoff_prob=1.0, cutoff_top_n=40,
                            ext_scorer=None):
  """probs_seq [T, V+1] softmax probabilities (blank last). Returns [(score, text)], best first."""
  probs_seq = np.asarray(probs_seq, dtype=np.float64)
  V = len(alphabet)
  assert probs_seq.shape[1] == V + 1
  blank = V
  space = alphabet.index(" ") if " " in alphabet else -2
  root = PathTrie()
  root.score = root.b_prev = 0.0
  if ext_scorer is not None and not ext_scorer.is_character_based:
    root.dict = ext_scorer.dictionary
  prefixes = [root]
  for t in range(probs_seq.shape[0]):
    prob = probs_seq[t]
    min_cutoff, full_beam = NEG, False
    if ext_scorer is not None:
      n = min(len(prefixes), beam_size)
      prefixes[:n] = sorted(prefixes[:n], key=_sort_key)
      min_cutoff = prefixes[n - 1].score + math.log(prob[blank]) - max(0.0, ext_scorer.beta)
      full_beam = n == beam_size
    for c, lp in pruned_log_probs(prob, cutoff_prob, cutoff_top_n):
      for prefix in prefixes[:beam_size]:
        if full_beam and lp + prefix.score < min_cutoff:
          break
        if c == blank:
          prefix.b_cur = log_sum_exp(prefix.b_cur, lp + prefix.score)
          continue
        if c == prefix.character:
          prefix.nb_cur = log_sum_exp(prefix.nb_cur, lp + prefix.nb_prev)
        new = prefix.get_path_trie(c)
        if new is None:
          continue
        log_p = NEG
        if c == prefix.character and prefix.b_prev > NEG:
          log_p = lp + prefix.b_prev
        elif c != prefix.character:
          log_p = lp + prefix.score
        if ext_scorer is not None and (c == space or ext_scorer.is_character_based):
          to_score = new if ext_scorer.is_character_based else prefix
          log_p += ext_scorer.get_log_cond_prob(ext_scorer.make_ngram(to_score)) * ext_scorer.alpha
          log_p += ext_scorer.beta
        new.nb_cur = log_sum_exp(new.nb_cur, log_p)
    prefixes = []
    root.iterate_to_vec(prefixes)
    if len(prefixes) >= beam_size:
      prefixes.sort(key=_sort_key)
      for n in prefixes[beam_size:]:
        n.remove()
      prefixes = prefixes[:beam_size]
  if ext_scorer is not None and not ext_scorer.is_character_based:
    for prefix in prefixes[:beam_size]:
      if prefix.character != -1 and prefix.character != space:
        prefix.score += ext_scorer.get_log_cond_prob(ext_scorer.make_ngram(prefix)) * ext_scorer.alpha \
            + ext_scorer.beta
  best = sorted(prefixes[:beam_size], key=_sort_key)
  return [(p.score, "".join(alphabet[c] for c in p.path_vec()[0])) for p in best]

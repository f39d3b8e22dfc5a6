"""CPU restatement of the reference's CTC beam-search decoders.  TEST INFRASTRUCTURE ONLY:
nothing outside tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.

What is restated
----------------
* `ctc_beam_search(...)` — the prefix beam search of `CTCBeamSearchNormLogDecoder`
  (ctc_decoder_with_lm/beam_search.cc:245-447; the same algorithm as TensorFlow >= 1.11's
  `tf.nn.ctc_beam_search_decoder`, which is TF-internal and not in /root/reference): one
  `Step` per frame on log-softmax-normalised logits (:262-271), beams kept in a bounded
  top-N container ordered by `newp.total` (:279-313), children grown from every beam that
  would still fit (:321-378), `TopPaths` (:398-429) after the end-of-sequence rescoring of the
  op's Compute (:730-739).
* `BaseScorer` — `tensorflow::ctc::BaseBeamScorer`: every expansion score is the incoming
  score, end score 0.
* `WordLMScorer` — `WordLMBeamScorer` (ctc_decoder_with_lm/beam_search.h:32-217): a word
  n-gram language model is consulted each time a beam emits a space (and once more at the end
  for an unfinished word), `alpha * log10 P(word | history) + beta` is added to that
  expansion; inside a word `trie_weight * (min unigram score of the words below this letter
  prefix)` from the letter trie, or -100 when the prefix is no vocabulary word (:66-80).
  `ScoreNGram` (:172-200) pads a history shorter than the model order with `order - len`
  begin-of-sentence tokens and scores the last `order` words from the null context.
* `LetterTrie` — the text trie file (ctc_decoder_with_lm/trie_node.h:46-82,165-189).
* `NGramLM` — back-off n-gram scoring as KenLM's `BaseScore` does it (third-party dependency
  kenlm, fetched by the reference's scripts/install_kenlm.sh, absent from /root/reference;
  the published algorithm: longest matching n-gram's log10 probability plus the back-off
  weights of the longer contexts that did not match), loadable from an ARPA text file or from
  a KenLM binary of the one layout the reference's op accepts and ships a sample of
  (`QuantArrayTrieModel`, beam_search.h:22; ctc-test-lm.binary is a bigram model): see
  `read_kenlm_quant_array_trie`.

Parity pins (tests/test_oracle_ctc_beam_search.py): ctc_decoder_with_lm/ctc-test.py:60-78 —
'then seconds' / -1.1842575 without a scorer, 'ten seconds' / -4.619581 with
alpha=2.0, beta=0.5, trie_weight=0.1 and the sample model — and the alpha=beta=trie_weight=0
equivalence of :81-124.
"""
from __future__ import absolute_import, division, print_function

import math
import struct

import numpy as np

LOG_ZERO = -float("inf")


def log_sum_exp(a, b):
  if a == LOG_ZERO:
    return b
  if b == LOG_ZERO:
    return a
  return a + math.log1p(math.exp(b - a)) if a > b else b + math.log1p(math.exp(a - b))


# ------------------------------------------------------------------------------------------
# scorers
# ------------------------------------------------------------------------------------------
class BaseScorer(object):
  def initialize_state(self):
    return None

  def expand_state(self, from_state, from_label, to_label):
    return None

  def expand_state_end(self, state):
    return state

  def expansion_score(self, state, previous):
    return previous

  def end_expansion_score(self, state):
    return 0.0


class _LMState(object):
  __slots__ = ("lm_score", "score", "word", "node", "prefix", "new_word")

  def __init__(self):
    self.lm_score, self.score, self.word, self.node, self.prefix, self.new_word = \
        0.0, 0.0, "", None, (), False


class WordLMScorer(BaseScorer):
  """beam_search.h:32-217.  `alphabet`: list of label strings (no blank)."""

  def __init__(self, lm, trie_root, alphabet, alpha, beta, trie_weight):
    self.lm, self.root, self.alphabet = lm, trie_root, alphabet
    self.alpha, self.beta, self.trie_weight = alpha, beta, trie_weight

  def initialize_state(self):
    s = _LMState()
    s.node = self.root
    return s

  def _copy(self, f):
    s = _LMState()
    s.word, s.node, s.prefix = f.word, f.node, f.prefix
    return s

  def expand_state(self, f, from_label, to_label):
    s = self._copy(f)
    if self.alphabet[to_label] != " ":
      s.word = f.word + self.alphabet[to_label]
      score = -100.0
      node = f.node
      if node is not None:
        node = node.children.get(to_label)
        s.node = node
        if node is not None:
          score = node.min_unigram_score
      s.score = score
    else:
      if from_label == to_label:
        return s
      s.prefix = f.prefix + (s.word,)
      s.word = ""
      s.node = self.root
      s.new_word = True
      s.lm_score = self.score_ngram(s.prefix)
    return s

  def expand_state_end(self, s):
    if len(s.word) > 0:
      s.prefix = s.prefix + (s.word,)
      s.word = ""
      s.node = self.root
      s.new_word = True
      s.lm_score = self.score_ngram(s.prefix)
    return s

  def expansion_score(self, s, previous):
    if s.new_word:
      return previous + (self.alpha * s.lm_score + self.beta)
    return previous + self.trie_weight * s.score

  def end_expansion_score(self, s):
    return (self.alpha * s.lm_score + self.beta) if s.new_word else 0.0

  def score_ngram(self, prefix):
    order = self.lm.order
    hist = []
    if len(prefix) < order:
      hist = [self.lm.bos] * (order - len(prefix))
      words = prefix
    else:
      words = prefix[len(prefix) - order:]
    prob = 0.0
    for w in words:
      wid = self.lm.index(w)
      if wid == 0:
        return -100.0
      prob = self.lm.score(hist, wid)
      hist.append(wid)
    if not words:      # only begin-of-sentence tokens were scored
      prob = self.lm.score(hist[:-1], self.lm.bos)
    return prob


# ------------------------------------------------------------------------------------------
# letter trie (trie_node.h)
# ------------------------------------------------------------------------------------------
class TrieNode(object):
  __slots__ = ("prefix_count", "min_score_word", "min_unigram_score", "children")

  def __init__(self):
    self.prefix_count, self.min_score_word, self.min_unigram_score = 0, 0, float("inf")
    self.children = {}


TRIE_MAGIC = 0x54524945


def read_letter_trie(path, vocab_size):
  with open(path) as f:
    tok = f.read().split()
  if int(tok[0]) != TRIE_MAGIC or int(tok[1]) != 1 or int(tok[2]) != vocab_size:
    raise ValueError("bad trie header")
  pos = [3]

  def read():
    c = int(tok[pos[0]])
    pos[0] += 1
    if c == -1:
      return None
    n = TrieNode()
    n.prefix_count = c
    n.min_score_word = int(tok[pos[0]])
    n.min_unigram_score = float(tok[pos[0] + 1])
    pos[0] += 2
    for i in range(vocab_size):
      ch = read()
      if ch is not None:
        n.children[i] = ch
    return n

  import sys
  old = sys.getrecursionlimit()
  sys.setrecursionlimit(10000)
  try:
    return read()
  finally:
    sys.setrecursionlimit(old)


def build_letter_trie(words, alphabet, unigram_score):
  """generate_trie.cpp:12-60: one Insert per vocabulary word with its unigram score
  (log10 P(word) from the null context)."""
  to_label = {c: i for i, c in enumerate(alphabet)}
  root = TrieNode()
  for wid, w in words:
    sc = unigram_score(wid)
    node = root
    for i in range(len(w) + 1):
      node.prefix_count += 1
      if sc < node.min_unigram_score:
        node.min_unigram_score, node.min_score_word = sc, wid
      if i < len(w):
        node = node.children.setdefault(to_label[w[i]], TrieNode())
  return root


# ------------------------------------------------------------------------------------------
# n-gram language model
# ------------------------------------------------------------------------------------------
class NGramLM(object):
  """ngrams[n-1]: {tuple(word ids): (log10 prob, log10 backoff)}.  Word id 0 is <unk>."""

  def __init__(self, order, vocab, ngrams, hashed=None):
    self.order, self.vocab, self.ngrams = order, vocab, ngrams
    # kenlm "probing" models: orders >= 2 are only known by the 64-bit hash of their word ids
    # (hashed[n-1]: {key: (prob, backoff)}); ngrams[0] still holds the unigrams
    self.hashed = hashed
    self.words = {w: i for i, w in enumerate(vocab)}
    self.bos = self.words["<s>"]

  def index(self, w):
    return self.words.get(w, 0)

  def _lookup(self, key):
    """(prob, backoff) of the n-gram `key` (tuple of word ids) or None."""
    if self.hashed is None:
      return self.ngrams[len(key) - 1].get(key)
    if len(key) == 1:
      return self.ngrams[0].get(key)
    node = key[-1]                       # kenlm keys an n-gram by its LAST word first ...
    for w in reversed(key[:-1]):         # ... then the preceding words, most recent first
      node = combine_word_hash(node, w)
    return self.hashed[len(key) - 1].get(node)

  def score(self, hist, wid):
    """log10 P(wid | hist) by back-off over the last order-1 words of hist."""
    if self.hashed is not None:
      ctx = tuple(hist[-(self.order - 1):]) if self.order > 1 else ()
      for start in range(len(ctx) + 1):
        e = self._lookup(ctx[start:] + (wid,))
        if e is not None:
          p = e[0]
          for s in range(start):
            b = self._lookup(ctx[s:])
            if b is not None:
              p += b[1]
          return p
      raise KeyError("unigram %d missing" % wid)
    ctx = tuple(hist[-(self.order - 1):]) if self.order > 1 else ()
    # longest context that exists as an n-gram prefix is irrelevant: walk from the longest
    for start in range(len(ctx) + 1):
      key = ctx[start:] + (wid,)
      e = self.ngrams[len(key) - 1].get(key)
      if e is not None:
        p = e[0]
        # back-offs of the longer contexts that failed
        for s in range(start):
          c = ctx[s:]
          b = self.ngrams[len(c) - 1].get(c)
          if b is not None:
            p += b[1]
        return p
    raise KeyError("unigram %d missing" % wid)


def read_arpa(path):
  with open(path) as f:
    lines = [l.rstrip("\n") for l in f]
  i = 0
  while lines[i].strip() != "\\data\\":
    i += 1
  counts = {}
  i += 1
  while lines[i].startswith("ngram "):
    n, c = lines[i][6:].split("=")
    counts[int(n)] = int(c)
    i += 1
  order = max(counts)
  raw = [dict() for _ in range(order)]
  vocab = ["<unk>"]
  words = {"<unk>": 0}
  for n in range(1, order + 1):
    while lines[i].strip() != "\\%d-grams:" % n:
      i += 1
    i += 1
    while i < len(lines) and lines[i].strip() and not lines[i].startswith("\\"):
      parts = lines[i].split()
      prob = float(parts[0])
      toks = parts[1:1 + n]
      bo = float(parts[1 + n]) if len(parts) > 1 + n else 0.0
      if n == 1:
        w = toks[0]
        if w not in words:
          words[w] = len(vocab)
          vocab.append(w)
      raw[n - 1][tuple(words[t] for t in toks)] = (prob, bo)
      i += 1
  return NGramLM(order, vocab, raw)


def murmur_hash64a(data, seed=0):
  """util/murmur_hash.cc (kenlm): MurmurHash64A, the vocabulary hash of the binary format."""
  m = 0xc6a4a7935bd1e995
  mask = (1 << 64) - 1
  h = (seed ^ (len(data) * m)) & mask
  n8 = len(data) // 8
  for i in range(n8):
    k, = struct.unpack_from("<Q", data, 8 * i)
    k = (k * m) & mask
    k ^= k >> 47
    k = (k * m) & mask
    h ^= k
    h = (h * m) & mask
  tail = data[8 * n8:]
  if tail:
    for j in range(len(tail) - 1, -1, -1):
      h ^= tail[j] << (8 * j)
    h = (h * m) & mask
  h ^= h >> 47
  h = (h * m) & mask
  h ^= h >> 47
  return h


def _read_bits(buf, bit_off, nbits):
  v = 0
  byte = bit_off >> 3
  chunk = int.from_bytes(buf[byte:byte + 16], "little")
  v = (chunk >> (bit_off & 7)) & ((1 << nbits) - 1)
  return v


def _required_bits(x):
  return 0 if x == 0 else int(x).bit_length()


KENLM_MAGIC = b"mmap lm http://kheafield.com/code format version 5\n\0"


def read_kenlm_quant_array_trie(path):
  """KenLM binary, model type 5 (QUANT_ARRAY_TRIE) — what `lm::ngram::QuantArrayTrieModel`
  loads.  Layout (kenlm lm/binary_format.cc, lm/vocab.cc, lm/quantize.cc, lm/trie.cc;
  restated, validated on the reference's ctc-test-lm.binary):
    0x00  magic string (52 bytes padded to 56), sanity block (floats 0/1/-0.5, word sizes) to 0x58
    0x58  u8 order, f32 probing multiplier, u32 model type, u8 has_vocabulary, u32 search version
          u64 counts[order]; padded to 8
    vocabulary: u64 n (words without <unk>), n sorted MurmurHash64A values, one u64 slot spare;
          word id = 1 + rank of its hash, id 0 = <unk>
    quantisation: u8 version(2), u8 prob bits, u8 backoff bits, pad to 8; per middle order
          2^prob + 2^backoff float bins; for the highest order 2^prob float bins
    unigrams: (count + 2) x {f32 prob, f32 backoff, u64 next}
    middle orders (bit-packed, Bhiksha-compressed pointers) — only order-2 models (no middle
          layers) are accepted here: the reference ships no sample of a higher-order file to
          validate the pointer compression against, so those fail loudly;
    highest order: bit-packed {word id (bits of count[0]), quantised prob}, entries of
          predicted word w at [unigram[w].next, unigram[w+1].next) — the trie is keyed by the
          predicted word first, then by the preceding words.
  The word strings follow the search structures, in ARPA order, NUL separated."""
  with open(path, "rb") as f:
    d = f.read()
  if d[:len(KENLM_MAGIC)] != KENLM_MAGIC:
    raise ValueError("not a KenLM binary (format version 5)")
  off = 0x58
  order = d[off]
  model_type, = struct.unpack_from("<I", d, off + 8)
  has_vocab = d[off + 12]
  counts = struct.unpack_from("<%dQ" % order, d, off + 20)
  if model_type != 5:
    raise NotImplementedError("KenLM model type %d: the reference's op loads type 5 only" % model_type)
  if order != 2:
    raise NotImplementedError("KenLM quantised array tries of order > 2 are not validated; "
                              "use the ARPA file")
  if not has_vocab:
    raise ValueError("binary without the vocabulary strings")
  off = (off + 20 + 8 * order + 7) & ~7
  nvoc, = struct.unpack_from("<Q", d, off)
  hashes = struct.unpack_from("<%dQ" % nvoc, d, off + 8)
  off += 8 * (counts[0] + 1)
  ver, pbits, bbits = d[off], d[off + 1], d[off + 2]
  if ver != 2:
    raise ValueError("quantisation version %d" % ver)
  off += 8
  bins = np.frombuffer(d, dtype="<f4", count=1 << pbits, offset=off)
  off += 4 << pbits
  uni = []
  for i in range(counts[0] + 1):
    p, b, nxt = struct.unpack_from("<ffQ", d, off + 16 * i)
    uni.append((p, b, nxt))
  off += 16 * (counts[0] + 2)
  wbits = _required_bits(counts[0])
  tot = wbits + pbits
  long_base = off
  off += ((1 + counts[1]) * tot + 7) // 8 + 8
  strings = d[off:].split(b"\0")
  strings = [s.decode("utf-8") for s in strings[:counts[0]]]
  # word ids: <unk> = 0, the rest ranked by hash
  rank = {h: i + 1 for i, h in enumerate(hashes)}
  vocab = [None] * counts[0]
  for s in strings:
    if s == "<unk>":
      vocab[0] = s
    else:
      vocab[rank[murmur_hash64a(s.encode("utf-8"))]] = s
  if any(v is None for v in vocab):
    raise ValueError("vocabulary hashes do not match the strings")
  uni_d, bi_d = {}, {}
  for w in range(counts[0]):
    p, b, _ = uni[w]
    # the sign bit of an in-trie prob marks left-extension independence: value is -|p|
    uni_d[(w,)] = (-abs(p), b)
    for e in range(uni[w][2], uni[w + 1][2]):
      ctx = _read_bits(d, long_base * 8 + e * tot, wbits)
      q = _read_bits(d, long_base * 8 + e * tot + wbits, pbits)
      bi_d[(ctx, w)] = (float(bins[q]), 0.0)
  return NGramLM(2, vocab, [uni_d, bi_d])


def combine_word_hash(current, nxt):
  """lm/search_hashed.hh (kenlm): the chained key of the probing tables."""
  mask = (1 << 64) - 1
  return ((int(current) * 8978948897894561157) & mask) ^ (((1 + int(nxt)) * 17894857484156487943) & mask)


def read_kenlm_probing(path):
  """KenLM binary, model type 0 (PROBING, kenlm's default) — the `ctc_decoders` scorer loads any
  type through lm::ngram::LoadVirtual (decoders/scorer.cpp:56-63); the reference ships a trigram
  sample, open_seq2seq/test_utils/toy_speech_data/toy_data-lm.binary. Layout (restated; every
  piece is validated on that file: all 115 + 108 stored keys are reproduced by the hash chain,
  the vocabulary hashes match the strings, and every context sums to probability 1):
    header as for the trie models; then the vocabulary: {u32 version, u32 bound} + B(count[0])
    packed 12-byte buckets {u64 MurmurHash64A(word), u32 id}, B(n) = max(n + 1, int(multiplier *
    n)), empty = key 0, ids in ARPA order (= the order of the strings at the end of the file);
    unigrams: (count[0] + 1) x {f32 prob, f32 backoff};
    per middle order n: B(count[n-1]) x {u64 key, f32 prob, f32 backoff};
    highest order: B(count[-1]) x packed {u64 key, f32 prob};
    key of (w1 .. wn) = combine(...combine(combine(wn, wn-1), wn-2)..., w1).
  The sign bit of a stored probability is a flag: value = -|x|."""
  with open(path, "rb") as f:
    d = f.read()
  if d[:len(KENLM_MAGIC)] != KENLM_MAGIC:
    raise ValueError("not a KenLM binary (format version 5)")
  off = 0x58
  order = d[off]
  mult, = struct.unpack_from("<f", d, off + 4)
  model_type, = struct.unpack_from("<I", d, off + 8)
  if model_type != 0 or not d[off + 12]:
    raise NotImplementedError("not a probing model with vocabulary")
  counts = struct.unpack_from("<%dQ" % order, d, off + 20)
  off = (off + 20 + 8 * order + 7) & ~7

  def buckets(n):
    return max(n + 1, int(np.float32(mult) * np.float32(n)))

  off += 8 + 12 * buckets(counts[0])
  uni = {}
  for w in range(counts[0]):
    p, b = struct.unpack_from("<ff", d, off + 8 * w)
    uni[(w,)] = (-abs(p), b)
  off += 8 * (counts[0] + 1)
  hashed = [None]
  for n in range(2, order + 1):
    tab, nb = {}, buckets(counts[n - 1])
    if n < order:
      for i in range(nb):
        k, p, b = struct.unpack_from("<Qff", d, off + 16 * i)
        if k:
          tab[k] = (-abs(p), b)
      off += 16 * nb
    else:
      for i in range(nb):
        k, p = struct.unpack_from("<Qf", d, off + 12 * i)
        if k:
          tab[k] = (-abs(p), 0.0)
      off += 12 * nb
    if len(tab) != counts[n - 1]:
      raise ValueError("order %d: %d keys, header says %d" % (n, len(tab), counts[n - 1]))
    hashed.append(tab)
  vocab = [x.decode("utf-8") for x in d[off:].split(b"\0")[:counts[0]]]
  if len(vocab) != counts[0] or vocab[0] != "<unk>":
    raise ValueError("vocabulary strings")
  return NGramLM(order, vocab, [uni] + [dict() for _ in range(order - 1)], hashed=hashed)


def load_lm(path):
  with open(path, "rb") as f:
    head = f.read(len(KENLM_MAGIC))
  if head != KENLM_MAGIC:
    return read_arpa(path)
  with open(path, "rb") as f:
    f.seek(0x60)
    model_type, = struct.unpack("<I", f.read(4))
  return read_kenlm_probing(path) if model_type == 0 else read_kenlm_quant_array_trie(path)


# ------------------------------------------------------------------------------------------
# the prefix beam search
# ------------------------------------------------------------------------------------------
class _Prob(object):
  __slots__ = ("total", "blank", "label")

  def __init__(self):
    self.reset()

  def reset(self):
    self.total = self.blank = self.label = LOG_ZERO

  def assign(self, o):
    self.total, self.blank, self.label = o.total, o.blank, o.label


class _Entry(object):
  __slots__ = ("parent", "label", "children", "oldp", "newp", "state", "serial")
  _count = [0]

  def __init__(self, parent, label):
    self.parent, self.label, self.children = parent, label, {}
    self.oldp, self.newp, self.state = _Prob(), _Prob(), None
    _Entry._count[0] += 1
    self.serial = _Entry._count[0]

  def active(self):
    return self.newp.total != LOG_ZERO

  def label_seq(self, merge_repeated):
    out, prev, c = [], -1, self
    while c.parent is not None:
      if not merge_repeated or c.label != prev:
        out.append(c.label)
      prev = c.label
      c = c.parent
    return out[::-1]


class _TopN(object):
  """gtl::TopN with a greater-than comparer on newp.total: keeps the `limit` best."""

  def __init__(self, limit):
    self.limit, self.items = limit, []

  def size(self):
    return len(self.items)

  def push(self, e):
    if len(self.items) < self.limit:
      self.items.append(e)
      return
    bi = min(range(len(self.items)), key=lambda i: self.items[i].newp.total)
    if e.newp.total > self.items[bi].newp.total:
      self.items[bi] = e

  def peek_bottom(self):
    return min(self.items, key=lambda x: x.newp.total)

  def extract(self):
    out = sorted(self.items, key=lambda x: -x.newp.total)
    self.items = []
    return out


def ctc_beam_search(logits, beam_width, scorer=None, top_paths=1, merge_repeated=False):
  """logits [T, C] float (blank = C-1).  Returns ([label lists], [log probs])."""
  scorer = scorer or BaseScorer()
  logits = np.asarray(logits, dtype=np.float64)
  T, C = logits.shape
  blank = C - 1
  leaves = _TopN(beam_width)
  root = _Entry(None, -1)
  root.newp.total = 0.0
  root.newp.blank = 0.0
  root.state = scorer.initialize_state()
  leaves.push(root)

  def is_candidate(p):
    return p.total > LOG_ZERO and (leaves.size() < beam_width or
                                   p.total > leaves.peek_bottom().newp.total)

  for t in range(T):
    raw = logits[t]
    mx = raw.max()
    norm = mx + math.log(np.exp(raw - mx).sum())
    branches = leaves.extract()
    for b in branches:
      b.oldp.assign(b.newp)
    for b in branches:
      if b.parent is not None:
        if b.parent.active():
          prev = b.parent.oldp.blank if b.label == b.parent.label else b.parent.oldp.total
          b.newp.label = log_sum_exp(b.newp.label, scorer.expansion_score(b.state, prev))
        b.newp.label += raw[b.label] - norm
      b.newp.blank = b.oldp.total + raw[blank] - norm
      b.newp.total = log_sum_exp(b.newp.blank, b.newp.label)
      leaves.push(b)
    for b in branches:
      if not is_candidate(b.oldp):
        continue
      for label in range(C - 1):
        c = b.children.get(label)
        if c is None:
          c = b.children[label] = _Entry(b, label)
        if not c.active():
          c.newp.blank = LOG_ZERO
          c.state = scorer.expand_state(b.state, b.label, label)
          prev = b.oldp.blank if label == b.label else b.oldp.total
          c.newp.label = raw[label] - norm + scorer.expansion_score(c.state, prev)
          c.newp.total = c.newp.label
          if is_candidate(c.newp):
            if leaves.size() == beam_width:
              leaves.peek_bottom().newp.reset()
            leaves.push(c)
          else:
            c.oldp.reset()
            c.newp.reset()
  # end-of-sequence rescoring (beam_search.cc:730-739), then TopPaths
  final = leaves.extract()
  for e in final:
    e.state = scorer.expand_state_end(e.state)
    e.newp.total += scorer.end_expansion_score(e.state)
  final.sort(key=lambda x: -x.newp.total)
  if top_paths > len(final):
    raise ValueError("Less leaves in the beam search than requested.")
  return ([e.label_seq(merge_repeated) for e in final[:top_paths]],
          [e.newp.total for e in final[:top_paths]])

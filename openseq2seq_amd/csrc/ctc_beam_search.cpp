// CTC prefix beam search with a word n-gram language model — HOST code (no kernel): the wire
// step after the logits leave the GPU.
//
// Reference: the custom TensorFlow op CTCBeamSearchDecoderWithLM, registered for DEVICE_CPU only
// (ctc_decoder_with_lm/beam_search.cc:452-804), i.e. CTCBeamSearchNormLogDecoder::Step / TopPaths
// (:245-447) driven by WordLMBeamScorer (ctc_decoder_with_lm/beam_search.h:32-217) over a KenLM
// model (third-party, not in the reference tree), a letter trie (trie_node.h) and an alphabet
// file (alphabet.h); bound from FullyConnectedCTCDecoder.decode_with_lm
// (open_seq2seq/decoders/fc_decoders.py:206-235).
//
// Built from scratch here:
//  * beams live in a per-utterance arena; a child entry is only materialised when it enters the
//    beam (the reference allocates every label of every expanded beam and keeps it until Reset),
//    the bounded best-N container is a binary min-heap on the total log probability;
//  * the scorer state holds word ids of the last `order` words instead of the growing list of
//    word strings; the language model is a forward trie in flat hash tables keyed by
//    (context node, word id) — exact keys, no hash-collision risk — loaded from an ARPA file or
//    from a KenLM binary of the layout the reference's op accepts and ships a sample of
//    (quantised array trie, order 2; higher-order binaries are refused, see os2s.h);
//  * utterances of a batch are decoded on a pool of host threads.
// Arithmetic is float like the reference's (log1pf / expf), ties in the heap are broken by
// entry creation order so that results are deterministic.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <fstream>
#include <limits>
#include <memory>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/os2s.h"

namespace {

constexpr float kLogZero = -std::numeric_limits<float>::infinity();
constexpr int kMaxLmOrder = 8;

inline float log_sum_exp(float a, float b) {
  if (a == kLogZero) return b;
  if (b == kLogZero) return a;
  return a > b ? a + log1pf(expf(b - a)) : b + log1pf(expf(a - b));
}

// ---------------------------------------------------------------------------------------------
// n-gram language model: forward trie, node id = index of the n-gram in its order's table
// ---------------------------------------------------------------------------------------------
struct NGramLM {
  struct Entry { float prob, backoff; };
  int order = 0;
  std::unordered_map<std::string, uint32_t> words;      // <unk> = 0
  uint32_t bos = 0;
  std::vector<std::vector<Entry>> entries;                // [order][node]
  std::vector<std::unordered_map<uint64_t, uint32_t>> index;   // order n>=2: (ctx node, word) -> node

  // kenlm "probing" binaries: orders >= 2 are only known by the chained 64-bit hash of their
  // word ids (hashed[n-1]); the forward trie above stays empty for them
  bool use_hash = false;
  std::vector<std::unordered_map<uint64_t, Entry>> hashed;
  static uint64_t combine(uint64_t current, uint32_t next) {       // lm/search_hashed.hh (kenlm)
    return (current * 8978948897894561157ULL) ^ ((uint64_t)(1 + next) * 17894857484156487943ULL);
  }
  // entry of the n-gram ids[0..n): keyed by its LAST word first, then the preceding ones
  const Entry* lookup_hashed(const uint32_t* ids, int n) const {
    if (n == 1) return ids[0] < entries[0].size() ? &entries[0][ids[0]] : nullptr;
    uint64_t node = ids[n - 1];
    for (int i = n - 2; i >= 0; --i) node = combine(node, ids[i]);
    auto it = hashed[n - 1].find(node);
    return it == hashed[n - 1].end() ? nullptr : &it->second;
  }

  uint32_t word_index(const std::string& w) const {
    auto it = words.find(w);
    return it == words.end() ? 0u : it->second;
  }
  // node of the n-gram ids[0..n) or -1
  int64_t find(const uint32_t* ids, int n) const {
    if (n == 0) return -1;
    int64_t node = ids[0];
    if ((size_t)node >= entries[0].size()) return -1;
    for (int i = 1; i < n; ++i) {
      auto it = index[i].find(((uint64_t)node << 32) | ids[i]);
      if (it == index[i].end()) return -1;
      node = it->second;
    }
    return node;
  }
  uint32_t add(const uint32_t* ids, int n, float prob, float backoff) {
    if (n == 1) {
      if (entries[0].size() <= ids[0]) entries[0].resize(ids[0] + 1, Entry{0.f, 0.f});
      entries[0][ids[0]] = Entry{prob, backoff};
      return ids[0];
    }
    const int64_t ctx = find(ids, n - 1);
    if (ctx < 0) return UINT32_MAX;      // context n-gram missing: malformed model
    const uint32_t node = (uint32_t)entries[n - 1].size();
    entries[n - 1].push_back(Entry{prob, backoff});
    index[n - 1][((uint64_t)ctx << 32) | ids[n - 1]] = node;
    return node;
  }
  // log10 P(w | hist[0..nh)) — back-off over the last order-1 words (kenlm BaseScore)
  float score(const uint32_t* hist, int nh, uint32_t w) const {
    const int nc = std::min(nh, order - 1);
    const uint32_t* ctx = hist + (nh - nc);
    uint32_t key[16];
    if (use_hash) {
      for (int start = 0; start <= nc; ++start) {
        const int n = nc - start;
        for (int i = 0; i < n; ++i) key[i] = ctx[start + i];
        key[n] = w;
        const Entry* e = lookup_hashed(key, n + 1);
        if (!e) continue;
        float p = e->prob;
        for (int s = 0; s < start; ++s) {
          const Entry* b = lookup_hashed(ctx + s, nc - s);
          if (b) p += b->backoff;
        }
        return p;
      }
      return -100.f;
    }
    for (int start = 0; start <= nc; ++start) {
      const int n = nc - start;
      for (int i = 0; i < n; ++i) key[i] = ctx[start + i];
      key[n] = w;
      const int64_t node = find(key, n + 1);
      if (node < 0) continue;
      float p = entries[n][node].prob;
      for (int s = 0; s < start; ++s) {
        const int64_t c = find(ctx + s, nc - s);
        if (c >= 0) p += entries[nc - s - 1][c].backoff;
      }
      return p;
    }
    return -100.f;
  }
};

bool read_arpa(const std::string& path, NGramLM* lm) {
  std::ifstream in(path);
  if (!in) return false;
  std::string line;
  while (std::getline(in, line) && line.compare(0, 6, "\\data\\") != 0) {}
  int order = 0;
  std::vector<size_t> declared(16, 0);
  while (std::getline(in, line) && line.compare(0, 6, "ngram ") == 0) {
    const int n = atoi(line.c_str() + 6);
    order = std::max(order, n);
    const size_t eq = line.find('=');
    if (n >= 1 && n <= 15 && eq != std::string::npos) declared[n] = strtoull(line.c_str() + eq + 1, nullptr, 10);
  }
  if (order < 1 || order > 15) return false;
  lm->order = order;
  lm->entries.assign(order, {});
  lm->index.assign(order, {});
  // sized once from the header (no rehashing on big models) — but an n-gram line is at least 4
  // bytes, so a declared count above file_size / 4 is a corrupt header, not a reason to allocate
  size_t file_bytes = 0;
  {
    std::ifstream sz(path, std::ios::binary | std::ios::ate);
    if (sz) file_bytes = (size_t)sz.tellg();
  }
  for (int n = 1; n <= order; ++n) {
    if (declared[n] > file_bytes / 4 + 1) return false;
    lm->entries[n - 1].reserve(declared[n] + 1);
    if (n > 1) lm->index[n - 1].reserve(declared[n]);
  }
  lm->words.clear();
  lm->words["<unk>"] = 0;
  lm->entries[0].push_back(NGramLM::Entry{-100.f, 0.f});
  int n = 0;
  std::vector<std::string> tok;
  do {
    if (line.empty()) continue;
    if (line[0] == '\\') {
      if (line.compare(0, 5, "\\end\\") == 0) break;
      n = atoi(line.c_str() + 1);
      if (n < 1 || n > order) return false;
      continue;
    }
    if (n == 0) continue;
    tok.clear();
    size_t p = 0;
    while (p < line.size()) {
      while (p < line.size() && (line[p] == ' ' || line[p] == '\t' || line[p] == '\r')) ++p;
      size_t q = p;
      while (q < line.size() && line[q] != ' ' && line[q] != '\t' && line[q] != '\r') ++q;
      if (q > p) tok.emplace_back(line, p, q - p);
      p = q;
    }
    if ((int)tok.size() < n + 1) return false;
    const float prob = strtof(tok[0].c_str(), nullptr);
    const float bo = (int)tok.size() > n + 1 ? strtof(tok[n + 1].c_str(), nullptr) : 0.f;
    uint32_t ids[16];
    for (int i = 0; i < n; ++i) {
      if (n == 1) {
        auto it = lm->words.find(tok[1]);
        if (it == lm->words.end()) it = lm->words.emplace(tok[1], (uint32_t)lm->words.size()).first;
        ids[0] = it->second;
      } else {
        ids[i] = lm->word_index(tok[1 + i]);
      }
    }
    if (lm->add(ids, n, prob, bo) == UINT32_MAX) return false;
  } while (std::getline(in, line));
  auto it = lm->words.find("<s>");
  if (it == lm->words.end()) return false;
  lm->bos = it->second;
  if (lm->entries[0].size() < lm->words.size()) lm->entries[0].resize(lm->words.size(), NGramLM::Entry{-100.f, 0.f});
  return true;
}

uint64_t murmur_hash64a(const void* key, size_t len, uint64_t seed) {
  const uint64_t m = 0xc6a4a7935bd1e995ULL;
  const int r = 47;
  uint64_t h = seed ^ (len * m);
  const unsigned char* data = (const unsigned char*)key;
  const size_t n8 = len / 8;
  for (size_t i = 0; i < n8; ++i) {
    uint64_t k;
    memcpy(&k, data + 8 * i, 8);
    k *= m; k ^= k >> r; k *= m;
    h ^= k; h *= m;
  }
  const unsigned char* tail = data + 8 * n8;
  const size_t rem = len & 7;
  if (rem) {
    for (size_t j = rem; j-- > 0;) h ^= (uint64_t)tail[j] << (8 * j);
    h *= m;
  }
  h ^= h >> r; h *= m; h ^= h >> r;
  return h;
}

inline uint64_t read_bits(const std::vector<unsigned char>& d, uint64_t bit_off, int nbits) {
  uint64_t lo = 0, hi = 0;
  const size_t byte = bit_off >> 3;
  memcpy(&lo, d.data() + byte, std::min<size_t>(8, d.size() - byte));
  if (byte + 8 < d.size()) memcpy(&hi, d.data() + byte + 8, std::min<size_t>(8, d.size() - byte - 8));
  const int sh = bit_off & 7;
  uint64_t v = lo >> sh;
  if (sh) v |= hi << (64 - sh);
  return nbits >= 64 ? v : v & ((1ULL << nbits) - 1);
}

inline int required_bits(uint64_t x) { int b = 0; while (x) { ++b; x >>= 1; } return b; }

const char kKenlmMagic[] = "mmap lm http://kheafield.com/code format version 5\n";

// KenLM binary, model type 0 (probing hash tables, kenlm's default; what the ctc_decoders scorer
// loads through LoadVirtual). Layout: include/os2s.h — validated on the reference's
// toy_data-lm.binary (every stored key reproduced by the hash chain, every context sums to 1).
int read_kenlm_probing(const std::vector<unsigned char>& d, size_t size, NGramLM* lm) {
  size_t off = 0x58;
  const int order = d[off];
  float mult;
  memcpy(&mult, d.data() + off + 4, 4);
  if (order < 1 || order > kMaxLmOrder || !d[off + 12]) return OS2S_ERR_UNSUPPORTED;
  std::vector<uint64_t> counts(order);
  if (off + 20 + 8 * (size_t)order > size) return OS2S_ERR_INVALID_ARG;
  memcpy(counts.data(), d.data() + off + 20, 8 * (size_t)order);
  off = (off + 20 + 8 * (size_t)order + 7) & ~(size_t)7;
  // every stored entry takes at least 8 bytes: counts beyond size / 8 (or a multiplier that is
  // not a small finite number) can only come from a corrupt header and would overflow the
  // offset arithmetic below
  if (!(mult >= 1.f && mult <= 16.f)) return OS2S_ERR_INVALID_ARG;
  for (int n = 0; n < order; ++n)
    if (counts[n] > size / 8) return OS2S_ERR_INVALID_ARG;
  auto buckets = [&](uint64_t n) { return std::max<uint64_t>(n + 1, (uint64_t)(mult * (float)n)); };
  off += 8 + 12 * buckets(counts[0]);
  if (off + 8 * (counts[0] + 1) > size) return OS2S_ERR_INVALID_ARG;
  lm->order = order;
  lm->entries.assign(order, {});
  lm->index.assign(order, {});
  lm->hashed.assign(order, {});
  lm->use_hash = true;
  lm->words.clear();
  lm->entries[0].resize(counts[0]);
  for (uint64_t w = 0; w < counts[0]; ++w) {
    float pb[2];
    memcpy(pb, d.data() + off + 8 * w, 8);
    lm->entries[0][w] = NGramLM::Entry{-fabsf(pb[0]), pb[1]};     // sign bit = flag
  }
  off += 8 * (counts[0] + 1);
  for (int n = 2; n <= order; ++n) {
    const uint64_t nb = buckets(counts[n - 1]);
    const size_t esz = n < order ? 16 : 12;
    if (off + esz * nb > size) return OS2S_ERR_INVALID_ARG;
    auto& tab = lm->hashed[n - 1];
    tab.reserve(counts[n - 1] * 2);
    for (uint64_t i = 0; i < nb; ++i) {
      uint64_t key;
      float pb[2] = {0.f, 0.f};
      memcpy(&key, d.data() + off + esz * i, 8);
      memcpy(pb, d.data() + off + esz * i + 8, esz - 8);
      if (key) tab[key] = NGramLM::Entry{-fabsf(pb[0]), n < order ? pb[1] : 0.f};
    }
    if (tab.size() != counts[n - 1]) return OS2S_ERR_INVALID_ARG;
    off += esz * nb;
  }
  // strings in id order
  size_t p = off;
  for (uint64_t i = 0; i < counts[0] && p < size; ++i) {
    const char* str = (const char*)d.data() + p;
    const size_t len = strnlen(str, size - p);
    lm->words[std::string(str, len)] = (uint32_t)i;
    p += len + 1;
  }
  if (lm->words.size() != counts[0]) return OS2S_ERR_INVALID_ARG;
  auto unk = lm->words.find("<unk>");
  auto bos = lm->words.find("<s>");
  if (unk == lm->words.end() || unk->second != 0 || bos == lm->words.end()) return OS2S_ERR_INVALID_ARG;
  lm->bos = bos->second;
  return OS2S_OK;
}

// KenLM binary: model type 0 (above) or model type 5 (quantised array trie), order 2. Layout in
// os2s.h / oracle.
int read_kenlm_binary(const std::string& path, NGramLM* lm) {
  std::ifstream in(path, std::ios::binary);
  if (!in) return OS2S_ERR_INVALID_ARG;
  std::vector<unsigned char> d((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  d.resize(d.size() + 16, 0);     // slack for unaligned 64-bit bit-field reads
  const size_t size = d.size() - 16;
  if (size < 0x80) return OS2S_ERR_INVALID_ARG;
  size_t off = 0x58;
  const int order = d[off];
  uint32_t model_type;
  memcpy(&model_type, d.data() + off + 8, 4);
  const int has_vocab = d[off + 12];
  if (model_type == 0) return read_kenlm_probing(d, size, lm);
  if (model_type != 5 || order != 2) return OS2S_ERR_UNSUPPORTED;
  if (!has_vocab) return OS2S_ERR_INVALID_ARG;
  uint64_t counts[2];
  memcpy(counts, d.data() + off + 20, 16);
  off = (off + 20 + 8 * order + 7) & ~(size_t)7;
  uint64_t nvoc;
  memcpy(&nvoc, d.data() + off, 8);
  if (nvoc + 1 != counts[0] || off + 8 * (counts[0] + 1) > size) return OS2S_ERR_INVALID_ARG;
  std::vector<uint64_t> hashes(nvoc);
  memcpy(hashes.data(), d.data() + off + 8, 8 * nvoc);
  off += 8 * (counts[0] + 1);
  const int ver = d[off], pbits = d[off + 1];
  if (ver != 2 || pbits < 1 || pbits > 25) return OS2S_ERR_INVALID_ARG;
  off += 8;
  const size_t bins_off = off;
  off += (size_t)4 << pbits;
  const size_t uni_off = off;
  off += 16 * (counts[0] + 2);
  const int wbits = required_bits(counts[0]);
  const int tot = wbits + pbits;
  const size_t long_off = off;
  off += ((1 + counts[1]) * tot + 7) / 8 + 8;
  if (off > size) return OS2S_ERR_INVALID_ARG;
  // strings, in ARPA order; ids: <unk> = 0, others 1 + rank of their hash
  lm->order = 2;
  lm->entries.assign(2, {});
  lm->index.assign(2, {});
  lm->words.clear();
  size_t p = off;
  for (uint64_t i = 0; i < counts[0] && p < size; ++i) {
    const char* s = (const char*)d.data() + p;
    const size_t len = strnlen(s, size - p);
    std::string w(s, len);
    p += len + 1;
    uint32_t id = 0;
    if (w != "<unk>") {
      const uint64_t h = murmur_hash64a(w.data(), w.size(), 0);
      auto it = std::lower_bound(hashes.begin(), hashes.end(), h);
      if (it == hashes.end() || *it != h) return OS2S_ERR_INVALID_ARG;
      id = 1 + (uint32_t)(it - hashes.begin());
    }
    lm->words[w] = id;
  }
  if (lm->words.size() != counts[0]) return OS2S_ERR_INVALID_ARG;
  auto bos = lm->words.find("<s>");
  if (bos == lm->words.end()) return OS2S_ERR_INVALID_ARG;
  lm->bos = bos->second;
  std::vector<uint64_t> next(counts[0] + 1);
  lm->entries[0].resize(counts[0]);
  for (uint64_t w = 0; w <= counts[0]; ++w) {
    float pb[2];
    memcpy(pb, d.data() + uni_off + 16 * w, 8);
    memcpy(&next[w], d.data() + uni_off + 16 * w + 8, 8);
    // the sign bit of an in-trie probability is a flag (left-extension independence)
    if (w < counts[0]) lm->entries[0][w] = NGramLM::Entry{-fabsf(pb[0]), pb[1]};
  }
  // the trie is keyed by the predicted word first: entries of word w are its one-word contexts
  for (uint64_t w = 0; w < counts[0]; ++w) {
    if (next[w] > next[w + 1] || next[w + 1] > counts[1]) return OS2S_ERR_INVALID_ARG;
    for (uint64_t e = next[w]; e < next[w + 1]; ++e) {
      const uint32_t ctx = (uint32_t)read_bits(d, long_off * 8 + e * tot, wbits);
      const uint32_t q = (uint32_t)read_bits(d, long_off * 8 + e * tot + wbits, pbits);
      float prob;
      memcpy(&prob, d.data() + bins_off + 4 * (size_t)q, 4);
      if (ctx >= counts[0]) return OS2S_ERR_INVALID_ARG;
      const uint32_t ids[2] = {ctx, (uint32_t)w};
      lm->add(ids, 2, prob, 0.f);
    }
  }
  return OS2S_OK;
}

// ---------------------------------------------------------------------------------------------
// letter trie (trie_node.h:165-189: text file, depth-first, -1 = no child)
// ---------------------------------------------------------------------------------------------
struct LetterTrie {
  struct Node { float min_unigram_score; int32_t first_child; };   // children at [first_child, +V)
  int V = 0;
  std::vector<Node> nodes;
  std::vector<int32_t> child;     // node index or -1

  int32_t child_of(int32_t node, int label) const { return child[(size_t)nodes[node].first_child + label]; }

  bool read(const std::string& path, int vocab_size) {
    std::ifstream in(path);
    if (!in) return false;
    long long magic = 0, version = 0, v = 0;
    in >> magic >> version >> v;
    if (magic != 0x54524945 || version != 1 || v != vocab_size) return false;
    V = vocab_size;
    // iterative depth-first parse (tries of real vocabularies are deep and wide)
    struct Frame { int32_t node; int next; };
    std::vector<Frame> stack;
    long long c;
    if (!(in >> c) || c == -1) return false;
    auto new_node = [&](void) -> int32_t {
      long long word; float score;
      in >> word >> score;
      Node n; n.min_unigram_score = score; n.first_child = (int32_t)child.size();
      child.resize(child.size() + V, -1);
      nodes.push_back(n);
      return (int32_t)nodes.size() - 1;
    };
    stack.push_back(Frame{new_node(), 0});
    while (!stack.empty()) {
      Frame& f = stack.back();
      if (f.next == V) { stack.pop_back(); continue; }
      const int slot = f.next++;
      const int32_t parent = f.node;
      if (!(in >> c)) return false;
      if (c == -1) continue;
      const int32_t n = new_node();
      child[(size_t)nodes[parent].first_child + slot] = n;
      stack.push_back(Frame{n, 0});
    }
    return true;
  }
};

// alphabet.h:24-40: one label per line, "\#" is '#', other lines starting with '#' are comments
bool read_alphabet(const std::string& path, std::vector<std::string>* labels) {
  std::ifstream in(path);
  if (!in) return false;
  for (std::string line; std::getline(in, line);) {
    if (line.size() == 2 && line[0] == '\\' && line[1] == '#') line = "#";
    else if (!line.empty() && line[0] == '#') continue;
    labels->push_back(line);
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// scorer
// ---------------------------------------------------------------------------------------------
constexpr int kMaxOrder = 8;
constexpr uint32_t kNoWord = 0xffffffffu;

struct Scorer {
  NGramLM lm;
  LetterTrie trie;
  std::vector<std::string> alphabet;
  std::vector<char> is_space;
  float alpha = 0.f, beta = 0.f, trie_weight = 0.f;
};

struct LMState {
  float lm_score = 0.f, score = 0.f;
  int32_t node = 0;                   // letter-trie node of the incomplete word, -1 = none
  bool new_word = false;
  uint8_t n_hist = 0;                 // valid entries of hist (<= order)
  uint32_t n_words = 0;               // words in the prefix so far
  uint32_t hist[kMaxOrder];           // ids of the last min(order, n_words) words, oldest first
  std::string word;                   // incomplete word
};

struct ScorerOps {
  const Scorer* s;
  void init(LMState* st) const { *st = LMState(); }
  void copy(const LMState& f, LMState* t) const {
    t->lm_score = 0.f; t->score = 0.f; t->new_word = false;
    t->word = f.word; t->node = f.node; t->n_hist = f.n_hist; t->n_words = f.n_words;
    memcpy(t->hist, f.hist, sizeof(uint32_t) * f.n_hist);
  }
  void push_word(LMState* st) const {
    const int order = s->lm.order;
    const uint32_t id = s->lm.word_index(st->word);
    if (st->n_hist == order) {
      memmove(st->hist, st->hist + 1, sizeof(uint32_t) * (order - 1));
      st->n_hist--;
    }
    st->hist[st->n_hist++] = id;
    st->n_words++;
    st->word.clear();
    st->node = 0;
    st->new_word = true;
    st->lm_score = score_ngram(*st);
  }
  // beam_search.h:172-200
  float score_ngram(const LMState& st) const {
    const int order = s->lm.order;
    uint32_t h[2 * kMaxOrder];
    int nh = 0;
    if ((int)st.n_words < order)
      for (int i = 0; i < order - (int)st.n_words; ++i) h[nh++] = s->lm.bos;
    float prob = 0.f;
    for (int i = 0; i < st.n_hist; ++i) {
      if (st.hist[i] == 0) return -100.f;
      prob = s->lm.score(h, nh, st.hist[i]);
      h[nh++] = st.hist[i];
    }
    return prob;
  }
  void expand(const LMState& f, int from_label, LMState* t, int to_label) const {
    copy(f, t);
    if (!s->is_space[to_label]) {
      t->word += s->alphabet[to_label];
      float sc = -100.f;
      int32_t node = f.node;
      if (node >= 0) {
        node = s->trie.child_of(node, to_label);
        t->node = node;
        if (node >= 0) sc = s->trie.nodes[node].min_unigram_score;
      }
      t->score = sc;
    } else {
      if (from_label == to_label) return;
      push_word(t);
    }
  }
  void expand_end(LMState* st) const { if (!st->word.empty()) push_word(st); }
  float expansion_score(const LMState& st, float previous) const {
    return st.new_word ? previous + (s->alpha * st.lm_score + s->beta) : previous + s->trie_weight * st.score;
  }
  float end_score(const LMState& st) const { return st.new_word ? s->alpha * st.lm_score + s->beta : 0.f; }
};

// ---------------------------------------------------------------------------------------------
// prefix beam search
// ---------------------------------------------------------------------------------------------
struct Prob {
  float total = kLogZero, blank = kLogZero, label = kLogZero;
  void reset() { total = blank = label = kLogZero; }
};

struct BeamEntry {
  BeamEntry* parent;
  int label;
  uint32_t serial;
  Prob oldp, newp;
  std::vector<std::pair<int, BeamEntry*>> children;
  LMState state;
  bool active() const { return newp.total != kLogZero; }
  BeamEntry* child(int l) const {
    for (auto& c : children) if (c.first == l) return c.second;
    return nullptr;
  }
};

inline bool better(const BeamEntry* a, const BeamEntry* b) {
  return a->newp.total > b->newp.total || (a->newp.total == b->newp.total && a->serial < b->serial);
}

// the `limit` best entries; heap front = the worst of them
struct Leaves {
  size_t limit;
  std::vector<BeamEntry*> h;
  static bool cmp(const BeamEntry* a, const BeamEntry* b) { return better(a, b); }   // min-heap of "better"
  size_t size() const { return h.size(); }
  BeamEntry* bottom() const { return h.front(); }
  void push(BeamEntry* e) {
    if (h.size() < limit) { h.push_back(e); std::push_heap(h.begin(), h.end(), cmp); return; }
    if (e->newp.total > h.front()->newp.total) {
      std::pop_heap(h.begin(), h.end(), cmp);
      h.back() = e;
      std::push_heap(h.begin(), h.end(), cmp);
    }
  }
  void extract(std::vector<BeamEntry*>* out) {
    out->assign(h.begin(), h.end());
    std::sort(out->begin(), out->end(), better);
    h.clear();
  }
};

struct Decoder {
  int C, beam_width;
  const Scorer* scorer;      // may be null
  std::deque<BeamEntry> arena;
  uint32_t serial = 0;
  Leaves leaves;
  std::vector<BeamEntry*> branches;

  BeamEntry* new_entry(BeamEntry* parent, int label) {
    arena.emplace_back();
    BeamEntry* e = &arena.back();
    e->parent = parent; e->label = label; e->serial = serial++;
    if (parent) parent->children.emplace_back(label, e);
    return e;
  }
  bool is_candidate(const Prob& p) const {
    return p.total > kLogZero && (leaves.size() < (size_t)beam_width || p.total > leaves.bottom()->newp.total);
  }

  // one utterance; logits row t at logits + t * ld
  int run(const float* logits, long long ld, int T, int top_paths, int merge_repeated,
          int32_t* out_ids, int out_ld, int32_t* out_len, float* out_logp) {
    ScorerOps ops{scorer};
    arena.clear();
    serial = 0;
    leaves.limit = beam_width;
    leaves.h.clear();
    BeamEntry* root = new_entry(nullptr, -1);
    root->newp.total = 0.f;
    root->newp.blank = 0.f;
    if (scorer) ops.init(&root->state);
    leaves.push(root);
    const int blank = C - 1;
    LMState tmp;
    for (int t = 0; t < T; ++t) {
      const float* raw = logits + (long long)t * ld;
      float mx = raw[0];
      for (int j = 1; j < C; ++j) mx = std::max(mx, raw[j]);
      float sum = 0.f;
      for (int j = 0; j < C; ++j) sum += expf(raw[j] - mx);
      const float norm = mx + logf(sum);
      leaves.extract(&branches);
      for (BeamEntry* b : branches) b->oldp = b->newp;
      for (BeamEntry* b : branches) {
        if (b->parent != nullptr) {
          if (b->parent->active()) {
            const float prev = b->label == b->parent->label ? b->parent->oldp.blank : b->parent->oldp.total;
            b->newp.label = log_sum_exp(b->newp.label, scorer ? ops.expansion_score(b->state, prev) : prev);
          }
          b->newp.label += raw[b->label] - norm;
        }
        b->newp.blank = b->oldp.total + raw[blank] - norm;
        b->newp.total = log_sum_exp(b->newp.blank, b->newp.label);
        leaves.push(b);
      }
      for (BeamEntry* b : branches) {
        if (!is_candidate(b->oldp)) continue;
        for (int label = 0; label < C - 1; ++label) {
          BeamEntry* c = b->child(label);
          if (c && c->active()) continue;
          const float prev = label == b->label ? b->oldp.blank : b->oldp.total;
          float ext = prev;
          if (scorer) {
            ops.expand(b->state, b->label, &tmp, label);
            ext = ops.expansion_score(tmp, prev);
          }
          Prob np;
          np.blank = kLogZero;
          np.label = raw[label] - norm + ext;
          np.total = np.label;
          if (is_candidate(np)) {
            if (!c) c = new_entry(b, label);
            if (scorer) std::swap(c->state, tmp);
            c->newp = np;
            if (leaves.size() == (size_t)beam_width) leaves.bottom()->newp.reset();
            leaves.push(c);
          } else if (c) {
            if (scorer) std::swap(c->state, tmp);
            c->oldp.reset();
            c->newp.reset();
          }
        }
      }
    }
    // end-of-sequence rescoring (beam_search.cc:730-739) and TopPaths (:398-429)
    leaves.extract(&branches);
    for (BeamEntry* e : branches) {
      if (scorer) {
        ops.expand_end(&e->state);
        e->newp.total += ops.end_score(e->state);
      }
    }
    std::sort(branches.begin(), branches.end(), better);
    if ((size_t)top_paths > branches.size()) return OS2S_ERR_INVALID_ARG;
    std::vector<int> seq;
    for (int i = 0; i < top_paths; ++i) {
      seq.clear();
      int prev = -1;
      for (const BeamEntry* c = branches[i]; c->parent != nullptr; c = c->parent) {
        if (!merge_repeated || c->label != prev) seq.push_back(c->label);
        prev = c->label;
      }
      const int n = (int)seq.size();
      if (n > out_ld) return OS2S_ERR_INVALID_ARG;
      for (int j = 0; j < n; ++j) out_ids[(long long)i * out_ld + j] = seq[n - 1 - j];
      for (int j = n; j < out_ld; ++j) out_ids[(long long)i * out_ld + j] = -1;
      out_len[i] = n;
      out_logp[i] = branches[i]->newp.total;
    }
    return OS2S_OK;
  }
};

}  // namespace

// Every model file goes through here. The counts in a model header are untrusted input: a
// corrupt file must come back as a status code, never as an exception crossing the C ABI
// (no exceptions cross the boundary).
static int load_lm(const char* lm_path, NGramLM* lm) {
  try {
    std::ifstream in(lm_path, std::ios::binary);
    if (!in) return OS2S_ERR_INVALID_ARG;
    char head[sizeof(kKenlmMagic)] = {0};
    in.read(head, sizeof(kKenlmMagic) - 1);
    in.close();
    if (memcmp(head, kKenlmMagic, sizeof(kKenlmMagic) - 1) == 0) return read_kenlm_binary(lm_path, lm);
    if (!read_arpa(lm_path, lm)) return OS2S_ERR_INVALID_ARG;
    return lm->order > kMaxOrder ? OS2S_ERR_UNSUPPORTED : OS2S_OK;
  } catch (const std::exception&) {           // bad_alloc / length_error from a hostile header
    return OS2S_ERR_INVALID_ARG;
  }
}

// generate_trie.cpp:32-64 + TrieNode::Insert / WriteToStream (trie_node.h:46-50,122-163)
extern "C" int os2s_ctc_generate_trie(const char* alphabet_path, const char* lm_path,
                                      const char* vocab_path, const char* trie_path) {
  if (!alphabet_path || !lm_path || !vocab_path || !trie_path) return OS2S_ERR_INVALID_ARG;
  std::vector<std::string> alphabet;
  if (!read_alphabet(alphabet_path, &alphabet) || alphabet.empty()) return OS2S_ERR_INVALID_ARG;
  std::unordered_map<std::string, int> to_label;
  for (size_t i = 0; i < alphabet.size(); ++i) to_label[alphabet[i]] = (int)i;
  NGramLM lm;
  const int rc = load_lm(lm_path, &lm);
  if (rc != OS2S_OK) return rc;
  std::ifstream ifs(vocab_path);
  if (!ifs) return OS2S_ERR_INVALID_ARG;
  const int V = (int)alphabet.size();
  struct Node { int prefix_count = 0; uint32_t min_word = 0; float min_score = std::numeric_limits<float>::max(); std::vector<int32_t> child; };
  std::vector<Node> nodes(1);
  nodes[0].child.assign(V, -1);
  std::string word;
  std::vector<int> labels;
  while (ifs >> word) {
    const uint32_t id = lm.word_index(word);
    const float score = lm.score(nullptr, 0, id);      // FullScore from the null context
    labels.clear();
    for (size_t p = 0; p < word.size();) {               // one label per UTF-8 code point
      const unsigned char c = (unsigned char)word[p];
      const size_t n = c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : 4;
      auto it = to_label.find(word.substr(p, n));
      if (it == to_label.end()) return OS2S_ERR_INVALID_ARG;   // LabelFromString aborts in the reference
      labels.push_back(it->second);
      p += n;
    }
    int32_t node = 0;
    for (size_t i = 0;; ++i) {
      Node& nd = nodes[node];
      nd.prefix_count++;
      if (score < nd.min_score) { nd.min_score = score; nd.min_word = id; }
      if (i == labels.size()) break;
      int32_t ch = nd.child[labels[i]];
      if (ch < 0) {
        ch = (int32_t)nodes.size();
        nodes[node].child[labels[i]] = ch;
        nodes.emplace_back();
        nodes.back().child.assign(V, -1);
      }
      node = ch;
    }
  }
  std::ofstream ofs(trie_path);
  if (!ofs) return OS2S_ERR_INVALID_ARG;
  ofs << 0x54524945 << "\n" << 1 << "\n" << V << "\n";
  // depth-first, children in label order, -1 for a missing child
  struct Frame { int32_t node; int next; };
  std::vector<Frame> stack;
  auto emit = [&](int32_t n) {
    ofs << nodes[n].prefix_count << "\n" << nodes[n].min_word << "\n" << nodes[n].min_score << "\n";
    stack.push_back(Frame{n, 0});
  };
  emit(0);
  while (!stack.empty()) {
    Frame& f = stack.back();
    if (f.next == V) { stack.pop_back(); continue; }
    const int32_t ch = nodes[f.node].child[f.next++];
    if (ch < 0) ofs << -1 << "\n";
    else emit(ch);
  }
  return ofs.good() ? OS2S_OK : OS2S_ERR_INVALID_ARG;
}

extern "C" int os2s_ctc_scorer_create(const char* lm_path, const char* trie_path,
                                      const char* alphabet_path, float alpha, float beta,
                                      float trie_weight, void** scorer) {
  if (!lm_path || !trie_path || !alphabet_path || !scorer) return OS2S_ERR_INVALID_ARG;
  std::unique_ptr<Scorer> s(new Scorer);
  s->alpha = alpha; s->beta = beta; s->trie_weight = trie_weight;
  if (!read_alphabet(alphabet_path, &s->alphabet) || s->alphabet.empty()) return OS2S_ERR_INVALID_ARG;
  for (auto& l : s->alphabet) s->is_space.push_back(l.size() == 1 && l[0] == ' ');
  const int rc = load_lm(lm_path, &s->lm);
  if (rc != OS2S_OK) return rc;
  if (!s->trie.read(trie_path, (int)s->alphabet.size())) return OS2S_ERR_INVALID_ARG;
  *scorer = s.release();
  return OS2S_OK;
}

// WordLMBeamScorer::SetAlpha / SetBeta / SetTrieWeight (beam_search.h:150-160)
extern "C" int os2s_ctc_scorer_set_weights(void* scorer, float alpha, float beta, float trie_weight) {
  if (!scorer) return OS2S_ERR_INVALID_ARG;
  Scorer* s = static_cast<Scorer*>(scorer);
  s->alpha = alpha; s->beta = beta; s->trie_weight = trie_weight;
  return OS2S_OK;
}

extern "C" void os2s_ctc_scorer_destroy(void* scorer) { delete static_cast<Scorer*>(scorer); }

extern "C" int os2s_ctc_scorer_ngram_score(const void* scorer, const char* const* words, int n_words,
                                           float* log10_prob) {
  if (!scorer || !words || n_words < 1 || !log10_prob) return OS2S_ERR_INVALID_ARG;
  const Scorer* s = static_cast<const Scorer*>(scorer);
  ScorerOps ops{s};
  LMState st;
  ops.init(&st);
  for (int i = 0; i < n_words; ++i) {
    st.word = words[i];
    ops.push_word(&st);
  }
  *log10_prob = st.lm_score;
  return OS2S_OK;
}

extern "C" int os2s_ctc_beam_search(const float* logits, long long ld_t, long long ld_b,
                                    const int32_t* seq_len, int T, int B, int C, int beam_width,
                                    int top_paths, int merge_repeated, const void* scorer,
                                    int n_threads, int32_t* out_ids, int32_t* out_len,
                                    float* out_log_prob) {
  if (!logits || !seq_len || !out_ids || !out_len || !out_log_prob) return OS2S_ERR_INVALID_ARG;
  if (T < 0 || B < 1 || C < 2 || beam_width < 1 || top_paths < 1 || top_paths > beam_width)
    return OS2S_ERR_INVALID_ARG;
  const Scorer* s = static_cast<const Scorer*>(scorer);
  if (s && (int)s->alphabet.size() != C - 1) return OS2S_ERR_INVALID_ARG;
  for (int b = 0; b < B; ++b)
    if (seq_len[b] < 0 || seq_len[b] > T) return OS2S_ERR_INVALID_ARG;
  int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
  nt = std::max(1, std::min(nt, B));
  std::atomic<int> next(0), status(OS2S_OK);
  auto work = [&]() {
    Decoder dec;
    dec.C = C; dec.beam_width = beam_width; dec.scorer = s;
    for (int b = next.fetch_add(1); b < B; b = next.fetch_add(1)) {
      const int rc = dec.run(logits + (long long)b * ld_b, ld_t, seq_len[b], top_paths, merge_repeated,
                             out_ids + (long long)b * top_paths * T, T, out_len + (long long)b * top_paths,
                             out_log_prob + (long long)b * top_paths);
      if (rc != OS2S_OK) status.store(rc);
    }
  };
  if (nt == 1) {
    work();
  } else {
    std::vector<std::thread> pool;
    for (int i = 0; i < nt; ++i) pool.emplace_back(work);
    for (auto& th : pool) th.join();
  }
  return status.load();
}

// =============================================================================================
// The reference's second decoder: the `ctc_decoders` module (decoders/*.cpp, a swig wrapper
// around a prefix beam search over softmax PROBABILITIES with an external scorer and a dictionary
// constraint), used offline by scripts/decode.py. Restated from
// decoders/ctc_beam_search_decoder.cpp:18-178, path_trie.cpp:37-158, scorer.cpp:71-230,
// decoder_utils.cpp:7-95. The OpenFST dictionary (a minimised automaton of {word + ' '}) is a
// character trie here: any deterministic automaton of that language answers the two questions
// the decoder asks (is there an arc, is the state final) identically.
// =============================================================================================
namespace {

constexpr float kFltMax = std::numeric_limits<float>::max();
constexpr float kFltMin = std::numeric_limits<float>::min();
constexpr float kNeg = -kFltMax;
constexpr double kOovScore = -1000.0;

template <typename T>
inline T lse2(T x, T y) {
  if (x <= -std::numeric_limits<T>::max()) return y;
  if (y <= -std::numeric_limits<T>::max()) return x;
  const T m = std::max(x, y);
  return std::log(std::exp(x - m) + std::exp(y - m)) + m;
}

struct DictScorer {
  NGramLM lm;
  std::vector<std::string> alphabet;
  double alpha = 0.0, beta = 0.0;
  bool char_based = true;
  int space_id = -1, max_order = 0, dict_size = 0;
  int V = 0;
  std::vector<int32_t> arcs;     // [state][label] -> state or -1
  std::vector<char> final_;

  int32_t arc(int32_t state, int label) const { return arcs[(size_t)state * V + label]; }

  // scorer.cpp:71-90
  double log_cond_prob(const std::vector<uint32_t>& ids) const {
    uint32_t hist[32];
    int nh = 0;
    double p = 0.0;
    for (uint32_t w : ids) {
      if (w == 0) return kOovScore;
      p = lm.score(hist, nh, w);
      if (nh == 32) { memmove(hist, hist + 1, sizeof(uint32_t) * 31); --nh; }
      hist[nh++] = w;
    }
    return p;
  }
};

struct PNode {
  float b_prev = kNeg, nb_prev = kNeg, b_cur = kNeg, nb_cur = kNeg, score = kNeg;
  int character = -1;
  PNode* parent = nullptr;
  bool exists = true;
  int32_t dict_state = 0;
  std::vector<std::pair<int, PNode*>> children;
};

void free_tree(PNode* n) {
  for (auto& c : n->children) free_tree(c.second);
  delete n;
}

struct DictDecoder {
  const DictScorer* sc;      // may be null
  bool use_dict;
  int V;                     // labels without blank

  // path_trie.cpp:37-86
  PNode* get_path_trie(PNode* n, int c) const {
    for (auto& ch : n->children)
      if (ch.first == c) {
        PNode* k = ch.second;
        if (!k->exists) {
          k->exists = true;
          k->b_prev = k->nb_prev = k->b_cur = k->nb_cur = kNeg;
        }
        return k;
      }
    int32_t state = 0;
    if (use_dict) {
      const int32_t nxt = sc->arc(n->dict_state, c);
      if (nxt < 0) {
        if (sc->final_[n->dict_state]) n->dict_state = 0;     // re-armed; this attempt is refused
        return nullptr;
      }
      state = nxt;
    }
    PNode* k = new PNode;
    k->character = c; k->parent = n; k->dict_state = state;
    n->children.emplace_back(c, k);
    return k;
  }
  // walks up to (not including) a node whose character == stop, or the root, or max_steps
  static PNode* path_vec(PNode* n, int stop, size_t max_steps, std::vector<int>* out) {
    out->clear();
    while (!(n->character == stop || n->character == -1 || out->size() == max_steps)) {
      out->push_back(n->character);
      n = n->parent;
    }
    std::reverse(out->begin(), out->end());
    return n;
  }
  static void iterate(PNode* n, std::vector<PNode*>* out) {
    if (n->exists) {
      n->b_prev = n->b_cur; n->nb_prev = n->nb_cur;
      n->b_cur = n->nb_cur = kNeg;
      n->score = lse2(n->b_prev, n->nb_prev);
      out->push_back(n);
    }
    for (size_t i = 0; i < n->children.size(); ++i) iterate(n->children[i].second, out);
  }
  static void remove(PNode* n) {
    n->exists = false;
    if (n->children.empty()) {
      PNode* p = n->parent;
      if (!p) return;      // the root is never freed here (an ancestor of every surviving prefix)
      for (auto it = p->children.begin(); it != p->children.end(); ++it)
        if (it->second == n) { p->children.erase(it); break; }
      if (p->children.empty() && !p->exists) remove(p);
      delete n;
    }
  }
  static bool compare(const PNode* x, const PNode* y) {
    if (x->score == y->score) return x->character != y->character && x->character < y->character;
    return x->score > y->score;
  }
  // scorer.cpp:166-200 (word ids instead of strings; "" and unknown words map to 0 = OOV)
  void make_ngram(PNode* prefix, std::vector<uint32_t>* ngram) const {
    ngram->clear();
    std::vector<int> vec;
    std::string word;
    PNode* cur = prefix;
    for (int order = 0; order < sc->max_order; ++order) {
      PNode* nn;
      if (sc->char_based) { nn = path_vec(cur, sc->space_id, 1, &vec); cur = nn; }
      else { nn = path_vec(cur, sc->space_id, (size_t)-1, &vec); cur = nn->parent; }
      word.clear();
      for (int c : vec) word += sc->alphabet[c];
      ngram->push_back(sc->lm.word_index(word));
      if (nn->character == -1) {
        for (int i = 0; i < sc->max_order - order - 1; ++i) ngram->push_back(sc->lm.bos);
        break;
      }
    }
    std::reverse(ngram->begin(), ngram->end());
  }

  int run(const float* probs, long long ld, int T, int beam_size, double cutoff_prob, int cutoff_top_n,
          int top_paths, int32_t* out_ids, int out_ld, int32_t* out_len, float* out_score) const {
    const int C = V + 1, blank = V;
    int space = -2;
    if (sc) space = sc->space_id >= 0 ? sc->space_id : -2;
    PNode* root = new PNode;
    root->score = root->b_prev = 0.f;
    std::vector<PNode*> prefixes{root};
    std::vector<std::pair<int, double>> pidx;
    std::vector<std::pair<int, float>> lpidx;
    std::vector<uint32_t> ngram;
    for (int t = 0; t < T; ++t) {
      const float* prob = probs + (long long)t * ld;
      float min_cutoff = kNeg;
      bool full_beam = false;
      if (sc) {
        const size_t n = std::min(prefixes.size(), (size_t)beam_size);
        std::sort(prefixes.begin(), prefixes.begin() + n, compare);
        min_cutoff = prefixes[n - 1]->score + std::log(prob[blank]) - (float)std::max(0.0, sc->beta);
        full_beam = n == (size_t)beam_size;
      }
      // decoder_utils.cpp:7-37
      pidx.clear();
      for (int i = 0; i < C; ++i) pidx.emplace_back(i, (double)prob[i]);
      size_t cutoff_len = C;
      if (cutoff_prob < 1.0 || (size_t)cutoff_top_n < cutoff_len) {
        std::sort(pidx.begin(), pidx.end(),
                  [](const std::pair<int, double>& a, const std::pair<int, double>& b) { return a.second > b.second; });
        if (cutoff_prob < 1.0) {
          double cum = 0.0;
          cutoff_len = 0;
          for (size_t i = 0; i < pidx.size(); ++i) {
            cum += pidx[i].second;
            cutoff_len += 1;
            if (cum >= cutoff_prob || cutoff_len >= (size_t)cutoff_top_n) break;
          }
        }
      }
      lpidx.clear();
      for (size_t i = 0; i < cutoff_len; ++i)
        lpidx.emplace_back(pidx[i].first, (float)std::log(pidx[i].second + kFltMin));
      for (auto& cl : lpidx) {
        const int c = cl.first;
        const float lp = cl.second;
        for (size_t i = 0; i < prefixes.size() && i < (size_t)beam_size; ++i) {
          PNode* prefix = prefixes[i];
          if (full_beam && lp + prefix->score < min_cutoff) break;
          if (c == blank) {
            prefix->b_cur = lse2(prefix->b_cur, lp + prefix->score);
            continue;
          }
          if (c == prefix->character) prefix->nb_cur = lse2(prefix->nb_cur, lp + prefix->nb_prev);
          PNode* nn = get_path_trie(prefix, c);
          if (!nn) continue;
          float log_p = kNeg;
          if (c == prefix->character && prefix->b_prev > kNeg) log_p = lp + prefix->b_prev;
          else if (c != prefix->character) log_p = lp + prefix->score;
          if (sc && (c == space || sc->char_based)) {
            make_ngram(sc->char_based ? nn : prefix, &ngram);
            const float score = (float)(sc->log_cond_prob(ngram) * sc->alpha);
            log_p += score;
            log_p += (float)sc->beta;
          }
          nn->nb_cur = lse2(nn->nb_cur, log_p);
        }
      }
      prefixes.clear();
      iterate(root, &prefixes);
      if (prefixes.size() >= (size_t)beam_size) {
        std::nth_element(prefixes.begin(), prefixes.begin() + beam_size, prefixes.end(), compare);
        for (size_t i = beam_size; i < prefixes.size(); ++i) remove(prefixes[i]);
        prefixes.resize(beam_size);
      }
    }
    if (sc && !sc->char_based) {
      for (size_t i = 0; i < (size_t)beam_size && i < prefixes.size(); ++i) {
        PNode* p = prefixes[i];
        if (p->character != -1 && p->character != space) {
          make_ngram(p, &ngram);
          p->score += (float)(sc->log_cond_prob(ngram) * sc->alpha + sc->beta);
        }
      }
    }
    const size_t n = std::min(prefixes.size(), (size_t)beam_size);
    std::sort(prefixes.begin(), prefixes.begin() + n, compare);
    int rc = OS2S_OK;
    if ((size_t)top_paths > n) rc = OS2S_ERR_INVALID_ARG;
    std::vector<int> path;
    for (int k = 0; k < top_paths && rc == OS2S_OK; ++k) {
      path_vec(prefixes[k], -1, (size_t)-1, &path);
      if ((int)path.size() > out_ld) { rc = OS2S_ERR_INVALID_ARG; break; }
      for (int j = 0; j < out_ld; ++j) out_ids[(long long)k * out_ld + j] = j < (int)path.size() ? path[j] : -1;
      out_len[k] = (int)path.size();
      out_score[k] = prefixes[k]->score;
    }
    free_tree(root);
    return rc;
  }
};

}  // namespace

// Scorer(alpha, beta, model_path, vocabulary) of decoders/scorer.cpp:16-52
extern "C" int os2s_ctc_dict_scorer_create(const char* lm_path, const char* const* vocabulary, int n_vocab,
                                           double alpha, double beta, void** scorer) {
  if (!lm_path || !vocabulary || n_vocab < 1 || !scorer) return OS2S_ERR_INVALID_ARG;
  std::unique_ptr<DictScorer> s(new DictScorer);
  s->alpha = alpha; s->beta = beta;
  const int rc = load_lm(lm_path, &s->lm);
  if (rc != OS2S_OK) return rc;
  s->max_order = s->lm.order;
  s->V = n_vocab;
  std::unordered_map<std::string, int> char_map;
  for (int i = 0; i < n_vocab; ++i) {
    if (!vocabulary[i]) return OS2S_ERR_INVALID_ARG;
    s->alphabet.emplace_back(vocabulary[i]);
    if (s->alphabet.back() == " ") s->space_id = i;
    char_map[s->alphabet.back()] = i;
  }
  auto utf8_chars = [](const std::string& w, std::vector<std::string>* out) {
    out->clear();
    for (size_t p = 0; p < w.size();) {
      size_t q = p + 1;
      while (q < w.size() && ((unsigned char)w[q] & 0xc0) == 0x80) ++q;
      out->emplace_back(w, p, q - p);
      p = q;
    }
  };
  std::vector<std::string> chars;
  for (auto& kv : s->lm.words) {
    if (kv.first == "<unk>" || kv.first == "<s>" || kv.first == "</s>") continue;
    utf8_chars(kv.first, &chars);
    if (chars.size() > 1) s->char_based = false;
  }
  if (!s->char_based) {
    // fill_dictionary(add_space = true), scorer.cpp:203-230
    s->arcs.assign((size_t)n_vocab, -1);
    s->final_.assign(1, 0);
    for (auto& kv : s->lm.words) {
      utf8_chars(kv.first, &chars);
      std::vector<int> seq;
      bool ok = !chars.empty();
      for (auto& ch : chars) {
        auto it = char_map.find(ch);
        if (it == char_map.end()) { ok = false; break; }
        seq.push_back(it->second);
      }
      if (!ok) continue;
      if (s->space_id >= 0) seq.push_back(s->space_id);
      int32_t st = 0;
      for (int l : seq) {
        int32_t nxt = s->arcs[(size_t)st * n_vocab + l];
        if (nxt < 0) {
          nxt = (int32_t)s->final_.size();
          s->final_.push_back(0);
          s->arcs.resize(s->arcs.size() + n_vocab, -1);
          s->arcs[(size_t)st * n_vocab + l] = nxt;
        }
        st = nxt;
      }
      s->final_[st] = 1;
      s->dict_size++;
    }
  }
  *scorer = s.release();
  return OS2S_OK;
}

extern "C" void os2s_ctc_dict_scorer_destroy(void* scorer) { delete static_cast<DictScorer*>(scorer); }

// reset_params / is_character_based / get_max_order / get_dict_size of the reference's Scorer
extern "C" int os2s_ctc_dict_scorer_set_weights(void* scorer, double alpha, double beta) {
  if (!scorer) return OS2S_ERR_INVALID_ARG;
  static_cast<DictScorer*>(scorer)->alpha = alpha;
  static_cast<DictScorer*>(scorer)->beta = beta;
  return OS2S_OK;
}

extern "C" int os2s_ctc_dict_scorer_info(const void* scorer, int* is_character_based, int* max_order,
                                         int* dict_size) {
  if (!scorer) return OS2S_ERR_INVALID_ARG;
  const DictScorer* s = static_cast<const DictScorer*>(scorer);
  if (is_character_based) *is_character_based = s->char_based;
  if (max_order) *max_order = s->max_order;
  if (dict_size) *dict_size = s->dict_size;
  return OS2S_OK;
}

extern "C" int os2s_ctc_dict_beam_search(const float* probs, long long ld_t, long long ld_b,
                                         const int32_t* seq_len, int T, int B, int C, int beam_size,
                                         double cutoff_prob, int cutoff_top_n, int top_paths,
                                         const void* scorer, int n_threads, int32_t* out_ids,
                                         int32_t* out_len, float* out_score) {
  if (!probs || !seq_len || !out_ids || !out_len || !out_score) return OS2S_ERR_INVALID_ARG;
  if (T < 1 || B < 1 || C < 2 || beam_size < 1 || top_paths < 1 || top_paths > beam_size || cutoff_top_n < 1)
    return OS2S_ERR_INVALID_ARG;
  const DictScorer* s = static_cast<const DictScorer*>(scorer);
  if (s && s->V != C - 1) return OS2S_ERR_INVALID_ARG;
  for (int b = 0; b < B; ++b)
    if (seq_len[b] < 0 || seq_len[b] > T) return OS2S_ERR_INVALID_ARG;
  int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
  nt = std::max(1, std::min(nt, B));
  std::atomic<int> next(0), status(OS2S_OK);
  auto work = [&]() {
    DictDecoder dec;
    dec.sc = s; dec.use_dict = s && !s->char_based; dec.V = C - 1;
    for (int b = next.fetch_add(1); b < B; b = next.fetch_add(1)) {
      const int rc = dec.run(probs + (long long)b * ld_b, ld_t, seq_len[b], beam_size, cutoff_prob, cutoff_top_n,
                             top_paths, out_ids + (long long)b * top_paths * T, T,
                             out_len + (long long)b * top_paths, out_score + (long long)b * top_paths);
      if (rc != OS2S_OK) status.store(rc);
    }
  };
  if (nt == 1) work();
  else {
    std::vector<std::thread> pool;
    for (int i = 0; i < nt; ++i) pool.emplace_back(work);
    for (auto& th : pool) th.join();
  }
  return status.load();
}

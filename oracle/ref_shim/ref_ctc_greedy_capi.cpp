// Test infrastructure only: a C-callable shim around the REFERENCE's
// ctc_greedy_decoder() (compiled from /root/reference/decoders/ctc_greedy_decoder.cpp
// where it lies; no reference source is copied into this repository).
// The reference returns a string built from a vocabulary; we pass a vocabulary
// of single bytes (id + 1) so the string maps 1:1 back to class ids.
#include <string>
#include <vector>
#include <cstdint>
#include "ctc_greedy_decoder.h"

extern "C" int ref_ctc_greedy_decode(const double* probs, int T, int V,
                                     int32_t* out_ids) {
  // probs: [T, V] row-major probabilities (the reference expects V == vocab+1,
  // blank last, strictly positive winners: its running max starts at 0.0).
  std::vector<std::vector<double>> seq(T, std::vector<double>(V));
  for (int t = 0; t < T; ++t)
    for (int v = 0; v < V; ++v) seq[t][v] = probs[(size_t)t * V + v];
  std::vector<std::string> vocab;
  for (int v = 0; v < V - 1; ++v) vocab.push_back(std::string(1, (char)(v + 1)));
  std::string s = ctc_greedy_decoder(seq, vocab);
  for (size_t i = 0; i < s.size(); ++i) out_ids[i] = (int32_t)((unsigned char)s[i]) - 1;
  return (int)s.size();
}

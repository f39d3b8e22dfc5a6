// Test infrastructure only. Stand-in for the reference's decoders/decoder_utils.h
// (which needs OpenFST's fst/log.h, absent here). The reference file
// decoders/ctc_greedy_decoder.cpp only uses VALID_CHECK_EQ from it
// (ctc_greedy_decoder.cpp:11-15). Pre-included with -include while the real
// header's guard DECODER_UTILS_H_ is pre-defined, so the real header is skipped.
#ifndef OS2S_REF_DECODER_UTILS_STUB_H_
#define OS2S_REF_DECODER_UTILS_STUB_H_
#include <cstdio>
#include <cstdlib>
#define VALID_CHECK_EQ(x, y, info)                                   \
  do {                                                               \
    if (!((x) == (y))) { std::fprintf(stderr, "%s\n", info); std::abort(); } \
  } while (0)
#endif

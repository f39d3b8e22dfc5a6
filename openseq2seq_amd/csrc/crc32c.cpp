// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78) for the TensorBundle checkpoint files
// (utils/tensor_bundle.py): every table block of <prefix>.index and every tensor in
// <prefix>.data-* carries one (tensorflow/core/lib/hash/crc32c.h semantics: Extend / Value; the
// masking is done by the caller). Host-only: SSE4.2 crc32 instruction when the CPU has it, else a
// slice-by-1 table.
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace {

uint32_t g_table[256];
bool g_table_ready = false;

void build_table() {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
    g_table[i] = c;
  }
  g_table_ready = true;
}

uint32_t crc_table(uint32_t c, const unsigned char* p, size_t n) {
  if (!g_table_ready) build_table();
  for (size_t i = 0; i < n; ++i) c = g_table[(c ^ p[i]) & 0xff] ^ (c >> 8);
  return c;
}

__attribute__((target("sse4.2"))) uint32_t crc_hw(uint32_t c, const unsigned char* p, size_t n) {
  uint64_t c64 = c;
  while (n && (reinterpret_cast<uintptr_t>(p) & 7)) { c64 = __builtin_ia32_crc32qi((uint32_t)c64, *p++); --n; }
  while (n >= 8) {
    uint64_t v;
    std::memcpy(&v, p, 8);
    c64 = __builtin_ia32_crc32di(c64, v);
    p += 8;
    n -= 8;
  }
  while (n) { c64 = __builtin_ia32_crc32qi((uint32_t)c64, *p++); --n; }
  return (uint32_t)c64;
}

}  // namespace

// crc = os2s_crc32c(0, data, n); extend with os2s_crc32c(crc, more, m)
extern "C" uint32_t os2s_crc32c(uint32_t init, const void* data, size_t n) {
  const unsigned char* p = static_cast<const unsigned char*>(data);
  uint32_t c = init ^ 0xffffffffu;
  c = __builtin_cpu_supports("sse4.2") ? crc_hw(c, p, n) : crc_table(c, p, n);
  return c ^ 0xffffffffu;
}

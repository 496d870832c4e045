// 64-bit digest of host memory blocks: what a device-resident copy of an emulator was made from
// (gp_content_digest, gp_mv_predict_host_checked in gp_abi.hip).  Plain host C++ -- no HIP -- so the hot loop can be
// compiled in per-ISA clones picked at load time.  Every byte counts: 32 interleaved rotate-add lanes over the
// 8-byte words (h = rotl(h, 5) + w: every step is a bijection of its lane, so a change of any one word changes
// the result), folded with multiplies at the end.  It runs at cache / memory speed; the checked calls run it
// while the device works and the calling thread would only wait.
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define GP_DIGEST_LANES 32

__attribute__((target_clones("avx512f", "avx2", "default")))
static void digest_words(uint64_t* __restrict h, const uint64_t* __restrict w, size_t n_groups) {
  uint64_t a[GP_DIGEST_LANES];
  for (int l = 0; l < GP_DIGEST_LANES; ++l) a[l] = h[l];
  for (size_t i = 0; i < n_groups; ++i) {
#pragma clang loop vectorize(enable) interleave(enable)
    for (int l = 0; l < GP_DIGEST_LANES; ++l)
      a[l] = ((a[l] << 5) | (a[l] >> 59)) + w[i * GP_DIGEST_LANES + l];
  }
  for (int l = 0; l < GP_DIGEST_LANES; ++l) h[l] = a[l];
}

uint64_t gp_host_content_digest(const void* const* blocks, const int64_t* nbytes, int n_blocks) {
  constexpr uint64_t K = 0x9E3779B97F4A7C15ull;
  uint64_t h[GP_DIGEST_LANES];
  for (int l = 0; l < GP_DIGEST_LANES; ++l) h[l] = K * (uint64_t)(l + 1);
  for (int b = 0; b < n_blocks; ++b) {
    const unsigned char* p8 = (const unsigned char*)blocks[b];
    const size_t n = (size_t)nbytes[b];
    const size_t nw = n / 8;
    size_t i = 0;
    if (((uintptr_t)p8 & 7) == 0) {
      const size_t groups = nw / GP_DIGEST_LANES;
      digest_words(h, (const uint64_t*)p8, groups);
      i = groups * GP_DIGEST_LANES;
    }
    for (; i < nw; ++i) {       // (unaligned block, or the words beyond the last whole group)
      uint64_t w;
      memcpy(&w, p8 + 8 * i, 8);
      uint64_t& x = h[i % GP_DIGEST_LANES];
      x = ((x << 5) | (x >> 59)) + w;
    }
    uint64_t tail = 0;
    memcpy(&tail, p8 + 8 * nw, n - 8 * nw);
    h[0] = (h[0] + tail + (uint64_t)n) * K;      // (the block's length counts too)
  }
  uint64_t r = 0;
  for (int l = 0; l < GP_DIGEST_LANES; ++l) r = (r ^ h[l]) * K + (r >> 29);
  return r;
}

// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11): the counter-based
// generator behind every random draw on the device.  The reference draws from one std::mt19937 in program
// order (CastPhotons.h:86-92); here a draw is a pure function of (key, counter), so it does not depend on
// which lane, wave or GPU makes it.
#ifndef PB_PHILOX_H
#define PB_PHILOX_H
#include "pb_internal.h"

PB_DI void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

// std::uniform_real_distribution<double>(0, 1): 53 bits of philox(counter = (c0, c1, 0, 0), key = (k0, k1)).
PB_DI double philox_canonical(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1) {
  uint32_t c[4] = {c0, c1, 0u, 0u};
  philox4x32_10(c, k0, k1);
  const uint64_t hi = c[0] >> 5, lo = c[1] >> 6;     // 27 + 26 bits
  return (double)((hi << 26) | lo) * (1.0 / 9007199254740992.0);
}

// Key word that separates the camera rays' draws (SHORT_VIEW, skeleton.cpp:442-449) from the photons'.
#define PB_VIEW_KEY 0x56494557u

#endif

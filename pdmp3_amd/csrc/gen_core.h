// gen_core.h -- the synthetic-workload generator of SURVEY.md 8d (configs C2
// and C5): integer-only, counter-based splitmix64, so every GPU can produce
// its own shard of the stream on device and the host can reproduce any frame.
//
// Product-side twin of oracle/pdmp3_oracle.c:orc_generate_frames (the oracle
// has its own independent copy; tests/test_generator.py checks they agree
// byte for byte).  Compiles for device (hipcc) and host (g++).
#pragma once

#include <stdint.h>
#include "../../include/pdmp3_hip.h"

#if defined(__HIPCC__)
#define PG_FN __host__ __device__ __forceinline__
#else
#define PG_FN static inline
#endif

namespace pdmp3 {

PG_FN uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// key = seed XOR (frame*4096 + gr*2048 + ch*1024 + slot); slots 0..575 are
// spectral lines, 576.. are side-info fields.
PG_FN uint64_t gen_r(uint64_t seed, int64_t frame, unsigned gr, unsigned ch, unsigned slot) {
  return splitmix64(seed ^ ((uint64_t)frame * 4096u + gr * 2048u + ch * 1024u + slot));
}

PG_FN unsigned gen_scalefac_l(uint64_t seed, int64_t frame, unsigned gr, unsigned ch, unsigned k) {
  return (unsigned)(gen_r(seed, frame, gr, ch, 600 + k) % 8);
}
PG_FN unsigned gen_scalefac_s(uint64_t seed, int64_t frame, unsigned gr, unsigned ch, unsigned k, unsigned w) {
  return (unsigned)(gen_r(seed, frame, gr, ch, 640 + k * 3 + w) % 8);
}

// One granule-channel; called by 64 lanes (lane = 0..63), lane l writes lines
// 9l..9l+8, lane 0 additionally writes the side record.
PG_FN void gen_gc(uint64_t seed, int64_t frame, unsigned gr, unsigned ch, int lane,
                  int16_t* spectra_gc, pdmp3_gc_side* side_gc) {
  const unsigned count1 = 2 * (240 + (unsigned)(gen_r(seed, frame, gr, ch, 576) % 49));
  for (int e = 0; e < 9; ++e) {
    const unsigned line = (unsigned)lane * 9 + e;
    int v = 0;
    if (line < count1) {
      const uint64_t q = gen_r(seed, frame, gr, ch, line);
      const unsigned A = 1 + 40 * (576 - line) / 576;
      const unsigned r2 = (unsigned)(q >> 16);
      const unsigned mag = ((q & 0xff) == 0) ? (r2 % 8207u) : (r2 % (A + 1));
      v = (q & 0x100) ? -(int)mag : (int)mag;
    }
    spectra_gc[line] = (int16_t)v;
  }
  if (lane != 0) return;
  pdmp3_gc_side s;
  unsigned char* raw = reinterpret_cast<unsigned char*>(&s);
  for (unsigned i = 0; i < sizeof s; ++i) raw[i] = 0;
  s.count1 = (uint16_t)count1;
  s.global_gain = (uint8_t)(130 + gen_r(seed, frame, gr, ch, 577) % 30);
  uint64_t r = gen_r(seed, frame, gr, ch, 578);
  unsigned flags = 0;
  if (r & 1) flags |= PDMP3_GC_SCALEFAC_SCALE;
  if (r & 2) flags |= PDMP3_GC_PREFLAG;
  const unsigned pct = (unsigned)((r >> 8) % 100);
  unsigned bt = 0, mixed = 0;
  if (pct < 85) bt = 0;
  else if (pct < 90) bt = 1;
  else if (pct < 95) { bt = 2; mixed = (unsigned)((r >> 20) & 1); }
  else bt = 3;
  if (bt != 0) flags |= PDMP3_GC_WIN_SWITCH;
  flags |= bt << PDMP3_GC_BLOCK_TYPE_SHIFT;
  if (mixed) flags |= PDMP3_GC_MIXED;
  s.flags = (uint8_t)flags;
  r = gen_r(seed, frame, gr, ch, 579);
  for (unsigned k = 0; k < 3; ++k) s.subblock_gain[k] = (uint8_t)((r >> (8 * k)) % 4);
  // 44.1 kHz, joint stereo, MS on, intensity off
  s.frame = (uint8_t)(0u | (1u << PDMP3_FR_MODE_SHIFT) | (2u << PDMP3_FR_MODEEXT_SHIFT));
  if (frame == 0) s.frame |= PDMP3_FR_RESET;
  for (unsigned k = 0; k < 21; ++k) s.scalefac_l[k] = (uint8_t)gen_scalefac_l(seed, frame, gr, ch, k);
  for (unsigned k = 0; k < 12; ++k)
    for (unsigned w = 0; w < 3; ++w) s.scalefac_s[k][w] = (uint8_t)gen_scalefac_s(seed, frame, gr, ch, k, w);
  // out-of-bounds scalefactor reads of the reference, by its memory layout
  // (SURVEY H4/H5): (g,0)->[g][1][0]; (0,1)->[1][0][0]; (1,1): l[21]->scalefac_s[0][0][0][0], s[12][w]->PEEK
  if (ch == 0) {
    s.scalefac_l[21] = (uint8_t)gen_scalefac_l(seed, frame, gr, 1, 0);
    for (unsigned w = 0; w < 3; ++w) s.scalefac_s[12][w] = (uint8_t)gen_scalefac_s(seed, frame, gr, 1, 0, w);
  } else if (gr == 0) {
    s.scalefac_l[21] = (uint8_t)gen_scalefac_l(seed, frame, 1, 0, 0);
    for (unsigned w = 0; w < 3; ++w) s.scalefac_s[12][w] = (uint8_t)gen_scalefac_s(seed, frame, 1, 0, 0, w);
  } else {
    s.scalefac_l[21] = (uint8_t)gen_scalefac_s(seed, frame, 0, 0, 0, 0);
    for (unsigned w = 0; w < 3; ++w) s.scalefac_s[12][w] = PDMP3_SF_PEEK;
  }
  *side_gc = s;
}

}  // namespace pdmp3

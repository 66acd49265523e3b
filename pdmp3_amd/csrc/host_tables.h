// host_tables.h -- host-side construction of the tables the device pipeline
// reads (decode_core.h: ConstBank, GlobalTables).
//
// Literal tables come from tables_data.h (data carried from pdmp3.c:572-870,
// 879-892, 2123); the libm-derived ones are generated here with the very
// expressions the reference uses, so they are bit-identical on any host with a
// correctly rounding libm (SURVEY appendix C lists their CRC-32s; the test
// suite checks them):
//   pow43[i]  = (float)pow((float)i, 4.0/3.0)            pdmp3.c:979
//   t1[n]     = (float)pow(2.0, -(0.5*n))                pdmp3.c:2127, 2144
//   t2[k]     = (float)pow(2.0, 0.25*(k-266))            pdmp3.c:2128, 2145-2146
#pragma once

#include <math.h>
#include <stdint.h>
#include <string.h>
#include <vector>

#include "decode_core.h"
#include "tables_data.h"

namespace pdmp3 {

struct HostTables {
  ConstBank cb;
  std::vector<float> pow43;        // 8207
  std::vector<uint8_t> band;       // 3*3*576
  std::vector<uint16_t> src_idx;   // 3*2*576
};

inline const uint16_t* sfb_long_of(int sfreq) { return sfreq == 0 ? kSfbLong0 : (sfreq == 1 ? kSfbLong1 : kSfbLong2); }
inline const uint16_t* sfb_short_of(int sfreq) { return sfreq == 0 ? kSfbShort0 : (sfreq == 1 ? kSfbShort1 : kSfbShort2); }

inline void build_host_tables(HostTables& H) {
  ConstBank& cb = H.cb;
  memset(&cb, 0, sizeof cb);
  for (int m = 0; m < 18; m++)
    for (int p = 0; p < 36; p++) cb.c36t[p][m] = kCosN36[m * 36 + p];
  for (int m = 0; m < 6; m++)
    for (int p = 0; p < 12; p++) cb.c12t[p][m] = kCosN12[m * 12 + p];
  for (int b = 0; b < 4; b++)
    for (int p = 0; p < 36; p++) cb.win[b][p] = kImdctWin[b * 36 + p];
  for (int i = 0; i < 8; i++) { cb.cs[i] = kAliasCs[i]; cb.ca[i] = kAliasCa[i]; }
  for (int i = 0; i < 16; i++) {
    // pdmp3.c:2166-2172; is_pos == 6 is special-cased, 7 means "off", >= 8 is
    // out of bounds in the reference (H3) and defined here as ratio 0.
    float t = (i < 6) ? kIsRatios[i] : 0.0f;
    if (i == 6) { cb.isr_l[i] = 1.0f; cb.isr_r[i] = 0.0f; }
    else { cb.isr_l[i] = t / (1.0f + t); cb.isr_r[i] = 1.0f / (1.0f + t); }
  }
  for (int i = 0; i < 512; i++) cb.dwin[i] = kSynthD[i];
  for (int n = 0; n < kT1Size; n++) cb.t1[n] = (float)pow(2.0, -(0.5 * n));
  for (int k = 0; k < kT2Size; k++) cb.t2[k] = (float)pow(2.0, 0.25 * (k - 266));
  for (int f = 0; f < 3; f++) {
    for (int i = 0; i < 23; i++) cb.sfb_l[f][i] = sfb_long_of(f)[i];
    for (int i = 0; i < 14; i++) cb.sfb_s[f][i] = sfb_short_of(f)[i];
  }
  for (int i = 0; i < 22; i++) cb.pretab[i] = kPretab[i];

  H.pow43.resize(8207);
  for (int i = 0; i < 8207; i++) H.pow43[i] = (float)pow((float)i, 4.0 / 3.0);

  H.band.assign(3 * 3 * 576, 0);
  H.src_idx.assign(3 * 2 * 576, 0);
  for (int f = 0; f < 3; f++) {
    const uint16_t* l = sfb_long_of(f);
    const uint16_t* s = sfb_short_of(f);
    uint8_t* bl = &H.band[(f * 3 + 0) * 576];
    uint8_t* bs = &H.band[(f * 3 + 1) * 576];
    uint8_t* bm = &H.band[(f * 3 + 2) * 576];
    for (int sfb = 0; sfb < 22; sfb++)
      for (int n = l[sfb]; n < l[sfb + 1]; n++) bl[n] = (uint8_t)sfb;
    uint16_t* ps = &H.src_idx[(f * 2 + 0) * 576];
    uint16_t* pm = &H.src_idx[(f * 2 + 1) * 576];
    for (int sfb = 0; sfb < 13; sfb++) {
      const int start = 3 * s[sfb], len = s[sfb + 1] - s[sfb];
      for (int win = 0; win < 3; win++)
        for (int j = 0; j < len; j++) {
          const int src = start + win * len + j;          // as Huffman-decoded: [win][j]
          const int dst = start + 3 * j + win;            // after L3_Reorder: [j][win]
          bs[src] = (uint8_t)(22 + sfb * 3 + win);
          ps[dst] = (uint16_t)src;
        }
    }
    for (int n = 0; n < 576; n++) {
      bm[n] = (n < 36) ? bl[n] : bs[n];                   // mixed: 2 long subbands, then short from sfb 3
      pm[n] = (n < 36) ? (uint16_t)n : ps[n];
    }
  }
}

}  // namespace pdmp3

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
#include "unpack_core.h"
#include "tables_data.h"
#include "lsf_tables.h"

namespace pdmp3 {

constexpr int kT1Size = 304;   // 2^(-n/2), n = 0..303 (0 from n = 300 on)
constexpr int kT2Size = 312;   // 2^((k-266)/4), k = 0..311

struct HostTables {
  ConstBank cb;
  std::vector<float> pow43;        // 8207
  std::vector<uint16_t> linetab;   // 3*3*576: source line | scale index << 10, per reordered line
  std::vector<float> win;          // 4*36
  std::vector<float> frag_long, frag_short, frag_mat;   // MFMA B fragments, [fragment][64 lanes]
  std::vector<float> taps;         // [16][64]: per-lane window coefficients (decode_core.h lane_init)
  std::vector<TabLds> tab_image;   // [3 sfreq]: what the kernels copy to LDS (decode_core.h tab_load_image)
  // the reference's libm expressions, kept to verify the device's ldexp forms
  // (decode_core.h: pow2_neg_half / pow2_quarter) over their whole index range
  std::vector<float> t1, t2;
  bool ldexp_forms_exact;
};

// sfreq 0..2: the reference's MPEG-1 tables; 3..8: MPEG-2 LSF / MPEG-2.5 (lsf_tables.h)
inline const uint16_t* sfb_long_of(int sfreq) { return sfreq >= 3 ? kLsfSfbLong[sfreq - 3] : sfreq == 0 ? kSfbLong0 : (sfreq == 1 ? kSfbLong1 : kSfbLong2); }
inline const uint16_t* sfb_short_of(int sfreq) { return sfreq >= 3 ? kLsfSfbShort[sfreq - 3] : sfreq == 0 ? kSfbShort0 : (sfreq == 1 ? kSfbShort1 : kSfbShort2); }
constexpr int kNumSfreq = 9;

inline void build_tab_images(HostTables& H);
inline void build_host_tables(HostTables& H) {
  ConstBank& cb = H.cb;
  memset(&cb, 0, sizeof cb);
  for (int i = 0; i < 8; i++) { cb.cs[i] = kAliasCs[i]; cb.ca[i] = kAliasCa[i]; }
  for (int i = 0; i < 16; i++) {
    // pdmp3.c:2166-2172; is_pos == 6 is special-cased, 7 means "off", >= 8 is
    // out of bounds in the reference (H3) and defined here as ratio 0.
    float t = (i < 6) ? kIsRatios[i] : 0.0f;
    if (i == 6) { cb.isr_l[i] = 1.0f; cb.isr_r[i] = 0.0f; }
    else { cb.isr_l[i] = t / (1.0f + t); cb.isr_r[i] = 1.0f / (1.0f + t); }
  }
  for (int i = 0; i < 512; i++) cb.dwin[i] = kSynthD[i];
  for (int f = 0; f < kNumSfreq; f++) {
    for (int i = 0; i < 23; i++) cb.sfb_l[f][i] = sfb_long_of(f)[i];
    for (int i = 0; i < 14; i++) cb.sfb_s[f][i] = sfb_short_of(f)[i];
  }
  // LSF intensity stereo (13818-3 2.4.3.2; the oracle evaluates the same expression): position p scales one channel by
  // i0^((p + 1) / 2), i0 = 2^(-1/4) (intensity_scale 0) or 2^(-1/2): p odd the left one, p even the right one
  for (int sc = 0; sc < 2; sc++)
    for (int pp = 0; pp < 32; pp++) {
      const float f = (float)pow(2.0, -(double)((sc + 1) * ((pp + 1) >> 1)) / 4.0);
      cb.isr_lsf_l[sc][pp] = (pp & 1) ? f : 1.0f;
      cb.isr_lsf_r[sc][pp] = (pp & 1) ? 1.0f : f;
    }
  H.win.assign(kImdctWin, kImdctWin + 144);
  H.t1.resize(kT1Size);
  H.t2.resize(kT2Size);
  H.ldexp_forms_exact = true;
  for (int n = 0; n < kT1Size; n++) {
    H.t1[n] = (float)pow(2.0, -(0.5 * n));
    if (f2u(H.t1[n]) != f2u(pow2_neg_half((uint32_t)n))) H.ldexp_forms_exact = false;
  }
  for (int k = 0; k < kT2Size; k++) {
    H.t2[k] = (float)pow(2.0, 0.25 * (k - 266));
    if (f2u(H.t2[k]) != f2u(pow2_quarter(k - 266))) H.ldexp_forms_exact = false;
  }

  H.pow43.resize(8207);
  for (int i = 0; i < 8207; i++) H.pow43[i] = (float)pow((float)i, 4.0 / 3.0);

  H.linetab.assign(kNumSfreq * 3 * 576, 0);
  for (int f = 0; f < kNumSfreq; f++) {
    const uint16_t* l = sfb_long_of(f);
    const uint16_t* s = sfb_short_of(f);
    uint16_t* tl = &H.linetab[(f * 3 + 0) * 576];
    uint16_t* ts = &H.linetab[(f * 3 + 1) * 576];
    uint16_t* tm = &H.linetab[(f * 3 + 2) * 576];
    for (int sfb = 0; sfb < 22; sfb++)
      for (int n = l[sfb]; n < l[sfb + 1]; n++) tl[n] = (uint16_t)(n | (sfb << 10));
    for (int sfb = 0; sfb < 13; sfb++) {
      const int start = 3 * s[sfb], len = s[sfb + 1] - s[sfb];
      for (int win = 0; win < 3; win++)
        for (int j = 0; j < len; j++) {
          const int src = start + win * len + j;          // as Huffman-decoded: [win][j]
          const int dst = start + 3 * j + win;            // after L3_Reorder (P:1786-1823): [j][win]
          ts[dst] = (uint16_t)(src | ((22 + sfb * 3 + win) << 10));
        }
    }
    for (int n = 0; n < 576; n++) tm[n] = (n < 36) ? tl[n] : ts[n];   // mixed: 2 long subbands, then short from sfb 3
  }

  // ---- MFMA B-operand fragments (decode_core.h: ph_mfma).  Lane l = (j = l & 15, kq = l >> 4)
  // holds B[k = 4 kk + kq][column map(nt, j)].
  auto col_of = [](int nt, int j) -> int { return nt == 0 ? j : 18 + j; };   // IMDCT output index p of tile column j
  // short transform as one 18 x 36 matrix, window folded in: out[6 w + p + 6] += in[w + 3 m] cos_N12[m][p] win[2][p]
  float bs[18][36];
  memset(bs, 0, sizeof bs);
  for (int w = 0; w < 3; w++)
    for (int m = 0; m < 6; m++)
      for (int p = 0; p < 12; p++) bs[w + 3 * m][6 * w + p + 6] = kCosN12[m * 12 + p] * kImdctWin[2 * 36 + p];
  H.frag_long.assign(10 * 64, 0.0f);
  H.frag_short.assign(10 * 64, 0.0f);
  for (int kk = 0; kk < 5; kk++)
    for (int nt = 0; nt < 2; nt++)
      for (int l = 0; l < 64; l++) {
        const int j = l & 15, k = 4 * kk + (l >> 4), p = col_of(nt, j);
        if (k < 18) {
          H.frag_long[(kk * 2 + nt) * 64 + l] = kCosN36[k * 36 + p];
          H.frag_short[(kk * 2 + nt) * 64 + l] = bs[k][p];
        }
      }
  // the four remaining IMDCT columns, computed on the VALU with scalar operands
  static const int xcol[4] = {16, 17, 34, 35};
  for (int q = 0; q < 4; q++)
    for (int m = 0; m < 18; m++) {
      PD_C36(&cb, c36x, q, m) = kCosN36[m * 36 + xcol[q]];
      PD_C36(&cb, s36x, q, m) = bs[m][xcol[q]];
    }
  // matrixing: C[n] = sum_sb s[sb] cos((2 sb + 1) n pi / 64), n = 0..31, taken from the reference's own matrix
  // N[i][sb] = (float)cos((float)((16 + i) (2 sb + 1)) * (pi / 64)) (pdmp3.c:1992):  C[n] = v[n - 16] (n >= 16),
  // C[n] = -v[48 - n] (n < 16); folded into the even / odd halves
  //   C[2m]   = sum_{k<16} (s[k] + s[31-k]) cos((2k+1) 2m pi/64),  C[2m+1] = sum_{k<16} (s[k] - s[31-k]) cos((2k+1)(2m+1) pi/64).
  // Fragment (eo, step r): lane (j, kq) holds the coefficient of k = 4 kq + r for output n = 2 j + eo.
  H.frag_mat.assign(8 * 64, 0.0f);
  for (int eo = 0; eo < 2; eo++)
    for (int r = 0; r < 4; r++)
      for (int l = 0; l < 64; l++) {
        const int j = l & 15, k = 4 * (l >> 4) + r, n = 2 * j + eo;
        const int i = n >= 16 ? n - 16 : 48 - n;
        const float nref = (float)cos(((float)(16 + i) * (2 * k + 1)) * (3.14159265358979323846 / 64.0));
        H.frag_mat[(eo * 4 + r) * 64 + l] = n >= 16 ? nref : -nref;
      }
  build_tab_images(H);
}


// the per-lane polyphase window coefficients and the LDS table images (after build_host_tables)
inline void build_tab_images(HostTables& H) {
  H.taps.assign(16 * 64, 0.0f);
  for (int lane = 0; lane < 64; lane++) {
    const int i = lane & 31;
    const float sgn_e = (i < 16) ? 1.0f : ((i == 16) ? 0.0f : -1.0f);
    for (int k = 0; k < 8; k++) {
      H.taps[k * 64 + lane] = sgn_e * kSynthD[64 * k + i];
      H.taps[(8 + k) * 64 + lane] = -kSynthD[64 * k + 32 + i];
    }
  }
  H.tab_image.resize(kNumSfreq);
  for (int sf = 0; sf < kNumSfreq; sf++) {
    TabLds& S = H.tab_image[sf];
    memset(&S, 0, sizeof S);
    for (int k = 0; k < 144; k++) (&S.win[0][0])[k] = H.win[k];
    for (int k = 0; k < 2 * kPow43Small; k++) {
      const int v = k - kPow43Small;
      const float p = H.pow43[v < 0 ? -v : v];
      S.pow43z[k] = v < 0 ? -p : p;
    }
    memcpy(&S.ltab[0][0], &H.linetab[(size_t)sf * 3 * 576], 3 * 576 * sizeof(uint16_t));
    for (int f = 0; f < 3; f++)
      for (int i = 0; i < 5; i++)
        for (int lane = 0; lane < 64; lane++)
          S.bandaddr[f][i][lane] = (uint8_t)((H.linetab[(size_t)f * 3 * 576 + fast_line(lane, i)] >> 10) << 2);
    S.sfreq = sf;
    for (int k = 0; k < 16; k++)
      for (int i = 0; i < 32; i++) S.taps[k][i] = H.taps[k * 64 + i];
  }
}

// Code books (tables_data.h: kHuffBooks, derived from the reference's tree arrays P:160-520) -> the two-level
// lookup of unpack_core.h: 8-bit first level whose every entry leads to a second-level table -- under a prefix that
// longer codes share, one just wide enough for the longest of them; for a code word of <= 8 bits its one leaf.
// Returns false if the blob does not fit.
inline bool build_unpack_tables(UnpackTables& U) {
  memset(&U, 0, sizeof U);
  for (int t = 0; t < 34; t++) { U.book_of_table[t] = (int8_t)kHuffBookOfTable[t]; U.linbits[t] = (uint8_t)kHuffLinbits[t]; }
  U.book_of_table[34] = PDMP3_HUFF_BOOK_ISO33;           // count1table_select = 2: the standard's table B (PDMP3_ISO_TABLE33, not the reference's H1)
  for (int i = 0; i < 32; i++) U.slen[i] = kSlen[i];
  for (int sf = 0; sf < 3; sf++) {
    for (int i = 0; i < 23; i++) U.sfb_l[sf][i] = sfb_long_of(sf)[i];
    U.sfb_l[sf][23] = sfb_short_of(sf)[0];
    for (int i = 0; i < 14; i++) U.sfb_s[sf][i] = sfb_short_of(sf)[i];
  }
  constexpr int HL = kHuffFirstBits;
  auto link = [](int sub_bits, uint32_t index) { return (uint32_t)(31 - sub_bits) | (index << 10); };   // (byte offset << 8)
  uint32_t n = 1;                                        // lut[0] = 0: the leaf of "no code word"
  for (int b = 0; b < PDMP3_NUM_HUFF_BOOKS; b++) {
    const pdmp3_hcode* codes = kHuffBooks[b];
    const int nc = kHuffBookSize[b];
    if (n + (1u << HL) > (uint32_t)kHuffLutMax) return false;
    U.book_base[b] = (uint16_t)n;
    uint32_t* first = U.lut + n;
    n += 1u << HL;
    for (int p = 0; p < (1 << HL); p++) first[p] = link(0, 0);
    int deepest[1 << HL];
    for (int p = 0; p < (1 << HL); p++) deepest[p] = 0;
    for (int i = 0; i < nc; i++)
      if (codes[i].len > HL) {
        const uint32_t p = codes[i].code >> (codes[i].len - HL);
        if (codes[i].len - HL > deepest[p]) deepest[p] = codes[i].len - HL;
      }
    uint32_t sub_of[1 << HL];
    for (int p = 0; p < (1 << HL); p++)
      if (deepest[p]) {
        if (n + (1u << deepest[p]) > (uint32_t)kHuffLutMax) return false;
        sub_of[p] = n;
        first[p] = link(deepest[p], n);
        n += 1u << deepest[p];
      }
    const bool quads = b == kHuffBookOfTable[32] || b == kHuffBookOfTable[33] || b == PDMP3_HUFF_BOOK_ISO33;
    for (int i = 0; i < nc; i++) {
      const int len = codes[i].len;
      const uint32_t val = codes[i].err ? 0 : codes[i].val;
      uint32_t nsign, nlin = 0;
      if (quads) nsign = (val & 1) + (val >> 1 & 1) + (val >> 2 & 1) + (val >> 3 & 1);
      else {
        nsign = ((val >> 4) != 0) + ((val & 15) != 0);
        nlin = ((val >> 4) == 15) + ((val & 15) == 15);
      }
      const uint32_t leaf = val | ((uint32_t)len << 8) | (((uint32_t)len + nsign) << 16) | (nlin << 24);
      if (len <= HL) {
        if (n + 1 > (uint32_t)kHuffLutMax) return false;
        U.lut[n] = leaf;
        const uint32_t base = codes[i].code << (HL - len);
        for (uint32_t k = 0; k < (1u << (HL - len)); k++) first[base + k] = link(0, n);
        n++;
      } else {
        const uint32_t p = codes[i].code >> (len - HL);
        const int sb = deepest[p], extra = len - HL;
        uint32_t* sub = U.lut + sub_of[p];
        const uint32_t base = (codes[i].code & ((1u << extra) - 1)) << (sb - extra);
        for (uint32_t k = 0; k < (1u << (sb - extra)); k++) sub[base + k] = leaf;
      }
    }
  }
  if (n + (1u << HL) > (uint32_t)kHuffLutMax) return false;
  static_assert(kZeroBook < 20 && kZeroBook >= PDMP3_NUM_HUFF_BOOKS, "the zero book takes a free slot of book_base");
  U.book_base[kZeroBook] = (uint16_t)n;
  for (int p = 0; p < (1 << HL); p++) U.lut[n + p] = link(0, 0);
  n += 1u << HL;
  U.n_lut = n;
  return true;
}


}  // namespace pdmp3

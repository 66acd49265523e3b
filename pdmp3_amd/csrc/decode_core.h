// decode_core.h -- the wave-level Layer-III transform pipeline for gfx950.
//
// One 64-lane wavefront owns one CHUNK of consecutive frames of a stream and
// walks it granule by granule; all inter-granule state lives on chip:
//   * IMDCT overlap (the reference's `store[2][32][18]`, pdmp3.c:1755) in the
//     VGPRs of the lane that owns (channel, subband),
//   * the polyphase history (the reference's 1024-float V FIFO per channel,
//     pdmp3.c:1983,2006) as 15 slots of 32 DCT coefficients per channel in LDS.
// Stages per granule (reference stage it replaces, "P:n" = pdmp3.c line n):
//   ph_load     global -> LDS copy of 2304 B int16 spectra + 2 side records
//   ph_scales   per-band requantisation scale 2^-(sfm*sf) * 2^(gain/4)   P:2127-2128, P:2144-2146
//   ph_requant  |is|^(4/3) * scale, short-block reorder as a gather,
//               MS / intensity stereo                                     P:1829, P:1786, P:1911
//   ph_imdct    alias reduction fused into the operand fetch, 18->36 (or
//               3 x 6->12) IMDCT with scalar-broadcast coefficients,
//               window, overlap-add, frequency inversion                  P:1706, P:1649, P:1752, P:1738
//   ph_dct32    32-point DCT-II (Lee) per time slot = the 64x32 matrixing
//               folded by its cosine symmetries                           P:2010-2014
//   ph_window   512-tap D window as 16 FMAs per sample against the slot
//               history, float -> int16 exactly as P:2028-2031
//   ph_store    coalesced PCM store, history shift
//
// The file is plain C++ that compiles for the device with hipcc AND for the
// host with g++ (tests/host_emul): the host build runs each phase for lanes
// 0..63 in turn and is used ONLY by the CPU test-suite to validate indexing
// without a GPU.  It is not a product path.
#pragma once

#include <stdint.h>
#include <string.h>
#include "../../include/pdmp3_hip.h"
#include "dct32_consts.h"

#if defined(__HIPCC__)
#define PD_FN __device__ __forceinline__
#define PD_MFN __device__ __forceinline__
#define PD_MUL(a, b) __fmul_rn((a), (b))
#define PD_ADD(a, b) __fadd_rn((a), (b))
#define PD_SUB(a, b) __fsub_rn((a), (b))
#define PD_FMA(a, b, c) __builtin_fmaf((a), (b), (c))
#define PD_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define PD_FN static inline
#define PD_MFN inline
#define PD_MUL(a, b) ((a) * (b))
#define PD_ADD(a, b) ((a) + (b))
#define PD_SUB(a, b) ((a) - (b))
#define PD_FMA(a, b, c) ((a) * (b) + (c))
#define PD_SCHED_FENCE() ((void)0)
#endif

namespace pdmp3 {

constexpr int kHaloGranules = 3;      // (f-2,gr1) (f-1,gr0) (f-1,gr1): SURVEY 8e incl. the H5 corner
constexpr int kHistSlots = 15;        // polyphase history depth (P:2015-2019 reaches 15 slots back)
constexpr int kT1Size = 304;          // 2^(-n/2), n = 0..303 (beyond: 0 as binary32)
constexpr int kT2Size = 312;          // 2^((k-266)/4), k = 0..311

// Small tables; lives in __constant__ memory on the device so that
// wave-uniform indices become scalar loads.
struct ConstBank {
  float c36t[36][18];   // cos_N36 (P:620-729) transposed to [p][m]
  float c12t[12][6];    // cos_N12 (P:606-619) transposed to [p][m]
  float win[4][36];     // g_imdct_win (P:577-603)
  float cs[8], ca[8];   // P:573-574
  float isr_l[16];      // is_ratio_l for is_pos 0..6 (P:2166-2172); [7] unused;
  float isr_r[16];      // [8..15]: the reference reads past is_ratios[] (H3) -> defined as t = 0
  float dwin[512];      // g_synth_dtbl (P:740-870)
  float t1[kT1Size];    // (float)pow(2.0, -0.5*n)   covers P:2127, P:2144
  float t2[kT2Size];    // (float)pow(2.0, 0.25*(k-266)) covers P:2128, P:2145
  uint16_t sfb_l[3][24];  // g_sf_band_indices[].l (P:879-892), padded
  uint16_t sfb_s[3][16];  // g_sf_band_indices[].s
  uint8_t pretab[24];   // P:2123 (+ [21] = 0, H4)
};

// Pointer to the bank.  On the device it is address-space-4 (constant) typed so
// that wave-uniform reads are scalar loads, and it is "laundered" through an
// empty asm once per granule so that the compiler cannot hoist the ~800
// loop-invariant scalar loads out of the granule loop (which spills hundreds of
// SGPRs into VGPR lanes).
#if defined(__HIPCC__)
typedef const __attribute__((address_space(4))) ConstBank* BankPtr;
#define PD_LAUNDER(p) asm volatile("" : "+s"(p))
#else
typedef const ConstBank* BankPtr;
#define PD_LAUNDER(p) ((void)0)
#endif

// Large tables in global memory.
struct GlobalTables {
  const float* pow43;       // [8207] (float)pow((float)i, 4.0/3.0), P:979
  const uint8_t* band;      // [3 sfreq][3 kind][576]: scale-table index of SOURCE line n;
                            //   kind 0 long: sfb; 1 short: 22+sfb*3+win; 2 mixed
  const uint16_t* src_idx;  // [3 sfreq][2 (short, mixed)][576]: reordered line d <- source line (P:1786-1823)
};

// LDS per wave (~7.9 KB).  Buffers whose lifetimes do not overlap share storage:
//   spec (ph_load .. ph_requant)            | pcm  (ph_window .. ph_store)
//   xr   (ph_requant .. ph_fetch)           | hyb  (ph_imdct .. ph_dct32) | vnew (ph_dct32 .. ph_window)
// hyb/vnew rows are [slot t][33]: the DCT lane that owns slot t transforms its row in place.
struct WaveLds {
  union {
    alignas(16) int16_t spec[2][576];
    alignas(16) int16_t pcm[1152];
  };
  alignas(16) uint8_t side[2][128];
  float scale[2][64];
  union {
    float xr[2][576];
    float hyb[2][18][33];
  };
  float peek[4];
};

struct LaneRegs {
  float ovl[18];     // IMDCT overlap of (ch = lane>>5, sb = lane&31): the reference's store[ch][sb][] (P:1755)
  float we[8];       // window coefficients of (ch, i = lane&31): even taps, sign folded
  float wo[8];       // odd taps
  float he[15];      // polyphase history: coefficient idx_e of the last 15 slots (oldest first)
  float ho[15];      // same for idx_o   (together = what the lane needs of v_vec[ch][], P:1983)
  float in[18];      // scratch: this granule's antialiased IMDCT input
  int idx_e, idx_o;  // which DCT coefficient the lane reads from even-/odd-aged slots
};

// Uniform per-granule facts, decoded from the side records in LDS.
struct GranuleInfo {
  int nch, sfreq, mode, mode_ext;
  int count1[2], flags[2];
  PD_MFN bool is_short(int ch) const {
    return (flags[ch] & PDMP3_GC_WIN_SWITCH) && ((flags[ch] & PDMP3_GC_BLOCK_TYPE_MASK) >> PDMP3_GC_BLOCK_TYPE_SHIFT) == 2;
  }
  PD_MFN bool is_mixed(int ch) const { return (flags[ch] & PDMP3_GC_MIXED) != 0; }
  PD_MFN int block_type(int ch) const { return (flags[ch] & PDMP3_GC_BLOCK_TYPE_MASK) >> PDMP3_GC_BLOCK_TYPE_SHIFT; }
};

PD_FN GranuleInfo granule_info(const WaveLds& L) {
  GranuleInfo g;
  int fr = L.side[0][7];
  g.sfreq = fr & PDMP3_FR_SFREQ_MASK;
  if (g.sfreq > 2) g.sfreq = 2;
  g.mode = (fr & PDMP3_FR_MODE_MASK) >> PDMP3_FR_MODE_SHIFT;
  g.mode_ext = (fr & PDMP3_FR_MODEEXT_MASK) >> PDMP3_FR_MODEEXT_SHIFT;
  g.nch = (g.mode == 3) ? 1 : 2;
  for (int ch = 0; ch < 2; ch++) {
    g.count1[ch] = L.side[ch][0] | (L.side[ch][1] << 8);
    g.flags[ch] = L.side[ch][3];
  }
  return g;
}

PD_FN uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

// ---------------------------------------------------------------------------
// chunk prologue: per-lane constants
// ---------------------------------------------------------------------------
PD_FN void state_zero(int lane, LaneRegs& R);
PD_FN void lane_init(int lane, LaneRegs& R, BankPtr cb) {
  const int i = lane & 31;
  // v[i] = C[16+i] (i<16), 0 (i==16), -C[48-i] (i>16);  v[32+i] = -C[16-i] (i<=16), -C[i-16] (i>16)
  // where C = 32-point DCT-II of the slot (derivation: DESIGN.md "polyphase").
  float sgn_e = (i < 16) ? 1.0f : ((i == 16) ? 0.0f : -1.0f);
  R.idx_e = (i < 16) ? 16 + i : ((i == 16) ? 0 : 48 - i);
  R.idx_o = (i <= 16) ? 16 - i : i - 16;
  for (int k = 0; k < 8; k++) {
    R.we[k] = sgn_e * cb->dwin[64 * k + i];
    R.wo[k] = -cb->dwin[64 * k + 32 + i];
  }
  state_zero(lane, R);
}

PD_FN void state_zero(int lane, LaneRegs& R) {
  (void)lane;
  for (int m = 0; m < 18; m++) R.ovl[m] = 0.0f;
  for (int s = 0; s < kHistSlots; s++) { R.he[s] = 0.0f; R.ho[s] = 0.0f; }
}

// state layout (opaque to callers): float ovl[64 lanes][18]; float he[64][15]; float ho[64][15]
constexpr int kStateFloats = 64 * (18 + 2 * kHistSlots);

PD_FN void state_load(int lane, LaneRegs& R, const float* st) {
  for (int m = 0; m < 18; m++) R.ovl[m] = st[m * 64 + lane];
  for (int s = 0; s < kHistSlots; s++) {
    R.he[s] = st[(18 + s) * 64 + lane];
    R.ho[s] = st[(18 + kHistSlots + s) * 64 + lane];
  }
}

PD_FN void state_store(int lane, const LaneRegs& R, float* st) {
  for (int m = 0; m < 18; m++) st[m * 64 + lane] = R.ovl[m];
  for (int s = 0; s < kHistSlots; s++) {
    st[(18 + s) * 64 + lane] = R.he[s];
    st[(18 + kHistSlots + s) * 64 + lane] = R.ho[s];
  }
}

// ---------------------------------------------------------------------------
// ph_load: 2304 B spectra + 256 B side, 16 B per lane per access
// ---------------------------------------------------------------------------
struct Chunk16 { uint32_t x, y, z, w; };

PD_FN void ph_load(int lane, WaveLds& L, const int16_t* spectra_g, const pdmp3_gc_side* side_g) {
  const Chunk16* src = reinterpret_cast<const Chunk16*>(spectra_g);
  Chunk16* dst = reinterpret_cast<Chunk16*>(&L.spec[0][0]);
  dst[lane] = src[lane];
  dst[lane + 64] = src[lane + 64];
  if (lane < 16) {
    dst[lane + 128] = src[lane + 128];
    reinterpret_cast<Chunk16*>(&L.side[0][0])[lane] = reinterpret_cast<const Chunk16*>(side_g)[lane];
  }
}

// ---------------------------------------------------------------------------
// ph_scales: scale[ch][e], e = 0..21 long sfb, 22+sfb*3+win short
// ---------------------------------------------------------------------------
PD_FN void ph_scales(int lane, WaveLds& L, BankPtr cb) {
  if (lane >= 61) return;
  for (int ch = 0; ch < 2; ch++) {
    const uint8_t* s = L.side[ch];
    const int gg = s[2], flags = s[3];
    const bool sfscale = flags & PDMP3_GC_SCALEFAC_SCALE;
    float t1, t2;
    if (lane < 22) {
      int x = s[8 + lane] + ((flags & PDMP3_GC_PREFLAG) ? cb->pretab[lane] : 0);
      int n = sfscale ? 2 * x : x;
      t1 = cb->t1[n];
      t2 = cb->t2[gg + 56];
    } else {
      const int q = lane - 22, sfb = q / 3, win = q - 3 * sfb;
      uint32_t sf = s[30 + q];
      if (sfb == 12 && s[30 + 36] == PDMP3_SF_PEEK) sf = f2u(L.peek[win]);   // H5
      if (sf > 400u) sf = 400u;
      uint32_t n = sfscale ? 2 * sf : sf;
      t1 = (n < (uint32_t)kT1Size) ? cb->t1[n] : 0.0f;
      t2 = cb->t2[gg + 56 - 8 * s[4 + win]];
    }
    L.scale[ch][lane] = PD_MUL(t1, t2);
  }
}

// ---------------------------------------------------------------------------
// ph_requant: requantise + reorder (gather) + stereo, in reordered line order
// ---------------------------------------------------------------------------
template <bool DUMP>
PD_FN void ph_requant(int lane, WaveLds& L, BankPtr cb, const GlobalTables& T,
                      float* dump0, float* dump1) {
  const GranuleInfo g = granule_info(L);
  const bool joint = (g.nch == 2) && (g.mode == 1) && (g.mode_ext != 0);
  const bool ms = joint && (g.mode_ext & 2);
  const bool is = joint && (g.mode_ext & 1);
  const int cmin = (g.count1[0] > g.count1[1]) ? g.count1[1] : g.count1[0];   // P:1920 (H2)
  int kind[2];
  for (int ch = 0; ch < 2; ch++) kind[ch] = g.is_short(ch) ? (g.is_mixed(ch) ? 2 : 1) : 0;
  const uint8_t* bandL = T.band + (g.sfreq * 3 + 0) * 576;
  const uint8_t* bandS = T.band + (g.sfreq * 3 + 1) * 576;

  for (int r = 0; r < 2; r++) {
    const int base = 512 * r + 8 * lane;
    if (base >= 576) break;
    for (int e = 0; e < 8; e++) {
      const int d = base + e;
      float x[2] = {0.0f, 0.0f};
      for (int ch = 0; ch < g.nch; ch++) {
        const int k = kind[ch];
        const int sidx = (k == 0) ? d : T.src_idx[(g.sfreq * 2 + (k - 1)) * 576 + d];
        const int v = L.spec[ch][sidx];
        const int a = v < 0 ? -v : v;
        const float p = T.pow43[a > 8206 ? 8206 : a];
        const float t3 = v < 0 ? -p : p;
        const int b = T.band[(g.sfreq * 3 + k) * 576 + sidx];
        x[ch] = PD_MUL(L.scale[ch][b], t3);
      }
      if (DUMP) {
        for (int ch = 0; ch < g.nch; ch++) dump0[ch * 4 * 576 + d] = x[ch];
      }
      if (ms && d < cmin) {   // P:1921-1928
        const float sum = PD_ADD(x[0], x[1]), dif = PD_SUB(x[0], x[1]);
        x[0] = (float)((double)sum * 0.70710678118654752440);
        x[1] = (float)((double)dif * 0.70710678118654752440);
      }
      if (is) {               // P:1932-1971; block shape taken from channel 0
        const uint8_t* s0 = L.side[0];
        const int c1 = g.count1[1];
        bool do_long = false, do_short = false;
        int sfb = 0, win = 0;
        if (kind[0] == 0) {
          sfb = bandL[d];
          do_long = (sfb < 21);
        } else if (kind[0] == 2 && d < 36) {
          sfb = bandL[d];
          do_long = (sfb < 8);
        } else {
          const int q = bandS[d] - 22;
          sfb = q / 3; win = q - 3 * sfb;
          do_short = (sfb < 12) && (kind[0] == 1 || sfb >= 3);
        }
        if (do_long && (int)cb->sfb_l[g.sfreq][sfb] >= c1) {
          const int is_pos = s0[8 + sfb];
          if (is_pos != 7) {
            const float l = PD_MUL(cb->isr_l[is_pos & 15], x[0]);
            const float rr = PD_MUL(cb->isr_r[is_pos & 15], x[0]);
            x[0] = l;
            x[1] = rr;
          }
        }
        if (do_short && 3 * (int)cb->sfb_s[g.sfreq][sfb] >= c1) {
          const int is_pos = s0[30 + sfb * 3 + win];
          if (is_pos != 7) {   // H3: sample forced through `unsigned` (x86-64 conversion semantics)
            const float xv = x[0];
            long long t = (xv >= 9.2233720368547758e18f || xv < -9.2233720368547758e18f || xv != xv)
                              ? (long long)0x8000000000000000ull : (long long)xv;
            const float vv = (float)(uint32_t)(unsigned long long)t;
            x[0] = vv; x[1] = vv;
          }
        }
      }
      for (int ch = 0; ch < g.nch; ch++) L.xr[ch][d] = x[ch];
      if (DUMP) {
        for (int ch = 0; ch < g.nch; ch++) dump1[ch * 4 * 576 + d] = x[ch];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// ph_imdct: antialias (fused) + IMDCT + window + overlap-add + freq inversion
// ---------------------------------------------------------------------------
template <bool DUMP>
PD_FN void ph_fetch(int lane, const WaveLds& L, LaneRegs& R, BankPtr cb, float* dump2) {
  const GranuleInfo g = granule_info(L);
  const int ch = lane >> 5, sb = lane & 31;
  if (ch >= g.nch) return;
  const bool shrt = g.is_short(ch), mixed = g.is_mixed(ch);
  const float* x = L.xr[ch];
  float* in = R.in;
  for (int m = 0; m < 18; m++) in[m] = x[18 * sb + m];
  // P:1706-1732: butterflies across the boundary below (index sb) and above (sb+1)
  const bool aa_lo = (sb >= 1) && (!shrt || (mixed && sb == 1));
  const bool aa_hi = (sb <= 30) && (!shrt || (mixed && sb == 0));
  if (aa_lo) {
    for (int i = 0; i < 8; i++) {
      const float lo = x[18 * sb - 1 - i];
      in[i] = PD_ADD(PD_MUL(in[i], cb->cs[i]), PD_MUL(lo, cb->ca[i]));           // ub, P:1726
    }
  }
  if (aa_hi) {
    for (int i = 0; i < 8; i++) {
      const float up = x[18 * (sb + 1) + i];
      in[17 - i] = PD_SUB(PD_MUL(in[17 - i], cb->cs[i]), PD_MUL(up, cb->ca[i]));  // lb, P:1725
    }
  }
  if (DUMP) {
    for (int m = 0; m < 18; m++) dump2[ch * 4 * 576 + 18 * sb + m] = in[m];
  }
}

template <bool DUMP>
PD_FN void ph_imdct(int lane, WaveLds& L, LaneRegs& R, BankPtr cb, float* dump3) {
  const GranuleInfo g = granule_info(L);
  const int ch = lane >> 5, sb = lane & 31;
  if (ch >= g.nch) return;
  const bool mixed = g.is_mixed(ch);
  const bool wsf = (g.flags[ch] & PDMP3_GC_WIN_SWITCH) != 0;
  const float* in = R.in;
  const int bt = (wsf && mixed && sb < 2) ? 0 : g.block_type(ch);   // P:1769-1771
  const bool odd_sb = sb & 1;
  float res[18];
  if (bt != 2) {
    // out[p] = (sum_m in[m] cos_N36[m][p]) * win[bt][p]  (P:1689-1698); the 18
    // coefficients of one p are wave-uniform => scalar operands.
    for (int p = 0; p < 18; p++) {
      float sa = 0.0f, sb2 = 0.0f;
      for (int m = 0; m < 18; m++) {
        sa = PD_FMA(in[m], cb->c36t[p][m], sa);
        sb2 = PD_FMA(in[m], cb->c36t[p + 18][m], sb2);
      }
      const float w0 = (bt == 0) ? cb->win[0][p] : ((bt == 1) ? cb->win[1][p] : cb->win[3][p]);
      const float w1 = (bt == 0) ? cb->win[0][p + 18] : ((bt == 1) ? cb->win[1][p + 18] : cb->win[3][p + 18]);
      res[p] = sa * w0 + R.ovl[p];                                             // P:1775
      R.ovl[p] = sb2 * w1;                                                     // P:1776
      if ((p & 1) == 1) PD_SCHED_FENCE();
    }
  } else {
    float raw[36];
    for (int p = 0; p < 36; p++) raw[p] = 0.0f;
    for (int wn = 0; wn < 3; wn++)                                             // P:1675-1685
      for (int p = 0; p < 12; p++) {
        float sum = 0.0f;
        for (int m = 0; m < 6; m++) sum = PD_FMA(in[wn + 3 * m], cb->c12t[p][m], sum);
        raw[6 * wn + p + 6] += sum * cb->win[2][p];
      }
    for (int p = 0; p < 18; p++) { res[p] = raw[p] + R.ovl[p]; R.ovl[p] = raw[p + 18]; }
  }
  for (int p = 0; p < 18; p++) {
    float y = res[p];
    if (odd_sb && (p & 1)) y = -y;                                             // P:1738-1746
    L.hyb[ch][p][sb] = y;
    if (DUMP) dump3[ch * 4 * 576 + 18 * sb + p] = y;
    if (lane == 0 && p < 3) L.peek[p] = y;                                     // H5 source
  }
}

// ---------------------------------------------------------------------------
// 32-point DCT-II, Lee's recursion:  X[k] = sum_n x[n] cos(pi (2n+1) k / 2N)
// ---------------------------------------------------------------------------
template <int N> struct LeeC;
template <> struct LeeC<32> { static constexpr float v[16] = PDMP3_LEE32; };
template <> struct LeeC<16> { static constexpr float v[8] = PDMP3_LEE16; };
template <> struct LeeC<8> { static constexpr float v[4] = PDMP3_LEE8; };
template <> struct LeeC<4> { static constexpr float v[2] = PDMP3_LEE4; };
template <> struct LeeC<2> { static constexpr float v[1] = PDMP3_LEE2; };

template <int N>
PD_FN void dct2_lee(const float* in, float* out) {
  if constexpr (N == 1) {
    out[0] = in[0];
  } else {
    float a[N / 2], b[N / 2], A[N / 2], B[N / 2];
    for (int n = 0; n < N / 2; n++) {
      a[n] = in[n] + in[N - 1 - n];
      b[n] = (in[n] - in[N - 1 - n]) * LeeC<N>::v[n];
    }
    dct2_lee<N / 2>(a, A);
    dct2_lee<N / 2>(b, B);
    for (int k = 0; k < N / 2; k++) {
      out[2 * k] = A[k];
      out[2 * k + 1] = (k + 1 < N / 2) ? B[k] + B[k + 1] : B[k];
    }
  }
}

PD_FN void ph_dct32(int lane, WaveLds& L) {
  const GranuleInfo g = granule_info(L);
  if (lane >= 18 * g.nch) return;
  const int ch = lane / 18, t = lane - 18 * ch;
  float x[32], c[32];
  for (int j = 0; j < 32; j++) x[j] = L.hyb[ch][t][j];
  dct2_lee<32>(x, c);
  for (int n = 0; n < 32; n++) L.hyb[ch][t][n] = c[n];   // in place: the row now holds the slot's C[0..31]
}

// float -> int16 exactly as P:2028-2031 on x86-64 (cvttsd2si: out of range => INT32_MIN)
PD_FN int pcm_from_sum(float sum) {
  const double d = (double)sum * 32767.0;
  int s;
  if (!(d > -2147483649.0 && d < 2147483648.0)) s = (int)0x80000000;
  else s = (int)d;
  if (s > 32767) s = 32767;
  else if (s < -32767) s = -32767;
  return s;
}

PD_FN void ph_window(int lane, WaveLds& L, LaneRegs& R) {
  const GranuleInfo g = granule_info(L);
  const int ch = lane >> 5, i = lane & 31;
  if (ch >= g.nch) return;
  // E[s], O[s]: the lane's two coefficients of slot s; s = 0..14 history, 15..32 this granule
  float E[kHistSlots + 18], O[kHistSlots + 18];
  for (int s = 0; s < kHistSlots; s++) { E[s] = R.he[s]; O[s] = R.ho[s]; }
  for (int t = 0; t < 18; t++) {
    E[kHistSlots + t] = L.hyb[ch][t][R.idx_e];
    O[kHistSlots + t] = L.hyb[ch][t][R.idx_o];
  }
  int16_t out[18];
  for (int t = 0; t < 18; t++) {
    float sum = 0.0f;
    for (int k = 0; k < 8; k++) {     // P:2021-2026: u[32j+i], j = 2k (age 2k), 2k+1 (age 2k+1)
      sum = PD_FMA(R.we[k], E[kHistSlots + t - 2 * k], sum);
      sum = PD_FMA(R.wo[k], O[kHistSlots + t - 2 * k - 1], sum);
    }
    out[t] = (int16_t)pcm_from_sum(sum);
  }
  for (int s = 0; s < kHistSlots; s++) { R.he[s] = E[18 + s]; R.ho[s] = O[18 + s]; }
  // pcm aliases spec, which is dead since ph_requant; hyb reads above are done
  for (int t = 0; t < 18; t++) L.pcm[(t * 32 + i) * g.nch + ch] = out[t];
}

// PCM: granule = 576 sample-frames = 1152*nch bytes
PD_FN void ph_store(int lane, WaveLds& L, int16_t* pcm_g, bool emit) {
  const GranuleInfo g = granule_info(L);
  if (emit) {
    const int n16 = (576 * 2 * g.nch) / 16;   // 144 or 72 chunks of 16 B
    const Chunk16* src = reinterpret_cast<const Chunk16*>(L.pcm);
    Chunk16* dst = reinterpret_cast<Chunk16*>(pcm_g);
    for (int c = lane; c < n16; c += 64) dst[c] = src[c];
  }
}

// ---------------------------------------------------------------------------
// One chunk = one wavefront.  PD_PHASE runs its body for every lane and then
// synchronises: on the device `lane` is threadIdx.x and the barrier is a
// (single-wave) __syncthreads(); the host test build loops lanes 0..63.
// ---------------------------------------------------------------------------
struct DecodeArgs {
  const int16_t* spectra;        // [n_frames][2][2][576]
  const pdmp3_gc_side* side;     // [n_frames][2][2]
  int16_t* pcm;                  // 2304 int16 per frame
  const float* state_in;         // kStateFloats or null (read by chunk 0)
  float* state_out;              // kStateFloats or null (written by the last chunk; must not alias state_in
                                 // when there is more than one chunk)
  float* stages;                 // [n_frames][2][2][4][576] or null (DUMP builds)
  int n_frames;
  int chunk_frames;
};

#if defined(__HIPCC__)
#define PD_NLANES 1
#define PD_PHASE(...)                                   \
  {                                                     \
    const int lane = threadIdx.x;                       \
    LaneRegs& R = Rs[0];                                \
    (void)R; (void)lane;                                \
    __VA_ARGS__;                                        \
  }                                                     \
  __syncthreads();
#else
#define PD_NLANES 64
#define PD_PHASE(...)                                   \
  for (int lane = 0; lane < 64; ++lane) {               \
    LaneRegs& R = Rs[lane];                             \
    (void)R;                                            \
    __VA_ARGS__;                                        \
  }
#endif

template <bool DUMP>
PD_FN void run_chunk(const DecodeArgs& a, const GlobalTables& T, BankPtr cb, int chunk, WaveLds& L) {
  LaneRegs Rs[PD_NLANES];
  const int f0 = chunk * a.chunk_frames;
  int f1 = f0 + a.chunk_frames;
  if (f1 > a.n_frames) f1 = a.n_frames;
  const int g_begin = 2 * f0, g_end = 2 * f1;
  const int g_start = (chunk == 0) ? 0 : g_begin - kHaloGranules;
  const bool last = (f1 == a.n_frames);

  PD_PHASE(
    lane_init(lane, R, cb);
    if (chunk == 0 && a.state_in) state_load(lane, R, a.state_in);
    if (lane < 4) L.peek[lane] = 1.0f;
  )
  for (int g = g_start; g < g_end; ++g) {
    const int f = g >> 1, gr = g & 1;
    PD_LAUNDER(cb);
    PD_PHASE(ph_load(lane, L, a.spectra + (size_t)g * 1152, a.side + (size_t)g * 2))
    PD_PHASE(
      if (gr == 0 && (L.side[0][7] & PDMP3_FR_RESET)) state_zero(lane, R);
      ph_scales(lane, L, cb);
    )
    float* dmp = DUMP ? a.stages + ((size_t)f * 16 + gr * 8) * 576 : nullptr;
    PD_PHASE(ph_requant<DUMP>(lane, L, cb, T, dmp, dmp + 576))
    PD_PHASE(ph_fetch<DUMP>(lane, L, R, cb, dmp + 2 * 576))
    PD_PHASE(ph_imdct<DUMP>(lane, L, R, cb, dmp + 3 * 576))
    PD_PHASE(ph_dct32(lane, L))
    PD_PHASE(ph_window(lane, L, R))
    PD_PHASE(
      const int nch = ((L.side[0][7] & PDMP3_FR_MODE_MASK) >> PDMP3_FR_MODE_SHIFT) == 3 ? 1 : 2;
      ph_store(lane, L, a.pcm + (size_t)f * 2304 + gr * 576 * nch, g >= g_begin);
    )
  }
  if (last && a.state_out) {
    PD_PHASE(state_store(lane, R, a.state_out))
  }
}

}  // namespace pdmp3

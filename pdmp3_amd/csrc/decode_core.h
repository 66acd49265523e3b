// decode_core.h -- the wave-level Layer-III transform pipeline for gfx950.
//
// One 64-lane wavefront owns one CHUNK of consecutive frames of a stream and
// walks it granule by granule; all inter-granule state lives on chip:
//   * IMDCT overlap (the reference's `store[2][32][18]`, pdmp3.c:1755) in VGPRs, in the C/D register layout of
//     the matrix instruction that produces it,
//   * the polyphase history (the reference's 1024-float V FIFO per channel,
//     pdmp3.c:1983,2006) as the 2 x 15 DCT coefficients each lane will need
//     again, also in VGPRs.
// Stages per granule (reference stage it replaces, "P:n" = pdmp3.c line n):
//   ph_commit    prefetched 2304 B int16 spectra + 2 side records -> LDS
//   ph_scales    per-band requantisation scale 2^-(sfm*sf) * 2^(gain/4)   P:2127-2128, P:2144-2146
//   ph_requant   |is|^(4/3) * scale, short-block reorder as a gather,
//                MS / intensity stereo                                     P:1829, P:1786, P:1911
//   ph_antialias alias reduction in place                                  P:1706
//   ph_mfma      18->36 (or 3 x 6->12) IMDCT, window, overlap-add, frequency inversion, and the 64x32 matrixing
//                folded to 16x16 DCT halves -- all on v_mfma_f32_16x16x4_f32   P:1649, P:1752, P:1738, P:2010-2014
//   ph_window    512-tap D window as 16 FMAs per sample against the slot
//                history, float -> int16 exactly as P:2028-2031
//   ph_store     PCM store
//
// ONE formulation, two builds: hipcc compiles this file for the device; g++ compiles THE SAME code for the
// CPU test-suite (tests/host_emul), where the 64 lanes of a wave run as 64 cooperatively scheduled fibers and
// the cross-lane operations (matrix instruction, lane shuffles, ballots, phase fences) are provided by the
// test harness through the `pdmp3::emu` hooks declared below -- with the device's fragment layouts and the
// device's arithmetic (k-ordered fmaf chains).  The host build is test infrastructure, not a product path.
#pragma once

#include <math.h>
#include <stdint.h>
#include <string.h>
#include "../../include/pdmp3_hip.h"

namespace pdmp3 {
struct ConstBank;
}

// ---------------------------------------------------------------------------
// THE ONE PLACE where the device build (hipcc) and the host test build (g++, tests/host_emul) differ: function
// attributes, the cross-lane primitives, the matrix instruction, vector types, the address-space-typed pointers and the
// three spots where the device form is an instruction the host has to spell out (RTZ multiplies, v_med3, readfirstlane of
// a pointer).  Everything below this block is one text for both.
// ---------------------------------------------------------------------------
#if defined(__HIPCC__)
#define PD_FN __device__ __forceinline__
#define PD_SLOW_FN __device__ __attribute__((noinline))      /* a real call: its registers and spills are its own */
#define PD_MFN __device__ __forceinline__
#define PD_HD __host__ __device__ __forceinline__
// Built with -ffp-contract=off: a*b+c stays two roundings (as in the reference's
// x86-64 build) unless PD_FMA is written out.
#define PD_FMA(a, b, c) __builtin_fmaf((a), (b), (c))
#define PD_CLOCK() __builtin_amdgcn_s_memtime()
#define PD_UNROLL _Pragma("unroll")
#define PD_UNROLL_N(n) _Pragma(PD_STR_(unroll n))
#define PD_STR_(x) #x
#define PD_NOUNROLL _Pragma("nounroll")
#define PD_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// pins a VGPR value at this program point (keeps IR passes from sinking the
// arithmetic that produced it past the scheduling fences)
#define PD_PIN(x) asm volatile("" : "+v"(x))
#define PD_ANY(c) (__builtin_amdgcn_ballot_w64(c) != 0ull)     // wave-uniform: any lane
#define PD_LANE() ((int)(threadIdx.x & 63))
// a value every lane holds alike, moved to a scalar register: conditions on it become scalar branches instead of
// exec-masked regions, and loads under them can be hoisted
#define PD_UNIFORM(x) __builtin_amdgcn_readfirstlane((int)(x))
#define PD_SHFL_XOR(v, m) __shfl_xor((v), (m))
// A workgroup's waves never share a WaveLds: LDS operations of a wave execute in order, so the
// phase hand-offs through LDS need no s_barrier and -- unlike __syncthreads() -- no
// drain of the vector-memory counter: the next granule's prefetch loads and this
// granule's PCM stores stay in flight across phases.  The wave barrier only stops
// the compiler from moving memory operations across the phase boundary.
#define PD_WAVE_SYNC()                                       \
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
  __builtin_amdgcn_wave_barrier();                          \
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront")
// granule kernel: the last wave of a workgroup hands its granule's state to the first wave of the next workgroup through
// global memory.  The two may sit on different XCDs, whose L2s are not coherent for ordinary accesses.  Device-scope
// release / acquire FENCES write back / invalidate a whole L2 here (tried in round 2: a 37 us launch went to 98 us), so
// the state travels in device-scope relaxed ATOMIC accesses (sc1: written through, read past the reader's L1), ordered
// against the flag by the wave's own vector-memory counter; nothing else is flushed.  The waiting wave sleeps between polls.
#define PD_STORE_DEVICE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define PD_LOAD_DEVICE(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define PD_VMEM_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define PD_SLEEP() __builtin_amdgcn_s_sleep(4)
#define PD_SETPRIO(x) __builtin_amdgcn_s_setprio(x)
// a flag word that is KNOWN to be in LDS, read / written as such (through a struct handed to a called function the
// compiler loses the address space and falls back to flat accesses, which wait for every counter)
#define PD_LDS_FLAG(p) (*(volatile __attribute__((address_space(3))) unsigned*)(p))
#define PD_BALLOT(c) ((unsigned long long)__builtin_amdgcn_ballot_w64(c))
// Pointer to the constant bank: address-space-4 (constant) typed so that wave-uniform reads are scalar loads, and
// "laundered" through an empty asm once per granule so that the compiler cannot hoist the ~800 loop-invariant scalar
// loads out of the granule loop (which spills hundreds of SGPRs into VGPR lanes).  PD_LAUNDER does the same for any
// scalar-register value.
namespace pdmp3 {
typedef const __attribute__((address_space(4))) ConstBank* BankPtr;
}
#define PD_LAUNDER(p) asm volatile("" : "+s"(p))
// a pointer argument of a CALLED function arrives in vector registers: back into scalar ones
#define PD_UNIFORM_PTR(T, p) ((T)(((unsigned long long)(unsigned)PD_UNIFORM((int)((unsigned long long)(p) >> 32)) << 32) | (unsigned)PD_UNIFORM((int)(unsigned long long)(p))))
// index into a table whose out-of-range reads are harmless on the device (somewhere in LDS) and replaced by the caller
#define PD_UNCHECKED_INDEX(v, lo, n) (v)
#define PD_FMED3(x, lo, hi) __builtin_amdgcn_fmed3f((x), (lo), (hi))
namespace pdmp3 {
// two / four binary32 values in consecutive registers: element-wise + - * on the pairs are the packed VALU forms
// (v_pk_add_f32, v_pk_mul_f32: one issue slot for two results; each element rounded exactly like the scalar operation)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// element-wise fused multiply-add of two pairs: v_pk_fma_f32 (each element = fmaf)
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
// v_mfma_f32_16x16x4_f32 (layout: "MFMA formulation" below)
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// v_permlane32_swap_b32: lanes 32..63 of a exchange with lanes 0..31 of b
__device__ __forceinline__ void permlane32_swap(int& a, int& b) {
  const auto r = __builtin_amdgcn_permlane32_swap((unsigned)a, (unsigned)b, false, false);
  a = (int)r[0];
  b = (int)r[1];
}
// p[t] = x[t] * 32767 rounded TOWARD ZERO, t = 0..17: the FP32 rounding mode is switched to RTZ and back inside one asm
// statement (two of them, nine products each: an asm statement takes 30 operands), the compiler never sees another mode
// (see pcm_convert18).  Out of place -- the products land in registers of their own (early-clobber), so that the compiler
// is free to let the sums die where they were accumulated: in place it copied all eighteen first.
__device__ __forceinline__ void mul9_rtz_32767(const float* x, float* p) {
#define PD_M(n, m) "v_mul_f32 %" #n ", 0x46fffe00, %" #m "\n\t"
  asm volatile(
      "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
      "s_nop 1\n\t"
      PD_M(0, 9) PD_M(1, 10) PD_M(2, 11) PD_M(3, 12) PD_M(4, 13) PD_M(5, 14) PD_M(6, 15) PD_M(7, 16) PD_M(8, 17)
      "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\t"
      "s_nop 1"
      : "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5]), "=&v"(p[6]), "=&v"(p[7]), "=&v"(p[8])
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]));
#undef PD_M
}
__device__ __forceinline__ void mul18_rtz_32767(const float* x, float* p) {
  mul9_rtz_32767(x, p);
  mul9_rtz_32767(x + 9, p + 9);
}
}  // namespace pdmp3
#else
namespace pdmp3 {
namespace emu {   // provided by tests/host_emul/wave_emul.h: one wave = 64 fibers
int lane();                                   // the calling fiber's lane
void wave_sync();                             // returns when all 64 lanes have arrived
void wave_yield();                            // all 64 lanes arrive, the other live waves run
float shfl_xor(float v, int mask);
bool any(bool c);
unsigned long long ballot(bool c);
void mfma16(float a, float b, float* cd);     // v_mfma_f32_16x16x4_f32: cd[4] in and out
void permlane32_swap(int* a, int* b);         // v_permlane32_swap_b32 vdst = a, src = b
}  // namespace emu
}  // namespace pdmp3
#define PD_FN static inline
#define PD_SLOW_FN static
#define PD_MFN inline
#define PD_HD static inline
#define PD_FMA(a, b, c) __builtin_fmaf((a), (b), (c))   /* a true fused multiply-add, as on the device */
#define PD_CLOCK() 0ull
#define PD_UNROLL
#define PD_UNROLL_N(n)
#define PD_NOUNROLL
#define PD_SCHED_FENCE() ((void)0)
#define PD_PIN(x) ((void)0)
#define PD_ANY(c) (::pdmp3::emu::any(c))
#define PD_LANE() (::pdmp3::emu::lane())
#define PD_UNIFORM(x) ((int)(x))
#define PD_SHFL_XOR(v, m) (::pdmp3::emu::shfl_xor((v), (m)))
#define PD_WAVE_SYNC() ::pdmp3::emu::wave_sync()
// (the host build runs the waves one after the other in frame order: a flag is always set before it is waited for)
#define PD_STORE_DEVICE(p, v) (*(p) = (v))
#define PD_LOAD_DEVICE(p) (*(p))
#define PD_VMEM_DRAIN() ::pdmp3::emu::wave_sync()
#define PD_SLEEP() ::pdmp3::emu::wave_yield()    /* the other live waves of the workgroup run (a lone wave comes straight back) */
#define PD_SETPRIO(x) ((void)0)
#define PD_LDS_FLAG(p) (*(volatile unsigned*)(p))
#define PD_BALLOT(c) (::pdmp3::emu::ballot(c))
namespace pdmp3 {
typedef const ConstBank* BankPtr;
}
#define PD_LAUNDER(p) ((void)0)
#define PD_UNIFORM_PTR(T, p) (p)
#define PD_UNCHECKED_INDEX(v, lo, n) ((unsigned)((v) - (lo)) < (unsigned)(n) ? (v) : 0)
namespace pdmp3 {
typedef float f32x2 __attribute__((vector_size(8)));
typedef unsigned short u16x2 __attribute__((vector_size(4)));
typedef float f32x4 __attribute__((vector_size(16)));
static inline f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return (f32x2){__builtin_fmaf(a[0], b[0], c[0]), __builtin_fmaf(a[1], b[1], c[1])}; }
static inline f32x4 mfma16(float a, float b, f32x4 c) {
  float cd[4] = {c[0], c[1], c[2], c[3]};
  emu::mfma16(a, b, cd);
  return (f32x4){cd[0], cd[1], cd[2], cd[3]};
}
static inline void permlane32_swap(int& a, int& b) { emu::permlane32_swap(&a, &b); }
// v_med3_f32 with the hardware's NaN rule: any NaN among the operands -> min3 of the others
static inline float pd_fmed3_host(float x, float lo, float hi) {
  if (x != x) return lo < hi ? lo : hi;
  return x < lo ? lo : (x > hi ? hi : x);
}
// the binary32 product x * 32767 rounded toward zero: the binary64 product is exact; round it to nearest and step back
// when that went away from zero
static inline void mul18_rtz_32767(const float* x, float* p) {
  for (int t = 0; t < 18; t++) {
    const double d = (double)x[t] * 32767.0;
    float f = (float)d;
    if (f == f && !__builtin_isinf(f) && __builtin_fabs((double)f) > __builtin_fabs(d)) f = __builtin_nextafterf(f, 0.0f);
    else if (__builtin_isinf(f) && !__builtin_isinf(x[t])) f = __builtin_copysignf(0x1.fffffep+127f, f);   // RTZ never rounds up to infinity
    p[t] = f;
  }
}
}  // namespace pdmp3
#define PD_FMED3(x, lo, hi) (::pdmp3::pd_fmed3_host((x), (lo), (hi)))
#endif

// development only (tools/ab_pmc.sh): phases switched off for instruction-count / timing experiments -- results are wrong
#ifndef PD_EXP_SKIP
#define PD_EXP_SKIP 0
#endif
// the VALU columns' coefficients as [m][q] (q fastest): the pair (q, q + 1) of one m is an aligned scalar-register pair, a
// v_pk_fma_f32 operand as it comes out of the s_load -- as [q][m] every packed FMA needed one or two s_mov_b32 in front of
// it and the block two more scalar loads (131072 frames 1.037 -> 1.005 ms, round 5; 0 = the old layout, for A/B)
#ifndef PD_C36_MQ
#define PD_C36_MQ 1
#endif
// A/B switches of round 5 (tools/build_variants.sh): the chunk kernel's requantisation with the long-block fast path;
// the band scales of the next granule in ONE phase with the window (their LDS round trips side by side)
#ifndef PD_CHUNK_FAST_REQUANT
#define PD_CHUNK_FAST_REQUANT 0
#endif
// issue priorities of the chunk kernel's phases (s_setprio 0..3; A/B in profiles/r05_kernel_experiments.txt)
#ifndef PD_PRIO_MFMA
#define PD_PRIO_MFMA 2
#endif
#ifndef PD_PRIO_WINDOW
#define PD_PRIO_WINDOW 0
#endif
#ifndef PD_PRIO_REQUANT
#define PD_PRIO_REQUANT 1
#endif
#ifndef PD_PRIO_REST
#define PD_PRIO_REST 0
#endif
#ifndef PD_PRIO_AA
#define PD_PRIO_AA 1
#endif

#ifndef PD_PRIO_SCALES
#define PD_PRIO_SCALES PD_PRIO_REST
#endif
#ifndef PD_SCALES_WITH_WINDOW
#define PD_SCALES_WITH_WINDOW 0
#endif


namespace pdmp3 {

constexpr int kOvlRegs = 18;          // overlap: [ch][h][r] in MFMA C/D layout for p = 18 + j, plus p = 34, 35 of the lane's own (ch, sb)

constexpr int kHaloGranules = 2;      // (f-1,gr0) (f-1,gr1): overlap depth 1, polyphase history 15 slots
constexpr int kHaloGranulesH5 = 3;    // + (f-2,gr1) when (f-1,gr1,ch1) is a short block: its requantisation peeks at
                                      //   (f-1,gr0)'s synthesis output, which needs one more granule of overlap
constexpr int kHistSlots = 15;        // polyphase history depth (P:2015-2019 reaches 15 slots back)
constexpr int kPow43Small = 128;      // |is| below this come from the LDS copy of the table

// Small tables; lives in __constant__ memory on the device so that
// wave-uniform indices become scalar loads.
struct ConstBank {
#if PD_C36_MQ
  float c36x[18][4];      // cos_N36 (P:620-729) columns p = 16, 17, 34, 35 as [m][q]: the columns left to the VALU; q fastest, so
  float s36x[18][4];      //   that the pair (q, q + 1) of one m is an aligned pair of scalar registers (v_pk_fma_f32 operand)
#define PD_C36(cb, t, q, m) ((cb)->t[m][q])
#else
  float c36x[4][18];      // cos_N36 (P:620-729) columns p = 16, 17, 34, 35 as [q][m]: the columns left to the VALU
  float s36x[4][18];      // the same columns of the short-block matrix (3 x 12-point IMDCT, win[2] folded in)
#define PD_C36(cb, t, q, m) ((cb)->t[q][m])
#endif
  float cs[8], ca[8];     // P:573-574
  float isr_l[16];        // is_ratio_l for is_pos 0..6 (P:2166-2172); [7] unused;
  float isr_r[16];        // [8..15]: the reference reads past is_ratios[] (H3) -> defined as t = 0
  float dwin[512];        // g_synth_dtbl (P:740-870)
  uint16_t sfb_l[9][24];  // [0..2]: g_sf_band_indices[].l (P:879-892), padded; [3..8]: the LSF rates (lsf_tables.h; not the reference)
  uint16_t sfb_s[9][16];  // g_sf_band_indices[].s
  float isr_lsf_l[2][32]; // LSF intensity stereo (13818-3 2.4.3.2): [intensity_scale][is_pos] -> the left channel's factor ...
  float isr_lsf_r[2][32]; // ... and the right one's: i0^((p + 1) / 2) for the channel p's parity picks, 1 for the other
};

// (BankPtr, the pointer to the bank, and PD_LAUNDER: top of the file)

// Large tables in global memory (L2-resident; copied to LDS per chunk where hot).
struct GlobalTables {
  const float* pow43;       // [8207] (float)pow((float)i, 4.0/3.0), P:979
  const uint16_t* linetab;  // [3 sfreq][3 kind][576]: for REORDERED line d: source line (10 bits) |
                            //   scale-table index of that source line << 10.
                            //   kind 0 long, 1 short, 2 mixed; index 0..21 long sfb, 22+sfb*3+win short
  const float* win;         // [4][36] g_imdct_win (P:577-603)
  // MFMA B-operand fragments, lane-indexed [fragment][64 lanes] (host_tables.h: build_fragments)
  const float* frag_long;   // [5 kk][2 nt]   cos_N36 columns {p = j | p = 18 + j}
  const float* frag_short;  // [5 kk][2 nt]   3 x 12-point IMDCT with win[2] folded in, same column map
  const float* frag_mat;    // [2 even/odd][4 k-steps]: 16 x 16 halves of the 32-point DCT-II, rows in register order
  const float* taps;        // [16][64 lanes]: the lane's window coefficients we[0..7], wo[0..7] (g_synth_dtbl, sign folded)
  const void* tab_image;    // [3 sfreq] images of TabLds (host_tables.h: build_tab_images), copied to LDS as they are
};

// LDS.  WaveData is a wave's own working set; TabLds are the hot tables -- one copy per wave in the chunk kernels
// (WaveLds), one per WORKGROUP in the granule kernel (run_granule).  Buffers whose lifetimes do not overlap share
// storage:   xr (ph_requant .. ph_mfma's operand fetch)   |   hyb (ph_mfma .. ph_window)
// hyb rows are [slot t][33]: the DCT lane that owns slot t transforms its row in place.
struct WaveData {
  alignas(16) int16_t spec[2][576];     // committed one granule ahead (before the previous granule's PCM stores)
  alignas(16) int16_t pcm[576];
  alignas(16) uint8_t side[2][128];
  float scale[2][64];
  union alignas(16) {
    float xr[2][576];
    float hyb[2][18][33];
  };
  float peek[4];
  float lo[2][4][16];       // MFMA build: even / odd folded time slots 16, 17: [a|b][2 ch + (t - 16)][k]
};
struct alignas(16) TabLds {
  alignas(16) float win[4][36];
  float pow43z[2 * kPow43Small];   // [128 + v] = sign(v) |v|^(4/3) for v = -128 .. 127: magnitude lookup and sign in one read
  alignas(16) uint16_t ltab[3][576];
  uint8_t bandaddr[3][5][64];      // long blocks (ph_requant_long): [sfreq] 4 x scale index of the lane's lines, see fast_line()
  int sfreq;                       // the sampling frequency ltab is for
  int pad_[3];
  // the window taps of lane i = lane & 31 (GlobalTables::taps holds them per lane; both channels' are alike): the granule
  // kernel reads them from here -- its sixteen waves took 16 x 256 B each through the CU's L1 at their entry, 64 KB per
  // workgroup; as part of the table image that is 2 KB per workgroup
  alignas(16) float taps[16][32];
};
static_assert(sizeof(TabLds) % 16 == 0, "copied in 16-byte pieces");
struct WaveLds : WaveData {
  TabLds tab;
};

typedef uint32_t Chunk16 __attribute__((vector_size(16)));   // one 16-byte global/LDS access

struct LaneRegs {
  float ovl[kOvlRegs];  // IMDCT overlap, the reference's store[ch][sb][] (P:1755)
  float bi[10];      // MFMA build: B fragments of the long IMDCT matrix
  float bm[8];       // MFMA build: B fragments of the matrixing (even / odd 16 x 16 halves of the 32-point DCT-II)
  float we[8];       // window coefficients of (ch, i = lane&31): even taps, sign folded
  float wo[8];       // odd taps
  float he[15];      // polyphase history: coefficient idx_e of the last 15 slots (oldest first)
  float ho[15];      // same for idx_o   (together = what the lane needs of v_vec[ch][], P:1983)
  float in[18];      // scratch: this granule's antialiased IMDCT input
  Chunk16 pf0, pf1, pf2, pf3;   // next granule's spectra / side records in flight
  int idx_e, idx_o;  // which DCT coefficient the lane reads from even-/odd-aged slots
};

// Uniform per-granule facts, decoded from the side records in LDS.
struct GranuleInfo {
  int fr;                  // the frame's flag byte (PDMP3_FR_*)
  int nch, sfreq, mode, mode_ext;   // sfreq: 0..2 MPEG-1; 3..8 LSF (3 * version + the frame byte's field: lsf_tables.h)
  int ver;                 // 0 MPEG-1, 1 MPEG-2 LSF, 2 MPEG-2.5 (pdmp3_gc_side.lsf; granules of an LSF launch come in pairs of FRAMES: engine.hip k_lsf_pair)
  int lsf_scale;           // LSF: intensity_scale of channel 1
  int iso;                 // PDMP3_GC_ISO_* of the frame's records (0: the reference's behaviour); all set for LSF
  int count1_0, count1_1, flags0, flags1;
  PD_MFN int flags(int ch) const { return ch ? flags1 : flags0; }
  PD_MFN bool is_short(int ch) const {
    const int f = flags(ch);
    return (f & PDMP3_GC_WIN_SWITCH) && ((f & PDMP3_GC_BLOCK_TYPE_MASK) >> PDMP3_GC_BLOCK_TYPE_SHIFT) == 2;
  }
  PD_MFN bool is_mixed(int ch) const { return (flags(ch) & PDMP3_GC_MIXED) != 0; }
  PD_MFN int block_type(int ch) const { return (flags(ch) & PDMP3_GC_BLOCK_TYPE_MASK) >> PDMP3_GC_BLOCK_TYPE_SHIFT; }
  PD_MFN int kind(int ch) const { return is_short(ch) ? (is_mixed(ch) ? 2 : 1) : 0; }
};

// RARE = false: the caller knows that the granule is neither LSF nor intensity stereo (frame_is_rare below: the kernels
// send such frames down a second copy of their code, so that the copy every ordinary granule goes through is as tight as
// it was before those features existed -- inlined into the one loop they cost the chunk kernel 5 % in round 6)
template <bool RARE = true>
PD_FN GranuleInfo granule_info(const WaveData& L) {
  GranuleInfo g;
  // (the first dword of each record: count1 u16, global_gain, flags; byte 7: frame flags)
  const uint32_t w0 = (uint32_t)PD_UNIFORM(*reinterpret_cast<const uint32_t*>(&L.side[0][0]));
  const uint32_t w1 = (uint32_t)PD_UNIFORM(*reinterpret_cast<const uint32_t*>(&L.side[1][0]));
  const int fr = PD_UNIFORM(L.side[0][7]);
  g.fr = fr;
  g.iso = PD_UNIFORM(L.side[0][offsetof(pdmp3_gc_side, iso)]);
  g.sfreq = fr & PDMP3_FR_SFREQ_MASK;
  if (g.sfreq > 2) g.sfreq = 2;
  g.ver = 0;
  g.lsf_scale = 0;
  if (RARE) {
    const int lsf0 = PD_UNIFORM(L.side[0][offsetof(pdmp3_gc_side, lsf)]);
    g.ver = lsf0 & PDMP3_LSF_VERSION_MASK;
    if (g.ver > 2) g.ver = 2;
  }
  if (RARE && g.ver) {                // (wave-uniform; an MPEG-1 granule pays one scalar compare for all of this)
    g.sfreq += 3 * g.ver;
    g.iso |= PDMP3_GC_ISO_MS_ALL | PDMP3_GC_ISO_IS_SHORT | PDMP3_GC_ISO_IS_STD;     // no reference behaviour exists for LSF: the standard's
    g.lsf_scale = (PD_UNIFORM(L.side[1][offsetof(pdmp3_gc_side, lsf)]) & PDMP3_LSF_IS_SCALE) ? 1 : 0;
  }
  g.mode = (fr & PDMP3_FR_MODE_MASK) >> PDMP3_FR_MODE_SHIFT;
  g.mode_ext = (fr & PDMP3_FR_MODEEXT_MASK) >> PDMP3_FR_MODEEXT_SHIFT;
  g.nch = (g.mode == 3) ? 1 : 2;
  g.count1_0 = (int)(w0 & 0xffffu);
  g.count1_1 = (int)(w1 & 0xffffu);
  g.flags0 = (int)(w0 >> 24);
  g.flags1 = (int)(w1 >> 24);
  return g;
}

// A frame the fast copies of the kernels do not take: intensity stereo (joint stereo with mode_extension bit 0) or LSF.
// rec = any gc record of the frame.
PD_FN bool frame_is_rare(const pdmp3_gc_side* rec) {
  const uint8_t* b = reinterpret_cast<const uint8_t*>(rec);
  const unsigned fr = b[7];
  const bool is = ((fr & PDMP3_FR_MODE_MASK) >> PDMP3_FR_MODE_SHIFT) == 1 && (fr & (1u << PDMP3_FR_MODEEXT_SHIFT));
  return is || (b[offsetof(pdmp3_gc_side, lsf)] & PDMP3_LSF_VERSION_MASK) != 0;
}

PD_HD uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
PD_HD float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }


// ---------------------------------------------------------------------------
// chunk prologue: per-lane constants, LDS copies of hot tables
// ---------------------------------------------------------------------------
PD_FN void state_zero(int lane, LaneRegs& R) {
  (void)lane;
  for (int m = 0; m < kOvlRegs; m++) R.ovl[m] = 0.0f;
  for (int s = 0; s < kHistSlots; s++) { R.he[s] = 0.0f; R.ho[s] = 0.0f; }
}

// the per-lane constants of a wave
template <bool TAPS = true>          // TAPS = false: the caller takes we / wo from the workgroup's table block (TabLds::taps)
PD_FN void lane_init(int lane, WaveData& L, LaneRegs& R, BankPtr cb, const GlobalTables& T) {
  const int i = lane & 31;
  // v[i] = C[16+i] (i<16), 0 (i==16), -C[48-i] (i>16);  v[32+i] = -C[16-i] (i<=16), -C[i-16] (i>16)
  // where C = 32-point DCT-II of the slot (derivation: DESIGN.md "polyphase").  The window coefficients that go with
  // them -- sgn_e * D[64 k + i] and -D[64 k + 32 + i] -- come per lane from T.taps (host_tables.h).
  (void)cb;
  R.idx_e = (i < 16) ? 16 + i : ((i == 16) ? 0 : 48 - i);
  R.idx_o = (i <= 16) ? 16 - i : i - 16;
  if (TAPS) {
    for (int k = 0; k < 8; k++) {
      R.we[k] = T.taps[k * 64 + lane];
      R.wo[k] = T.taps[(8 + k) * 64 + lane];
    }
  }
  state_zero(lane, R);
  if (lane < 4) L.peek[lane] = 1.0f;
  for (int k = 0; k < 10; k++) R.bi[k] = T.frag_long[k * 64 + lane];
  for (int k = 0; k < 8; k++) R.bm[k] = T.frag_mat[k * 64 + lane];
}

// Lines of a lane in the long-block fast path (ph_requant_long): the PAIRS (2 l + 128 i, 2 l + 1 + 128 i), i = 0..3 --
// one LDS dword of int16 spectra each, and always inside one scalefactor band (every long band boundary of P:879-892
// is even) -- and the single line 512 + l (i = 4).
PD_HD int fast_line(int lane, int i) { return i < 4 ? 2 * lane + 128 * i : 512 + lane; }
PD_FN unsigned band_addr_of(const GlobalTables& T, int sfreq, int line) {      // 4 x long-block band of a line
  return (unsigned)((T.linetab[(size_t)sfreq * 3 * 576 + line] >> 10) << 2);
}

// The tables, brought in by `nthr` threads (tid = 0 .. nthr - 1): the image of TabLds for one sampling frequency, built on
// the host (host_tables.h: the signed |is|^(4/3) table, the IMDCT windows, the band addresses of all three sampling
// frequencies, the line tables of this one), copied as it is.
PD_FN void tab_load_image(int tid, int nthr, TabLds& S, const GlobalTables& T, int sfreq) {
  const Chunk16* src = reinterpret_cast<const Chunk16*>(reinterpret_cast<const char*>(T.tab_image) + (size_t)sfreq * sizeof(TabLds));
  Chunk16* dst = reinterpret_cast<Chunk16*>(&S);
  for (int k = tid; k < (int)(sizeof(TabLds) / 16); k += nthr) dst[k] = src[k];
}
PD_FN void load_linetab(int lane, TabLds& S, const GlobalTables& T, int sfreq) { tab_load_image(lane, 64, S, T, sfreq); }

// state layout (opaque to callers): float ovl[kOvlRegs][64 lanes]; float he[15][64]; float ho[15][64]
constexpr int kStateFloats = 64 * (kOvlRegs + 2 * kHistSlots);

PD_FN void state_load(int lane, LaneRegs& R, const float* st) {
  for (int m = 0; m < kOvlRegs; m++) R.ovl[m] = st[m * 64 + lane];
  for (int s = 0; s < kHistSlots; s++) {
    R.he[s] = st[(kOvlRegs + s) * 64 + lane];
    R.ho[s] = st[(kOvlRegs + kHistSlots + s) * 64 + lane];
  }
}

// channel 1's part of the carried state only: its overlap tails (ovl[8..15] in every lane, ovl[16..17] in the
// (ch 1, sb) lanes) and its polyphase history; what a run of mono frames leaves untouched (P:1777, P:2126 are
// indexed by channel)
PD_FN void state_load_ch1(int lane, LaneRegs& R, const float* st) {
  for (int m = 8; m < 16; m++) R.ovl[m] = st[m * 64 + lane];
  if (lane >= 32) {
    R.ovl[16] = st[16 * 64 + lane];
    R.ovl[17] = st[17 * 64 + lane];
    for (int s = 0; s < kHistSlots; s++) {
      R.he[s] = st[(kOvlRegs + s) * 64 + lane];
      R.ho[s] = st[(kOvlRegs + kHistSlots + s) * 64 + lane];
    }
  }
}

PD_FN void state_store(int lane, const LaneRegs& R, float* st) {
  for (int m = 0; m < kOvlRegs; m++) st[m * 64 + lane] = R.ovl[m];
  for (int s = 0; s < kHistSlots; s++) {
    st[(kOvlRegs + s) * 64 + lane] = R.he[s];
    st[(kOvlRegs + kHistSlots + s) * 64 + lane] = R.ho[s];
  }
}

// ---------------------------------------------------------------------------
// ph_prefetch / ph_commit: 2304 B spectra + 256 B side of one granule,
// 16 B per lane per access, issued one granule ahead of its use
// ---------------------------------------------------------------------------
PD_FN void ph_prefetch(int lane, LaneRegs& R, const int16_t* spectra_g, const pdmp3_gc_side* side_g) {
  const Chunk16* src = reinterpret_cast<const Chunk16*>(spectra_g);
  R.pf0 = src[lane];
  R.pf1 = src[lane + 64];
  if (lane < 16) {
    R.pf2 = src[lane + 128];
    R.pf3 = reinterpret_cast<const Chunk16*>(side_g)[lane];
  }
}

PD_FN void ph_commit(int lane, WaveData& L, const LaneRegs& R) {
  Chunk16* dst = reinterpret_cast<Chunk16*>(&L.spec[0][0]);
  dst[lane] = R.pf0;
  dst[lane + 64] = R.pf1;
  if (lane < 16) {
    dst[lane + 128] = R.pf2;
    reinterpret_cast<Chunk16*>(&L.side[0][0])[lane] = R.pf3;
  }
}

// ---------------------------------------------------------------------------
// ph_scales: scale[ch][e], e = 0..21 long sfb, 22+sfb*3+win short.
//   t1 = (float)pow(2.0, -(sfm*(sf+pre)))  = {1, 2^-1/2}[n&1] * 2^-(n>>1), n = sf+pre (or twice that)
//   t2 = (float)pow(2.0, 0.25*k)           = {2^(j/4)}[k&3] * 2^(k>>2)
// Power-of-two scaling of a binary32 is exact, so both equal the reference's
// libm values bit for bit (host_tables.h re-checks this against pow() for the
// whole index range when the engine is created).
// ---------------------------------------------------------------------------
PD_HD float pow2_quarter(int k) {        // 2^(k/4), k in [-266, 45]
  const int j = k & 3;
  const float base = (j == 0) ? 1.0f : (j == 1) ? 0x1.306fe0p+0f : (j == 2) ? 0x1.6a09e6p+0f : 0x1.ae89fap+0f;
  return ldexpf(base, k >> 2);
}
PD_HD float pow2_neg_half(uint32_t n) {  // 2^(-n/2); 0 from n = 300 on (binary32 underflow)
  const float v = ldexpf((n & 1) ? 0x1.6a09e6p-1f : 1.0f, -(int)((n < 300u ? n : 300u) >> 1));
  return n >= 300u ? 0.0f : v;
}

// straight-line (select-only) so that the LDS reads of both channels are in flight together
PD_FN void ph_scales(int lane, WaveData& L) {
  const bool is_long = lane < 22;
  const int ll = is_long ? lane : 21;
  const int q = is_long ? 0 : (lane < 61 ? lane - 22 : 38);
  const int sfb = q / 3, win = q - 3 * sfb;
  const int pre = (int)((0x2fe95400000ull >> (2 * ll)) & 3);     // pretab P:2123 (+ [21] = 0, H4), 2 bits per entry
  const float pk = L.peek[win];
  float res[2];
  PD_UNROLL for (int ch = 0; ch < 2; ch++) {
    const uint8_t* s = L.side[ch];
    const int gg = s[2], flags = s[3];
    const uint32_t sl = s[8 + ll], ss = s[30 + q], mark = s[30 + 36], sbg = s[4 + win];
    const bool sfscale = flags & PDMP3_GC_SCALEFAC_SCALE;
    uint32_t sf_short = (sfb == 12 && mark == PDMP3_SF_PEEK) ? f2u(pk) : ss;                 // H5
    sf_short = sf_short > 400u ? 400u : sf_short;
    const uint32_t x = is_long ? sl + ((flags & PDMP3_GC_PREFLAG) ? (uint32_t)pre : 0u) : sf_short;
    const float t1 = pow2_neg_half(sfscale ? 2 * x : x);
    const float t2 = pow2_quarter(gg - 210 - (is_long ? 0 : 8 * (int)sbg));
    res[ch] = t1 * t2;
  }
  if (lane < 61) { L.scale[0][lane] = res[0]; L.scale[1][lane] = res[1]; }
}

// ---------------------------------------------------------------------------
// ph_requant: requantise + reorder (gather) + stereo.  Lane l owns the
// REORDERED lines l, l + 64, ..., l + 512 of both channels (consecutive lanes
// touch consecutive LDS words: no bank conflicts on the int16 / u16 tables).
// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
// ph_stereo_is: joint stereo of a granule that HAS intensity stereo (mode_extension bit 0) -- the reference's
// (P:1932-1971 with its H3 defects) or the standard's (PDMP3_GC_ISO_IS_STD; always for LSF) -- plus the granule's M/S
// rotation.  Works in place on L.xr[ch][reordered line] (ph_requant has stored the requantised lines there), lines
// lane + 64 i, i < ni (9; 1 for the peek-only halo granule).  Only in the RARE copies of the kernels' code (ph_requant).
// ---------------------------------------------------------------------------
PD_FN void ph_stereo_is(int lane, WaveData* Lp, const TabLds* Sp, BankPtr cb, const uint16_t* linetab, GranuleInfo g, int ni, int tg) {
  WaveData& L = *Lp;
  const TabLds& S = *Sp;
  const uint16_t* gtab = linetab + (size_t)g.sfreq * 3 * 576;
  const bool ms = (g.mode_ext & 2) != 0;
  const int cmin = (g.iso & PDMP3_GC_ISO_MS_ALL) ? 576 : (g.count1_0 > g.count1_1 ? g.count1_1 : g.count1_0);
  const int kind0 = g.kind(0), kind1 = g.kind(1);
  const bool is_std = (g.iso & PDMP3_GC_ISO_IS_STD) != 0;
  constexpr int kIsNone = 255;
  const uint8_t* sd0 = L.side[0];
  const uint8_t* sd1 = L.side[1];
  // PDMP3_GC_ISO_IS_STD (NOT the reference; pinned against FFmpeg): the standard's intensity stereo.
  // A band (of a window, in short blocks) is intensity coded when the RIGHT channel holds no non-zero value in it or
  // above it and the right channel's scalefactor there -- the last band borrows the one below -- is not the "off"
  // value (7; LSF: the largest its slen holds); the block shape is the right channel's.
  float lastl = -1.0f, last0 = -1.0f, last1 = -1.0f, last2 = -1.0f;
  bool any_short = false;
  if (is_std) {
    // the last band with a non-zero line: of the long part, and of each window of the short part (as floats: PD_SHFL_XOR)
    PD_NOUNROLL for (int i = 0; i < 9; i++) {        // (all 576 lines, also for the peek-only granule)
      const int d = lane + 64 * i;
      const unsigned e = tg ? gtab[kind1 * 576 + d] : S.ltab[kind1][d];
      const int idx = (int)(e >> 10);
      if (L.spec[1][e & 1023] != 0) {
        const float b = (float)(idx < 22 ? idx : (idx - 22) / 3);
        const int w = idx < 22 ? -1 : (idx - 22) % 3;
        lastl = (w < 0 && b > lastl) ? b : lastl;
        last0 = (w == 0 && b > last0) ? b : last0;
        last1 = (w == 1 && b > last1) ? b : last1;
        last2 = (w == 2 && b > last2) ? b : last2;
      }
    }
    PD_NOUNROLL for (int m = 1; m < 64; m <<= 1) {
      const float ol = PD_SHFL_XOR(lastl, m), o0 = PD_SHFL_XOR(last0, m), o1 = PD_SHFL_XOR(last1, m), o2 = PD_SHFL_XOR(last2, m);
      lastl = ol > lastl ? ol : lastl;
      last0 = o0 > last0 ? o0 : last0;
      last1 = o1 > last1 ? o1 : last1;
      last2 = o2 > last2 ? o2 : last2;
    }
    any_short = last0 >= 0.0f || last1 >= 0.0f || last2 >= 0.0f;
  }
  PD_NOUNROLL for (int i = 0; i < ni; i++) {
    const int d = lane + 64 * i;
    float a0 = L.xr[0][d], a1 = L.xr[1][d];
    if (is_std) {
      const unsigned e = tg ? gtab[kind1 * 576 + d] : S.ltab[kind1][d];
      const int idx = (int)(e >> 10);
      bool coded;
      int pos, bi;               // bi: the scalefactor's number in transmission order (LSF: which partition it came in)
      if (idx < 22) {
        coded = !any_short && (float)idx > lastl;
        bi = idx < 21 ? idx : 20;
        pos = sd1[8 + bi];
      } else {
        const int b = (idx - 22) / 3, w = (idx - 22) % 3, bb = b < 12 ? b : 11;
        const float lw = w == 0 ? last0 : (w == 1 ? last1 : last2);
        coded = (float)b > lw;
        pos = sd1[30 + bb * 3 + w];
        bi = (kind1 == 2 ? 6 + (bb - 3) * 3 : bb * 3) + w;
      }
      int ispos;
      if (g.ver) {
        // 13818-3 2.4.3.2: "not intensity coded" is the largest value the scalefactor's slen holds
        int acc = 0, ill = 0;
        bool found = false;
        PD_NOUNROLL for (int k = 0; k < 4; k++) {
          acc += sd1[offsetof(pdmp3_gc_side, lsf_nsfb) + k];
          const int v = (1 << sd1[offsetof(pdmp3_gc_side, lsf_slen) + k]) - 1;
          ill = (!found && bi < acc) ? v : ill;
          found = found || bi < acc;
        }
        ispos = (coded && pos != ill) ? (pos & 31) : kIsNone;
      } else ispos = (coded && pos < 7) ? pos : kIsNone;
      if (ispos == kIsNone) {
        if (ms && d < cmin) {    // P:1921-1928
          const float sum = a0 + a1, dif = a0 - a1;
          a0 = (float)((double)sum * 0.70710678118654752440);
          a1 = (float)((double)dif * 0.70710678118654752440);
        }
      } else {
        const float kl = g.ver ? cb->isr_lsf_l[g.lsf_scale][ispos & 31] : cb->isr_l[ispos & 7];
        const float kr = g.ver ? cb->isr_lsf_r[g.lsf_scale][ispos & 31] : cb->isr_r[ispos & 7];
        const float x = a0;
        a0 = kl * x;
        a1 = kr * x;
      }
    } else {
      // the reference's: M/S first (P:1921-1928), then P:1932-1971 with the block shape taken from channel 0
      if (ms && d < cmin) {
        const float sum = a0 + a1, dif = a0 - a1;
        a0 = (float)((double)sum * 0.70710678118654752440);
        a1 = (float)((double)dif * 0.70710678118654752440);
      }
      const int c1 = g.count1_1;
      bool do_long = false, do_short = false;
      int sfb = 0, win = 0;
      if (kind0 == 0) {
        sfb = (tg ? gtab[d] : S.ltab[0][d]) >> 10;
        do_long = (sfb < 21);
      } else if (kind0 == 2 && d < 36) {
        sfb = (tg ? gtab[d] : S.ltab[0][d]) >> 10;
        do_long = (sfb < 8);
      } else {
        // band of POSITION d in the un-reordered [win][j] layout (P:2203)
        PD_NOUNROLL for (int b = 0; b < 13; b++) {
          const int lo = 3 * cb->sfb_s[g.sfreq][b], hi = 3 * cb->sfb_s[g.sfreq][b + 1];
          // (ISO switch: the lines are in reordered order by now, position lo + 3 j + w belongs to window w)
          if (d >= lo && d < hi) { sfb = b; win = (g.iso & PDMP3_GC_ISO_IS_SHORT) ? (d - lo) % 3 : (d - lo) / ((hi - lo) / 3); }
        }
        do_short = (sfb < 12) && (kind0 == 1 || sfb >= 3);
      }
      if (do_long && (int)cb->sfb_l[g.sfreq][sfb] >= c1) {
        const int is_pos = sd0[8 + sfb];
        if (is_pos != 7) {
          const float l = cb->isr_l[is_pos & 15] * a0;
          const float r = cb->isr_r[is_pos & 15] * a0;
          a0 = l; a1 = r;
        }
      }
      if (do_short && 3 * (int)cb->sfb_s[g.sfreq][sfb] >= c1) {
        const int is_pos = sd0[30 + sfb * 3 + win];
        if (is_pos != 7 && (g.iso & PDMP3_GC_ISO_IS_SHORT)) {   // the standard's: the ratios of the long case, multiplied
          const float l = cb->isr_l[is_pos & 15] * a0;
          const float r = cb->isr_r[is_pos & 15] * a0;
          a0 = l; a1 = r;
        } else if (is_pos != 7) {   // H3: sample forced through `unsigned` (x86-64 conversion semantics)
          const float xv = a0;
          long long t = (xv >= 9.2233720368547758e18f || xv < -9.2233720368547758e18f || xv != xv)
                            ? (long long)0x8000000000000000ull : (long long)xv;
          const float vv = (float)(uint32_t)(unsigned long long)t;
          a0 = vv; a1 = vv;
        }
      }
    }
    L.xr[0][d] = a0;
    L.xr[1][d] = a1;
  }
}

template <bool TG, bool SCALES> PD_FN void ph_requant_long(int lane, WaveData& L, const TabLds& S, const GlobalTables& T, const GranuleInfo& g);   // (below)
PD_FN void ph_scales(int lane, WaveData& L);

// NI = 9: the whole granule.  NI = 1: only the reordered lines 0..63 (the peek-only halo granule, see run_chunk).
// FAST: granules of long blocks take ph_requant_long.  TG: the tables in S may be for another sampling frequency than
// the granule's (granule kernel: one table block per workgroup) -- then the line table is read from global memory.
// SCALES: the band scales (ph_scales) are computed in here -- in the fast path while the loads from the full |is|^(4/3)
// table are in flight (straight-line callers, which have nothing else to put there).
template <bool DUMP, int NI = 9, bool FAST = false, bool TG = false, bool SCALES = false, bool RARE = true>
PD_FN void ph_requant(int lane, WaveData& L, const TabLds& S, BankPtr cb, const GlobalTables& T, float* dump0, float* dump1,
                      const GranuleInfo* gi = nullptr) {
  // (gi: the granule's facts, read once by the caller -- every granule_info() is an LDS round trip and four
  //  v_readfirstlane at the head of a phase, and the phase fences keep the compiler from sharing one between phases)
  const GranuleInfo g = gi ? *gi : granule_info<RARE>(L);
  const bool tg = TG && (g.sfreq != S.sfreq);                       // wave-uniform
  const uint16_t* gtab = T.linetab + (size_t)g.sfreq * 3 * 576;
  const bool joint = (g.nch == 2) && (g.mode == 1) && (g.mode_ext != 0);
  const bool ms = joint && (g.mode_ext & 2);
  const bool is = RARE && joint && (g.mode_ext & 1);                // (RARE = false: no such granule comes this way)
  const int cmin = (g.iso & PDMP3_GC_ISO_MS_ALL) ? 576 : (g.count1_0 > g.count1_1 ? g.count1_1 : g.count1_0);   // P:1920 (H2): the smaller; ISO switch: every line
  const int kind0 = g.kind(0), kind1 = g.kind(1);
  if (FAST && !DUMP && NI == 9) {
    if (kind0 == 0 && (g.nch == 1 || kind1 == 0) && !is && g.ver == 0) {     // wave-uniform (the fast path's band addresses are MPEG-1's)
      ph_requant_long<TG, SCALES>(lane, L, S, T, g);
      return;
    }
  }
  if (SCALES) { ph_scales(lane, L); PD_WAVE_SYNC(); }
  float x0[NI], x1[NI];
#define PD_LINE(i) (lane + 64 * (i))
  {
    unsigned e0[NI], e1[NI];
    // (the values are pinned inside each branch: merged, the two would become one FLAT load of a selected address.
    //  ALL the loads first, then the pins: a pin is an asm statement that reads the value, so a pin right behind its load
    //  is a wait for that load -- eighteen LDS round trips one after the other, which is what this was until round 5)
    if (tg) {
      PD_UNROLL for (int i = 0; i < NI; i++) e0[i] = gtab[kind0 * 576 + PD_LINE(i)];
      PD_UNROLL for (int i = 0; i < NI; i++) e1[i] = gtab[kind1 * 576 + PD_LINE(i)];
      PD_UNROLL for (int i = 0; i < NI; i++) { PD_PIN(e0[i]); PD_PIN(e1[i]); }
    } else {
      PD_UNROLL for (int i = 0; i < NI; i++) e0[i] = S.ltab[kind0][PD_LINE(i)];
      PD_UNROLL for (int i = 0; i < NI; i++) e1[i] = S.ltab[kind1][PD_LINE(i)];
      PD_UNROLL for (int i = 0; i < NI; i++) { PD_PIN(e0[i]); PD_PIN(e1[i]); }
    }
    int v0[NI], v1[NI];
    float s0[NI], s1[NI];
    PD_UNROLL for (int i = 0; i < NI; i++) { v0[i] = L.spec[0][e0[i] & 1023]; s0[i] = L.scale[0][e0[i] >> 10]; }
    PD_UNROLL for (int i = 0; i < NI; i++) { v1[i] = L.spec[1][e1[i] & 1023]; s1[i] = L.scale[1][e1[i] >> 10]; }
    // |is|^(4/3): LDS copy for magnitudes < 128; for the rare larger ones every lane issues an
    // unconditional gather from the full table (index 0 when not needed, i.e. one shared line), all
    // 18 in flight together with the LDS lookups -- no per-value branch, one wait.
    int a0[NI], a1[NI];
    float pg0[NI], pg1[NI], ps0[NI], ps1[NI];
    PD_UNROLL for (int i = 0; i < NI; i++) { a0[i] = v0[i] < 0 ? -v0[i] : v0[i]; a1[i] = v1[i] < 0 ? -v1[i] : v1[i]; }
    PD_UNROLL for (int i = 0; i < NI; i++) {
      // (wave-uniform skip: most groups of 64 lines hold no magnitude above the LDS copy's range, and a gather is
      // a 64-bit address per lane plus a trip through the texture addresser even when every lane asks for [0])
      pg0[i] = 0.0f; pg1[i] = 0.0f;
      if (PD_ANY(a0[i] >= kPow43Small || a1[i] >= kPow43Small)) {
        pg0[i] = T.pow43[a0[i] >= kPow43Small ? (a0[i] > 8206 ? 8206 : a0[i]) : 0];
        pg1[i] = T.pow43[a1[i] >= kPow43Small ? (a1[i] > 8206 ? 8206 : a1[i]) : 0];
      }
    }
    PD_UNROLL for (int i = 0; i < NI; i++) { ps0[i] = S.pow43z[kPow43Small + (a0[i] & (kPow43Small - 1))]; ps1[i] = S.pow43z[kPow43Small + (a1[i] & (kPow43Small - 1))]; }
    PD_UNROLL for (int i = 0; i < NI; i++) {
      const float p = a0[i] >= kPow43Small ? pg0[i] : ps0[i];
      x0[i] = s0[i] * (v0[i] < 0 ? -p : p);                 // (t1*t2)*t3, P:2132
    }
    PD_UNROLL for (int i = 0; i < NI; i++) {
      const float p = a1[i] >= kPow43Small ? pg1[i] : ps1[i];
      x1[i] = (g.nch == 2) ? s1[i] * (v1[i] < 0 ? -p : p) : 0.0f;
    }
  }
  if (DUMP) {
    PD_UNROLL for (int i = 0; i < NI; i++) {
      dump0[PD_LINE(i)] = x0[i];
      if (g.nch == 2) dump0[4 * 576 + PD_LINE(i)] = x1[i];
    }
  }
  // Intensity stereo is rare (no common encoder emits it) and its code is long: it is a function of its own that works
  // on the lines in LDS (ph_stereo_is) and exists only in the RARE copies of the kernels' code.  The M/S rotation of such
  // a granule is done there too (the standard's intensity stereo leaves the coded lines out of it).
  if (ms && !is) {   // P:1921-1928
    PD_UNROLL for (int i = 0; i < NI; i++) {
      const float sum = x0[i] + x1[i], dif = x0[i] - x1[i];
      const float l = (float)((double)sum * 0.70710678118654752440);
      const float r = (float)((double)dif * 0.70710678118654752440);
      const bool in = PD_LINE(i) < cmin;
      x0[i] = in ? l : x0[i];
      x1[i] = in ? r : x1[i];
    }
  }
  PD_UNROLL for (int i = 0; i < NI; i++) {
    L.xr[0][PD_LINE(i)] = x0[i];
    if (g.nch == 2) L.xr[1][PD_LINE(i)] = x1[i];
  }
  if (is) {
    PD_WAVE_SYNC();
    ph_stereo_is(lane, &L, &S, cb, T.linetab, g, NI, tg ? 1 : 0);
    PD_WAVE_SYNC();
  }
  if (DUMP) {
    PD_UNROLL for (int i = 0; i < NI; i++) {
      dump1[PD_LINE(i)] = L.xr[0][PD_LINE(i)];
      if (g.nch == 2) dump1[4 * 576 + PD_LINE(i)] = L.xr[1][PD_LINE(i)];
    }
  }
#undef PD_LINE
}

// ---------------------------------------------------------------------------
// ph_requant_long: the same stage for the common granule -- long blocks in every channel, no intensity stereo
// (wave-uniform).  Reorder is the identity there and the band of a line is a constant of the lane, so nothing goes
// through the line table: lane l owns the line pairs (2 l + 128 i, 2 l + 1 + 128 i), i = 0..3, and line 512 + l
// (fast_line): one LDS dword of spectra per pair, the band's scale address from bandaddr[], |is|^(4/3) WITH its sign
// from pow43z[] in one read, the product and the MS sum / difference as packed operations, 8-byte stores.
// Same arithmetic per line as ph_requant: (t1 t2) (+-t3), P:2132; MS in binary64 like P:1923-1926.
// ---------------------------------------------------------------------------
PD_FN float pow43z_at(const TabLds& S, int v) {
  // |v| >= 128 reads past the table on the device (somewhere in LDS or beyond it: harmless); such values are replaced by the caller
  return S.pow43z[kPow43Small + PD_UNCHECKED_INDEX(v, -kPow43Small, 2 * kPow43Small)];
}
// the full table for the values outside -128 .. 127 (rare): every lane issues its loads unconditionally (entry 0 when
// its value is small -- one shared line), so that all of a group's gathers are in flight together
PD_FN int pow43_big_index(int v) {
  const int a = v < 0 ? -v : v;
  return a < kPow43Small ? 0 : (a > 8206 ? 8206 : a);
}
PD_FN float pow43_big_pick(int v, float small, float big) {
  const int a = v < 0 ? -v : v;
  return a < kPow43Small ? small : (v < 0 ? -big : big);
}
PD_FN float ms_scale(float x) { return (float)((double)x * 0.70710678118654752440); }      // P:1923-1926

// |is|^(4/3) with its sign comes from the LDS table for -128 .. 127.  Values outside it are a few per granule at most:
// the lanes that hold one fetch theirs from the full table under their own execution mask (a skipped block for
// everybody else: one compare and a scalar branch per pair of lines) -- all of a granule's loads are issued before
// anything waits for one, the scales are computed under them, the results are picked up afterwards.
template <bool TG, bool SCALES>
PD_FN void ph_requant_long(int lane, WaveData& L, const TabLds& S, const GlobalTables& T, const GranuleInfo& g) {
  const bool two = g.nch == 2;
  const bool ms = two && (g.mode == 1) && (g.mode_ext & 2);
  const int cmin = (g.iso & PDMP3_GC_ISO_MS_ALL) ? 576 : (g.count1_0 > g.count1_1 ? g.count1_1 : g.count1_0);   // P:1920 (H2): the smaller; ISO switch: every line
  const uint32_t* sp0 = reinterpret_cast<const uint32_t*>(&L.spec[0][0]);
  const uint32_t* sp1 = reinterpret_cast<const uint32_t*>(&L.spec[1][0]);
  const char* sc0 = reinterpret_cast<const char*>(&L.scale[0][0]);
  const char* sc1 = reinterpret_cast<const char*>(&L.scale[1][0]);
  uint32_t w0[4], w1[4];
  unsigned ba[5];
  PD_UNROLL for (int i = 0; i < 4; i++) { w0[i] = sp0[lane + 64 * i]; w1[i] = two ? sp1[lane + 64 * i] : 0u; }
  const int v0s = L.spec[0][512 + lane], v1s = two ? L.spec[1][512 + lane] : 0;
  PD_UNROLL for (int i = 0; i < 5; i++) ba[i] = S.bandaddr[g.sfreq][i][lane];
  // a 16-bit value is inside -128 .. 127 <=> its bits 15..7 are all alike <=> bits 15..8 of v ^ (v << 1) are zero
  bool big0[4], big1[4];
  float ga0[4], gb0[4], ga1[4], gb1[4], gs0 = 0.0f, gs1 = 0.0f;
  PD_UNROLL for (int i = 0; i < 4; i++) {
    big0[i] = ((w0[i] ^ (w0[i] << 1)) & 0xff00ff00u) != 0u;
    big1[i] = ((w1[i] ^ (w1[i] << 1)) & 0xff00ff00u) != 0u;
    ga0[i] = gb0[i] = ga1[i] = gb1[i] = 0.0f;
    // (PD_PIN keeps the blocks conditional: left alone, the compiler speculates the loads -- entry 0 is always safe --
    //  and every lane computes 18 indices and issues 18 gathers per granule for the handful that are wanted)
    if (big0[i]) {
      int ia = pow43_big_index((int)(int16_t)(w0[i] & 0xffffu)), ib = pow43_big_index((int)w0[i] >> 16);
      PD_PIN(ia); PD_PIN(ib);
      ga0[i] = T.pow43[ia];
      gb0[i] = T.pow43[ib];
    }
    if (big1[i]) {
      int ia = pow43_big_index((int)(int16_t)(w1[i] & 0xffffu)), ib = pow43_big_index((int)w1[i] >> 16);
      PD_PIN(ia); PD_PIN(ib);
      ga1[i] = T.pow43[ia];
      gb1[i] = T.pow43[ib];
    }
  }
  const bool bigs0 = (unsigned)(v0s + kPow43Small) >= 2u * kPow43Small, bigs1 = (unsigned)(v1s + kPow43Small) >= 2u * kPow43Small;
  if (bigs0) { int ia = pow43_big_index(v0s); PD_PIN(ia); gs0 = T.pow43[ia]; }
  if (bigs1) { int ia = pow43_big_index(v1s); PD_PIN(ia); gs1 = T.pow43[ia]; }
  if (SCALES) { ph_scales(lane, L); PD_WAVE_SYNC(); }       // (the loads above are in flight)
  PD_UNROLL for (int i = 0; i < 5; i++) {
    const float s0 = *reinterpret_cast<const float*>(sc0 + ba[i]);
    const float s1 = *reinterpret_cast<const float*>(sc1 + ba[i]);
    f32x2 x0, x1;
    if (i < 4) {
      const int a0 = (int)(int16_t)(w0[i] & 0xffffu), b0 = (int)w0[i] >> 16;
      const int a1 = (int)(int16_t)(w1[i] & 0xffffu), b1 = (int)w1[i] >> 16;
      float pa0 = pow43z_at(S, a0), pb0 = pow43z_at(S, b0), pa1 = pow43z_at(S, a1), pb1 = pow43z_at(S, b1);
      if (big0[i]) { pa0 = pow43_big_pick(a0, pa0, ga0[i]); pb0 = pow43_big_pick(b0, pb0, gb0[i]); PD_PIN(pa0); PD_PIN(pb0); }
      if (big1[i]) { pa1 = pow43_big_pick(a1, pa1, ga1[i]); pb1 = pow43_big_pick(b1, pb1, gb1[i]); PD_PIN(pa1); PD_PIN(pb1); }
      const f32x2 p0 = {pa0, pb0}, p1 = {pa1, pb1};
      const f32x2 t0 = {s0, s0}, t1 = {s1, s1};
      x0 = t0 * p0;
      x1 = t1 * p1;
    } else {
      float q0 = pow43z_at(S, v0s), q1 = pow43z_at(S, v1s);
      if (bigs0) { q0 = pow43_big_pick(v0s, q0, gs0); PD_PIN(q0); }
      if (bigs1) { q1 = pow43_big_pick(v1s, q1, gs1); PD_PIN(q1); }
      x0 = (f32x2){s0 * q0, 0.0f};
      x1 = (f32x2){s1 * q1, 0.0f};
    }
    if (ms) {   // P:1921-1928: lines below the smaller count1 only
      const f32x2 sum = x0 + x1, dif = x0 - x1;
      f32x2 l, r;
      l[0] = ms_scale(sum[0]); r[0] = ms_scale(dif[0]);
      if (i < 4) { l[1] = ms_scale(sum[1]); r[1] = ms_scale(dif[1]); } else { l[1] = 0.0f; r[1] = 0.0f; }
      const int end = i < 4 ? 128 * i + 128 : 576;                 // the group's last line + 1
      if (end <= cmin) { x0 = l; x1 = r; }                         // wave-uniform: the whole group is below the bound
      else {
        const bool in_a = fast_line(lane, i) < cmin, in_b = fast_line(lane, i) + 1 < cmin;
        x0[0] = in_a ? l[0] : x0[0]; x0[1] = in_b ? l[1] : x0[1];
        x1[0] = in_a ? r[0] : x1[0]; x1[1] = in_b ? r[1] : x1[1];
      }
    }
    if (i < 4) {
      *reinterpret_cast<f32x2*>(&L.xr[0][2 * lane + 128 * i]) = x0;
      if (two) *reinterpret_cast<f32x2*>(&L.xr[1][2 * lane + 128 * i]) = x1;
    } else {
      L.xr[0][512 + lane] = x0[0];
      if (two) L.xr[1][512 + lane] = x1[0];
    }
  }
}

// ---------------------------------------------------------------------------
// MFMA formulation of IMDCT + matrixing (device build)
//
// v_mfma_f32_16x16x4_f32: lane l = (j = l & 15, kq = l >> 4) holds A[row j][k = kq], B[k = kq][col j] and
// D[row 4 kq + r][col j], r = 0..3; the result is a k-ordered fmaf chain, i.e. the same arithmetic as the
// scalar formulation's PD_FMA chain.
//
//   IMDCT      D[(ch, sb)][p] = sum_k xr[ch][18 sb + k] * cos_N36[k][p]; 4 row tiles (ch, h: sb = 16 h + ..),
//              3 column tiles {p = j | p = 18 + j | p = 16, 17, 34, 35}, 5 k-steps (k = 18, 19 are zero)
//   epilogue   window, overlap-add (overlap stays in D layout), frequency inversion -- in registers
//   matrixing  C[(ch, t)][n] = sum_sb hyb[ch][sb][t] * cos((2 sb + 1) n pi / 64), folded once more:
//              C[2m] = sum_{k<16} (x[k] + x[31-k]) cos((2k+1) 2m pi/64),  C[2m+1] = sum_{k<16} (x[k] - x[31-k]) cos((2k+1)(2m+1) pi/64).
//              The second IMDCT row tile of a channel holds its subbands REVERSED (row i = subband 31 - i), so x[k]
//              and x[31-k] sit in the same lane and register: the butterflies are lane-local adds and the epilogue's
//              registers ARE the A fragments (row = t = j, k = 4 kq + r arrives as k-step r) -- the hybrid output
//              never goes through LDS and the matrixing costs 8 instead of 16 MFMAs per row tile.
// ---------------------------------------------------------------------------
// (mfma16(a, b, c) and f32x4: top of the file)

// alias reduction in place (P:1706-1732): lane (ch, sb) owns the boundary below subband sb
PD_FN void ph_antialias(int lane, WaveData& L, BankPtr cb, bool only_first = false, const GranuleInfo* gi = nullptr) {   // only_first: boundary sb 0 | 1 alone
  const GranuleInfo g = gi ? *gi : granule_info(L);
  const int ch = lane >> 5, sb = lane & 31;
  if (ch >= g.nch || sb == 0) return;
  if (only_first && sb != 1) return;
  const bool shrt = g.is_short(ch), mixed = g.is_mixed(ch);
  if (shrt && !(mixed && sb == 1)) return;
  float* x = L.xr[ch];
  float lo[8], up[8];
  PD_UNROLL for (int i = 0; i < 8; i++) { lo[i] = x[18 * sb - 1 - i]; up[i] = x[18 * sb + i]; }
  PD_UNROLL for (int i = 0; i < 8; i++) {
    x[18 * sb - 1 - i] = lo[i] * cb->cs[i] - up[i] * cb->ca[i];   // lb, P:1725
    x[18 * sb + i] = up[i] * cb->cs[i] + lo[i] * cb->ca[i];       // ub, P:1726
  }
}

// The peek-only halo granule (run_chunk): all that is wanted from it are the three IMDCT tail values p = 18, 19, 20 of
// (channel 0, subband 0) -- the overlap that the next granule's hybrid output (ch 0, sb 0, t 0..2) adds, which is what the
// H5 scalefactor peek of the granule after that reads.  Lanes 0..2 = j: an 18-term dot product against column 18 + j of
// the same matrices the MFMA path uses (fragment layout: element (k, n) of column tile nt sits at [(k / 4) * 2 + nt][(k % 4) * 16 + n]),
// windowed like the epilogue of ph_mfma, left in the lane's overlap register of (ch 0, h 0, r 0).
PD_FN void ph_peek_tail(int lane, const WaveData& L, const TabLds& S, LaneRegs& R, const GlobalTables& T) {
  if (lane >= 3) return;
  const GranuleInfo g = granule_info(L);
  const bool shrt = g.is_short(0);
  const bool lowrow = (g.flags(0) & PDMP3_GC_WIN_SWITCH) && g.is_mixed(0);     // subbands 0, 1 of a mixed block: long transform, window 0
  const float* frag = (shrt && !lowrow) ? T.frag_short : T.frag_long;
  float y = 0.0f;
  PD_UNROLL for (int m = 0; m < 18; m++) y = PD_FMA(L.xr[0][m], frag[((m >> 2) * 2 + 1) * 64 + (m & 3) * 16 + lane], y);
  if (!(shrt && !lowrow)) y = y * S.win[lowrow ? 0 : g.block_type(0)][18 + lane];
  R.ovl[0] = y;
}

// The same for the granule kernel's H5 waves: lanes 0..2 = j get the IMDCT output p = j of (channel 0, subband 0) --
// windowed, WITHOUT the overlap (the caller adds the tail of the granule before) -- i.e. what ph_imdct leaves in
// y1[0] of lanes 0..2: the same fmaf chain over the same fragment values.
PD_FN float ph_peek_head(int lane, const WaveData& L, const TabLds& S, const GlobalTables& T) {
  const GranuleInfo g = granule_info(L);
  const bool shrt = g.is_short(0);
  const bool lowrow = (g.flags(0) & PDMP3_GC_WIN_SWITCH) && g.is_mixed(0);     // subbands 0, 1 of a mixed block: long transform, window 0
  const float* frag = (shrt && !lowrow) ? T.frag_short : T.frag_long;
  const int n = lane < 3 ? lane : 0;
  float y = 0.0f;
  PD_UNROLL for (int m = 0; m < 18; m++) y = PD_FMA(L.xr[0][m], frag[((m >> 2) * 2 + 0) * 64 + (m & 3) * 16 + n], y);
  if (!(shrt && !lowrow)) y = y * S.win[lowrow ? 0 : g.block_type(0)][n];
  return y;
}

template <bool DUMP>
PD_FN void ph_mfma(int lane, WaveData& L, const TabLds& S, LaneRegs& R, BankPtr cb, const GlobalTables& T, float* dump2, float* dump3,
                   bool do_matrix, const GranuleInfo* gi = nullptr) {   // do_matrix (wave-uniform) = false: a halo granule whose polyphase input nobody reads
  const GranuleInfo g = gi ? *gi : granule_info(L);
  const int j = lane & 15, kq = lane >> 4;
  if (DUMP) {   // stage 2 = lines after alias reduction
    PD_UNROLL for (int i = 0; i < 9; i++) {
      dump2[lane + 64 * i] = L.xr[0][lane + 64 * i];
      if (g.nch == 2) dump2[4 * 576 + lane + 64 * i] = L.xr[1][lane + 64 * i];
    }
  }
  const bool any_short = g.is_short(0) || (g.nch == 2 && g.is_short(1));
  // ---- IMDCT outputs p = 16, 17, 34, 35 on the VALU: lane = (cl, sb), scalar-broadcast coefficients.
  // (As a third MFMA column tile they would fill 4 of 16 columns.)  Must precede the hyb writes below:
  // hyb aliases xr.
  {
    const int cl = lane >> 5, sb = lane & 31;
    const bool act = cl < g.nch;
    float o16, o17;
    const bool lwsf = (g.flags(cl) & PDMP3_GC_WIN_SWITCH) != 0;
    const bool llow = lwsf && g.is_mixed(cl) && sb < 2;
    const bool lshort = g.is_short(cl) && !llow;
    const float* x = &L.xr[cl][18 * sb];
    float in[18];
    PD_UNROLL for (int m = 0; m < 18; m++) in[m] = x[m];
    float y[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    PD_UNROLL for (int m = 0; m < 18; m++)
      PD_UNROLL for (int q = 0; q < 4; q++) y[q] = PD_FMA(in[m], PD_C36(cb, c36x, q, m), y[q]);
    const float* w = S.win[llow ? 0 : g.block_type(cl)];
    y[0] = y[0] * w[16]; y[1] = y[1] * w[17]; y[2] = y[2] * w[34]; y[3] = y[3] * w[35];
    if (any_short) {                      // wave-uniform
      float ys[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      PD_UNROLL for (int m = 0; m < 18; m++)
        PD_UNROLL for (int q = 0; q < 4; q++) ys[q] = PD_FMA(in[m], PD_C36(cb, s36x, q, m), ys[q]);
      PD_UNROLL for (int q = 0; q < 4; q++) y[q] = lshort ? ys[q] : y[q];
    }
    o16 = y[0] + R.ovl[16]; o17 = y[1] + R.ovl[17];                       // P:1775
    R.ovl[16] = act ? y[2] : R.ovl[16];                                   // P:1776
    R.ovl[17] = act ? y[3] : R.ovl[17];
    if (sb & 1) o17 = -o17;                                               // P:1738-1746
    if (DUMP) { if (act) { dump3[cl * 4 * 576 + 18 * sb + 16] = o16; dump3[cl * 4 * 576 + 18 * sb + 17] = o17; } }
    // matrixing fold: x[k] +- x[31 - k]; subband 31 - sb of the same channel is lane ^ 31
    if (do_matrix) {
      const float p16 = PD_SHFL_XOR(o16, 31), p17 = PD_SHFL_XOR(o17, 31);
      if (sb < 16) {
        L.lo[0][2 * cl + 0][sb] = o16 + p16; L.lo[1][2 * cl + 0][sb] = o16 - p16;
        L.lo[0][2 * cl + 1][sb] = o17 + p17; L.lo[1][2 * cl + 1][sb] = o17 - p17;
      }
    }
  }
  // ---- channel 1 first: its matrixing output overwrites xr[1][18..] and nothing of xr[0];
  // channel 0's output then overwrites xr[0] and xr[1][0..17], both consumed by then.
  PD_UNROLL for (int cc = 0; cc < 2; cc++) {
    const int ch = 1 - cc;
    if (ch < g.nch) {                    // wave-uniform
      float outa[8];
      const bool shrt = g.is_short(ch);
      const bool wsf = (g.flags(ch) & PDMP3_GC_WIN_SWITCH) != 0;
      const bool mixrows = wsf && g.is_mixed(ch);        // subbands 0, 1 use window/transform 0 (P:1769-1771)
      const int bt = g.block_type(ch);
      float bfr[10];
      if (shrt) {
        // (pinned = waited for INSIDE this branch: the wait counters are per wave, not per path -- with the short blocks' loads
        //  still in flight where the paths meet, the long-block path too sat through `s_waitcnt vmcnt(9) .. vmcnt(0)` between
        //  its matrix instructions, i.e. waited for the prefetch of the next granule, asked for a moment before, every granule)
        PD_UNROLL for (int k = 0; k < 10; k++) bfr[k] = T.frag_short[k * 64 + lane];
        PD_UNROLL for (int k = 0; k < 10; k++) PD_PIN(bfr[k]);
      }
      else { PD_UNROLL for (int k = 0; k < 10; k++) bfr[k] = R.bi[k]; }
      f32x4 acc[2][2];
      float afr[2][5];
      PD_UNROLL for (int h = 0; h < 2; h++)
        PD_UNROLL for (int kk = 0; kk < 5; kk++) {
          const int k = 4 * kk + kq;
          afr[h][kk] = L.xr[ch][18 * (h ? 31 - j : j) + (k < 18 ? k : 0)];   // second row tile: subbands reversed
          if (k >= 18) afr[h][kk] = 0.0f;
        }
      PD_UNROLL for (int h = 0; h < 2; h++)
        PD_UNROLL for (int nt = 0; nt < 2; nt++) acc[h][nt] = mfma16(afr[h][0], bfr[nt], (f32x4){0, 0, 0, 0});
      PD_UNROLL for (int kk = 1; kk < 5; kk++)
        PD_UNROLL for (int h = 0; h < 2; h++)
          PD_UNROLL for (int nt = 0; nt < 2; nt++) acc[h][nt] = mfma16(afr[h][kk], bfr[kk * 2 + nt], acc[h][nt]);
      f32x4 accl[2];
      PD_UNROLL for (int nt = 0; nt < 2; nt++) accl[nt] = acc[0][nt];
      if (shrt && mixrows) {             // wave-uniform: rows sb 0, 1 of the first tile take the long transform
        PD_UNROLL for (int nt = 0; nt < 2; nt++) accl[nt] = mfma16(afr[0][0], R.bi[nt], (f32x4){0, 0, 0, 0});
        PD_UNROLL for (int kk = 1; kk < 5; kk++)
          PD_UNROLL for (int nt = 0; nt < 2; nt++) accl[nt] = mfma16(afr[0][kk], R.bi[kk * 2 + nt], accl[nt]);
      }
      // window factors of this lane's columns: t = j and p = 18 + j
      const float* wb = S.win[bt];
      const float* w0 = S.win[0];
      const float wb1 = wb[j], wb2 = wb[18 + j];
      const float w01 = w0[j], w02 = w0[18 + j];
      PD_UNROLL for (int h = 0; h < 2; h++)
        PD_UNROLL for (int r = 0; r < 4; r++) {
          const bool lowrow = mixrows && h == 0 && kq == 0 && r < 2;      // subbands 0, 1 (first tile, rows 0, 1)
          float y1 = acc[h][0][r], y2 = acc[h][1][r];
          if (h == 0 && lowrow) { y1 = accl[0][r]; y2 = accl[1][r]; }
          const bool win_folded = shrt && !lowrow;                        // short transform: window is in the matrix
          const float f1 = lowrow ? w01 : wb1, f2 = lowrow ? w02 : wb2;
          if (!win_folded) { y1 = y1 * f1; y2 = y2 * f2; }
          const int oi = ch * 8 + h * 4 + r;
          float o = y1 + R.ovl[oi];                                       // P:1775
          R.ovl[oi] = y2;                                                 // P:1776
          const int sb = h ? 31 - (4 * kq + r) : 4 * kq + r;              // subband of this row (tile 1 is reversed)
          if ((sb & 1) && (j & 1)) o = -o;                                // P:1738-1746: odd subband, odd sample
          outa[h * 4 + r] = o;
          if (DUMP) dump3[ch * 4 * 576 + 18 * sb + j] = o;
          if (ch == 0 && h == 0 && r == 0 && kq == 0 && j < 3) L.peek[j] = o;   // H5 source: (ch 0, sb 0, t 0..2)
        }
      // matrixing of time slots t = j (rows) of this channel: butterflies, then even / odd 16 x 16 products
      if (do_matrix) {
        f32x4 me = (f32x4){0, 0, 0, 0}, mo = (f32x4){0, 0, 0, 0};
        PD_UNROLL for (int r = 0; r < 4; r++) {
          const float a = outa[r] + outa[4 + r], b = outa[r] - outa[4 + r];   // x[k] +- x[31 - k], k = 4 kq + r
          me = mfma16(a, R.bm[r], me);
          mo = mfma16(b, R.bm[4 + r], mo);
        }
        PD_UNROLL for (int r = 0; r < 4; r++) {      // D rows = time slot 4 kq + r, cols: C[2 j] and C[2 j + 1]
          L.hyb[ch][4 * kq + r][2 * j] = me[r];
          L.hyb[ch][4 * kq + r][2 * j + 1] = mo[r];
        }
      }
    }
  }
  // ---- the four left-over time slots (rows: ch 0 t 16, ch 0 t 17, ch 1 t 16, ch 1 t 17) as one more row tile
  PD_WAVE_SYNC();                          // lo[] was written by other lanes
  if (do_matrix) {
    f32x4 me = (f32x4){0, 0, 0, 0}, mo = (f32x4){0, 0, 0, 0};
    PD_UNROLL for (int r = 0; r < 4; r++) {
      const float a = (j < 4) ? L.lo[0][j & 3][4 * kq + r] : 0.0f;
      const float b = (j < 4) ? L.lo[1][j & 3][4 * kq + r] : 0.0f;
      me = mfma16(a, R.bm[r], me);
      mo = mfma16(b, R.bm[4 + r], mo);
    }
    if (kq == 0) {
      PD_UNROLL for (int r = 0; r < 4; r++) {
        const int ch = r >> 1, t = 16 + (r & 1);
        if (ch < g.nch) { L.hyb[ch][t][2 * j] = me[r]; L.hyb[ch][t][2 * j + 1] = mo[r]; }
      }
    }
  }
}
// float -> int16 exactly as P:2028-2031 on x86-64 (cvttsd2si: out of range / NaN => INT32_MIN);
// the f64 product of a binary32 and 32767 is exact, so this is bit-for-bit the reference's conversion
PD_FN int pcm_from_sum(float sum) {
  const double d = (double)sum * 32767.0;
  int s;
  if (!(d > -2147483649.0 && d < 2147483648.0)) s = (int)0x80000000;
  else s = (int)d;
  if (s > 32767) s = 32767;
  else if (s < -32767) s = -32767;
  return s;
}

// The same conversion for the 18 samples of a lane without binary64 (half-rate on the VALU, 5 instructions per
// sample): truncation of the EXACT product sum * 32767 equals truncation of the binary32 product rounded TOWARD
// ZERO (no integer lies strictly between a value and its round-toward-zero neighbour, integers below 2^24 being
// representable), so the 18 multiplies run with the FP32 rounding mode switched to RTZ -- set and restored inside
// one asm statement, the compiler never sees another mode.  Then: clamp to +-32767 in binary32 (exact), convert,
// and the reference's wrap-around: a product at or beyond 2^31 (sum > 65538, the largest binary32 whose product
// is still below 2^31) or a NaN comes out of cvttsd2si as INT32_MIN and is then clipped to -32767.
// `wrap` (wave-uniform): some lane of the wave holds a sum beyond 65538 or a NaN -- the per-sample form of the wrap-around.
// Pure per lane: the host build tests it by itself against pcm_from_sum (tests/test_pipeline_emul.py).
PD_FN void pcm_convert18_lane(const float* sum, int* out, bool wrap) {
  float p[18];
  mul18_rtz_32767(sum, p);
  if (wrap) {
    PD_UNROLL for (int t = 0; t < 18; t++) {
      const int n = (int)PD_FMED3(p[t], -32767.0f, 32767.0f);
      out[t] = (sum[t] <= 65538.0f) ? n : -32767;
    }
  } else {
    // (a NaN sum gives a NaN product, which v_med3 turns into -32767 by itself -- its NaN rule is min3: the same result
    //  as the branch above, whatever the wave-wide test saw)
    PD_UNROLL for (int t = 0; t < 18; t++) out[t] = (int)PD_FMED3(p[t], -32767.0f, 32767.0f);
  }
}
// The wrap-around is a property of signals driven 65 000 x past full scale: one test for the wave (the largest of the
// 18 sums of every lane) instead of a compare and a select per sample.  fmax passes over NaNs: a NaN among finite sums
// does not raise `wrap` -- and need not: both forms of pcm_convert18_lane give -32767 for it (the host test checks
// exactly that, lane by lane, against pcm_from_sum).  Called by every lane of the wave (idle ones: act = false).
PD_FN bool pcm_wave_wraps(const float* sum, bool act) {
  bool big = false;
  if (act) {
    float mx = sum[0];
    PD_UNROLL for (int t = 1; t < 18; t++) mx = __builtin_fmaxf(mx, sum[t]);
    big = !(mx <= 65538.0f);
  }
  return PD_ANY(big);
}

// the 18 sums of a lane (P:2028) -> PCM of the granule: conversion, channel pairing, stores
// ALL: a stereo granule -- every lane active, nch = 2: no execution masks anywhere
template <bool F32, bool ALL = false>
PD_FN void pcm_emit(int lane, WaveData& L, int nch_arg, bool act_arg, const float* sum, int16_t* pcm_g, float* pcmf_g) {
  const int i = lane & 31;
  const int nch = ALL ? 2 : nch_arg;
  const bool act = ALL ? true : act_arg;
  int out[18];
  const bool wrap = F32 ? false : pcm_wave_wraps(sum, act);
  if (act) {
    if (F32) {
      // float PCM: the binary32 `sum` of P:2028 itself, before the scaling to int16
      PD_UNROLL for (int t = 0; t < 18; t++) out[t] = (int)f2u(sum[t]);
    } else pcm_convert18_lane(sum, out, wrap);
  }
  if (F32) {
    if (nch == 2) {
      // same pairing as below; a lane then owns one interleaved sample-frame of two floats: 8-byte stores,
      // 64 consecutive sample-frames per instruction
      PD_UNROLL for (int q = 0; q < 9; q++) {
        int a = out[2 * q], b = out[2 * q + 1];
        permlane32_swap(a, b);
        reinterpret_cast<f32x2*>(pcmf_g)[lane + 64 * q] = (f32x2){u2f((uint32_t)a), u2f((uint32_t)b)};
      }
    } else if (act) {
      PD_UNROLL for (int t = 0; t < 18; t++) pcmf_g[32 * t + i] = u2f((uint32_t)out[t]);
    }
    return;
  }
  if (nch == 2) {
    // Lanes i and i + 32 hold the left and the right value of the same sample.  For a pair of time slots (t, t + 1) one
    // half exchange leaves lane i with L, R of sample (t, i) and lane i + 32 with L, R of sample (t + 1, i): every lane
    // owns one interleaved sample-frame (P:2032-2041, P:2307-2345), lane l the dword 32 t + l of the granule.
    // 9 stores of one dword per lane, 64 consecutive dwords per instruction.
    uint32_t* dst = reinterpret_cast<uint32_t*>(pcm_g);
    PD_UNROLL for (int q = 0; q < 9; q++) {
      int a = out[2 * q], b = out[2 * q + 1];
      permlane32_swap(a, b);
      dst[64 * q + lane] = ((unsigned)a & 0xffffu) | ((unsigned)b << 16);
    }
  } else {
    // mono (pcm aliases nothing live: spec is dead since ph_requant; hyb reads above are done)
    if (act) { PD_UNROLL for (int t = 0; t < 18; t++) L.pcm[t * 32 + i] = (int16_t)out[t]; }
  }
}

// The window sums of a granule (P:2015-2026) and its PCM.  sum[t] = sum_k we[k] E[15 + t - 2 k] + wo[k] O[14 + t - 2 k], the 16
// FMAs in this order (k ascending, we before wo), where E[s], O[s] are the lane's two DCT coefficients of slot s: s = 0..14
// the history (slots 3..17 of the granule before), s = 15..32 this granule's slots t = 0..17.  Two consecutive sums share
// every coefficient and read neighbouring slots, so the phase is written on PAIRS: Ep[j] = (E[2 j + 1], E[2 j + 2]),
// Op[j] = (O[2 j], O[2 j + 1]), sum pair q = (sum[2 q], sum[2 q + 1]) = sum_k we[k] Ep[7 + q - k] + wo[k] Op[7 + q - k] --
// one v_pk_fma_f32 per two FMAs, every operand an aligned register pair BY CONSTRUCTION: the pairs of this granule's slots are
// loaded as pairs (ds_read2_b32), and the history of the next granule is pairs 9..15 as they are (E[18 + s] = next E[s]: the
// pairing survives the shift by 18), so nothing is moved to form an operand.  (Left to the SLP vectoriser the same FMAs came
// out packed too, but with ~40 register moves per granule to re-pair O, whose natural load pairs (t, t + 1) start at the
// wrong parity.)  E[0] is never read (the oldest slot enters with wo only); it stays in the state for its layout's sake.
// ALL: a stereo granule (every lane active).
template <bool F32, bool ALL>
PD_FN void ph_window_emit(int lane, WaveData& L, LaneRegs& R, int nch, int16_t* pcm_g, float* pcmf_g) {
  const int ch = lane >> 5;
  const bool act = ALL || ch < nch;            // (mono: lanes 32..63 idle, their history is channel 1's and stays)
  float sum[18];
  if (act) {
    const float* he = &L.hyb[ch][0][R.idx_e];  // slot t at he[33 t]
    const float* ho = &L.hyb[ch][0][R.idx_o];
    f32x2 Ep[16], Op[16];
    PD_UNROLL for (int j = 0; j < 7; j++) { Ep[j] = (f32x2){R.he[2 * j + 1], R.he[2 * j + 2]}; Op[j] = (f32x2){R.ho[2 * j], R.ho[2 * j + 1]}; }
    // (the compiler merges loads off one base register into ds_read2_b32 in the order of their offsets: (0, 1), (2, 3), ...
    //  O's pairs start at an odd slot, so its slot 0 is read through a base of its own -- else every register of O is moved)
    int io0 = R.idx_o;
    PD_PIN(io0);
    // three sums pairs at a time: a v_pk_fma_f32 that reads the result of the one before it costs a wait state
#define PD_WIN_LOAD(j0, j1) \
    PD_UNROLL for (int j = (j0); j < (j1); j++) { \
      Ep[j] = (f32x2){he[33 * (2 * j - 14)], he[33 * (2 * j - 13)]}; \
      if (j >= 8) Op[j] = (f32x2){ho[33 * (2 * j - 15)], ho[33 * (2 * j - 14)]}; \
    }
#define PD_WIN_SUMS(q0) { \
      f32x2 acc[3] = {(f32x2){0.0f, 0.0f}, (f32x2){0.0f, 0.0f}, (f32x2){0.0f, 0.0f}}; \
      PD_UNROLL for (int k = 0; k < 8; k++) { \
        PD_UNROLL for (int u = 0; u < 3; u++) acc[u] = fma2((f32x2){R.we[k], R.we[k]}, Ep[7 + (q0) + u - k], acc[u]); \
        PD_UNROLL for (int u = 0; u < 3; u++) acc[u] = fma2((f32x2){R.wo[k], R.wo[k]}, Op[7 + (q0) + u - k], acc[u]); \
      } \
      PD_UNROLL for (int u = 0; u < 3; u++) { \
        if (F32 && ALL) { \
          /* float PCM of a stereo granule: a pair of sums is a pair of time slots -- stored as soon as it is there (the \
             eighteen sums held until the end were 20 spilled registers in this kernel) */ \
          int a_ = (int)f2u(acc[u][0]), b_ = (int)f2u(acc[u][1]); \
          permlane32_swap(a_, b_); \
          /* (one base + a constant offset per pair, one 8-byte store: the granule's PCM is 16-byte aligned by contract) */ \
          reinterpret_cast<f32x2*>(pcmf_g)[lane + 64 * ((q0) + u)] = (f32x2){u2f((uint32_t)a_), u2f((uint32_t)b_)}; \
        } else { sum[2 * ((q0) + u)] = acc[u][0]; sum[2 * ((q0) + u) + 1] = acc[u][1]; } \
      } \
    }
    PD_WIN_LOAD(7, 16)
    const float o17 = ho[33 * 17];
    Op[7] = (f32x2){R.ho[14], L.hyb[ch][0][io0]};
    PD_WIN_SUMS(0)
    PD_WIN_SUMS(3)
    PD_WIN_SUMS(6)
#undef PD_WIN_LOAD
#undef PD_WIN_SUMS
    R.he[0] = Ep[8][1];
    PD_UNROLL for (int j = 0; j < 7; j++) {
      R.he[2 * j + 1] = Ep[j + 9][0]; R.he[2 * j + 2] = Ep[j + 9][1];
      R.ho[2 * j] = Op[j + 9][0]; R.ho[2 * j + 1] = Op[j + 9][1];
    }
    R.ho[14] = o17;
  }
  if (!(F32 && ALL)) pcm_emit<F32, ALL>(lane, L, nch, act, sum, pcm_g, pcmf_g);
}

// full = false (wave-uniform): the last halo granule -- only its slots 3..17 are wanted, as the next granule's history
template <bool F32>
PD_FN void ph_window(int lane, WaveData& L, LaneRegs& R, bool full, int nch, int16_t* pcm_g, float* pcmf_g) {
  // (nch is a parameter: by now the side records in LDS are the NEXT granule's, see run_chunk)
  if (!full) {
    const int ch = lane >> 5;
    if (ch < nch) {
      PD_UNROLL for (int s = 0; s < kHistSlots; s++) {
        R.he[s] = L.hyb[ch][3 + s][R.idx_e];
        R.ho[s] = L.hyb[ch][3 + s][R.idx_o];
      }
    }
    return;
  }
  if (nch == 2) ph_window_emit<F32, true>(lane, L, R, 2, pcm_g, pcmf_g);      // wave-uniform
  else ph_window_emit<F32, false>(lane, L, R, nch, pcm_g, pcmf_g);
}

// PCM of a mono granule (576 samples = 1152 bytes) from the LDS staging buffer; stereo granules were stored by
// ph_window straight from registers.
PD_FN void ph_store(int lane, WaveData& L, int nch, int16_t* pcm_g, bool emit) {
  if (!emit) return;
  if (nch != 2) {
    const Chunk16* src = reinterpret_cast<const Chunk16*>(L.pcm);
    Chunk16* dst = reinterpret_cast<Chunk16*>(pcm_g);
    for (int c = lane; c < 72; c += 64) dst[c] = src[c];
  }
}

// Highest frame in [f_lo, f_hi) that has two channels or carries the RESET flag, or -1.  Wave-uniform.
PD_FN int last_stereo_or_reset(const pdmp3_gc_side* side, int f_lo, int f_hi) {
  // 256 frames per step: four independent byte loads per lane in flight (a step costs one memory round trip; an
  // all-mono batch makes its last chunk walk the whole batch, so the steps had better be few)
  const int lane = PD_LANE();
  for (int base = f_hi - 256;; base -= 256) {
    bool hit[4];
    PD_UNROLL for (int q = 0; q < 4; q++) {
      const int f = base + 64 * q + lane;
      hit[q] = false;
      if (f >= f_lo && f < f_hi) {
        const uint8_t b = reinterpret_cast<const uint8_t*>(side + (size_t)f * 4)[7];
        hit[q] = ((b & PDMP3_FR_MODE_MASK) >> PDMP3_FR_MODE_SHIFT) != 3 || (b & PDMP3_FR_RESET);
      }
    }
    PD_UNROLL for (int q = 3; q >= 0; q--) {
      const unsigned long long m = PD_BALLOT(hit[q]);
      if (m) return base + 64 * q + 63 - __builtin_clzll(m);
    }
    if (base <= f_lo) return -1;
  }
}

// ---------------------------------------------------------------------------
// One chunk = one wavefront.
// ---------------------------------------------------------------------------
struct DecodeArgs;
PD_FN bool chunk_is_rare(const DecodeArgs& a, int chunk);      // (below)
struct DecodeArgs {
  const int16_t* spectra;        // [n_frames][2][2][576]
  const pdmp3_gc_side* side;     // [n_frames][2][2]
  int16_t* pcm;                  // 2304 int16 per frame
  float* pcm_f32;                // F32 kernels: 2304 floats per frame instead (the sums of P:2028, unscaled)
  const float* state_in;         // kStateFloats or null (read by chunk 0)
  float* state_out;              // kStateFloats or null (written by the last chunk; must not alias state_in
                                 // when there is more than one chunk)
  float* stages;                 // [n_frames][2][2][4][576] or null (DUMP builds)
  int n_frames;
  int chunk_frames;
  unsigned long long* prof;      // PROF builds: [n_chunks][kProfSlots] shader-clock ticks per phase
  // granule kernel (run_granule): chain_state = [2 n_frames][kGranFloats], chain_flag = two flags per granule
  // (== chain_epoch once that part of the granule's state is there)
  float* chain_state;
  unsigned* chain_flag;
  unsigned chain_epoch;          // 0: independent chunks (halo), as described above
  unsigned debug_flags;          // tests: PD_DEBUG_FAR_TIMEOUT = every wait for another workgroup gives up at once
  int sf_hint;                   // the sampling frequency the workgroups' line tables are loaded for
  int n_gran;                    // granules in the batch when that is not 2 n_frames (0 = it is): an LSF launch of an ODD number of
                                 // frames -- its last record-frame holds one (engine.hip k_lsf_pair; chunk kernels only)
};
constexpr unsigned PD_DEBUG_FAR_TIMEOUT = 1u;

// Does the chunk -- its frames, its halo (four frames back cover the three granules of the longest one) and, when the
// frame before it is mono, the far stereo frames run_chunk's pre-halo goes back to -- hold a frame_is_rare() frame?  Then
// the wave runs the RARE copy of run_chunk.  Wave-uniform; one byte pair per lane and 64 frames.
PD_FN bool chunk_is_rare(const DecodeArgs& a, int chunk) {
  const int lane = PD_LANE();
  if (a.n_gran > 0) return true;                                  // an LSF launch
  const int f0 = chunk * a.chunk_frames;
  int f1 = f0 + a.chunk_frames;
  if (f1 > a.n_frames) f1 = a.n_frames;
  bool rare = false;
  for (int base = f0 - 4 < 0 ? 0 : f0 - 4; base < f1; base += 64) {
    const int f = base + lane;
    rare = rare || (f < f1 && frame_is_rare(a.side + (size_t)f * 4));
  }
  if (PD_ANY(rare)) return true;
  if (f0 > 0) {
    const uint8_t fb = reinterpret_cast<const uint8_t*>(a.side + (size_t)(f0 - 1) * 4)[7];
    if (((fb & PDMP3_FR_MODE_MASK) >> PDMP3_FR_MODE_SHIFT) == 3 && !(fb & PDMP3_FR_RESET)) {
      const int fs = last_stereo_or_reset(a.side, 0, f0 - 1);     // (where the pre-halo starts: run_chunk)
      if (fs >= 0 && (frame_is_rare(a.side + (size_t)fs * 4) || (fs > 0 && frame_is_rare(a.side + (size_t)(fs - 1) * 4)))) return true;
    }
  }
  return false;
}

constexpr int kProfSlots = 12;

// PD_PHASE: a phase body, then the wave-level fence that orders its LDS traffic against the next phase's
// (device: compiler-level only, see PD_WAVE_SYNC; host test build: every fiber of the wave arrives).
#define PD_PHASE(...)                                   \
  {                                                     \
    __VA_ARGS__;                                        \
  }                                                     \
  PD_WAVE_SYNC();

struct GranPos;
PD_FN void gran_publish_regs(int lane, const LaneRegs& R, const DecodeArgs& a, int g, const GranPos& gp);   // (below)

// OWN_TABS: S is this wave's own table block -- filled here, and refilled when the stream changes its sampling
// frequency; otherwise (granule kernel: S belongs to the workgroup and is there already) granules of another sampling
// frequency read the global line table (ph_requant's TG).  gp: granule kernel only -- where the closing state goes.
// state_only: decode nothing -- only derive the state at the START of the chunk (its halo), into *state_only.
template <bool DUMP, bool PROF = false, bool F32 = false, bool OWN_TABS = true, bool RARE = true>
PD_FN void run_chunk(const DecodeArgs& a, const GlobalTables& T, BankPtr cb, int chunk, WaveData& L, TabLds& S,
                     const GranPos* gp = nullptr, LaneRegs* state_only = nullptr) {
  LaneRegs R;
  const int lane = PD_LANE();
  const unsigned long long t_wave_start = PROF ? PD_CLOCK() : 0ull;
  const int f0 = chunk * a.chunk_frames;
  int f1 = state_only ? f0 : f0 + a.chunk_frames;
  if (f1 > a.n_frames) f1 = a.n_frames;
  const int g_begin = 2 * f0;
  int g_end = 2 * f1;
  if (a.n_gran > 0 && g_end > a.n_gran) g_end = a.n_gran;
  int g_start = 0;
  // the third halo granule is there for ONE thing, the H5 peek two granules later: it is decoded "peek-only"
  // (26 lines, one subband, three IMDCT outputs; ph_peek_tail) unless it is also granule 0 of the batch, whose
  // state is the caller's
  int g_peek = -1;
  if (g_begin > 0) {
    // channel 1 of the granule just before the chunk: flags byte of its side record (wave-uniform)
    const uint8_t fl = reinterpret_cast<const uint8_t*>(a.side + (size_t)(g_begin - 1) * 2 + 1)[3];
    const bool shrt = (fl & PDMP3_GC_WIN_SWITCH) && ((fl & PDMP3_GC_BLOCK_TYPE_MASK) >> PDMP3_GC_BLOCK_TYPE_SHIFT) == 2;
    g_start = g_begin - (shrt ? kHaloGranulesH5 : kHaloGranules);
    if (shrt && g_start > 0) g_peek = g_start;
    if (g_start < 0) g_start = 0;
  }
  const bool last = (f1 == a.n_frames) && !state_only;
  // Channel 1 across mono frames.  The halo re-derives a channel's state from the last two granules in which
  // the channel was decoded; mono frames leave channel 1's overlap and polyphase history as the last stereo
  // frame left them (the reference's store[ch] / v_vec[ch], P:1777, P:2126), however long ago that was.  So
  // when the frame before the chunk is mono and channel 1 matters here (a stereo frame in the chunk, or this
  // chunk writes state_out), first run a PRE-halo at the last stereo frame before it: its two granules plus
  // the one in front (H5, as for the ordinary halo).  Channel 0 comes out of that wrong and is then
  // re-derived by the ordinary halo, which does not touch channel 1.  No stereo frame back to the start of
  // the batch: channel 1 is the caller's state_in (or zero); a RESET frame: zero.
  int pre_end = 0, npre = 0;
  int g_keep = -1;           // merged pre-halo: the last granule that decodes channel 1 (its history must be kept)
  bool ch1_from_state = false;
  if (g_start > 0) {
    const uint8_t fb = reinterpret_cast<const uint8_t*>(a.side + (size_t)(f0 - 1) * 4)[7];
    const bool prev_mono = ((fb & PDMP3_FR_MODE_MASK) >> PDMP3_FR_MODE_SHIFT) == 3 && !(fb & PDMP3_FR_RESET);
    if (prev_mono && ((last && a.state_out) || state_only || last_stereo_or_reset(a.side, f0, f1) >= 0)) {
      const int fs = last_stereo_or_reset(a.side, 0, f0 - 1);
      if (fs < 0) ch1_from_state = true;
      else {
        const uint8_t sb = reinterpret_cast<const uint8_t*>(a.side + (size_t)fs * 4)[7];
        if (((sb & PDMP3_FR_MODE_MASK) >> PDMP3_FR_MODE_SHIFT) != 3) {     // (a mono RESET frame: zero, nothing to do)
          pre_end = 2 * fs + 2;
          npre = (fs > 0 && !(sb & PDMP3_FR_RESET)) ? 3 : 2;
          if (npre == 3 && pre_end - 3 > 0) g_peek = pre_end - 3;
          if (pre_end >= g_start) { g_start = pre_end - npre; g_keep = pre_end - 1; npre = 0; pre_end = 0; }   // touches the ordinary halo: one run
        }
      }
    }
  }
  const int g_first = npre ? pre_end - npre : g_start;
  const bool from_stream_start = (g_first == 0);     // exact state: the caller's (or zero), no halo needed
  // the line tables of the first granule's sampling rate are fetched together with its spectra (one memory round
  // trip instead of two before the first granule can start)
  int cur_sfreq = reinterpret_cast<const uint8_t*>(a.side + (size_t)g_first * 2)[7] & PDMP3_FR_SFREQ_MASK;
  if (cur_sfreq > 2) cur_sfreq = 2;
  if (RARE) {
    int v = reinterpret_cast<const uint8_t*>(a.side + (size_t)g_first * 2)[offsetof(pdmp3_gc_side, lsf)] & PDMP3_LSF_VERSION_MASK;
    cur_sfreq += 3 * (v > 2 ? 2 : v);
  }

  PD_PHASE(
    ph_prefetch(lane, R, a.spectra + (size_t)g_first * 1152, a.side + (size_t)g_first * 2);
    if (OWN_TABS) load_linetab(lane, S, T, cur_sfreq);
    lane_init(lane, L, R, cb, T);
    if (a.state_in) {
      if (from_stream_start) state_load(lane, R, a.state_in);
      else if (ch1_from_state) state_load_ch1(lane, R, a.state_in);
    }
  )
  PD_PHASE(ph_commit(lane, L, R))
  // The band scales of a granule depend on its side info alone, which is in LDS one granule ahead: they are computed
  // in the store phase of the granule before (here: of nothing), not in a phase of their own.
  PD_PHASE(ph_scales(lane, L))
  unsigned long long acc[kProfSlots] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = PD_CLOCK();
  acc[11] = tprev;                 // end of the per-wave setup
#define PD_TICK(k) if (PROF) { const unsigned long long tn_ = PD_CLOCK(); acc[k] += tn_ - tprev; tprev = tn_; }
  // k < 0: the pre-halo granules pre_end + k; k >= 0: granule g_start + k (ordinary halo, then the chunk)
  for (int k = -npre; g_start + k < g_end; ++k) {
    const int g = k < 0 ? pre_end + k : g_start + k;
    const int g_next = k + 1 < 0 ? pre_end + k + 1 : g_start + k + 1;
    const int f = g >> 1, gr = g & 1;
    // Halo granules are decoded only for what they leave behind: the IMDCT tails (every one of them) and the
    // polyphase history, which is slots 3..17 of the LAST granule before a stretch that is emitted -- or, for the
    // pre-halo, before channel 1 goes quiet.  Their own PCM, and the matrixing of the earlier ones, is never read.
    const bool emit = (k >= 0) && (g >= g_begin);
    const bool feeds_next = (k == -1) || (k >= 0 && (g == g_begin - 1 || g == g_keep));
    PD_LAUNDER(cb);
    // the granule's facts (wave-uniform, scalar registers), read ONCE here for all its phases; the commit phase below
    // overwrites the side records with the next granule's
    const GranuleInfo gi = granule_info<RARE>(L);
    if (OWN_TABS && gi.sfreq != cur_sfreq) {      // wave-uniform: first granule, or the stream changed sampling rate
      PD_PHASE(load_linetab(lane, S, T, gi.sfreq))
      cur_sfreq = gi.sfreq;
    }
    PD_TICK(0)
    const bool reset_here = (gr == 0 || gi.ver) && (gi.fr & PDMP3_FR_RESET);     // (an LSF launch: every granule is a frame)
    const int nch_g = gi.nch;
    PD_TICK(1)
    float* dmp = DUMP ? a.stages + ((size_t)f * 16 + gr * 8) * 576 : nullptr;
    if (g == g_peek) {                     // wave-uniform: lines 0..63, boundary sb 0 | 1, three IMDCT outputs
      PD_PHASE(if (reset_here) state_zero(lane, R); ph_requant<false, 1, false, !OWN_TABS, false, RARE>(lane, L, S, cb, T, nullptr, nullptr, &gi))
      PD_TICK(2)
      PD_PHASE(
        if (g_next < g_end) ph_prefetch(lane, R, a.spectra + (size_t)g_next * 1152, a.side + (size_t)g_next * 2);
        ph_antialias(lane, L, cb, true, &gi);
      )
      PD_TICK(3)
      PD_PHASE(ph_peek_tail(lane, L, S, R, T))
      PD_TICK(4)
    } else {
      PD_SETPRIO(PD_PRIO_REQUANT);
      PD_PHASE(if (reset_here) state_zero(lane, R); if (!(PD_EXP_SKIP & 1)) ph_requant<DUMP, 9, PD_CHUNK_FAST_REQUANT != 0, !OWN_TABS, false, RARE>(lane, L, S, cb, T, dmp, dmp + 576, &gi))
      PD_TICK(2)
      PD_SETPRIO(PD_PRIO_AA);
      PD_PHASE(
        // the next granule's HBM reads fly during this granule's transforms
        if (g_next < g_end) ph_prefetch(lane, R, a.spectra + (size_t)g_next * 1152, a.side + (size_t)g_next * 2);
        if (!(PD_EXP_SKIP & 2)) ph_antialias(lane, L, cb, false, &gi);
      )
      PD_TICK(3)
      PD_SETPRIO(PD_PRIO_MFMA);
      PD_PHASE(if (!(PD_EXP_SKIP & 4)) ph_mfma<DUMP>(lane, L, S, R, cb, T, dmp + 2 * 576, dmp + 3 * 576, emit || feeds_next, &gi))
      PD_TICK(4)
    }
    PD_TICK(5)
    // The next granule is committed to LDS BEFORE this granule's PCM stores are issued (ph_window issues them): its
    // prefetch loads are older than those stores, so waiting for them never waits for a store.  (With the commit after
    // the window the wave sat through the acknowledgement of its nine stores every granule: 10 % of the loop.)
    PD_SETPRIO(PD_PRIO_SCALES);
    if (g_next < g_end) {
      PD_PHASE(ph_commit(lane, L, R))
      if (!PD_SCALES_WITH_WINDOW) { PD_PHASE(if (!(PD_EXP_SKIP & 8)) ph_scales(lane, L)) }    // next granule's (its side info was committed just above)
    }
    PD_TICK(6)
    // (PD_SCALES_WITH_WINDOW: the scales read the side records, the window reads hyb and writes PCM -- nothing of one is
    //  the other's, so they share a phase and their LDS round trips run side by side)
    if ((emit || feeds_next) && !(PD_EXP_SKIP & 16)) {
      PD_SETPRIO(PD_PRIO_WINDOW);
      PD_PHASE(if (PD_SCALES_WITH_WINDOW && g_next < g_end) ph_scales(lane, L);
               ph_window<F32>(lane, L, R, emit, nch_g, a.pcm + (size_t)f * 2304 + gr * 576 * nch_g,
                                F32 ? a.pcm_f32 + (size_t)f * 2304 + gr * 576 * nch_g : nullptr))
    } else if (PD_SCALES_WITH_WINDOW && g_next < g_end) {
      PD_PHASE(ph_scales(lane, L))
    }
    PD_SETPRIO(PD_PRIO_REST);
    PD_PHASE(if (!F32) ph_store(lane, L, nch_g, a.pcm + (size_t)f * 2304 + gr * 576 * nch_g, emit))
    PD_TICK(7)
  }
#undef PD_TICK
  if (PROF) {
    PD_PHASE(
      if (lane == 0) {
        for (int k = 0; k < 8; k++) a.prof[(size_t)chunk * kProfSlots + k] = acc[k];
        a.prof[(size_t)chunk * kProfSlots + 8] = (unsigned long long)(g_end - g_start + npre);
        a.prof[(size_t)chunk * kProfSlots + 9] = PD_CLOCK();
        a.prof[(size_t)chunk * kProfSlots + 10] = t_wave_start;
        a.prof[(size_t)chunk * kProfSlots + 11] = acc[11];
      }
    )
  }
  if (last && a.state_out) {
    PD_PHASE(state_store(lane, R, a.state_out))
  }
  if (state_only) {
    PD_UNROLL for (int m = 0; m < kOvlRegs; m++) state_only->ovl[m] = R.ovl[m];
    PD_UNROLL for (int k = 0; k < kHistSlots; k++) { state_only->he[k] = R.he[k]; state_only->ho[k] = R.ho[k]; }
    return;
  }
  if (!DUMP && !PROF && gp) {
    // granule kernel, a frame that took this path: the wave of the next frame may be waiting for the state it leaves
    // (stereo frames publish theirs; after a mono frame nobody takes anything from the chain)
    // (... nor after a frame_is_rare() frame: the frames around it treat it like a mono frame, run_granule_wave)
    const uint8_t fbl = reinterpret_cast<const uint8_t*>(a.side + (size_t)(f1 - 1) * 4)[7];
    if (((fbl & PDMP3_FR_MODE_MASK) >> PDMP3_FR_MODE_SHIFT) != 3 && !frame_is_rare(a.side + (size_t)(f1 - 1) * 4)) {
      PD_PHASE(gran_publish_regs(lane, R, a, 2 * f1 - 1, *gp))
    }
  }
}

// ---------------------------------------------------------------------------
// Granule kernel: ONE GRANULE PER WAVE, no loop, no halo (run_granule).
//
// What granule g needs from granule g - 1 is (i) its IMDCT tails -- a function of g - 1's own spectra -- and (ii) slots
// 3..17 of its matrixing output -- a function of its spectra and of the tails of g - 2.  Neither hangs on a chain:
//   a  spectra -> requantise -> alias reduction -> IMDCT, windowed: y1 (first halves), y2 (tails)      [own data only]
//   b  tails y2 -> chain_state[g], flag A
//   c  wait for flag A of g - 1 (published at ITS step b), overlap-add, frequency inversion, matrixing  -> hyb in LDS
//   d  matrixing rows 3..17 (+ the three hybrid outputs an H5 granule after it peeks at) -> chain_state[g], flag B
//   e  the window sums as far as they read this granule's own slots (registers)
//   f  wait for flag B of g - 1, the rest of the sums, PCM
// A wave is straight-line code whose state arrives late and leaves early: no loop-carried registers, no prefetch
// registers, no spills -- 128 VGPRs, four waves per SIMD, and a launch of N frames is 2 N waves that are ALL in flight
// up to 2048 frames (MI355X), every SIMD with four instruction streams to pick from instead of two.
// Same operations in the same order as run_chunk: PCM and carried state bit-identical (tests compare).
// Workgroup = WPW waves = WPW consecutive granules; its tables (TabLds) are shared.  Inside a workgroup the flags are
// in LDS and the state goes through ordinary stores / loads (one CU: one L1, one L2); the workgroup's last wave
// publishes with device scope for the first wave of the next one.  Only that first wave ever waits for ANOTHER
// workgroup, and that wait is BOUNDED: if the state does not come (the workgroup before it is not resident -- a
// partitioned or shared device, a dispatcher that does not go in order), the wave gives up, decodes its frame the
// independent way (run_chunk: halo) and publishes from there.  A launch therefore finishes whatever the residency and
// the dispatch order; waiting costs time, never progress.  (A ticket counter handing out the places in start order
// gave the same guarantee but serialised the launch's 512 workgroups on one atomic: +4 us on the C2 batch.)
// Mono frames, and stereo frames right after mono ones (channel 1's state is further back, run_chunk's pre-halo finds
// it): the wave of granule 0 runs run_chunk on the frame and publishes at its end, the wave of granule 1 leaves.
// ---------------------------------------------------------------------------
constexpr int kGranFloats = 64 * (kOvlRegs + kHistSlots);          // tails [18][64] | rows [15][64]
// Hand-over INSIDE a workgroup goes through LDS, into the receiver's own block ("mailboxes"), at no cost in space:
//   tails [18][64]  -> the receiver's xr, dead from the end of its ph_imdct until its ph_overlap_matrix writes hyb there
//   rows  [15][64]  -> the receiver's spec | pcm | side | scale (4224 B in a row), dead from the end of its ph_imdct on
// The receiver says when the space is free (GranMb::free, after its ph_imdct), the sender then writes and says so
// (tails_full / rows_full).  A sender waits for `free` without bound: it depends on the receiver's own progress only
// (its transforms up to the IMDCT need nothing from anybody).  LDS operations of a wave execute in order, so the
// flag follows the data.  Only the workgroup's last wave publishes through memory -- chain_state[g], device scope,
// flags in chain_flag -- and only its first wave reads that.
struct GranMb {                       // LDS, one per wave of the workgroup, zero at entry
  unsigned free, tails_full, rows_full;
  unsigned peek_full;                 // H5 frames: the three tail values below are there (from the wave of the frame's first granule)
  float peek_tail[3];
  unsigned pad;
};
struct GranPos {
  WaveData* wl;          // the workgroup's per-wave LDS blocks
  GranMb* mb;            // [wpw]
  int w;                 // place of the granule within its workgroup
  int wpw;
  unsigned* tabs_ready;  // LDS: waves of the workgroup that have stored their part of the tables
  // persistent form (run_granule_ring): the workgroup's waves go round a RANGE of granules [g_base, g_end), wave w taking
  // g_base + w, + wpw, ...; the mailboxes are a ring (place wpw - 1 hands on to place 0) and their flags carry the
  // receiving granule's number in the range + 1 instead of 0 / 1; nothing is handed on beyond g_end
  int ring;              // 0: one granule per wave, flags 0 -> 1 (k_decode_g)
  int g_base, g_end;
};
PD_FN int gran_next_place(const GranPos& gp) { return (gp.ring && gp.w + 1 == gp.wpw) ? 0 : gp.w + 1; }
PD_FN unsigned gran_tag(const GranPos& gp, int g) { return gp.ring ? (unsigned)(g - gp.g_base + 1) : 1u; }
static_assert(sizeof(WaveData::xr) >= kOvlRegs * 64 * sizeof(float), "tails mailbox");
static_assert(offsetof(WaveData, scale) + sizeof(WaveData::scale) - offsetof(WaveData, spec) >= kHistSlots * 64 * sizeof(float) &&
              offsetof(WaveData, xr) >= offsetof(WaveData, scale) + sizeof(WaveData::scale), "rows mailbox");
PD_FN float* gran_tails_box(WaveData& L) { return &L.xr[0][0]; }
PD_FN float* gran_rows_box(WaveData& L) { return reinterpret_cast<float*>(&L.spec[0][0]); }
PD_FN void gran_lds_flag(int lane, unsigned* p) { if (lane == 0) PD_LDS_FLAG(p) = 1u; }
// the workgroup's tables are complete once all its waves have stored their part (GranPos::tabs_ready, counted up by each wave
// after its stores -- the kernel has no barrier behind the table loads: they pass under the waves' first phases)
PD_FN void gran_tabs_wait(const GranPos& gp) {
  while ((int)PD_UNIFORM(PD_LDS_FLAG(gp.tabs_ready)) < gp.wpw) PD_SLEEP();
  asm volatile("" ::: "memory");
}
PD_FN void gran_lds_wait(unsigned* p) {
  while (PD_UNIFORM(PD_LDS_FLAG(p)) == 0) PD_SLEEP();
  asm volatile("" ::: "memory");
}
// ring form: the flag carries the number of the granule it is meant for
PD_FN void gran_lds_flag_tag(int lane, unsigned* p, unsigned tag) { if (lane == 0) PD_LDS_FLAG(p) = tag; }
PD_FN void gran_lds_wait_tag(unsigned* p, unsigned tag) {
  while ((unsigned)PD_UNIFORM(PD_LDS_FLAG(p)) != tag) PD_SLEEP();
  asm volatile("" ::: "memory");
}
PD_FN void gran_far_signal(int lane, const DecodeArgs& a, int g, int k) {
  PD_VMEM_DRAIN();
  if (lane == 0) PD_STORE_DEVICE(a.chain_flag + 2 * (size_t)g + k, a.chain_epoch);
}
// the tails of granule g (place gp) to whoever decodes granule g + 1
PD_FN void gran_send_tails(int lane, const float* y2, const DecodeArgs& a, int g, const GranPos& gp) {
  if (gp.w == gp.wpw - 1) {
    float* st = a.chain_state + (size_t)g * kGranFloats;
    for (int m = 0; m < kOvlRegs; m++) PD_STORE_DEVICE(&st[m * 64 + lane], y2[m]);
    gran_far_signal(lane, a, g, 0);
  } else {
    GranMb& mb = gp.mb[gp.w + 1];
    gran_lds_wait(&mb.free);
    float* box = gran_tails_box(gp.wl[gp.w + 1]);
    for (int m = 0; m < kOvlRegs; m++) box[m * 64 + lane] = y2[m];
    PD_WAVE_SYNC();
    gran_lds_flag(lane, &mb.tails_full);
  }
}
// slots 3..17 of granule g's matrixing output, from the wave that has them in LDS
PD_FN void gran_send_rows(int lane, const WaveData& L, const DecodeArgs& a, int g, const GranPos& gp) {
  const int ch = lane >> 5, i = lane & 31;
  if (gp.w == gp.wpw - 1) {
    float* st = a.chain_state + (size_t)g * kGranFloats + kOvlRegs * 64;
    // (all fifteen out of LDS first: a device-scope store is ordered after the load of its value, and load by load that is
    //  fifteen LDS round trips in a row in the workgroup's last wave -- the wave a launch ends with)
    float v[kHistSlots];
    PD_UNROLL for (int s = 0; s < kHistSlots; s++) v[s] = L.hyb[ch][3 + s][i];
    PD_UNROLL for (int s = 0; s < kHistSlots; s++) PD_STORE_DEVICE(&st[s * 64 + lane], v[s]);
    gran_far_signal(lane, a, g, 1);
  } else {
    GranMb& mb = gp.mb[gp.w + 1];
    gran_lds_wait(&mb.free);
    float* box = gran_rows_box(gp.wl[gp.w + 1]);
    for (int s = 0; s < kHistSlots; s++) box[s * 64 + lane] = L.hyb[ch][3 + s][i];
    PD_WAVE_SYNC();
    gran_lds_flag(lane, &mb.rows_full);
  }
}
// ring form of the two (persistent kernel): always through the next place's mailboxes, flags = the receiving granule's tag
PD_FN void gran_send_tails_ring(int lane, const float* y2, int g, const GranPos& gp) {
  const int nx = gran_next_place(gp);
  const unsigned tag = gran_tag(gp, g + 1);
  GranMb& mb = gp.mb[nx];
  gran_lds_wait_tag(&mb.free, tag);
  float* box = gran_tails_box(gp.wl[nx]);
  for (int m = 0; m < kOvlRegs; m++) box[m * 64 + lane] = y2[m];
  PD_WAVE_SYNC();
  gran_lds_flag_tag(lane, &mb.tails_full, tag);
}
PD_FN void gran_send_rows_ring(int lane, const WaveData& L, int g, const GranPos& gp) {
  const int ch = lane >> 5, i = lane & 31;
  const int nx = gran_next_place(gp);
  const unsigned tag = gran_tag(gp, g + 1);
  GranMb& mb = gp.mb[nx];
  gran_lds_wait_tag(&mb.free, tag);
  float* box = gran_rows_box(gp.wl[nx]);
  for (int s = 0; s < kHistSlots; s++) box[s * 64 + lane] = L.hyb[ch][3 + s][i];
  PD_WAVE_SYNC();
  gran_lds_flag_tag(lane, &mb.rows_full, tag);
}
// both parts from a wave that has the state in registers (run_chunk at the end of a frame), for granule g = the frame's
// second (gp = ITS place): coefficient 16 + i is he of lane i < 16, coefficient 16 - i is ho of lane i <= 16 -- together
// all 32 of a row
PD_FN void gran_publish_regs(int lane, const LaneRegs& R, const DecodeArgs& a, int g, const GranPos& gp) {
  // (does the frame after it take the state from the chain at all?  -- same facts as run_granule_wave)
  const int fn = (g >> 1) + 1;
  if (fn >= a.n_frames) return;
  const uint8_t nb = reinterpret_cast<const uint8_t*>(a.side + (size_t)fn * 4)[7];
  if (((nb & PDMP3_FR_MODE_MASK) >> PDMP3_FR_MODE_SHIFT) == 3 || (nb & PDMP3_FR_RESET)) return;
  if (frame_is_rare(a.side + (size_t)fn * 4)) return;             // (it goes the way mono frames go)
  const int ch = lane >> 5, i = lane & 31;
  if (gp.ring) {
    if (g + 1 >= gp.g_end) return;              // (the next range derives its opening state itself)
    const int nx = gran_next_place(gp);
    const unsigned tag = gran_tag(gp, g + 1);
    GranMb& mb = gp.mb[nx];
    gran_lds_wait_tag(&mb.free, tag);
    float* tb = gran_tails_box(gp.wl[nx]);
    float* rb = gran_rows_box(gp.wl[nx]);
    for (int m = 0; m < kOvlRegs; m++) tb[m * 64 + lane] = R.ovl[m];
    for (int s = 0; s < kHistSlots; s++) {
      if (i < 16) rb[s * 64 + ch * 32 + 16 + i] = R.he[s];
      if (i <= 16) rb[s * 64 + ch * 32 + 16 - i] = R.ho[s];
    }
    PD_WAVE_SYNC();
    if (lane == 0) { PD_LDS_FLAG(&mb.tails_full) = tag; PD_LDS_FLAG(&mb.rows_full) = tag; }
    return;
  }
  const bool far = gp.w == gp.wpw - 1;
  float* tails = far ? a.chain_state + (size_t)g * kGranFloats : nullptr;
  if (far) {
    for (int m = 0; m < kOvlRegs; m++) PD_STORE_DEVICE(&tails[m * 64 + lane], R.ovl[m]);
    float* rows = tails + kOvlRegs * 64;
    for (int s = 0; s < kHistSlots; s++) {
      if (i < 16) PD_STORE_DEVICE(&rows[s * 64 + ch * 32 + 16 + i], R.he[s]);
      if (i <= 16) PD_STORE_DEVICE(&rows[s * 64 + ch * 32 + 16 - i], R.ho[s]);
    }
    PD_VMEM_DRAIN();
    if (lane == 0) { PD_STORE_DEVICE(a.chain_flag + 2 * (size_t)g, a.chain_epoch); PD_STORE_DEVICE(a.chain_flag + 2 * (size_t)g + 1, a.chain_epoch); }
  } else {
    GranMb& mb = gp.mb[gp.w + 1];
    gran_lds_wait(&mb.free);
    float* tb = gran_tails_box(gp.wl[gp.w + 1]);
    float* rb = gran_rows_box(gp.wl[gp.w + 1]);
    for (int m = 0; m < kOvlRegs; m++) tb[m * 64 + lane] = R.ovl[m];
    for (int s = 0; s < kHistSlots; s++) {
      if (i < 16) rb[s * 64 + ch * 32 + 16 + i] = R.he[s];
      if (i <= 16) rb[s * 64 + ch * 32 + 16 - i] = R.ho[s];
    }
    PD_WAVE_SYNC();
    if (lane == 0) { PD_LDS_FLAG(&mb.tails_full) = 1u; PD_LDS_FLAG(&mb.rows_full) = 1u; }
  }
}
// Wave 0 of a workgroup: part k of granule g - 1 from the workgroup before.  The wait is BOUNDED (false: given up)
constexpr int kGranFarPolls = 1 << 13;      // x (a device-scope load + s_sleep): some milliseconds
PD_FN bool gran_far_wait(const DecodeArgs& a, int g, int k, bool bounded) {
  if (bounded && (a.debug_flags & PD_DEBUG_FAR_TIMEOUT)) return false;
  for (int n = 0; (unsigned)PD_UNIFORM(PD_LOAD_DEVICE(a.chain_flag + 2 * (size_t)(g - 1) + k)) != a.chain_epoch; ++n) {
    if (bounded && n >= kGranFarPolls) return false;
    PD_SLEEP();
  }
  asm volatile("" ::: "memory");
  return true;
}

// ---- ph_mfma in two stages (stereo granules).  ph_imdct: everything up to the windowed IMDCT outputs, indexed like
// R.ovl: y1 = first halves (what the overlap is added to), y2 = tails (the next granule's overlap).  xr is dead after it.
PD_FN void ph_imdct(int lane, const WaveData& L, const TabLds& S, const LaneRegs& R, BankPtr cb, const GlobalTables& T, float* y1, float* y2,
                    const GranuleInfo* gi = nullptr) {
  const GranuleInfo g = gi ? *gi : granule_info(L);
  const int j = lane & 15, kq = lane >> 4;
  const bool any_short = g.is_short(0) || g.is_short(1);
  {   // IMDCT outputs p = 16, 17, 34, 35 on the VALU: lane = (cl, sb), scalar-broadcast coefficients
    const int cl = lane >> 5, sb = lane & 31;
    const bool lwsf = (g.flags(cl) & PDMP3_GC_WIN_SWITCH) != 0;
    const bool llow = lwsf && g.is_mixed(cl) && sb < 2;
    const bool lshort = g.is_short(cl) && !llow;
    const float* x = &L.xr[cl][18 * sb];
    float in[18];
    PD_UNROLL for (int m = 0; m < 18; m++) in[m] = x[m];
    float y[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    PD_UNROLL for (int m = 0; m < 18; m++)
      PD_UNROLL for (int q = 0; q < 4; q++) y[q] = PD_FMA(in[m], PD_C36(cb, c36x, q, m), y[q]);
    const float* w = S.win[llow ? 0 : g.block_type(cl)];
    y[0] = y[0] * w[16]; y[1] = y[1] * w[17]; y[2] = y[2] * w[34]; y[3] = y[3] * w[35];
    if (any_short) {                      // wave-uniform
      float ys[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      PD_UNROLL for (int m = 0; m < 18; m++)
        PD_UNROLL for (int q = 0; q < 4; q++) ys[q] = PD_FMA(in[m], PD_C36(cb, s36x, q, m), ys[q]);
      PD_UNROLL for (int q = 0; q < 4; q++) y[q] = lshort ? ys[q] : y[q];
    }
    y1[16] = y[0]; y1[17] = y[1]; y2[16] = y[2]; y2[17] = y[3];
  }
  PD_UNROLL for (int cc = 0; cc < 2; cc++) {
    const int ch = 1 - cc;
    const bool shrt = g.is_short(ch);
    const bool wsf = (g.flags(ch) & PDMP3_GC_WIN_SWITCH) != 0;
    const bool mixrows = wsf && g.is_mixed(ch);        // subbands 0, 1 use window/transform 0 (P:1769-1771)
    const int bt = g.block_type(ch);
    float bfr[10];
    if (shrt) {
        // (pinned = waited for INSIDE this branch: the wait counters are per wave, not per path -- with the short blocks' loads
        //  still in flight where the paths meet, the long-block path too sat through `s_waitcnt vmcnt(9) .. vmcnt(0)` between
        //  its matrix instructions, i.e. waited for the prefetch of the next granule, asked for a moment before, every granule)
        PD_UNROLL for (int k = 0; k < 10; k++) bfr[k] = T.frag_short[k * 64 + lane];
        PD_UNROLL for (int k = 0; k < 10; k++) PD_PIN(bfr[k]);
      }
    else { PD_UNROLL for (int k = 0; k < 10; k++) bfr[k] = R.bi[k]; }
    f32x4 acc[2][2];
    float afr[2][5];
    PD_UNROLL for (int h = 0; h < 2; h++)
      PD_UNROLL for (int kk = 0; kk < 5; kk++) {
        const int k = 4 * kk + kq;
        afr[h][kk] = L.xr[ch][18 * (h ? 31 - j : j) + (k < 18 ? k : 0)];   // second row tile: subbands reversed
        if (k >= 18) afr[h][kk] = 0.0f;
      }
    PD_UNROLL for (int h = 0; h < 2; h++)
      PD_UNROLL for (int nt = 0; nt < 2; nt++) acc[h][nt] = mfma16(afr[h][0], bfr[nt], (f32x4){0, 0, 0, 0});
    PD_UNROLL for (int kk = 1; kk < 5; kk++)
      PD_UNROLL for (int h = 0; h < 2; h++)
        PD_UNROLL for (int nt = 0; nt < 2; nt++) acc[h][nt] = mfma16(afr[h][kk], bfr[kk * 2 + nt], acc[h][nt]);
    f32x4 accl[2];
    PD_UNROLL for (int nt = 0; nt < 2; nt++) accl[nt] = acc[0][nt];
    if (shrt && mixrows) {             // wave-uniform: rows sb 0, 1 of the first tile take the long transform
      PD_UNROLL for (int nt = 0; nt < 2; nt++) accl[nt] = mfma16(afr[0][0], R.bi[nt], (f32x4){0, 0, 0, 0});
      PD_UNROLL for (int kk = 1; kk < 5; kk++)
        PD_UNROLL for (int nt = 0; nt < 2; nt++) accl[nt] = mfma16(afr[0][kk], R.bi[kk * 2 + nt], accl[nt]);
    }
    const float* wb = S.win[bt];
    const float* w0 = S.win[0];
    const float wb1 = wb[j], wb2 = wb[18 + j];
    const float w01 = w0[j], w02 = w0[18 + j];
    PD_UNROLL for (int h = 0; h < 2; h++)
      PD_UNROLL for (int r = 0; r < 4; r++) {
        const bool lowrow = mixrows && h == 0 && kq == 0 && r < 2;      // subbands 0, 1 (first tile, rows 0, 1)
        float a1 = acc[h][0][r], a2 = acc[h][1][r];
        if (h == 0 && lowrow) { a1 = accl[0][r]; a2 = accl[1][r]; }
        const bool win_folded = shrt && !lowrow;                        // short transform: window is in the matrix
        const float f1 = lowrow ? w01 : wb1, f2 = lowrow ? w02 : wb2;
        if (!win_folded) { a1 = a1 * f1; a2 = a2 * f2; }
        y1[ch * 8 + h * 4 + r] = a1;
        y2[ch * 8 + h * 4 + r] = a2;
      }
  }
}
// ph_overlap_matrix: the rest -- overlap-add against the tails `ovl` of the granule before (P:1775), frequency
// inversion (P:1738-1746), the H5 peek values, matrixing -> hyb
PD_FN void ph_overlap_matrix(int lane, WaveData& L, const LaneRegs& R, const float* y1, const float* ovl) {
  const int j = lane & 15, kq = lane >> 4;
  {
    const int cl = lane >> 5, sb = lane & 31;
    const float o16 = y1[16] + ovl[16];
    float o17 = y1[17] + ovl[17];
    if (sb & 1) o17 = -o17;
    const float p16 = PD_SHFL_XOR(o16, 31), p17 = PD_SHFL_XOR(o17, 31);
    if (sb < 16) {
      L.lo[0][2 * cl + 0][sb] = o16 + p16; L.lo[1][2 * cl + 0][sb] = o16 - p16;
      L.lo[0][2 * cl + 1][sb] = o17 + p17; L.lo[1][2 * cl + 1][sb] = o17 - p17;
    }
  }
  PD_UNROLL for (int cc = 0; cc < 2; cc++) {
    const int ch = 1 - cc;
    float outa[8];
    PD_UNROLL for (int h = 0; h < 2; h++)
      PD_UNROLL for (int r = 0; r < 4; r++) {
        const int oi = ch * 8 + h * 4 + r;
        float o = y1[oi] + ovl[oi];
        const int sb = h ? 31 - (4 * kq + r) : 4 * kq + r;
        if ((sb & 1) && (j & 1)) o = -o;
        outa[h * 4 + r] = o;
        if (ch == 0 && h == 0 && r == 0 && kq == 0 && j < 3) L.peek[j] = o;
      }
    f32x4 me = (f32x4){0, 0, 0, 0}, mo = (f32x4){0, 0, 0, 0};
    PD_UNROLL for (int r = 0; r < 4; r++) {
      const float a = outa[r] + outa[4 + r], b = outa[r] - outa[4 + r];
      me = mfma16(a, R.bm[r], me);
      mo = mfma16(b, R.bm[4 + r], mo);
    }
    PD_UNROLL for (int r = 0; r < 4; r++) {
      L.hyb[ch][4 * kq + r][2 * j] = me[r];
      L.hyb[ch][4 * kq + r][2 * j + 1] = mo[r];
    }
  }
  PD_WAVE_SYNC();                          // lo[] was written by other lanes
  {
    f32x4 me = (f32x4){0, 0, 0, 0}, mo = (f32x4){0, 0, 0, 0};
    PD_UNROLL for (int r = 0; r < 4; r++) {
      const float a = (j < 4) ? L.lo[0][j & 3][4 * kq + r] : 0.0f;
      const float b = (j < 4) ? L.lo[1][j & 3][4 * kq + r] : 0.0f;
      me = mfma16(a, R.bm[r], me);
      mo = mfma16(b, R.bm[4 + r], mo);
    }
    if (kq == 0) {
      PD_UNROLL for (int r = 0; r < 4; r++) {
        const int ch = r >> 1, t = 16 + (r & 1);
        L.hyb[ch][t][2 * j] = me[r]; L.hyb[ch][t][2 * j + 1] = mo[r];
      }
    }
  }
}
// the window sums of a stereo granule in two parts (same chain of 16 FMAs per sum as ph_window, newest slot first):
// the terms that read the granule's own slots ...
PD_FN void ph_window_own(int lane, const WaveData& L, const TabLds& S, LaneRegs& R, float* acc) {
  const int ch = lane >> 5;
  PD_UNROLL for (int k = 0; k < 8; k++) { R.we[k] = S.taps[k][lane & 31]; R.wo[k] = S.taps[8 + k][lane & 31]; }
  float E[18], O[18];
  PD_UNROLL for (int t = 0; t < 18; t++) { E[t] = L.hyb[ch][t][R.idx_e]; O[t] = L.hyb[ch][t][R.idx_o]; }
  PD_UNROLL for (int t = 0; t < 18; t++) {
    float s = 0.0f;
    PD_UNROLL for (int k = 0; k < 8; k++) {
      if (t - 2 * k >= 0) s = PD_FMA(R.we[k], E[t - 2 * k], s);
      if (t - 2 * k - 1 >= 0) s = PD_FMA(R.wo[k], O[t - 2 * k - 1], s);
    }
    acc[t] = s;
  }
}
// ... and the ones that read the history R.he / R.ho (slots 3..17 of the granule before); PCM
template <bool F32>
PD_FN void ph_window_hist(int lane, WaveData& L, const LaneRegs& R, const float* acc, int16_t* pcm_g, float* pcmf_g) {
  float sum[18];
  PD_UNROLL for (int t = 0; t < 18; t++) {
    float s = acc[t];
    PD_UNROLL for (int k = 0; k < 8; k++) {
      if (t - 2 * k < 0) s = PD_FMA(R.we[k], R.he[kHistSlots + t - 2 * k], s);
      if (t - 2 * k - 1 < 0) s = PD_FMA(R.wo[k], R.ho[kHistSlots + t - 2 * k - 1], s);
    }
    sum[t] = s;
  }
  pcm_emit<F32, true>(lane, L, 2, true, sum, pcm_g, pcmf_g);
}

// run_chunk as a CALLED function, for the granule kernel's rare paths (frames that do not go the granule way; the state
// at the start of a frame when the bounded wait for another workgroup has run out).  Not inlined on purpose: inlined,
// its register demand and its spills shape the allocation of the whole kernel -- the hot path went from no spills to
// hundreds.  Arguments by value / as plain pointers (the LDS blocks become generic pointers: slower accesses, here only).
template <bool F32>
PD_SLOW_FN void gran_slow_chunk(DecodeArgs a, GlobalTables T, BankPtr cb, int f, WaveData* L, TabLds* S, GranPos second, bool publish,
                                LaneRegs* state_only) {
  // (arguments of a called function arrive in vector registers; the constant bank's address has to be scalar again)
  cb = PD_UNIFORM_PTR(BankPtr, cb);
  f = PD_UNIFORM(f);
  gran_tabs_wait(second);
  run_chunk<false, false, F32, false>(a, T, cb, f, *L, *S, publish ? &second : nullptr, state_only);
}

// One granule of a stereo frame whose predecessor's state comes through the chain (or is the caller's / zero: `fresh`).
// pf: the granule's spectra / side records, in flight since the kernel's entry; next_takes: the wave of granule g + 1
// will take this one's state from the chain.
// RING (persistent kernel, run_granule_ring): the wave goes round a range of granules -- its constants Rp are loaded once,
// the granule's input is in Rp->pf*, the hand-over is the ring form (no memory path); emit = false: a HALO granule in front
// of the range (its PCM belongs to the workgroup before: nothing is windowed or stored, tails and rows are handed on);
// zero_in: the first halo granule (whatever state it starts from does not reach the range: SURVEY 8e);
// halo_in: the range's first granule when no halo frame can be run in front of it (mono frames there): the state at
// its start is derived by this wave the independent way (run_chunk's halo as a called function);
// g_pf: the granule whose input is asked for while this one is windowed (this wave's next), -1: none.
template <bool F32, bool RING = false>
PD_FN void run_granule(const DecodeArgs& a, const GlobalTables& T, BankPtr cb, int g, WaveData& L, TabLds& S, const GranPos& gp,
                       bool fresh, bool h5, bool next_takes, const LaneRegs& pf, LaneRegs* Rp = nullptr, bool emit = true,
                       bool zero_in = false, bool halo_in = false, int g_pf = -1) {
  LaneRegs R;
  (void)Rp;
  int lane_ = PD_LANE();
  // (persistent kernel: the lane number is laundered every turn, so that nothing derived from it -- addresses, masks,
  //  table indices -- is hoisted out of the loop and held in registers across the whole body)
  if (RING) PD_PIN(lane_);
  const int lane = lane_;
  const int f = g >> 1, gr = g & 1;
  const bool from_caller = fresh && gr == 0 && f == 0 && a.state_in && !(reinterpret_cast<const uint8_t*>(a.side)[7] & PDMP3_FR_RESET);
  const bool from_zero = (fresh && gr == 0 && !from_caller) || (RING && zero_in);
  const bool from_chain = !from_caller && !from_zero;
  const unsigned tag = gran_tag(gp, g);
  // (development: shader-clock stamps per wave when a.prof is set -- tools/gran_profile.py)
#define PD_GT(k) if (a.prof) { const unsigned long long t_ = PD_CLOCK(); if (lane == 0) a.prof[(size_t)g * kProfSlots + (k)] = t_; }
  PD_GT(1)
  R.pf0 = pf.pf0; R.pf1 = pf.pf1; R.pf2 = pf.pf2; R.pf3 = pf.pf3;
  // A launch ends with its slowest wave, and those are the two of every workgroup whose hand-over goes through memory
  // (the last one publishes with stores it has to see acknowledged, the first one reads past its caches): the last wave
  // runs ahead of the other three of its SIMD until it has published, the first one catches up after it has taken
  const bool far_sender = !RING && gp.w == gp.wpw - 1 && next_takes;
  if (far_sender) PD_SETPRIO(2);
  PD_PHASE(lane_init<false>(lane, L, R, cb, T))
  if (h5 && (gr == 1 || !fresh)) {
    if (!RING) PD_SETPRIO(3);     // (these waves have an eighth more to do than the others of their SIMDs, and a launch ends with its last wave)
    // (wave-uniform) granule 1 / channel 1 of this frame is a short block: its scales read three hybrid outputs of granule 0
    // (SURVEY H5): the first-half IMDCT outputs p = 0..2 of granule 0's (channel 0, subband 0) plus the tails of the
    // granule before the frame.  Waiting for the waves that compute them anyway would put granule 1's wave most of a
    // granule behind all others -- and a launch ends with its last wave -- so the numbers are derived from data only:
    // lines 0..63 requantised, one alias boundary, three 18-term sums per granule (the sums ph_imdct forms for them).
    // The TWO waves of the frame share the work -- granule 0's wave takes the granule before the frame (its tails:
    // ph_peek_tail, as in run_chunk) and leaves the three values in the mailbox of granule 1's wave, which takes granule
    // 0 meanwhile: each is late by one pass instead of one of them by two (7.8 k ticks in front of its own granule:
    // the 115 such waves of a C2 launch made it 2 us longer; so shared, 0.7).
    float pk = 0.0f;
    gran_tabs_wait(gp);
    if (gr == 0) {
      LaneRegs R1;
      R1.ovl[0] = 0.0f;
      PD_PHASE(ph_prefetch(lane, R1, a.spectra + (size_t)(g - 1) * 1152, a.side + (size_t)(g - 1) * 2))
      PD_PHASE(ph_commit(lane, L, R1))
      PD_PHASE(ph_scales(lane, L))
      PD_PHASE((ph_requant<false, 1, false, true, false, false>(lane, L, S, cb, T, nullptr, nullptr)))
      PD_PHASE(ph_antialias(lane, L, cb, true))
      PD_PHASE(ph_peek_tail(lane, L, S, R1, T))
      GranMb& mb = gp.mb[gran_next_place(gp)];          // (a frame's two granules are neighbours in one workgroup: WPW is even)
      if (lane < 3) mb.peek_tail[lane] = R1.ovl[0];
      PD_WAVE_SYNC();
      if (RING) gran_lds_flag_tag(lane, &mb.peek_full, gran_tag(gp, g + 1));
      else gran_lds_flag(lane, &mb.peek_full);
    } else {
      LaneRegs R0;
      float tail = 0.0f;
      PD_PHASE(ph_prefetch(lane, R0, a.spectra + (size_t)(g - 1) * 1152, a.side + (size_t)(g - 1) * 2))
      if (fresh && f == 0 && a.state_in && !(reinterpret_cast<const uint8_t*>(a.side)[7] & PDMP3_FR_RESET) && lane < 3) tail = a.state_in[lane];
      PD_PHASE(ph_commit(lane, L, R0))
      PD_PHASE(ph_scales(lane, L))
      PD_PHASE((ph_requant<false, 1, false, true, false, false>(lane, L, S, cb, T, nullptr, nullptr)))
      PD_PHASE(ph_antialias(lane, L, cb, true))
      const float head = ph_peek_head(lane, L, S, T);
      PD_WAVE_SYNC();
      if (!fresh) {
        if (RING) gran_lds_wait_tag(&gp.mb[gp.w].peek_full, tag);
        else gran_lds_wait(&gp.mb[gp.w].peek_full);
        tail = lane < 3 ? gp.mb[gp.w].peek_tail[lane] : 0.0f;
      }
      pk = head + tail;
    }
    PD_PHASE(ph_commit(lane, L, R))
    if (gr == 1) { PD_PHASE(if (lane < 3) L.peek[lane] = pk) }
  } else {
    PD_PHASE(ph_commit(lane, L, R))
  }
  PD_GT(2)
  gran_tabs_wait(gp);
  PD_LAUNDER(cb);
  const GranuleInfo gi = granule_info<false>(L);  // (the granule's facts, read once for the three phases that want them; frames that are
                                                  //  frame_is_rare() do not come this way: run_granule_wave)
  if (!(PD_EXP_SKIP & 1)) { PD_PHASE((ph_requant<false, 9, true, true, true, false>(lane, L, S, cb, T, nullptr, nullptr, &gi))) }
  PD_GT(3)
  // A SIMD issues from its OLDEST wave first: of the four waves that share one -- places w, w + 4, w + 8, w + 12 of the
  // workgroup -- the first gets through requantisation in 5 k ticks and the last in 12 k (profiles/r04_gran_profile.txt),
  // and a workgroup holds its CU until its last wave is through.  From here on the younger a wave, the higher its
  // priority: the late ones catch up (same box, three runs each: C2 19.9-20.3 -> 19.4-19.8 us, 8192 frames 70.0 -> 68.7,
  // 12288 frames 102.4 -> 100.8; every other schedule tried -- from the start, mirrored, back to equal after the IMDCT --
  // was worse or the same).
  // (round 5: priorities by STAGE instead -- the further a wave has got, the lower, which is what paid in the chunk kernel --
  //  or stage + age class: C2 18.4-18.6 us as it is, 18.5-19.7 with every such schedule, profiles/r05_kernel_experiments.txt)
  if (!RING && !(h5 && (gr == 1 || !fresh))) {
    const int q = gp.w >> 2;
    if (q == 1) PD_SETPRIO(1); else if (q == 2) PD_SETPRIO(2); else if (q == 3) PD_SETPRIO(3); else PD_SETPRIO(0);
  }
  if (!(PD_EXP_SKIP & 2)) { PD_PHASE(ph_antialias(lane, L, cb, false, &gi)) }
  float y1[kOvlRegs], y2[kOvlRegs];
  if (PD_EXP_SKIP & 4) { PD_UNROLL for (int m = 0; m < kOvlRegs; m++) { y1[m] = L.xr[0][m * 64 + lane]; y2[m] = L.xr[1][m * 64 + lane]; } }
  else { PD_PHASE(ph_imdct(lane, L, S, R, cb, T, y1, y2, &gi)) }
  PD_GT(4)
  // from here on the wave reads nothing of spec / side / scale / xr any more: its mailboxes are free
  if (RING) gran_lds_flag_tag(lane, &gp.mb[gp.w].free, tag);
  else if (gp.w > 0) gran_lds_flag(lane, &gp.mb[gp.w].free);
  if (next_takes) {
    if (RING) { PD_PHASE(gran_send_tails_ring(lane, y2, g, gp)) }
    else { PD_PHASE(gran_send_tails(lane, y2, a, g, gp)) }
  }
  PD_GT(5)
  float ovl[kOvlRegs];
  bool have_halo = false;
  LaneRegs H;                  // (memory, and touched only if the wait below is given up)
  if (from_caller) { PD_UNROLL for (int m = 0; m < kOvlRegs; m++) ovl[m] = a.state_in[m * 64 + lane]; }
  else if (from_zero) { PD_UNROLL for (int m = 0; m < kOvlRegs; m++) ovl[m] = 0.0f; }
  else if (RING ? !halo_in : gp.w > 0) {
    if (RING) gran_lds_wait_tag(&gp.mb[gp.w].tails_full, tag);
    else gran_lds_wait(&gp.mb[gp.w].tails_full);
    const float* box = gran_tails_box(L);
    PD_UNROLL for (int m = 0; m < kOvlRegs; m++) ovl[m] = box[m * 64 + lane];
    PD_WAVE_SYNC();
  } else if (!RING && gran_far_wait(a, g, 0, true)) {
    const float* st = a.chain_state + (size_t)(g - 1) * kGranFloats;
    PD_UNROLL for (int m = 0; m < kOvlRegs; m++) ovl[m] = PD_LOAD_DEVICE(&st[m * 64 + lane]);
  } else {
    // The workgroup before this one has not delivered within the bound (it is not resident: a partitioned or shared
    // device, a dispatcher that does not go in order).  Do not depend on it: derive the state at the start of the
    // frame the independent way, from the granules before it (run_chunk's halo) -- waiting costs time, never progress.
    // (Persistent kernel: the first granule of a range whose halo frame cannot be run in front of it.)
    PD_WAVE_SYNC();
    gran_slow_chunk<F32>(a, T, cb, f, &L, &S, gp, false, &H);
    PD_WAVE_SYNC();
    PD_UNROLL for (int m = 0; m < kOvlRegs; m++) ovl[m] = H.ovl[m];
    have_halo = true;
  }
  PD_GT(6)
  if (PD_EXP_SKIP & 8) { L.hyb[0][lane & 15][lane >> 2] = y1[0] + ovl[1]; } else { PD_PHASE(ph_overlap_matrix(lane, L, R, y1, ovl)) }
  PD_GT(7)
  if (g == 2 * a.n_frames - 1 && a.state_out) {       // (wave-uniform) the launch's closing state, in the caller's form
    float* so = a.state_out;
    const int ch = lane >> 5;
    PD_UNROLL for (int m = 0; m < kOvlRegs; m++) so[m * 64 + lane] = y2[m];
    PD_UNROLL for (int s = 0; s < kHistSlots; s++) {
      so[(kOvlRegs + s) * 64 + lane] = L.hyb[ch][3 + s][R.idx_e];
      so[(kOvlRegs + kHistSlots + s) * 64 + lane] = L.hyb[ch][3 + s][R.idx_o];
    }
  }
  if (next_takes) {
    if (RING) { PD_PHASE(gran_send_rows_ring(lane, L, g, gp)) }
    else { PD_PHASE(gran_send_rows(lane, L, a, g, gp)) }
  }
  if (far_sender && !(gr == 1 && h5)) PD_SETPRIO(0);
  PD_GT(8)
  (void)g_pf;
  if (RING && !emit) {
    // a halo granule: no PCM.  The rows the wave before it sends are not wanted, but they arrive in this wave's spec | side
    // | scale: the next granule must not be committed there before they have
    if (from_chain) gran_lds_wait_tag(&gp.mb[gp.w].rows_full, tag);
    PD_GT(11)
    return;
  }
  float acc[18];
  if (PD_EXP_SKIP & 16) { PD_UNROLL for (int t = 0; t < 18; t++) acc[t] = L.hyb[0][t][lane & 31]; } else { PD_PHASE(ph_window_own(lane, L, S, R, acc)) }
  PD_GT(9)
  if (have_halo) { PD_UNROLL for (int s = 0; s < kHistSlots; s++) { R.he[s] = H.he[s]; R.ho[s] = H.ho[s]; } }
  else if (from_caller) {
    PD_UNROLL for (int s = 0; s < kHistSlots; s++) {
      R.he[s] = a.state_in[(kOvlRegs + s) * 64 + lane];
      R.ho[s] = a.state_in[(kOvlRegs + kHistSlots + s) * 64 + lane];
    }
  } else if (from_zero) { PD_UNROLL for (int s = 0; s < kHistSlots; s++) { R.he[s] = 0.0f; R.ho[s] = 0.0f; } }
  else if (RING || gp.w > 0) {
    if (RING) gran_lds_wait_tag(&gp.mb[gp.w].rows_full, tag);
    else gran_lds_wait(&gp.mb[gp.w].rows_full);
    const float* rows = gran_rows_box(L) + (lane >> 5) * 32;
    PD_UNROLL for (int s = 0; s < kHistSlots; s++) { R.he[s] = rows[s * 64 + R.idx_e]; R.ho[s] = rows[s * 64 + R.idx_o]; }
  } else {
    // (the tails came, so the workgroup before this one is running: its rows will come -- no bound needed)
    gran_far_wait(a, g, 1, false);
    const float* rows = a.chain_state + (size_t)(g - 1) * kGranFloats + kOvlRegs * 64 + (lane >> 5) * 32;
    PD_UNROLL for (int s = 0; s < kHistSlots; s++) { R.he[s] = PD_LOAD_DEVICE(&rows[s * 64 + R.idx_e]); R.ho[s] = PD_LOAD_DEVICE(&rows[s * 64 + R.idx_o]); }
  }
  (void)from_chain;
  PD_GT(10)
  if (PD_EXP_SKIP & 32) { a.pcm[(size_t)f * 2304 + gr * 1152 + lane] = (int16_t)(acc[0] + acc[17] + R.he[0] + R.ho[14]); } else { PD_PHASE(ph_window_hist<F32>(lane, L, R, acc, a.pcm + (size_t)f * 2304 + gr * 1152, F32 ? a.pcm_f32 + (size_t)f * 2304 + gr * 1152 : nullptr)) }
  PD_GT(11)
#undef PD_GT
}

// The wave of granule g: which way its frame goes (wave-uniform facts from the side records; both waves of a frame
// decide alike).  gp = this granule's place; pf = its input, prefetched.
// RING: see run_granule; g_emit = the range's first granule proper (granules before it are its halo frame).
template <bool F32, bool RING = false>
PD_FN void run_granule_wave(const DecodeArgs& a, const GlobalTables& T, BankPtr cb, int g, WaveData& L, TabLds& S, const GranPos& gp,
                            const LaneRegs& pf, LaneRegs* Rp = nullptr, int g_emit = 0, bool halo_by_wave = false, int g_pf = -1) {
  const int f = g >> 1, gr = g & 1;
  const uint8_t fb = reinterpret_cast<const uint8_t*>(a.side + (size_t)f * 4)[7];
  const uint8_t pb = reinterpret_cast<const uint8_t*>(a.side + (size_t)(f > 0 ? f - 1 : 0) * 4)[7];
  const uint8_t nb = reinterpret_cast<const uint8_t*>(a.side + (size_t)(f + 1 < a.n_frames ? f + 1 : f) * 4)[7];
  const uint8_t fl = reinterpret_cast<const uint8_t*>(a.side + (size_t)(2 * f + 1) * 2 + 1)[3];
  // (frames with intensity stereo -- and LSF ones, should a caller hand them to this kernel -- go the way mono frames
  //  go: decoded as a whole by the wave of their first granule, with run_chunk's RARE copy; to the frames around them
  //  they look like mono frames do: nothing is taken from them through the chain)
  const bool rare = frame_is_rare(a.side + (size_t)f * 4);
  const bool prev_rare = frame_is_rare(a.side + (size_t)(f > 0 ? f - 1 : 0) * 4);
  const bool next_rare = frame_is_rare(a.side + (size_t)(f + 1 < a.n_frames ? f + 1 : f) * 4);
  const bool stereo = ((fb & PDMP3_FR_MODE_MASK) >> PDMP3_FR_MODE_SHIFT) != 3 && !rare;
  const bool fresh = f == 0 || (fb & PDMP3_FR_RESET);       // its input state is the caller's / zero
  const bool prev_stereo = ((pb & PDMP3_FR_MODE_MASK) >> PDMP3_FR_MODE_SHIFT) != 3 && !prev_rare;
  const bool chained = stereo && (fresh || prev_stereo);
  const bool h5 = (fl & PDMP3_GC_WIN_SWITCH) && ((fl & PDMP3_GC_BLOCK_TYPE_MASK) >> PDMP3_GC_BLOCK_TYPE_SHIFT) == 2;
  bool next_takes = false;
  if (chained) {
    // who takes this granule's state from the chain: granule 1 of the same frame; or the next frame, if it is a stereo
    // frame that does not start from zero (this frame being stereo, it then goes this way too)
    next_takes = gr == 0 || (f + 1 < a.n_frames && ((nb & PDMP3_FR_MODE_MASK) >> PDMP3_FR_MODE_SHIFT) != 3 && !(nb & PDMP3_FR_RESET) && !next_rare);
    if (RING) {
      if (g + 1 >= gp.g_end) next_takes = false;          // (the next range derives its opening state itself)
      const bool halo_frame = g < g_emit;                 // (then this frame is chained or fresh: run_granule_ring chose it so)
      run_granule<F32, true>(a, T, cb, g, L, S, gp, fresh, h5, next_takes, pf, Rp, !halo_frame, halo_frame && gr == 0 && !fresh,
                             halo_by_wave && g == g_emit && !fresh, g_pf);
    } else run_granule<F32, false>(a, T, cb, g, L, S, gp, fresh, h5, next_takes, pf);
    return;
  }
  if (gr == 0) {                                             // the frame is decoded by the wave of its first granule
    GranPos second = gp;                                     // (a frame's two granules are in one workgroup: WPW is even)
    second.w = gp.w + 1;
    if (RING) {
      // this wave's own input was asked for before the loop / during the last window: its registers are dead here
      gran_slow_chunk<F32>(a, T, cb, f, &L, &S, second, true, nullptr);
    } else gran_slow_chunk<F32>(a, T, cb, f, &L, &S, second, true, nullptr);
  }
}

// ---------------------------------------------------------------------------
// Persistent granule kernel (k_decode_p): the launches too large for one granule per wave.
//
// A workgroup of WPW = 16 waves takes a contiguous RANGE of frames [f0, f1) and goes round it: wave w decodes granules
// 2 f0 + w, + 16, + 32, ... with run_granule's straight-line body (128 VGPRs, four waves per SIMD), its per-lane constants
// and the workgroup's LDS tables loaded ONCE, tails and rows handed from place to place through the LDS mailboxes as a
// ring (place 15 hands on to place 0; the flags carry the receiving granule's number).  The only loop-carried registers are
// constants and the next granule's input in flight.  Nothing crosses a range boundary: a range's opening state is
// derived from data -- ONE halo per range instead of one per wave (run_chunk) or a hand-over through memory
// (k_decode_g): the frame before the range is run through the same pipeline with its PCM suppressed (2 of the range's
// ~1000 granule slots; its own input state does not reach the range), or, where that frame does not go the
// granule way (mono, stereo right after mono), by the first wave calling run_chunk's halo.
// Same arithmetic in the same order as run_chunk / run_granule: PCM and carried state bit-identical (tests compare).
// ---------------------------------------------------------------------------
template <bool F32>
PD_FN void run_granule_ring(const DecodeArgs& a, const GlobalTables& T, BankPtr cb, WaveData& L, TabLds& S, GranPos gp, int f0, int f1) {
  const int lane = PD_LANE();
  // the halo (wave-uniform, the same for all waves of the workgroup)
  int g_emit = 2 * f0;
  bool halo_by_wave = false;
  gp.ring = 1;
  gp.g_base = g_emit;
  gp.g_end = 2 * f1;
  if (f0 > 0) {
    const uint8_t b0 = reinterpret_cast<const uint8_t*>(a.side + (size_t)f0 * 4)[7];
    const uint8_t b1 = reinterpret_cast<const uint8_t*>(a.side + (size_t)(f0 - 1) * 4)[7];
    const uint8_t b2 = reinterpret_cast<const uint8_t*>(a.side + (size_t)(f0 > 1 ? f0 - 2 : 0) * 4)[7];
    // ("stereo" as run_granule_wave reads it: frame_is_rare() frames go the way mono frames go)
    const bool st0 = ((b0 & PDMP3_FR_MODE_MASK) >> PDMP3_FR_MODE_SHIFT) != 3 && !frame_is_rare(a.side + (size_t)f0 * 4);
    const bool st1 = ((b1 & PDMP3_FR_MODE_MASK) >> PDMP3_FR_MODE_SHIFT) != 3 && !frame_is_rare(a.side + (size_t)(f0 - 1) * 4);
    const bool st2 = ((b2 & PDMP3_FR_MODE_MASK) >> PDMP3_FR_MODE_SHIFT) != 3 && !frame_is_rare(a.side + (size_t)(f0 > 1 ? f0 - 2 : 0) * 4);
    const bool needs_state = st0 && !(b0 & PDMP3_FR_RESET) && st1;          // (else: zero, or the frame derives its own -- run_granule_wave)
    if (needs_state) {
      const bool prev_goes_granule_way = (b1 & PDMP3_FR_RESET) || f0 == 1 || st2;
      if (prev_goes_granule_way) gp.g_base = g_emit - 2;
      else halo_by_wave = true;
    }
  }
  GlobalTables Tl = T;
  for (int g = gp.g_base + gp.w; g < gp.g_end; g += gp.wpw) {
    // (a turn is run_granule's straight-line body as it is: the per-lane constants are asked for again every turn -- they
    //  come from the L1 / L2 -- because held across the loop they cost what the chunk kernel pays: spills and moves; the
    //  table pointers are laundered so that the compiler does not hoist those loads out of the loop)
    PD_LAUNDER(Tl.taps); PD_LAUNDER(Tl.frag_long); PD_LAUNDER(Tl.frag_mat);
    if (a.prof) { const unsigned long long t_ = PD_CLOCK(); if (lane == 0) a.prof[(size_t)g * kProfSlots] = t_; }
    LaneRegs pf;
    ph_prefetch(lane, pf, a.spectra + (size_t)g * 1152, a.side + (size_t)g * 2);
    run_granule_wave<F32, true>(a, Tl, cb, g, L, S, gp, pf, nullptr, g_emit, halo_by_wave, -1);
    PD_WAVE_SYNC();
  }
}

}  // namespace pdmp3

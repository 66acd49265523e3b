// engine.hip -- gfx950 kernels + the C-ABI of include/pdmp3_hip.h.
//
// Launch geometry: one 64-lane workgroup (= one wavefront) per chunk of
// `chunk_frames` frames; a launch of N frames makes ceil(N/chunk) workgroups,
// so at throughput sizes (>= 10^5 frames) the grid is >> 256 CUs x 8 XCDs and
// consecutive workgroups (which the dispatcher round-robins over the XCDs)
// stream disjoint, contiguous spans of the spectra / PCM buffers.  There is no
// inter-workgroup communication: chunk boundaries are re-derived from a
// 3-granule halo (decode_core.h).
#include <hip/hip_runtime.h>

#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "decode_core.h"
#include "gen_core.h"
#include "host_tables.h"
#include "unpack_core.h"
#include <atomic>
#include <mutex>
#include <new>

using namespace pdmp3;

#ifndef PDMP3_WAVES_PER_EU
#define PDMP3_WAVES_PER_EU 2
#endif

__constant__ ConstBank c_bank;

// Workgroup b is observed to run on XCD b % 8, each XCD with its own L2.  A chunk's halo is the tail of the chunk
// before it, so neighbouring chunks should share an L2: XCD x gets the x-th contiguous eighth of the chunks
// (bijective for any count).  Purely a placement choice -- nothing is communicated between workgroups.
__device__ __forceinline__ int xcd_contiguous(int b, int n) {
  const int q = n >> 3, r = n & 7, x = b & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// Independent chunks: one wavefront (= one workgroup) per chunk of frames, a halo in front of each.
// F32: float PCM (the sums of P:2028 unscaled, DecodeArgs::pcm_f32) instead of int16
// (The wave's LDS block is addressed through its wave number although a workgroup is one wave: with the block at a
//  constant address the same source compiles to a kernel that issues 40 more vector loads per granule and takes 1.45 ms
//  instead of 1.05 ms for 131072 frames -- measured on MI355X, ROCm 7.2; profiles/r03_kernel_experiments.txt.)
template <bool DUMP, bool F32 = false, int WPW = 1>
__global__ __launch_bounds__(64 * WPW, PDMP3_WAVES_PER_EU) void k_decode(DecodeArgs a, GlobalTables T, int n_chunks, unsigned* rare_flag, unsigned rare_epoch) {
  __shared__ WaveLds L[WPW];
  const int w = threadIdx.x >> 6;
  const int n_wgs = (n_chunks + WPW - 1) / WPW;
  const int chunk = xcd_contiguous((int)blockIdx.x, n_wgs) * WPW + w;
  if (chunk >= n_chunks) return;
  // Two KERNELS per launch of chunks: this one is compiled without intensity stereo and LSF -- the code every ordinary
  // chunk runs, as tight as it was before those existed (both copies in one kernel cost it a spilled register and 4 % of
  // a 131072-frame launch: profiles/r06_kernel_experiments.txt) -- and leaves the chunks that hold such a frame
  // (decode_core.h chunk_is_rare: a look at the chunk's frame bytes) to k_decode_rare behind it, which it tells so.
  if (DUMP) { run_chunk<DUMP, false, F32, true, true>(a, T, (BankPtr)&c_bank, chunk, L[w], L[w].tab); return; }
  // (rare_flag: a word of the engine's, rare_epoch: this launch's number -- k_decode_rare goes home at once unless the word
  //  says that this launch has a chunk for it; never reset.  Kernel parameters, not DecodeArgs: the granule kernels, which
  //  keep every argument in scalar registers, slowed down by 2.5 % with two more of them)
  if (rare_flag && chunk_is_rare(a, chunk)) {
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(rare_flag, rare_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  run_chunk<DUMP, false, F32, true, false>(a, T, (BankPtr)&c_bank, chunk, L[w], L[w].tab);
}

// ... the second kernel of the launch: nothing to do unless the first one found a chunk with a frame_is_rare() frame
// (the flag is an epoch number: never reset), then those chunks, with the full code (intensity stereo, LSF)
template <bool F32>
__global__ __launch_bounds__(64, PDMP3_WAVES_PER_EU) void k_decode_rare(DecodeArgs a, GlobalTables T, int n_chunks, int always, const unsigned* rare_flag, unsigned rare_epoch) {
  __shared__ WaveLds L[1];
  const int chunk = (int)blockIdx.x;
  if (chunk >= n_chunks) return;
  if (!always) {
    if (__hip_atomic_load(rare_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != rare_epoch) return;
    if (!chunk_is_rare(a, chunk)) return;
  }
  run_chunk<false, false, F32, true, true>(a, T, (BankPtr)&c_bank, chunk, L[0], L[0].tab);
}

// One granule per wave (decode_core.h run_granule): WPW consecutive granules per workgroup, the workgroup's place in the
// chain is its blockIdx.  128 VGPRs and 9.3 KB of LDS per wave + one table block per workgroup: two workgroups = 16 waves
// per CU, four per SIMD.
// W = 16: one workgroup per CU holds the CU's sixteen waves (a launch of 2048 frames is one workgroup on every CU of
// an MI355X); W = 8 for the smaller launches: twice as many CUs share the work, two waves per SIMD -- which is also all
// the occupancy that form is compiled for (a workgroup of 8 waves is one per CU in the launches that take it; asked for
// four waves per SIMD the compiler could not get there and said so: 217 registers, occupancy 2).
template <bool F32, int W>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(W / 4, W / 4))) void k_decode_g(DecodeArgs a, GlobalTables T) {
  __shared__ WaveData L[W];
  __shared__ TabLds S;
  __shared__ GranMb mb[W];
  __shared__ unsigned tabs_ready;
  static_assert(sizeof(WaveData) * W + sizeof(TabLds) + sizeof(GranMb) * W + 16 <= 160 * 1024, "one workgroup of 16 waves per CU");
  const int tid = (int)threadIdx.x;
  const unsigned long long t_entry = a.prof ? PD_CLOCK() : 0ull;
  const int w = tid >> 6;
  const int g = (int)blockIdx.x * W + w;
  const bool valid = g < 2 * a.n_frames;
  // the wave's own input first: its round trip passes under the workgroup's table loads
  LaneRegs pf;
  if (valid) ph_prefetch(tid & 63, pf, a.spectra + (size_t)g * 1152, a.side + (size_t)g * 2);
  if (tid < W * (int)(sizeof(GranMb) / 4)) reinterpret_cast<unsigned*>(mb)[tid] = 0u;
  if (tid == 0) tabs_ready = 0u;
  __syncthreads();                       // (nothing to wait for in front of it: the waves arrive together)
  // the workgroup's tables; the line tables are for the sampling frequency the caller expects (granules of another
  // one read the global line table).  No barrier behind the loads: each wave counts itself in when its part is
  // stored (GranPos::tabs_ready) and whoever needs the tables waits for the count -- by then it is long there
  tab_load_image(tid, 64 * W, S, T, a.sf_hint);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if ((tid & 63) == 0) atomicAdd(&tabs_ready, 1u);
  if (!valid) return;
  if (a.prof && (tid & 63) == 0) a.prof[(size_t)g * kProfSlots] = t_entry;
  const GranPos gp{L, mb, w, W, &tabs_ready};
  run_granule_wave<F32>(a, T, (BankPtr)&c_bank, g, L[w], S, gp, pf);
}

// OPT-IN BUILD (make EXTRA=-DPDMP3_WITH_RING_KERNEL): measured 1.66 x slower than the engine's own choice at every size
// (profiles/r04_ring_bench.txt), so the default library does not carry its two 7 k-instruction kernels.
// Persistent form (decode_core.h run_granule_ring): a workgroup of 16 waves goes round a contiguous range of
// `frames_per_wg` frames, one granule per wave and turn; constants and tables once per wave / workgroup, hand-over through
// the LDS mailboxes as a ring, one halo per range.  One workgroup per CU (158 KB of LDS), four waves per SIMD.
#if defined(PDMP3_WITH_RING_KERNEL)
template <bool F32>
__global__ __launch_bounds__(64 * 16) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_decode_p(DecodeArgs a, GlobalTables T, int frames_per_wg) {
  constexpr int W = 16;
  __shared__ WaveData L[W];
  __shared__ TabLds S;
  __shared__ GranMb mb[W];
  __shared__ unsigned tabs_ready;
  const int tid = (int)threadIdx.x;
  const int w = tid >> 6;
  if (tid < W * (int)(sizeof(GranMb) / 4)) reinterpret_cast<unsigned*>(mb)[tid] = 0u;
  if (tid == 0) tabs_ready = 0u;
  __syncthreads();
  tab_load_image(tid, 64 * W, S, T, a.sf_hint);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if ((tid & 63) == 0) atomicAdd(&tabs_ready, 1u);
  const int f0 = (int)blockIdx.x * frames_per_wg;
  int f1 = f0 + frames_per_wg;
  if (f1 > a.n_frames) f1 = a.n_frames;
  const GranPos gp{L, mb, w, W, &tabs_ready, 1, 2 * f0, 2 * f1};
  run_granule_ring<F32>(a, T, (BankPtr)&c_bank, L[w], S, gp, f0, f1);
}
#endif

// same kernel with shader-clock stamps after every phase (tools/phase_profile.py)
__global__ __launch_bounds__(64, PDMP3_WAVES_PER_EU) void k_decode_prof(DecodeArgs a, GlobalTables T) {
  __shared__ WaveLds L[1];
  const int w = threadIdx.x >> 6;
  run_chunk<false, true, false, true, false>(a, T, (BankPtr)&c_bank, (int)blockIdx.x, L[w], L[w].tab);   // (the copy ordinary chunks run)
}

// LSF launches (launch_decode): record-frame p of the output = the first granules ([0][ch]: 2 x 576 int16 of spectra, 2 x 128
// bytes of side records) of the input's frames 2 p and 2 p + 1; a missing second one (odd n) is zeroed and never decoded.
__global__ __launch_bounds__(256) void k_lsf_pair(const int16_t* in_sp, const pdmp3_gc_side* in_sd, int n_frames, int16_t* out_sp, pdmp3_gc_side* out_sd) {
  const int p = blockIdx.x;
  const uint4 zero = make_uint4(0, 0, 0, 0);
  for (int gr = 0; gr < 2; gr++) {
    const int f = 2 * p + gr;
    const uint4* ssp = reinterpret_cast<const uint4*>(in_sp + (size_t)f * 2304);
    const uint4* ssd = reinterpret_cast<const uint4*>(in_sd + (size_t)f * 4);
    uint4* dsp = reinterpret_cast<uint4*>(out_sp + (size_t)p * 2304 + gr * 1152);
    uint4* dsd = reinterpret_cast<uint4*>(out_sd + (size_t)p * 4 + gr * 2);
    for (int k = threadIdx.x; k < 144 + 16; k += blockDim.x) {
      if (k < 144) dsp[k] = f < n_frames ? ssp[k] : zero;
      else dsd[k - 144] = f < n_frames ? ssd[k - 144] : zero;
    }
  }
}

__global__ __launch_bounds__(64) void k_generate(uint64_t seed, int64_t first, int16_t* spectra, pdmp3_gc_side* side) {
  const int64_t gc = blockIdx.x;           // (frame_local*4 + gr*2 + ch)
  const int64_t f = gc >> 2;
  gen_gc(seed, first + f, (unsigned)((gc >> 1) & 1), (unsigned)(gc & 1), (int)threadIdx.x,
         spectra + gc * 576, side + gc);
}

// ---------------------------------------------------------------------------
// main-data decoding on the device (unpack_core.h)
// ---------------------------------------------------------------------------
constexpr int kUnpackLanes = 64;                        // one wave = 16 frames per pass decodes ...
constexpr int kUnpackThreads = 256;                     // ... four bring the tables and the rows into LDS and zero the output
constexpr int kUnpackRows = kUnpackLanes / 4;
constexpr int kRowBytes = PDMP3_RESERVOIR_BYTES;
// LDS row stride in 32-bit words, chosen ODD: the 64 lanes read their rows at about the same offset at the same
// time, and with the natural stride (516 words) that is 8 of the 32 banks for the whole wave
constexpr int kRowStrideW = kRowBytes / 4 + 1;          // 517
constexpr int kFrameBitsW = sizeof(pdmp3_frame_bits) / 4;   // 20

// LDS: the 34 KB table blob, the 16 reservoir rows the workgroup works on (33 KB) and their side info, brought in with
// coalesced loads, and the 16 KB ring of symbol records: 85 KB, one workgroup per CU.  The kernel is a long dependent
// chain per lane -- where does the next code word start -- and a window of 2048 frames has only 8192 of them (128
// waves on 1024 SIMDs), so what counts is the length of that chain: WAVE 0 of the workgroup walks the 64 bit streams
// (unpack_core.h unpack_step: two table lookups and an add per symbol, pairs and quads in one loop) and leaves a record
// per symbol and lane in the ring; WAVES 1-3 take turns with the ring's rows (row i belongs to wave 1 + i % 3), read
// linbits and signs and store the lines (unpack_value).  Round 2's single loop did all of it in wave 0, a loop per
// symbol kind: 406 trips of ~700 cycles per window, 140 us; this one: <= 288 trips of the walker's half.
// The lines go straight to HBM, into spectra that two of the value waves zero with coalesced stores first.
constexpr int kRingRows = 16;                              // trips the walker may be ahead of the value waves (16, not 32: the ring's 8 KB
                                                           // are what lets TWO workgroups share a CU -- 80.0 KB each --, which is worth more on the
                                                           // windows of 4096 frames the whole-stream decoder now uses: 17.0 -> 18.5 M frames/s)
constexpr int kRingCheck = 8;                              // ... looked at every so many trips
static_assert(kRingRows % kRingCheck == 0 && kRingCheck >= 3, "blocks of trips do not wrap around the ring");
struct UnpackRing {
  SymRec rec[kRingRows][kUnpackLanes];                     // tag (trip / kRingRows) & 3 in bits 30-31 of .x: the row is of THIS turn
  unsigned next[3];                                        // value wave c: the first trip it has not taken yet
  unsigned zeroed;                                         // value waves 2 and 3: my half of the pass's spectra is zero
};
typedef unsigned ring_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ring_u32x2 ring_load(const SymRec* p) {
  return *(const volatile __attribute__((address_space(3))) ring_u32x2*)(p);
}
__device__ __forceinline__ void ring_store(SymRec* p, uint32_t x, uint32_t y) {
  ring_u32x2 v;
  v.x = x; v.y = y;
  *(volatile __attribute__((address_space(3))) ring_u32x2*)(p) = v;
}

__global__ __launch_bounds__(kUnpackThreads) void k_unpack(const UnpackTables* tabs, const pdmp3_frame_bits* bits,
                                                            const uint8_t* res, int n_frames, int16_t* spectra,
                                                            GcRaw* raw, int tab_n16, unsigned long long* prof) {
  __shared__ UnpackTables U;
  __shared__ uint32_t rows[kUnpackRows * kRowStrideW + 4];
  __shared__ uint32_t fbits[kUnpackRows * kFrameBitsW];
  __shared__ UnpackRing ring;
  // development only (PDMP3_HIP_UNPACK_PROF=1): s_memtime of workgroup's wave 0 at the steps of its first pass
#define PD_UP_STAMP(k) do { if (prof && threadIdx.x == 0) prof[blockIdx.x * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
  PD_UP_STAMP(0);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int fl = lane >> 2, g = lane & 3;                  // the granule-channel of this lane, in every wave
  bool first = true;
  for (int f0 = blockIdx.x * kUnpackRows; f0 < n_frames; f0 += gridDim.x * kUnpackRows) {
    const int nrows = n_frames - f0 < kUnpackRows ? n_frames - f0 : kUnpackRows;
    if (!first) __syncthreads();                           // (previous pass done with the buffers)
    {
      // the pass's rows, 16 bytes per lane and trip, all asked for before the tables (first pass) so that the two
      // round trips to HBM overlap
      constexpr int kRow16 = kRowBytes / 16, kRowTrips = (kUnpackRows * kRow16 + kUnpackThreads - 1) / kUnpackThreads;
      static_assert(kRowBytes % 16 == 0, "rows are copied 16 bytes at a time");
      const uint4* src = reinterpret_cast<const uint4*>(res + (size_t)f0 * kRowBytes);
      uint4 rv[kRowTrips];
      PD_UNROLL for (int k = 0; k < kRowTrips; ++k) {
        const int i = (int)threadIdx.x + k * kUnpackThreads;
        if (i < nrows * kRow16) rv[k] = src[i];
      }
      if (first) {
        const uint4* tsrc = reinterpret_cast<const uint4*>(tabs);
        uint4* dst = reinterpret_cast<uint4*>(&U);
        for (int i = threadIdx.x; i < tab_n16; i += kUnpackThreads) dst[i] = tsrc[i];
      }
      PD_UNROLL for (int k = 0; k < kRowTrips; ++k) {
        const int i = (int)threadIdx.x + k * kUnpackThreads;
        if (i < nrows * kRow16) {
          const int r = i / kRow16, c = i - r * kRow16;
          uint32_t* d = rows + r * kRowStrideW + 4 * c;    // (unpack_core.h PD_ROW_BE: big-endian words as numbers)
          d[0] = __builtin_bswap32(rv[k].x); d[1] = __builtin_bswap32(rv[k].y);
          d[2] = __builtin_bswap32(rv[k].z); d[3] = __builtin_bswap32(rv[k].w);
        }
      }
      const uint32_t* fsrc = reinterpret_cast<const uint32_t*>(bits + f0);
      for (int i = threadIdx.x; i < nrows * kFrameBitsW; i += kUnpackThreads) fbits[i] = fsrc[i];
      for (int i = threadIdx.x; i < kRingRows * kUnpackLanes; i += kUnpackThreads) (&ring.rec[0][0])[i].x = 3u << 30;   // "the turn before trip 0"
      if (threadIdx.x < 3) ring.next[threadIdx.x] = threadIdx.x;
      if (threadIdx.x == 3) ring.zeroed = 0;
    }
    __syncthreads();
    PD_UP_STAMP(1);
    const size_t idx = (size_t)(f0 + fl) * 4 + g;
    const uint8_t* row = reinterpret_cast<const uint8_t*>(rows + fl * kRowStrideW);
    int16_t* is = spectra + idx * 576;
    SymPlan P;
    SymState st;
    bool live = false;
    if (wave == 0) {
      // ---- the walker
      if (fl < nrows) live = unpack_plan(U, *reinterpret_cast<const pdmp3_frame_bits*>(fbits + fl * kFrameBitsW), g, P, st);
      PD_UP_STAMP(2);
      Win2 w;
      w2_open(w, row, live ? st.pos : 0u);
      if (!live) { st.pos = 0; st.line = 0; }
      // the plan in REGISTERS (as a struct it stays in memory and every trip starts with a load of its table base)
      unsigned qt0 = P.tab0, qt1 = P.tab1, qt2 = P.tab2, qq = P.qbase, qe0 = P.e0, qe1 = P.e1, qn = P.nbig, qend = P.end;
      PD_PIN(qt0); PD_PIN(qt1); PD_PIN(qt2); PD_PIN(qq); PD_PIN(qe0); PD_PIN(qe1); PD_PIN(qn); PD_PIN(qend);
      // kRingCheck trips at a time: room in the ring and "is any lane still at it" are looked at once per block (a lone
      // wave issues an instruction every ~6 cycles whatever it is: the loop's own bookkeeping was a third of a trip)
      const unsigned ztab = U.book_base[kZeroBook];
      unsigned trip = 0;
      for (;; trip += kRingCheck) {
        for (;;) {                                         // room for the block's rows?  (rarely not: three waves take them out)
          const unsigned n0 = PD_LDS_FLAG(&ring.next[0]), n1 = PD_LDS_FLAG(&ring.next[1]), n2 = PD_LDS_FLAG(&ring.next[2]);
          const unsigned lo = n0 < n1 ? (n0 < n2 ? n0 : n2) : (n1 < n2 ? n1 : n2);
          if (__builtin_amdgcn_readfirstlane(lo) + kRingRows >= trip + kRingCheck) break;
          PD_SLEEP();
        }
        if (!__any(live && sym_active(qn, qend, st))) break;
        SymRec* blk = &ring.rec[trip % kRingRows][lane];   // (kRingRows is a multiple of kRingCheck: the block does not wrap)
        const uint32_t tag = ((trip / kRingRows) & 3u) << 30;
        PD_UNROLL for (int j = 0; j < kRingCheck; ++j) {
          const SymRec r = unpack_step(U.lut, qt0, qt1, qt2, qq, ztab, qe0, qe1, qn, live && sym_active(qn, qend, st), st, w);
          ring_store(blk + j * kUnpackLanes, r.x | tag, r.y);
        }
      }
      PD_UP_STAMP(3);
      if (prof && threadIdx.x == 0) prof[blockIdx.x * 8 + 6] = trip;
      for (int k = 0; k < 3; ++k)                          // one end row per value wave (the block's room was checked)
        ring_store(&ring.rec[(trip + k) % kRingRows][lane], (kRecNopLine << 16) | (((trip + k) / kRingRows) & 3u) << 30, kRecEnd);
    } else {
      // ---- a value wave: rows wave - 1, wave + 2, ...  The first one writes the records' side fields and scalefactors
      // before it joins in (the ring holds what the walker produces meanwhile; the other two are taking rows out already)
      // and the other two zero the pass's spectra (lines are stored only where the stream has any) -- all of it beside
      // the walker's first trips instead of in front of them
      if (wave == 1) {
        if (fl < nrows)
          unpack_scalefactors(U, row, *reinterpret_cast<const pdmp3_frame_bits*>(fbits + fl * kFrameBitsW), g, raw + idx);
      } else {
        uint4* z = reinterpret_cast<uint4*>(spectra + (size_t)f0 * 4 * 576);
        for (int i = (int)threadIdx.x - 128; i < nrows * 4 * 72; i += 128) z[i] = make_uint4(0, 0, 0, 0);
        PD_VMEM_DRAIN();                                   // the zeroes have arrived before any wave stores a line
        if (lane == 0) atomicAdd(&ring.zeroed, 1u);
      }
      while (PD_UNIFORM(PD_LDS_FLAG(&ring.zeroed)) < 2) PD_SLEEP();
      asm volatile("" ::: "memory");
      for (unsigned trip = (unsigned)wave - 1;; trip += 3) {
        const SymRec* slot = &ring.rec[trip % kRingRows][lane];
        const unsigned tag = (trip / kRingRows) & 3u;
        ring_u32x2 r;
        for (;;) {
          r = ring_load(slot);
          if (__all((r.x >> 30) == tag)) break;
          __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0) PD_LDS_FLAG(&ring.next[wave - 1]) = trip + 3;
        if (r.y & kRecEnd) break;
        unpack_value(row, SymRec{r.x, r.y}, is);
      }
    }
    __syncthreads();                                       // every line of every record is stored, and wave 1's part of `raw`
    PD_UP_STAMP(4);
    if (wave == 0 && live) unpack_tail(U, U.lut, row, P, st, is, raw + idx);
    PD_UP_STAMP(5);
    first = false;
  }
#undef PD_UP_STAMP
}

// reservoir rows from the pool (unpack_core.h row_chunk16): a wave per frame, 16 bytes per lane and trip (a 4-byte word
// per thread was 21 us for 8192 frames, 1.2 TB/s)
constexpr int kRowsWaves = 4;
__global__ __launch_bounds__(64 * kRowsWaves) void k_rows(const pdmp3_row_desc* desc, const uint8_t* pool, uint8_t* rows, int n_frames) {
  const int f = blockIdx.x * kRowsWaves + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (f >= n_frames) return;
  const pdmp3_row_desc* d = desc + f;
  uint4* out = reinterpret_cast<uint4*>(rows + (size_t)f * kRowBytes);
  for (int c = lane; c < kRowBytes / 16; c += 64) {
    uint32_t v[4];
    row_chunk16(d, pool, 16u * (unsigned)c, v);
    out[c] = make_uint4(v[0], v[1], v[2], v[3]);
  }
}

// The merge (unpack_core.h "The same merge by BLOCKS"): a workgroup per block of 32 frames, a lane per surviving value.
// k_merge_outcome: the block's 10 KB of merge input into LDS with 16-byte loads, every lane walks the 32 frames for its
// slot, a row of outcomes goes out; the workgroup that is through last among the eight of a super-block (a counter that
// wraps back to zero by itself) composes the eight rows into the super-block's.  What the eight exchange travels as the
// granule kernel's hand-overs do: device-scope relaxed atomic accesses, ordered by the wait for the stores and the counter.
constexpr int kMergeRaw16 = kMergeBlk * 4 * (int)sizeof(GcRaw) / 16;              // 640
constexpr int kMergeBits16 = kMergeBlk * (int)sizeof(pdmp3_frame_bits) / 16;      // 160
constexpr int kMergeSide16 = kMergeBlk * PDMP3_FRAME_SIDE_BYTES / 16;             // 1024
constexpr int kMergeRow16 = kMergeLanes * 4 / 16;                                 // a row of outcomes: 64
constexpr int kMergeStageRows = 40;                                               // rows of outcomes a block looks at in one go (a window of 8192 frames: 38 at most)
constexpr int kMergeRawPer = (kMergeRaw16 + kMergeLanes - 1) / kMergeLanes, kMergeBitsPer = (kMergeBits16 + kMergeLanes - 1) / kMergeLanes;
static_assert(sizeof(GcRaw) % 16 == 0 && sizeof(pdmp3_frame_bits) % 16 == 0 && kMergeSlots <= kMergeLanes, "the merge copies 16 bytes at a time");
__global__ __launch_bounds__(kMergeLanes) void k_merge_outcome(const GcRaw* raw, const pdmp3_frame_bits* bits, int n_frames, uint32_t* outc,
                                                                uint32_t* sup, unsigned* counters) {
  __shared__ uint4 raw_s[kMergeRaw16];
  __shared__ uint8_t fr_s[kMergeBlk];
  __shared__ unsigned last_s;
  const int b = blockIdx.x, f0 = b * kMergeBlk, t = threadIdx.x;
  const int nb = n_frames - f0 < kMergeBlk ? n_frames - f0 : kMergeBlk;
  const uint4* src = reinterpret_cast<const uint4*>(raw + (size_t)f0 * 4);
  for (int i = t; i < nb * 20; i += kMergeLanes) raw_s[i] = src[i];
  if (t < nb) fr_s[t] = bits[f0 + t].frame;
  __syncthreads();
  unsigned o = 0;
  if (t < kMergeSlots) o = merge_block_outcome(t, reinterpret_cast<const GcRaw*>(raw_s), fr_s, nb, PD_UNIFORM(merge_wave_kind(t & ~63)));
  const int sb = b / kMergeSuper, b0 = sb * kMergeSuper;
  const int n_in = (int)gridDim.x - b0 < kMergeSuper ? (int)gridDim.x - b0 : kMergeSuper;
  if (n_in == 1) {                                         // (a super-block of one block: its outcome is the block's)
    outc[(size_t)b * kMergeLanes + t] = o;
    sup[(size_t)sb * kMergeLanes + t] = o;
    return;
  }
  __hip_atomic_store(outc + (size_t)b * kMergeLanes + t, o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  PD_VMEM_DRAIN();
  __syncthreads();
  if (t == 0) last_s = atomicInc(counters + sb, (unsigned)n_in - 1u) == (unsigned)n_in - 1u;
  __syncthreads();
  if (!last_s) return;
  const int tw = t < kMergeSlots ? merge_twin(t) : -1;
  unsigned oj[kMergeSuper], o0j[kMergeSuper];
  PD_UNROLL for (int j = 0; j < kMergeSuper; j++) {
    const int jj = j < n_in ? j : n_in - 1;
    oj[j] = PD_LOAD_DEVICE(outc + (size_t)(b0 + jj) * kMergeLanes + t);
    o0j[j] = tw >= 0 ? PD_LOAD_DEVICE(outc + (size_t)(b0 + jj) * kMergeLanes + tw) : 0u;
  }
  unsigned acc = 0, acc0 = 0;
  PD_UNROLL for (int j = 0; j < kMergeSuper; j++)
    if (j < n_in) merge_compose_step(oj[j], o0j[j], acc, acc0);
  sup[(size_t)sb * kMergeLanes + t] = acc;
}

// k_merge_apply: block b asks for its frames' inputs, walks the rows of outcomes in front of it -- super-blocks 0 .. b / 8 - 1,
// then the blocks of its own super-block -- from the values the window before left, then its own frames; the records are
// built in LDS -- zeroes, the frame's own side fields (a thread per granule-channel), scalefactors and count1 (a thread per
// slot: byte stores) -- and leave with 16-byte stores: the whole of `side` is written here, once.  The last block leaves
// the values for the next window.
__global__ __launch_bounds__(kMergeLanes) void k_merge_apply(const GcRaw* raw, const pdmp3_frame_bits* bits, int n_frames, const uint32_t* outc,
                                                              const uint32_t* sup, const uint16_t* state_in, uint16_t* state_out, pdmp3_gc_side* side) {
  __shared__ uint4 raw_s[kMergeRaw16];
  __shared__ uint4 bits_s[kMergeBits16];
  __shared__ uint4 img_s[kMergeSide16];
  __shared__ uint4 stage_s[kMergeStageRows * kMergeRow16];
  __shared__ uint32_t meta_s[kMergeBlk];
  __shared__ uint8_t trash_s[kMergeLanes];
  const int b = blockIdx.x, f0 = b * kMergeBlk, t = threadIdx.x;
  const int nb = n_frames - f0 < kMergeBlk ? n_frames - f0 : kMergeBlk;
  uint4 rv[kMergeRawPer], bv[kMergeBitsPer];
  uint32_t hw = 0;
  {
    if (t < nb) hw = *reinterpret_cast<const uint32_t*>(bits + f0 + t);      // frame | scfsi | iso
    const uint4* src = reinterpret_cast<const uint4*>(raw + (size_t)f0 * 4);
    // (values, not arrays in memory: every element is assigned on every path)
    PD_UNROLL for (int k = 0; k < kMergeRawPer; k++) { const int i = t + k * kMergeLanes; rv[k] = make_uint4(0, 0, 0, 0); if (i < nb * 20) rv[k] = src[i]; }
    const uint4* bsrc = reinterpret_cast<const uint4*>(bits + f0);
    PD_UNROLL for (int k = 0; k < kMergeBitsPer; k++) { const int i = t + k * kMergeLanes; bv[k] = make_uint4(0, 0, 0, 0); if (i < nb * 5) bv[k] = bsrc[i]; }
  }
  const int tw = t < kMergeSlots ? merge_twin(t) : -1;
  const unsigned kind = PD_UNIFORM(merge_wave_kind(t & ~63));
  unsigned val = t < kMergeSlots ? state_in[t] : 0u, val0 = tw >= 0 ? state_in[tw] : 0u;
  const int nsup = b / kMergeSuper, total = nsup + b % kMergeSuper;
  for (int c0 = 0; c0 < total; c0 += kMergeStageRows) {
    const int nc = total - c0 < kMergeStageRows ? total - c0 : kMergeStageRows;
    if (c0) __syncthreads();
    for (int i = t; i < nc * kMergeRow16; i += kMergeLanes) {
      const int r = c0 + i / kMergeRow16;
      const uint32_t* row = r < nsup ? sup + (size_t)r * kMergeLanes : outc + (size_t)(nsup * kMergeSuper + r - nsup) * kMergeLanes;
      stage_s[i] = reinterpret_cast<const uint4*>(row)[i % kMergeRow16];
    }
    __syncthreads();
    if (t < kMergeSlots) {
      if (kind & 1u) merge_carry_rows<true>(reinterpret_cast<const uint32_t*>(stage_s), nc, t, tw, val, val0);
      else merge_carry_rows<false>(reinterpret_cast<const uint32_t*>(stage_s), nc, t, tw, val, val0);
    }
  }
  PD_UNROLL for (int k = 0; k < kMergeRawPer; k++) { const int i = t + k * kMergeLanes; if (i < nb * 20) raw_s[i] = rv[k]; }
  PD_UNROLL for (int k = 0; k < kMergeBitsPer; k++) { const int i = t + k * kMergeLanes; if (i < nb * 5) bits_s[i] = bv[k]; }
  for (int i = t; i < kMergeSide16; i += kMergeLanes) img_s[i] = make_uint4(0, 0, 0, 0);
  if (t < nb) meta_s[t] = merge_frame_meta(hw & 0xffu, hw >> 24);
  __syncthreads();
  const pdmp3_frame_bits* F = reinterpret_cast<const pdmp3_frame_bits*>(bits_s);
  pdmp3_gc_side* img = reinterpret_cast<pdmp3_gc_side*>(img_s);
  if ((t >> 2) < nb) side_fields(F[t >> 2], t & 3, img + t);
  if (t < kMergeSlots) {
    val = merge_block_apply(t, val, val0, reinterpret_cast<const GcRaw*>(raw_s), meta_s, nb, img, trash_s + t, kind);
    if (f0 + nb == n_frames) state_out[t] = (uint16_t)val;
  }
  __syncthreads();
  uint4* dst = reinterpret_cast<uint4*>(side + (size_t)f0 * 4);
  for (int i = t; i < nb * (PDMP3_FRAME_SIDE_BYTES / 16); i += kMergeLanes) dst[i] = img_s[i];
}

// Scratch of a chained launch (DecodeArgs::chain_*): launches that are ordered one after the other share a buffer --
// those of one pdmp3_hip_stream (its slots' kernels are chained by the state event), or those of bare calls on one HIP
// stream; launches that may overlap never do.
struct ChainBuf {
  const void* key;          // whose launches share it: a pdmp3_hip_stream's state scratch, or the HIP stream of bare calls
  bool used;
  int cap;                  // frames
  unsigned epoch;           // of the last launch that used it; flags of older launches are smaller, never equal
  float* state;             // cap x 2 kGranFloats floats
  unsigned* flag;           // cap x 4 flags
  unsigned long long last_use;   // launch counter value of its latest use (the least recently used one is evicted)
  // bare calls on this HIP stream (no stream object, which has its own): where the kernel leaves the closing state before it
  // is copied over the caller's, and an LSF launch's regrouped records.  Stream-ordered like the rest: a call's launches are
  // through with them before the next call's on the same stream start.
  float* state_tmp;
  int16_t* pair_sp; pdmp3_gc_side* pair_sd; int pair_cap;   // record-frames
};
// what a launch gets of it (a copy made under the lock: another thread's launch may replace the buffers right after)
struct ChainUse { float* state; unsigned* flag; unsigned epoch; };
constexpr int kChainBufs = 32;

constexpr int kRareSlots = 4096;      // a flag word per launch, taken round robin: two launches share one only if 4096 others lie between them
struct pdmp3_hip_ctx {
  int device;
  int wave_slots;           // waves of k_decode the device holds at once (CUs x 4 SIMDs x 2)
  UnpackTables* d_unpack;
  int unpack_n16;                // its used part, in 16-byte units
  unsigned long long* d_uprof;   // development only: PDMP3_HIP_UNPACK_PROF=1
  float* d_pow43;
  uint16_t* d_linetab;
  float* d_win;
  float* d_frag;            // frag_long [10][64] | frag_short [10][64] | frag_mat [8][64] | taps [16][64]
  void* d_tab_image;        // [kNumSfreq] TabLds images
  unsigned* d_rare_flags;   // [kRareSlots] epoch numbers (DecodeArgs::rare_flag): "this launch of chunks holds a chunk for k_decode_rare"
  std::atomic<unsigned> rare_epoch;
  int chain_mode;           // PDMP3_HIP_CHAIN=0: independent chunks with halos everywhere; otherwise launches up to
                            // gran_max_frames take the granule kernel (k_decode_g)
  int gran_max_frames;      // launches up to this many frames take the granule kernel (PDMP3_HIP_GRAN_MAX)
  int ring_min_frames;      // launches from this many frames on take the persistent granule kernel (PDMP3_HIP_RING_MIN; 0: never)
  int cus;
  int wave_slots_gran;      // waves of k_decode_g the device holds at once (CUs x 4 SIMDs x 4)
  unsigned debug_flags;     // PDMP3_HIP_DEBUG_FAR_TIMEOUT=1: every wait for another workgroup gives up at once (tests)
  std::atomic<int> last_kind;   // PDMP3_HIP_LAUNCH_* of the latest decode launch, any thread (reports only)
  int direct_max_frames;    // record batches of a stream up to this size run on the pinned host buffers directly (PDMP3_HIP_DIRECT_MAX)
  int gran_w8;              // development: PDMP3_HIP_GRAN_W=8 -- every launch of the granule kernel in workgroups of 8 waves
  int sf_hint;              // sampling-frequency index the granule kernel's line tables are loaded for (PDMP3_HIP_SF_HINT; 0 = 44.1 kHz)
  std::mutex chain_mu;
  unsigned long long chain_clock;
  ChainBuf chain[kChainBufs];
};

static thread_local char g_err[256] = "";

static int fail(int code, const char* what, hipError_t e) {
  snprintf(g_err, sizeof g_err, "%s: %s", what, e == hipSuccess ? "" : hipGetErrorString(e));
  return code;
}

#define HIP_TRY(call, what)                                   \
  do {                                                        \
    hipError_t e_ = (call);                                   \
    if (e_ != hipSuccess) return fail(PDMP3_HIP_EDEVICE, what, e_); \
  } while (0)

extern "C" const char* pdmp3_hip_last_error(void) { return g_err; }
// (for the library's other translation unit, node.hip: the calling thread's error text)
extern "C" void pdmp3_hip_set_error_(const char* text) { snprintf(g_err, sizeof g_err, "%s", text ? text : ""); }

extern "C" size_t pdmp3_hip_state_bytes(void) { return (size_t)kStateFloats * sizeof(float); }
extern "C" int pdmp3_hip_pci_bus_id(const pdmp3_hip_ctx* c, char* buf, int len) {
  if (!c || !buf || len < 16) return PDMP3_HIP_EINVAL;
  if (hipDeviceGetPCIBusId(buf, len, c->device) != hipSuccess) { buf[0] = 0; return PDMP3_HIP_EDEVICE; }
  return PDMP3_HIP_OK;
}

extern "C" int pdmp3_hip_last_launch_kind(const pdmp3_hip_ctx* c) { return c ? c->last_kind.load(std::memory_order_relaxed) : PDMP3_HIP_LAUNCH_NONE; }

extern "C" void pdmp3_hip_destroy(pdmp3_hip_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipFree(c->d_pow43);
  (void)hipFree(c->d_linetab);
  (void)hipFree(c->d_win);
  (void)hipFree(c->d_frag);
  (void)hipFree(c->d_tab_image);
  (void)hipFree(c->d_rare_flags);
  (void)hipFree(c->d_unpack);
  (void)hipFree(c->d_uprof);
  for (ChainBuf& b : c->chain) { (void)hipFree(b.state); (void)hipFree(b.flag); }
  delete c;
}

extern "C" int pdmp3_hip_create(int device, pdmp3_hip_ctx** out) {
  if (!out) return fail(PDMP3_HIP_EINVAL, "pdmp3_hip_create: out is NULL", hipSuccess);
  *out = nullptr;
  HIP_TRY(hipSetDevice(device), "hipSetDevice");
  HostTables H;
  build_host_tables(H);
  if (!H.ldexp_forms_exact)
    return fail(PDMP3_HIP_EDEVICE, "this host's libm pow() disagrees with the device's ldexp forms of 2^(k/4), 2^(-n/2)", hipSuccess);
  pdmp3_hip_ctx* c = new (std::nothrow) pdmp3_hip_ctx();      // (value-initialised: every pointer null)
  if (!c) return fail(PDMP3_HIP_ENOMEM, "new", hipSuccess);
  c->device = device;
  {
    const char* e = getenv("PDMP3_HIP_CHAIN");
    c->chain_mode = (e && *e == '0') ? 0 : 1;
  }
  {
    hipDeviceProp_t prop;
    const int cus = (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    c->wave_slots = cus * 4 * PDMP3_WAVES_PER_EU;
    c->wave_slots_gran = cus * 4 * 4;
    const char* dm = getenv("PDMP3_HIP_DIRECT_MAX");
    c->direct_max_frames = dm ? atoi(dm) : 32;
    { const char* gw = getenv("PDMP3_HIP_GRAN_W"); c->gran_w8 = gw && atoi(gw) == 8; }
    const char* h = getenv("PDMP3_HIP_SF_HINT");
    c->sf_hint = (h && *h >= '0' && *h <= '2') ? *h - '0' : 0;
    const char* d = getenv("PDMP3_HIP_DEBUG_FAR_TIMEOUT");
    c->debug_flags = (d && *d == '1') ? PD_DEBUG_FAR_TIMEOUT : 0u;
    const char* e = getenv("PDMP3_HIP_GRAN_MAX");
    c->cus = cus;
    { const char* rm = getenv("PDMP3_HIP_RING_MIN"); c->ring_min_frames = rm ? atoi(rm) : 0; }
    c->gran_max_frames = e && atoi(e) > 0 ? atoi(e) : c->wave_slots_gran * 3;     // (measured on MI355X: faster than chunks with halos up to about 14000 frames)
  }
  // every failure from here on releases what was allocated so far (pdmp3_hip_destroy takes a partly built context)
  UnpackTables* U = new UnpackTables;
  const char* what = nullptr;
  hipError_t e = hipSuccess;
#define CREATE_STEP(call, text) if ((e = (call)) != hipSuccess) { what = text; break; }
  do {
    if (!build_unpack_tables(*U)) { what = "Huffman lookup tables exceed kHuffLutMax"; break; }
    CREATE_STEP(hipMemcpyToSymbol(HIP_SYMBOL(c_bank), &H.cb, sizeof(ConstBank)), "upload const bank")
    CREATE_STEP(hipMalloc(&c->d_pow43, H.pow43.size() * sizeof(float)), "hipMalloc pow43")
    CREATE_STEP(hipMalloc(&c->d_linetab, H.linetab.size() * sizeof(uint16_t)), "hipMalloc linetab")
    CREATE_STEP(hipMalloc(&c->d_win, H.win.size() * sizeof(float)), "hipMalloc win")
    CREATE_STEP(hipMalloc(&c->d_frag, (10 + 10 + 8 + 16) * 64 * sizeof(float)), "hipMalloc frag")
    CREATE_STEP(hipMalloc(&c->d_tab_image, kNumSfreq * sizeof(TabLds)), "hipMalloc table images")
    CREATE_STEP(hipMalloc((void**)&c->d_unpack, sizeof(UnpackTables)), "hipMalloc unpack tables")
    { const char* up = getenv("PDMP3_HIP_UNPACK_PROF");
      if (up && *up == '1') CREATE_STEP(hipMalloc((void**)&c->d_uprof, 2048 * 8 * sizeof(unsigned long long)), "hipMalloc unpack prof") }
    CREATE_STEP(hipMemcpy(c->d_pow43, H.pow43.data(), H.pow43.size() * sizeof(float), hipMemcpyHostToDevice), "upload pow43")
    CREATE_STEP(hipMemcpy(c->d_linetab, H.linetab.data(), H.linetab.size() * sizeof(uint16_t), hipMemcpyHostToDevice), "upload linetab")
    CREATE_STEP(hipMemcpy(c->d_win, H.win.data(), H.win.size() * sizeof(float), hipMemcpyHostToDevice), "upload win")
    CREATE_STEP(hipMemcpy(c->d_frag, H.frag_long.data(), 10 * 64 * sizeof(float), hipMemcpyHostToDevice), "upload frag_long")
    CREATE_STEP(hipMemcpy(c->d_frag + 10 * 64, H.frag_short.data(), 10 * 64 * sizeof(float), hipMemcpyHostToDevice), "upload frag_short")
    CREATE_STEP(hipMemcpy(c->d_frag + 20 * 64, H.frag_mat.data(), 8 * 64 * sizeof(float), hipMemcpyHostToDevice), "upload frag_mat")
    CREATE_STEP(hipMemcpy(c->d_frag + 28 * 64, H.taps.data(), 16 * 64 * sizeof(float), hipMemcpyHostToDevice), "upload taps")
    CREATE_STEP(hipMemcpy(c->d_tab_image, H.tab_image.data(), kNumSfreq * sizeof(TabLds), hipMemcpyHostToDevice), "upload table images")
    CREATE_STEP(hipMalloc((void**)&c->d_rare_flags, kRareSlots * sizeof(unsigned)), "hipMalloc rare flags")
    CREATE_STEP(hipMemset(c->d_rare_flags, 0, kRareSlots * sizeof(unsigned)), "hipMemset rare flags")
    c->rare_epoch.store(1);
    CREATE_STEP(hipMemcpy(c->d_unpack, U, sizeof(UnpackTables), hipMemcpyHostToDevice), "upload unpack tables")
    c->unpack_n16 = (int)((offsetof(UnpackTables, lut) + (size_t)U->n_lut * 4 + 15) / 16);
    CREATE_STEP(hipDeviceSynchronize(), "sync after uploads")
  } while (0);
#undef CREATE_STEP
  delete U;
  if (what) {
    pdmp3_hip_destroy(c);
    return fail(PDMP3_HIP_EDEVICE, what, e);
  }
  *out = c;
  return PDMP3_HIP_OK;
}

// Frames per chunk (= per wave).  The kernel holds 2 waves per SIMD, so a launch runs in rounds of `slots` waves
// (2048 on MI355X: 256 CUs x 4 SIMDs x 2); a partly filled last round costs as much as a full one (measured: 131072
// frames at 48 frames per chunk = 1.33 rounds take 20 % longer than at 32 or 64).  So: the fewest rounds that keep a
// chunk at <= 32 frames (halo overhead 2-3 granules per chunk), and the chunk size that spreads the frames evenly
// over them.  Up to one round of frames: one frame per chunk (measured fastest for the 2048-frame batch).
static int auto_chunk(int n_frames, int slots) {
  if (slots < 64) slots = 2048;
  if (n_frames <= slots) return 1;
  const long long per_round_max = (long long)slots * 32;
  const long long rounds = (n_frames + per_round_max - 1) / per_round_max;
  const long long L = (n_frames + slots * rounds - 1) / (slots * rounds);
  return (int)(L < 1 ? 1 : L);
}

// The scratch of a chained launch of n_frames frames on stream s; false: none to be had, the chunks stay independent.
// NO hipMallocAsync in here (until round 6 the scratch below, a bare call's state block and an LSF launch's regrouped records
// were stream-ordered allocations).  On ROCm 7.0.2 / MI355X memory fresh from the stream-ordered allocator can lose the writes of
// the first kernels that touch it, in a process's first moments: tools/ubench/malloc_async_probe.cpp -- allocate, one trivial
// kernel writes, the next reads -- finds up to 2304 of 2560 16-byte words unwritten in 7 of 300 fresh processes (hipMalloc: 0 of
// 300; a stream synchronise between allocation and use: 2 of 300).  It showed as an LSF stream's second batch through the
// streaming API decoding single channels of single frames from zero spectra, in 3-30 % of fresh processes depending on the box
// (tests/fuzz_gpu.py -> tests/test_gpu_lsf.py's CLI test).  Everything is hipMalloc'ed once now and kept: a free or a regrow
// synchronises, which happens when a stream's launches grow, not per launch.
static void chain_free(ChainBuf* b) {                      // (hipFree waits for whatever still uses the memory)
  (void)hipFree(b->state); (void)hipFree(b->flag); (void)hipFree(b->state_tmp); (void)hipFree(b->pair_sp); (void)hipFree(b->pair_sd);
  *b = ChainBuf{};
}
static ChainBuf* chain_entry(pdmp3_hip_ctx* c, const void* key) {   // (c->chain_mu held)
  ChainBuf* b = nullptr;
  for (ChainBuf& x : c->chain) if (x.used && x.key == key) { b = &x; break; }
  if (!b) for (ChainBuf& x : c->chain) if (!x.used) { b = &x; *b = ChainBuf{}; b->used = true; b->key = key; break; }
  if (!b) {
    // every slot is taken: the least recently used one goes (a HIP stream that bare calls once ran on may be long gone,
    // and its scratch -- 17 KB per frame -- would otherwise stay until the engine is destroyed)
    ChainBuf* lru = &c->chain[0];
    for (ChainBuf& x : c->chain) if (x.last_use < lru->last_use) lru = &x;
    chain_free(lru);
    lru->used = true; lru->key = key;
    b = lru;
  }
  b->last_use = ++c->chain_clock;
  return b;
}
// a bare call's state block / regrouped LSF records on stream `key` (see ChainBuf); false: no memory
static bool bare_state_tmp(pdmp3_hip_ctx* c, const void* key, float** out) {
  std::lock_guard<std::mutex> lock(c->chain_mu);
  ChainBuf* b = chain_entry(c, key);
  if (!b->state_tmp && hipMalloc((void**)&b->state_tmp, pdmp3_hip_state_bytes()) != hipSuccess) { (void)hipGetLastError(); b->state_tmp = nullptr; return false; }
  *out = b->state_tmp;
  return true;
}
static bool bare_pairs(pdmp3_hip_ctx* c, const void* key, int np, int16_t** sp, pdmp3_gc_side** sd) {
  std::lock_guard<std::mutex> lock(c->chain_mu);
  ChainBuf* b = chain_entry(c, key);
  if (b->pair_cap < np) {
    (void)hipFree(b->pair_sp); (void)hipFree(b->pair_sd);          // (synchronises: earlier launches are done with them)
    b->pair_sp = nullptr; b->pair_sd = nullptr; b->pair_cap = 0;
    const int cap = np < 64 ? 64 : np;
    if (hipMalloc((void**)&b->pair_sp, (size_t)cap * PDMP3_FRAME_SPECTRA_BYTES) != hipSuccess ||
        hipMalloc((void**)&b->pair_sd, (size_t)cap * PDMP3_FRAME_SIDE_BYTES) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipFree(b->pair_sp); b->pair_sp = nullptr; b->pair_sd = nullptr;
      return false;
    }
    b->pair_cap = cap;
  }
  *sp = b->pair_sp; *sd = b->pair_sd;
  return true;
}
static bool chain_get(pdmp3_hip_ctx* c, const void* key, hipStream_t s, int n_frames, ChainUse* use) {
  std::lock_guard<std::mutex> lock(c->chain_mu);
  ChainBuf* b = chain_entry(c, key);
  constexpr size_t kFloatsPerFrame = (size_t)2 * kGranFloats;
  if (b->cap < n_frames) {
    (void)hipFree(b->state); (void)hipFree(b->flag);       // (synchronises: earlier launches are done with the old ones)
    b->state = nullptr; b->flag = nullptr; b->cap = 0;
    const int cap = n_frames < 256 ? 256 : n_frames;
    const size_t flag_bytes = (size_t)cap * 4 * sizeof(unsigned);
    if (hipMalloc((void**)&b->state, (size_t)cap * kFloatsPerFrame * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&b->flag, flag_bytes) != hipSuccess ||
        hipMemsetAsync(b->flag, 0, flag_bytes, s) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipFree(b->state); (void)hipFree(b->flag);
      b->state = nullptr; b->flag = nullptr;
      return false;
    }
    b->cap = cap;
    b->epoch = 0;
  }
  if (b->epoch == 0xffffffffu) {                           // (never in practice: flags start over)
    if (hipMemsetAsync(b->flag, 0, (size_t)b->cap * 4 * sizeof(unsigned), s) != hipSuccess) return false;
    b->epoch = 0;
  }
  b->epoch++;
  use->state = b->state;
  use->flag = b->flag;
  use->epoch = b->epoch;
  return true;
}

static void chain_release(pdmp3_hip_ctx* c, const void* key) {     // (its launches are complete)
  std::lock_guard<std::mutex> lock(c->chain_mu);
  for (ChainBuf& x : c->chain)
    if (x.used && x.key == key) chain_free(&x);
}

// d_state_tmp: where the kernel leaves the new state before it is copied over d_state (chunk 0 and the channel-1
// pre-halo read the OLD state while the last chunk writes the new one).  Streams own one; a bare
// pdmp3_hip_decode_frames call takes the one kept for its HIP stream (bare_state_tmp), so that calls on different HIP streams never share it.
static int launch_decode(pdmp3_hip_ctx* c, const int16_t* d_spectra, const pdmp3_gc_side* d_side, int n_frames,
                         void* d_state, int16_t* d_pcm, float* d_stages, int chunk_frames, void* stream,
                         unsigned long long* d_prof = nullptr, float* d_state_tmp = nullptr, float* d_pcm_f32 = nullptr,
                         const void* owner = nullptr, bool leave_state_in_tmp = false, bool lsf = false,
                         int16_t* pair_sp = nullptr, pdmp3_gc_side* pair_sd = nullptr) {
  // owner: the stream object whose launches these are (they are ordered: one chain scratch for all of them);
  // leave_state_in_tmp: the caller swaps its two state buffers instead of having the new state copied back
  if (!c || n_frames < 0) return fail(PDMP3_HIP_EINVAL, "pdmp3_hip_decode_frames: bad argument", hipSuccess);
  if (n_frames == 0) return PDMP3_HIP_OK;
  if (!d_spectra || !d_side || !(d_pcm || d_pcm_f32))
    return fail(PDMP3_HIP_EINVAL, "pdmp3_hip_decode_frames: NULL buffer", hipSuccess);
  if (((uintptr_t)d_spectra | (uintptr_t)d_side | (uintptr_t)d_pcm | (uintptr_t)d_pcm_f32) & 15)
    return fail(PDMP3_HIP_EINVAL, "pdmp3_hip_decode_frames: buffers must be 16-byte aligned", hipSuccess);
  if (n_frames == 0) return PDMP3_HIP_OK;
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipSetDevice(c->device), "hipSetDevice");     // (a bare call may come from a thread whose current device is another one)
  // LSF (pdmp3_gc_side.lsf != 0): a frame is ONE granule.  The kernels keep their two-granule frames: the
  // launch's n frames are regrouped on the device into ceil(n / 2) record-frames whose granules are consecutive FRAMES
  // (k_lsf_pair: [0][ch] of frame 2 p and of frame 2 p + 1), decoded by the chunk kernel -- an odd last frame is a
  // record-frame of one granule (DecodeArgs::n_gran) -- and the PCM comes out in stream order, 576 sample-frames a frame.
  int16_t* d_pair_sp = nullptr;
  pdmp3_gc_side* d_pair_sd = nullptr;
  int n_gran = 0;
  if (lsf) {
    if (d_stages || d_prof) return fail(PDMP3_HIP_EINVAL, "pdmp3_hip_decode_lsf_frames: no stage dumps / profiles of LSF launches", hipSuccess);
    const int np = (n_frames + 1) / 2;
    // (a stream object brings its own buffers for the pairs, kept from batch to batch: pair_sp / pair_sd; a bare call's are the
    //  stream-ordered allocator's)
    if (pair_sp) { d_pair_sp = pair_sp; d_pair_sd = pair_sd; }
    else if (!bare_pairs(c, owner ? owner : (const void*)s, np, &d_pair_sp, &d_pair_sd)) return fail(PDMP3_HIP_ENOMEM, "hipMalloc LSF pairs", hipSuccess);
    hipLaunchKernelGGL(k_lsf_pair, dim3(np), dim3(256), 0, s, d_spectra, d_side, n_frames, d_pair_sp, d_pair_sd);
    n_gran = n_frames;
    d_spectra = d_pair_sp; d_side = d_pair_sd; n_frames = np;
    if (chunk_frames <= 1) chunk_frames = 0;             // (never the granule kernels: they hand on whole frames)
  }
  const int chunk_frames_arg = lsf ? 2 : chunk_frames;   // (0 = the engine's choice; an LSF launch: chunks)
  if (chunk_frames <= 0) chunk_frames = auto_chunk(n_frames, c->wave_slots);
  if (d_stages || chunk_frames > n_frames) chunk_frames = n_frames;
  DecodeArgs a;
  a.spectra = d_spectra;
  a.side = d_side;
  a.pcm = d_pcm;
  a.pcm_f32 = d_pcm_f32;
  a.state_in = (const float*)d_state;
  const void* chain_key = owner ? owner : (const void*)s;   // (a stream object, or the bare call's HIP stream)
  if (d_state && !d_state_tmp && !bare_state_tmp(c, chain_key, &d_state_tmp)) return fail(PDMP3_HIP_ENOMEM, "hipMalloc state", hipSuccess);
  a.state_out = d_state ? d_state_tmp : nullptr;
  a.stages = d_stages;
  a.n_frames = n_frames;
  a.chunk_frames = chunk_frames;
  a.prof = d_prof;
  a.chain_state = nullptr; a.chain_flag = nullptr; a.chain_epoch = 0; a.debug_flags = c->debug_flags; a.sf_hint = c->sf_hint;
  a.n_gran = n_gran;
  bool gran = false, ring = false;
  const bool plain = !d_stages && !d_prof;
  // the persistent granule kernel: launches of at least ring_min_frames frames (and chunk_frames = -3: always)
  const bool ring_prof = d_prof && chunk_frames_arg == -4;          // (development: the persistent kernel with per-turn stamps)
  if ((plain || ring_prof) && c->chain_mode != 0 && n_frames >= 16 &&
      (chunk_frames_arg == PDMP3_HIP_CHUNK_PERSISTENT || ring_prof || (chunk_frames_arg <= 1 && chunk_frames_arg >= 0 && c->ring_min_frames > 0 && n_frames >= c->ring_min_frames)))
    ring = true;
#if !defined(PDMP3_WITH_RING_KERNEL)
  if (ring && (chunk_frames_arg == PDMP3_HIP_CHUNK_PERSISTENT || ring_prof)) {
    return fail(PDMP3_HIP_EINVAL, "this build of libpdmp3_hip.so does not carry the persistent kernel (make -C pdmp3_amd/csrc EXTRA=-DPDMP3_WITH_RING_KERNEL)", hipSuccess);
  }
  ring = false;
#endif
  const bool gran_prof = d_prof && chunk_frames_arg == -2;          // (development: the granule kernel with per-wave stamps)
  if (!ring && (plain || gran_prof) && c->chain_mode != 0 && chunk_frames_arg <= 1 && n_frames <= c->gran_max_frames) {
    // one granule per wave (run_granule): tails and matrixing rows are handed from wave to wave, no halo.  Waits for
    // another workgroup are bounded (then: halo), so the launch finishes whatever part of it is resident; up to
    // gran_max_frames all of it is
    ChainUse u;
    if (chain_get(c, chain_key, s, n_frames, &u)) {
      a.chain_state = u.state; a.chain_flag = u.flag; a.chain_epoch = u.epoch;
      a.chunk_frames = 1;
      gran = true;
    }
  }
  GlobalTables T{c->d_pow43, c->d_linetab, c->d_win, c->d_frag, c->d_frag + 10 * 64, c->d_frag + 20 * 64, c->d_frag + 28 * 64, c->d_tab_image};
  const int nchunks = (n_frames + a.chunk_frames - 1) / a.chunk_frames;
#if defined(PDMP3_WITH_RING_KERNEL)
  if (ring) {
    // one workgroup per CU while a range keeps at least 8 frames (= one turn of the 16 waves)
    const int cus = c->cus > 0 ? c->cus : 256;
    int per = (n_frames + cus - 1) / cus;
    if (per < 8) per = 8;
    const int n_wgs = (n_frames + per - 1) / per;
    a.chunk_frames = 1;
    c->last_kind = PDMP3_HIP_LAUNCH_PERSISTENT;
    if (d_pcm_f32) hipLaunchKernelGGL((k_decode_p<true>), dim3(n_wgs), dim3(64 * 16), 0, s, a, T, per);
    else hipLaunchKernelGGL((k_decode_p<false>), dim3(n_wgs), dim3(64 * 16), 0, s, a, T, per);
  } else
#endif
  if (gran) {
    // (workgroups of 8 waves while that gives every CU at most one of them)
    const bool small = 2 * n_frames <= c->wave_slots_gran / 2 || c->gran_w8;
    const int W = small ? 8 : 16;
    c->last_kind = W;
    const int n_wgs = (2 * n_frames + W - 1) / W;
    if (small) {
      if (d_pcm_f32) hipLaunchKernelGGL((k_decode_g<true, 8>), dim3(n_wgs), dim3(64 * W), 0, s, a, T);
      else hipLaunchKernelGGL((k_decode_g<false, 8>), dim3(n_wgs), dim3(64 * W), 0, s, a, T);
    } else {
      if (d_pcm_f32) hipLaunchKernelGGL((k_decode_g<true, 16>), dim3(n_wgs), dim3(64 * W), 0, s, a, T);
      else hipLaunchKernelGGL((k_decode_g<false, 16>), dim3(n_wgs), dim3(64 * W), 0, s, a, T);
    }
  } else {
    c->last_kind = PDMP3_HIP_LAUNCH_CHUNKS;
    if (d_prof) hipLaunchKernelGGL(k_decode_prof, dim3(nchunks), dim3(64), 0, s, a, T);
    else if (d_stages) hipLaunchKernelGGL(k_decode<true>, dim3(nchunks), dim3(64), 0, s, a, T, nchunks, (unsigned*)nullptr, 0u);
    else if (lsf) {                                // every chunk of an LSF launch is k_decode_rare's
      if (d_pcm_f32) hipLaunchKernelGGL(k_decode_rare<true>, dim3(nchunks), dim3(64), 0, s, a, T, nchunks, 1, (const unsigned*)nullptr, 0u);
      else hipLaunchKernelGGL(k_decode_rare<false>, dim3(nchunks), dim3(64), 0, s, a, T, nchunks, 1, (const unsigned*)nullptr, 0u);
    } else {
      const unsigned ep = c->rare_epoch.fetch_add(1);
      unsigned* flag = c->d_rare_flags + (ep % kRareSlots);
      if (d_pcm_f32) {
        hipLaunchKernelGGL((k_decode<false, true>), dim3(nchunks), dim3(64), 0, s, a, T, nchunks, flag, ep);
        hipLaunchKernelGGL(k_decode_rare<true>, dim3(nchunks), dim3(64), 0, s, a, T, nchunks, 0, (const unsigned*)flag, ep);
      } else {
        // (development: PDMP3_HIP_DEBUG_LDS_PAD = bytes of dynamic LDS added to every workgroup of the chunk kernel, which
        //  lowers the number of waves a CU holds -- 24576: one wave per SIMD instead of two; tools/occupancy_scaling.py)
        static const int lds_pad = [] { const char* e = getenv("PDMP3_HIP_DEBUG_LDS_PAD"); return e ? atoi(e) : 0; }();
        hipLaunchKernelGGL(k_decode<false>, dim3(nchunks), dim3(64), (size_t)lds_pad, s, a, T, nchunks, flag, ep);
        hipLaunchKernelGGL(k_decode_rare<false>, dim3(nchunks), dim3(64), 0, s, a, T, nchunks, 0, (const unsigned*)flag, ep);
      }
    }
  }
  hipError_t e = hipGetLastError();
  const char* what = "launch k_decode";
  if (e == hipSuccess && d_state && !leave_state_in_tmp) {
    what = "state copy";
    e = hipMemcpyAsync(d_state, d_state_tmp, pdmp3_hip_state_bytes(), hipMemcpyDeviceToDevice, s);
  }
  if (e != hipSuccess) return fail(PDMP3_HIP_EDEVICE, what, e);
  return PDMP3_HIP_OK;
}

// The hand-over scratch the engine keeps for bare decode calls on `stream` (17 KB per frame of the largest granule-kernel
// launch seen there, until 32 other streams have been used or the engine is destroyed): given back now.  Blocks until the
// launches on that stream that use it are complete.
extern "C" int pdmp3_hip_release_stream_scratch(pdmp3_hip_ctx* ctx, void* stream) {
  if (!ctx) return fail(PDMP3_HIP_EINVAL, "pdmp3_hip_release_stream_scratch: NULL", hipSuccess);
  HIP_TRY(hipSetDevice(ctx->device), "hipSetDevice");
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream), "stream sync");
  chain_release(ctx, (const void*)stream);
  return PDMP3_HIP_OK;
}

extern "C" int pdmp3_hip_decode_frames(pdmp3_hip_ctx* ctx, const int16_t* d_spectra, const pdmp3_gc_side* d_side,
                                       int n_frames, void* d_state, int16_t* d_pcm, int chunk_frames, void* stream) {
  return launch_decode(ctx, d_spectra, d_side, n_frames, d_state, d_pcm, nullptr, chunk_frames, stream);
}

extern "C" int pdmp3_hip_decode_frames_f32(pdmp3_hip_ctx* ctx, const int16_t* d_spectra, const pdmp3_gc_side* d_side,
                                           int n_frames, void* d_state, float* d_pcm, int chunk_frames, void* stream) {
  if (!d_pcm) return fail(PDMP3_HIP_EINVAL, "pdmp3_hip_decode_frames_f32: d_pcm is NULL", hipSuccess);
  return launch_decode(ctx, d_spectra, d_side, n_frames, d_state, nullptr, nullptr, chunk_frames, stream, nullptr, nullptr, d_pcm);
}

// LSF frames (include/pdmp3_hip.h): n_frames one-granule frames of one channel count -> PCM in stream order
extern "C" int pdmp3_hip_decode_lsf_frames(pdmp3_hip_ctx* ctx, const int16_t* d_spectra, const pdmp3_gc_side* d_side,
                                           int n_frames, void* d_state, int16_t* d_pcm, void* stream) {
  return launch_decode(ctx, d_spectra, d_side, n_frames, d_state, d_pcm, nullptr, 0, stream, nullptr, nullptr, nullptr, nullptr, false, true);
}
extern "C" int pdmp3_hip_decode_lsf_frames_f32(pdmp3_hip_ctx* ctx, const int16_t* d_spectra, const pdmp3_gc_side* d_side,
                                               int n_frames, void* d_state, float* d_pcm, void* stream) {
  if (!d_pcm) return fail(PDMP3_HIP_EINVAL, "pdmp3_hip_decode_lsf_frames_f32: d_pcm is NULL", hipSuccess);
  return launch_decode(ctx, d_spectra, d_side, n_frames, d_state, nullptr, nullptr, 0, stream, nullptr, nullptr, d_pcm, nullptr, false, true);
}

extern "C" int pdmp3_hip_decode_frames_stages(pdmp3_hip_ctx* ctx, const int16_t* d_spectra, const pdmp3_gc_side* d_side,
                                              int n_frames, void* d_state, int16_t* d_pcm, float* d_stages, void* stream) {
  if (!d_stages) return fail(PDMP3_HIP_EINVAL, "pdmp3_hip_decode_frames_stages: d_stages is NULL", hipSuccess);
  return launch_decode(ctx, d_spectra, d_side, n_frames, d_state, d_pcm, d_stages, 0, stream);
}

// ---------------------------------------------------------------------------
// host-buffer streaming helper (pinned staging, hipMemcpyAsync both ways)
// ---------------------------------------------------------------------------
// One pdmp3_hip_stream = one decoder's carried state + up to kMaxSlots staging slots.  Each slot has its own
// HIP stream (H2D -> k_decode -> D2H), so slot w+1's upload overlaps slot w's kernel and download over the
// two PCIe directions; the kernels themselves are chained in submit order through `ev_state` because each one
// starts from the synthesis state its predecessor left (and they share the stream's d_state_tmp).
constexpr int kMaxSlots = 8;
struct StreamSlot {
  hipStream_t stream;
  hipEvent_t done;
  int16_t* h_spectra; pdmp3_gc_side* h_side; int16_t* h_pcm;     // pinned
  int16_t* d_spectra; pdmp3_gc_side* d_side; int16_t* d_pcm;
  int16_t* d_pair_sp; pdmp3_gc_side* d_pair_sd;                   // LSF launches: the regrouped records (allocated on first use, (max_frames + 1) / 2 frames)
  // bitstream-level input (allocated on first use)
  pdmp3_frame_bits* h_bits; uint8_t* h_res;                       // pinned
  pdmp3_frame_bits* d_bits; uint8_t* d_res; GcRaw* d_raw; uint32_t* d_outc; unsigned* d_mcnt;
  pdmp3_row_desc* h_desc; pdmp3_row_desc* d_desc; uint8_t* d_pool;   // compact bits input: pinned descriptors; the pool is h_res
  uint8_t* h_in; uint8_t* d_in;   // the blocks h_desc | h_bits | h_res and d_desc | d_bits | d_pool point into
  int busy;
  int direct;                     // the latest record submit ran on the pinned host buffers themselves (submit_records)
};
struct pdmp3_hip_stream {
  pdmp3_hip_ctx* ctx;
  int max_frames, n_slots;
  StreamSlot s[kMaxSlots];
  hipEvent_t ev_state;       // recorded after the latest kernel + state copy
  int have_state_ev;
  float* d_state;
  float* d_state_tmp;
  float* d_state_prev;       // d_state as it was before the latest submit of decoded records (pdmp3_hip_stream_rewind)
  uint16_t* d_sfstate;       // [2][256]: scalefactors / count1 carried from frame to frame (unpack_core.h), double-buffered
  int sf_cur;
  int have_bits;
  int f32;                   // PCM as float (pdmp3_hip_stream_set_f32): the slots' PCM buffers hold 9216 bytes per frame
  int lsf;                   // the records of the submits are LSF frames (pdmp3_hip_stream_set_lsf): pdmp3_hip_decode_lsf_frames' layout
};

extern "C" void pdmp3_hip_stream_destroy(pdmp3_hip_stream* hs) {
  if (!hs) return;
  (void)hipSetDevice(hs->ctx->device);
  for (int i = 0; i < hs->n_slots; ++i) {
    StreamSlot& t = hs->s[i];
    if (t.stream) { (void)hipStreamSynchronize(t.stream); (void)hipStreamDestroy(t.stream); }
    if (t.done) (void)hipEventDestroy(t.done);
    (void)hipHostFree(t.h_spectra); (void)hipHostFree(t.h_side); (void)hipHostFree(t.h_pcm);
    (void)hipFree(t.d_spectra); (void)hipFree(t.d_side); (void)hipFree(t.d_pcm); (void)hipFree(t.d_pair_sp); (void)hipFree(t.d_pair_sd);
    (void)hipHostFree(t.h_in);
    (void)hipFree(t.d_in); (void)hipFree(t.d_res); (void)hipFree(t.d_raw); (void)hipFree(t.d_outc); (void)hipFree(t.d_mcnt);
  }
  (void)hipFree(hs->d_sfstate);
  if (hs->ev_state) (void)hipEventDestroy(hs->ev_state);
  (void)hipFree(hs->d_state);
  chain_release(hs->ctx, hs);
  (void)hipFree(hs->d_state_tmp);
  (void)hipFree(hs->d_state_prev);
  free(hs);
}

extern "C" int pdmp3_hip_stream_create_slots(pdmp3_hip_ctx* ctx, int max_frames, int n_slots, pdmp3_hip_stream** out) {
  if (!ctx || !out || max_frames < 1 || n_slots < 1 || n_slots > kMaxSlots)
    return fail(PDMP3_HIP_EINVAL, "pdmp3_hip_stream_create: bad argument", hipSuccess);
  *out = nullptr;
  HIP_TRY(hipSetDevice(ctx->device), "hipSetDevice");
  pdmp3_hip_stream* hs = (pdmp3_hip_stream*)calloc(1, sizeof *hs);
  if (!hs) return fail(PDMP3_HIP_ENOMEM, "calloc", hipSuccess);
  hs->ctx = ctx;
  hs->max_frames = max_frames;
  hs->n_slots = n_slots;
  const size_t n = (size_t)max_frames;
#define HS_TRY(call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) { pdmp3_hip_stream_destroy(hs); return fail(PDMP3_HIP_EDEVICE, what, e_); } } while (0)
  for (int i = 0; i < n_slots; ++i) {
    StreamSlot& t = hs->s[i];
    HS_TRY(hipStreamCreateWithFlags(&t.stream, hipStreamNonBlocking), "hipStreamCreate");
    HS_TRY(hipEventCreateWithFlags(&t.done, hipEventDisableTiming), "hipEventCreate");
    HS_TRY(hipHostMalloc((void**)&t.h_spectra, n * PDMP3_FRAME_SPECTRA_BYTES, hipHostMallocDefault), "hipHostMalloc spectra");
    HS_TRY(hipHostMalloc((void**)&t.h_side, n * PDMP3_FRAME_SIDE_BYTES, hipHostMallocDefault), "hipHostMalloc side");
    HS_TRY(hipHostMalloc((void**)&t.h_pcm, n * PDMP3_FRAME_PCM_BYTES, hipHostMallocDefault), "hipHostMalloc pcm");
    HS_TRY(hipMalloc((void**)&t.d_spectra, n * PDMP3_FRAME_SPECTRA_BYTES), "hipMalloc spectra");
    HS_TRY(hipMalloc((void**)&t.d_side, n * PDMP3_FRAME_SIDE_BYTES), "hipMalloc side");
    HS_TRY(hipMalloc((void**)&t.d_pcm, n * PDMP3_FRAME_PCM_BYTES), "hipMalloc pcm");
  }
  HS_TRY(hipEventCreateWithFlags(&hs->ev_state, hipEventDisableTiming), "hipEventCreate");
  HS_TRY(hipMalloc((void**)&hs->d_state, pdmp3_hip_state_bytes()), "hipMalloc state");
  HS_TRY(hipMalloc((void**)&hs->d_state_tmp, pdmp3_hip_state_bytes()), "hipMalloc state");
  HS_TRY(hipMalloc((void**)&hs->d_state_prev, pdmp3_hip_state_bytes()), "hipMalloc state");
  HS_TRY(hipMemsetAsync(hs->d_state, 0, pdmp3_hip_state_bytes(), hs->s[0].stream), "memset state");
  HS_TRY(hipStreamSynchronize(hs->s[0].stream), "sync");
#undef HS_TRY
  *out = hs;
  return PDMP3_HIP_OK;
}

extern "C" int pdmp3_hip_stream_create(pdmp3_hip_ctx* ctx, int max_frames, pdmp3_hip_stream** out) {
  return pdmp3_hip_stream_create_slots(ctx, max_frames, 1, out);
}

static int drain_slots(pdmp3_hip_stream* hs) {
  for (int i = 0; i < hs->n_slots; ++i) {
    HIP_TRY(hipStreamSynchronize(hs->s[i].stream), "stream sync");
    hs->s[i].busy = 0;
  }
  return PDMP3_HIP_OK;
}

extern "C" int pdmp3_hip_stream_reset(pdmp3_hip_stream* hs) {
  if (!hs) return fail(PDMP3_HIP_EINVAL, "pdmp3_hip_stream_reset: NULL", hipSuccess);
  HIP_TRY(hipSetDevice(hs->ctx->device), "hipSetDevice");
  int rc = drain_slots(hs);
  if (rc != PDMP3_HIP_OK) return rc;
  hs->have_state_ev = 0;
  HIP_TRY(hipMemsetAsync(hs->d_state, 0, pdmp3_hip_state_bytes(), hs->s[0].stream), "memset state");
  if (hs->d_sfstate) HIP_TRY(hipMemsetAsync(hs->d_sfstate, 0, 2 * 256 * sizeof(uint16_t), hs->s[0].stream), "memset sfstate");
  HIP_TRY(hipStreamSynchronize(hs->s[0].stream), "sync");
  return PDMP3_HIP_OK;
}

#define SLOT_OK(hs, i) ((hs) && (i) >= 0 && (i) < (hs)->n_slots)
extern "C" int pdmp3_hip_stream_slots(const pdmp3_hip_stream* hs) { return hs ? hs->n_slots : 0; }
extern "C" int pdmp3_hip_stream_capacity(const pdmp3_hip_stream* hs) { return hs ? hs->max_frames : 0; }
extern "C" int16_t* pdmp3_hip_stream_slot_spectra(pdmp3_hip_stream* hs, int slot) { return SLOT_OK(hs, slot) ? hs->s[slot].h_spectra : nullptr; }
extern "C" pdmp3_gc_side* pdmp3_hip_stream_slot_side(pdmp3_hip_stream* hs, int slot) { return SLOT_OK(hs, slot) ? hs->s[slot].h_side : nullptr; }
extern "C" const int16_t* pdmp3_hip_stream_slot_pcm(pdmp3_hip_stream* hs, int slot) { return SLOT_OK(hs, slot) ? hs->s[slot].h_pcm : nullptr; }
extern "C" int16_t* pdmp3_hip_stream_spectra(pdmp3_hip_stream* hs) { return pdmp3_hip_stream_slot_spectra(hs, 0); }
extern "C" pdmp3_gc_side* pdmp3_hip_stream_side(pdmp3_hip_stream* hs) { return pdmp3_hip_stream_slot_side(hs, 0); }
extern "C" const int16_t* pdmp3_hip_stream_pcm(pdmp3_hip_stream* hs) { return pdmp3_hip_stream_slot_pcm(hs, 0); }

// PCM of a batch to its destination: into the slot's pinned buffer, or -- when the caller hands over pinned host
// memory (pdmp3_hip_host_alloc) or DEVICE memory of its own -- straight to where it is wanted, `row` bytes per frame (4608; 2304 = mono frames
// packed densely out of their 4608-byte slots).
static int download_pcm(StreamSlot& t, size_t n, void* host_dst, int row, bool lsf = false) {
  if (lsf && host_dst) {
    // LSF frames (pdmp3_hip_decode_lsf_frames): stereo frames lie back to back, 2304 bytes each; mono frames in pairs in the
    // first half of a 4608-byte place
    if (row == PDMP3_FRAME_PCM_BYTES / 2) {
      HIP_TRY(hipMemcpyAsync(host_dst, t.d_pcm, n * (size_t)row, hipMemcpyDefault, t.stream), "pcm (direct, LSF)");
    } else {
      if (n / 2) HIP_TRY(hipMemcpy2DAsync(host_dst, 2304, t.d_pcm, PDMP3_FRAME_PCM_BYTES, 2304, n / 2, hipMemcpyDefault, t.stream), "pcm (direct, LSF mono)");
      if (n & 1) HIP_TRY(hipMemcpyAsync((char*)host_dst + (n / 2) * 2304, (const char*)t.d_pcm + (n / 2) * PDMP3_FRAME_PCM_BYTES, 1152, hipMemcpyDefault, t.stream), "pcm (direct, LSF mono tail)");
    }
    return PDMP3_HIP_OK;
  }
  if (!host_dst) {
    HIP_TRY(hipMemcpyAsync(t.h_pcm, t.d_pcm, n * PDMP3_FRAME_PCM_BYTES, hipMemcpyDeviceToHost, t.stream), "D2H pcm");
  } else if (row == PDMP3_FRAME_PCM_BYTES) {
    HIP_TRY(hipMemcpyAsync(host_dst, t.d_pcm, n * PDMP3_FRAME_PCM_BYTES, hipMemcpyDefault, t.stream), "pcm (direct)");
  } else {
    HIP_TRY(hipMemcpy2DAsync(host_dst, (size_t)row, t.d_pcm, PDMP3_FRAME_PCM_BYTES, (size_t)row, n, hipMemcpyDefault, t.stream),
            "pcm (direct, packed)");
  }
  return PDMP3_HIP_OK;
}

extern "C" int pdmp3_hip_host_alloc(size_t bytes, void** out) {
  if (!out) return fail(PDMP3_HIP_EINVAL, "pdmp3_hip_host_alloc: out is NULL", hipSuccess);
  *out = nullptr;
  HIP_TRY(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault), "hipHostMalloc");
  return PDMP3_HIP_OK;
}
extern "C" void pdmp3_hip_host_free(void* p) { if (p) (void)hipHostFree(p); }
extern "C" int pdmp3_hip_host_is_pinned(const void* p, size_t bytes) {
  if (!p) return 0;
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return 0; }
  if (a.type != hipMemoryTypeHost && a.type != hipMemoryTypeDevice) return 0;
  if (bytes > 1) {
    hipPointerAttribute_t e;
    if (hipPointerGetAttributes(&e, (const char*)p + bytes - 1) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (e.type != a.type) return 0;
  }
  return a.type == hipMemoryTypeHost ? 1 : 2;
}

extern "C" int pdmp3_hip_copy_to_dest(void* dst, const void* src_host, size_t bytes) {
  if (!bytes) return PDMP3_HIP_OK;
  if (!dst || !src_host) return fail(PDMP3_HIP_EINVAL, "pdmp3_hip_copy_to_dest: NULL", hipSuccess);
  HIP_TRY(hipMemcpy(dst, src_host, bytes, hipMemcpyDefault), "copy to destination");
  return PDMP3_HIP_OK;
}

// PCM of the record-level submits (pdmp3_hip_stream_submit / _decode) as float from now on (on != 0) or int16 again.
// Call with nothing in flight; the slots' PCM buffers are re-allocated.  The accessors return the same pointers'
// new values, to be read as float: frame f at floats [f*2304, f*2304+2304).
extern "C" int pdmp3_hip_stream_set_f32(pdmp3_hip_stream* hs, int on) {
  if (!hs) return fail(PDMP3_HIP_EINVAL, "pdmp3_hip_stream_set_f32: NULL", hipSuccess);
  on = on ? 1 : 0;
  if (hs->f32 == on) return PDMP3_HIP_OK;
  HIP_TRY(hipSetDevice(hs->ctx->device), "hipSetDevice");
  int rc = drain_slots(hs);
  if (rc != PDMP3_HIP_OK) return rc;
  const size_t bytes = (size_t)hs->max_frames * (on ? PDMP3_FRAME_PCM_F32_BYTES : PDMP3_FRAME_PCM_BYTES);
  for (int i = 0; i < hs->n_slots; ++i) {
    StreamSlot& t = hs->s[i];
    (void)hipHostFree(t.h_pcm); t.h_pcm = nullptr;
    (void)hipFree(t.d_pcm); t.d_pcm = nullptr;
    HIP_TRY(hipHostMalloc((void**)&t.h_pcm, bytes, hipHostMallocDefault), "hipHostMalloc pcm");
    HIP_TRY(hipMalloc((void**)&t.d_pcm, bytes), "hipMalloc pcm");
  }
  hs->f32 = on;
  return PDMP3_HIP_OK;
}

static int submit_records(pdmp3_hip_stream* hs, int slot, int n_frames, void* host_dst, int row);
extern "C" int pdmp3_hip_stream_submit(pdmp3_hip_stream* hs, int slot, int n_frames) {
  return submit_records(hs, slot, n_frames, nullptr, PDMP3_FRAME_PCM_BYTES);
}
extern "C" int pdmp3_hip_stream_submit_to(pdmp3_hip_stream* hs, int slot, int n_frames, void* pinned_dst, int row_bytes) {
  if (pinned_dst && hs && hs->lsf) {
    if (row_bytes != PDMP3_FRAME_PCM_BYTES / 2 && row_bytes != PDMP3_FRAME_PCM_BYTES / 4)
      return fail(PDMP3_HIP_EINVAL, "pdmp3_hip_stream_submit_to: row_bytes of LSF frames must be 2304 (stereo) or 1152 (mono)", hipSuccess);
  } else
  if (pinned_dst && row_bytes != PDMP3_FRAME_PCM_BYTES && row_bytes != PDMP3_FRAME_PCM_BYTES / 2)
    return fail(PDMP3_HIP_EINVAL, "pdmp3_hip_stream_submit_to: row_bytes must be 4608 or 2304", hipSuccess);
  return submit_records(hs, slot, n_frames, pinned_dst, row_bytes);
}
// the slot's buffers for an LSF launch's regrouped records (launch_decode pair_sp / pair_sd): allocated once, on the first LSF submit
static int slot_pairs(pdmp3_hip_stream* hs, StreamSlot& t) {
  if (!hs->lsf || t.d_pair_sp) return PDMP3_HIP_OK;
  const size_t np = ((size_t)hs->max_frames + 1) / 2;
  HIP_TRY(hipMalloc((void**)&t.d_pair_sp, np * PDMP3_FRAME_SPECTRA_BYTES), "hipMalloc LSF pairs");
  if (hipMalloc((void**)&t.d_pair_sd, np * PDMP3_FRAME_SIDE_BYTES) != hipSuccess) { (void)hipFree(t.d_pair_sp); t.d_pair_sp = nullptr; return fail(PDMP3_HIP_ENOMEM, "hipMalloc LSF pairs", hipGetLastError()); }
  return PDMP3_HIP_OK;
}
static int submit_records(pdmp3_hip_stream* hs, int slot, int n_frames, void* host_dst, int row) {
  if (!SLOT_OK(hs, slot) || n_frames < 0 || n_frames > hs->max_frames)
    return fail(PDMP3_HIP_EINVAL, "pdmp3_hip_stream_submit: bad argument", hipSuccess);
  StreamSlot& t = hs->s[slot];
  if (t.busy) return fail(PDMP3_HIP_EINVAL, "pdmp3_hip_stream_submit: slot still in flight (wait for it first)", hipSuccess);
  if (n_frames == 0) return PDMP3_HIP_OK;
  HIP_TRY(hipSetDevice(hs->ctx->device), "hipSetDevice");
  { const int rcp = slot_pairs(hs, t); if (rcp != PDMP3_HIP_OK) return rcp; }
  const size_t n = (size_t)n_frames;
  if (hs->f32 && host_dst) return fail(PDMP3_HIP_EINVAL, "pdmp3_hip_stream_submit_to: not with float PCM", hipSuccess);
  // Small batches (the streaming API's read-ahead: a handful of frames per call): no copies at all.  The kernel reads the
  // records from the slot's PINNED host buffers and writes the PCM into pinned host memory itself -- a few KB each way,
  // one PCIe round trip under the waves' first phase instead of two copy commands in front of the kernel and one behind
  // it -- and the three state buffers rotate instead of being copied (previous <- current <- next).  Per batch: one
  // launch, two event records, one wait.  (PDMP3_HIP_DIRECT_MAX: largest such batch in frames, 0 = never.)
  t.direct = 0;
  // (the kernel itself stores into host_dst on this path: only where it certainly can -- pinned host memory, or memory of
  //  THIS device; registered / managed memory and another GPU's memory take the copy path, whose hipMemcpyAsync sorts it out)
  bool dst_ok = true;
  if (host_dst) {
    hipPointerAttribute_t pa;
    if (hipPointerGetAttributes(&pa, host_dst) != hipSuccess) { (void)hipGetLastError(); dst_ok = false; }
    else dst_ok = (pa.type == hipMemoryTypeHost && !pa.isManaged) || (pa.type == hipMemoryTypeDevice && pa.device == hs->ctx->device);
  }
  if (n_frames <= hs->ctx->direct_max_frames && dst_ok && (!host_dst || row == PDMP3_FRAME_PCM_BYTES) && !(hs->lsf && host_dst)) {
    // (a stream object with ONE slot -- the streaming API's -- has one HIP stream: its batches are in order anyway, and
    //  its wait is for that stream: no events at all, each of which is a call here and a packet of its own on the queue)
    const bool lone = hs->n_slots == 1;
    if (!lone && hs->have_state_ev) HIP_TRY(hipStreamWaitEvent(t.stream, hs->ev_state, 0), "wait for the previous batch's state");
    void* dst = host_dst ? host_dst : (void*)t.h_pcm;
    int rc = hs->f32
        ? launch_decode(hs->ctx, t.h_spectra, t.h_side, n_frames, hs->d_state, nullptr, nullptr, 0, t.stream, nullptr, hs->d_state_tmp, (float*)dst, hs, true, hs->lsf != 0, t.d_pair_sp, t.d_pair_sd)
        : launch_decode(hs->ctx, t.h_spectra, t.h_side, n_frames, hs->d_state, (int16_t*)dst, nullptr, 0, t.stream, nullptr, hs->d_state_tmp, nullptr, hs, true, hs->lsf != 0, t.d_pair_sp, t.d_pair_sd);
    if (rc != PDMP3_HIP_OK) return rc;
    float* const was_prev = hs->d_state_prev;
    hs->d_state_prev = hs->d_state;            // (what pdmp3_hip_stream_rewind goes back to)
    hs->d_state = hs->d_state_tmp;             // the kernel left the new state here
    hs->d_state_tmp = was_prev;
    if (!lone) {
      HIP_TRY(hipEventRecord(hs->ev_state, t.stream), "record state event");
      hs->have_state_ev = 1;
      HIP_TRY(hipEventRecord(t.done, t.stream), "record done event");
    }
    t.busy = 1;
    t.direct = lone ? 2 : 1;                   // 2: pdmp3_hip_stream_wait synchronises the stream
    return PDMP3_HIP_OK;
  }
  HIP_TRY(hipMemcpyAsync(t.d_spectra, t.h_spectra, n * PDMP3_FRAME_SPECTRA_BYTES, hipMemcpyHostToDevice, t.stream), "H2D spectra");
  HIP_TRY(hipMemcpyAsync(t.d_side, t.h_side, n * PDMP3_FRAME_SIDE_BYTES, hipMemcpyHostToDevice, t.stream), "H2D side");
  if (hs->have_state_ev) HIP_TRY(hipStreamWaitEvent(t.stream, hs->ev_state, 0), "wait for the previous batch's state");
  HIP_TRY(hipMemcpyAsync(hs->d_state_prev, hs->d_state, pdmp3_hip_state_bytes(), hipMemcpyDeviceToDevice, t.stream), "keep the state");
  int rc = hs->f32
      ? launch_decode(hs->ctx, t.d_spectra, t.d_side, n_frames, hs->d_state, nullptr, nullptr, 0, t.stream, nullptr, hs->d_state_tmp, (float*)t.d_pcm, hs, false, hs->lsf != 0, t.d_pair_sp, t.d_pair_sd)
      : launch_decode(hs->ctx, t.d_spectra, t.d_side, n_frames, hs->d_state, t.d_pcm, nullptr, 0, t.stream, nullptr, hs->d_state_tmp, nullptr, hs, false, hs->lsf != 0, t.d_pair_sp, t.d_pair_sd);
  if (rc != PDMP3_HIP_OK) return rc;
  HIP_TRY(hipEventRecord(hs->ev_state, t.stream), "record state event");
  hs->have_state_ev = 1;
  if (hs->f32) HIP_TRY(hipMemcpyAsync(t.h_pcm, t.d_pcm, n * PDMP3_FRAME_PCM_F32_BYTES, hipMemcpyDeviceToHost, t.stream), "D2H pcm");
  else rc = download_pcm(t, n, host_dst, row, hs->lsf != 0);
  if (rc != PDMP3_HIP_OK) return rc;
  HIP_TRY(hipEventRecord(t.done, t.stream), "record done event");
  t.busy = 1;
  return PDMP3_HIP_OK;
}

// The records of this stream object's following submits are LSF frames (on != 0: decoded like pdmp3_hip_decode_lsf_frames,
// the PCM in its layout) or MPEG-1 frames again.  A setting of the host's for the next submit; the synthesis state is
// the same block either way.
extern "C" int pdmp3_hip_stream_set_lsf(pdmp3_hip_stream* hs, int on) {
  if (!hs) return fail(PDMP3_HIP_EINVAL, "pdmp3_hip_stream_set_lsf: NULL", hipSuccess);
  hs->lsf = on != 0;
  return PDMP3_HIP_OK;
}

extern "C" int pdmp3_hip_stream_wait(pdmp3_hip_stream* hs, int slot) {
  if (!SLOT_OK(hs, slot)) return fail(PDMP3_HIP_EINVAL, "pdmp3_hip_stream_wait: bad argument", hipSuccess);
  StreamSlot& t = hs->s[slot];
  if (!t.busy) return PDMP3_HIP_OK;
  HIP_TRY(hipSetDevice(hs->ctx->device), "hipSetDevice");
  if (t.direct == 2) HIP_TRY(hipStreamSynchronize(t.stream), "stream sync");
  else HIP_TRY(hipEventSynchronize(t.done), "event sync");
  t.busy = 0;
  return PDMP3_HIP_OK;
}

// 1: pdmp3_hip_stream_wait(hs, slot) would return at once (nothing submitted, or the GPU is through with it); 0: not yet.
// For the thread that would call the wait (the whole-stream decoder asks how much the GPU still has to do before it
// decides how many frames the next window gets).
extern "C" int pdmp3_hip_stream_done(pdmp3_hip_stream* hs, int slot) {
  if (!SLOT_OK(hs, slot)) return fail(PDMP3_HIP_EINVAL, "pdmp3_hip_stream_done: bad argument", hipSuccess);
  StreamSlot& t = hs->s[slot];
  if (!t.busy) return 1;
  const hipError_t e = t.direct == 2 ? hipStreamQuery(t.stream) : hipEventQuery(t.done);
  if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
  return 1;                                     // (done, or an error the wait will report)
}

// Undo the slot's latest pdmp3_hip_stream_submit beyond its first keep_frames frames: the carried synthesis state
// becomes what it was after frame keep_frames - 1 of that batch (the state before the batch, then the kept frames
// again -- their records are still in the slot's device buffers).  Blocks until done.
extern "C" int pdmp3_hip_stream_rewind(pdmp3_hip_stream* hs, int slot, int keep_frames) {
  if (!SLOT_OK(hs, slot) || keep_frames < 0 || keep_frames > hs->max_frames)
    return fail(PDMP3_HIP_EINVAL, "pdmp3_hip_stream_rewind: bad argument", hipSuccess);
  HIP_TRY(hipSetDevice(hs->ctx->device), "hipSetDevice");
  StreamSlot& t = hs->s[slot];
  HIP_TRY(hipStreamSynchronize(t.stream), "stream sync");
  t.busy = 0;
  HIP_TRY(hipMemcpyAsync(hs->d_state, hs->d_state_prev, pdmp3_hip_state_bytes(), hipMemcpyDeviceToDevice, t.stream), "restore the state");
  if (keep_frames) {
    // (the records are where the submit took them from: the pinned host buffers if it ran on those)
    const int16_t* sp = t.direct ? t.h_spectra : t.d_spectra;
    const pdmp3_gc_side* sd = t.direct ? t.h_side : t.d_side;
    const int rc = hs->f32
        ? launch_decode(hs->ctx, sp, sd, keep_frames, hs->d_state, nullptr, nullptr, 0, t.stream, nullptr, hs->d_state_tmp, (float*)t.d_pcm, hs, false, hs->lsf != 0, t.d_pair_sp, t.d_pair_sd)
        : launch_decode(hs->ctx, sp, sd, keep_frames, hs->d_state, t.d_pcm, nullptr, 0, t.stream, nullptr, hs->d_state_tmp, nullptr, hs, false, hs->lsf != 0, t.d_pair_sp, t.d_pair_sd);
    if (rc != PDMP3_HIP_OK) return rc;
  }
  HIP_TRY(hipEventRecord(hs->ev_state, t.stream), "record state event");
  hs->have_state_ev = 1;
  HIP_TRY(hipStreamSynchronize(t.stream), "stream sync");
  return PDMP3_HIP_OK;
}

// ---- bitstream-level input ------------------------------------------------
static int ensure_bits(pdmp3_hip_stream* hs) {
  if (hs->have_bits) return PDMP3_HIP_OK;
  HIP_TRY(hipSetDevice(hs->ctx->device), "hipSetDevice");
  const size_t n = (size_t)hs->max_frames;
  for (int i = 0; i < hs->n_slots; ++i) {
    StreamSlot& t = hs->s[i];
    // descriptors | side info | rows or pool: ONE pinned block and one device block with the same layout, so that the
    // compact input of a window goes up in one copy (a hipMemcpyAsync call costs the submitting thread 20-40 us here)
    const size_t desc_bytes = n * sizeof(pdmp3_row_desc), bits_bytes = n * sizeof(pdmp3_frame_bits);
    const size_t pool_cap = n * PDMP3_RESERVOIR_BYTES + PDMP3_POOL_SLACK_BYTES + 16;
    HIP_TRY(hipHostMalloc((void**)&t.h_in, desc_bytes + bits_bytes + pool_cap, hipHostMallocDefault), "hipHostMalloc window input");
    HIP_TRY(hipMalloc((void**)&t.d_in, desc_bytes + bits_bytes + pool_cap), "hipMalloc window input");
    t.h_desc = reinterpret_cast<pdmp3_row_desc*>(t.h_in);
    t.h_bits = reinterpret_cast<pdmp3_frame_bits*>(t.h_in + desc_bytes);
    t.h_res = t.h_in + desc_bytes + bits_bytes;
    t.d_desc = reinterpret_cast<pdmp3_row_desc*>(t.d_in);
    t.d_bits = reinterpret_cast<pdmp3_frame_bits*>(t.d_in + desc_bytes);
    t.d_pool = t.d_in + desc_bytes + bits_bytes;
    HIP_TRY(hipMalloc((void**)&t.d_res, n * PDMP3_RESERVOIR_BYTES + 16), "hipMalloc reservoir");
    HIP_TRY(hipMalloc((void**)&t.d_raw, n * 4 * sizeof(GcRaw)), "hipMalloc raw");
    // rows of outcomes: one per block of kMergeBlk frames, then one per super-block; the super-blocks' counters (zero between launches)
    const size_t mblk = (n + kMergeBlk - 1) / kMergeBlk, msup = (mblk + kMergeSuper - 1) / kMergeSuper;
    HIP_TRY(hipMalloc((void**)&t.d_outc, (mblk + msup) * kMergeLanes * sizeof(uint32_t)), "hipMalloc merge outcomes");
    HIP_TRY(hipMalloc((void**)&t.d_mcnt, (msup + 1) * sizeof(unsigned)), "hipMalloc merge counters");
    HIP_TRY(hipMemset(t.d_mcnt, 0, (msup + 1) * sizeof(unsigned)), "memset merge counters");
  }
  HIP_TRY(hipMalloc((void**)&hs->d_sfstate, 2 * 256 * sizeof(uint16_t)), "hipMalloc sfstate");
  HIP_TRY(hipMemset(hs->d_sfstate, 0, 2 * 256 * sizeof(uint16_t)), "memset sfstate");
  // hipMemset returns before the device has run it, and the slots' streams are non-blocking: they do not wait for the null
  // stream.  Without this wait the zeroing of the scalefactor state could land AFTER the first window's k_merge_apply had
  // written it -- the second window of a fresh decoder then started from zeros (round 6: one whole-stream decode in ~2000
  // with fresh decoders differed by 1-3 LSB in frames 17-18; tools/ubench/malloc_async_probe.cpp mode M shows the mechanism)
  HIP_TRY(hipDeviceSynchronize(), "device sync after the memsets");
  hs->have_bits = 1;
  return PDMP3_HIP_OK;
}

extern "C" pdmp3_frame_bits* pdmp3_hip_stream_slot_bits(pdmp3_hip_stream* hs, int slot) {
  if (!SLOT_OK(hs, slot) || ensure_bits(hs) != PDMP3_HIP_OK) return nullptr;
  return hs->s[slot].h_bits;
}
extern "C" uint8_t* pdmp3_hip_stream_slot_reservoir(pdmp3_hip_stream* hs, int slot) {
  if (!SLOT_OK(hs, slot) || ensure_bits(hs) != PDMP3_HIP_OK) return nullptr;
  return hs->s[slot].h_res;
}

extern "C" pdmp3_row_desc* pdmp3_hip_stream_slot_rowdesc(pdmp3_hip_stream* hs, int slot) {
  if (!SLOT_OK(hs, slot) || ensure_bits(hs) != PDMP3_HIP_OK) return nullptr;
  return hs->s[slot].h_desc;
}
extern "C" uint8_t* pdmp3_hip_stream_slot_pool(pdmp3_hip_stream* hs, int slot) { return pdmp3_hip_stream_slot_reservoir(hs, slot); }
extern "C" size_t pdmp3_hip_stream_pool_bytes(const pdmp3_hip_stream* hs) { return hs ? (size_t)hs->max_frames * PDMP3_RESERVOIR_BYTES + PDMP3_POOL_SLACK_BYTES : 0; }

static int submit_bits(pdmp3_hip_stream* hs, int slot, int n_frames, void* host_dst, int row, size_t pool_bytes = 0);
extern "C" int pdmp3_hip_stream_submit_pool_to(pdmp3_hip_stream* hs, int slot, int n_frames, size_t pool_bytes, void* pinned_dst, int row_bytes) {
  if (pinned_dst && row_bytes != PDMP3_FRAME_PCM_BYTES && row_bytes != PDMP3_FRAME_PCM_BYTES / 2)
    return fail(PDMP3_HIP_EINVAL, "pdmp3_hip_stream_submit_pool_to: row_bytes must be 4608 or 2304", hipSuccess);
  if (!pool_bytes || !hs || pool_bytes > pdmp3_hip_stream_pool_bytes(hs))
    return fail(PDMP3_HIP_EINVAL, "pdmp3_hip_stream_submit_pool_to: bad pool size", hipSuccess);
  return submit_bits(hs, slot, n_frames, pinned_dst, row_bytes, pool_bytes);
}
extern "C" int pdmp3_hip_stream_submit_bits(pdmp3_hip_stream* hs, int slot, int n_frames) {
  return submit_bits(hs, slot, n_frames, nullptr, PDMP3_FRAME_PCM_BYTES);
}
extern "C" int pdmp3_hip_stream_submit_bits_to(pdmp3_hip_stream* hs, int slot, int n_frames, void* pinned_dst, int row_bytes) {
  if (pinned_dst && row_bytes != PDMP3_FRAME_PCM_BYTES && row_bytes != PDMP3_FRAME_PCM_BYTES / 2)
    return fail(PDMP3_HIP_EINVAL, "pdmp3_hip_stream_submit_bits_to: row_bytes must be 4608 or 2304", hipSuccess);
  return submit_bits(hs, slot, n_frames, pinned_dst, row_bytes);
}
static int submit_bits(pdmp3_hip_stream* hs, int slot, int n_frames, void* host_dst, int row, size_t pool_bytes) {
  if (!SLOT_OK(hs, slot) || n_frames < 0 || n_frames > hs->max_frames)
    return fail(PDMP3_HIP_EINVAL, "pdmp3_hip_stream_submit_bits: bad argument", hipSuccess);
  StreamSlot& t = hs->s[slot];
  if (t.busy) return fail(PDMP3_HIP_EINVAL, "pdmp3_hip_stream_submit_bits: slot still in flight (wait for it first)", hipSuccess);
  if (n_frames == 0) return PDMP3_HIP_OK;
  int rc = ensure_bits(hs);
  if (rc != PDMP3_HIP_OK) return rc;
  HIP_TRY(hipSetDevice(hs->ctx->device), "hipSetDevice");
  const size_t n = (size_t)n_frames;
  t.direct = 0;               // (the records of this submit are the device's: a rewind replays d_spectra, never h_spectra)
  if (pool_bytes) {           // compact input: descriptors, side info and pool up in one copy, rows rebuilt on the device
    // (Tried: k_rows reading descriptors, side info and pool straight from the pinned host block, no copy at all -- the
    //  kernel then runs at PCIe speed and the pipeline, which is bound by the kernels of a window, lost 20 %.)
    const size_t head = (size_t)hs->max_frames * (sizeof(pdmp3_row_desc) + sizeof(pdmp3_frame_bits));
    HIP_TRY(hipMemcpyAsync(t.d_in, t.h_in, head + pool_bytes, hipMemcpyHostToDevice, t.stream), "H2D window input");
    hipLaunchKernelGGL(k_rows, dim3((unsigned)((n_frames + kRowsWaves - 1) / kRowsWaves)), dim3(64 * kRowsWaves), 0, t.stream, t.d_desc, t.d_pool, t.d_res, n_frames);
    HIP_TRY(hipGetLastError(), "launch k_rows");
  } else {
    HIP_TRY(hipMemcpyAsync(t.d_bits, t.h_bits, n * sizeof(pdmp3_frame_bits), hipMemcpyHostToDevice, t.stream), "H2D bits");
    HIP_TRY(hipMemcpyAsync(t.d_res, t.h_res, n * PDMP3_RESERVOIR_BYTES, hipMemcpyHostToDevice, t.stream), "H2D reservoir");
  }
  {
    int blocks = (n_frames + kUnpackRows - 1) / kUnpackRows;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_unpack, dim3(blocks), dim3(kUnpackThreads), 0, t.stream, hs->ctx->d_unpack, t.d_bits, t.d_res,
                       n_frames, t.d_spectra, t.d_raw, hs->ctx->unpack_n16, hs->ctx->d_uprof);
    HIP_TRY(hipGetLastError(), "launch k_unpack");
    if (hs->ctx->d_uprof) {                            // development only: serialises, prints one line per launch
      static std::vector<unsigned long long> hp(2048 * 8);
      HIP_TRY(hipStreamSynchronize(t.stream), "unpack prof sync");
      HIP_TRY(hipMemcpy(hp.data(), hs->ctx->d_uprof, (size_t)blocks * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost), "unpack prof D2H");
      double d[5] = {0, 0, 0, 0, 0}, trips = 0, tmax = 0;
      for (int b = 0; b < blocks; b++) {
        for (int k = 0; k < 5; k++) d[k] += (double)(hp[b * 8 + k + 1] - hp[b * 8 + k]);
        trips += (double)hp[b * 8 + 6];
        const double tot = (double)(hp[b * 8 + 5] - hp[b * 8]);
        if (tot > tmax) tmax = tot;
      }
      fprintf(stderr, "k_unpack prof: %d frames %d wgs | ticks/wg: setup %.0f head %.0f loop %.0f drain %.0f tail %.0f | trips %.1f -> %.1f ticks/trip | longest wg %.0f\n",
              n_frames, blocks, d[0] / blocks, d[1] / blocks, d[2] / blocks, d[3] / blocks, d[4] / blocks, trips / blocks,
              trips > 0 ? d[2] / trips : 0.0, tmax);
    }
  }
  // what each block of 64 frames does to the values that survive frames needs nothing of the batch before ...
  const unsigned merge_blocks_n = (unsigned)((n_frames + kMergeBlk - 1) / kMergeBlk);
  uint32_t* d_sup = t.d_outc + (size_t)merge_blocks_n * kMergeLanes;
  hipLaunchKernelGGL(k_merge_outcome, dim3(merge_blocks_n), dim3(kMergeLanes), 0, t.stream, t.d_raw, t.d_bits, n_frames, t.d_outc, d_sup, t.d_mcnt);
  HIP_TRY(hipGetLastError(), "launch k_merge_outcome");
  // ... everything from here on continues it (scalefactor / count1 carry, synthesis state)
  if (hs->have_state_ev) HIP_TRY(hipStreamWaitEvent(t.stream, hs->ev_state, 0), "wait for the previous batch's state");
  hipLaunchKernelGGL(k_merge_apply, dim3(merge_blocks_n), dim3(kMergeLanes), 0, t.stream, t.d_raw, t.d_bits, n_frames, t.d_outc, d_sup,
                     hs->d_sfstate + 256 * hs->sf_cur, hs->d_sfstate + 256 * (hs->sf_cur ^ 1), t.d_side);
  HIP_TRY(hipGetLastError(), "launch k_merge_apply");
  hs->sf_cur ^= 1;
  // A destination in THIS device's memory that takes whole 4608-byte rows: the kernel stores the PCM there itself (the
  // copy from the slot's buffer was 11 us of a window's 235 -- 75 MB through HBM for 8192 frames; end to end, A/B on one
  // box, four runs each: 26.5 against 25.9 M frames/s).  Pinned host memory
  // stays with the copy command: stores over PCIe from 256 CUs are slower than the DMA engine.
  int16_t* pcm_out = t.d_pcm;
  if (host_dst && row == PDMP3_FRAME_PCM_BYTES && !((uintptr_t)host_dst & 15)) {
    hipPointerAttribute_t pa, pe;
    if (hipPointerGetAttributes(&pa, host_dst) == hipSuccess &&
        hipPointerGetAttributes(&pe, (const char*)host_dst + n * PDMP3_FRAME_PCM_BYTES - 1) == hipSuccess) {
      if (pa.type == hipMemoryTypeDevice && pe.type == hipMemoryTypeDevice && pa.device == hs->ctx->device && pe.device == hs->ctx->device && !pa.isManaged)
        pcm_out = (int16_t*)host_dst;
    } else (void)hipGetLastError();
  }
  rc = launch_decode(hs->ctx, t.d_spectra, t.d_side, n_frames, hs->d_state, pcm_out, nullptr, 0, t.stream, nullptr, hs->d_state_tmp, nullptr, hs, true);
  if (rc != PDMP3_HIP_OK) return rc;
  { float* x = hs->d_state; hs->d_state = hs->d_state_tmp; hs->d_state_tmp = x; }   // (the new state is where the kernel left it)
  HIP_TRY(hipEventRecord(hs->ev_state, t.stream), "record state event");
  hs->have_state_ev = 1;
  if (pcm_out == t.d_pcm) {
    rc = download_pcm(t, n, host_dst, row);
    if (rc != PDMP3_HIP_OK) return rc;
  }
  HIP_TRY(hipEventRecord(t.done, t.stream), "record done event");
  t.busy = 1;
  return PDMP3_HIP_OK;
}

extern "C" int pdmp3_hip_stream_fetch_records(pdmp3_hip_stream* hs, int slot, int n_frames, int16_t* spectra, pdmp3_gc_side* side) {
  if (!SLOT_OK(hs, slot) || n_frames < 0 || n_frames > hs->max_frames || !spectra || !side)
    return fail(PDMP3_HIP_EINVAL, "pdmp3_hip_stream_fetch_records: bad argument", hipSuccess);
  HIP_TRY(hipSetDevice(hs->ctx->device), "hipSetDevice");
  HIP_TRY(hipStreamSynchronize(hs->s[slot].stream), "stream sync");
  HIP_TRY(hipMemcpy(spectra, hs->s[slot].d_spectra, (size_t)n_frames * PDMP3_FRAME_SPECTRA_BYTES, hipMemcpyDeviceToHost), "D2H spectra");
  HIP_TRY(hipMemcpy(side, hs->s[slot].d_side, (size_t)n_frames * PDMP3_FRAME_SIDE_BYTES, hipMemcpyDeviceToHost), "D2H side");
  return PDMP3_HIP_OK;
}

extern "C" int pdmp3_hip_stream_decode(pdmp3_hip_stream* hs, int n_frames) {
  int rc = pdmp3_hip_stream_submit(hs, 0, n_frames);
  if (rc != PDMP3_HIP_OK) return rc;
  return pdmp3_hip_stream_wait(hs, 0);
}

extern "C" int pdmp3_hip_generate_frames(pdmp3_hip_ctx* ctx, uint64_t seed, int64_t first_frame, int n_frames,
                                         int16_t* d_spectra, pdmp3_gc_side* d_side, void* stream) {
  if (!ctx || !d_spectra || !d_side || n_frames < 0)
    return fail(PDMP3_HIP_EINVAL, "pdmp3_hip_generate_frames: bad argument", hipSuccess);
  if (n_frames == 0) return PDMP3_HIP_OK;
  HIP_TRY(hipSetDevice(ctx->device), "hipSetDevice");
  hipLaunchKernelGGL(k_generate, dim3((unsigned)n_frames * 4u), dim3(64), 0, (hipStream_t)stream, seed, first_frame,
                     d_spectra, d_side);
  HIP_TRY(hipGetLastError(), "launch k_generate");
  return PDMP3_HIP_OK;
}

extern "C" int pdmp3_host_generate_frames(uint64_t seed, int64_t first_frame, int n_frames, int16_t* spectra,
                                          pdmp3_gc_side* side) {
  if (!spectra || !side || n_frames < 0) return fail(PDMP3_HIP_EINVAL, "pdmp3_host_generate_frames: bad argument", hipSuccess);
  for (int f = 0; f < n_frames; ++f)
    for (int gc = 0; gc < 4; ++gc)
      for (int lane = 0; lane < 64; ++lane)
        gen_gc(seed, first_frame + f, (unsigned)(gc >> 1), (unsigned)(gc & 1), lane,
               spectra + ((size_t)f * 4 + gc) * 576, side + (size_t)f * 4 + gc);
  return PDMP3_HIP_OK;
}

// Test entry (not part of the product boundary): the two merge kernels alone on merge input from the host -- raw: n_frames x 4
// GcRaw (320 bytes per frame, unpack_core.h), bits: the frames' side-info records, state_in / state_out: 256 uint16 each, side:
// n_frames x 4 records out.  tests/test_gpu_bulk.py runs it on random input beside the rule's host form.
extern "C" int pdmp3_hip_debug_merge(pdmp3_hip_ctx* ctx, const void* raw, const pdmp3_frame_bits* bits, int n_frames,
                                     const uint16_t* state_in, uint16_t* state_out, pdmp3_gc_side* side) {
  if (!ctx || !raw || !bits || !state_in || !state_out || !side || n_frames <= 0)
    return fail(PDMP3_HIP_EINVAL, "pdmp3_hip_debug_merge: bad argument", hipSuccess);
  HIP_TRY(hipSetDevice(ctx->device), "hipSetDevice");
  const size_t n = (size_t)n_frames, nblk = (n + kMergeBlk - 1) / kMergeBlk, nsup = (nblk + kMergeSuper - 1) / kMergeSuper;
  GcRaw* d_raw = nullptr; pdmp3_frame_bits* d_bits = nullptr; uint32_t* d_outc = nullptr; unsigned* d_cnt = nullptr;
  uint16_t* d_st = nullptr; pdmp3_gc_side* d_side = nullptr;
  int rc = PDMP3_HIP_OK;
  do {
#define DM_STEP(call, text) if ((call) != hipSuccess) { rc = fail(PDMP3_HIP_EDEVICE, text, hipGetLastError()); break; }
    DM_STEP(hipMalloc((void**)&d_raw, n * 4 * sizeof(GcRaw)), "hipMalloc raw")
    DM_STEP(hipMalloc((void**)&d_bits, n * sizeof(pdmp3_frame_bits)), "hipMalloc bits")
    DM_STEP(hipMalloc((void**)&d_outc, (nblk + nsup) * kMergeLanes * sizeof(uint32_t)), "hipMalloc outcomes")
    DM_STEP(hipMalloc((void**)&d_cnt, (nsup + 1) * sizeof(unsigned)), "hipMalloc counters")
    DM_STEP(hipMalloc((void**)&d_st, 2 * 256 * sizeof(uint16_t)), "hipMalloc state")
    DM_STEP(hipMalloc((void**)&d_side, n * 4 * sizeof(pdmp3_gc_side)), "hipMalloc side")
    DM_STEP(hipMemset(d_cnt, 0, (nsup + 1) * sizeof(unsigned)), "memset counters")
    DM_STEP(hipMemset(d_side, 0xee, n * 4 * sizeof(pdmp3_gc_side)), "memset side")          // (every byte of the records is the kernel's to write)
    DM_STEP(hipMemcpy(d_raw, raw, n * 4 * sizeof(GcRaw), hipMemcpyHostToDevice), "H2D raw")
    DM_STEP(hipMemcpy(d_bits, bits, n * sizeof(pdmp3_frame_bits), hipMemcpyHostToDevice), "H2D bits")
    DM_STEP(hipMemcpy(d_st, state_in, 256 * sizeof(uint16_t), hipMemcpyHostToDevice), "H2D state")
    hipLaunchKernelGGL(k_merge_outcome, dim3((unsigned)nblk), dim3(kMergeLanes), 0, 0, d_raw, d_bits, n_frames, d_outc, d_outc + nblk * kMergeLanes, d_cnt);
    hipLaunchKernelGGL(k_merge_apply, dim3((unsigned)nblk), dim3(kMergeLanes), 0, 0, d_raw, d_bits, n_frames, d_outc, d_outc + nblk * kMergeLanes,
                       d_st, d_st + 256, d_side);
    DM_STEP(hipGetLastError(), "launch merge kernels")
    DM_STEP(hipDeviceSynchronize(), "sync")
    DM_STEP(hipMemcpy(side, d_side, n * 4 * sizeof(pdmp3_gc_side), hipMemcpyDeviceToHost), "D2H side")
    DM_STEP(hipMemcpy(state_out, d_st + 256, 256 * sizeof(uint16_t), hipMemcpyDeviceToHost), "D2H state")
    unsigned left[64] = {0};                              // (the super-blocks' counters are back at zero: the next launch finds them so)
    DM_STEP(hipMemcpy(left, d_cnt, (nsup < 64 ? nsup : 64) * sizeof(unsigned), hipMemcpyDeviceToHost), "D2H counters")
    for (size_t i = 0; i < (nsup < 64 ? nsup : 64); i++) if (left[i]) rc = fail(PDMP3_HIP_EDEVICE, "pdmp3_hip_debug_merge: a super-block's counter did not wrap", hipSuccess);
#undef DM_STEP
  } while (0);
  (void)hipFree(d_raw); (void)hipFree(d_bits); (void)hipFree(d_outc); (void)hipFree(d_cnt); (void)hipFree(d_st); (void)hipFree(d_side);
  return rc;
}

// Debug/profiling entry (not part of the product boundary): per-chunk shader-clock
// ticks spent in each pipeline phase.  d_prof: uint64 [n_chunks][10].
extern "C" int pdmp3_hip_debug_profile_phases(pdmp3_hip_ctx* ctx, const int16_t* d_spectra, const pdmp3_gc_side* d_side,
                                              int n_frames, int16_t* d_pcm, int chunk_frames,
                                              unsigned long long* d_prof, void* stream) {
  if (!d_prof) return fail(PDMP3_HIP_EINVAL, "pdmp3_hip_debug_profile_phases: d_prof is NULL", hipSuccess);
  return launch_decode(ctx, d_spectra, d_side, n_frames, nullptr, d_pcm, nullptr, chunk_frames, stream, d_prof);
}

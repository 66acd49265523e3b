// emul.cpp -- CPU build of pdmp3_amd/csrc/decode_core.h for the test-suite.
//
// TEST INFRASTRUCTURE.  Runs the device pipeline -- the same source the GPU runs, matrix-instruction fragment
// layouts included -- as 64 fibers per wave (wave_emul.h) so that `pytest -m "not gpu"` can check the kernel's
// indexing, table construction, fragment layouts and chunk/halo logic against the oracle in a
// container without a GPU.  Never loaded by the product (pdmp3_amd/).
#include "wave_emul.h"
#include "../../pdmp3_amd/csrc/decode_core.h"
#include "../../pdmp3_amd/csrc/host_tables.h"
#include "../../pdmp3_amd/csrc/gen_core.h"

#include <memory>
#include <type_traits>
#include <vector>

using namespace pdmp3;

extern "C" int emul_decode_frames(const int16_t* spectra, const pdmp3_gc_side* side, int n_frames,
                                  float* state, int16_t* pcm, float* stages, int chunk_frames) {
  static HostTables H;
  static bool ready = false;
  if (!ready) { build_host_tables(H); ready = true; }
  GlobalTables T{H.pow43.data(), H.linetab.data(), H.win.data(), H.frag_long.data(), H.frag_short.data(), H.frag_mat.data(), H.taps.data(), H.tab_image.data()};
  if (chunk_frames <= 0) chunk_frames = n_frames;
  if (stages) chunk_frames = n_frames;
  // like engine.hip: the kernel writes the new state to scratch (any chunk may read the old one), then it is copied
  std::vector<float> state_next(kStateFloats);
  DecodeArgs a{spectra, side, pcm, nullptr, state, state ? state_next.data() : nullptr, stages, n_frames, chunk_frames, nullptr, nullptr, nullptr, 0u};
  const int nchunks = (n_frames + chunk_frames - 1) / chunk_frames;
  auto L = std::make_unique<WaveLds>();
  for (int c = 0; c < nchunks; ++c) {
    WaveLds& Lr = *L;
    // like engine.hip k_decode: the copy of the chunk's code without intensity stereo / LSF unless the chunk holds such a frame
    if (stages) emu::run_wave([&] { run_chunk<true>(a, T, &H.cb, c, Lr, Lr.tab); });
    else emu::run_wave([&] {
      if (chunk_is_rare(a, c)) run_chunk<false, false, false, true, true>(a, T, &H.cb, c, Lr, Lr.tab);
      else run_chunk<false, false, false, true, false>(a, T, &H.cb, c, Lr, Lr.tab);
    });
  }
  if (state) std::copy(state_next.begin(), state_next.end(), state);
  return 0;
}

// LSF frames (pdmp3_hip_decode_lsf_frames: engine.hip k_lsf_pair + the chunk kernel with DecodeArgs::n_gran): the pairing
// done here on the host, then the same run_chunk
extern "C" int emul_decode_lsf_frames(const int16_t* spectra, const pdmp3_gc_side* side, int n_frames,
                                      float* state, int16_t* pcm, float* pcm_f32, int chunk_frames) {
  static HostTables H;
  static bool ready = false;
  if (!ready) { build_host_tables(H); ready = true; }
  GlobalTables T{H.pow43.data(), H.linetab.data(), H.win.data(), H.frag_long.data(), H.frag_short.data(), H.frag_mat.data(), H.taps.data(), H.tab_image.data()};
  const int np = (n_frames + 1) / 2;
  std::vector<int16_t> psp((size_t)np * 2304, 0);
  std::vector<pdmp3_gc_side> psd((size_t)np * 4);
  memset(psd.data(), 0, psd.size() * sizeof(pdmp3_gc_side));
  for (int f = 0; f < n_frames; f++) {
    memcpy(&psp[(size_t)(f >> 1) * 2304 + (f & 1) * 1152], spectra + (size_t)f * 2304, 1152 * sizeof(int16_t));
    memcpy(&psd[(size_t)(f >> 1) * 4 + (f & 1) * 2], side + (size_t)f * 4, 2 * sizeof(pdmp3_gc_side));
  }
  if (chunk_frames <= 0) chunk_frames = np;
  std::vector<float> state_next(kStateFloats);
  DecodeArgs a{psp.data(), psd.data(), pcm, pcm_f32, state, state ? state_next.data() : nullptr, nullptr, np, chunk_frames, nullptr, nullptr, nullptr, 0u};
  a.n_gran = n_frames;
  const int nchunks = (np + chunk_frames - 1) / chunk_frames;
  auto L = std::make_unique<WaveLds>();
  for (int c = 0; c < nchunks; ++c) {
    WaveLds& Lr = *L;
    if (pcm_f32) emu::run_wave([&] { run_chunk<false, false, true>(a, T, &H.cb, c, Lr, Lr.tab); });
    else emu::run_wave([&] { run_chunk<false>(a, T, &H.cb, c, Lr, Lr.tab); });
  }
  if (state) std::copy(state_next.begin(), state_next.end(), state);
  return 0;
}

// float PCM form (pdmp3_hip_decode_frames_f32)
extern "C" int emul_decode_frames_f32(const int16_t* spectra, const pdmp3_gc_side* side, int n_frames,
                                      float* state, float* pcm, int chunk_frames) {
  static HostTables H;
  static bool ready = false;
  if (!ready) { build_host_tables(H); ready = true; }
  GlobalTables T{H.pow43.data(), H.linetab.data(), H.win.data(), H.frag_long.data(), H.frag_short.data(), H.frag_mat.data(), H.taps.data(), H.tab_image.data()};
  if (chunk_frames <= 0) chunk_frames = n_frames;
  std::vector<float> state_next(kStateFloats);
  DecodeArgs a{spectra, side, nullptr, pcm, state, state ? state_next.data() : nullptr, nullptr, n_frames, chunk_frames, nullptr, nullptr, nullptr, 0u};
  const int nchunks = (n_frames + chunk_frames - 1) / chunk_frames;
  auto L = std::make_unique<WaveLds>();
  for (int c = 0; c < nchunks; ++c) {
    WaveLds& Lr = *L;
    emu::run_wave([&] {
      if (chunk_is_rare(a, c)) run_chunk<false, false, true, true, true>(a, T, &H.cb, c, Lr, Lr.tab);
      else run_chunk<false, false, true, true, false>(a, T, &H.cb, c, Lr, Lr.tab);
    });
  }
  if (state) std::copy(state_next.begin(), state_next.end(), state);
  return 0;
}

// one granule per wave (run_granule_wave), as the engine's granule kernel launches it: workgroups of 16 consecutive granules
// sharing one table block; the waves of a workgroup are live together (they hand their states on through each other's
// LDS), the workgroups run one after the other
extern "C" int emul_decode_frames_granules(const int16_t* spectra, const pdmp3_gc_side* side, int n_frames,
                                           float* state, int16_t* pcm, float* pcm_f32, unsigned debug_flags, int sf_hint) {
  static HostTables H;
  static bool ready = false;
  if (!ready) { build_host_tables(H); ready = true; }
  GlobalTables T{H.pow43.data(), H.linetab.data(), H.win.data(), H.frag_long.data(), H.frag_short.data(), H.frag_mat.data(), H.taps.data(), H.tab_image.data()};
  std::vector<float> state_next(kStateFloats);
  std::vector<float> cstate((size_t)n_frames * 2 * kGranFloats);
  std::vector<unsigned> cflag((size_t)n_frames * 4, 0u);
  DecodeArgs a{spectra, side, pcm, pcm_f32, state, state ? state_next.data() : nullptr, nullptr, n_frames, 1, nullptr,
               cstate.data(), cflag.data(), 7u, debug_flags, sf_hint};
  constexpr int WPW = 16;
  auto L = std::make_unique<WaveData[]>(WPW);
  auto S = std::make_unique<TabLds>();
  emu::run_wave([&] { tab_load_image(emu::lane(), 64, *S, T, sf_hint); });
  unsigned tabs_ready = WPW;
  GranMb mb[WPW];
  for (int g0 = 0; g0 < 2 * n_frames; g0 += WPW) {
    const int nw = 2 * n_frames - g0 < WPW ? 2 * n_frames - g0 : WPW;
    memset(mb, 0, sizeof mb);
    std::function<void()> bodies[WPW];
    for (int w = 0; w < nw; ++w) {
      bodies[w] = [&, w] {
        const int g = g0 + w;
        const GranPos gp{L.get(), mb, w, WPW, &tabs_ready};
        LaneRegs pf;
        ph_prefetch(emu::lane(), pf, a.spectra + (size_t)g * 1152, a.side + (size_t)g * 2);
        if (pcm_f32) run_granule_wave<true>(a, T, &H.cb, g, L[w], *S, gp, pf);
        else run_granule_wave<false>(a, T, &H.cb, g, L[w], *S, gp, pf);
      };
    }
    emu::run_waves(bodies, nw);
  }
  if (state) std::copy(state_next.begin(), state_next.end(), state);
  return 0;
}

// the persistent granule kernel (run_granule_ring, engine.hip k_decode_p): workgroups of 16 waves, each going round a
// contiguous range of `frames_per_wg` frames; the waves of a workgroup are live together, the workgroups run one after the other
extern "C" int emul_decode_frames_ring(const int16_t* spectra, const pdmp3_gc_side* side, int n_frames,
                                       float* state, int16_t* pcm, float* pcm_f32, int frames_per_wg, int sf_hint) {
  static HostTables H;
  static bool ready = false;
  if (!ready) { build_host_tables(H); ready = true; }
  GlobalTables T{H.pow43.data(), H.linetab.data(), H.win.data(), H.frag_long.data(), H.frag_short.data(), H.frag_mat.data(), H.taps.data(), H.tab_image.data()};
  std::vector<float> state_next(kStateFloats);
  DecodeArgs a{spectra, side, pcm, pcm_f32, state, state ? state_next.data() : nullptr, nullptr, n_frames, 1, nullptr,
               nullptr, nullptr, 0u, 0u, sf_hint};
  constexpr int WPW = 16;
  auto L = std::make_unique<WaveData[]>(WPW);
  auto S = std::make_unique<TabLds>();
  emu::run_wave([&] { tab_load_image(emu::lane(), 64, *S, T, sf_hint); });
  unsigned tabs_ready = WPW;
  GranMb mb[WPW];
  if (frames_per_wg < 1) frames_per_wg = 1;
  for (int f0 = 0; f0 < n_frames; f0 += frames_per_wg) {
    const int f1 = f0 + frames_per_wg < n_frames ? f0 + frames_per_wg : n_frames;
    memset(mb, 0, sizeof mb);
    std::function<void()> bodies[WPW];
    for (int w = 0; w < WPW; ++w) {
      bodies[w] = [&, w] {
        const GranPos gp{L.get(), mb, w, WPW, &tabs_ready, 1, 2 * f0, 2 * f1};
        if (pcm_f32) run_granule_ring<true>(a, T, &H.cb, L[w], *S, gp, f0, f1);
        else run_granule_ring<false>(a, T, &H.cb, L[w], *S, gp, f0, f1);
      };
    }
    emu::run_waves(bodies, WPW);
  }
  if (state) std::copy(state_next.begin(), state_next.end(), state);
  return 0;
}

extern "C" size_t emul_state_floats() { return kStateFloats; }

extern "C" void emul_generate_frames(uint64_t seed, int64_t first, int n, int16_t* spectra, pdmp3_gc_side* side) {
  for (int f = 0; f < n; ++f)
    for (int gc = 0; gc < 4; ++gc)
      for (int lane = 0; lane < 64; ++lane)
        gen_gc(seed, first + f, gc >> 1, gc & 1, lane, spectra + ((size_t)f * 4 + gc) * 576, side + (size_t)f * 4 + gc);
}

extern "C" int emul_ldexp_forms_exact() { static HostTables H; build_host_tables(H); return H.ldexp_forms_exact ? 1 : 0; }

extern "C" void emul_tables(float* pow43, float* t1, float* t2) {
  static HostTables H;
  build_host_tables(H);
  memcpy(pow43, H.pow43.data(), 8207 * 4);
  memcpy(t1, H.t1.data(), kT1Size * 4);
  memcpy(t2, H.t2.data(), kT2Size * 4);
}

// ---- unpack_core.h on the host: every granule-channel, then every merge slot --------------------------------
#include "../../pdmp3_amd/csrc/unpack_core.h"
extern "C" int emul_unpack_frames(const pdmp3_frame_bits* bits, const uint8_t* res, int n_frames, uint16_t* state /*[256]*/,
                                  int16_t* spectra, pdmp3_gc_side* side) {
  static UnpackTables* U = nullptr;
  if (!U) { U = new UnpackTables; if (!build_unpack_tables(*U)) return -1; }
  std::vector<GcRaw> raw((size_t)n_frames * 4);
  std::vector<uint32_t> padded(((size_t)n_frames * PDMP3_RESERVOIR_BYTES + 16) / 4, 0);   // peek64 may touch 4 bytes past the last row
  memcpy(padded.data(), res, (size_t)n_frames * PDMP3_RESERVOIR_BYTES);
  res = reinterpret_cast<const uint8_t*>(padded.data());
  memset(spectra, 0, (size_t)n_frames * 2304 * sizeof(int16_t));
  for (int f = 0; f < n_frames; ++f)
    for (int g = 0; g < 4; ++g)
      unpack_gc(*U, U->lut, res + (size_t)f * PDMP3_RESERVOIR_BYTES, bits[f], g, spectra + ((size_t)f * 4 + g) * 576,
                side + (size_t)f * 4 + g, &raw[(size_t)f * 4 + g]);
  uint16_t st_in[256];
  memcpy(st_in, state, sizeof st_in);
  // the merge by blocks of 32 frames (what k_merge_outcome + k_merge_apply run) on a copy of the records, beside the rule
  // itself (merge_slot: one slot through all the frames in order): the two must agree in every byte and in the state
  std::vector<pdmp3_gc_side> side_b(side, side + (size_t)n_frames * 4);
  std::vector<uint32_t> outc((size_t)merge_outcome_rows(n_frames) * kMergeLanes + 1, 0);
  uint16_t state_b[256];
  memcpy(state_b, state, sizeof state_b);
  merge_blocks(raw.data(), bits, n_frames, st_in, state_b, side_b.data(), outc.data());
  for (int t = 0; t < kMergeSlots; ++t) merge_slot(t, raw.data(), bits, n_frames, st_in, state, side);
  if (n_frames > 0 && memcmp(side_b.data(), side, (size_t)n_frames * 4 * sizeof(pdmp3_gc_side)) != 0) return -2;
  if (memcmp(state_b, state, kMergeSlots * sizeof(uint16_t)) != 0) return -3;
  return (int)U->n_lut;
}


// the merge alone on RANDOM merge input (every combination of set / copy bits, mono frames, new streams in the middle of a
// block, the two ISO switches that keep the one-past-the-end slots zero): merge_blocks -- the form of the two kernels --
// against merge_slot, records and carried state.  Returns 0, or 1 + the index of the first frame whose records differ,
// or -1 when only the state differs.
// (emul_merge_case: the same, and the case goes to the caller -- the merge input, the frames' records, the incoming state, and
//  what the rule makes of them -- for the GPU test that runs the two kernels on it; any of the pointers may be NULL)
extern "C" int emul_merge_case(uint64_t seed, int n_frames, int p_set, int p_copy, int p_mono, int p_new, void* raw_out, pdmp3_frame_bits* bits_out,
                               uint16_t* st_in_out, pdmp3_gc_side* side_ref, uint16_t* st_ref);
extern "C" int emul_merge_fuzz(uint64_t seed, int n_frames, int p_set, int p_copy, int p_mono, int p_new) {
  return emul_merge_case(seed, n_frames, p_set, p_copy, p_mono, p_new, nullptr, nullptr, nullptr, nullptr, nullptr);
}
extern "C" int emul_merge_raw_bytes() { return (int)sizeof(GcRaw); }
extern "C" int emul_merge_case(uint64_t seed, int n_frames, int p_set, int p_copy, int p_mono, int p_new, void* raw_out, pdmp3_frame_bits* bits_out,
                               uint16_t* st_in_out, pdmp3_gc_side* side_ref, uint16_t* st_ref) {
  auto rnd = [&seed]() { seed = seed * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(seed >> 33); };
  std::vector<GcRaw> raw((size_t)n_frames * 4);
  std::vector<pdmp3_frame_bits> bits((size_t)n_frames);
  memset(raw.data(), 0, raw.size() * sizeof(GcRaw));
  memset(bits.data(), 0, bits.size() * sizeof(pdmp3_frame_bits));
  for (int f = 0; f < n_frames; f++) {
    const bool mono = (int)(rnd() % 100) < p_mono;
    bits[f].frame = (uint8_t)((mono ? 3u : rnd() % 3u) << PDMP3_FR_MODE_SHIFT);
    if (f == 0 || (int)(rnd() % 1000) < p_new) bits[f].frame |= PDMP3_FR_NEWSTREAM;
    bits[f].iso = (uint8_t)((rnd() % 8 == 0 ? 0x08u : 0u) | (rnd() % 8 == 0 ? 0x10u : 0u));
    for (int g = 0; g < 4; g++) {
      GcRaw& r = raw[(size_t)f * 4 + g];
      if (mono && (g & 1)) continue;                          // (what unpack_scalefactors leaves for a channel that is not there: zeroes)
      if ((int)(rnd() % 100) < p_set) { r.count1 = (uint16_t)(rnd() % 577); r.count1_set = 1; }
      const bool shrt = rnd() % 4 == 0, mixed = shrt && rnd() % 2;
      if (shrt) {
        if (mixed) { r.sf_l_set = 0xffu; for (int b = 0; b < 8; b++) r.sf_l[b] = (uint8_t)(rnd() % 16); }
        unsigned set = 0;
        for (int b = mixed ? 3 : 0; b < 12; b++) { set |= 1u << b; for (int w = 0; w < 3; w++) r.sf_s[b * 3 + w] = (uint8_t)(rnd() % 16); }
        r.sf_s_set = (uint16_t)set;
      } else if ((int)(rnd() % 100) < p_set) {
        unsigned set = 0, copy = 0;
        for (int g4 = 0; g4 < 4; g4++) {
          const int lo = g4 ? 1 + 5 * g4 : 0, hi = 6 + 5 * g4;
          if (g >= 2 && (int)(rnd() % 100) < p_copy) copy |= 1u << g4;
          else for (int b = lo; b < hi; b++) { r.sf_l[b] = (uint8_t)(rnd() % 16); set |= 1u << b; }
        }
        r.sf_l_set = set; r.sf_l_copy = (uint8_t)copy;
      }
    }
  }
  uint16_t st_in[256], st_a[256], st_b[256];
  for (int i = 0; i < 256; i++) st_in[i] = st_a[i] = st_b[i] = (uint16_t)(rnd() % (i >= 228 ? 577 : 16));
  std::vector<pdmp3_gc_side> side_a((size_t)n_frames * 4), side_b((size_t)n_frames * 4);
  memset(side_a.data(), 0, side_a.size() * sizeof(pdmp3_gc_side));
  for (int f = 0; f < n_frames; f++) for (int g = 0; g < 4; g++) side_fields(bits[f], g, &side_a[(size_t)f * 4 + g]);
  side_b = side_a;
  std::vector<uint32_t> outc((size_t)merge_outcome_rows(n_frames) * kMergeLanes + 1, 0);
  for (int t = 0; t < kMergeSlots; ++t) merge_slot(t, raw.data(), bits.data(), n_frames, st_in, st_a, side_a.data());
  merge_blocks(raw.data(), bits.data(), n_frames, st_in, st_b, side_b.data(), outc.data());
  if (raw_out) memcpy(raw_out, raw.data(), raw.size() * sizeof(GcRaw));
  if (bits_out) memcpy(bits_out, bits.data(), bits.size() * sizeof(pdmp3_frame_bits));
  if (st_in_out) memcpy(st_in_out, st_in, sizeof st_in);
  if (side_ref) memcpy(side_ref, side_a.data(), side_a.size() * sizeof(pdmp3_gc_side));
  if (st_ref) memcpy(st_ref, st_a, sizeof st_a);
  for (int f = 0; f < n_frames; f++)
    if (memcmp(&side_a[(size_t)f * 4], &side_b[(size_t)f * 4], 4 * sizeof(pdmp3_gc_side)) != 0) return 1 + f;
  return memcmp(st_a, st_b, kMergeSlots * sizeof(uint16_t)) != 0 ? -1 : 0;
}

// reservoir rows from the pool (unpack_core.h row_chunk16: what k_rows runs per frame), and word by word (row_word: the
// rule itself; the tests ask for both and want them equal)
extern "C" void emul_rows(const pdmp3_row_desc* desc, const uint8_t* pool, int n_frames, uint8_t* rows) {
  for (int f = 0; f < n_frames; ++f)
    for (unsigned c = 0; c < PDMP3_RESERVOIR_BYTES / 16; ++c) {
      uint32_t v[4];
      row_chunk16(desc + f, pool, 16 * c, v);
      memcpy(rows + (size_t)f * PDMP3_RESERVOIR_BYTES + 16 * c, v, 16);
    }
}
extern "C" void emul_rows_by_word(const pdmp3_row_desc* desc, const uint8_t* pool, int n_frames, uint8_t* rows) {
  for (int f = 0; f < n_frames; ++f)
    for (unsigned w = 0; w < PDMP3_RESERVOIR_BYTES / 4; ++w) {
      const uint32_t v = row_word(desc + f, pool, 4 * w);
      memcpy(rows + (size_t)f * PDMP3_RESERVOIR_BYTES + 4 * w, &v, 4);
    }
}

// pcm_convert18_lane (the device's int16 conversion: RTZ product, v_med3 clamp, the reference's wrap-around) for n groups
// of 18 sums, both forms (`wrap` = what the wave-wide test would have said), beside pcm_from_sum (P:2028-2031 restated
// with the binary64 product): the test compares the three, NaNs, infinities and sums beyond the wrap point included
extern "C" void emul_pcm_convert18(const float* sums, int n_groups, int* out_fast, int* out_wrap, int* out_exact) {
  for (int g = 0; g < n_groups; ++g) {
    pcm_convert18_lane(sums + 18 * g, out_fast + 18 * g, false);
    pcm_convert18_lane(sums + 18 * g, out_wrap + 18 * g, true);
    for (int t = 0; t < 18; ++t) out_exact[18 * g + t] = pcm_from_sum(sums[18 * g + t]);
  }
}

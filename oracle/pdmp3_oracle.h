/*
 * pdmp3_oracle.h -- CPU ORACLE for the PDMP3 hot path.  TEST INFRASTRUCTURE.
 *
 * This is a plain-C restatement of the reference algorithm (technosaurus/PDMP3
 * pdmp3.c) used ONLY as the checker: by tests/, by __graft_entry__.smoke() and
 * by bench.py's cpu_baseline leg.  Nothing under pdmp3_amd/ may include, link
 * or call it; the product path fails loudly when its HIP library is missing.
 *
 * Parity status: PINNED.  The reference ships no golden vectors,
 * so the oracle is pinned against the reference itself, compiled from
 * /root/reference by oracle/Makefile into oracle/_ref/ (tests/test_oracle_vs_ref.py,
 * bit-exact, 0 LSB / 0 ulp), and against the committed fixtures those runs
 * produced (tests/golden/, generator: tools/make_golden.py).
 *
 * "P:n" = /root/reference/pdmp3.c line n.
 */
#ifndef PDMP3_ORACLE_H
#define PDMP3_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include "../include/pdmp3_hip.h"   /* the gc record = the boundary under test */

#ifdef __cplusplus
extern "C" {
#endif

/* Synthesis state carried from granule to granule (per stream; the reference
 * keeps these as function-statics, P:1755 and P:1983). */
typedef struct orc_synth {
  float store[2][32][18];
  float v_vec[2][1024];
} orc_synth;

void orc_synth_reset(orc_synth* st);

/* Decode n_frames frames given as gc records (same layout the HIP engine
 * takes).  pcm: int16, 4608 bytes per frame (mono: first 2304 used).
 * stages (nullable): float [n_frames][2][2][4][576], see pdmp3_hip.h.
 * Returns 0. */
int orc_decode_frames(orc_synth* st, const int16_t* spectra,
                      const pdmp3_gc_side* side, int n_frames,
                      int16_t* pcm, float* stages);

/* the same, also handing out the binary32 synthesis sums that P:2028-2031 scale, truncate and clip into int16
 * (float PCM): pcm_f32 (nullable) interleaved like pcm, 2304 floats per frame; pcm nullable too */
int orc_decode_frames_f32(orc_synth* st, const int16_t* spectra,
                          const pdmp3_gc_side* side, int n_frames,
                          int16_t* pcm, float* pcm_f32, float* stages);

/* SURVEY 8d synthetic generator (C2/C5), integer-only. */
void orc_generate_frames(uint64_t seed, int64_t first_frame, int n_frames,
                         int16_t* spectra, pdmp3_gc_side* side);

/* libm-derived tables exactly as the reference builds them; exposed so tests
 * can compare the product's uploaded copies.  Returned pointers are static. */
const float* orc_table_pow43(void);     /* [8207]   P:979  */
const float* orc_table_nwin(void);      /* [64][32] P:1992 */

/* ---------------- bitstream front end (host stage restated) ------------- */

typedef struct orc_stream orc_stream;

#define ORC_OK          0
#define ORC_ERR        -1
#define ORC_NEED_MORE -10
#define ORC_NEW_FORMAT -11
#define ORC_NO_SPACE    7

orc_stream* orc_stream_new(void);                  /* pdmp3_new + zeroed memory (H13) */
void orc_stream_delete(orc_stream*);
int orc_stream_open_feed(orc_stream*);             /* P:2369 */
int orc_stream_feed(orc_stream*, const unsigned char* in, size_t size);      /* P:2391 */
int orc_stream_read(orc_stream*, unsigned char* out, size_t outsize, size_t* done); /* P:2431 */
int orc_stream_decode(orc_stream*, const unsigned char* in, size_t insize,
                      unsigned char* out, size_t outsize, size_t* done);     /* P:2491 */
int orc_stream_getformat(orc_stream*, long* rate, int* channels, int* enc);  /* P:2526 */

/* Frame-record tap: when set, every successfully parsed frame's 4 gc records
 * and spectra (taken after Read_Frame, before the transforms, exactly what the
 * HIP engine would be given) are appended here. */
typedef struct orc_tap {
  int16_t* spectra;        /* capacity cap_frames * 2304 int16 */
  pdmp3_gc_side* side;     /* capacity cap_frames * 4 */
  int cap_frames;
  int n_frames;
} orc_tap;
void orc_stream_set_tap(orc_stream*, orc_tap* tap);
/* NOT the reference: the ISO-correct switches of include/pdmp3.h (PDMP3_ISO_*, same bit values) restated, so that the
 * library's modes have something to be compared with.  Nothing pins them ("parity unpinned"). */
void orc_stream_set_quirks(orc_stream*, unsigned iso_mask);
/* != 0 once a frame made the reference's line counter wrap (P:2106 on a counter below 4): the reference's behaviour from
 * there on is undefined (it reads and writes past its arrays); the oracle clamps the counter and flags the stream */
int orc_stream_undefined(const orc_stream*);
/* tests only: the 24 kHz long band table with FFmpeg's / mpg123's entry 330 where the standard has 332 (pdmp3_oracle.c) */
void orc_debug_24k_330(int on);

/* Whole-buffer convenience that mimics the CLI driver loop pdmp3() (P:2540):
 * read(16 KiB) -> on NEED_MORE feed 4096 bytes -> ... ; returns PCM bytes. */
size_t orc_decode_buffer_like_cli(const unsigned char* mp3, size_t n,
                                  unsigned char* pcm, size_t pcm_cap,
                                  orc_tap* tap);

#ifdef __cplusplus
}
#endif
#endif

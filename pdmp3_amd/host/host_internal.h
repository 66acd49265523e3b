/*
 * host_internal.h -- what the files of libpdmp3.so share (nothing here is exported: HOST_LOCAL).
 *
 * libpdmp3.so is the libmpg123-style streaming API of PDMP3 (include/pdmp3.h) and the whole-stream decoder
 * (include/pdmp3_bulk.h) over the MI355X transform engine (include/pdmp3_hip.h).  Its host stage -- plain C on the
 * box's host cores, as BASELINE north_star asks -- is the input ring, header sync, side info, bit reservoir,
 * scalefactors and a table-driven Huffman decoder; each parsed frame becomes four granule-channel records
 * (pdmp3_gc_side + int16 spectra) written straight into the engine's pinned staging buffers.
 *
 *   huffman_lut.c   code books -> two-level lookup tables, the frame-size table
 *   frame_parse.c   one frame: header sync, side info, bit reservoir, main data (scalefactors + Huffman), records
 *   stream_api.c    the handle; pdmp3_feed / pdmp3_read / pdmp3_decode / pdmp3_getformat; read-ahead batches and the
 *                   helper threads that decode a batch's main data
 *   cpus.c          how many CPUs the process may use, which of them sit next to the GPU
 *   bulk.c          whole-stream decoder: stages A-D on one scanning thread, worker pool, submitter, main-data copies
 *   split_scan.c    whole-stream decoder: the scan split over threads (pre-pass, hop threads, scanners, stitcher)
 *   bulk_api.c      whole-stream decoder: pdmp3_amd_bulk_* entry points (new / delete / decode / wait / parse hooks)
 *   corpus.c        a corpus of files dealt over the GPUs of a node
 *   wav_cli.c       pdmp3() -- the reference's CLI contract -- and the .raw / .wav sinks
 *
 * Behavioural contract = the reference's (file:line cited at each function, "P:n" = /root/reference/pdmp3.c line n),
 * including the quirks SURVEY.md lists as H1, H6, H7, H9, H10, H16-H18.  The code is written from that contract, not from
 * the reference's source: e.g. Huffman decoding is a two-level lookup built from code books (tables_data.h) instead of the
 * reference's bit-serial tree walk.
 *
 * There is no CPU fallback for the transforms: without the engine library or a HIP device pdmp3_new() fails.
 */
#ifndef PDMP3_HOST_INTERNAL_H
#define PDMP3_HOST_INTERNAL_H
#define _GNU_SOURCE
#include "../../include/pdmp3.h"
#include "../../include/pdmp3_hip.h"
#include "../csrc/tables_data.h"
#include "../csrc/lsf_tables.h"

#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define INBUF_SIZE 16384u            /* P:123 */
#define BATCH_MAX 16                 /* frames per GPU batch: > 16 KiB / 1152 B frames */
#define BYTE_EOF 0xffffffffu


#define HOST_LOCAL __attribute__((visibility("hidden")))   /* shared by the library's files, not exported */

/* ------------------------------------------------------------------------ */
/* Huffman code books -> two-level lookup tables                             */
/* ------------------------------------------------------------------------ */
#define HL_BITS 10
typedef struct {
  uint16_t first[1 << HL_BITS];   /* adv<<8 | val   or   0x8000 | subtable index */
  uint16_t* sub;                  /* subtables of 1 << sub_bits entries: adv<<8 | val */
  int sub_bits;
  int quads;                      /* count1 book: val = v w x y */
} huff_lut;
/* adv = bits of the whole code word + one sign bit per value != 0: what a symbol without linbits takes in all, so that
 * the position of the next symbol -- the loop's dependent chain -- is one table lookup and one add away */
static inline unsigned leaf_nsign(int quads, unsigned val) {
  return quads ? (val & 1) + (val >> 1 & 1) + (val >> 2 & 1) + (val >> 3 & 1) : (unsigned)((val >> 4) != 0) + (unsigned)((val & 15) != 0);
}

/* huffman_lut.c (pthread_once(&g_lut_once, build_luts) before the first parse) */
HOST_LOCAL void build_luts(void);
HOST_LOCAL extern huff_lut g_lut[PDMP3_NUM_HUFF_BOOKS];
HOST_LOCAL extern pthread_once_t g_lut_once;
HOST_LOCAL extern uint16_t g_frame_q[15][3];

/* ------------------------------------------------------------------------ */
/* handle                                                                    */
/* ------------------------------------------------------------------------ */
typedef struct {
  unsigned id, layer, protection, bitrate_index, sfreq, padding, mode, mode_ext;
  unsigned ver;                      /* 0 = MPEG-1 (all the reference takes, P:1293); 1 = MPEG-2 LSF, 2 = MPEG-2.5: only with PDMP3_ISO_LSF */
} frame_header;
/* samples per channel a frame decodes to: 1152; an LSF frame is ONE granule */
static inline unsigned frame_samples(const frame_header* H) { return H->ver ? 576u : 1152u; }
static inline unsigned sfreq9(const frame_header* H) { return 3 * H->ver + H->sfreq; }

typedef struct {
  unsigned main_data_begin, scfsi[2][4];
  unsigned part2_3_length[2][2], big_values[2][2], global_gain[2][2], scalefac_compress[2][2];
  unsigned win_switch[2][2], block_type[2][2], mixed[2][2], table_select[2][2][3], subblock_gain[2][2][3];
  unsigned region0_count[2][2], region1_count[2][2], preflag[2][2], scalefac_scale[2][2], count1table_select[2][2];
} side_info;

/* what one frame's main data yields; merged into the handle's persistent state in frame order */
typedef struct {
  int16_t* is;                       /* [gr][ch][576] destination: the engine's staging spectra of this frame */
  uint16_t count1[2][2];
  uint8_t count1_set[2][2];          /* 0 when part2_3_length == 0: count1 keeps its old value (H6) */
  uint8_t sf_l[2][2][21];
  uint32_t sf_l_set[2][2];           /* bit sfb: scalefac_l[gr][ch][sfb] was read from the stream */
  uint8_t sf_l_copy[2];              /* [ch] bit b: granule 1 takes band group b from granule 0 (scfsi) */
  uint8_t sf_s[2][2][12][3];
  uint16_t sf_s_set[2][2];           /* bit sfb: scalefac_s[gr][ch][sfb][0..2] were read */
} main_out;

/* everything a frame's parse can change (read_ahead below undoes frames with it) */
typedef struct {
  size_t processed; unsigned istart;
  frame_header hdr; side_info si;
  uint8_t scalefac_l[2][2][21]; uint8_t scalefac_s[2][2][12][3]; uint16_t count1[2][2];
  uint8_t main_vec[2048 + 16]; unsigned main_top;
  uint8_t side_vec[64 + 8]; unsigned side_ptr, side_idx;
  int new_header, need_reset, tap_n;
} parse_snap;

struct pdmp3_handle {
  /* input ring, P:126-128 */
  size_t processed;
  unsigned istart, iend;
  unsigned char in[INBUF_SIZE];
  /* output cursor into the last decoded frame, P:127 (ostart), P:129 (out) */
  unsigned ostart;
  int16_t last_pcm[2304 * 2];      /* (as float when enc_f32: 2304 floats) */
  int enc_f32;                     /* pdmp3_amd_set_encoding: PCM as float (not in the reference) */
  unsigned last_nch;
  /* parse state that survives frames (the reference never clears it) */
  frame_header hdr;
  side_info si;
  uint8_t scalefac_l[2][2][21];
  uint8_t scalefac_s[2][2][12][3];
  uint16_t count1[2][2];
  uint8_t main_vec[2048 + 16];     /* bit reservoir, P:137 */
  unsigned main_top;
  main_out scratch_out;            /* this frame's decoded main data (inline path) */
  uint8_t side_vec[64 + 8];        /* side info bytes, P:138 */
  unsigned side_ptr, side_idx;
  int new_header;                  /* P:147 */
  int need_reset;                  /* hsynth_init / synth_init, P:134-135 */
  int ring_short;                  /* set whenever a parse step found fewer bytes in the ring than it wanted */
  /* Whole-stream decoding (bulk_drive) reads a stream that is in memory anyway: the ring is then only its index
   * arithmetic (istart / iend / processed move exactly as with real feeds) and the bytes come from the buffer:
   * the ring slot ring_filled() places before the write index holds stream byte vfed - ring_filled(), stale slots
   * of a replayed ring included (`processed` is no position: the header search resets it, P:1322-1340). */
  const unsigned char* vsrc;
  size_t vfed;                     /* bytes fed so far in virtual mode */
  /* whole-stream decoding in bits mode: the side info goes straight into the engine's record (read_side_info_bits);
   * fb_cur is valid for the frame just parsed when fb_valid is set */
  int side_to_bits, fb_valid;
  int bits_scan;                   /* this handle scans for a device-Huffman decoder (bits mode) */
  int lsf_seen;                    /* bits mode + PDMP3_ISO_LSF: the scan met an LSF frame, which the device's Huffman stage cannot take --
                                      the whole stream goes to the host-Huffman decoder instead (bulk_api.c bulk_decode_impl) */
  pdmp3_frame_bits fb_cur;
  struct bulk* pool_sink;          /* bits mode with an engine: Get_Main_Data appends to the window's pool (fill_reservoir_pool) */
  /* Read-ahead of pdmp3_read (see read_ahead below).  The parser above may be AHEAD of the stream position the
   * reference would have at this point of the call sequence; what the API shows is the logical view: */
  size_t l_processed;              /* id->processed of the reference */
  unsigned l_istart;               /* id->istart of the reference: pdmp3_feed's free space, the 1152-byte rule */
  frame_header l_hdr;              /* g_frame_header of the reference: pdmp3_getformat, the partial-frame cursor */
  int l_new_header;                /* id->new_header of the reference */
  struct ra_entry {                /* a frame parsed and sent to the engine but not handed out yet */
    size_t processed_after; unsigned istart_after; frame_header hdr; uint8_t nch, nh;
  } ra[BATCH_MAX];                 /* (a batch is homogeneous: all MPEG-1, or all LSF of one version and channel count) */
  int ra_head, ra_n, ra_inflight;  /* ra_inflight: the batch is still on the GPU */
  parse_snap ra_before[BATCH_MAX]; /* the parser as it was before each of these frames */
  main_out ra_out[BATCH_MAX];      /* their decoded main data (the helpers of read_ahead write these) */
  /* engine */
  pdmp3_hip_stream* hs;
  int host_only;                   /* test hook: parse without an engine (no decode possible) */
  unsigned iso;                    /* PDMP3_ISO_*: the standard's behaviour instead of the reference's (pdmp3_amd_set_quirks) */
  /* record tap for tests (host logic without GPU) */
  int16_t* tap_spectra; pdmp3_gc_side* tap_side; int tap_cap, tap_n;
};

/* ------------------------------------------------------------------------ */
/* input ring (P:1062-1086, P:1464-1474)                                     */
/* ------------------------------------------------------------------------ */
static inline unsigned ring_filled(const pdmp3_handle* id) {
  return (id->istart <= id->iend) ? (id->iend - id->istart) : (INBUF_SIZE - id->istart + id->iend);
}
/* the same two as the reference's caller sees them (the parser may have read ahead, see read_ahead) */
static inline unsigned ring_filled_logical(const pdmp3_handle* id) {
  return (id->l_istart <= id->iend) ? (id->iend - id->l_istart) : (INBUF_SIZE - id->l_istart + id->iend);
}
static inline unsigned ring_free_logical(const pdmp3_handle* id) {
  return (id->iend < id->l_istart) ? (id->l_istart - id->iend) : (INBUF_SIZE - id->iend + id->l_istart);
}
/* the parser stands where the reference stands (nothing read ahead): used by the paths that drive it directly */
static inline void sync_logical(pdmp3_handle* id) {
  id->l_istart = id->istart; id->l_processed = id->processed; id->l_hdr = id->hdr;
  if (!id->l_new_header && id->new_header) id->l_new_header = 1;
}
static inline unsigned ring_byte(pdmp3_handle* id) {
  if (id->istart == id->iend) { id->ring_short = 1; return BYTE_EOF; }
  unsigned v = id->vsrc ? id->vsrc[id->vfed - ring_filled(id)] : id->in[id->istart];
  id->istart++;
  if (id->istart == INBUF_SIZE) id->istart = 0;
  id->processed++;
  return v;
}

/* n bytes (n <= ring_filled) from the read index to dst; the read index moves past them */
static inline void ring_take(pdmp3_handle* id, uint8_t* dst, unsigned n) {
  if (id->vsrc) memcpy(dst, id->vsrc + id->vfed - ring_filled(id), n);
  else {
    unsigned first = INBUF_SIZE - id->istart;
    if (first > n) first = n;
    memcpy(dst, id->in + id->istart, first);
    memcpy(dst + first, id->in, n - first);
  }
  id->istart = (id->istart + n) % INBUF_SIZE;
  id->processed += n;
}

#define RESERVOIR_BYTES (2048 + 16)

static inline unsigned frame_bytes(const frame_header* H) {   /* P:1135-1138; the 42 quotients there are, computed once */
  if (H->ver) return lsf_frame_bytes(H->ver, kLsfBitrates[H->bitrate_index], sfreq9(H), H->padding);
  return g_frame_q[H->bitrate_index][H->sfreq] + H->padding;
}
static inline uint64_t side_word(const uint8_t* base, unsigned pos) {   /* >= 57 valid bits from bit `pos`, at the top */
  uint64_t w;
  memcpy(&w, base + (pos >> 3), 8);
  return __builtin_bswap64(w) << (pos & 7);
}
static inline void hp_pause(void) {
#if defined(__x86_64__)
  __builtin_ia32_pause();
#else
  sched_yield();
#endif
}

/* frame_parse.c */
HOST_LOCAL int search_header(pdmp3_handle* id);
HOST_LOCAL void read_side_info_bits(pdmp3_handle* id);
HOST_LOCAL int fill_reservoir(pdmp3_handle* id, unsigned size, unsigned begin);
HOST_LOCAL void decode_main(const uint8_t* reservoir, const frame_header* H, const side_info* S, main_out* out);
HOST_LOCAL void apply_main(pdmp3_handle* id, const frame_header* H, const main_out* out);
HOST_LOCAL int read_frame_staged(pdmp3_handle* id);
HOST_LOCAL int read_frame(pdmp3_handle* id, int16_t* spectra);
HOST_LOCAL void emit_records(pdmp3_handle* id, const frame_header* H, const side_info* S, int reset,
                         int16_t* spectra, pdmp3_gc_side* sd);
HOST_LOCAL size_t drain_frame(pdmp3_handle* id, unsigned char* out, size_t buflen);
/* stream_api.c: one engine per HIP device and process; $PDMP3_DEVICE */
HOST_LOCAL pdmp3_hip_ctx* shared_ctx_on(int dev);
HOST_LOCAL int default_device(void);
/* bulk.c: Get_Main_Data into the window's pool (device Huffman path) */
struct bulk;
HOST_LOCAL int fill_reservoir_pool(pdmp3_handle* id, unsigned size, unsigned begin);
/* cpus.c */
HOST_LOCAL int usable_cpus(void);

#endif

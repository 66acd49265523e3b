/*
 * pdmp3_host.c -- libpdmp3.so: the libmpg123-style streaming API of PDMP3
 * (include/pdmp3.h) over the MI355X transform engine (include/pdmp3_hip.h).
 *
 * Host stage (plain C, runs on the box's host cores, as BASELINE north_star
 * asks): input ring, header sync, side info, bit reservoir, scalefactors and a
 * table-driven Huffman decoder.  Each parsed frame becomes four granule-channel
 * records (pdmp3_gc_side + int16 spectra) written straight into the engine's
 * pinned staging buffers; pdmp3_read() parses as many frames as its output
 * buffer can take, decodes them as ONE batch on the GPU
 * (pdmp3_hip_stream_decode: hipMemcpyAsync H2D -> k_decode -> D2H) and then
 * hands the PCM out with the reference's partial-frame cursor semantics.
 *
 * Behavioural contract = the reference's (file:line cited at each function,
 * "P:n" = /root/reference/pdmp3.c line n), including the quirks SURVEY.md
 * lists as H1, H6, H7, H9, H10, H16-H18.  The code is written from that
 * contract, not from the reference's source: e.g. Huffman decoding is a
 * two-level lookup built from code books (tables_data.h) instead of the
 * reference's bit-serial tree walk.
 *
 * There is no CPU fallback for the transforms: without the engine library or
 * a HIP device pdmp3_new() fails.
 */
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

static huff_lut g_lut[PDMP3_NUM_HUFF_BOOKS];
static pthread_once_t g_lut_once = PTHREAD_ONCE_INIT;

static uint16_t g_frame_q[15][3];       /* frame sizes without padding (frame_bytes); filled here: handles are created on any thread */
static void build_luts(void) {
  for (unsigned b = 1; b < 15; b++)
    for (unsigned f = 0; f < 3; f++) g_frame_q[b][f] = (uint16_t)(144u * kBitratesL3[b] / kSampleRates[f]);
  for (int b = 0; b < PDMP3_NUM_HUFF_BOOKS; b++) {
    huff_lut* L = &g_lut[b];
    const pdmp3_hcode* codes = kHuffBooks[b];
    const int n = kHuffBookSize[b];
    int maxlen = 0;
    for (int i = 0; i < n; i++) if (codes[i].len > maxlen) maxlen = codes[i].len;
    L->sub_bits = maxlen > HL_BITS ? maxlen - HL_BITS : 0;
    L->quads = b == kHuffBookOfTable[32] || b == kHuffBookOfTable[33] || b == PDMP3_HUFF_BOOK_ISO33;
    int nsub = 0;
    memset(L->first, 0, sizeof L->first);
    for (int i = 0; i < n; i++) {
      if (codes[i].len > HL_BITS) {
        uint32_t prefix = codes[i].code >> (codes[i].len - HL_BITS);
        if (!(L->first[prefix] & 0x8000)) L->first[prefix] = (uint16_t)(0x8000 | nsub++);
      }
    }
    L->sub = nsub ? (uint16_t*)calloc((size_t)nsub << L->sub_bits, sizeof(uint16_t)) : NULL;
    for (int i = 0; i < n; i++) {
      const int len = codes[i].len;
      const uint16_t val = codes[i].err ? 0 : codes[i].val;
      if (len <= HL_BITS) {
        const uint32_t base = codes[i].code << (HL_BITS - len);
        for (uint32_t k = 0; k < (1u << (HL_BITS - len)); k++) L->first[base + k] = (uint16_t)(((len + leaf_nsign(L->quads, val)) << 8) | val);
      } else {
        const uint32_t prefix = codes[i].code >> (len - HL_BITS);
        const int si = L->first[prefix] & 0x7fff;
        const int extra = len - HL_BITS;
        const uint32_t rest = codes[i].code & ((1u << extra) - 1);
        const uint32_t base = rest << (L->sub_bits - extra);
        for (uint32_t k = 0; k < (1u << (L->sub_bits - extra)); k++)
          L->sub[((size_t)si << L->sub_bits) + base + k] = (uint16_t)(((len + leaf_nsign(L->quads, val)) << 8) | val);
      }
    }
  }
}

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

static pthread_mutex_t g_ctx_lock = PTHREAD_MUTEX_INITIALIZER;
#define MAX_DEVICES 16
static pdmp3_hip_ctx* g_ctx[MAX_DEVICES];

/* one engine (tables in HBM) per HIP device, shared by every handle / bulk decoder of the process on that device */
static pdmp3_hip_ctx* shared_ctx_on(int dev) {
  if (dev < 0 || dev >= MAX_DEVICES) return NULL;
  pthread_mutex_lock(&g_ctx_lock);
  /* Several decoders = several HIP streams whose kernels should overlap; HIP multiplexes streams onto 4 hardware
   * queues by default and kernels of one queue run one after the other (measured, 4 decoders on the C4 corpus:
   * 5.5 M frames/s with 4 queues, 7.1 M with 16).  Only a default, and only effective if HIP is not up yet. */
  setenv("GPU_MAX_HW_QUEUES", "16", 0);
  if (!g_ctx[dev] && pdmp3_hip_create(dev, &g_ctx[dev]) != PDMP3_HIP_OK) g_ctx[dev] = NULL;
  pdmp3_hip_ctx* c = g_ctx[dev];
  pthread_mutex_unlock(&g_ctx_lock);
  return c;
}

static int default_device(void) {
  const char* e = getenv("PDMP3_DEVICE");
  return e ? atoi(e) : 0;
}

static pdmp3_hip_ctx* shared_ctx(void) { return shared_ctx_on(default_device()); }

/* P:2351: pdmp3_new(decoder, error) -- `decoder` is ignored like in the reference */
pdmp3_handle* pdmp3_new(const char* decoder, int* error) {
  (void)decoder;
  pthread_once(&g_lut_once, build_luts);
  pdmp3_handle* id = (pdmp3_handle*)calloc(1, sizeof *id);
  if (!id) { if (error) *error = PDMP3_ERR; return NULL; }
  pdmp3_hip_ctx* ctx = shared_ctx();
  if (!ctx || pdmp3_hip_stream_create(ctx, BATCH_MAX, &id->hs) != PDMP3_HIP_OK) {
    fprintf(stderr, "pdmp3: no MI355X transform engine: %s\n", pdmp3_hip_last_error());
    free(id);
    if (error) *error = PDMP3_ERR;
    return NULL;
  }
  if (error) *error = PDMP3_OK;
  return id;
}

/* Test hook (host-logic tests on machines without a GPU): a handle that can
 * parse and tap records but not decode.  Not part of the reference API. */
pdmp3_handle* pdmp3_amd_new_parse_only(void) {
  pthread_once(&g_lut_once, build_luts);
  pdmp3_handle* id = (pdmp3_handle*)calloc(1, sizeof *id);
  if (id) id->host_only = 1;
  return id;
}

void pdmp3_amd_set_tap(pdmp3_handle* id, int16_t* spectra, pdmp3_gc_side* side, int cap_frames) {
  id->tap_spectra = spectra; id->tap_side = side; id->tap_cap = cap_frames; id->tap_n = 0;
}
int pdmp3_amd_tap_count(const pdmp3_handle* id) { return id->tap_n; }

static int ra_rollback(pdmp3_handle* id);      /* read-ahead of pdmp3_read, below */

/* P:2360 */
void pdmp3_delete(pdmp3_handle* id) {
  if (!id) return;
  if (id->ra_inflight && id->hs) (void)pdmp3_hip_stream_wait(id->hs, 0);
  if (id->hs) pdmp3_hip_stream_destroy(id->hs);
  free(id);
}

/* P:2369-2384 */
int pdmp3_open_feed(pdmp3_handle* id) {
  if (!id) return PDMP3_ERR;
  if (ra_rollback(id) != PDMP3_OK) return PDMP3_ERR;   /* (what survives open_feed is the parse state of the reference's position) */
  id->ostart = 0; id->istart = 0; id->iend = 0; id->processed = 0; id->new_header = 0;
  id->l_istart = 0; id->l_processed = 0; id->l_new_header = 0;
  id->need_reset = 1;
  id->main_top = 0;
  return PDMP3_OK;
}

/* ------------------------------------------------------------------------ */
/* input ring (P:1062-1086, P:1464-1474)                                     */
/* ------------------------------------------------------------------------ */
static unsigned ring_filled(const pdmp3_handle* id) {
  return (id->istart <= id->iend) ? (id->iend - id->istart) : (INBUF_SIZE - id->istart + id->iend);
}
/* the same two as the reference's caller sees them (the parser may have read ahead, see read_ahead) */
static unsigned ring_filled_logical(const pdmp3_handle* id) {
  return (id->l_istart <= id->iend) ? (id->iend - id->l_istart) : (INBUF_SIZE - id->l_istart + id->iend);
}
static unsigned ring_free_logical(const pdmp3_handle* id) {
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

/* P:2391-2423: all-or-nothing copy into the ring */
int pdmp3_feed(pdmp3_handle* id, const unsigned char* in, size_t size) {
  if (!(id && in && size)) return PDMP3_ERR;
  if (size > (size_t)ring_free_logical(id)) return PDMP3_NO_SPACE;
  size_t first;
  const int real = id->vsrc == NULL;             /* (virtual ring: the same index arithmetic, no bytes moved) */
  if (!real) id->vfed += size;
  if (id->iend < id->l_istart) {
    first = id->l_istart - id->iend;
    if (size < first) first = size;
    if (real) memcpy(id->in + id->iend, in, first);
    id->iend += (unsigned)first;
  } else {
    first = INBUF_SIZE - id->iend;
    if (size < first) first = size;
    if (first) { if (real) memcpy(id->in + id->iend, in, first); id->iend += (unsigned)first; size -= first; }
    if (size) { if (real) memcpy(id->in, in + first, size); id->iend = (unsigned)size; }
  }
  /* a feed that fills the ring exactly leaves iend == istart, which the reference reads as EMPTY (P:1062-1068): it
   * will not get to the frames read ahead before its next feeds have overwritten them */
  if (id->ra_head != id->ra_n && id->iend == id->l_istart && ra_rollback(id) != PDMP3_OK) return PDMP3_ERR;
  return PDMP3_OK;
}

/* ------------------------------------------------------------------------ */
/* header sync (P:1252-1340)                                                 */
/* ------------------------------------------------------------------------ */
static int read_header(pdmp3_handle* id) {
  unsigned b[4];
  if (id->vsrc && ring_filled(id) >= 4) {         /* (virtual ring: the four bytes lie in a row) */
    uint8_t q[4];
    ring_take(id, q, 4);
    b[0] = q[0]; b[1] = q[1]; b[2] = q[2]; b[3] = q[3];
  } else
  for (int i = 0; i < 4; i++) b[i] = ring_byte(id);
  if (b[0] == BYTE_EOF || b[1] == BYTE_EOF || b[2] == BYTE_EOF || b[3] == BYTE_EOF) return PDMP3_ERR;
  uint32_t h = (b[0] << 24) | (b[1] << 16) | (b[2] << 8) | b[3];
  /* PDMP3_ISO_LSF (not the reference): eleven sync bits, so that MPEG-2.5's 0xFFE + ID 0 is a header too; never in bits
   * mode (the device's Huffman stage reads MPEG-1 side info only: include/pdmp3_bulk.h) */
  const int lsf_ok = (id->iso & PDMP3_ISO_LSF) && !id->side_to_bits;
  const uint32_t sync = lsf_ok ? 0xffe00000u : 0xfff00000u;
  while ((h & sync) != sync) {                   /* byte-aligned 12-bit sync */
    unsigned nb = ring_byte(id);
    if (nb == BYTE_EOF) return PDMP3_ERR;
    h = (h << 8) | nb;
  }
  frame_header* H = &id->hdr;
  H->ver = 0;
  if (lsf_ok) {
    const unsigned v = (h >> 19) & 3;              /* 11 MPEG-1, 10 MPEG-2 LSF, 00 MPEG-2.5, 01 reserved */
    if (v == 1) { H->layer = 0; return PDMP3_ERR; }
    H->ver = v == 3 ? 0 : v == 2 ? 1 : 2;
  }
  H->id = (h >> 19) & 1; H->layer = (h >> 17) & 3; H->protection = (h >> 16) & 1;
  H->bitrate_index = (h >> 12) & 15; H->sfreq = (h >> 10) & 3; H->padding = (h >> 9) & 1;
  H->mode = (h >> 6) & 3; H->mode_ext = (h >> 4) & 3;
  /* MPEG-1 only; free format, index 15, sfreq 3 and layer 0 rejected (P:1293-1315) */
  if ((H->id != 1 && !H->ver) || H->bitrate_index == 0 || H->bitrate_index == 15 || H->sfreq == 3 || H->layer == 0)
    return PDMP3_ERR;
  H->layer = 4 - H->layer;
  if (!id->new_header) id->new_header = 1;
  return PDMP3_OK;
}

/* P:1322-1340: retry from the next byte after the mark; give up after 1152 tries */
static int search_header(pdmp3_handle* id) {
  const size_t pos = id->processed;
  unsigned mark = id->istart;
  int res = PDMP3_NEED_MORE, tries = 0;
  while (ring_filled(id) > 4) {
    res = read_header(id);
    if (id->hdr.layer == 3 && (res == PDMP3_OK || res == PDMP3_NEW_FORMAT)) break;
    if (++mark == INBUF_SIZE) mark = 0;
    id->istart = mark;
    id->processed = pos;
    if (++tries > 1152) return PDMP3_ERR;
  }
  if (!(id->hdr.layer == 3 && (res == PDMP3_OK || res == PDMP3_NEW_FORMAT))) id->ring_short = 1;   /* left by the fill test */
  return res;
}

/* ------------------------------------------------------------------------ */
/* side info (P:1129-1200)                                                   */
/* ------------------------------------------------------------------------ */
/* bit cursor over side_vec, kept in registers while one frame's side info is parsed (pos in bits) */
typedef struct { const uint8_t* base; unsigned pos; } side_cur;
static inline unsigned side_bits(side_cur* c, unsigned n) {          /* n <= 12 */
  uint64_t w;
  memcpy(&w, c->base + ((c->pos >> 3) & 63), 8);
  w = __builtin_bswap64(w) << (c->pos & 7);
  c->pos += n;
  return (unsigned)(w >> (64 - n));
}

static unsigned frame_bytes(const frame_header* H) {   /* P:1135-1138; the 42 quotients there are, computed once */
  if (H->ver) return lsf_frame_bytes(H->ver, kLsfBitrates[H->bitrate_index], sfreq9(H), H->padding);
  return g_frame_q[H->bitrate_index][H->sfreq] + H->padding;
}
static inline unsigned side_info_bytes(const frame_header* H) {
  const unsigned nch = H->mode == 3 ? 1 : 2;
  return H->ver ? (nch == 1 ? 9 : 17) : (nch == 1 ? 17 : 32);
}

/* 13818-3 2.4.1.7 (not the reference): ONE granule; main_data_begin 8 bits, 1 / 2 private bits, no scfsi; a 9-bit
 * scalefac_compress, no preflag bit (scalefac_compress >= 500 implies it, except for the right channel of an
 * intensity-stereo frame, whose scalefac_compress is a different code: lsf_tables.h) */
static void read_side_info_lsf(pdmp3_handle* id) {
  const unsigned nch = id->hdr.mode == 3 ? 1 : 2, nbytes = side_info_bytes(&id->hdr);
  unsigned got = ring_filled(id);
  if (got > nbytes) got = nbytes;
  else if (got < nbytes) id->ring_short = 1;
  ring_take(id, id->side_vec, got);
  if (got == nbytes) { id->side_ptr = 0; id->side_idx = 0; }
  side_info* S = &id->si;
  side_cur sc = {id->side_vec, id->side_ptr * 8 + id->side_idx};
  S->main_data_begin = side_bits(&sc, 8);
  (void)side_bits(&sc, nch == 1 ? 1 : 2);
  for (unsigned ch = 0; ch < nch; ch++) {
    for (unsigned b = 0; b < 4; b++) S->scfsi[ch][b] = 0;
    S->part2_3_length[0][ch] = side_bits(&sc, 12);
    S->big_values[0][ch] = side_bits(&sc, 9);
    S->global_gain[0][ch] = side_bits(&sc, 8);
    S->scalefac_compress[0][ch] = side_bits(&sc, 9);
    S->win_switch[0][ch] = side_bits(&sc, 1);
    if (S->win_switch[0][ch]) {
      S->block_type[0][ch] = side_bits(&sc, 2);
      S->mixed[0][ch] = side_bits(&sc, 1);
      S->table_select[0][ch][0] = side_bits(&sc, 5);
      S->table_select[0][ch][1] = side_bits(&sc, 5);
      for (unsigned w = 0; w < 3; w++) S->subblock_gain[0][ch][w] = side_bits(&sc, 3);
      S->region0_count[0][ch] = (S->block_type[0][ch] == 2 && !S->mixed[0][ch]) ? 8 : 7;
      S->region1_count[0][ch] = 20 - S->region0_count[0][ch];
    } else {
      for (unsigned r = 0; r < 3; r++) S->table_select[0][ch][r] = side_bits(&sc, 5);
      S->region0_count[0][ch] = side_bits(&sc, 4);
      S->region1_count[0][ch] = side_bits(&sc, 3);
      S->block_type[0][ch] = 0;
      S->mixed[0][ch] = 0;
    }
    const int is_right = id->hdr.mode == 1 && (id->hdr.mode_ext & 1) && ch == 1;
    S->preflag[0][ch] = !is_right && S->scalefac_compress[0][ch] >= 500;
    S->scalefac_scale[0][ch] = side_bits(&sc, 1);
    S->count1table_select[0][ch] = side_bits(&sc, 1) ? 2 : 0;       /* 2: the standard's table B (as with PDMP3_ISO_TABLE33) */
  }
  id->side_ptr = sc.pos >> 3;
  id->side_idx = sc.pos & 7;
}

static void read_side_info(pdmp3_handle* id) {
  if (id->hdr.ver) { read_side_info_lsf(id); return; }
  const unsigned nch = id->hdr.mode == 3 ? 1 : 2, nbytes = nch == 1 ? 17 : 32;
  unsigned got = ring_filled(id);
  if (got > nbytes) got = nbytes;
  else if (got < nbytes) id->ring_short = 1;
  ring_take(id, id->side_vec, got);
  if (got == nbytes) { id->side_ptr = 0; id->side_idx = 0; }   /* pointers move only on a full read (P:1576-1586) */
  side_info* S = &id->si;
  side_cur sc = {id->side_vec, id->side_ptr * 8 + id->side_idx};
  S->main_data_begin = side_bits(&sc, 9);
  (void)side_bits(&sc, nch == 1 ? 5 : 3);
  for (unsigned ch = 0; ch < nch; ch++)
    for (unsigned b = 0; b < 4; b++) S->scfsi[ch][b] = side_bits(&sc, 1);
  for (unsigned gr = 0; gr < 2; gr++)
    for (unsigned ch = 0; ch < nch; ch++) {
      S->part2_3_length[gr][ch] = side_bits(&sc, 12);
      S->big_values[gr][ch] = side_bits(&sc, 9);
      S->global_gain[gr][ch] = side_bits(&sc, 8);
      S->scalefac_compress[gr][ch] = side_bits(&sc, 4);
      S->win_switch[gr][ch] = side_bits(&sc, 1);
      if (S->win_switch[gr][ch]) {
        S->block_type[gr][ch] = side_bits(&sc, 2);
        S->mixed[gr][ch] = side_bits(&sc, 1);
        S->table_select[gr][ch][0] = side_bits(&sc, 5);
        S->table_select[gr][ch][1] = side_bits(&sc, 5);
        for (unsigned w = 0; w < 3; w++) S->subblock_gain[gr][ch][w] = side_bits(&sc, 3);
        S->region0_count[gr][ch] = (S->block_type[gr][ch] == 2 && !S->mixed[gr][ch]) ? 8 : 7;   /* implicit */
        S->region1_count[gr][ch] = 20 - S->region0_count[gr][ch];
      } else {
        for (unsigned r = 0; r < 3; r++) S->table_select[gr][ch][r] = side_bits(&sc, 5);
        S->region0_count[gr][ch] = side_bits(&sc, 4);
        S->region1_count[gr][ch] = side_bits(&sc, 3);
        S->block_type[gr][ch] = 0;             /* mixed / subblock_gain stay stale (H20) */
      }
      S->preflag[gr][ch] = side_bits(&sc, 1);
      S->scalefac_scale[gr][ch] = side_bits(&sc, 1);
      S->count1table_select[gr][ch] = side_bits(&sc, 1);
      if ((id->iso & PDMP3_ISO_TABLE33) && S->count1table_select[gr][ch]) S->count1table_select[gr][ch] = 2;
    }
  id->side_ptr = sc.pos >> 3;
  id->side_idx = sc.pos & 7;
}

/* The same parse (P:1129-1200) for a frame whose side info the ring holds completely, written as the engine's
 * pdmp3_frame_bits in one go: the whole-stream decoder's scan is one host thread, and field-by-field parsing into
 * side_info plus the repacking (fill_frame_bits) was two thirds of its time per frame.  A granule-channel is 59 bits:
 * 34 fixed, 22 that depend on window_switching, 3 flags.  What the reference leaves stale from earlier frames is kept
 * in `si` exactly as read_side_info keeps it (H20: table_select[2] and subblock_gain are not written by every frame),
 * so frames parsed by either function can follow each other. */
static inline uint64_t side_word(const uint8_t* base, unsigned pos) {   /* >= 57 valid bits from bit `pos`, at the top */
  uint64_t w;
  memcpy(&w, base + (pos >> 3), 8);
  return __builtin_bswap64(w) << (pos & 7);
}
static void read_side_info_bits(pdmp3_handle* id) {
  static const uint8_t rev4[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
  const unsigned nch = id->hdr.mode == 3 ? 1 : 2, nbytes = nch == 1 ? 17 : 32;
  ring_take(id, id->side_vec, nbytes);
  id->side_ptr = 0; id->side_idx = 0;
  side_info* S = &id->si;
  pdmp3_frame_bits* fb = &id->fb_cur;
  const uint8_t* v = id->side_vec;
  memset(fb, 0, sizeof *fb);
  const uint64_t head = side_word(v, 0);
  S->main_data_begin = (unsigned)(head >> 55);
  unsigned pos;
  if (nch == 1) { fb->scfsi[0] = rev4[(head >> 46) & 15]; pos = 18; }
  else { fb->scfsi[0] = rev4[(head >> 48) & 15]; fb->scfsi[1] = rev4[(head >> 44) & 15]; pos = 20; }
  for (unsigned gr = 0; gr < 2; gr++)
    for (unsigned ch = 0; ch < nch; ch++, pos += 59) {
      const uint64_t x = side_word(v, pos);
      const unsigned tail = (unsigned)(side_word(v, pos + 56) >> 61);     /* preflag, scalefac_scale, count1table_select */
      pdmp3_gc_bits* g = &fb->gc[gr * 2 + ch];
      g->part2_3_length = (uint16_t)(x >> 52);
      g->big_values = (uint16_t)((x >> 43) & 0x1ff);
      g->global_gain = (uint8_t)(x >> 35);
      g->scalefac_compress = (uint8_t)((x >> 31) & 15);
      const unsigned ws = (unsigned)(x >> 30) & 1, y = (unsigned)(x >> 8) & 0x3fffff;
      unsigned flags = ((tail & 2) ? PDMP3_GC_SCALEFAC_SCALE : 0) | ((tail & 4) ? PDMP3_GC_PREFLAG : 0);
      if (ws) {
        const unsigned bt = y >> 20, mixed = (y >> 19) & 1;
        flags |= PDMP3_GC_WIN_SWITCH | (bt << PDMP3_GC_BLOCK_TYPE_SHIFT) | (mixed ? PDMP3_GC_MIXED : 0);
        S->mixed[gr][ch] = mixed;
        g->table_select[0] = (uint8_t)((y >> 14) & 31);
        g->table_select[1] = (uint8_t)((y >> 9) & 31);
        g->table_select[2] = (uint8_t)S->table_select[gr][ch][2];          /* stale */
        for (unsigned w = 0; w < 3; w++) g->subblock_gain[w] = (uint8_t)(S->subblock_gain[gr][ch][w] = (y >> (6 - 3 * w)) & 7);
        g->region0_count = (bt == 2 && !mixed) ? 8 : 7;
        g->region1_count = (uint8_t)(20 - g->region0_count);
      } else {
        g->table_select[0] = (uint8_t)(y >> 17);
        g->table_select[1] = (uint8_t)((y >> 12) & 31);
        g->table_select[2] = (uint8_t)(S->table_select[gr][ch][2] = (y >> 7) & 31);
        for (unsigned w = 0; w < 3; w++) g->subblock_gain[w] = (uint8_t)S->subblock_gain[gr][ch][w];   /* stale */
        g->region0_count = (uint8_t)((y >> 3) & 15);
        g->region1_count = (uint8_t)(y & 7);
      }
      g->flags = (uint8_t)flags;
      g->count1table_select = (uint8_t)(((tail & 1) && (id->iso & PDMP3_ISO_TABLE33)) ? 2 : (tail & 1));
    }
  fb->iso = (uint8_t)id->iso;
  id->side_ptr = pos >> 3;
  id->side_idx = pos & 7;
  id->fb_valid = 1;
}

/* ------------------------------------------------------------------------ */
/* bit reservoir (P:1096-1122) and main-data bit reader                      */
/* ------------------------------------------------------------------------ */
static int fill_reservoir(pdmp3_handle* id, unsigned size, unsigned begin) {
  uint8_t* dst;
  int ok = begin <= id->main_top;
  if (ok) {
    memmove(id->main_vec, id->main_vec + id->main_top - begin, begin);
    dst = id->main_vec + begin;
    id->main_top = begin + size;
  } else {            /* not enough history: keep the bytes for later frames, skip this one (H9) */
    dst = id->main_vec + id->main_top;
    id->main_top += size;
  }
  /* as many of `size` bytes as the ring holds and main_vec has room for; a short read is ignored (H18) */
  const size_t off = (size_t)(dst - id->main_vec);
  unsigned n = off >= sizeof id->main_vec ? 0 : (unsigned)(sizeof id->main_vec - off);
  if (n > size) n = size;
  if (n > ring_filled(id)) { n = ring_filled(id); id->ring_short = 1; }
  ring_take(id, dst, n);
  return ok ? PDMP3_OK : PDMP3_NEED_MORE;
}

/* ------------------------------------------------------------------------ */
/* main data of ONE frame: scalefactors + Huffman, as a pure function of the */
/* reservoir bytes, the header and the side info.  This is the part of the   */
/* host stage that is independent from frame to frame: pdmp3_read runs it    */
/* inline, the bulk entry point fans it out over host threads.               */
/* ------------------------------------------------------------------------ */
#define RESERVOIR_BYTES (2048 + 16)

typedef struct {
  const uint8_t* buf;       /* RESERVOIR_BYTES readable */
  unsigned bitpos;
} bitreader;

static inline uint32_t peek32(const bitreader* b) {   /* next 25+ valid bits, MSB first */
  const unsigned byte = b->bitpos >> 3;
  const uint8_t* p = b->buf + (byte < RESERVOIR_BYTES - 5 ? byte : RESERVOIR_BYTES - 5);
  uint64_t w = ((uint64_t)p[0] << 32) | ((uint64_t)p[1] << 24) | ((uint64_t)p[2] << 16) | ((uint64_t)p[3] << 8) | p[4];
  return (uint32_t)(w >> (8 - (b->bitpos & 7)));
}
static inline unsigned get_bits(bitreader* b, unsigned n) {
  if (!n) return 0;
  unsigned v = peek32(b) >> (32 - n);
  b->bitpos += n;
  return v;
}

/* one code word of `book`: returns the leaf value (x<<4 | y) */
static inline unsigned huff_symbol(bitreader* b, int book) {
  const huff_lut* L = &g_lut[book];
  const uint32_t w = peek32(b);
  unsigned e = L->first[w >> (32 - HL_BITS)];
  if (e & 0x8000) {
    const unsigned rest = (w << HL_BITS) >> (32 - L->sub_bits);
    e = L->sub[((size_t)(e & 0x7fff) << L->sub_bits) + rest];
  }
  b->bitpos += (e >> 8) - leaf_nsign(L->quads, e & 0xff);     /* the code word alone: the caller reads the signs */
  return e & 0xff;
}

/* one big_values pair, field by field (only used within 8 bytes of the end of the reservoir buffer) */
static void pair_slow(bitreader* b, int book, unsigned linbits, int* px, int* py) {
  const unsigned leaf = huff_symbol(b, book);
  int x = leaf >> 4, y = leaf & 15;
  if (linbits && x == 15) x += (int)get_bits(b, linbits);
  if (x > 0 && get_bits(b, 1)) x = -x;
  if (linbits && y == 15) y += (int)get_bits(b, linbits);
  if (y > 0 && get_bits(b, 1)) y = -y;
  *px = x; *py = y;
}

#define FAST_LIMIT ((RESERVOIR_BYTES - 8) * 8u)    /* bit positions from which one 8-byte load is in bounds */
static inline uint64_t peek64(const bitreader* b) {           /* >= 57 valid bits, MSB first */
  uint64_t w;
  memcpy(&w, b->buf + (b->bitpos >> 3), 8);
  return __builtin_bswap64(w) << (b->bitpos & 7);
}

/* pairs [pos, end) of one region.  A pair is at most 19 + 2 * (13 + 1) = 47 bits: one window per pair.
 * The body is branch-free apart from the second-level lookup: on dense material "is x zero", "is it negative" are
 * coin flips, and three mispredicted branches per pair were most of this loop's time (7 us per 320 kbps frame).
 * `lin` is a compile-time flag: the tables without linbits (1-15) get a loop without the linbits arithmetic. */
static inline __attribute__((always_inline)) unsigned decode_pairs_body(bitreader* b, const huff_lut* L, int book, unsigned linbits,
                                                                        const int lin, unsigned pos, unsigned end, int16_t* is) {
  for (; pos < end; pos += 2) {
    int x, y;
    if (__builtin_expect(b->bitpos <= FAST_LIMIT, 1)) {
      const uint64_t w = peek64(b);
      unsigned e = L->first[w >> (64 - HL_BITS)];
      if (__builtin_expect(e & 0x8000, 0)) {
        const unsigned rest = (unsigned)((w << HL_BITS) >> (64 - L->sub_bits));
        e = L->sub[((size_t)(e & 0x7fff) << L->sub_bits) + rest];
      }
      if (!lin) b->bitpos += e >> 8;               /* (the next pair's window does not wait for the values) */
      x = (e >> 4) & 15; y = e & 15;
      uint64_t v = w << ((e >> 8) - (x != 0) - (y != 0));   /* what follows the code word: <= 28 bits are looked at */
      unsigned lx = 0, ly = 0;
      if (lin) {                                   /* ((v >> 1) >> (63 - n)) == v >> (64 - n) for n = 1..63 and 0 for n = 0 */
        lx = x == 15 ? linbits : 0;
        x += (int)((v >> 1) >> (63 - lx));
        v <<= lx;
      }
      const unsigned nzx = x != 0;
      const int sx = (int)(v >> 63) & (int)nzx;    /* a sign bit follows a value != 0 */
      x = (x ^ -sx) + sx;
      v <<= nzx;
      if (lin) {
        ly = y == 15 ? linbits : 0;
        y += (int)((v >> 1) >> (63 - ly));
        v <<= ly;
      }
      const unsigned nzy = y != 0;
      const int sy = (int)(v >> 63) & (int)nzy;
      y = (y ^ -sy) + sy;
      if (lin) b->bitpos += (e >> 8) + lx + ly;
    } else pair_slow(b, book, linbits, &x, &y);
    if (pos < 576) is[pos] = (int16_t)x;           /* big_values > 288 is not checked by the reference (H8) */
    if (pos + 1 < 576) is[pos + 1] = (int16_t)y;
  }
  return pos;
}

static unsigned decode_pairs(bitreader* b, unsigned tn, unsigned pos, unsigned end, int16_t* is) {
  const int book = kHuffBookOfTable[tn];
  if (book < 0) {                                  /* table 0 (and the unused 4, 14): no bits, zeros */
    for (; pos < end; pos += 2) {
      if (pos < 576) is[pos] = 0;
      if (pos + 1 < 576) is[pos + 1] = 0;
    }
    return pos;
  }
  const unsigned linbits = kHuffLinbits[tn];
  return linbits ? decode_pairs_body(b, &g_lut[book], book, linbits, 1, pos, end, is)
                 : decode_pairs_body(b, &g_lut[book], book, 0, 0, pos, end, is);
}

/* P:2051-2115 */
static void decode_huffman(bitreader* b, const frame_header* H, const side_info* S, unsigned part2_start,
                           unsigned gr, unsigned ch, main_out* out) {
  int16_t* is = out->is + (gr * 2 + ch) * 576;
  if (S->part2_3_length[gr][ch] == 0) {           /* all zero; count1 keeps its old value (H6) */
    memset(is, 0, 576 * sizeof *is);
    out->count1_set[gr][ch] = 0;
    if (H->ver) { out->count1[gr][ch] = 0; out->count1_set[gr][ch] = 1; }      /* (LSF: nothing of the reference's to reproduce) */
    return;
  }
  const unsigned end = part2_start + S->part2_3_length[gr][ch] - 1;   /* last bit of this part */
  /* Every line is defined: the ones neither a pair nor a quad writes are zero.  (They are the rzero region, which
   * the reference zeroes too -- except when its line counter wraps below zero on a corrupt part2_3_length, P:2106:
   * then it requantises the FLOATS the previous frame's synthesis left in is[], which no int16 record can carry.
   * Host and device Huffman both give zeros there.) */
  memset(is, 0, 576 * sizeof *is);
  unsigned r1, r2;
  if (H->ver) {
    /* LSF: the same rule over the LSF band tables; nothing lies beyond band 22; at 8 kHz three short bands are 72 lines */
    const uint16_t* l = kLsfSfbLong[sfreq9(H) - 3];
    if (S->win_switch[gr][ch] && S->block_type[gr][ch] == 2) { r1 = sfreq9(H) == 8 ? 72 : 36; r2 = 576; }
    else {
      const unsigned i1 = S->region0_count[gr][ch] + 1, i2 = S->region0_count[gr][ch] + S->region1_count[gr][ch] + 2;
      r1 = l[i1 > 22 ? 22 : i1];
      r2 = l[i2 > 22 ? 22 : i2];
    }
  } else
  if (S->win_switch[gr][ch] && S->block_type[gr][ch] == 2) { r1 = 36; r2 = 576; }
  else {
    /* l[23] s[14] are contiguous in the reference: indices 23, 24 read s[0], s[1] (H7) */
    const uint16_t* l = H->sfreq == 0 ? kSfbLong0 : H->sfreq == 1 ? kSfbLong1 : kSfbLong2;
    const uint16_t* s = H->sfreq == 0 ? kSfbShort0 : H->sfreq == 1 ? kSfbShort1 : kSfbShort2;
    const unsigned i1 = S->region0_count[gr][ch] + 1, i2 = S->region0_count[gr][ch] + S->region1_count[gr][ch] + 2;
    r1 = i1 < 23 ? l[i1] : s[i1 - 23];
    r2 = i2 < 23 ? l[i2] : s[i2 - 23];
  }
  /* the pair at (even) pos takes table 0 while pos < r1, table 1 while pos < r2, else table 2 */
  const unsigned nbig = S->big_values[gr][ch] * 2;
  unsigned e0 = (r1 + 1) & ~1u, e1 = (r2 + 1) & ~1u;
  if (e0 > nbig) e0 = nbig;
  if (e1 > nbig) e1 = nbig;
  if (e1 < e0) e1 = e0;
  unsigned pos = decode_pairs(b, S->table_select[gr][ch][0], 0, e0, is);
  pos = decode_pairs(b, S->table_select[gr][ch][1], pos, e1, is);
  pos = decode_pairs(b, S->table_select[gr][ch][2], pos, nbig, is);
  /* count1 region: table 32, or the reference's mis-pointed table 33 (H1).  The loop runs while a whole quad
   * fits below 576, so the reference's mid-quad bound check can never fire. */
  /* (count1table_select = 2: PDMP3_ISO_TABLE33 was set when the side info was read -- the standard's table B) */
  const int qbook = S->count1table_select[gr][ch] == 2 ? PDMP3_HUFF_BOOK_ISO33 : kHuffBookOfTable[32 + S->count1table_select[gr][ch]];
  const huff_lut* Q = &g_lut[qbook];
  while (pos <= 572 && b->bitpos <= end) {
    unsigned leaf;
    int q[4];
    if (__builtin_expect(b->bitpos <= FAST_LIMIT && Q->sub_bits == 0, 1)) {
      const uint64_t w = peek64(b);
      const unsigned e = Q->first[w >> (64 - HL_BITS)];
      b->bitpos += e >> 8;
      leaf = e & 0xff;
      uint64_t v = w << ((e >> 8) - leaf_nsign(1, leaf));
      for (int k = 0; k < 4; k++) {                /* v w x y, branch-free like the pairs */
        const unsigned nz = (leaf >> (3 - k)) & 1;
        const int sg = (int)(v >> 63) & (int)nz;
        q[k] = ((int)nz ^ -sg) + sg;
        v <<= nz;
      }
    } else {
      leaf = huff_symbol(b, qbook);
      for (int k = 0; k < 4; k++) {
        q[k] = (int)(leaf >> (3 - k)) & 1;
        if (q[k] && get_bits(b, 1)) q[k] = -1;
      }
    }
    is[pos] = (int16_t)q[0]; is[pos + 1] = (int16_t)q[1]; is[pos + 2] = (int16_t)q[2]; is[pos + 3] = (int16_t)q[3];
    pos += 4;
  }
  if (b->bitpos > end + 1) pos -= 4;               /* overshoot: drop the last quad */
  if (pos > 576) pos = 576;                        /* (unsigned wrap of the reference on pos < 4: corrupt input) */
  out->count1[gr][ch] = (uint16_t)pos;
  out->count1_set[gr][ch] = 1;
  if (pos < 576) memset(is + pos, 0, (576 - pos) * sizeof *is);
  b->bitpos = end + 1;
}

/* P:1376-1437: scalefactors, then Huffman, for every granule / channel of the frame */
static void decode_main(const uint8_t* reservoir, const frame_header* H, const side_info* S, main_out* out) {
  const unsigned nch = H->mode == 3 ? 1 : 2;
  bitreader b = {reservoir, 0};
  memset(out->sf_l_set, 0, sizeof out->sf_l_set);
  memset(out->sf_s_set, 0, sizeof out->sf_s_set);
  out->sf_l_copy[0] = out->sf_l_copy[1] = 0;
  if (H->ver) {
    /* 13818-3 2.4.3.2: scalefac_compress -> four slen and, by block shape, four partition sizes (lsf_tables.h); the
     * scalefactors come in band order (short: band by band, window by window; mixed: 6 long bands, then short bands
     * 3..11); what is not transmitted is 0.  Every scalefactor of the granule is (re)written: nothing is carried. */
    for (unsigned ch = 0; ch < nch; ch++) {
      const unsigned part2_start = b.bitpos;
      uint8_t slen[4];
      int pf;
      const int cls = lsf_slen_of(S->scalefac_compress[0][ch], H->mode == 1 && (H->mode_ext & 1) && ch == 1, slen, &pf);
      const int shortb = S->win_switch[0][ch] && S->block_type[0][ch] == 2, mixed = shortb && S->mixed[0][ch];
      const uint8_t* nsf = kLsfNsfb[cls][shortb ? (mixed ? 2 : 1) : 0];
      uint8_t vals[40];
      unsigned n = 0;
      for (unsigned k = 0; k < 4; k++)
        for (unsigned i = 0; i < nsf[k]; i++) vals[n++] = (uint8_t)get_bits(&b, slen[k]);
      for (; n < 40; n++) vals[n] = 0;
      memset(out->sf_l[0][ch], 0, sizeof out->sf_l[0][ch]);
      memset(out->sf_s[0][ch], 0, sizeof out->sf_s[0][ch]);
      if (!shortb) memcpy(out->sf_l[0][ch], vals, 21);
      else if (!mixed) memcpy(out->sf_s[0][ch], vals, 36);
      else { memcpy(out->sf_l[0][ch], vals, 6); memcpy(out->sf_s[0][ch][3], vals + 6, 27); }
      out->sf_l_set[0][ch] = 0x1fffff;
      out->sf_s_set[0][ch] = 0xfff;
      decode_huffman(&b, H, S, part2_start, 0, ch, out);
    }
    return;
  }
  for (unsigned gr = 0; gr < 2; gr++)
    for (unsigned ch = 0; ch < nch; ch++) {
      const unsigned part2_start = b.bitpos;
      const unsigned slen1 = kSlen[S->scalefac_compress[gr][ch] * 2], slen2 = kSlen[S->scalefac_compress[gr][ch] * 2 + 1];
      if (S->win_switch[gr][ch] && S->block_type[gr][ch] == 2) {
        unsigned first_short = 0;
        if (S->mixed[gr][ch]) {
          for (unsigned sfb = 0; sfb < 8; sfb++) out->sf_l[gr][ch][sfb] = (uint8_t)get_bits(&b, slen1);
          out->sf_l_set[gr][ch] |= 0xffu;
          first_short = 3;
        }
        for (unsigned sfb = first_short; sfb < 12; sfb++) {
          for (unsigned w = 0; w < 3; w++) out->sf_s[gr][ch][sfb][w] = (uint8_t)get_bits(&b, sfb < 6 ? slen1 : slen2);
          out->sf_s_set[gr][ch] |= (uint16_t)(1u << sfb);
        }
      } else {
        static const uint8_t lo[5] = {0, 6, 11, 16, 21};
        for (unsigned g4 = 0; g4 < 4; g4++) {
          const unsigned nb = g4 < 2 ? slen1 : slen2;
          if (gr == 1 && S->scfsi[ch][g4]) {       /* reuse granule 0's factors (whatever they are by then) */
            out->sf_l_copy[ch] |= (uint8_t)(1u << g4);
          } else {
            for (unsigned sfb = lo[g4]; sfb < lo[g4 + 1]; sfb++) {
              out->sf_l[gr][ch][sfb] = (uint8_t)get_bits(&b, nb);
              out->sf_l_set[gr][ch] |= 1u << sfb;
            }
          }
        }
      }
      decode_huffman(&b, H, S, part2_start, gr, ch, out);
    }
}

/* merge one frame's main data into the state that survives frames (scalefactors, count1, is) */
static void apply_main(pdmp3_handle* id, const frame_header* H, const main_out* out) {
  static const uint8_t lo[5] = {0, 6, 11, 16, 21};
  const unsigned nch = H->mode == 3 ? 1 : 2, ngr = H->ver ? 1 : 2;
  for (unsigned gr = 0; gr < ngr; gr++)
    for (unsigned ch = 0; ch < nch; ch++) {
      for (unsigned sfb = 0; sfb < 21; sfb++)
        if (out->sf_l_set[gr][ch] >> sfb & 1) id->scalefac_l[gr][ch][sfb] = out->sf_l[gr][ch][sfb];
      if (gr == 1)
        for (unsigned g4 = 0; g4 < 4; g4++)
          if (out->sf_l_copy[ch] >> g4 & 1)
            for (unsigned sfb = lo[g4]; sfb < lo[g4 + 1]; sfb++) id->scalefac_l[1][ch][sfb] = id->scalefac_l[0][ch][sfb];
      for (unsigned sfb = 0; sfb < 12; sfb++)
        if (out->sf_s_set[gr][ch] >> sfb & 1) memcpy(id->scalefac_s[gr][ch][sfb], out->sf_s[gr][ch][sfb], 3);
      if (out->count1_set[gr][ch]) id->count1[gr][ch] = out->count1[gr][ch];
    }
}

static int fill_reservoir_pool(pdmp3_handle* id, unsigned size, unsigned begin);   /* bulk pipeline, below */

/* P:1346-1374: sizes + bit reservoir; the frame's bytes leave the ring here */
static int stage_main_data(pdmp3_handle* id) {
  const unsigned nch = id->hdr.mode == 3 ? 1 : 2;
  const unsigned fb = frame_bytes(&id->hdr);
  if (fb > 2000) return PDMP3_ERR;
  unsigned size = fb - side_info_bytes(&id->hdr) - 4;
  (void)nch;
  if (id->hdr.protection == 0) size -= 2;
  if (id->pool_sink) return fill_reservoir_pool(id, size, id->si.main_data_begin);
  return fill_reservoir(id, size, id->si.main_data_begin);
}

/* P:1217-1244.  With `defer` the main data is left undecoded in the reservoir (the bulk path snapshots
 * it and decodes on another thread); everything that touches the input ring has happened either way. */
static int read_frame_staged(pdmp3_handle* id) {
  if (search_header(id) != PDMP3_OK) return PDMP3_ERR;
  if (id->hdr.protection == 0) {                   /* CRC is skipped, never checked (P:1206-1210) */
    if (ring_byte(id) != BYTE_EOF) (void)ring_byte(id);
  }
  if (id->hdr.layer != 3) return PDMP3_ERR;
  id->fb_valid = 0;
  if (frame_bytes(&id->hdr) <= 2000) {
    if (id->side_to_bits && ring_filled(id) >= 32 && !id->hdr.ver) read_side_info_bits(id);
    else read_side_info(id);
  }
  return stage_main_data(id);
}

/* one whole frame, inline: `spectra` (2304 int16) receives is[gr][ch][576] of this frame */
static int read_frame(pdmp3_handle* id, int16_t* spectra) {
  const int res = read_frame_staged(id);
  if (res != PDMP3_OK) return res;
  main_out* out = &id->scratch_out;
  out->is = spectra;
  decode_main(id->main_vec, &id->hdr, &id->si, out);
  apply_main(id, &id->hdr, out);
  return PDMP3_OK;
}

/* ------------------------------------------------------------------------ */
/* parsed frame -> 4 gc records (the engine boundary)                        */
/* ------------------------------------------------------------------------ */
/* `spectra` already holds is[gr][ch] of the channels the frame has (decode_main wrote them there) */
static void emit_records(pdmp3_handle* id, const frame_header* H, const side_info* S, int reset,
                         int16_t* spectra, pdmp3_gc_side* sd) {
  const unsigned nch = H->mode == 3 ? 1 : 2;
  memset(sd, 0, 4 * sizeof *sd);
  const uint8_t fr = (uint8_t)((H->sfreq & 3) | (H->mode << PDMP3_FR_MODE_SHIFT) |
                               (H->mode_ext << PDMP3_FR_MODEEXT_SHIFT) | (reset ? PDMP3_FR_RESET : 0));
  for (unsigned g = 0; g < 4; g++) {
    const unsigned gr = g >> 1, ch = g & 1;
    pdmp3_gc_side* r = &sd[g];
    r->frame = fr;
    r->lsf = (uint8_t)H->ver;
    r->iso = (uint8_t)(((id->iso & PDMP3_ISO_MS_BOUND) ? PDMP3_GC_ISO_MS_ALL : 0) | ((id->iso & PDMP3_ISO_IS_SHORT) ? PDMP3_GC_ISO_IS_SHORT : 0) |
                       ((id->iso & PDMP3_ISO_IS_BOUND) ? PDMP3_GC_ISO_IS_STD : 0));
    if (ch >= nch || (H->ver && gr == 1)) { memset(spectra + g * 576, 0, 576 * sizeof(int16_t)); continue; }   /* (an LSF frame is one granule) */
    if (H->ver && ch == 1 && H->mode == 1 && (H->mode_ext & 1)) {
      /* channel 1 of an LSF intensity-stereo frame: its scalefactors are intensity positions, whose "not intensity
       * coded" value depends on the partition a position came in (include/pdmp3_hip.h) */
      uint8_t slen[4];
      int pf;
      const int cls = lsf_slen_of(S->scalefac_compress[0][1], 1, slen, &pf);
      const int shortb = S->win_switch[0][1] && S->block_type[0][1] == 2, mixed = shortb && S->mixed[0][1];
      if (S->scalefac_compress[0][1] & 1) r->lsf |= PDMP3_LSF_IS_SCALE;
      for (unsigned k = 0; k < 4; k++) { r->lsf_slen[k] = slen[k]; r->lsf_nsfb[k] = kLsfNsfb[cls][shortb ? (mixed ? 2 : 1) : 0][k]; }
    }
    r->count1 = id->count1[gr][ch];
    r->global_gain = (uint8_t)S->global_gain[gr][ch];
    r->flags = (uint8_t)((S->scalefac_scale[gr][ch] ? PDMP3_GC_SCALEFAC_SCALE : 0) |
                         (S->preflag[gr][ch] ? PDMP3_GC_PREFLAG : 0) |
                         (S->win_switch[gr][ch] ? PDMP3_GC_WIN_SWITCH : 0) |
                         ((S->block_type[gr][ch] & 3) << PDMP3_GC_BLOCK_TYPE_SHIFT) |
                         ((S->win_switch[gr][ch] && S->mixed[gr][ch]) ? PDMP3_GC_MIXED : 0));
    for (unsigned w = 0; w < 3; w++) r->subblock_gain[w] = (uint8_t)S->subblock_gain[gr][ch][w];
    memcpy(r->scalefac_l, id->scalefac_l[gr][ch], 21);
    memcpy(r->scalefac_s, id->scalefac_s[gr][ch], 36);
    /* What the reference reads one element past each array:
     * the first element of the NEXT [gr][ch] block, and for the last block
     * the start of the following member (scalefac_s, resp. the float bits of
     * is[0][0][w], which only the device knows). */
    if (g < 3) {
      r->scalefac_l[21] = id->scalefac_l[(g + 1) >> 1][(g + 1) & 1][0];
      memcpy(r->scalefac_s[12], id->scalefac_s[(g + 1) >> 1][(g + 1) & 1][0], 3);
    } else {
      r->scalefac_l[21] = id->scalefac_s[0][0][0][0];
      r->scalefac_s[12][0] = r->scalefac_s[12][1] = r->scalefac_s[12][2] = PDMP3_SF_PEEK;
    }
    /* the ISO switches (pdmp3_amd_set_quirks; not the reference): bands 21 / 12 have scalefactor 0 */
    if ((id->iso & PDMP3_ISO_SF21) || H->ver) r->scalefac_l[21] = 0;
    if ((id->iso & PDMP3_ISO_SF12) || H->ver) r->scalefac_s[12][0] = r->scalefac_s[12][1] = r->scalefac_s[12][2] = 0;
  }
  if (id->tap_side) {
    if (id->tap_n < id->tap_cap) {
      memcpy(id->tap_spectra + (size_t)id->tap_n * 2304, spectra, 2304 * sizeof(int16_t));
      memcpy(id->tap_side + (size_t)id->tap_n * 4, sd, 4 * sizeof *sd);
    }
    id->tap_n++;
  }
}

/* P:2307-2345: hand out up to buflen bytes of the frame under the cursor */
static size_t drain_frame(pdmp3_handle* id, unsigned char* out, size_t buflen) {
  const unsigned nch = id->l_hdr.mode == 3 ? 1 : 2;        /* the CURRENT header's channel count, as in the reference */
  const unsigned sh = (id->enc_f32 ? 2 : 1) + (nch - 1), bps = 1u << sh;
  const unsigned spf = frame_samples(&id->l_hdr);         /* 1152; 576 for an LSF frame (one granule) */
  size_t n = buflen >> sh;
  if (n > spf - id->ostart) n = spf - id->ostart;
  if (out) memcpy(out, (const unsigned char*)id->last_pcm + (size_t)id->ostart * bps, n * bps);
  id->ostart += (unsigned)n;
  if (id->ostart >= spf) id->ostart = 0;
  return n * bps;
}

/* ------------------------------------------------------------------------ */
/* pdmp3_read (P:2431-2481), batched                                         */
/* ------------------------------------------------------------------------ */
struct bulk;
static int bulk_push(struct bulk* b);             /* snapshot the frame read_frame_staged just staged */
static int bulk_at_limit(const struct bulk* b);

/* The whole-stream decoder's form of the read loop (`sink`): the call only does what touches the input ring and the
 * output cursor; main data decoding, the transforms and the PCM copy are the sink's business (bulk path below),
 * byte counts are the same.  The parser is never ahead here. */
static int read_impl_sink(pdmp3_handle* id, size_t outsize, size_t* done, struct bulk* sink) {
  *done = 0;
  int res = PDMP3_ERR;
  if (id->ostart) {                               /* rest of the frame a previous call could not fit */
    const size_t n = drain_frame(id, NULL, outsize);
    *done = n; outsize -= n;
    res = PDMP3_OK;
  }
  while (outsize) {
    if (ring_filled(id) < 1152) { res = PDMP3_NEED_MORE; break; }      /* H10 */
    if (bulk_at_limit(sink)) { res = PDMP3_OK; break; }                /* (split scan: this scanner's span ends here) */
    const size_t pos = id->processed;
    const unsigned mark = id->istart;
    res = read_frame_staged(id);
    if (res != PDMP3_OK && res != PDMP3_NEW_FORMAT) {   /* failed: rewind to the frame start (P:2459-2462) */
      id->processed = pos; id->istart = mark;
      sync_logical(id);
      break;
    }
    sync_logical(id);
    if (bulk_push(sink) != PDMP3_OK) return PDMP3_ERR;
    id->last_nch = id->hdr.mode == 3 ? 1 : 2;
    const size_t n = drain_frame(id, NULL, outsize);      /* (the cursor is NOT reset for a new frame: P:2307-2345) */
    outsize -= n; *done += n;
  }
  if (id->l_new_header == 1 && res == PDMP3_OK) res = PDMP3_NEW_FORMAT;
  return res;
}

/* ------------------------------------------------------------------------ */
/* Read-ahead.  pdmp3_read is synchronous per call (P:2431-2481): it parses   */
/* and decodes as many frames as the caller's buffer takes.  A GPU batch of   */
/* the three or four frames a 16 KiB buffer takes is all launch latency, so   */
/* when the parser has to parse a frame anyway it goes on through EVERY       */
/* complete frame the ring already holds and the engine decodes them as one   */
/* batch; later calls hand those frames out without parsing or launching.     */
/*                                                                            */
/* What the caller can observe must not change, so the handle keeps two       */
/* views of the stream: the parser's (istart, processed, hdr, new_header:     */
/* possibly ahead) and the reference's (l_*: what P:2431-2481 would have      */
/* consumed by now) -- pdmp3_feed's free space, the 1152-byte rule (H10),     */
/* pdmp3_getformat and the return codes use the second.  A frame is parsed    */
/* ahead only if the reference is certain to parse it to the same result      */
/* later: at least 1152 bytes are buffered behind its start NOW (more can     */
/* only be fed), the parse succeeded, and no step of it found the ring short  */
/* of bytes (ring_short) -- bytes fed later cannot change it then.  The       */
/* first frame of a batch is the one the reference parses in this very call:  */
/* its failures keep their side effects (H9: the reservoir keeps the bytes,   */
/* the ring is rewound); a frame read ahead that fails is undone completely   */
/* (snapshot) and left for the call in which the reference gets to it.        */
/* ------------------------------------------------------------------------ */
static void snap_save(const pdmp3_handle* id, parse_snap* p) {
  p->processed = id->processed; p->istart = id->istart; p->hdr = id->hdr; p->si = id->si;
  memcpy(p->scalefac_l, id->scalefac_l, sizeof p->scalefac_l);
  memcpy(p->scalefac_s, id->scalefac_s, sizeof p->scalefac_s);
  memcpy(p->count1, id->count1, sizeof p->count1);
  memcpy(p->main_vec, id->main_vec, sizeof p->main_vec); p->main_top = id->main_top;
  memcpy(p->side_vec, id->side_vec, sizeof p->side_vec); p->side_ptr = id->side_ptr; p->side_idx = id->side_idx;
  p->new_header = id->new_header; p->need_reset = id->need_reset; p->tap_n = id->tap_n;
}
static void snap_restore(pdmp3_handle* id, const parse_snap* p) {
  id->processed = p->processed; id->istart = p->istart; id->hdr = p->hdr; id->si = p->si;
  memcpy(id->scalefac_l, p->scalefac_l, sizeof p->scalefac_l);
  memcpy(id->scalefac_s, p->scalefac_s, sizeof p->scalefac_s);
  memcpy(id->count1, p->count1, sizeof p->count1);
  memcpy(id->main_vec, p->main_vec, sizeof p->main_vec); id->main_top = p->main_top;
  memcpy(id->side_vec, p->side_vec, sizeof p->side_vec); id->side_ptr = p->side_ptr; id->side_idx = p->side_idx;
  id->new_header = p->new_header; id->need_reset = p->need_reset; id->tap_n = p->tap_n;
}

static void ra_push(pdmp3_handle* id) {
  struct ra_entry* e = &id->ra[id->ra_n++];
  e->processed_after = id->processed; e->istart_after = id->istart; e->hdr = id->hdr;
  e->nch = (uint8_t)(id->hdr.mode == 3 ? 1 : 2);
  e->nh = (uint8_t)(id->new_header != 0);
}

/* Take back the frames read ahead but not handed out: the parser returns to the reference's position, the engine's
 * synthesis state to the last frame handed out.  Needed when the reference will NOT find what was read ahead: a
 * feed that fills the ring exactly makes it look empty to the reference (iend == istart, P:1062-1068), whose next
 * feeds then overwrite the unread frames; pdmp3_open_feed keeps the parse state of ITS position (H4-H6). */
static int ra_rollback(pdmp3_handle* id) {
  if (id->ra_head == id->ra_n) return PDMP3_OK;
  snap_restore(id, &id->ra_before[id->ra_head]);
  int rc = PDMP3_OK;
  if (id->hs && pdmp3_hip_stream_rewind(id->hs, 0, id->ra_head) != PDMP3_HIP_OK) {
    fprintf(stderr, "pdmp3: engine failure: %s\n", pdmp3_hip_last_error());
    rc = PDMP3_ERR;
  }
  id->ra_head = id->ra_n = id->ra_inflight = 0;
  return rc;
}

/* ------------------------------------------------------------------------ */
/* Helpers for a read-ahead batch.  What is sequential in a frame -- ring,    */
/* header, side info, bit reservoir -- is a fraction of a microsecond; its    */
/* main data (scalefactors + Huffman, 4-5 us at 320 kbps) only needs the      */
/* reservoir as that frame left it, and read_ahead has a copy of exactly that */
/* per frame (the snapshots it keeps for undoing frames).  So a batch's main  */
/* data is decoded by the caller AND a few helper threads, frame by frame off */
/* one counter; the merge into the state that survives frames (apply_main)    */
/* stays sequential.  The helpers are per process, started on first use, spin */
/* for a short while after a batch (the next one is usually 50-100 us away)   */
/* and then sleep.  PDMP3_STREAM_THREADS = helpers (default min(3, CPUs - 1); */
/* 0: none).  A second handle that reads while the helpers are busy decodes   */
/* its batch alone.                                                           */
/* ------------------------------------------------------------------------ */
#define HP_MAX 15
typedef struct { const uint8_t* res; const frame_header* H; const side_info* S; main_out* out; } hp_job;
static struct {
  pthread_mutex_t own;             /* one batch at a time */
  pthread_mutex_t m; pthread_cond_t cv;
  int started, n, sleepers;
  hp_job job[BATCH_MAX];
  _Atomic uint64_t state;          /* batch number << 32 | frames of the batch << 16 | next frame: ONE word, so that a
                                      helper that is late for a batch can never take a frame of it by the numbers of the next */
  _Atomic int done;
} g_hp = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, 0, 0, 0, {{0, 0, 0, 0}}, 0, 0};

static inline void hp_pause(void) {
#if defined(__x86_64__)
  __builtin_ia32_pause();
#else
  sched_yield();
#endif
}
/* Takes frames until the current batch has none left; returns that batch's number.  Whatever the fetch-and-add
 * returns IS a claim -- batch, frame count and index come out of one word -- also for a helper that arrives here
 * still thinking of the batch before: it must decode the frame it drew, nobody else will. */
static uint32_t hp_take(void) {
  for (;;) {
    const uint64_t v = atomic_fetch_add_explicit(&g_hp.state, 1, memory_order_acq_rel);
    const uint32_t k = (uint32_t)(v & 0xffff), n = (uint32_t)(v >> 16 & 0xffff);
    if (k >= n) return (uint32_t)(v >> 32);
    const hp_job* j = &g_hp.job[k];
    decode_main(j->res, j->H, j->S, j->out);
    atomic_fetch_add_explicit(&g_hp.done, 1, memory_order_release);
  }
}
/* How long a helper looks for the next batch before it sleeps on the condition: PDMP3_STREAM_SPIN = pause instructions
 * (default 1000: ~15 us; rounds 3-5: 20000, 0.2-0.5 ms -- three cores at 100 % per streaming handle; the
 * next batch of a caller that reads at the reference driver's cadence is 50-100 us away, measured with both:
 * profiles/r06_stream_api.json). */
static int g_hp_spin = 1000;
static void* hp_worker(void* arg) {
  (void)arg;
  uint32_t seen = 0;
  for (;;) {
    int spins = 0;
    while ((uint32_t)(atomic_load_explicit(&g_hp.state, memory_order_acquire) >> 32) == seen) {
      if (++spins < g_hp_spin) { hp_pause(); continue; }      /* a short look, then sleep */
      pthread_mutex_lock(&g_hp.m);
      g_hp.sleepers++;
      while ((uint32_t)(atomic_load_explicit(&g_hp.state, memory_order_acquire) >> 32) == seen) pthread_cond_wait(&g_hp.cv, &g_hp.m);
      g_hp.sleepers--;
      pthread_mutex_unlock(&g_hp.m);
      spins = 0;
    }
    seen = hp_take();
  }
  return NULL;
}
static int usable_cpus(void);
static void hp_start(void) {                                   /* (g_hp.own held) */
  g_hp.started = 1;
  const char* e = getenv("PDMP3_STREAM_THREADS");
  int n = e ? atoi(e) : usable_cpus() - 1;
  if (!e && n > 3) n = 3;
  if (n > HP_MAX) n = HP_MAX;
  const char* sp = getenv("PDMP3_STREAM_SPIN");
  if (sp && atoi(sp) >= 0) g_hp_spin = atoi(sp);
  for (int i = 0; i < n; i++) {
    pthread_t t;
    if (pthread_create(&t, NULL, hp_worker, NULL) != 0) break;
    pthread_detach(t);
    g_hp.n++;
  }
}
/* decode_main of jobs[0..n) */
static void hp_run(const hp_job* jobs, int n) {
  if (n > 1 && pthread_mutex_trylock(&g_hp.own) == 0) {
    if (!g_hp.started) hp_start();
    if (g_hp.n > 0) {
      memcpy(g_hp.job, jobs, (size_t)n * sizeof *jobs);
      atomic_store_explicit(&g_hp.done, 0, memory_order_relaxed);
      const uint32_t batch = (uint32_t)(atomic_load_explicit(&g_hp.state, memory_order_relaxed) >> 32) + 1;
      atomic_store_explicit(&g_hp.state, (uint64_t)batch << 32 | (uint64_t)n << 16, memory_order_release);
      pthread_mutex_lock(&g_hp.m);
      if (g_hp.sleepers) pthread_cond_broadcast(&g_hp.cv);
      pthread_mutex_unlock(&g_hp.m);
      (void)hp_take();
      while (atomic_load_explicit(&g_hp.done, memory_order_acquire) < n) hp_pause();
      pthread_mutex_unlock(&g_hp.own);
      return;
    }
    pthread_mutex_unlock(&g_hp.own);
  }
  for (int i = 0; i < n; i++) decode_main(jobs[i].res, jobs[i].H, jobs[i].S, jobs[i].out);
}

/* Parse the frame the reference parses now and, behind it, every frame that qualifies; send them to the engine.
 * Called with nothing read ahead (parser == logical view).  Returns the code of the FIRST frame's Read_Frame.
 * Three steps: everything that touches the ring, frame after frame; the frames' main data (hp_run); the merge
 * into the scalefactor / count1 state and the records, frame after frame. */
static int read_ahead(pdmp3_handle* id) {
  static _Thread_local int16_t scratch_sp[BATCH_MAX * 2304];      /* parse-only test handles: records go nowhere */
  static _Thread_local pdmp3_gc_side scratch_sd[BATCH_MAX * 4];
  int16_t* spectra = id->hs ? pdmp3_hip_stream_spectra(id->hs) : scratch_sp;
  pdmp3_gc_side* side = id->hs ? pdmp3_hip_stream_side(id->hs) : scratch_sd;
  id->ra_head = id->ra_n = 0;
  const size_t pos = id->processed;
  const unsigned mark = id->istart;
  snap_save(id, &id->ra_before[0]);
  const int res = read_frame_staged(id);
  if (res != PDMP3_OK) {                                  /* failed: rewind to the frame start (P:2459-2462) */
    id->processed = pos; id->istart = mark;
    sync_logical(id);                                     /* (the header it read stays, as in the reference) */
    return res;
  }
  const int reset0 = id->need_reset;
  id->need_reset = 0;
  ra_push(id);
  const int cap = getenv("PDMP3_NO_READAHEAD") ? 1 : BATCH_MAX;
  while (id->ra_n < cap) {
    /* bytes behind the parser, counted from the reference's cursor (a cursor that has crossed the end of the ring
     * while pdmp3_feed has iend parked there, P:2410-2417, sees the ring full of its own stale bytes) */
    const size_t ahead = id->processed - id->l_processed;
    const unsigned have = ring_filled_logical(id);
    const unsigned avail = have > ahead ? (unsigned)(have - ahead) : 0;
    if (avail < 1152) break;                              /* H10: the reference would not attempt it yet */
    parse_snap* snap = &id->ra_before[id->ra_n];
    snap_save(id, snap);
    id->ring_short = 0;
    const int r = read_frame_staged(id);
    /* undone unless it succeeded on bytes that were all there -- and belongs into this batch: the engine takes LSF
     * frames in launches of their own, all of one channel count (pdmp3_hip_stream_set_lsf) */
    if (r != PDMP3_OK || id->ring_short || id->processed - snap->processed > avail ||
        id->hdr.ver != id->ra[0].hdr.ver || (id->hdr.ver && (id->hdr.mode == 3) != (id->ra[0].hdr.mode == 3))) {
      snap_restore(id, snap);
      break;
    }
    ra_push(id);
  }
  /* frame i's header, side info and reservoir: what the snapshot taken before frame i + 1 holds -- the parser itself for the last */
  const int n = id->ra_n;
  hp_job jobs[BATCH_MAX];
  for (int i = 0; i < n; i++) {
    const parse_snap* nx = i + 1 < n ? &id->ra_before[i + 1] : NULL;
    jobs[i].res = nx ? nx->main_vec : id->main_vec;
    jobs[i].H = nx ? &nx->hdr : &id->hdr;
    jobs[i].S = nx ? &nx->si : &id->si;
    jobs[i].out = &id->ra_out[i];
    id->ra_out[i].is = spectra + (size_t)i * 2304;
  }
  hp_run(jobs, n);
  for (int i = 0; i < n; i++) {
    if (i) {                                              /* the snapshot before frame i gets the state the frames before it left */
      parse_snap* sn = &id->ra_before[i];
      memcpy(sn->scalefac_l, id->scalefac_l, sizeof sn->scalefac_l);
      memcpy(sn->scalefac_s, id->scalefac_s, sizeof sn->scalefac_s);
      memcpy(sn->count1, id->count1, sizeof sn->count1);
      sn->tap_n = id->tap_n;
    }
    apply_main(id, jobs[i].H, jobs[i].out);
    emit_records(id, jobs[i].H, jobs[i].S, i == 0 ? reset0 : 0, spectra + (size_t)i * 2304, side + (size_t)i * 4);
  }
  if (id->hs) {
    (void)pdmp3_hip_stream_set_lsf(id->hs, id->ra[0].hdr.ver != 0);
    if (pdmp3_hip_stream_submit(id->hs, 0, id->ra_n) != PDMP3_HIP_OK) {
      fprintf(stderr, "pdmp3: engine failure: %s\n", pdmp3_hip_last_error());
      id->ra_n = 0;
      return PDMP3_ERR;
    }
    id->ra_inflight = 1;
  }
  return res;
}

/* pdmp3_read (P:2431-2481) */
static int read_impl(pdmp3_handle* id, unsigned char* outmemory, size_t outsize, size_t* done) {
  *done = 0;
  int res = PDMP3_ERR;
  if (id->ostart) {                               /* rest of the frame a previous call could not fit */
    const size_t n = drain_frame(id, outmemory, outsize);
    *done = n; outsize -= n; outmemory += n;
    res = PDMP3_OK;
  }
  while (outsize) {
    if (id->ra_head == id->ra_n) {                /* nothing read ahead: the reference's own step */
      if (ring_filled_logical(id) < 1152) { res = PDMP3_NEED_MORE; break; }      /* H10 */
      res = read_ahead(id);
      if (id->ra_n == 0) {
        if (res == PDMP3_OK || res == PDMP3_NEW_FORMAT) return PDMP3_ERR;        /* engine failure */
        break;
      }
    } else {
      if (ring_filled_logical(id) < 1152) {       /* (cannot be after a plain feed; the net under ra_rollback's cases) */
        if (ra_rollback(id) != PDMP3_OK) return PDMP3_ERR;
        res = PDMP3_NEED_MORE;
        break;
      }
      res = PDMP3_OK;                             /* Read_Frame of a frame read ahead: it succeeded */
    }
    /* Decode_L3 + Convert_Frame_S16 of the frame at the head */
    const struct ra_entry* e = &id->ra[id->ra_head];
    if (!id->hs && !id->host_only) return PDMP3_ERR;
    if (id->ra_inflight) {
      if (pdmp3_hip_stream_wait(id->hs, 0) != PDMP3_HIP_OK) {
        fprintf(stderr, "pdmp3: engine failure: %s\n", pdmp3_hip_last_error());
        return PDMP3_ERR;
      }
      id->ra_inflight = 0;
    }
    id->l_processed = e->processed_after; id->l_istart = e->istart_after; id->l_hdr = e->hdr;
    if (!id->l_new_header && e->nh) id->l_new_header = 1;
    /* where frame ra_head of the batch lies in the slot's PCM (include/pdmp3_hip.h): an MPEG-1 frame in its own 4608-byte
     * place (9216 as float); LSF frames -- half the samples -- back to back when stereo, in pairs per place when mono */
    const size_t place = id->enc_f32 ? 9216u : 4608u;
    const size_t fbytes = (e->hdr.ver ? place / 4 : place / 2) * e->nch;
    const size_t off = !e->hdr.ver ? (size_t)id->ra_head * place
                       : e->nch == 2 ? (size_t)id->ra_head * fbytes : (size_t)(id->ra_head >> 1) * place + (size_t)(id->ra_head & 1) * fbytes;
    const unsigned char* pcm = id->hs ? (const unsigned char*)pdmp3_hip_stream_pcm(id->hs) + off : NULL;
    id->ra_head++;
    if (pcm && id->ostart == 0 && outsize >= fbytes) {    /* whole frame fits: copy straight through */
      memcpy(outmemory, pcm, fbytes);
      outmemory += fbytes; outsize -= fbytes; *done += fbytes;
      id->last_nch = e->nch;
    } else {
      /* Convert_Frame_S16 (P:2307-2345) starts at the cursor it finds: a new frame does not reset it.  (The cursor
       * is not 0 here only after a call whose buffer ended inside a sample-frame -- 1..3 stray bytes -- which makes
       * the reference decode and drop frames; the new frame is then handed out from that sample on.) */
      if (pcm) memcpy(id->last_pcm, pcm, fbytes);
      else memset(id->last_pcm, 0, fbytes);       /* parse-only test handle: silence */
      id->last_nch = e->nch;
      const size_t n = drain_frame(id, outmemory, outsize);
      outmemory += n; outsize -= n; *done += n;
    }
  }
  if (id->l_new_header == 1 && res == PDMP3_OK) res = PDMP3_NEW_FORMAT;
  return res;
}

int pdmp3_read(pdmp3_handle* id, unsigned char* outmemory, size_t outsize, size_t* done) {
  if (!(id && outmemory && outsize && done)) return PDMP3_ERR;
  return read_impl(id, outmemory, outsize, done);
}

/* parse-only variant of the read loop for host-logic tests: parses every frame
 * the ring allows, taps records, produces no PCM */
int pdmp3_amd_parse_available(pdmp3_handle* id) {
  int16_t sp[2304];
  pdmp3_gc_side sd[4];
  int res = PDMP3_NEED_MORE;
  while (ring_filled(id) >= 1152) {
    const size_t pos = id->processed;
    const unsigned mark = id->istart;
    res = read_frame(id, sp);
    if (res != PDMP3_OK && res != PDMP3_NEW_FORMAT) { id->processed = pos; id->istart = mark; sync_logical(id); return res; }
    emit_records(id, &id->hdr, &id->si, id->need_reset, sp, sd);
    id->need_reset = 0;
    sync_logical(id);
  }
  return PDMP3_NEED_MORE;
}

/* P:2491-2520 */
int pdmp3_decode(pdmp3_handle* id, const unsigned char* in, size_t insize, unsigned char* out, size_t outsize, size_t* done) {
  size_t take = ring_free_logical(id);
  *done = 0;
  if (take > insize) take = insize;               /* the surplus is silently dropped (H16) */
  int res = pdmp3_feed(id, in, take);
  if (res != PDMP3_OK) return res;
  if (out && outsize) {
    size_t got;
    res = pdmp3_read(id, out, outsize, &got);
    *done = got;
  } else if (id->l_processed == 0) {              /* probe: peek at the first header, then rewind */
    const size_t pos = id->processed;             /* (nothing is read ahead before the first frame is handed out) */
    const unsigned mark = id->istart;
    res = search_header(id);
    id->processed = pos; id->istart = mark;
    sync_logical(id);
    if (id->l_new_header == 1) res = PDMP3_NEW_FORMAT;
  }
  return res;
}

/* include/pdmp3.h: float output (not in the reference) */
int pdmp3_amd_set_encoding(pdmp3_handle* id, int encoding) {
  if (!id || (encoding != PDMP3_ENC_SIGNED_16 && encoding != PDMP3_ENC_FLOAT_32)) return PDMP3_ERR;
  const int want = encoding == PDMP3_ENC_FLOAT_32;
  if (want == id->enc_f32) return PDMP3_OK;
  if (ra_rollback(id) != PDMP3_OK) return PDMP3_ERR;          /* frames read ahead were decoded in the other format */
  if (id->ostart) {                                            /* the frame under the cursor was, too: convert what is left */
    const unsigned nch = id->l_hdr.mode == 3 ? 1 : 2;
    const unsigned spf = frame_samples(&id->l_hdr);
    if (want) { float* f = (float*)id->last_pcm; for (int k = (int)(spf * nch) - 1; k >= 0; k--) f[k] = (float)id->last_pcm[k] / 32767.0f; }
    else { const float* f = (const float*)id->last_pcm; for (unsigned k = 0; k < spf * nch; k++) { float v = f[k] * 32767.0f; id->last_pcm[k] = (int16_t)(v > 32767.0f ? 32767 : v < -32767.0f ? -32767 : (int)v); } }
  }
  if (id->hs && pdmp3_hip_stream_set_f32(id->hs, want) != PDMP3_HIP_OK) return PDMP3_ERR;
  id->enc_f32 = want;
  return PDMP3_OK;
}

/* P:2526-2535 */
/* ISO-correct switches (include/pdmp3.h): from the next frame parsed on.  Frames that pdmp3_read has
 * parsed ahead keep the mode they were parsed in. */
int pdmp3_amd_set_quirks(pdmp3_handle* id, unsigned iso_mask) {
  if (!id || (iso_mask & ~(PDMP3_ISO_ALL | PDMP3_ISO_LSF))) return PDMP3_ERR;
  id->iso = iso_mask;
  return PDMP3_OK;
}

int pdmp3_getformat(pdmp3_handle* id, long* rate, int* channels, int* encoding) {
  if (!(id && rate && channels && encoding)) return PDMP3_ERR;
  *encoding = id->enc_f32 ? PDMP3_ENC_FLOAT_32 : PDMP3_ENC_SIGNED_16;
  *rate = (long)kLsfSampleRates[sfreq9(&id->l_hdr)];
  *channels = id->l_hdr.mode == 3 ? 1 : 2;
  id->new_header = -1;
  id->l_new_header = -1;
  return PDMP3_OK;
}

/* ------------------------------------------------------------------------ */
/* Bulk decode of one whole in-memory stream (first "next" row:   */
/* the host Huffman stage in front of the transforms).  Not a reference      */
/* entry point; its OUTPUT is defined by one: byte for byte what pdmp3()     */
/* (P:2540-2589) writes for the same file.                                   */
/*                                                                          */
/*   A  sequential   the reference's read loop at the CLI's cadence (4096 B  */
/*                   feeds, 16 KiB reads): ring, header sync, side info, bit */
/*                   reservoir -- everything whose result depends on the     */
/*                   previous frame.  Each frame leaves a job: header, side  */
/*                   info and a snapshot of the reservoir.                   */
/*   B  parallel     decode_main() per job on the worker threads, spectra    */
/*                   written straight into the engine's pinned staging slot. */
/*   C  sequential   apply_main() + emit_records() in frame order: the       */
/*                   scalefactor / count1 state that survives frames.        */
/*   D  GPU, async   pdmp3_hip_stream_submit(); the PCM of window w-2 is     */
/*                   copied out while w-1 is on the GPU and w is in B.       */
/* ------------------------------------------------------------------------ */
#define BULK_SLOTS 6
#define BULK_GATH_EXTRA 16             /* copy-list entries beyond one per frame: segment images of a split scan's windows */
#define GATHER_MAX_HELPERS 8
#define GATHER_QUEUE 512
#define GATHER_TASK_ENTRIES 512          /* copy-list entries per task: half a megabyte of main data */
#define PAR_MAX_BATCH 64               /* private windows of a split scan that go into one window of the engine, at most */
#include <time.h>
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
#define PDMP3_BULK_REPLAY (-2)         /* see bulk_drive */
#define BULK_GRAB 8                   /* frames a worker takes per trip to the counter */
#define BULK_COPY_PIECE ((size_t)256 << 10)

/* PCM out of a pinned slot into the caller's (pageable) memory.  The destination is written once and not read here
 * again, the source was written by the DMA engine and is in no cache: on x86-64 the copy goes through non-temporal
 * stores -- no read-for-ownership of the destination lines, a third less memory traffic per byte than memcpy below
 * glibc's own non-temporal threshold (tens of megabytes; the pieces here are 256 KB) -- with the source prefetched a
 * few lines ahead.  With four copy threads a window's 19 MB have to leave at 11 GB/s per thread to keep up with
 * PCIe; a host whose cores do 7 GB/s with plain memcpy was the 6.4 M frames/s of the round-4 driver run (the same
 * build did 9.7 M elsewhere).  PDMP3_BULK_PLAIN_COPY=1: memcpy. */
#if defined(__x86_64__)
#include <emmintrin.h>
static int g_plain_copy = -1;
static void copy_out(unsigned char* dst, const unsigned char* src, size_t n) {
  int plain = __atomic_load_n(&g_plain_copy, __ATOMIC_RELAXED);
  if (plain < 0) { const char* e = getenv("PDMP3_BULK_PLAIN_COPY"); plain = (e && *e == '1') ? 1 : 0; __atomic_store_n(&g_plain_copy, plain, __ATOMIC_RELAXED); }
  if (plain || n < 4096) { memcpy(dst, src, n); return; }
  const size_t head = (size_t)(-(uintptr_t)dst & 63);          /* up to the destination's next cache line */
  if (head) { memcpy(dst, src, head); dst += head; src += head; n -= head; }
  size_t lines = n >> 6;
  while (lines--) {
    _mm_prefetch((const char*)src + 512, _MM_HINT_NTA);
    const __m128i a = _mm_loadu_si128((const __m128i*)src), b = _mm_loadu_si128((const __m128i*)src + 1);
    const __m128i c = _mm_loadu_si128((const __m128i*)src + 2), d = _mm_loadu_si128((const __m128i*)src + 3);
    _mm_stream_si128((__m128i*)dst, a); _mm_stream_si128((__m128i*)dst + 1, b);
    _mm_stream_si128((__m128i*)dst + 2, c); _mm_stream_si128((__m128i*)dst + 3, d);
    src += 64; dst += 64;
  }
  _mm_sfence();
  if (n & 63) memcpy(dst, src, n & 63);
}
#else
static void copy_out(unsigned char* dst, const unsigned char* src, size_t n) { memcpy(dst, src, n); }
#endif

typedef struct {
  frame_header hdr;
  side_info si;
  uint8_t reset;
  uint8_t res[RESERVOIR_BYTES];
} frame_job;

typedef struct {
  frame_job* jobs;
  main_out* outs;
  int n;
  int16_t* spectra;                   /* destination of this window's records */
  pdmp3_gc_side* side;
  int slot;
} bulk_window;

typedef struct {                      /* a window that is on the GPU */
  int n, active, all_stereo;          /* all_stereo: 2 = every frame stereo, 1 = every frame mono, 0 = mixed */
  unsigned char* dst;                 /* where this window's PCM goes (caller memory) and how much room is left there */
  size_t dst_cap;
  int direct;                         /* the GPU downloads straight to dst (pinned caller memory): nothing to copy */
  uint8_t* nch;
  long long sub_seq;                  /* its place in the submitter's queue */
  int lsf;                            /* a window of LSF frames (all of one version and channel count): half the PCM per frame,
                                         laid out as include/pdmp3_hip.h pdmp3_hip_decode_lsf_frames says */
} bulk_flight;

/* split scan: a window as a scanner thread leaves it -- what bits_push / fill_reservoir_pool write into an engine slot,
 * in private memory.  The bytes the scanner itself puts into the pool (a segment's image of the reservoir buffer) are
 * kept in `arena` and entered in the copy list like the main data the submitter gathers from the stream. */
#define PW_ARENA_BYTES (24u << 10)
struct par_cache;
typedef struct pre_window {
  long long index;                    /* window number within the stream */
  int cap;                            /* frames its arrays hold */
  struct par_cache* home;             /* where it goes when the stitcher is through with it (NULL: freed) */
  int n, gath_n;
  size_t pool_tail;
  pdmp3_frame_bits* bits;
  pdmp3_row_desc* desc;
  uint8_t* nch;
  void* gath;                         /* struct pool_copy[] */
  uint8_t* arena;
  size_t arena_len;
  double t_take, t_begin, t_done;     /* trace: taken by a scanner, its snapshot there, pushed */
} pre_window;
struct par_scan;

struct bulk {
  pdmp3_handle* id;
  int cap;                            /* frames a window holds */
  int target;                         /* frames the one-thread scan gives a window (<= cap) */
  int ramp_on;                        /* this stream's first windows are short (win_ramp): long streams only */
  int cur_target;                     /* this stream's window size: the slot's capacity for a stream that fits one slot, else `target` */
  int trace2; double tr_t0;           /* $PDMP3_BULK_TRACE >= 2: per-window lines, times from the stream's start */
  int count_only;                     /* scan: stage A alone */
  int bits_mode;                      /* main data goes to the device undecoded (pdmp3_hip_stream_submit_bits) */
  int device, window_arg;             /* what bulk_new was given (the LSF decoder below is made with the same) */
  struct bulk* lsf_alt;               /* bits mode + PDMP3_ISO_LSF: the host-Huffman decoder LSF streams go through (bulk_decode_impl) */
  pdmp3_frame_bits* bits_dst; uint8_t* res_dst;   /* where stage A writes the current window (bits mode) */
  int bits_n, bits_slot, bits_open;
  pdmp3_frame_bits* rec_bits; uint8_t* rec_res;   /* parse-only bits mode: caller memory */
  /* compact bits input (include/pdmp3_hip.h: pdmp3_row_desc): res_dst is the window's pool */
  int pool_mode;                      /* 1: rows go up as pool + descriptors; 0: as 2064-byte snapshots */
  pdmp3_row_desc* desc_dst;
  size_t pool_tail, pool_cap;
  int seg_first;                      /* index in the window of the first frame of the current segment */
  uint32_t seg_s_off;
  int need_segment;                   /* id->main_vec is the live buffer: the next regular frame starts a segment */
  uint32_t cur_row_off; unsigned cur_top; int cur_explicit, cur_staged;   /* what the frame just staged leaves for its descriptor */
  /* The frames' main data is not copied by the scanning thread (it is memory-bound there: two thirds of its time per
   * frame): stage A only notes where the bytes are in the caller's stream and where they go in the pool, and the
   * submitter thread copies them just before the window goes up (pool_gather).  What the scanner itself needs from
   * the pool before that -- a few KB per window, pool_materialize -- it copies early (pool_ensure). */
  struct pool_copy { const uint8_t* src; uint32_t dst, n; } *gath[BULK_SLOTS], *gath_cur;
  int gath_n, gath_cap;                           /* entries written / entries gath_cur has room for */
  cpu_set_t near_gpu;                             /* the CPUs of the GPU's NUMA node this process may use (empty: no binding) */
  int sky[RESERVOIR_BYTES + 1], sky_n;            /* frames of the segment no later frame has topped yet (pdmp3_row_desc.up) */
  pdmp3_row_desc* rec_desc; size_t rec_pool_cap;  /* parse-only pool mode (host tests): caller memory, one window */
  bulk_window win[2];
  int cur;                            /* window stage A is filling */
  bulk_window* in_b;                  /* window the workers hold, or NULL */
  long long windows;                  /* windows handed to the workers so far */
  long long frames;
  /* workers */
  pthread_t* th;
  int nth;
  pthread_mutex_t mu;
  pthread_cond_t cv_work, cv_done;
  bulk_window* work;
  const unsigned char* copy_src; unsigned char* copy_dst; size_t copy_bytes, copy_next, copy_row;
  long long gen;
  int next, active, quit;
  /* sinks: the engine (hs) or caller memory (parse only, host tests) */
  pdmp3_hip_stream* hs;
  bulk_flight flight[BULK_SLOTS];
  int16_t* rec_spectra; pdmp3_gc_side* rec_side; size_t rec_cap;
  unsigned char* pcm; size_t pcm_cap;
  size_t pcm_emitted;                 /* PCM bytes of all frames handed to stage C so far */
  int pcm_pinned;                     /* the caller's PCM buffer is pinned host memory (1) or device memory (2): direct target */
  int failed, busy;
  int carry;                          /* keep parse state (host handle / device sfstate) from the previous stream */
  /* bits mode: the engine calls of a window (H2D, kernels, D2H: ~40 us of driver time) are issued by a thread
   * of their own, in window order, so that the scanning thread does not stop for them */
  pthread_t sub_th;
  int sub_started, sub_quit, sub_rc;
  pthread_mutex_t sub_mu;
  pthread_cond_t sub_cv, sub_done_cv;
  int sub_slot[8], sub_n[8], sub_row[8];
  size_t sub_pool[8];                 /* pool bytes of the window (0: snapshot rows) */
  int sub_gath[8];                    /* entries of the slot's copy list */
  void* sub_dst[8];
  long long sub_head, sub_tail;       /* jobs enqueued / completed */
  long long sub_copied;               /* jobs whose main data is out of the caller's stream (<= sub_tail + 1) */
  long long par_taken, par_given_up;  /* streams the split scan took to their end / gave up half way (pdmp3_amd_bulk_split_scans) */
  size_t next_copy_row;               /* row size of the copy job bulk_collect last handed out */
  /* split scan (par_scan below): a scanner that fills private windows (struct pre_window) instead of engine slots */
  int scan_threads;                   /* scanners of the split scan (0: stage A on the calling thread alone) */
  int scan_forced;                    /* PDMP3_BULK_SCAN_THREADS was given: split whatever the destination */
  long long stream_win;               /* windows of the CURRENT stream closed so far (the first ones are short: win_frames) */
  /* the submitter's helpers: a window's main data (1 KB per frame, out of the caller's stream into the slot's pinned pool)
   * is copied by several threads at once -- on one thread it is 9 ms of an hour of audio, which is what bounds the
   * pipeline once the scan is split */
  /* the main-data copies of the windows (pool_gather), as tasks: helper threads take them, and so does whoever waits for
   * a slot's copies to be complete.  The one-thread scan's windows are cut into tasks when the submitter gets them; the
   * split scan's as its private windows are put into the slot, so that the copies are under way while the window fills. */
  pthread_t gh_th[GATHER_MAX_HELPERS];
  int gh_n, gh_quit;
  struct { uint8_t* pool; const void* list; int n, slot; } gq[GATHER_QUEUE];
  unsigned gq_head, gq_tail;          /* pushed / taken */
  int g_out[BULK_SLOTS];              /* tasks of the slot not finished yet */
  int g_pushed[BULK_SLOTS];           /* the open window's copies have been handed out as it was filled */
  pthread_mutex_t gh_mu; pthread_cond_t gh_cv, gh_done_cv;
  int win_sink;
  struct pre_window* pw_cur;
  struct par_scan* par;               /* where finished windows go */
  struct par_cache* pc;               /* the split scan's memory, kept from stream to stream */
  long long limit_frames;             /* stop parsing when `frames` reaches this (0: no limit) */
  double tr_take, tr_begin;           /* trace stamps for the window being scanned */
  uint8_t* priv_pool;                 /* win_sink: this scanner's pool of the window it is filling */
  void* slot_arena[BULK_SLOTS][PAR_MAX_BATCH];   /* the literal bytes (segment images) a slot's copy list points into, from its pre_windows */
  int slot_arena_n[BULK_SLOTS];
  double t_submit, t_gpuwait, t_poolwait;   /* PDMP3_BULK_TRACE=1: where the scanning thread waits */
  double t_sub_gather, t_sub_call, t_drive, t_subwait, t_tail;          /* ... and what the submitter thread spends on the main-data copies / the engine calls */
};

static void* bulk_worker(void* arg) {
  struct bulk* b = (struct bulk*)arg;
  long long seen = 0;
  for (;;) {
    pthread_mutex_lock(&b->mu);
    while (b->gen == seen && !b->quit) pthread_cond_wait(&b->cv_work, &b->mu);
    if (b->quit) { pthread_mutex_unlock(&b->mu); return NULL; }
    seen = b->gen;
    bulk_window* w = b->work;
    const unsigned char* csrc = b->copy_src;
    unsigned char* cdst = b->copy_dst;
    const size_t cbytes = b->copy_bytes, crow = b->copy_row;
    pthread_mutex_unlock(&b->mu);
    while (w) {
      const int i0 = __atomic_fetch_add(&b->next, BULK_GRAB, __ATOMIC_RELAXED);
      if (i0 >= w->n) break;
      const int i1 = i0 + BULK_GRAB < w->n ? i0 + BULK_GRAB : w->n;
      for (int i = i0; i < i1; i++) {
        main_out* o = &w->outs[i];
        o->is = w->spectra + (size_t)i * 2304;
        decode_main(w->jobs[i].res, &w->jobs[i].hdr, &w->jobs[i].si, o);
      }
    }
    for (;;) {                                     /* PCM of an older window: pinned slot -> caller memory */
      const size_t c0 = __atomic_fetch_add(&b->copy_next, BULK_COPY_PIECE, __ATOMIC_RELAXED);
      if (c0 >= cbytes) break;
      const size_t c1 = cbytes - c0 < BULK_COPY_PIECE ? cbytes : c0 + BULK_COPY_PIECE;
      if (crow == 4608) copy_out(cdst + c0, csrc + c0, c1 - c0);
      else {                                       /* mono: 2304-byte frames out of 4608-byte slots */
        for (size_t off = c0; off < c1;) {
          const size_t r = off / 2304, w = off % 2304;
          const size_t n = 2304 - w < c1 - off ? 2304 - w : c1 - off;
          memcpy(cdst + off, csrc + r * 4608 + w, n);
          off += n;
        }
      }
    }
    pthread_mutex_lock(&b->mu);
    if (--b->active == 0) pthread_cond_signal(&b->cv_done);
    pthread_mutex_unlock(&b->mu);
  }
}

/* hand the workers a window to decode (or NULL) and a byte range to copy (or none) */
static void bulk_wait_b(struct bulk* b);
static void bulk_start_b(struct bulk* b, bulk_window* w, const unsigned char* src, unsigned char* dst, size_t nbytes) {
  bulk_wait_b(b);                                 /* one job at a time */
  pthread_mutex_lock(&b->mu);
  b->work = w; b->next = 0;
  b->copy_src = src; b->copy_dst = dst; b->copy_bytes = nbytes; b->copy_next = 0; b->copy_row = b->next_copy_row;
  b->active = b->nth; b->gen++;
  pthread_cond_broadcast(&b->cv_work);
  pthread_mutex_unlock(&b->mu);
  b->in_b = w;
  b->busy = 1;
}
static void bulk_wait_b(struct bulk* b) {
  if (!b->busy) return;
  const double t0 = now_s();
  pthread_mutex_lock(&b->mu);
  while (b->active) pthread_cond_wait(&b->cv_done, &b->mu);
  pthread_mutex_unlock(&b->mu);
  b->t_poolwait += now_s() - t0;
  b->busy = 0;
}

/* PCM of a finished slot -> caller memory.  An all-stereo window is one contiguous block: with `job` it is
 * left to the workers (returned through src/dst/nbytes), otherwise copied here. */
static int bulk_collect(struct bulk* b, int slot, const unsigned char** jsrc, unsigned char** jdst, size_t* jbytes) {
  bulk_flight* f = &b->flight[slot];
  if (jbytes) *jbytes = 0;
  if (!f->active) return PDMP3_OK;
  const double t0 = now_s();
  const int wrc = pdmp3_hip_stream_wait(b->hs, slot);
  b->t_gpuwait += now_s() - t0;
  if (wrc != PDMP3_HIP_OK) return PDMP3_ERR;
  f->active = 0;
  if (f->direct) return PDMP3_OK;
  const unsigned char* src = (const unsigned char*)pdmp3_hip_stream_slot_pcm(b->hs, slot);
  const int to_device = b->pcm_pinned == 2;       /* memory the host cannot write: copies go through the engine */
  if (f->lsf) {
    /* stereo frames lie back to back (2304 bytes each), mono frames in pairs in the first half of a 4608-byte place */
    size_t off = 0;
    const size_t fb = f->all_stereo == 2 ? 2304 : 1152;
    for (int i = 0; i < f->n && off < f->dst_cap;) {
      const int run = f->all_stereo == 2 ? f->n - i : ((i & 1) || i + 1 == f->n ? 1 : 2);
      const unsigned char* from = f->all_stereo == 2 ? src + (size_t)i * 2304 : src + (size_t)(i >> 1) * 4608 + (size_t)(i & 1) * 1152;
      size_t n = (size_t)run * fb;
      if (n > f->dst_cap - off) n = f->dst_cap - off;
      if (to_device) { if (pdmp3_hip_copy_to_dest(f->dst + off, from, n) != PDMP3_HIP_OK) return PDMP3_ERR; }
      else memcpy(f->dst + off, from, n);
      off += (size_t)run * fb;
      i += run;
    }
    return PDMP3_OK;
  }
  if (f->all_stereo == 2 || (f->all_stereo == 1 && jbytes && !to_device)) {
    const size_t row = f->all_stereo == 2 ? 4608 : 2304;
    size_t n = (size_t)f->n * row;
    if (n > f->dst_cap) n = f->dst_cap;
    if (!n) return PDMP3_OK;
    if (to_device) return pdmp3_hip_copy_to_dest(f->dst, src, n) == PDMP3_HIP_OK ? PDMP3_OK : PDMP3_ERR;
    if (jbytes) { *jsrc = src; *jdst = f->dst; *jbytes = n; b->next_copy_row = row; }
    else memcpy(f->dst, src, n);
    return PDMP3_OK;
  }
  /* frames of both kinds (or mono frames without the pool): frame by frame, runs of equal frames as one copy */
  size_t off = 0;
  for (int i = 0; i < f->n;) {
    const size_t fb = 2304u * f->nch[i];
    int k = i + 1;
    if (fb == 4608) while (k < f->n && f->nch[k] == 2) k++;       /* stereo frames are dense in the slot */
    const size_t run = (size_t)(k - i) * fb;
    if (off < f->dst_cap) {
      const size_t n = run < f->dst_cap - off ? run : f->dst_cap - off;
      if (to_device) { if (pdmp3_hip_copy_to_dest(f->dst + off, src + (size_t)i * 4608, n) != PDMP3_HIP_OK) return PDMP3_ERR; }
      else memcpy(f->dst + off, src + (size_t)i * 4608, n);
    }
    off += run;
    i = k;
  }
  return PDMP3_OK;
}

/* a window whose frames all have the same channel count and that fits its destination goes there directly when
 * the destination is pinned */
static void flight_plan(struct bulk* b, bulk_flight* f) {
  const size_t row = (f->all_stereo == 2 ? 4608 : 2304) >> (f->lsf ? 1 : 0);
  f->direct = b->pcm_pinned && f->all_stereo != 0 && f->dst && (size_t)f->n * row <= f->dst_cap;
  /* (a window that cannot go there directly -- mixed mono / stereo frames, or the tail that does not fit -- is staged in
   * the slot's pinned buffer and copied by bulk_collect, through the engine when the destination is device memory) */
}

/* stage C + D of the window the workers have just finished */
static int bulk_finish_b(struct bulk* b) {
  bulk_window* w = b->in_b;
  bulk_wait_b(b);
  if (!w) return PDMP3_OK;
  b->in_b = NULL;
  pdmp3_handle* id = b->id;
  bulk_flight* f = b->hs ? &b->flight[w->slot] : NULL;
  if (f) {
    f->dst = b->pcm_emitted < b->pcm_cap ? b->pcm + b->pcm_emitted : NULL;
    f->dst_cap = b->pcm_emitted < b->pcm_cap ? b->pcm_cap - b->pcm_emitted : 0;
    f->n = w->n; f->all_stereo = -1;
  }
  for (int i = 0; i < w->n; i++) {
    const frame_job* j = &w->jobs[i];
    apply_main(id, &j->hdr, &w->outs[i]);
    emit_records(id, &j->hdr, &j->si, j->reset, w->spectra + (size_t)i * 2304, w->side + (size_t)i * 4);
    const unsigned nch = j->hdr.mode == 3 ? 1 : 2;
    if (f) { f->nch[i] = (uint8_t)nch; f->all_stereo = f->all_stereo < 0 ? (int)nch : (f->all_stereo == (int)nch ? f->all_stereo : 0); }
    b->pcm_emitted += 2u * frame_samples(&j->hdr) * nch;
  }
  if (f) {
    f->lsf = w->n && w->jobs[0].hdr.ver != 0;     /* (bulk_push closes a window where the version -- or an LSF stream's channel count -- changes) */
    flight_plan(b, f);
    (void)pdmp3_hip_stream_set_lsf(b->hs, f->lsf);
    if (pdmp3_hip_stream_submit_to(b->hs, w->slot, w->n, f->direct ? f->dst : NULL, (f->all_stereo == 1 ? 2304 : 4608) >> (f->lsf ? 1 : 0)) != PDMP3_HIP_OK) {
      fprintf(stderr, "pdmp3: engine failure: %s\n", pdmp3_hip_last_error());
      return PDMP3_ERR;
    }
    f->active = 1;
  }
  return PDMP3_OK;
}

/* The window stage A has filled (k) goes to the workers, together with the PCM copy of window k-3, whose slot it
 * takes over; before that, window k-1 leaves the workers for stage C and the GPU. */
static int bulk_rotate(struct bulk* b) {
  bulk_window* w = &b->win[b->cur];
  if (bulk_finish_b(b) != PDMP3_OK) return PDMP3_ERR;
  if (!w->n) return PDMP3_OK;
  const unsigned char* src = NULL;
  unsigned char* dst = NULL;
  size_t nbytes = 0;
  if (b->hs) {
    w->slot = (int)(b->windows % BULK_SLOTS);
    if (bulk_collect(b, w->slot, &src, &dst, &nbytes) != PDMP3_OK) return PDMP3_ERR;   /* window k-3 */
    w->spectra = pdmp3_hip_stream_slot_spectra(b->hs, w->slot);
    w->side = pdmp3_hip_stream_slot_side(b->hs, w->slot);
  } else {
    const size_t first = (size_t)b->frames - (size_t)w->n;
    if (first + (size_t)w->n > b->rec_cap) return PDMP3_ERR;
    w->spectra = b->rec_spectra + first * 2304;
    w->side = b->rec_side + first * 4;
  }
  bulk_start_b(b, w, src, dst, nbytes);
  b->windows++;
  b->cur ^= 1;
  b->win[b->cur].n = 0;
  return PDMP3_OK;
}

/* ---- bits mode: stage A writes side info + reservoir snapshot straight into the engine's pinned slot; scale-
 * factors, Huffman and the frame-to-frame merge run on the device (include/pdmp3_hip.h, submit_bits) ---- */
static void fill_frame_bits(const pdmp3_handle* id, pdmp3_frame_bits* fb, int newstream) {
  const frame_header* H = &id->hdr;
  const side_info* S = &id->si;
  const unsigned nch = H->mode == 3 ? 1 : 2;
  memset(fb, 0, sizeof *fb);
  fb->frame = (uint8_t)((H->sfreq & 3) | (H->mode << PDMP3_FR_MODE_SHIFT) | (H->mode_ext << PDMP3_FR_MODEEXT_SHIFT) |
                        (id->need_reset ? PDMP3_FR_RESET : 0) | (newstream ? PDMP3_FR_NEWSTREAM : 0));
  fb->iso = (uint8_t)id->iso;
  for (unsigned ch = 0; ch < nch; ch++)
    for (unsigned g4 = 0; g4 < 4; g4++) if (S->scfsi[ch][g4]) fb->scfsi[ch] |= (uint8_t)(1u << g4);
  for (unsigned gr = 0; gr < 2; gr++)
    for (unsigned ch = 0; ch < nch; ch++) {
      pdmp3_gc_bits* g = &fb->gc[gr * 2 + ch];
      g->part2_3_length = (uint16_t)S->part2_3_length[gr][ch];
      g->big_values = (uint16_t)S->big_values[gr][ch];
      g->global_gain = (uint8_t)S->global_gain[gr][ch];
      g->scalefac_compress = (uint8_t)S->scalefac_compress[gr][ch];
      g->flags = (uint8_t)((S->scalefac_scale[gr][ch] ? PDMP3_GC_SCALEFAC_SCALE : 0) |
                           (S->preflag[gr][ch] ? PDMP3_GC_PREFLAG : 0) |
                           (S->win_switch[gr][ch] ? PDMP3_GC_WIN_SWITCH : 0) |
                           ((S->block_type[gr][ch] & 3) << PDMP3_GC_BLOCK_TYPE_SHIFT) |
                           ((S->win_switch[gr][ch] && S->mixed[gr][ch]) ? PDMP3_GC_MIXED : 0));
      for (unsigned k = 0; k < 3; k++) {
        g->table_select[k] = (uint8_t)S->table_select[gr][ch][k];
        g->subblock_gain[k] = (uint8_t)S->subblock_gain[gr][ch][k];
      }
      g->region0_count = (uint8_t)S->region0_count[gr][ch];
      g->region1_count = (uint8_t)S->region1_count[gr][ch];
      g->count1table_select = (uint8_t)S->count1table_select[gr][ch];
    }
}

/* the main data a window's frames left in the caller's stream, into its pool (entries the scanner needed early have n = 0) */
/* (The pool is written once and read by the copy engine: non-temporal stores for the whole cache lines of an entry -- no
 *  read-for-ownership of 9 MB per window, nothing of it in the caches the scanners work in.  A frame's main data is about a
 *  kilobyte, its first and last partial line go the ordinary way.  PDMP3_BULK_GATHER_NT=0: memcpy.) */
static int g_gather_nt = -1;
static void pool_gather(uint8_t* pool, const struct pool_copy* g, int n) {
#if defined(__x86_64__)
  int nt = __atomic_load_n(&g_gather_nt, __ATOMIC_RELAXED);
  if (nt < 0) { const char* e = getenv("PDMP3_BULK_GATHER_NT"); nt = !(e && *e == '0'); __atomic_store_n(&g_gather_nt, nt, __ATOMIC_RELAXED); }
  if (nt) {
    for (int i = 0; i < n; i++) {
      size_t len = g[i].n;
      if (!len) continue;
      unsigned char* dst = pool + g[i].dst;
      const unsigned char* src = g[i].src;
      if (len < 256) { memcpy(dst, src, len); continue; }
      const size_t head = (size_t)(-(uintptr_t)dst & 63);
      if (head) { memcpy(dst, src, head); dst += head; src += head; len -= head; }
      for (size_t lines = len >> 6; lines; lines--) {
        const __m128i a = _mm_loadu_si128((const __m128i*)src), b = _mm_loadu_si128((const __m128i*)src + 1);
        const __m128i c = _mm_loadu_si128((const __m128i*)src + 2), d = _mm_loadu_si128((const __m128i*)src + 3);
        _mm_stream_si128((__m128i*)dst, a); _mm_stream_si128((__m128i*)dst + 1, b);
        _mm_stream_si128((__m128i*)dst + 2, c); _mm_stream_si128((__m128i*)dst + 3, d);
        src += 64; dst += 64;
      }
      if (len & 63) memcpy(dst, src, len & 63);
    }
    _mm_sfence();
    return;
  }
#endif
  for (int i = 0; i < n; i++) if (g[i].n) memcpy(pool + g[i].dst, g[i].src, g[i].n);
}
/* (gh_mu held) one task off the queue and done; 0: the queue is empty */
static int gather_take_locked(struct bulk* b) {
  if (b->gq_tail == b->gq_head) return 0;
  const unsigned k = b->gq_tail++ % GATHER_QUEUE;
  uint8_t* pool = b->gq[k].pool;
  const struct pool_copy* list = (const struct pool_copy*)b->gq[k].list;
  const int n = b->gq[k].n, slot = b->gq[k].slot;
  pthread_mutex_unlock(&b->gh_mu);
  pool_gather(pool, list, n);
  pthread_mutex_lock(&b->gh_mu);
  if (--b->g_out[slot] == 0) pthread_cond_broadcast(&b->gh_done_cv);
  return 1;
}
static void* gather_helper(void* arg) {
  struct bulk* b = (struct bulk*)arg;
  pthread_mutex_lock(&b->gh_mu);
  for (;;) {
    while (!b->gh_quit && b->gq_tail == b->gq_head) pthread_cond_wait(&b->gh_cv, &b->gh_mu);
    if (b->gh_quit) break;
    (void)gather_take_locked(b);
  }
  pthread_mutex_unlock(&b->gh_mu);
  return NULL;
}
/* the copies g[0, n) of `slot`'s window as tasks (a full queue: the caller does the copy itself) */
static void gather_push(struct bulk* b, int slot, uint8_t* pool, const struct pool_copy* g, int n) {
  for (int lo = 0; lo < n; lo += GATHER_TASK_ENTRIES) {
    const int k = n - lo < GATHER_TASK_ENTRIES ? n - lo : GATHER_TASK_ENTRIES;
    pthread_mutex_lock(&b->gh_mu);
    if (b->gq_head - b->gq_tail >= GATHER_QUEUE) { pthread_mutex_unlock(&b->gh_mu); pool_gather(pool, g + lo, k); continue; }
    const unsigned q = b->gq_head++ % GATHER_QUEUE;
    b->gq[q].pool = pool; b->gq[q].list = g + lo; b->gq[q].n = k; b->gq[q].slot = slot;
    b->g_out[slot]++;
    pthread_cond_signal(&b->gh_cv);
    pthread_mutex_unlock(&b->gh_mu);
  }
}
/* until the slot's copies are complete; takes tasks (any slot's) while it waits */
static void gather_wait(struct bulk* b, int slot) {
  pthread_mutex_lock(&b->gh_mu);
  while (b->g_out[slot]) if (!gather_take_locked(b)) pthread_cond_wait(&b->gh_done_cv, &b->gh_mu);
  pthread_mutex_unlock(&b->gh_mu);
}
static void* bulk_submitter(void* arg) {
  struct bulk* b = (struct bulk*)arg;
  for (;;) {
    pthread_mutex_lock(&b->sub_mu);
    while (b->sub_tail == b->sub_head && !b->sub_quit) pthread_cond_wait(&b->sub_cv, &b->sub_mu);
    if (b->sub_tail == b->sub_head) { pthread_mutex_unlock(&b->sub_mu); return NULL; }
    const int slot = b->sub_slot[b->sub_tail & 7], n = b->sub_n[b->sub_tail & 7], row = b->sub_row[b->sub_tail & 7];
    void* dst = b->sub_dst[b->sub_tail & 7];
    const size_t pool = b->sub_pool[b->sub_tail & 7];
    const int gn = b->sub_gath[b->sub_tail & 7];
    pthread_mutex_unlock(&b->sub_mu);
    const double t0 = now_s();
    if (pool) {
      if (gn > 0) gather_push(b, slot, pdmp3_hip_stream_slot_pool(b->hs, slot), b->gath[slot], gn);   /* (gn < 0: handed out while the window was filled) */
      gather_wait(b, slot);
    }
    const double t1 = now_s();
    pthread_mutex_lock(&b->sub_mu);                /* (whoever only waits for the stream's bytes to be free need not sit through the engine call) */
    b->sub_copied = b->sub_tail + 1;
    pthread_cond_broadcast(&b->sub_done_cv);
    pthread_mutex_unlock(&b->sub_mu);
    const int rc = pool ? pdmp3_hip_stream_submit_pool_to(b->hs, slot, n, pool, dst, row)
                        : pdmp3_hip_stream_submit_bits_to(b->hs, slot, n, dst, row);
    if (rc != PDMP3_HIP_OK) fprintf(stderr, "pdmp3: engine failure: %s\n", pdmp3_hip_last_error());
    const double t2 = now_s();
    b->t_sub_gather += t1 - t0; b->t_sub_call += t2 - t1;
    if (b->trace2) fprintf(stderr, "  submitter: %d frames to slot %d: taken at %.2f ms, copies %.0f us, engine call %.0f us\n", n, slot, (t0 - b->tr_t0) * 1e3, (t1 - t0) * 1e6, (t2 - t1) * 1e6);
    pthread_mutex_lock(&b->sub_mu);
    if (rc != PDMP3_HIP_OK) b->sub_rc = rc;
    b->sub_tail++;
    pthread_cond_broadcast(&b->sub_done_cv);
    pthread_mutex_unlock(&b->sub_mu);
  }
}
static long long sub_enqueue(struct bulk* b, int slot, int n, void* dst, int row, size_t pool_bytes, int gath_n) {
  pthread_mutex_lock(&b->sub_mu);
  const long long seq = b->sub_head;
  b->sub_slot[b->sub_head & 7] = slot; b->sub_n[b->sub_head & 7] = n;
  b->sub_dst[b->sub_head & 7] = dst; b->sub_row[b->sub_head & 7] = row; b->sub_pool[b->sub_head & 7] = pool_bytes;
  b->sub_gath[b->sub_head & 7] = gath_n;
  b->sub_head++;
  pthread_cond_signal(&b->sub_cv);
  pthread_mutex_unlock(&b->sub_mu);
  return seq;
}
static int sub_wait_seq(struct bulk* b, long long seq) {   /* window number `seq` of the queue has been handed to the GPU */
  if (!b->sub_started) return PDMP3_OK;
  pthread_mutex_lock(&b->sub_mu);
  while (b->sub_tail <= seq) pthread_cond_wait(&b->sub_done_cv, &b->sub_mu);
  const int rc = b->sub_rc;
  pthread_mutex_unlock(&b->sub_mu);
  return rc == PDMP3_HIP_OK ? PDMP3_OK : PDMP3_ERR;
}
/* every enqueued window's main data has been copied out of the caller's stream (the windows themselves may still be on
 * their way to the GPU: a failure there shows at the next call or at the wait) */
static int sub_drain_copied(struct bulk* b) {
  if (!b->sub_started) return PDMP3_OK;
  pthread_mutex_lock(&b->sub_mu);
  while (b->sub_copied < b->sub_head) pthread_cond_wait(&b->sub_done_cv, &b->sub_mu);
  const int rc = b->sub_rc;
  pthread_mutex_unlock(&b->sub_mu);
  return rc == PDMP3_HIP_OK ? PDMP3_OK : PDMP3_ERR;
}
static int sub_drain(struct bulk* b) {            /* every enqueued window has been handed to the GPU */
  if (!b->sub_started) return PDMP3_OK;
  pthread_mutex_lock(&b->sub_mu);
  while (b->sub_tail != b->sub_head) pthread_cond_wait(&b->sub_done_cv, &b->sub_mu);
  const int rc = b->sub_rc;
  pthread_mutex_unlock(&b->sub_mu);
  return rc == PDMP3_HIP_OK ? PDMP3_OK : PDMP3_ERR;
}

/* make the slot of window `windows` writable: its previous occupant (window - BULK_SLOTS) must be off the GPU; its PCM
 * goes home on the worker pool while stage A fills the slot's input side */
static int bulk_at_limit(const struct bulk* b) { return b->limit_frames && b->frames >= b->limit_frames; }
/* A stream's first windows are short -- cap / 8, cap / 8, cap / 4, cap / 2, then cap frames each: the GPU has something to
 * do after an eighth of a window's scan instead of a whole one (0.33 ms of a 7 ms decode at 4096 frames, twice that at
 * 8192), and the four together are exactly one full window, so every later window starts where it would have.  Device
 * Huffman with the compact upload only, and only for windows large enough to notice. */
static int win_ramp(const struct bulk* b) { return b->ramp_on && b->bits_mode && b->pool_mode && !b->win_sink && b->target >= 1024 && b->target % 8 == 0; }
static int win_frames(const struct bulk* b, long long w) {
  if (b->win_sink) return b->cap;                 /* (a split scan's private window) */
  if (!win_ramp(b) || w >= 4) return b->cur_target > 0 ? b->cur_target : b->target;
  return w < 2 ? b->target / 8 : w == 2 ? b->target / 4 : b->target / 2;
}
static pre_window* pw_new_in(struct par_cache* pc, int cap, long long index);
static void pw_free(pre_window* w);
/* bytes the scanner itself puts into the window's pool (a segment's image of the reservoir buffer, a frame's own image):
 * straight into the pool, and for a split scan's private window also into its arena and its copy list */
static int pool_literal(struct bulk* b, size_t dst, const uint8_t* src, size_t n) {
  memcpy(b->res_dst + dst, src, n);
  if (!b->win_sink) return PDMP3_OK;
  pre_window* w = b->pw_cur;
  if (w->arena_len + n > PW_ARENA_BYTES || b->gath_n >= b->gath_cap - 1) return PDMP3_ERR;
  memcpy(w->arena + w->arena_len, src, n);
  struct pool_copy* g = &b->gath_cur[b->gath_n++];
  g->src = w->arena + w->arena_len; g->dst = (uint32_t)dst; g->n = (uint32_t)n;
  w->arena_len += n;
  return PDMP3_OK;
}
static int bits_open_window(struct bulk* b) {
  b->bits_n = 0;
  b->bits_open = 1;
  b->pool_tail = 0; b->need_segment = 1; b->seg_first = 0; b->cur_explicit = 0; b->cur_staged = 0; b->sky_n = 0;
  b->gath_n = 0;
  if (b->win_sink) {                              /* split scan: a private window */
    b->pw_cur = pw_new_in(b->pc, b->cap, b->stream_win);
    if (!b->pw_cur) return PDMP3_ERR;
    b->bits_dst = b->pw_cur->bits; b->desc_dst = b->pw_cur->desc;
    b->gath_cur = (struct pool_copy*)b->pw_cur->gath; b->gath_cap = b->cap + BULK_GATH_EXTRA;
    b->res_dst = b->priv_pool;
    b->pool_cap = (size_t)b->cap * RESERVOIR_BYTES + PDMP3_POOL_SLACK_BYTES;
    return PDMP3_OK;
  }
  if (!b->hs) {                                   /* parse only: one "window" = the caller's arrays */
    b->bits_dst = b->rec_bits;
    b->res_dst = b->rec_res;
    b->desc_dst = b->rec_desc; b->pool_cap = b->rec_pool_cap;
    if (b->pool_mode) {
      free(b->gath[0]);
      b->gath_cur = b->gath[0] = (struct pool_copy*)malloc((b->rec_cap + 1) * sizeof(struct pool_copy));
      if (!b->gath_cur) return PDMP3_ERR;
      b->gath_cap = b->rec_cap + 1;
    }
    return PDMP3_OK;
  }
  b->bits_slot = (int)(b->windows % BULK_SLOTS);
  const unsigned char* src; unsigned char* dst; size_t nbytes;
  if (b->flight[b->bits_slot].active) {           /* (long done: BULK_SLOTS windows ago) */
    const double t0 = now_s();
    const int rc = sub_wait_seq(b, b->flight[b->bits_slot].sub_seq);   /* (not the windows queued after it) */
    b->t_subwait += now_s() - t0;
    if (rc != PDMP3_OK) return PDMP3_ERR;
  }
  if (bulk_collect(b, b->bits_slot, &src, &dst, &nbytes) != PDMP3_OK) return PDMP3_ERR;
  if (nbytes) bulk_start_b(b, NULL, src, dst, nbytes);
  b->g_pushed[b->bits_slot] = 0;
  for (int i = 0; i < b->slot_arena_n[b->bits_slot]; i++) free(b->slot_arena[b->bits_slot][i]);       /* (its window was gathered long ago) */
  b->slot_arena_n[b->bits_slot] = 0;
  b->bits_dst = pdmp3_hip_stream_slot_bits(b->hs, b->bits_slot);
  b->res_dst = pdmp3_hip_stream_slot_reservoir(b->hs, b->bits_slot);
  if (b->pool_mode) {
    b->desc_dst = pdmp3_hip_stream_slot_rowdesc(b->hs, b->bits_slot);
    b->pool_cap = pdmp3_hip_stream_pool_bytes(b->hs);
    b->gath_cur = b->gath[b->bits_slot]; b->gath_cap = b->cap + BULK_GATH_EXTRA * (PAR_MAX_BATCH + 1);
    if (!b->desc_dst || !b->gath_cur) return PDMP3_ERR;
  }
  return b->bits_dst && b->res_dst ? PDMP3_OK : PDMP3_ERR;
}

/* ---- compact bits input: the window's pool (include/pdmp3_hip.h, pdmp3_row_desc) ----
 * While a segment runs, id->main_vec is NOT updated: it keeps the buffer as it was when the segment began (that image
 * is in the pool at seg_s_off), and "the buffer's valid bytes [0, main_top) are the last main_top bytes of the pool"
 * holds.  pool_materialize() brings main_vec up to date again from the segment's frames (the same rule the device
 * applies, unpack_core.h row_byte): before anything irregular touches the buffer, and when the window closes. */
/* pool bytes [lo, hi) are needed now: the copies that are still only noted and touch them (the list is in pool order) */
static void pool_ensure(struct bulk* b, size_t lo, size_t hi) {
  for (int i = b->gath_n - 1; i >= 0; i--) {
    struct pool_copy* g = &b->gath_cur[i];
    if ((size_t)g->dst + g->n <= lo && g->n) break;           /* (entries done earlier have n = 0: keep looking) */
    if (g->n && g->dst < hi) {
      memcpy(b->res_dst + g->dst, g->src, g->n);
      if (!b->win_sink) g->n = 0;                 /* (a split scan's private pool is not the one that goes up: the entry stays) */
      else if ((size_t)g->dst < lo) break;
    }
  }
}
static void pool_materialize(struct bulk* b) {
  pdmp3_handle* id = b->id;
  if (b->need_segment) return;                    /* main_vec is live */
  unsigned covered = 0;
  if (b->sky_n)                                   /* the last frame, then up its links: each hop has a larger top */
    for (const pdmp3_row_desc* d = &b->desc_dst[b->sky[b->sky_n - 1]];; d -= d->up) {
      pool_ensure(b, (size_t)d->row_off + covered, (size_t)d->row_off + d->top);
      memcpy(id->main_vec + covered, b->res_dst + d->row_off + covered, d->top - covered);
      covered = d->top;
      if (!d->up) break;
    }
  b->need_segment = 1;
}

/* room for a segment start (2064 + 511), a frame's main data (< 2000) and an explicit image (2064) */
#define POOL_ROOM 6700u
_Static_assert(POOL_ROOM <= PDMP3_POOL_SLACK_BYTES, "a fresh window has room for its first frame");

/* Get_Main_Data (P:1096-1122) of the frame being staged, into the pool.  Same return codes and the same effect on
 * main_top and the ring as fill_reservoir. */
static int fill_reservoir_pool(pdmp3_handle* id, unsigned size, unsigned begin) {
  struct bulk* b = id->pool_sink;
  if (!b->bits_open && bits_open_window(b) != PDMP3_OK) { b->failed = 1; return PDMP3_ERR; }
  if (b->cur_staged) { b->failed = 1; return PDMP3_ERR; }   /* (every frame staged here is pushed: the pool is the only copy) */
  const int regular = begin <= id->main_top && size <= ring_filled(id) && begin + size <= sizeof id->main_vec;
  if (!regular) {
    /* reservoir underflow (H9), a frame the ring does not hold completely (H18), or more bytes than the buffer
     * takes: the reference's buffer arithmetic on the real buffer; a frame that is decoded all the same carries its
     * own image of the result */
    pool_materialize(b);
    b->cur_explicit = 1;
    return fill_reservoir(id, size, begin);
  }
  if (b->need_segment) {                          /* the buffer as it is now, then its valid tail once more */
    const unsigned h = id->main_top < 511 ? id->main_top : 511;
    b->seg_s_off = (uint32_t)b->pool_tail;
    if (pool_literal(b, b->pool_tail, id->main_vec, RESERVOIR_BYTES) != PDMP3_OK) { b->failed = 1; return PDMP3_ERR; }
    b->pool_tail += RESERVOIR_BYTES;
    if (h && pool_literal(b, b->pool_tail, id->main_vec + id->main_top - h, h) != PDMP3_OK) { b->failed = 1; return PDMP3_ERR; }
    b->pool_tail += h;
    b->seg_first = b->bits_n;
    b->sky_n = 0;
    b->need_segment = 0;
  }
  b->cur_row_off = (uint32_t)(b->pool_tail - begin);
  b->cur_top = begin + size;
  b->cur_explicit = 0;
  b->cur_staged = 1;
  if (id->vsrc) {                                 /* the bytes stay where they are for now (pool_gather) */
    /* the copy list holds one entry per frame and BULK_GATH_EXTRA literals (a split scan's private window; an engine
     * window that private windows are stitched into: that many per private window): never write past it, whatever
     * PW_ARENA_BYTES / RESERVOIR_BYTES let through (the window fails; the stream then takes the one-thread scan) */
    if (b->gath_n >= b->gath_cap) { b->failed = 1; return PDMP3_ERR; }
    struct pool_copy* g = &b->gath_cur[b->gath_n++];
    g->src = id->vsrc + id->vfed - ring_filled(id); g->dst = (uint32_t)b->pool_tail; g->n = size;
    id->istart = (id->istart + size) % INBUF_SIZE;
    id->processed += size;
  } else ring_take(id, b->res_dst + b->pool_tail, size);
  b->pool_tail += size;
  id->main_top = begin + size;
  return PDMP3_OK;
}

static int bits_close_window(struct bulk* b) {
  if (!b->bits_open) return PDMP3_OK;
  if (b->pool_mode) pool_materialize(b);          /* the next window starts from the live buffer */
  b->bits_open = 0;
  if (!b->bits_n) return PDMP3_OK;
  if (b->hs) {
    bulk_wait_b(b);                               /* the slot's old PCM has been copied out */
    bulk_flight* f = &b->flight[b->bits_slot];
    f->n = b->bits_n;
    f->dst = b->pcm_emitted < b->pcm_cap ? b->pcm + b->pcm_emitted : NULL;
    f->dst_cap = b->pcm_emitted < b->pcm_cap ? b->pcm_cap - b->pcm_emitted : 0;
    f->all_stereo = f->nch[0];
    for (int i = 0; i < f->n; i++) {
      if (f->nch[i] != f->nch[0]) f->all_stereo = 0;
      b->pcm_emitted += 2304u * f->nch[i];
    }
    const double t0 = now_s();
    flight_plan(b, f);
    f->sub_seq = sub_enqueue(b, b->bits_slot, b->bits_n, f->direct ? f->dst : NULL, f->all_stereo == 1 ? 2304 : 4608, b->pool_mode ? b->pool_tail : 0, b->g_pushed[b->bits_slot] ? -1 : b->gath_n);
    b->t_submit += now_s() - t0;
    f->active = 1;
  }
  b->windows++;
  b->stream_win++;
  return PDMP3_OK;
}

static int pw_close_window(struct bulk* b);       /* split scan, below */
static int bits_push(struct bulk* b) {
  pdmp3_handle* id = b->id;
  if (!b->bits_open && bits_open_window(b) != PDMP3_OK) { b->failed = 1; return PDMP3_ERR; }
  if (!b->hs && !b->win_sink && (size_t)b->frames > b->rec_cap) { b->failed = 1; return PDMP3_ERR; }
  const int i = b->bits_n++;
  if (id->fb_valid) {                             /* read_side_info_bits has built the record */
    const frame_header* H = &id->hdr;
    id->fb_cur.frame = (uint8_t)((H->sfreq & 3) | (H->mode << PDMP3_FR_MODE_SHIFT) | (H->mode_ext << PDMP3_FR_MODEEXT_SHIFT) |
                                 (id->need_reset ? PDMP3_FR_RESET : 0) | (b->frames == 1 && !b->carry ? PDMP3_FR_NEWSTREAM : 0));
    b->bits_dst[i] = id->fb_cur;
  } else
  fill_frame_bits(id, &b->bits_dst[i], b->frames == 1 && !b->carry);   /* a fresh handle's parse state is zero */
  id->need_reset = 0;
  if (b->pool_mode) {
    pdmp3_row_desc* d = &b->desc_dst[i];
    b->cur_staged = 0;
    if (b->cur_explicit) {                        /* its own image of the buffer (main_vec is live here) */
      d->row_off = d->s_off = (uint32_t)b->pool_tail;
      d->top = RESERVOIR_BYTES; d->back = 0; d->up = 0; d->reserved = 0;
      if (pool_literal(b, b->pool_tail, id->main_vec, RESERVOIR_BYTES) != PDMP3_OK) { b->failed = 1; return PDMP3_ERR; }
      b->pool_tail += RESERVOIR_BYTES;
      b->cur_explicit = 0;
    } else {
      d->row_off = b->cur_row_off; d->s_off = b->seg_s_off;
      d->top = (uint16_t)b->cur_top; d->back = (uint16_t)(i - b->seg_first); d->reserved = 0;
      /* previous frame of the segment with a larger top: the skyline seen from this frame (a stack of strictly
       * decreasing tops, so never deeper than 2064) */
      while (b->sky_n && b->desc_dst[b->sky[b->sky_n - 1]].top <= d->top) b->sky_n--;
      d->up = (uint16_t)(b->sky_n ? i - b->sky[b->sky_n - 1] : 0);
      b->sky[b->sky_n++] = i;
      if (i - b->seg_first >= 65000) pool_materialize(b);   /* (`back` / `up` are 16 bits: a very long window starts a new segment) */
    }
  } else memcpy(b->res_dst + (size_t)i * RESERVOIR_BYTES, id->main_vec, RESERVOIR_BYTES);
  if (b->hs) b->flight[b->bits_slot].nch[i] = (uint8_t)(id->hdr.mode == 3 ? 1 : 2);
  if (b->win_sink) b->pw_cur->nch[i] = (uint8_t)(id->hdr.mode == 3 ? 1 : 2);
  const int full = b->bits_n >= win_frames(b, b->stream_win) || (b->pool_mode && b->pool_tail + POOL_ROOM > b->pool_cap);
  if (b->win_sink && full && pw_close_window(b) != PDMP3_OK) { b->failed = 1; return PDMP3_ERR; }
  if (b->win_sink) return PDMP3_OK;
  if (b->hs && full && bits_close_window(b) != PDMP3_OK) { b->failed = 1; return PDMP3_ERR; }
  if (!b->hs && !b->win_sink && b->pool_mode && b->pool_tail + POOL_ROOM > b->pool_cap) { b->failed = 1; return PDMP3_ERR; }
  return PDMP3_OK;
}

static int bulk_push(struct bulk* b) {
  pdmp3_handle* id = b->id;
  b->frames++;
  if (b->count_only) { id->need_reset = 0; return PDMP3_OK; }
  if (b->bits_mode) return bits_push(b);
  bulk_window* w = &b->win[b->cur];
  /* the engine takes LSF frames in launches of their own, all of one channel count: such a frame opens a new window */
  if (w->n && (id->hdr.ver != w->jobs[0].hdr.ver || (id->hdr.ver && (id->hdr.mode == 3) != (w->jobs[0].hdr.mode == 3)))) {
    if (bulk_rotate(b) != PDMP3_OK) { b->failed = 1; return PDMP3_ERR; }
    w = &b->win[b->cur];
  }
  frame_job* j = &w->jobs[w->n++];
  j->hdr = id->hdr;
  j->si = id->si;
  j->reset = (uint8_t)id->need_reset;
  id->need_reset = 0;
  memcpy(j->res, id->main_vec, RESERVOIR_BYTES);
  if (w->n == b->cap && bulk_rotate(b) != PDMP3_OK) { b->failed = 1; return PDMP3_ERR; }
  return PDMP3_OK;
}

/* stage A: the CLI's loop (P:2566-2583) over a memory buffer.  Returns the PCM bytes pdmp3() would write. */
static long long bulk_drive(struct bulk* b, const unsigned char* mp3, size_t n) {
  pdmp3_handle* id = b->id;
  pdmp3_open_feed(id);
  id->vsrc = mp3; id->vfed = 0;                   /* the ring's bytes are the buffer's: only its indices move */
  size_t fed = 0, done, total = 0;
  int res;
  while ((res = read_impl_sink(id, INBUF_SIZE, &done, b)) != PDMP3_ERR) {
    total += done;
    /* More bytes consumed than were ever fed: the ring is being replayed.  pdmp3_feed parks iend AT INBUF_SIZE
     * when a feed ends exactly at the end of the ring (P:2410-2417); a frame that then ends exactly there (only
     * 1152-byte frames can: 32 kHz / 256 kbps, the H10 limit) wraps the read index to 0 != iend and the ring
     * looks full of its own stale contents (P:1464-1474).  The reference -- and pdmp3_read / pdmp3(), which keep
     * its behaviour -- then emit the last 16 KiB again, often forever.  There is no finite reference output to
     * match, so the whole-stream entry points stop here. */
    if (id->processed > fed) { id->vsrc = NULL; b->failed = 2; return PDMP3_BULK_REPLAY; }
    if (res == PDMP3_NEED_MORE) {
      size_t take = n - fed < 4096 ? n - fed : 4096;
      if (!take) break;
      if (id->vsrc && take > ring_free_logical(id)) {
        /* the CLI drops a feed the ring has no room for (H16, after an underflow's NEED_MORE): from here on the ring's
         * bytes are no longer the buffer's at `processed` -- give the ring its real contents and go on with copies */
        for (unsigned k = 0, f = ring_filled(id); k < f; k++) id->in[(id->istart + k) % INBUF_SIZE] = id->vsrc[id->vfed - f + k];
        id->vsrc = NULL;
      }
      (void)pdmp3_feed(id, mp3 + fed, take);
      fed += take;
    }
  }
  id->vsrc = NULL;
  return (long long)total;
}

/* ------------------------------------------------------------------------ */
/* Split scan (round 4).  Stage A is ~300 cycles of bit-field parsing and      */
/* index arithmetic per frame on ONE thread (profiles/r04_scan_sections.txt),  */
/* and with the PCM left on the device it is what bounds the whole-stream      */
/* decoder.  What is strictly sequential in it is little: a frame's position   */
/* follows from the header before it, the ring's indices from the CLI's feed   */
/* cadence, the reservoir's fill from the frame before.  So: a PRE-PASS hops    */
/* from header to header with the real ring arithmetic (pdmp3_feed on a        */
/* scratch handle), checks that every frame takes the scanner's regular path    */
/* and leaves, at a few window boundaries, what a scanner needs to start there  */
/* -- ring indices, reservoir fill, which earlier frames' bytes are still in    */
/* the reservoir buffer, which frames last set the side-info fields the         */
/* reference leaves stale (H20); SCANNER threads run the unchanged stage-A code */
/* from those points into private windows (pre_window); the calling thread      */
/* moves the windows into the engine's slots in stream order.  Anything the     */
/* pre-pass does not recognise as regular (resync, underflow H9, a frame the    */
/* ring does not hold H18, a dropped feed H16, a replayed ring) makes the whole */
/* stream go the one-thread way from the start: results are the sequential      */
/* scanner's by construction, bit for bit (tests compare).                      */
/* ------------------------------------------------------------------------ */
typedef struct {              /* what a frame's header and side info say by themselves (hop_parse) */
  uint32_t x;
  uint16_t fb, begin, top;    /* frame bytes, main_data_begin, begin + main-data bytes */
  uint8_t nch, crc;
  uint8_t ws;                 /* win_switch_flag of granule-channel g = gr * 2 + ch in bit g */
  uint8_t pad;
} hop1;
/* The pre-pass is itself a chain -- where a header is follows from the one before -- but only from a known header on:
 * HOP threads start at guessed places (the stream cut into equal parts), look for a header there that three more
 * headers follow, and hop from it; the pre-pass proper walks the first part itself and from there on reads the hop
 * threads' records instead of the stream (4 ns a frame instead of 27).  A guess is right when the part before it lands
 * exactly on it; one that is not (a header-like pattern inside main data that chains three times) makes the stream go
 * the one-thread way, as everything else the pre-pass does not like does. */
enum { SEG_RUNNING = 0, SEG_AT_NEXT, SEG_AT_END, SEG_BAD };
typedef struct pre_seg {
  struct par_scan* P; int j;
  size_t guess;
  hop1* rec; long long cap;
  long long x_start;          /* where its first header is: -2 not known yet, -1 none found (atomic) */
  char pad0[64];              /* (what the hop thread keeps writing has cache lines of its own: its neighbours write theirs as often) */
  long long count;            /* records written (atomic, release; moved on every 8 frames and at the end) */
  int state;                  /* SEG_* (atomic, release; final once not SEG_RUNNING) */
  double t_sync, t_done;      /* trace */
  char pad1[64];
} pre_seg;
#define PAR_MAX_SCANNERS 16
#define PAR_MAX_SEGS 8
#define PAR_AHEAD 64
/* What the split scan allocates per stream is tens of megabytes in blocks large enough for malloc to map and unmap each
 * time: every page of them faults in again on every stream, in threads that share one address space (3 ms of a 6 ms
 * decode on a 256-core host).  The decoder keeps them instead: the records of the pre-pass, each scanner's scratch, the
 * private windows (as many as can be in flight). */
struct hop_rec_s;
typedef struct par_cache {
  pthread_mutex_t mu;
  struct hop_rec_s* rec; long long rec_cap;
  hop1* seg_rec[PAR_MAX_SEGS]; long long seg_cap[PAR_MAX_SEGS];
  struct { struct bulk* wb; pdmp3_handle* id; uint8_t* pool; size_t pool_bytes; } scan[PAR_MAX_SCANNERS];
  pre_window* spare[2 * PAR_AHEAD]; int n_spare;
  /* the threads of the scan (hop threads, pre-pass, scanners): started when a stream first needs them, kept for the next
   * stream -- fourteen pthread_create / pthread_join per stream are 0.2 ms, as much as the scan of a five-minute file */
  pthread_mutex_t crew_mu; pthread_cond_t crew_cv, crew_done_cv;
  pthread_t crew[PAR_MAX_SCANNERS + PAR_MAX_SEGS];
  int crew_n, crew_busy, crew_quit;               /* threads / jobs taken or waiting to be taken */
  cpu_set_t near_gpu;                             /* where they run (struct bulk::near_gpu; empty: anywhere) */
  struct { void* (*fn)(void*); void* arg; int* left; } jobs[PAR_MAX_SCANNERS + PAR_MAX_SEGS];
  unsigned job_head, job_tail;
} par_cache;
typedef struct hop_rec_s {
  uint32_t x;                 /* offset of the frame's header in the stream */
  uint32_t md_src;            /* offset of its main data */
  uint64_t md_end;            /* main-data bytes of the stream up to and including this frame */
  uint16_t fb, begin, top;    /* frame bytes, main_data_begin, reservoir fill after it */
  uint8_t nch, crc;
} hop_rec;

typedef struct {              /* the scanner's state in front of frame `frame` (a window boundary) */
  long long frame;
  unsigned istart, iend;
  size_t processed, vfed, fed;
  unsigned main_top;
  int sky_n;
  int* sky;                   /* frames whose bytes are still visible in the reservoir buffer, oldest (largest top) first */
  long long last_ws0[4], last_ws1[4];   /* the last frame before it whose gc g had win_switch_flag 0 / 1 (-1: none) */
  int ready;
} span_snap;

struct par_scan {
  /* ---- set before the threads start, read by all of them (the pre-pass looks at abort / quit once per frame: none of this
   *      shares a cache line with what the scanners and the stitcher write) */
  struct bulk* b;
  const unsigned char* mp3; size_t n;
  int K;
  /* Scanners take WINDOWS in turn (a shared counter): window w is scanned from the snapshot the pre-pass leaves at its
   * first frame.  (Contiguous spans per scanner starve the GPU while the first scanner works through its span alone: one
   * scanner produces windows at half the rate the GPU takes them.  Window by window, w is ready at
   * w x [pre-pass time per window] + [scan time of one window], always ahead of the GPU's w x 157 us.) */
  int sub;                    /* frames of a private window (the engine's windows are made of several: par_drive) */
  int spin;                   /* a scanner whose snapshot is the next or the one after spins for it (hosts with cores to spare); else it yields */
  int one_window;             /* by its first frame's size the stream fits one window of the engine */
  span_snap* snap; long long snap_cap;   /* by window index; [0] unused (a fresh handle) */
  hop_rec* rec; long long rec_cap;
  pre_window** win; long long win_cap;    /* finished windows by stream index (entries: under the mutex) */
  int J;                      /* parts of the pre-pass: [0] is the pre-pass thread's own, the others have a hop thread each */
  struct scanner_arg* args;
  double t0;
  atomic_int abort;           /* (set under the mutex, so that waiters wake; also looked at in loops that hold no lock) */
  atomic_int quit;            /* the stitcher has left: nobody wants further windows (not an error) */
  char pad0[64];
  /* ---- the pre-pass's: how far it is.  The scanners do not sleep on a condition for their snapshots -- with a dozen of them
   *      waiting, every broadcast (a snapshot, a finished window, a window taken) woke them all and the mutex they then queued
   *      for was the pre-pass's too: 2.7 ms of pre-pass with 8 scanners, 5.8 ms with 16 -- they watch this counter */
  long long published;        /* windows < this have their snapshot (atomic, release; 1 from the start: window 0 needs none) */
  long long n_frames;         /* valid once prepass_done */
  int prepass_done, irregular;   /* (prepass_done: atomic, release; set under the mutex as well: the stitcher sleeps on the condition) */
  double t_prepass, t_pre_part0, t_pre_wait;
  char pad1[64];
  /* ---- the scanners' and the stitcher's */
  long long next_win;         /* the next window nobody has taken yet (atomic) */
  long long stitched;         /* windows the stitcher has taken (atomic; written under the mutex) */
  int scanners_done;
  pthread_mutex_t mu; pthread_cond_t cv;
  int jobs_left, hops_left;   /* pre-pass and scanners / hop threads that have not returned (under the crew's mutex) */
  char pad2[64];
  pre_seg seg[PAR_MAX_SEGS];
};

static void pw_destroy(pre_window* w) {
  if (!w) return;
  free(w->bits); free(w->desc); free(w->nch); free(w->gath); free(w->arena);
  free(w);
}
static pre_window* pw_new_in(par_cache* pc, int cap, long long index) {
  pre_window* w = NULL;
  if (pc) {
    pthread_mutex_lock(&pc->mu);
    while (pc->n_spare && !w) {
      w = pc->spare[--pc->n_spare];
      if (w->cap < cap) { pw_destroy(w); w = NULL; }    /* (too small for this stream's private windows; a larger one does) */
    }
    pthread_mutex_unlock(&pc->mu);
  }
  if (w) {
    w->index = index; w->n = w->gath_n = 0; w->pool_tail = 0; w->arena_len = 0;
    if (!w->arena && !(w->arena = (uint8_t*)malloc(PW_ARENA_BYTES))) { pw_destroy(w); return NULL; }    /* (it went with a slot) */
    return w;
  }
  w = (pre_window*)calloc(1, sizeof *w);
  if (!w) return NULL;
  w->index = index; w->cap = cap; w->home = pc;
  w->bits = (pdmp3_frame_bits*)malloc((size_t)cap * sizeof(pdmp3_frame_bits));
  w->desc = (pdmp3_row_desc*)malloc((size_t)cap * sizeof(pdmp3_row_desc));
  w->nch = (uint8_t*)malloc((size_t)cap);
  w->gath = malloc(((size_t)cap + BULK_GATH_EXTRA) * sizeof(struct pool_copy));
  w->arena = (uint8_t*)malloc(PW_ARENA_BYTES);
  if (!w->bits || !w->desc || !w->nch || !w->gath || !w->arena) { pw_destroy(w); return NULL; }
  return w;
}
static void pw_free(pre_window* w) {
  if (!w) return;
  par_cache* pc = w->home;
  if (pc) {
    pthread_mutex_lock(&pc->mu);
    if (pc->n_spare < (int)(sizeof pc->spare / sizeof pc->spare[0])) { pc->spare[pc->n_spare++] = w; w = NULL; }
    pthread_mutex_unlock(&pc->mu);
  }
  pw_destroy(w);
}
static par_cache* pc_new(void) {
  par_cache* pc = (par_cache*)calloc(1, sizeof *pc);
  if (pc) {
    pthread_mutex_init(&pc->mu, NULL);
    pthread_mutex_init(&pc->crew_mu, NULL); pthread_cond_init(&pc->crew_cv, NULL); pthread_cond_init(&pc->crew_done_cv, NULL);
  }
  return pc;
}
#define CREW_MAX (PAR_MAX_SCANNERS + PAR_MAX_SEGS)
static void* crew_main(void* arg) {
  par_cache* pc = (par_cache*)arg;
  pthread_mutex_lock(&pc->crew_mu);
  for (;;) {
    while (!pc->crew_quit && pc->job_tail == pc->job_head) pthread_cond_wait(&pc->crew_cv, &pc->crew_mu);
    if (pc->job_tail == pc->job_head) break;            /* (quit, nothing left) */
    const unsigned k = pc->job_tail++ % CREW_MAX;
    void* (*fn)(void*) = pc->jobs[k].fn;
    void* a = pc->jobs[k].arg;
    int* left = pc->jobs[k].left;
    pthread_mutex_unlock(&pc->crew_mu);
    (void)fn(a);
    pthread_mutex_lock(&pc->crew_mu);
    pc->crew_busy--;
    (*left)--;                                          /* (under crew_mu: whoever waits for 0 may free what `left` is part of) */
    pthread_cond_broadcast(&pc->crew_done_cv);
  }
  pthread_mutex_unlock(&pc->crew_mu);
  return NULL;
}
/* fn(arg) on a thread of its own, at once (the jobs of a stream wait for each other: each needs a thread); *left counts
 * the stream's jobs that have not returned.  -1: no thread to be had. */
static int crew_run(par_cache* pc, void* (*fn)(void*), void* arg, int* left) {
  pthread_mutex_lock(&pc->crew_mu);
  if (pc->crew_busy >= pc->crew_n) {
    if (pc->crew_n >= CREW_MAX || pthread_create(&pc->crew[pc->crew_n], NULL, crew_main, pc) != 0) { pthread_mutex_unlock(&pc->crew_mu); return -1; }
    if (CPU_COUNT(&pc->near_gpu) > 0) (void)pthread_setaffinity_np(pc->crew[pc->crew_n], sizeof pc->near_gpu, &pc->near_gpu);
    pc->crew_n++;
  }
  const unsigned k = pc->job_head++ % CREW_MAX;
  pc->jobs[k].fn = fn; pc->jobs[k].arg = arg; pc->jobs[k].left = left;
  pc->crew_busy++;
  (*left)++;
  pthread_mutex_unlock(&pc->crew_mu);
  return 0;
}
/* the jobs queued so far may start: ONE wake-up for all of them (a signal per job is a system call per job on the calling
 * thread -- eleven of them in front of a file of a few minutes that is scanned in 0.2 ms) */
static void crew_kick(par_cache* pc) {
  pthread_mutex_lock(&pc->crew_mu);
  pthread_cond_broadcast(&pc->crew_cv);
  pthread_mutex_unlock(&pc->crew_mu);
}
static void crew_wait(par_cache* pc, int* left) {
  pthread_mutex_lock(&pc->crew_mu);
  while (*left) pthread_cond_wait(&pc->crew_done_cv, &pc->crew_mu);
  pthread_mutex_unlock(&pc->crew_mu);
}
static void pc_free(par_cache* pc) {
  if (!pc) return;
  pthread_mutex_lock(&pc->crew_mu); pc->crew_quit = 1; pthread_cond_broadcast(&pc->crew_cv); pthread_mutex_unlock(&pc->crew_mu);
  for (int i = 0; i < pc->crew_n; i++) pthread_join(pc->crew[i], NULL);
  pthread_mutex_destroy(&pc->crew_mu); pthread_cond_destroy(&pc->crew_cv); pthread_cond_destroy(&pc->crew_done_cv);
  for (int i = 0; i < pc->n_spare; i++) pw_destroy(pc->spare[i]);
  for (int j = 0; j < PAR_MAX_SEGS; j++) free(pc->seg_rec[j]);
  for (int k = 0; k < PAR_MAX_SCANNERS; k++) { free(pc->scan[k].wb); free(pc->scan[k].id); free(pc->scan[k].pool); }
  free(pc->rec);
  pthread_mutex_destroy(&pc->mu);
  free(pc);
}
/* the private window is complete: the reservoir buffer is brought up to date for the next one (as bits_close_window
 * does) and the window goes to whoever stitches the stream together */
static int pw_close_window(struct bulk* b) {
  if (!b->bits_open) return PDMP3_OK;
  pool_materialize(b);
  b->bits_open = 0;
  pre_window* w = b->pw_cur;
  b->pw_cur = NULL;
  if (!b->bits_n) { pw_free(w); return PDMP3_OK; }
  w->n = b->bits_n; w->gath_n = b->gath_n; w->pool_tail = b->pool_tail;
  w->t_take = b->tr_take; w->t_begin = b->tr_begin; w->t_done = now_s();
  struct par_scan* P = b->par;
  pthread_mutex_lock(&P->mu);
  if (w->index < P->win_cap && !P->win[w->index]) { P->win[w->index] = w; w = NULL; }
  pthread_cond_broadcast(&P->cv);
  pthread_mutex_unlock(&P->mu);
  if (w) { pw_free(w); return PDMP3_ERR; }          /* (cannot happen: more windows than the stream has bytes for) */
  b->windows++;
  b->stream_win++;
  return PDMP3_OK;
}

static void header_fields(uint32_t h, frame_header* H) {
  H->id = (h >> 19) & 1; H->layer = 4 - ((h >> 17) & 3); H->protection = (h >> 16) & 1;
  H->bitrate_index = (h >> 12) & 15; H->sfreq = (h >> 10) & 3; H->padding = (h >> 9) & 1;
  H->mode = (h >> 6) & 3; H->mode_ext = (h >> 4) & 3;
  H->ver = 0;                                      /* (the split scan and the window estimates are MPEG-1's: bits mode never takes LSF) */
}
static inline uint32_t be32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

/* A frame's own fields at offset x (at least 40 bytes of stream from there); -1: not a frame the regular path takes */
static int hop_parse(const unsigned char* mp3, size_t x, hop1* r) {
  const uint32_t hw = be32(mp3 + x);
  if ((hw & 0xfff00000u) != 0xfff00000u) return -1;     /* the scanner would search for the next sync */
  frame_header H;
  header_fields(hw, &H);
  if (H.id != 1 || H.bitrate_index == 0 || H.bitrate_index == 15 || H.sfreq == 3 || H.layer != 3) return -1;
  const unsigned nch = H.mode == 3 ? 1 : 2, nbytes = nch == 1 ? 17 : 32, crc = H.protection == 0 ? 2 : 0;
  const unsigned fb = frame_bytes(&H);
  if (fb > 2000) return -1;
  const unsigned char* v = mp3 + x + 4 + crc;
  const unsigned begin = ((unsigned)v[0] << 1) | (v[1] >> 7);
  const unsigned size = fb - nbytes - 4 - crc;
  if (begin + size > RESERVOIR_BYTES) return -1;        /* (overflow of the reservoir buffer) */
  unsigned ws = 0;
  for (unsigned gr = 0; gr < 2; gr++)                   /* which frame last set the fields the reference leaves stale (H20) */
    for (unsigned ch = 0; ch < nch; ch++) {
      const unsigned pos = (nch == 1 ? 18u : 20u) + 59u * (gr * nch + ch) + 33u;
      ws |= ((v[pos >> 3] >> (7 - (pos & 7))) & 1u) << (gr * 2 + ch);
    }
  r->x = (uint32_t)x; r->fb = (uint16_t)fb; r->begin = (uint16_t)begin; r->top = (uint16_t)(begin + size);
  r->nch = (uint8_t)nch; r->crc = (uint8_t)crc; r->ws = (uint8_t)ws; r->pad = 0;
  return 0;
}
/* this is a chain of dependent cache misses -- where the next header is follows from this one -- unless the lines are
 * asked for ahead of time: frames of a constant-bitrate stream are as long as each other to within the padding byte, so
 * the header and side info of the 12th frame from here lie within 12 bytes of x + 12 fb (two lines cover them); on a
 * variable-bitrate stream the guess is wrong and costs nothing */
static inline void hop_prefetch(const unsigned char* mp3, size_t x, unsigned fb) {
  __builtin_prefetch(mp3 + x + 12u * fb, 0, 3);
  __builtin_prefetch(mp3 + x + 12u * fb + 64, 0, 3);
}
#define HOP_END_BYTES 1152u   /* the CLI's loop stops asking once fewer bytes than this are left (H10) */
static long long seg_wait_start(pre_seg* S) {           /* (found within microseconds of the thread's start) */
  long long v;
  while ((v = __atomic_load_n(&S->x_start, __ATOMIC_ACQUIRE)) == -2) {
    if (S->P->abort || S->P->quit) return -1;
    sched_yield();
  }
  return v;
}
static void* par_hop_thread(void* arg) {
  pre_seg* S = (pre_seg*)arg;
  struct par_scan* P = S->P;
  const unsigned char* mp3 = P->mp3;
  const size_t n = P->n;
  /* a header like the stream's first (MPEG-1 Layer III, same sampling rate) that three more follow */
  const uint32_t like = be32(mp3) & 0xfffe0c00u;
  long long found = -1;
  const size_t stop = S->guess + 65536 < n ? S->guess + 65536 : n;
  for (size_t x = S->guess; x < stop && x + 4u * 2000u + 64u <= n; x++) {
    if (mp3[x] != 0xff || (be32(mp3 + x) & 0xfffe0c00u) != like) continue;
    size_t y = x;
    int ok = 1;
    for (int k = 0; k < 4 && ok; k++) {
      hop1 r;
      if ((be32(mp3 + y) & 0xfffe0c00u) != like || hop_parse(mp3, y, &r) != 0) ok = 0;
      else y += r.fb;
    }
    if (ok) { found = (long long)x; break; }
  }
  __atomic_store_n(&S->x_start, found, __ATOMIC_RELEASE);
  S->t_sync = now_s() - P->t0;
  int state = SEG_BAD;
  if (found >= 0) {
    long long next_start = -3;                          /* (-3: the last part) */
    if (S->j + 1 < P->J) next_start = seg_wait_start(&P->seg[S->j + 1]);
    size_t x = (size_t)found;
    long long c = 0;
    for (;;) {
      if (next_start == -1 || P->abort) break;          /* (the next part has no start: given up) */
      if (next_start >= 0 && x >= (size_t)next_start) { state = x == (size_t)next_start ? SEG_AT_NEXT : SEG_BAD; break; }
      if (n - x < HOP_END_BYTES) { state = SEG_AT_END; break; }
      if (c >= S->cap || hop_parse(mp3, x, &S->rec[c]) != 0) break;
      hop_prefetch(mp3, x, S->rec[c].fb);
      x += S->rec[c].fb;
      c++;
      if (!(c & 7)) __atomic_store_n(&S->count, c, __ATOMIC_RELEASE);
    }
    __atomic_store_n(&S->count, c, __ATOMIC_RELEASE);
  }
  S->t_done = now_s() - P->t0;
  __atomic_store_n(&S->state, state, __ATOMIC_RELEASE);
  return NULL;
}

/* The pre-pass.  Returns 0 when the whole stream is regular (P->n_frames frames), -1 otherwise. */
static int par_prepass(struct par_scan* P) {
  const unsigned char* mp3 = P->mp3;
  const size_t n = P->n;
  pdmp3_handle* h = (pdmp3_handle*)calloc(1, sizeof *h);
  int* sky = (int*)malloc((RESERVOIR_BYTES + 2) * sizeof(int));
  if (!h || !sky) { free(h); free(sky); return -1; }
  h->host_only = 1;
  h->vsrc = mp3; h->vfed = 0;
  size_t fed = 0;
  unsigned main_top = 0;
  uint64_t md_end = 0;
  int sky_n = 0, next = 1, rc = -1;
  long long f = 0, ws0[4] = {-1, -1, -1, -1}, ws1[4] = {-1, -1, -1, -1};
  int part = 0;                                         /* whose records: 0 = none, the stream itself */
  long long part_i = 0;
  long long part_end = P->J > 1 ? seg_wait_start(&P->seg[1]) : -3;     /* where part 0 ends */
  if (part_end == -1) goto out;
  for (;;) {
    if (P->abort || P->quit) goto out;                  /* (given up, or the stitcher has left: rc stays "not regular") */
    if (next < P->snap_cap && f == (long long)next * P->sub) {        /* window `next` starts here */
      span_snap* S = &P->snap[next];
      S->frame = f; S->istart = h->istart; S->iend = h->iend; S->processed = h->processed; S->vfed = h->vfed; S->fed = fed;
      S->main_top = main_top; S->sky_n = sky_n;
      S->sky = (int*)malloc((size_t)(sky_n + 1) * sizeof(int));
      if (!S->sky) goto out;
      memcpy(S->sky, sky, (size_t)sky_n * sizeof(int));
      memcpy(S->last_ws0, ws0, sizeof ws0); memcpy(S->last_ws1, ws1, sizeof ws1);
      S->ready = 1;
      __atomic_store_n(&P->published, (long long)next + 1, __ATOMIC_RELEASE);
      next++;
    }
    while (ring_filled(h) < HOP_END_BYTES) {            /* H10 + the CLI's feeds (bulk_drive) */
      const size_t take = n - fed < 4096 ? n - fed : 4096;
      if (!take) { rc = 0; goto out; }                  /* the stream ends here: what is left is dropped, as the CLI drops it */
      if (take > ring_free_logical(h)) goto out;        /* a feed the CLI would drop (H16) */
      if (pdmp3_feed(h, mp3 + fed, take) != PDMP3_OK) goto out;
      fed += take;
    }
    const size_t x = h->vfed - ring_filled(h);
    const unsigned avail = ring_filled(h);
    hop1 own;
    const hop1* q = &own;
    if (part == 0 && part_end >= 0 && x >= (size_t)part_end) {         /* part 0 is through: the hop threads' records from here */
      if (x != (size_t)part_end) goto out;              /* (the guess was not a frame boundary) */
      part = 1; part_i = 0;
      P->t_pre_part0 = now_s() - P->t0;
    }
    if (part == 0) {
      if (hop_parse(mp3, x, &own) != 0) goto out;
      hop_prefetch(mp3, x, own.fb);
    } else {
      for (;;) {
        pre_seg* G = &P->seg[part];
        /* (the records were written by another core, most of them a while ago: they come from its cache or from memory,
         *  a line per four frames -- asked for well ahead, or the loop runs at one such miss per line) */
        if (part_i < __atomic_load_n(&G->count, __ATOMIC_ACQUIRE)) { __builtin_prefetch(&G->rec[part_i + 64], 0, 3); q = &G->rec[part_i++]; break; }
        const int st = __atomic_load_n(&G->state, __ATOMIC_ACQUIRE);
        if (st == SEG_RUNNING) { if (P->abort || P->quit) goto out; const double tw = now_s(); sched_yield(); P->t_pre_wait += now_s() - tw; continue; }
        if (part_i < __atomic_load_n(&G->count, __ATOMIC_ACQUIRE)) continue;   /* (its last records came with the state) */
        if (st != SEG_AT_NEXT || part + 1 >= P->J) goto out;           /* a header the regular path does not take, or a guess that was none */
        part++; part_i = 0;
      }
      if (q->x != x) goto out;
    }
    const unsigned nch = q->nch, nbytes = nch == 1 ? 17 : 32, crc = q->crc, fb = q->fb, begin = q->begin, top = q->top;
    if (fb > avail) goto out;                           /* (a frame the ring does not hold completely: H18) */
    if (begin > main_top) goto out;                     /* reservoir underflow (H9) */
    if (f >= P->rec_cap) goto out;
    hop_rec* r = &P->rec[f];
    md_end += top - begin;
    r->x = (uint32_t)x; r->md_src = (uint32_t)(x + 4 + crc + nbytes); r->md_end = md_end;
    r->fb = (uint16_t)fb; r->begin = (uint16_t)begin; r->top = (uint16_t)top; r->nch = (uint8_t)nch; r->crc = (uint8_t)crc;
    main_top = top;
    while (sky_n && P->rec[sky[sky_n - 1]].top <= main_top) sky_n--;
    sky[sky_n++] = (int)f;
    {                                                   /* (no branches on the stream's bits: they do not predict) */
      const unsigned wsb = q->ws, live = nch == 2 ? 0xfu : 0x5u;
      for (unsigned g = 0; g < 4; g++) {
        const long long on = -(long long)((wsb >> g) & (live >> g) & 1u), off = -(long long)((~wsb >> g) & (live >> g) & 1u);
        ws1[g] = (ws1[g] & ~on) | (f & on);
        ws0[g] = (ws0[g] & ~off) | (f & off);
      }
    }
    h->istart = (h->istart + fb) % INBUF_SIZE;
    h->processed += fb;
    h->l_istart = h->istart; h->l_processed = h->processed;
    if (h->processed > fed) goto out;                   /* (a replayed ring) */
    f++;
  }
out:
  free(h); free(sky);
  pthread_mutex_lock(&P->mu);
  P->n_frames = f;
  P->irregular = rc != 0;
  __atomic_store_n(&P->prepass_done, 1, __ATOMIC_RELEASE);
  pthread_cond_broadcast(&P->cv);
  pthread_mutex_unlock(&P->mu);
  return rc;
}
static void* par_prepass_thread(void* arg) {
  struct par_scan* P = (struct par_scan*)arg;
  const double t0 = now_s();
  (void)par_prepass(P);
  const double dt = now_s() - t0;
  pthread_mutex_lock(&P->mu); P->t_prepass = dt; pthread_mutex_unlock(&P->mu);     /* (read by the stitcher's trace) */
  return NULL;
}

/* `len` bytes of the stream's main data, from position `off` of their concatenation, whose last byte belongs to frame `g` or an earlier one */
static void md_read(const struct par_scan* P, long long g, uint64_t off, unsigned len, uint8_t* out) {
  while (g > 0 && P->rec[g - 1].md_end > off) g--;      /* the frame that holds byte `off` */
  while (len) {
    const hop_rec* r = &P->rec[g];
    const uint64_t start = r->md_end - (uint64_t)(r->top - r->begin);
    const unsigned in = (unsigned)(off - start), have = (unsigned)(r->md_end - off);
    const unsigned k = have < len ? have : len;
    memcpy(out, P->mp3 + r->md_src + in, k);
    out += k; off += k; len -= k; g++;
  }
}

/* a scanner's handle as the sequential scanner's would be in front of frame S->frame */
static void span_init(const struct par_scan* P, const span_snap* S, pdmp3_handle* id) {
  id->vsrc = P->mp3; id->vfed = S->vfed;
  id->istart = S->istart; id->iend = S->iend; id->processed = S->processed;
  id->l_istart = S->istart; id->l_processed = S->processed;
  id->new_header = 1; id->l_new_header = 1; id->need_reset = 0; id->ostart = 0;
  const hop_rec* last = &P->rec[S->frame - 1];
  header_fields(be32(P->mp3 + last->x), &id->hdr);
  id->l_hdr = id->hdr;
  id->last_nch = last->nch;
  /* the reservoir buffer: [0, top) of the newest frame, above it what older frames with larger tops left (sky), zero
   * where no frame ever reached */
  id->main_top = S->main_top;
  memset(id->main_vec, 0, sizeof id->main_vec);
  unsigned covered = 0;
  for (int i = S->sky_n - 1; i >= 0; i--) {
    const long long g = S->sky[i];
    const hop_rec* r = &P->rec[g];
    if (r->top <= covered) continue;
    md_read(P, g, r->md_end - r->top + covered, r->top - covered, id->main_vec + covered);
    covered = r->top;
  }
  /* side-info fields that a frame only sets on one side of win_switch_flag and otherwise leaves as they were (H20) */
  for (unsigned g = 0; g < 4; g++) {
    const unsigned gr = g >> 1, ch = g & 1;
    for (int which = 0; which < 2; which++) {
      const long long f = which ? S->last_ws1[g] : S->last_ws0[g];
      if (f < 0) continue;
      const hop_rec* r = &P->rec[f];
      const uint8_t* v = P->mp3 + r->x + 4 + r->crc;
      const unsigned pos = (r->nch == 1 ? 18u : 20u) + 59u * (gr * r->nch + ch);
      uint8_t tmp[48];
      memcpy(tmp, v, 40); memset(tmp + 40, 0, 8);       /* (side_word reads 8 bytes at a time) */
      const uint64_t xw = side_word(tmp, pos);
      const unsigned y = (unsigned)(xw >> 8) & 0x3fffff;
      if (which) for (unsigned w = 0; w < 3; w++) id->si.subblock_gain[gr][ch][w] = (y >> (6 - 3 * w)) & 7;
      else id->si.table_select[gr][ch][2] = (y >> 7) & 31;
    }
  }
}

typedef struct scanner_arg { struct par_scan* P; int k; int rc; } scanner_arg;
static void* par_scanner(void* arg) {
  scanner_arg* A = (scanner_arg*)arg;
  struct par_scan* P = A->P;
  A->rc = -1;
  par_cache* pc = P->b->pc;                             /* (scanner k's scratch is its own: no lock) */
  const size_t pool_bytes = (size_t)P->sub * RESERVOIR_BYTES + PDMP3_POOL_SLACK_BYTES + 64;
  if (!pc->scan[A->k].wb) pc->scan[A->k].wb = (struct bulk*)calloc(1, sizeof(struct bulk));
  if (!pc->scan[A->k].id) pc->scan[A->k].id = (pdmp3_handle*)malloc(sizeof(pdmp3_handle));
  if (pc->scan[A->k].pool_bytes < pool_bytes) {
    free(pc->scan[A->k].pool);
    pc->scan[A->k].pool = (uint8_t*)malloc(pool_bytes);
    pc->scan[A->k].pool_bytes = pc->scan[A->k].pool ? pool_bytes : 0;
  }
  struct bulk* wb = pc->scan[A->k].wb;
  pdmp3_handle* id = pc->scan[A->k].id;
  uint8_t* pool = pc->scan[A->k].pool;
  if (!wb || !id || !pool) goto done;
  const int whole = P->K == 1;                          /* one scanner: the sequential stage A, from frame 0 to the end */
  for (;;) {
    const long long w = __atomic_fetch_add(&P->next_win, 1, __ATOMIC_RELAXED);
    const double t_take = now_s();
    /* its snapshot: a few tens of microseconds away as a rule (the pre-pass leaves one every 10-25 us and the scanners take
     * them in turn).  The scanner that is next, or next but one, keeps looking; those further ahead sleep 20 us at a time.
     * (Not further than PAR_AHEAD windows in front of the stitcher either: finished windows are memory.) */
    int ready = 0;
    while (!P->abort && !P->quit) {
      const long long pub = __atomic_load_n(&P->published, __ATOMIC_ACQUIRE);
      if (w < pub) {
        if (w < __atomic_load_n(&P->stitched, __ATOMIC_RELAXED) + PAR_AHEAD) { ready = 1; break; }
      } else if (w >= P->snap_cap) break;
      else if (__atomic_load_n(&P->prepass_done, __ATOMIC_ACQUIRE)) {
        /* the stream ended (or went irregular) before this window -- unless its snapshot came with the end */
        if (w < __atomic_load_n(&P->published, __ATOMIC_ACQUIRE)) continue;
        break;
      }
      if (w >= pub && w < pub + 2) {
        if (P->spin) for (int i = 0; i < 64; i++) hp_pause(); else sched_yield();
      } else {
        const struct timespec nap = {0, 20000};
        (void)nanosleep(&nap, NULL);
      }
    }
    const int stop = P->abort;
    if (stop) break;
    if (!ready || (whole && w > 0)) { A->rc = 0; break; }
    memset(id, 0, sizeof *id);
    memset(wb, 0, sizeof *wb);
    id->host_only = 1;
    id->iso = P->b->id->iso;
    id->side_to_bits = 1;
    id->pool_sink = wb;
    wb->id = id; wb->cap = P->sub; wb->bits_mode = 1; wb->pool_mode = 1; wb->win_sink = 1; wb->par = P; wb->pc = pc; wb->priv_pool = pool;
    wb->carry = P->b->carry;
    wb->stream_win = w;
    size_t fed = 0;
    if (w == 0) { pdmp3_open_feed(id); id->vsrc = P->mp3; id->vfed = 0; }
    else {
      const span_snap* S = &P->snap[w];
      span_init(P, S, id);
      fed = S->fed;
      wb->frames = S->frame;
    }
    wb->limit_frames = whole ? 0 : (w + 1) * P->sub;
    wb->tr_take = t_take; wb->tr_begin = now_s();
    size_t done;
    int res;
    while (!bulk_at_limit(wb) && (res = read_impl_sink(id, INBUF_SIZE, &done, wb)) != PDMP3_ERR) {
      if (P->abort || P->quit || wb->failed) break;
      if (id->processed > fed) break;                   /* (the pre-pass will have said so) */
      if (res == PDMP3_NEED_MORE) {
        const size_t take = P->n - fed < 4096 ? P->n - fed : 4096;
        if (!take) break;
        if (take > ring_free_logical(id)) break;
        (void)pdmp3_feed(id, P->mp3 + fed, take);
        fed += take;
      }
    }
    if (wb->failed || P->abort) break;
    if (P->quit) { A->rc = 0; break; }
    if (pw_close_window(wb) != PDMP3_OK) break;         /* (the window, full or -- the stream's last -- partly filled) */
  }
done:
  if (wb && wb->pw_cur) { pw_free(wb->pw_cur); wb->pw_cur = NULL; }
  pthread_mutex_lock(&P->mu);
  if (A->rc != 0) P->abort = 1;
  P->scanners_done++;
  pthread_cond_broadcast(&P->cv);
  pthread_mutex_unlock(&P->mu);
  return NULL;
}

/* Starts the hop threads, the pre-pass and the scanners for `mp3`; NULL when the stream is too short to bother or frame 0 is
 * not where a regular stream has it. */
#define PAR_MIN_WINDOWS 4                /* private windows (tests, a forced split scan); 12 otherwise */
#define PAR_MIN_PART_BYTES (1u << 18)   /* (a file of a few minutes -- 2 to 4 MB -- in six parts: the hop is a chain of cache misses, 0.1 us a frame on a stream that is not in the caches) */
static void par_free(struct par_scan* P) {          /* (the records and the hop threads' arrays are the cache's) */
  if (P->snap) for (long long w = 0; w < P->snap_cap; w++) free(P->snap[w].sky);
  free(P->win); free(P->snap); free(P->args); free(P);
}
static struct par_scan* par_start(struct bulk* b, const unsigned char* mp3, size_t n, int K, int sub, int min_windows, int max_parts) {
  if (K < 1 || n < 4096 || n > 0xfff00000u || (mp3[0] != 0xff) || (mp3[1] & 0xf0) != 0xf0) return NULL;
  frame_header H;
  header_fields(be32(mp3), &H);
  if (H.id != 1 || H.bitrate_index == 0 || H.bitrate_index == 15 || H.sfreq == 3 || H.layer != 3) return NULL;
  const unsigned fb0 = frame_bytes(&H);
  const long long est = (long long)(n / fb0);
  /* sub = 0: the caller leaves the private windows' size to the decoder: 256 frames.  (Until round 5: 1024 for long streams, 512
   * and 256 for files of a few minutes.  The host side takes as long either way; what the shorter ones buy is at the stream's
   * start -- the first window is with the stitcher after 0.05 ms instead of 0.13 -- and in how evenly the engine's windows fill:
   * 137812 frames with the PCM left in HBM, five runs each, interleaved: 31.6 M frames/s with 1024, 32.8 with 512, 34.2 with 256.) */
  if (sub == 0) sub = 256;
  if (sub > b->cap) sub = b->cap;
  if (sub < 1) return NULL;
  const long long est_windows = (est + sub - 1) / sub;
  if (est_windows < min_windows) return NULL;
  if (K > PAR_MAX_SCANNERS) K = PAR_MAX_SCANNERS;
  if (K > est_windows) K = (int)est_windows;
  /* parts of the pre-pass: $PDMP3_BULK_PREPASS_THREADS, else by the host's cores; none shorter than a megabyte */
  const int cores = usable_cpus();
  int J = cores >= 16 ? 6 : cores >= 12 ? 3 : 1;
  const char* ev = getenv("PDMP3_BULK_PREPASS_THREADS");
  size_t min_part = PAR_MIN_PART_BYTES;
  if (ev && atoi(ev) >= 1) { J = atoi(ev); min_part = 16384; }   /* (forced: tests split short streams) */
  if (K == 1) J = 1;                                    /* (one scanner: the sequential stage A, nothing to hurry for) */
  if (J > PAR_MAX_SEGS) J = PAR_MAX_SEGS;
  if (J > max_parts) J = max_parts;
  /* The parts grow: 1 : 1 : 1.5 : 2.25 : ... of the stream.  The pre-pass walks part 0 itself, at the hop's own speed (a chain
   * of cache misses: 30 ns a frame on a stream that is not in the caches), and reads the hop threads' records from there on at
   * 12 ns a frame -- but only as far as they have got: behind a part 0 of a 24th, five EQUAL parts had it follow the first hop
   * thread at that thread's pace through a fifth of the stream (the scanners, and the GPU behind them, waiting for snapshots for
   * the first third of the decode).  A part that is half as long again as the one before is through when the pre-pass gets there.
   * Measured, six interleaved runs each: the hour with the PCM left in HBM 34.5 against 34.4 M frames/s (the pre-pass's own 14 ns a
   * frame bound it either way), the corpus of files of a few minutes with one decoder 9.2 against 7.6 M (a file's part 0 is a
   * fourteenth of it instead of a quarter MB).  PDMP3_BULK_PREPASS_EQUAL=1: the old division. */
  double pw[PAR_MAX_SEGS], pw_sum = 0;
  const char* eq = getenv("PDMP3_BULK_PREPASS_EQUAL");
  const int equal_parts = eq && *eq == '1';
  for (;;) {
    pw_sum = 0;
    for (int j = 0; j < J; j++) { pw[j] = equal_parts ? (j == 0 ? 1.0 : 23.0 / (J > 1 ? J - 1 : 1)) : j < 2 ? 1.0 : pw[j - 1] * 1.5; pw_sum += pw[j]; }
    if (J == 1 || (double)n * pw[0] / pw_sum >= (double)min_part) break;
    J--;
  }
  size_t part_at[PAR_MAX_SEGS + 1];
  { double acc = 0; for (int j = 0; j < J; j++) { part_at[j] = (size_t)((double)n * acc / pw_sum); acc += pw[j]; } part_at[J] = n; }
  struct par_scan* P = (struct par_scan*)calloc(1, sizeof *P);
  if (!P) return NULL;
  P->b = b; P->mp3 = mp3; P->n = n; P->K = K; P->J = J; P->sub = sub; P->t0 = now_s();
  { const char* sp = getenv("PDMP3_BULK_SCAN_SPIN"); P->spin = sp ? atoi(sp) != 0 : cores >= 2 * K + 8; }
  P->published = 1;
  P->one_window = est + est / 16 <= b->cap;
  if (!b->pc) {
    if (!(b->pc = pc_new())) { free(P); return NULL; }
    b->pc->near_gpu = b->near_gpu;
  }
  par_cache* pc = b->pc;
  P->rec_cap = (long long)(n / 96) + 8;                 /* (no Layer III frame is shorter than 96 bytes) */
  if (pc->rec_cap < P->rec_cap) {
    free(pc->rec);
    pc->rec = (hop_rec*)malloc((size_t)P->rec_cap * sizeof(hop_rec));
    pc->rec_cap = pc->rec ? P->rec_cap : 0;
  }
  P->rec = pc->rec;
  P->win_cap = P->rec_cap / sub + 8;
  P->win = (pre_window**)calloc((size_t)P->win_cap, sizeof(pre_window*));
  P->snap_cap = P->win_cap;
  P->snap = (span_snap*)calloc((size_t)P->snap_cap, sizeof(span_snap));
  P->args = (scanner_arg*)calloc((size_t)K, sizeof(scanner_arg));
  int ok = P->rec && P->win && P->snap && P->args;
  for (int j = 1; ok && j < J; j++) {
    pre_seg* S = &P->seg[j];
    const size_t share = part_at[j + 1] - part_at[j];
    S->P = P; S->j = j; S->guess = part_at[j]; S->x_start = -2;
    S->cap = (long long)((share + 65536 + 4096) / 96) + 8;
    if (pc->seg_cap[j] < S->cap) {
      free(pc->seg_rec[j]);
      pc->seg_rec[j] = (hop1*)malloc((size_t)S->cap * sizeof(hop1));
      pc->seg_cap[j] = pc->seg_rec[j] ? S->cap : 0;
    }
    S->rec = pc->seg_rec[j];
    if (!S->rec) ok = 0;
  }
  if (!ok) { par_free(P); return NULL; }
  pthread_mutex_init(&P->mu, NULL); pthread_cond_init(&P->cv, NULL);
  /* (a thread that cannot be had: the ones that run are told to stop and waited for, and the stream goes the one-thread way) */
  int started = 1;
  for (int j = 1; started && j < J; j++) started = crew_run(pc, par_hop_thread, &P->seg[j], &P->hops_left) == 0;
  if (started) started = crew_run(pc, par_prepass_thread, P, &P->jobs_left) == 0;
  for (int k = 0; started && k < K; k++) {
    P->args[k].P = P; P->args[k].k = k; P->args[k].rc = -1;
    started = crew_run(pc, par_scanner, &P->args[k], &P->jobs_left) == 0;
  }
  crew_kick(pc);
  if (!started) {
    pthread_mutex_lock(&P->mu); P->abort = 1; pthread_cond_broadcast(&P->cv); pthread_mutex_unlock(&P->mu);
    crew_wait(pc, &P->jobs_left); crew_wait(pc, &P->hops_left);
    pthread_mutex_destroy(&P->mu); pthread_cond_destroy(&P->cv);
    par_free(P);
    return NULL;
  }
  return P;
}
/* joins the threads and frees everything; returns the pre-pass's verdict: 0 = the stream was regular and complete */
static int par_finish(struct par_scan* P) {
  par_cache* pc = P->b->pc;
  pthread_mutex_lock(&P->mu); P->quit = 1; pthread_cond_broadcast(&P->cv); pthread_mutex_unlock(&P->mu);
  /* (the pre-pass -- its loop and its waits for the hop threads -- and the scanners leave on `quit` or `abort`; the hop
   * threads look at abort only -- they are through long before unless the stitcher gave up early: told to stop once the
   * verdict is taken, which the others' end no longer changes.  A pre-pass that left on `quit` before its end reports
   * "not regular": the verdict below is then -1, as for any stream the stitcher did not see to its end) */
  crew_wait(pc, &P->jobs_left);
  pthread_mutex_lock(&P->mu); const int ok = !P->irregular && !P->abort; P->abort = 1; pthread_cond_broadcast(&P->cv); pthread_mutex_unlock(&P->mu);
  crew_wait(pc, &P->hops_left);
  for (long long w = 0; w < P->win_cap; w++) pw_free(P->win[w]);
  pthread_mutex_destroy(&P->mu); pthread_cond_destroy(&P->cv);
  par_free(P);
  return ok ? 0 : -1;
}
/* next finished window in stream order, or NULL: the stream is complete (*end = 1) or the scan was given up (*end = -1) */
static pre_window* par_next_window_wait(struct par_scan* P, long long w, int* end, double wait_s) {
  pre_window* pw = NULL;
  *end = 0;
  pthread_mutex_lock(&P->mu);
  for (;;) {
    if (P->abort || (P->prepass_done && P->irregular)) { *end = -1; break; }
    if (w < P->win_cap && P->win[w]) { pw = P->win[w]; P->win[w] = NULL; __atomic_store_n(&P->stitched, w + 1, __ATOMIC_RELAXED); break; }
    if (P->prepass_done && w >= (P->n_frames + P->sub - 1) / P->sub) { *end = 1; break; }
    if (P->scanners_done == P->K && P->prepass_done) { *end = -1; break; }      /* (a window is missing: should not happen) */
    if (wait_s < 0) { pthread_cond_wait(&P->cv, &P->mu); continue; }
    if (wait_s == 0) break;                             /* (not there yet: *end stays 0) */
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    ts.tv_nsec += (long)(wait_s * 1e9);
    if (ts.tv_nsec >= 1000000000L) { ts.tv_sec++; ts.tv_nsec -= 1000000000L; }
    (void)pthread_cond_timedwait(&P->cv, &P->mu, &ts);
    wait_s = 0;                                         /* (one more look, then back to the caller) */
  }
  pthread_mutex_unlock(&P->mu);
  return pw;
}
static pre_window* par_next_window(struct par_scan* P, long long w, int* end) { return par_next_window_wait(P, w, end, -1.0); }

static void par_trace_prepass(const struct par_scan* P) {
  fprintf(stderr, "  pre-pass in %d parts: its own part done at %.2f ms, waited %.2f ms for hop threads, through at %.2f ms;", P->J, P->t_pre_part0 * 1e3,
          P->t_pre_wait * 1e3, P->t_prepass * 1e3);
  for (int j = 1; j < P->J; j++) fprintf(stderr, " hop %d: start found %.2f, done %.2f (%lld frames);", j, P->seg[j].t_sync * 1e3, P->seg[j].t_done * 1e3, P->seg[j].count);
  fprintf(stderr, "\n");
}
static void pw_trace(const pre_window* pw, long long w, double t_start, double t0, double t1) {
  fprintf(stderr, "  window %lld: %d frames, taken %.2f, snapshot there %.2f, scanned %.2f, stitcher asked %.2f, got it %.2f ms\n", w, pw->n,
          (pw->t_take - t_start) * 1e3, (pw->t_begin - t_start) * 1e3, (pw->t_done - t_start) * 1e3, (t0 - t_start) * 1e3, (t1 - t_start) * 1e3);
}

/* windows of the engine that are still the GPU's (or the submitter's: handed over, not yet launched) */
static int bulk_in_flight(struct bulk* b) {
  int n = 0;
  for (int slot = 0; slot < BULK_SLOTS; slot++) {
    const bulk_flight* f = &b->flight[slot];
    if (!f->active) continue;
    pthread_mutex_lock(&b->sub_mu);
    const int launched = b->sub_tail > f->sub_seq;
    pthread_mutex_unlock(&b->sub_mu);
    if (!launched || pdmp3_hip_stream_done(b->hs, slot) == 0) n++;
  }
  return n;
}
/* a private window onto the end of the engine's open window: its pool behind what is there, offsets moved accordingly */
static int par_fits(const struct bulk* b, const pre_window* pw, int lim) {
  return (b->bits_n + pw->n <= lim || (!b->bits_n && pw->n <= b->cap)) && b->pool_tail + pw->pool_tail + POOL_ROOM <= b->pool_cap &&
         b->gath_n + pw->gath_n <= b->cap + BULK_GATH_EXTRA * (PAR_MAX_BATCH + 1) && b->slot_arena_n[b->bits_slot] < PAR_MAX_BATCH;
}
static void par_append(struct bulk* b, pre_window* pw) {
  const int at = b->bits_n;
  const uint32_t base = (uint32_t)b->pool_tail;
  memcpy(b->bits_dst + at, pw->bits, (size_t)pw->n * sizeof(pdmp3_frame_bits));
  memcpy(b->flight[b->bits_slot].nch + at, pw->nch, (size_t)pw->n);
  pdmp3_row_desc* d = b->desc_dst + at;
  for (int i = 0; i < pw->n; i++) { d[i] = pw->desc[i]; d[i].row_off += base; d[i].s_off += base; }
  struct pool_copy* g = b->gath_cur + b->gath_n;
  const struct pool_copy* gs = (const struct pool_copy*)pw->gath;
  for (int i = 0; i < pw->gath_n; i++) { g[i] = gs[i]; g[i].dst += base; }
  gather_push(b, b->bits_slot, b->res_dst, g, pw->gath_n);       /* (under way while the window fills) */
  b->g_pushed[b->bits_slot] = 1;
  b->bits_n += pw->n; b->gath_n += pw->gath_n; b->pool_tail += pw->pool_tail;
  b->slot_arena[b->bits_slot][b->slot_arena_n[b->bits_slot]++] = pw->arena; pw->arena = NULL;   /* (the copy list points into it until the submitter is through) */
}

/* The whole-stream decoder's stage A on several threads.  Returns the PCM byte count like bulk_drive, or -3: the stream is
 * not one the split scan takes (nothing has been changed), or -4: it was given up half way (windows of the stream's start
 * may have gone to the engine: the caller drains the pipeline and decodes the stream again the sequential way -- same
 * PCM for the frames both saw, so nothing wrong is ever left in the caller's buffer).
 *
 * The scanners' private windows are SHORT (`sub` frames: the first is there after 0.1 ms) and the engine's windows are
 * made of as many of them as there are when a slot is free, up to the slot's capacity: the GPU has something to do at
 * once, and once it is busy the windows grow by themselves to the size at which the device's Huffman stage fills the
 * chip (k_unpack: one workgroup per 16 frames, two per CU -- 8192 frames).  While two windows or more are still the
 * GPU's, a window that is not full waits for more. */
#define PAR_NOT_TAKEN (-3)
#define PAR_GIVEN_UP (-4)
static atomic_int g_par_active;                    /* split scans under way in this process */
static long long par_drive(struct bulk* b, const unsigned char* mp3, size_t n, int K) {
  const double t_start = now_s();
  int sub = 0;                                          /* (by the stream's length: par_start) */
  const char* se = getenv("PDMP3_BULK_SUB_FRAMES");
  if (se && atoi(se) >= 1) sub = atoi(se) < b->cap ? atoi(se) : b->cap;
  /* streams from 12 private windows on (3072 frames: a scan of 0.25 ms on one thread); a forced split scan: from 4 */
  /* several decoders of one process at it at once (a corpus, a decoder per few files): they share the host's cores --
   * the second takes half the scanners, the third and fourth a third and a quarter (two at least), and only the first
   * has hop threads */
  const int others = atomic_fetch_add(&g_par_active, 1);
  if (others > 0) { K = K / (others + 1); if (K < 2) K = 2; }
  struct par_scan* P = par_start(b, mp3, n, K, sub, b->scan_forced ? PAR_MIN_WINDOWS : 3 * PAR_MIN_WINDOWS, others > 0 ? 1 : PAR_MAX_SEGS);
  if (!P) { atomic_fetch_sub(&g_par_active, 1); return PAR_NOT_TAKEN; }
  sub = P->sub;
  K = P->K;
  long long total = 0, frames = 0, w = 0;
  int end = 0, engine_ok = 1, n_windows = 0, gave_up = 0;
  double t_win = 0, t_open = 0, t_fill = 0, t_more = 0;
  const char* tr = getenv("PDMP3_BULK_TRACE");
  const int trace2 = tr && atoi(tr) >= 2;
  b->trace2 = trace2; b->tr_t0 = t_start;
  pre_window* held = NULL;                              /* taken from the scanners, did not fit the window before */
  for (;;) {
    const double t0 = now_s();
    const int was_held = held != NULL;                  /* (taken from the scanners -- and counted -- when it did not fit the window before) */
    pre_window* pw = held ? held : par_next_window(P, w, &end);
    held = NULL;
    const double t1 = now_s();
    t_win += t1 - t0;
    if (!pw) break;
    if (!was_held) {
      if (trace2) pw_trace(pw, w, t_start, t0, t1);
      w++;
    }
    const int opened = engine_ok && bits_open_window(b) == PDMP3_OK;
    const double t2 = now_s();
    t_open += t2 - t1;
    /* (Tried in round 5: a short LAST window -- the stream's end known from the pre-pass, the window that would leave less than
     *  2048 frames behind stopping that far short of it -- so that the caller waits for a shorter chain of kernels at the end:
     *  32.2 against 33.8 M frames/s without, five interleaved runs.  The GPU is the bound by then, and two windows cost it more
     *  than one.  And capped FIRST windows -- cap / 8, cap / 8, cap / 4, cap / 2, as the one-thread scan has them -- so that the
     *  third window does not wait until it is full while two tiny ones are the GPU's: 33.4 against 33.7 M, six runs each.) */
    /* (a destination in host memory: windows of `target` frames -- 4096 unless the caller named a size -- or, for a stream that
     *  fits a slot, the slot: bulk_decode_impl's cur_target) */
    const int lim = b->pcm_pinned == 2 || b->cur_target <= 0 || b->cur_target > b->cap ? b->cap : b->cur_target;
    if (opened && par_fits(b, pw, lim)) {
      par_append(b, pw);
      for (int i = 0; i < pw->n; i++) total += 2304 * pw->nch[i];
      frames += pw->n;
      pw_free(pw);
      const double t3 = now_s();
      t_fill += t3 - t2;
      while (b->bits_n < lim) {                         /* what else is there, or worth waiting for */
        int e2;
        pre_window* more = par_next_window_wait(P, w, &e2, 0);
        /* (a stream that fits one slot goes up as ONE window -- a file of a few minutes: every window costs the GPU its
         *  150 us whatever it holds, and in a corpus the GPU has the file before to work on meanwhile) */
        const int hold = e2 == 0 && !more && (P->one_window || bulk_in_flight(b) >= 2);
        if (hold) more = par_next_window_wait(P, w, &e2, 50e-6);
        if (!more) { if (e2 == 0 && (P->one_window || bulk_in_flight(b) >= 2)) continue; break; }
        if (trace2) pw_trace(more, w, t_start, t3, now_s());
        w++;
        if (!par_fits(b, more, lim)) { held = more; break; }
        par_append(b, more);
        for (int i = 0; i < more->n; i++) total += 2304 * more->nch[i];
        frames += more->n;
        pw_free(more);
      }
      const double t4 = now_s();
      t_more += t4 - t3;
      b->frames = frames;
      if (trace2) fprintf(stderr, "  -> window of %d frames to slot %d at %.2f ms\n", b->bits_n, b->bits_slot, (t4 - t_start) * 1e3);
      n_windows++;
      if (bits_close_window(b) != PDMP3_OK) engine_ok = 0;
      t_fill += now_s() - t4;
    } else if (opened) {
      /* a private window that does not fit an EMPTY window of the engine (slots of a handful of frames: the window's pool has
       * no room for the reservoir image a private window starts with): not a stream for the split scan -- the one-thread scan
       * decodes it again from its first frame (PAR_GIVEN_UP), nothing is wrong with the engine */
      pw_free(pw);
      gave_up = 1;
      pthread_mutex_lock(&P->mu); P->abort = 1; pthread_cond_broadcast(&P->cv); pthread_mutex_unlock(&P->mu);
      break;
    } else { engine_ok = 0; pw_free(pw); }
    if (!engine_ok) { pthread_mutex_lock(&P->mu); P->abort = 1; pthread_cond_broadcast(&P->cv); pthread_mutex_unlock(&P->mu); break; }
  }
  pw_free(held);
  pthread_mutex_lock(&P->mu); const double t_pre = P->t_prepass; pthread_mutex_unlock(&P->mu);
  const long long nf = P->n_frames;
  if (trace2) par_trace_prepass(P);
  uint32_t last_hw = 0;
  if (end == 1 && nf > 0) last_hw = be32(mp3 + P->rec[nf - 1].x);
  const int fin = par_finish(P);
  const int ok = fin == 0 && end == 1 && engine_ok && frames == nf;
  atomic_fetch_sub(&g_par_active, 1);
  if (!ok && tr) fprintf(stderr, "bulk trace: split scan given up: pre-pass verdict %d, end %d, engine %d, frames %lld of %lld, %d windows%s\n", fin, end, engine_ok, frames, nf, n_windows,
                         gave_up ? " (a private window larger than an empty window of the engine)" : "");
  if (!engine_ok) { b->failed = 1; return -1; }
  if (!ok) { b->par_given_up++; return PAR_GIVEN_UP; }
  b->par_taken++;
  if (nf > 0) { header_fields(last_hw, &b->id->hdr); b->id->l_hdr = b->id->hdr; }
  if (getenv("PDMP3_BULK_TRACE"))
    fprintf(stderr, "bulk trace: split scan, %d scanners, %lld frames in private windows of %d, %d windows to the engine, pre-pass %.2f ms; stitch %.2f ms = "
            "waiting for the first private window of each %.2f + for slots %.2f + for more of them %.2f + filling and closing %.2f (this stream)\n",
            K, nf, sub, n_windows, t_pre * 1e3, (now_s() - t_start) * 1e3, t_win * 1e3, t_open * 1e3, t_more * 1e3, t_fill * 1e3);
  return total;
}

/* Host tests (no engine): the split scan's windows of a stream, in order, as one byte string -- per window n, the copy
 * list's length, the pool's fill, the side-info records, the row descriptors, the channel counts and the copy list (an
 * entry whose source is the stream as its offset, one whose source is the window's arena as its bytes).  K = 1 is the
 * unchanged stage-A code on one scanner from frame 0; K > 1 must give the same string.  Returns its length, -3 / -4 like
 * par_drive, -1 when `out` is too small. */
/* (the hook's stand-in for a decoder lives as long as the process, as a decoder's memory does from stream to stream;
 * one caller at a time) */
static pthread_mutex_t g_hook_mu = PTHREAD_MUTEX_INITIALIZER;
static struct bulk* g_hook_b;
static struct bulk* hook_get(void) {
  pthread_mutex_lock(&g_hook_mu);
  if (!g_hook_b) {
    g_hook_b = (struct bulk*)calloc(1, sizeof *g_hook_b);
    if (g_hook_b && !(g_hook_b->id = (pdmp3_handle*)calloc(1, sizeof *g_hook_b->id))) { free(g_hook_b); g_hook_b = NULL; }
  }
  if (!g_hook_b) pthread_mutex_unlock(&g_hook_mu);
  return g_hook_b;
}
static void hook_put(struct bulk* b) { (void)b; pthread_mutex_unlock(&g_hook_mu); }
long long pdmp3_amd_test_split_scan(const unsigned char* mp3, size_t n, int window_frames, int K, unsigned iso,
                                    unsigned char* out, size_t out_cap, long long* frames) {
  pthread_once(&g_lut_once, build_luts);
  struct bulk* b = hook_get();
  if (!b) return -1;
  b->cap = b->target = window_frames > 0 ? window_frames : 2048;
  b->bits_mode = 1; b->pool_mode = 1;             /* (what the scanners' sinks are: the window schedule depends on it) */
  b->id->iso = iso;
  const double t_start = now_s();
  struct par_scan* P = par_start(b, mp3, n, K, b->cap, PAR_MIN_WINDOWS, PAR_MAX_SEGS);
  if (!P) { hook_put(b); return PAR_NOT_TAKEN; }
  size_t o = 0;
  int end = 0, fit = 1;
  long long nf = 0;
#define PUT(ptr, len) do { if (o + (len) <= out_cap) memcpy(out + o, (ptr), (len)); else fit = 0; o += (len); } while (0)
  const char* tr = getenv("PDMP3_BULK_TRACE");
  const int trace2 = tr && atoi(tr) >= 2;
  for (long long w = 0;; w++) {
    const double t0 = trace2 ? now_s() : 0;
    pre_window* pw = par_next_window(P, w, &end);
    if (!pw) break;
    if (trace2) pw_trace(pw, w, t_start, t0, now_s());
    const int32_t hd[2] = {pw->n, pw->gath_n};
    const uint64_t pt = pw->pool_tail;
    PUT(hd, sizeof hd); PUT(&pt, sizeof pt);
    PUT(pw->bits, (size_t)pw->n * sizeof(pdmp3_frame_bits)); PUT(pw->desc, (size_t)pw->n * sizeof(pdmp3_row_desc)); PUT(pw->nch, (size_t)pw->n);
    const struct pool_copy* g = (const struct pool_copy*)pw->gath;
    for (int i = 0; i < pw->gath_n; i++) {
      const int lit = !(g[i].src >= mp3 && g[i].src < mp3 + n);
      const uint32_t e[3] = {(uint32_t)lit, g[i].dst, g[i].n};
      PUT(e, sizeof e);
      if (lit) PUT(g[i].src, g[i].n);
      else { const uint64_t off = (uint64_t)(g[i].src - mp3); PUT(&off, sizeof off); }
    }
    nf += pw->n;
    pw_free(pw);
  }
#undef PUT
  const long long pf = P->n_frames;
  if (trace2) par_trace_prepass(P);
  const int ok = par_finish(P) == 0 && end == 1 && nf == pf;
  hook_put(b);
  if (frames) *frames = nf;
  if (!ok) return PAR_GIVEN_UP;
  return fit ? (long long)o : -1;
}

void pdmp3_amd_bulk_delete(struct bulk* b) {
  if (!b) return;
  if (b->lsf_alt) { pdmp3_amd_bulk_delete(b->lsf_alt); b->lsf_alt = NULL; }
  if (b->th) {
    pthread_mutex_lock(&b->mu);
    b->quit = 1;
    pthread_cond_broadcast(&b->cv_work);
    pthread_mutex_unlock(&b->mu);
    for (int i = 0; i < b->nth; i++) pthread_join(b->th[i], NULL);
    free(b->th);
    pthread_mutex_destroy(&b->mu); pthread_cond_destroy(&b->cv_work); pthread_cond_destroy(&b->cv_done);
  }
  if (b->sub_started) {
    pthread_mutex_lock(&b->sub_mu);
    b->sub_quit = 1;
    pthread_cond_signal(&b->sub_cv);
    pthread_mutex_unlock(&b->sub_mu);
    pthread_join(b->sub_th, NULL);
    pthread_mutex_destroy(&b->sub_mu); pthread_cond_destroy(&b->sub_cv); pthread_cond_destroy(&b->sub_done_cv);
    pthread_mutex_lock(&b->gh_mu);
    b->gh_quit = 1;
    pthread_cond_broadcast(&b->gh_cv);
    pthread_mutex_unlock(&b->gh_mu);
    for (int i = 0; i < b->gh_n; i++) pthread_join(b->gh_th[i], NULL);
    pthread_mutex_destroy(&b->gh_mu); pthread_cond_destroy(&b->gh_cv); pthread_cond_destroy(&b->gh_done_cv);
  }
  for (int i = 0; i < 2; i++) { free(b->win[i].jobs); free(b->win[i].outs); }
  for (int i = 0; i < BULK_SLOTS; i++) {
    free(b->flight[i].nch); free(b->gath[i]);
    for (int k = 0; k < b->slot_arena_n[i]; k++) free(b->slot_arena[i][k]);
  }
  if (b->hs) pdmp3_hip_stream_destroy(b->hs);
  pc_free(b->pc);
  free(b->id);
  free(b);
}

/* CPUs this process can keep busy: the affinity mask, capped by the cgroup CPU quota (containers often show every
 * CPU of the host but run under a quota of a few; threads beyond it only get throttled) */
static int usable_cpus(void) {
  long n = sysconf(_SC_NPROCESSORS_ONLN);
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0 && CPU_COUNT(&set) < n) n = CPU_COUNT(&set);
  FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r");
  if (f) {
    char q[32];
    long long period = 0;
    if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") && period > 0) {
      const long long quota = atoll(q);
      const long c = (long)((quota + period / 2) / period);
      if (c >= 1 && c < n) n = c;
    }
    fclose(f);
  }
  return n < 1 ? 1 : (int)n;
}

/* The CPUs next to the GPU (/sys/bus/pci/devices/<address>/local_cpulist), as far as this process may run on them.
 * The GPU boxes are two-socket machines: the decoder's pinned buffers sit behind one socket's memory controllers and
 * the GPU behind one socket's PCIe root; helper threads that wander to the other socket copy every PCM byte across the
 * socket link and back (measured on such a box, PCM to pageable memory, the same build run after run: 6.1 .. 9.1 M
 * frames/s; the round-4 driver run's 6.4 M against 9.7 M elsewhere).  So the decoder's own threads -- copy pool,
 * submitter, gather helpers, the split scan's crew -- are kept on the GPU's node, and its pinned buffers are allocated
 * from a thread that is there.  PDMP3_BULK_NUMA=0: leave everything to the scheduler.  Returns the number of CPUs. */
static int gpu_local_cpus(pdmp3_hip_ctx* ctx, cpu_set_t* out) {
  CPU_ZERO(out);
  const char* e = getenv("PDMP3_BULK_NUMA");
  if (e && *e == '0') return 0;
  char bdf[64], path[160], list[1024];
  if (pdmp3_hip_pci_bus_id(ctx, bdf, (int)sizeof bdf) != PDMP3_HIP_OK || !bdf[0]) return 0;
  for (char* p = bdf; *p; p++) if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');
  snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/local_cpulist", bdf);
  FILE* f = fopen(path, "r");
  if (!f) return 0;
  const int ok = fgets(list, sizeof list, f) != NULL;
  fclose(f);
  if (!ok) return 0;
  cpu_set_t mine;
  if (sched_getaffinity(0, sizeof mine, &mine) != 0) return 0;
  for (const char* p = list; *p;) {                     /* "0-63,128-191" */
    char* end;
    const long a = strtol(p, &end, 10);
    if (end == p) break;
    long b = a;
    p = end;
    if (*p == '-') { b = strtol(p + 1, &end, 10); p = end; }
    for (long c = a; c <= b && c < CPU_SETSIZE; c++) if (c >= 0 && CPU_ISSET((int)c, &mine)) CPU_SET((int)c, out);
    while (*p == ',' || *p == ' ' || *p == '\n') p++;
  }
  const int n = CPU_COUNT(out);
  if (n == CPU_COUNT(&mine)) { CPU_ZERO(out); return 0; }           /* one node, or already bound: nothing to do */
  return n;
}
static void bind_thread(pthread_t t, const cpu_set_t* set) { if (CPU_COUNT(set) > 0) (void)pthread_setaffinity_np(t, sizeof *set, set); }

/* threads <= 0: one per usable CPU (at most 64); window_frames <= 0: 2048.  with_engine = 0 gives a
 * parse-only decoder (host tests on machines without a GPU). */
static struct bulk* bulk_new(int threads, int window_frames, int with_engine, int bits_mode, int device) {
  pthread_once(&g_lut_once, build_luts);
  const int window_arg = window_frames;
  if (threads <= 0) {
    const int c = usable_cpus();
    threads = c > 64 ? 64 : c;
    /* (the pool only copies PCM out: at most 4 threads -- the scan, the submitter and on a device destination the split scan's
     *  threads want cores too, and more copy threads take memory bandwidth from the scanning thread, which bounds the
     *  host-memory destinations: the hour to pageable memory, five interleaved runs, 3 / 4 / 6 / 8 threads: 9.0 / 9.0 / 8.2 /
     *  8.1 M frames/s; six was the default for a while in round 5) */
    if (bits_mode && threads > 4) threads = 4;
  }
  /* frames per GPU batch.  Bits mode: k_unpack is bound by the length of one lane's chain, not by throughput, and two of
   * its workgroups (16 frames each) fit a CU: 8192 frames fill the chip once, 68 -> 84 us against 4096.  The slots hold
   * 8192; the split scan (PCM left on the device) fills them as far as its scanners have got when a slot is free, the
   * one-thread scan (host destinations: the PCM's way home over PCIe is what bounds those, and a window's PCM leaves when
   * the window is done) closes its windows at 4096 -- measured, pinned / pageable: 10.6 / 9.5 M frames/s against 9.6 / 8.1
   * with windows of 8192.  A window size the caller names is both. */
  int target = window_frames;
  if (window_frames <= 0) { window_frames = bits_mode ? 8192 : 2048; target = bits_mode ? 4096 : 2048; }
  if (window_frames > 32768) window_frames = target = 32768;
  struct bulk* b = (struct bulk*)calloc(1, sizeof *b);
  if (!b) return NULL;
  b->cap = window_frames;
  b->target = target;
  b->device = device;
  b->window_arg = window_arg;
  {
    /* split scan (par_drive): 12 scanners where the process has 32 CPUs, 8 with 16 (they live for the few milliseconds of a
     * stream's scan; with the PCM left in HBM the scanners, the upload and the kernels all take about 0.2 ms per 8192 frames, and
     * four more keep the scan off the critical path: 33.6 against 31.5-33.5 M frames/s, less spread), fewer on smaller quotas, none below 6 CPUs; PDMP3_BULK_SCAN_THREADS = 0 .. PAR_MAX_SCANNERS (16) overrides (0: one thread, as before) */
    const int c = usable_cpus();
    const char* e = getenv("PDMP3_BULK_SCAN_THREADS");
    b->scan_threads = e ? atoi(e) : (c >= 32 ? 12 : c >= 16 ? 8 : c >= 12 ? 4 : c >= 6 ? 2 : 0);
    b->scan_forced = e != NULL;
    if (b->scan_threads < 0) b->scan_threads = 0;
    if (b->scan_threads > PAR_MAX_SCANNERS) b->scan_threads = PAR_MAX_SCANNERS;
  }
  b->bits_mode = bits_mode;
  b->id = (pdmp3_handle*)calloc(1, sizeof *b->id);
  if (!b->id) { free(b); return NULL; }
  b->id->host_only = 1;
  for (int i = 0; i < 2 && !bits_mode; i++) {
    b->win[i].jobs = (frame_job*)malloc((size_t)b->cap * sizeof(frame_job));
    b->win[i].outs = (main_out*)malloc((size_t)b->cap * sizeof(main_out));
    if (!b->win[i].jobs || !b->win[i].outs) { pdmp3_amd_bulk_delete(b); return NULL; }
  }
  if (with_engine) {
    pdmp3_hip_ctx* ctx = shared_ctx_on(device);
    int made = PDMP3_HIP_EINVAL;
    if (ctx) {
      /* the slots' pinned buffers: allocated (and first touched, by the pinning) from a thread on the GPU's node */
      cpu_set_t before;
      const int rebind = gpu_local_cpus(ctx, &b->near_gpu) > 0 && sched_getaffinity(0, sizeof before, &before) == 0;
      if (rebind) (void)sched_setaffinity(0, sizeof b->near_gpu, &b->near_gpu);
      /* (the scanners / hop threads / gather helpers are SIZED by the whole affinity mask and BOUND to the
       *  GPU's node: on a host whose node holds half of the usable CPUs that oversubscribes the node.  Sizing them by the
       *  node was tried in round 6 and not kept without a measurement on such a host: on the 16-CPU quota of the test boxes
       *  it would cut the scanners from 8 to 2.  PDMP3_BULK_SCAN_THREADS / _PREPASS_THREADS / _GATHER_THREADS size them.) */
      made = pdmp3_hip_stream_create_slots(ctx, b->cap, BULK_SLOTS, &b->hs);
      if (rebind) (void)sched_setaffinity(0, sizeof before, &before);
    }
    if (made != PDMP3_HIP_OK) {
      fprintf(stderr, "pdmp3: no MI355X transform engine: %s\n", pdmp3_hip_last_error());
      b->hs = NULL;
      pdmp3_amd_bulk_delete(b);
      return NULL;
    }
    for (int i = 0; i < BULK_SLOTS; i++) {
      b->flight[i].nch = (uint8_t*)malloc((size_t)b->cap);
      b->gath[i] = (struct pool_copy*)malloc(((size_t)b->cap + BULK_GATH_EXTRA * (PAR_MAX_BATCH + 1)) * sizeof(struct pool_copy));
      if (!b->flight[i].nch || !b->gath[i]) { pdmp3_amd_bulk_delete(b); return NULL; }
    }
    if (bits_mode) {
      const char* snap = getenv("PDMP3_BULK_SNAPSHOT_ROWS");          /* 1: the 2064-byte-per-frame form of the input */
      b->pool_mode = !(snap && *snap && *snap != '0');
      pthread_mutex_init(&b->sub_mu, NULL); pthread_cond_init(&b->sub_cv, NULL); pthread_cond_init(&b->sub_done_cv, NULL);
      if (pthread_create(&b->sub_th, NULL, bulk_submitter, b) != 0) { pdmp3_amd_bulk_delete(b); return NULL; }
      b->sub_started = 1;
      bind_thread(b->sub_th, &b->near_gpu);
      pthread_mutex_init(&b->gh_mu, NULL); pthread_cond_init(&b->gh_cv, NULL); pthread_cond_init(&b->gh_done_cv, NULL);
      {
        const char* ge = getenv("PDMP3_BULK_GATHER_THREADS");      /* helpers for the windows' main-data copies (0 .. 8) */
        int want = ge ? atoi(ge) : (b->scan_threads >= 8 ? 6 : b->scan_threads > 0 ? 3 : 0);   /* (4 instead of 6: the same with the PCM left in HBM, 9.0 against 9.7 M frames/s to pinned memory) */
        if (want > GATHER_MAX_HELPERS) want = GATHER_MAX_HELPERS;
        for (b->gh_n = 0; b->gh_n < want; b->gh_n++)
          if (pthread_create(&b->gh_th[b->gh_n], NULL, gather_helper, b) != 0) break;
          else bind_thread(b->gh_th[b->gh_n], &b->near_gpu);
      }
    }
  }
  pthread_mutex_init(&b->mu, NULL); pthread_cond_init(&b->cv_work, NULL); pthread_cond_init(&b->cv_done, NULL);
  b->th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
  if (!b->th) { pdmp3_amd_bulk_delete(b); return NULL; }
  for (b->nth = 0; b->nth < threads; b->nth++)
    if (pthread_create(&b->th[b->nth], NULL, bulk_worker, b) != 0) break;
    else if (bits_mode) bind_thread(b->th[b->nth], &b->near_gpu);   /* (a host-Huffman pool is compute: it takes every socket) */
  if (b->nth == 0) { pdmp3_amd_bulk_delete(b); return NULL; }
  return b;
}

/* default: Huffman decoding on the device; PDMP3_BULK_HOST_HUFFMAN=1 (or _new_ex) keeps it on the host pool */
struct bulk* pdmp3_amd_bulk_new_on(int threads, int window_frames, int host_huffman, int device) {
  return bulk_new(threads, window_frames, 1, !host_huffman, device);
}
struct bulk* pdmp3_amd_bulk_new_ex(int threads, int window_frames, int host_huffman) {
  return bulk_new(threads, window_frames, 1, !host_huffman, default_device());
}
struct bulk* pdmp3_amd_bulk_new(int threads, int window_frames) {
  const char* e = getenv("PDMP3_BULK_HOST_HUFFMAN");
  return bulk_new(threads, window_frames, 1, !(e && *e && *e != '0'), default_device());
}
struct bulk* pdmp3_amd_bulk_new_parse_only(int threads, int window_frames) { return bulk_new(threads, window_frames, 0, 0, 0); }
struct bulk* pdmp3_amd_bulk_new_parse_bits(void) { return bulk_new(1, 1, 0, 1, 0); }
int pdmp3_amd_bulk_threads(const struct bulk* b) { return b ? b->nth : 0; }
/* streams this decoder's split scan (several scanner threads: device destinations, or PDMP3_BULK_SCAN_THREADS) decoded to
 * their end, and streams it gave up half way and decoded again with the one-thread scan (irregular ones: resync, tags,
 * truncation in the middle of the ring's cadence) -- same PCM either way; for tests and for whoever wonders about the rate */
void pdmp3_amd_bulk_split_scans(const struct bulk* b, long long* taken, long long* given_up) {
  if (taken) *taken = b ? b->par_taken : 0;
  if (given_up) *given_up = b ? b->par_given_up : 0;
}
/* the ISO-correct switches (include/pdmp3.h: pdmp3_amd_set_quirks) for the streams this decoder is given from now on */
int pdmp3_amd_bulk_set_quirks(struct bulk* b, unsigned iso_mask) { return b ? pdmp3_amd_set_quirks(b->id, iso_mask) : PDMP3_ERR; }

static void bulk_begin(struct bulk* b) {
  pdmp3_handle* id = b->id;
  /* a fresh handle per stream -- unless the caller is pdmp3(), which decodes all its files with ONE handle: parse
   * state left by the previous file shows in the next one, so it is kept (b->carry) */
  if (!b->carry) {
    const unsigned iso = id->iso;                 /* (a setting of the decoder, not parse state: pdmp3_amd_bulk_set_quirks) */
    memset(id, 0, sizeof *id);
    id->host_only = 1;
    id->iso = iso;
  }
  id->pool_sink = b->pool_mode ? b : NULL;
  id->side_to_bits = b->bits_mode && !getenv("PDMP3_BULK_SLOW_SIDE_INFO");
  /* (windows, flights, a running copy job: the pipeline keeps going across streams) */
  b->win[b->cur].n = 0;
  b->frames = 0; b->pcm_emitted = 0; b->count_only = 0; b->failed = 0;
  b->bits_open = 0; b->bits_n = 0;            /* (a failed submit stays failed: sub_rc is sticky) */
  b->stream_win = 0;
}

/* frames and PCM bytes pdmp3() would produce for this stream: stage A alone */
long long pdmp3_amd_scan_buffer_iso(const unsigned char* mp3, size_t n, unsigned iso_mask, long long* frames);
long long pdmp3_amd_scan_buffer(const unsigned char* mp3, size_t n, long long* frames) { return pdmp3_amd_scan_buffer_iso(mp3, n, 0, frames); }
/* ... with the switches a decoder was given (pdmp3_amd_bulk_set_quirks): only PDMP3_ISO_LSF changes what a scan counts */
long long pdmp3_amd_scan_buffer_iso(const unsigned char* mp3, size_t n, unsigned iso_mask, long long* frames) {
  pthread_once(&g_lut_once, build_luts);
  struct bulk b;
  memset(&b, 0, sizeof b);
  b.id = (pdmp3_handle*)calloc(1, sizeof *b.id);
  if (!b.id) return -1;
  b.id->host_only = 1;
  b.id->iso = iso_mask & (PDMP3_ISO_ALL | PDMP3_ISO_LSF);
  b.count_only = 1;
  const long long total = bulk_drive(&b, mp3 ? mp3 : (const unsigned char*)"", mp3 ? n : 0);
  if (frames) *frames = b.frames;
  free(b.id);
  return total;                                   /* PDMP3_BULK_REPLAY (-2) passes through */
}

/* everything submitted so far is decoded and its PCM in caller memory */
static int bulk_drain(struct bulk* b) {
  int ok = sub_drain(b) == PDMP3_OK;
  bulk_wait_b(b);
  for (int i = 0; i < BULK_SLOTS && ok; i++) {         /* what is still on the GPU: the pool copies it out */
    const unsigned char* src; unsigned char* dst; size_t nbytes;
    ok = bulk_collect(b, i, &src, &dst, &nbytes) == PDMP3_OK;
    if (ok && nbytes) { bulk_start_b(b, NULL, src, dst, nbytes); bulk_wait_b(b); }
  }
  bulk_wait_b(b);
  b->in_b = NULL;
  if (!ok) for (int i = 0; i < BULK_SLOTS; i++) { (void)pdmp3_hip_stream_wait(b->hs, i); b->flight[i].active = 0; }
  return ok ? PDMP3_OK : PDMP3_ERR;
}

/* scan + submit one stream; with drain = 0 its last windows may still be on their way when this returns */
static long long bulk_decode_impl(struct bulk* b, const unsigned char* mp3, size_t n, unsigned char* pcm, size_t pcm_cap,
                                  long* rate, int* channels, int drain) {
  if (!b || !b->hs || (!mp3 && n) || (!pcm && pcm_cap)) return -1;
  /* PDMP3_ISO_LSF on a decoder whose Huffman stage is on the device: that stage reads MPEG-1 side info only, so a stream
   * that opens with an MPEG-2 LSF / 2.5 header goes through a host-Huffman decoder this one keeps for the purpose
   * (same device, same threads and window, same switches; the call is synchronous then) */
  if (b->bits_mode && (b->id->iso & PDMP3_ISO_LSF) && n >= 4 && mp3[0] == 0xff && (mp3[1] & 0xe0) == 0xe0 && (mp3[1] & 0x18) != 0x18 && (mp3[1] & 0x18) != 0x08) {
    if (!b->lsf_alt) b->lsf_alt = bulk_new(b->nth, b->window_arg, 1, 0, b->device);
    if (!b->lsf_alt) return -1;
    if (bulk_drain(b) != PDMP3_OK) return -1;     /* (this decoder's own streams first: the PCM destinations may overlap) */
    b->lsf_alt->id->iso = b->id->iso;
    return bulk_decode_impl(b->lsf_alt, mp3, n, pcm, pcm_cap, rate, channels, 1);
  }
  bulk_begin(b);
  /* Device Huffman: nothing to reset on the host side -- the stream's first frame carries PDMP3_FR_RESET (synthesis
   * state) and, unless parse state is carried over (pdmp3()), PDMP3_FR_NEWSTREAM (scalefactors / count1), so
   * streams follow each other through the pipeline without a stop.  Host Huffman: the pipeline is idle here. */
  if (!b->bits_mode && !b->carry && pdmp3_hip_stream_reset(b->hs) != PDMP3_HIP_OK) return -1;
  b->pcm = pcm; b->pcm_cap = pcm_cap;
  b->pcm_pinned = pcm_cap ? pdmp3_hip_host_is_pinned(pcm, pcm_cap) : 0;      /* 1 pinned host memory, 2 device memory */
  const double t_in = now_s();
  long long total = PAR_NOT_TAKEN;
  /* short first windows (win_ramp) only for streams long enough to gain from them: a file of a few minutes would go up
   * in five windows instead of two, and a window costs the GPU 150 us whatever its size (C4 corpus to pageable memory:
   * 5.2 -> 4.0 M frames/s with the ramp on every file) */
  b->ramp_on = 0;
  b->cur_target = b->target;
  if (n >= 4 && mp3[0] == 0xff && (mp3[1] & 0xf0) == 0xf0) {
    frame_header H0;
    header_fields(((uint32_t)mp3[0] << 24) | ((uint32_t)mp3[1] << 16) | ((uint32_t)mp3[2] << 8) | mp3[3], &H0);
    if (H0.id == 1 && H0.layer == 3 && H0.bitrate_index != 0 && H0.bitrate_index != 15 && H0.sfreq != 3)
    {
      const long long est0 = (long long)(n / frame_bytes(&H0));
      b->ramp_on = est0 >= 8LL * b->target;
      /* a file of a few minutes that fits one slot goes up as one window instead of a full one and a remainder (4096 + 32
       * frames, say: the remainder costs the GPU as much as the full one) */
      if (b->bits_mode && est0 > b->target && est0 + est0 / 16 <= b->cap) b->cur_target = b->cap;
    }
  }
  /* The split scan: with the PCM left in device memory the scan is the bound without it (13 -> 18 M frames/s in round 4, 35 now).
   * Towards host memory the link bounds the pipeline -- 12.7 ms for the hour's PCM at 50 GB/s -- but the ONE-thread scan takes
   * 12.4 ms of its own beside it and loses to every disturbance (pinned 8.9-9.8 M frames/s, pageable 8.2-9.6 on one box): since
   * round 5 host destinations take the split scan as well, with four scanners and the engine's windows closed at 4096 frames (a
   * window's PCM leaves when the window is done): pinned 10.2-10.4, pageable 9.35-9.54, five interleaved runs.  (Round 4's split
   * scan, with its pre-pass on the scanners' mutex and eight scanners, LOST there: pinned 10.4 -> 8.3.)
   * (Decoders whose windows the caller made shorter than 1024 frames, and hosts with fewer than 12 usable CPUs -- two scanners by
   * default: measured nowhere --, keep the one-thread scan for host destinations.)
   * PDMP3_BULK_SCAN_THREADS=0: the one-thread scan everywhere. */
  if (b->bits_mode && b->pool_mode && !b->carry && b->scan_threads > 0 &&
      (b->pcm_pinned == 2 || b->scan_forced || (b->target >= 1024 && b->scan_threads >= 4))) {
    /* (host destinations: four scanners at most -- the link bounds the pipeline there, the scan only has to stay off its path) */
    total = par_drive(b, mp3, n, b->pcm_pinned == 2 || b->scan_forced || b->scan_threads < 4 ? b->scan_threads : 4);
    if (total == PAR_GIVEN_UP) {
      /* not a stream the split scan can take after all (something irregular further in): what has gone to the engine
       * is let through, then the stream is decoded again from its first frame by the one-thread scanner */
      (void)bulk_drain(b);
      bulk_begin(b);
      b->pcm = pcm; b->pcm_cap = pcm_cap;
    }
  }
  if (total == PAR_NOT_TAKEN || total == PAR_GIVEN_UP) total = bulk_drive(b, mp3, n);
  const double t_driven = now_s();
  b->t_drive += t_driven - t_in;
  int ok = !b->failed;
  if (b->bits_mode) ok = bits_close_window(b) == PDMP3_OK && ok;
  else {
    ok = ok && bulk_rotate(b) == PDMP3_OK;               /* the partly filled last window */
    ok = ok && bulk_finish_b(b) == PDMP3_OK;
  }
  if (drain || !b->bits_mode || !ok) ok = bulk_drain(b) == PDMP3_OK && ok;
  else if (b->pool_mode) ok = sub_drain_copied(b) == PDMP3_OK && ok;      /* the submitter has taken the main data out of `mp3` */
  b->t_tail += now_s() - t_driven;
  if (rate) *rate = (long)kLsfSampleRates[sfreq9(&b->id->hdr)];
  if (channels) *channels = b->id->hdr.mode == 3 ? 1 : 2;
  if (getenv("PDMP3_BULK_TRACE")) {
    fprintf(stderr, "bulk trace: scan loop %.2f ms (of it waiting for the submitter %.2f ms), tail %.2f ms (cumulative)\n",
            b->t_drive * 1e3, b->t_subwait * 1e3, b->t_tail * 1e3);
    fprintf(stderr, "bulk trace: submit %.2f ms, gpu wait %.2f ms, pool wait %.2f ms; submitter: copies %.2f ms, engine calls %.2f ms (cumulative)\n",
            b->t_submit * 1e3, b->t_gpuwait * 1e3, b->t_poolwait * 1e3, b->t_sub_gather * 1e3, b->t_sub_call * 1e3);
  }
  if (total == PDMP3_BULK_REPLAY) return PDMP3_BULK_REPLAY;
  return ok ? total : -1;
}

/* PCM buffers in pinned host memory: the GPU downloads each window straight into them, the pool has nothing to copy */
void* pdmp3_amd_pcm_alloc(size_t bytes) {
  void* p = NULL;
  return pdmp3_hip_host_alloc(bytes, &p) == PDMP3_HIP_OK ? p : NULL;
}
void pdmp3_amd_pcm_free(void* p) { pdmp3_hip_host_free(p); }

/* Decode a whole stream.  Returns the PCM byte count pdmp3() writes for it (the first min(that, pcm_cap)
 * bytes are in `pcm`), or -1 on an engine failure.  rate / channels: format of the last header seen. */
long long pdmp3_amd_bulk_decode(struct bulk* b, const unsigned char* mp3, size_t n, unsigned char* pcm, size_t pcm_cap,
                                long* rate, int* channels) {
  return bulk_decode_impl(b, mp3, n, pcm, pcm_cap, rate, channels, 1);
}

/* The same without waiting for the tail: returns as soon as the stream is scanned and its windows are queued
 * (`mp3` may be released then); `pcm` is complete after pdmp3_amd_bulk_wait().  The next stream's scan overlaps
 * the previous one's GPU work and copy-out -- for corpora of many files.  (Host-Huffman decoders wait anyway.) */
long long pdmp3_amd_bulk_decode_async(struct bulk* b, const unsigned char* mp3, size_t n, unsigned char* pcm, size_t pcm_cap,
                                      long* rate, int* channels) {
  return bulk_decode_impl(b, mp3, n, pcm, pcm_cap, rate, channels, 0);
}

int pdmp3_amd_bulk_wait(struct bulk* b) {
  if (!b || !b->hs) return -1;
  return bulk_drain(b) == PDMP3_OK ? 0 : -1;
}

/* Host stages A-C only: the records the engine would be given, into caller memory (cap_frames frames).
 * Returns the frame count, or -1 when they do not fit. */
long long pdmp3_amd_bulk_parse(struct bulk* b, const unsigned char* mp3, size_t n, int16_t* spectra, pdmp3_gc_side* side,
                               size_t cap_frames, long long* pcm_bytes) {
  if (!b || b->hs || b->bits_mode || (!mp3 && n)) return -1;
  bulk_begin(b);
  b->rec_spectra = spectra; b->rec_side = side; b->rec_cap = cap_frames;
  const long long total = bulk_drive(b, mp3, n);
  int ok = !b->failed && bulk_rotate(b) == PDMP3_OK;
  ok = ok && bulk_finish_b(b) == PDMP3_OK;
  bulk_wait_b(b);
  b->in_b = NULL;
  if (pcm_bytes) *pcm_bytes = total;
  if (total == PDMP3_BULK_REPLAY) return PDMP3_BULK_REPLAY;
  return ok ? b->frames : -1;
}

/* Stage A only, bits mode, into caller memory: what pdmp3_hip_stream_submit_bits would be given (host tests) */
long long pdmp3_amd_bulk_parse_bits(struct bulk* b, const unsigned char* mp3, size_t n, pdmp3_frame_bits* bits, uint8_t* res,
                                    size_t cap_frames, long long* pcm_bytes) {
  if (!b || b->hs || !b->bits_mode || (!mp3 && n)) return -1;
  bulk_begin(b);
  b->rec_bits = bits; b->rec_res = res; b->rec_cap = cap_frames;
  const long long total = bulk_drive(b, mp3, n);
  if (pcm_bytes) *pcm_bytes = total;
  if (total == PDMP3_BULK_REPLAY) return PDMP3_BULK_REPLAY;
  return b->failed ? -1 : b->frames;
}

/* the same in the compact form (include/pdmp3_hip.h: pdmp3_row_desc): side info, row descriptors and the pool of ONE
 * window that holds the whole stream (host tests of the pool rule against the snapshot rows) */
long long pdmp3_amd_bulk_parse_pool(struct bulk* b, const unsigned char* mp3, size_t n, pdmp3_frame_bits* bits,
                                    pdmp3_row_desc* desc, uint8_t* pool, size_t pool_cap, size_t cap_frames, size_t* pool_bytes) {
  if (!b || b->hs || !b->bits_mode || (!mp3 && n)) return -1;
  b->pool_mode = 1;
  bulk_begin(b);
  b->rec_bits = bits; b->rec_res = pool; b->rec_desc = desc; b->rec_pool_cap = pool_cap; b->rec_cap = cap_frames;
  const long long total = bulk_drive(b, mp3, n);
  if (b->bits_open) { pool_materialize(b); pool_gather(pool, b->gath_cur, b->gath_n); }
  if (pool_bytes) *pool_bytes = b->pool_tail;
  b->pool_mode = 0; b->id->pool_sink = NULL;
  if (total == PDMP3_BULK_REPLAY) return PDMP3_BULK_REPLAY;
  return b->failed ? -1 : b->frames;
}

/* ------------------------------------------------------------------------ */
/* A corpus of files over the GPUs of a node ("C4: whole files per */
/* GPU, largest first"): the C form of pdmp3_amd/sharding.py assign_files +   */
/* one whole-stream decoder per device.  include/pdmp3_bulk.h                  */
/* ------------------------------------------------------------------------ */
/* largest-first greedy; ties go to the lower rank (== sharding.assign_files): rank_of[i] = the device slot of file i */
void pdmp3_amd_corpus_assign(const size_t* sizes, int n_files, int world, int* rank_of) {
  if (!sizes || !rank_of || n_files <= 0 || world <= 0) return;
  int* order = (int*)malloc((size_t)n_files * sizeof *order);
  unsigned long long* load = (unsigned long long*)calloc((size_t)world, sizeof *load);
  if (!order || !load) { free(order); free(load); for (int i = 0; i < n_files; i++) rank_of[i] = i % world; return; }
  for (int i = 0; i < n_files; i++) order[i] = i;
  for (int i = 1; i < n_files; i++) {             /* stable insertion sort by size, descending (corpora are thousands of files at most) */
    const int k = order[i];
    int j = i;
    while (j > 0 && sizes[order[j - 1]] < sizes[k]) { order[j] = order[j - 1]; j--; }
    order[j] = k;
  }
  for (int q = 0; q < n_files; q++) {
    int r = 0;
    for (int k = 1; k < world; k++) if (load[k] < load[r]) r = k;
    rank_of[order[q]] = r;
    load[r] += sizes[order[q]];
  }
  free(order); free(load);
}

struct corpus_job {
  int device, slot, world, n_files, host_huffman, threads, window;
  unsigned iso;
  const int* rank_of;
  const unsigned char* const* mp3s; const size_t* sizes;
  unsigned char* const* pcm; const size_t* pcm_caps; long long* pcm_bytes;
  int rc;
};
static void* corpus_worker(void* arg) {
  struct corpus_job* j = (struct corpus_job*)arg;
  struct bulk* b = pdmp3_amd_bulk_new_on(j->threads, j->window, j->host_huffman, j->device);
  if (!b) { j->rc = -1; return NULL; }
  (void)pdmp3_amd_bulk_set_quirks(b, j->iso);
  for (int i = 0; i < j->n_files && j->rc == 0; i++) {
    if (j->rank_of[i] != j->slot) continue;
    /* (asynchronous: the next file's scan runs under this one's GPU work and copy-out) */
    const long long got = pdmp3_amd_bulk_decode_async(b, j->mp3s[i], j->sizes[i], j->pcm[i], j->pcm_caps[i], NULL, NULL);
    j->pcm_bytes[i] = got;
    if (got < 0 && got != PDMP3_BULK_REPLAY) j->rc = -1;
  }
  if (pdmp3_amd_bulk_wait(b) != 0) j->rc = -1;
  pdmp3_amd_bulk_delete(b);
  return NULL;
}
/* n_files whole streams over n_devices HIP devices (a device may be listed more than once: that many decoders share it), one
 * host thread and one whole-stream decoder per entry, files dealt largest first.  pcm[i] (capacity pcm_caps[i]) receives file
 * i's PCM, pcm_bytes[i] what pdmp3() would write for it (or PDMP3_BULK_REPLAY).  0, or -1 on an engine failure. */
int pdmp3_amd_corpus_decode(const int* devices, int n_devices, const unsigned char* const* mp3s, const size_t* sizes, int n_files,
                            unsigned char* const* pcm, const size_t* pcm_caps, long long* pcm_bytes, unsigned iso_mask,
                            int threads_per_decoder, int window_frames, int host_huffman) {
  if (!devices || n_devices < 1 || n_devices > 64 || n_files < 0 || (n_files && (!mp3s || !sizes || !pcm || !pcm_caps || !pcm_bytes))) return -1;
  if (!n_files) return 0;
  int* rank_of = (int*)malloc((size_t)n_files * sizeof *rank_of);
  if (!rank_of) return -1;
  pdmp3_amd_corpus_assign(sizes, n_files, n_devices, rank_of);
  struct corpus_job jobs[64];
  pthread_t th[64];
  int started = 0, rc = 0;
  for (int k = 0; k < n_devices; k++) {
    jobs[k] = (struct corpus_job){devices[k], k, n_devices, n_files, host_huffman, threads_per_decoder, window_frames, iso_mask,
                                  rank_of, mp3s, sizes, pcm, pcm_caps, pcm_bytes, 0};
    if (pthread_create(&th[k], NULL, corpus_worker, &jobs[k]) != 0) { rc = -1; break; }
    started++;
  }
  for (int k = 0; k < started; k++) { pthread_join(th[k], NULL); if (jobs[k].rc) rc = -1; }
  free(rank_of);
  return rc;
}

/* ------------------------------------------------------------------------ */
/* CLI driver (P:2540-2589) with the raw sink (P:2236-2257)                   */
/* ------------------------------------------------------------------------ */
/* RIFF/WAVE header for interleaved PCM: 16-bit integer (format 1) or 32-bit float (format 3); data_bytes = 0xffffffff
 * when the length is not known yet (a pipe) */
static void wav_header(unsigned char h[44], long rate, int channels, int float32, uint32_t data_bytes) {
  const uint32_t bps = float32 ? 4 : 2, align = bps * (uint32_t)channels;
  const uint32_t riff = data_bytes == 0xffffffffu ? 0xffffffffu : data_bytes + 36;
#define LE32(p, v) ((p)[0] = (unsigned char)(v), (p)[1] = (unsigned char)((v) >> 8), (p)[2] = (unsigned char)((v) >> 16), (p)[3] = (unsigned char)((v) >> 24))
#define LE16(p, v) ((p)[0] = (unsigned char)(v), (p)[1] = (unsigned char)((v) >> 8))
  memcpy(h, "RIFF", 4); LE32(h + 4, riff); memcpy(h + 8, "WAVEfmt ", 8); LE32(h + 16, 16u);
  LE16(h + 20, float32 ? 3u : 1u); LE16(h + 22, (uint32_t)channels); LE32(h + 24, (uint32_t)rate);
  LE32(h + 28, (uint32_t)rate * align); LE16(h + 32, align); LE16(h + 34, bps * 8);
  memcpy(h + 36, "data", 4); LE32(h + 40, data_bytes);
#undef LE32
#undef LE16
}

/* include/pdmp3_bulk.h: a whole PCM buffer as a .wav file */
int pdmp3_amd_write_wav(const char* path, const void* pcm, size_t bytes, long rate, int channels, int float32) {
  if (!path || (!pcm && bytes) || rate <= 0 || channels < 1 || channels > 2 || bytes > 0xfffffff0u) return PDMP3_ERR;
  FILE* f = fopen(path, "wb");
  if (!f) return PDMP3_ERR;
  unsigned char h[44];
  wav_header(h, rate, channels, float32, (uint32_t)bytes);
  int ok = fwrite(h, 1, 44, f) == 44 && (bytes == 0 || fwrite(pcm, 1, bytes, f) == bytes);
  ok = (fclose(f) == 0) && ok;
  return ok ? PDMP3_OK : PDMP3_ERR;
}

/* The driver's sink.  Default: the reference's OUTPUT_RAW writer (P:2236-2257) -- "<first name>.raw", opened once for
 * the FIRST file name only, O_CREAT without O_TRUNC, "-" = stdout.  PDMP3_CLI_WAV=1: the same samples as
 * "<first name>.wav" (truncated, 44-byte header with the first stream's rate and channel count, sizes filled in when
 * the driver is done; to stdout with unknown-length sizes). */
static int g_out_fd = -2, g_out_wav = 0, g_out_hdr = 0;
static uint64_t g_out_bytes = 0;
static long g_out_rate = 44100;
static int g_out_ch = 2;

static void write_all(int fd, const unsigned char* data, size_t nbytes) {
  size_t off = 0;
  while (off < nbytes) {
    ssize_t w = write(fd, data + off, nbytes - off);
    if (w <= 0) { fputs("Unable to write raw data\n", stderr); exit(-1); }
    off += (size_t)w;
  }
}

static void write_raw(const char* filename, const unsigned char* data, size_t nbytes, long rate, int channels) {
  if (g_out_fd == -2) {
    const char* w = getenv("PDMP3_CLI_WAV");
    g_out_wav = w && *w && *w != '0';
    if (strcmp(filename, "-")) {
      char name[1024];
      snprintf(name, sizeof name, g_out_wav ? "%s.wav" : "%s.raw", filename);
      g_out_fd = open(name, O_WRONLY | O_CREAT | (g_out_wav ? O_TRUNC : 0), 0666);  /* raw: no O_TRUNC, like the reference */
      if (g_out_fd == -1) { perror(name); exit(-1); }
    } else g_out_fd = 1;
  }
  /* the sink is opened by the FIRST call, data or not -- the reference calls its writer after every pdmp3_read, the
   * first NEED_MORE with nothing decoded included (P:2565-2566, P:2239-2251), so the output is named after the first
   * file even if that one yields no PCM; only the WAV header waits for the first data (rate and channels) */
  if (g_out_wav && !g_out_hdr && nbytes) {
    unsigned char h[44];
    g_out_rate = rate > 0 ? rate : 44100; g_out_ch = channels == 1 ? 1 : 2;
    wav_header(h, g_out_rate, g_out_ch, 0, 0xffffffffu);
    write_all(g_out_fd, h, 44);
    g_out_hdr = 1;
  }
  write_all(g_out_fd, data, nbytes);
  g_out_bytes += nbytes;
}

static void finish_output(void) {
  if (g_out_fd >= 0 && g_out_wav && !g_out_hdr) {            /* no data at all: an empty WAV file */
    unsigned char h[44];
    wav_header(h, g_out_rate, g_out_ch, 0, 0);
    write_all(g_out_fd, h, 44);
    g_out_hdr = 1;
    return;
  }
  if (g_out_fd >= 0 && g_out_wav && g_out_fd != 1 && g_out_bytes <= 0xfffffff0u && lseek(g_out_fd, 0, SEEK_SET) == 0) {
    unsigned char h[44];
    wav_header(h, g_out_rate, g_out_ch, 0, (uint32_t)g_out_bytes);
    write_all(g_out_fd, h, 44);
  }
}

/* one file through the reference's own loop (P:2566-2583): stdin, and files the whole-stream path declines */
static void cli_stream_file(pdmp3_handle* id, const char* filename, FILE* fp) {
  unsigned char out[INBUF_SIZE];
  pdmp3_open_feed(id);
  size_t done;
  int res;
  while ((res = pdmp3_read(id, out, INBUF_SIZE, &done)) != PDMP3_ERR) {
    write_raw(filename, out, done, done ? (long)kLsfSampleRates[sfreq9(&id->l_hdr)] : 0, id->l_hdr.mode == 3 ? 1 : 2);
    if (res == PDMP3_NEED_MORE) {
      unsigned char in[4096];
      const size_t n = fread(in, 1, sizeof in, fp);
      if (!n) break;
      (void)pdmp3_feed(id, in, n);
    }
  }
}

/* The streaming API driven from a memory buffer, in C (include/pdmp3_bulk.h): pdmp3_new, pdmp3_open_feed, then
 * pdmp3_read(read_bytes) until PDMP3_ERR, feeding feed_bytes on PDMP3_NEED_MORE -- the reference driver's loop
 * (P:2564-2584) with its two sizes as parameters (4096 / 16384 there).  eager != 0: the caller keeps the ring as full
 * as feed_bytes-sized feeds allow instead of waiting for PDMP3_NEED_MORE. */
long long pdmp3_amd_stream_loop(const unsigned char* mp3, size_t n, unsigned char* out, size_t cap,
                                size_t feed_bytes, size_t read_bytes, int eager) {
  if (!mp3 || !feed_bytes || !read_bytes) return -1;
  pdmp3_handle* id = pdmp3_new(NULL, NULL);
  if (!id) return -1;
  unsigned char* buf = (unsigned char*)malloc(read_bytes);
  if (!buf) { pdmp3_delete(id); return -1; }
  pdmp3_open_feed(id);
  size_t fed = 0, done, total = 0;
  int res;
  for (;;) {
    if (eager)
      while (fed < n) {
        const size_t take = n - fed < feed_bytes ? n - fed : feed_bytes;
        /* never to the last byte: a ring filled exactly looks EMPTY to the reference (iend == istart, P:1062-1068)
         * and the next feeds would overwrite it -- a caller of the real API has to keep count for this itself */
        if (take >= ring_free_logical(id) || pdmp3_feed(id, mp3 + fed, take) != PDMP3_OK) break;
        fed += take;
      }
    res = pdmp3_read(id, buf, read_bytes, &done);
    if (res == PDMP3_ERR) break;
    if (out && total < cap) memcpy(out + total, buf, done < cap - total ? done : cap - total);
    total += done;
    if (res == PDMP3_NEED_MORE) {
      if (fed >= n) break;
      if (!eager) {
        const size_t take = n - fed < feed_bytes ? n - fed : feed_bytes;
        (void)pdmp3_feed(id, mp3 + fed, take);
        fed += take;
      }
    }
  }
  free(buf);
  pdmp3_delete(id);
  return (long long)total;
}

/* Same contract as the reference's driver: every named file is decoded to interleaved int16 and appended to
 * "<first name>.raw".  Regular files take the whole-stream path (include/pdmp3_bulk.h), whose output is by
 * definition what the loop above produces -- parse state carried from file to file like the reference's single
 * handle does; PDMP3_CLI_STREAMING=1 forces the loop. */
void pdmp3(char* const* mp3s) {
  if (*mp3s && !strncmp("/dev/dsp", *mp3s, 8)) mp3s++;      /* OSS device argument accepted, playback not supported */
  pdmp3_handle* id = pdmp3_new(NULL, NULL);
  if (!id) { fputs("Cannot open stream API (no transform engine)\n", stderr); exit(0); }
  const char* force = getenv("PDMP3_CLI_STREAMING");
  const int streaming_only = force && *force && *force != '0';
  /* $PDMP3_CLI_ISO = mask of PDMP3_ISO_* (include/pdmp3.h): the standard's behaviour instead of the reference's; default 0 */
  const char* iso_env = getenv("PDMP3_CLI_ISO");
  const unsigned iso = iso_env ? (unsigned)strtoul(iso_env, NULL, 0) & (PDMP3_ISO_ALL | PDMP3_ISO_LSF) : 0u;
  (void)pdmp3_amd_set_quirks(id, iso);
  struct bulk* b = NULL;
  int bulk_used = 0, loop_used = 0;
  for (; *mp3s; mp3s++) {
    const char* filename = *mp3s;
    FILE* fp = strcmp(filename, "-") ? fopen(filename, "r") : stdin;
    if (!fp) { fputs("Cannot open file\n", stderr); exit(0); }
    unsigned char* data = NULL;
    long size = -1;
    /* the two paths keep their parse state in different handles: once one of them has decoded a file, later
     * files stay on it */
    if (fp != stdin && !streaming_only && !loop_used && fseek(fp, 0, SEEK_END) == 0 && (size = ftell(fp)) >= 0 &&
        fseek(fp, 0, SEEK_SET) == 0) {
      data = (unsigned char*)malloc((size_t)size + 1);
      if (data && fread(data, 1, (size_t)size, fp) != (size_t)size) { free(data); data = NULL; }
      if (!data) (void)fseek(fp, 0, SEEK_SET);
    }
    long long total = -1;
    if (data) {
      total = pdmp3_amd_scan_buffer_iso(data, (size_t)size, iso, NULL);
      if (total == PDMP3_BULK_REPLAY && bulk_used) {            /* the reference would not terminate on this file */
        fprintf(stderr, "pdmp3: %s: the reference decoder replays its input ring on this stream; skipped\n", filename);
        total = 0;
      }
    }
    if (data && total >= 0) {
      if (!b) { b = pdmp3_amd_bulk_new(0, 0); if (b) (void)pdmp3_amd_bulk_set_quirks(b, iso); }
      unsigned char* pcm = (unsigned char*)malloc((size_t)total + 1);
      if (!b || !pcm) { fputs("Cannot open stream API (no transform engine)\n", stderr); exit(0); }
      b->carry = bulk_used;                       /* first file: fresh state, like the reference's new handle */
      long rate = 0; int ch = 0;
      const long long got = pdmp3_amd_bulk_decode(b, data, (size_t)size, pcm, (size_t)total, &rate, &ch);
      if (got != total) { fputs("pdmp3: engine failure\n", stderr); exit(-1); }
      write_raw(filename, pcm, (size_t)total, rate, ch);
      free(pcm);
      bulk_used = 1;
    } else {
      if (data) (void)fseek(fp, 0, SEEK_SET);
      cli_stream_file(id, filename, fp);
      loop_used = 1;
    }
    free(data);
    if (fp != stdin) fclose(fp);
  }
  finish_output();
  if (b) pdmp3_amd_bulk_delete(b);
  pdmp3_delete(id);
}

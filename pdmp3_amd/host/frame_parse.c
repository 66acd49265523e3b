/* frame_parse.c -- libpdmp3.so: everything that parses ONE frame (P:1252-1474, 2051-2115): header sync, side info (MPEG-1 and
 * LSF), bit reservoir, main data = scalefactors + table-driven Huffman as a pure function (decode_main), the merge into
 * the state that survives frames (apply_main), and the record builder at the engine boundary (emit_records).
 * See host_internal.h for the map of the library. */
#include "host_internal.h"

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
  /* PDMP3_ISO_LSF (not the reference): eleven sync bits, so that MPEG-2.5's 0xFFE + ID 0 is a header too.  In bits mode (the
   * device's Huffman stage reads MPEG-1 side info only: include/pdmp3_bulk.h) an LSF frame ends the scan -- lsf_seen, below --
   * and the stream goes to the decoder's host-Huffman twin from its first byte: what is a frame and what is junk must not
   * depend on which stage decodes the Huffman data */
  const int lsf_ok = (id->iso & PDMP3_ISO_LSF) != 0;
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
  if (H->ver && id->bits_scan) {                  /* (only Layer III frames count: search_header goes on otherwise, as with any junk) */
    if (H->layer == 3) id->lsf_seen = 1;
    H->layer = 0;
    return PDMP3_ERR;
  }
  if (!id->new_header) id->new_header = 1;
  return PDMP3_OK;
}

/* P:1322-1340: retry from the next byte after the mark; give up after 1152 tries */
int search_header(pdmp3_handle* id) {
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
void read_side_info_bits(pdmp3_handle* id) {
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
int fill_reservoir(pdmp3_handle* id, unsigned size, unsigned begin) {
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
void decode_main(const uint8_t* reservoir, const frame_header* H, const side_info* S, main_out* out) {
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
void apply_main(pdmp3_handle* id, const frame_header* H, const main_out* out) {
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
int read_frame_staged(pdmp3_handle* id) {
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
int read_frame(pdmp3_handle* id, int16_t* spectra) {
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
void emit_records(pdmp3_handle* id, const frame_header* H, const side_info* S, int reset,
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
size_t drain_frame(pdmp3_handle* id, unsigned char* out, size_t buflen) {
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


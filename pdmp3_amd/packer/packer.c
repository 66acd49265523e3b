/*
 * packer.c -- MPEG-1 Layer III bitstream generator (include/pdmp3_packer.h; SURVEY 8f #3: the corpus generator of
 * the stream-level workloads, `python -m pdmp3_amd.packer`).
 *
 * There is no MP3 encoder in this image and no network, so the corpora for the
 * stream-level configs are made here: syntactically valid
 * frames with random content -- headers (CBR with ISO padding or VBR), CRC
 * words, side info, scfsi, long / start / short / stop / mixed blocks, every
 * Huffman table incl. linbits and both count1 tables, and a bit reservoir
 * (main_data_begin up to 511).  What a frame "should" decode to is defined by
 * the reference decoder (oracle), not by this tool, so the only requirement is
 * validity and variety.  Bitstream facts follow SURVEY appendix A.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/pdmp3_packer.h"
#include "../csrc/tables_data.h"
#include "../csrc/lsf_tables.h"

/* ---------- rng ---------- */
static uint64_t rng_state;
static uint64_t rnd(void) {
  uint64_t x = (rng_state += 0x9E3779B97F4A7C15ull);
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
static unsigned rndn(unsigned n) { return n ? (unsigned)(rnd() % n) : 0; }

/* ---------- bit writer ---------- */
typedef struct { uint8_t* p; size_t cap; size_t bits; } bw;
static void bw_put(bw* w, uint32_t v, unsigned n) {
  for (int i = (int)n - 1; i >= 0; i--) {
    size_t byte = w->bits >> 3;
    if (byte >= w->cap) { w->bits++; continue; }
    if ((w->bits & 7) == 0) w->p[byte] = 0;
    if ((v >> i) & 1) w->p[byte] |= (uint8_t)(0x80 >> (w->bits & 7));
    w->bits++;
  }
}

/* ---------- encode tables: val -> code ---------- */
static struct { uint32_t code; uint8_t len; } g_enc[PDMP3_NUM_HUFF_BOOKS][256];
static int g_enc_ready = 0;
static void build_enc(void) {
  if (g_enc_ready) return;
  memset(g_enc, 0, sizeof g_enc);
  for (int b = 0; b < PDMP3_NUM_HUFF_BOOKS; b++)
    for (int i = 0; i < kHuffBookSize[b]; i++) {
      const pdmp3_hcode* c = &kHuffBooks[b][i];
      if (c->err) continue;
      g_enc[b][c->val].code = c->code;
      g_enc[b][c->val].len = c->len;
    }
  g_enc_ready = 1;
}

/* max x/y of each pair table (ISO 11172-3 table B.7 dimensions) */
static const int kTabMax[32] = {0, 1, 2, 2, 0, 3, 3, 5, 5, 5, 7, 7, 7, 15, 0, 15,
                                15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15};

typedef struct {
  unsigned part2_3_length, big_values, global_gain, scalefac_compress, win_switch, block_type, mixed;
  unsigned table_select[3], subblock_gain[3], region0, region1, preflag, scalefac_scale, count1table;
} gc_side;

static const uint16_t* sfb_l_of(int f) { return f == 0 ? kSfbLong0 : f == 1 ? kSfbLong1 : kSfbLong2; }
/* version 0 = MPEG-1 (the reference's streams), 1 = MPEG-2 LSF, 2 = MPEG-2.5: lsf_tables.h */
static const uint16_t* sfb_l_v(const pk_cfg* c) { return c->version ? kLsfSfbLong[(c->version - 1) * 3 + c->sfreq] : sfb_l_of(c->sfreq); }
static const uint16_t* sfb_s_v(const pk_cfg* c) {
  return c->version ? kLsfSfbShort[(c->version - 1) * 3 + c->sfreq] : c->sfreq == 0 ? kSfbShort0 : c->sfreq == 1 ? kSfbShort1 : kSfbShort2;
}

/* writes one granule-channel's main data (scalefactors + Huffman) into `w`,
 * using at most `budget` bits; fills the side-info fields it decides */
static unsigned draw_block_type(const pk_cfg* c) {
  unsigned pct = rndn(100), bt = 0, acc = 0;
  for (int k = 0; k < 4; k++) { acc += (unsigned)c->block_pct[k]; if (pct < acc) { bt = (unsigned)k; break; } }
  return bt;
}

/* pre_bt < 0: the block type is drawn here (the streams of rounds 1-5, bit for bit); otherwise it was drawn by the
 * caller (iso_strict: scfsi depends on both granules' types; pre_mixed >= 0 likewise).  line_cap: no line at or above it is coded (the right
 * channel of an intensity-stereo granule: is_cut_pct). */
static void gen_gc(const pk_cfg* c, bw* w, gc_side* s, unsigned budget, int gr, const unsigned scfsi[4], unsigned* used,
                   int pre_bt, int pre_mixed, unsigned line_cap, int is_right) {
  const size_t start = w->bits;
  memset(s, 0, sizeof *s);
  const unsigned bt = pre_bt < 0 ? draw_block_type(c) : (unsigned)pre_bt;
  s->win_switch = bt != 0;
  s->block_type = bt;
  s->mixed = pre_mixed >= 0 ? (unsigned)(bt == 2 && pre_mixed) : (unsigned)(bt == 2 && (int)rndn(100) < c->mixed_pct);
  s->global_gain = (unsigned)c->gain_lo + rndn((unsigned)(c->gain_hi - c->gain_lo + 1));
  s->scalefac_compress = rndn(c->narrow_scales ? 14 : 16);     /* narrow: scalefactors <= 7 (slen <= 3) */
  s->preflag = rndn(2);
  s->scalefac_scale = rndn(2);
  s->count1table = ((int)rndn(100) < c->table33_pct) ? 1 : 0;
  for (int k = 0; k < 3; k++) s->subblock_gain[k] = rndn(c->narrow_scales ? 3 : 8);
  if (budget < 160) {            /* nothing fits: an empty granule (part2_3_length = 0) */
    s->part2_3_length = 0;
    s->big_values = 0;
    *used = 0;
    return;
  }
  /* scalefactors */
  if (c->version) {
    /* LSF (13818-3 2.4.3.2): 9-bit scalefac_compress -> four slen and the partition sizes of lsf_tables.h; no scfsi, no
     * preflag bit (implied by scalefac_compress >= 500).  is_right: channel 1 of an intensity-stereo frame has its own
     * classes, and its scalefactors are intensity positions: the largest value of a partition means "not intensity
     * coded" (which not every decoder honours: iso_strict streams do not use it) */
    uint8_t slen[4];
    int pf, cls, again;
    const int shape = (s->win_switch && s->block_type == 2) ? (s->mixed ? 2 : 1) : 0;
    do {
      s->scalefac_compress = rndn(512);
      cls = lsf_slen_of(s->scalefac_compress, is_right, slen, &pf);
      again = c->narrow_scales && (slen[0] > 3 || slen[1] > 3 || slen[2] > 3 || slen[3] > 3);
      /* iso_strict: no intensity position that means "not intensity coded" -- the largest value of its slen, which for
       * slen 0 is the only value there is (FFmpeg, the fixtures' decoder, reads those as positions) */
      if (is_right && c->iso_strict)
        for (int k = 0; k < 4; k++) if (kLsfNsfb[cls][shape][k] && !slen[k]) again = 1;
    } while (again);
    s->preflag = (unsigned)pf;
    for (int k = 0; k < 4; k++)
      for (int i = 0; i < kLsfNsfb[cls][shape][k]; i++) {
        unsigned range = 1u << slen[k];
        if (is_right && c->iso_strict && range > 1) range -= 1;
        bw_put(w, rndn(range), slen[k]);
      }
  } else {
  const unsigned slen1 = kSlen[s->scalefac_compress * 2], slen2 = kSlen[s->scalefac_compress * 2 + 1];
  if (s->win_switch && s->block_type == 2) {
    unsigned first = 0;
    if (s->mixed) { for (int k = 0; k < 8; k++) bw_put(w, rndn(1u << slen1), slen1); first = 3; }
    for (unsigned sfb = first; sfb < 12; sfb++)
      for (int win = 0; win < 3; win++) { unsigned nb = sfb < 6 ? slen1 : slen2; bw_put(w, rndn(1u << nb), nb); }
  } else {
    static const int lo[5] = {0, 6, 11, 16, 21};
    for (int b = 0; b < 4; b++) {
      if (gr == 1 && scfsi[b]) continue;
      unsigned nb = b < 2 ? slen1 : slen2;
      for (int k = lo[b]; k < lo[b + 1]; k++) bw_put(w, rndn(1u << nb), nb);
    }
  }
  }
  /* tables / regions */
  if (s->win_switch) {
    s->region0 = (s->block_type == 2 && !s->mixed) ? 8 : 7;
    s->region1 = 20 - s->region0;
    for (int k = 0; k < 2; k++) s->table_select[k] = rndn(32);
  } else {
    s->region0 = rndn(14);                         /* region0 + region1 + 2 <= 22 */
    s->region1 = rndn(8);
    if (!c->iso_strict && rndn(50) == 0) { s->region0 = 15; s->region1 = 6 + rndn(2); }   /* indices 23/24: reference H7 */
    for (int k = 0; k < 3; k++) s->table_select[k] = rndn(32);
  }
  for (int k = 0; k < 3; k++) if (s->table_select[k] == 4 || s->table_select[k] == 14) s->table_select[k] = rndn(2) ? 0 : 15;
  unsigned r1, r2;
  if (s->win_switch && s->block_type == 2) { r1 = (c->version == 2 && c->sfreq == 2) ? 72 : 36; r2 = 576; }   /* (8 kHz: three short bands are 72 lines) */
  else {
    const uint16_t* l = sfb_l_v(c);
    const uint16_t* sh = sfb_s_v(c);
    const unsigned i1 = s->region0 + 1, i2 = s->region0 + s->region1 + 2;
    r1 = i1 < 23 ? l[i1] : sh[i1 - 23];
    r2 = i2 < 23 ? l[i2] : sh[i2 - 23];
  }
  /* big values */
  const unsigned want_big = rndn(289);            /* <= 288 (reference H8) */
  unsigned nbig = 0, pos = 0;
  const unsigned soft = (unsigned)((uint64_t)budget * (60 + rndn(41)) / 100);   /* leave room for count1 */
  while (nbig < want_big && pos + 2 <= line_cap) {
    const unsigned tn = pos < r1 ? s->table_select[0] : pos < r2 ? s->table_select[1] : s->table_select[2];
    const int book = kHuffBookOfTable[tn];
    unsigned need = 0;
    int x = 0, y = 0;
    if (book >= 0) {
      const int mx = kTabMax[tn];
      const unsigned lb = kHuffLinbits[tn];
      /* decaying magnitudes; occasionally the linbits range */
      const int cap = 1 + (int)((uint64_t)mx * (576 - pos) / 576);
      x = (int)rndn((unsigned)(cap < mx ? cap : mx) + 1);
      y = (int)rndn((unsigned)(cap < mx ? cap : mx) + 1);
      unsigned lx = 0, ly = 0;
      if (lb && mx == 15 && (int)rndn(1000) < c->big_pct) { x = 15; lx = rndn(1u << lb); }
      if (lb && mx == 15 && (int)rndn(1000) < c->big_pct) { y = 15; ly = rndn(1u << lb); }
      const unsigned val = (unsigned)(x << 4 | y);
      need = g_enc[book][val].len + (x ? 1 : 0) + (y ? 1 : 0) + ((lb && x == 15) ? lb : 0) + ((lb && y == 15) ? lb : 0);
      if ((w->bits - start) + need > soft) break;
      bw_put(w, g_enc[book][val].code, g_enc[book][val].len);
      if (lb && x == 15) bw_put(w, lx, lb);
      if (x) bw_put(w, rndn(2), 1);
      if (lb && y == 15) bw_put(w, ly, lb);
      if (y) bw_put(w, rndn(2), 1);
    }
    nbig++; pos += 2;
  }
  s->big_values = nbig;
  /* count1 quads with the ISO code books (table 32 = book 15, table 33 = ISO book) */
  const int qbook = s->count1table ? PDMP3_HUFF_BOOK_ISO33 : kHuffBookOfTable[32];
  const unsigned want_q = rndn((576 - pos) / 4 + 1);
  for (unsigned q = 0; q < want_q && pos + 4 <= line_cap; q++) {
    const unsigned val = rndn(16);
    unsigned nz = 0;
    for (int k = 0; k < 4; k++) nz += (val >> k) & 1;
    const unsigned need = g_enc[qbook][val].len + nz;
    if ((w->bits - start) + need > budget) break;
    bw_put(w, g_enc[qbook][val].code, g_enc[qbook][val].len);
    for (unsigned k = 0; k < nz; k++) bw_put(w, rndn(2), 1);
    pos += 4;
  }
  /* a few stuffing bits now and then (the decoder jumps to part2_3 end, P:2113) */
  /* (not under iso_strict: what follows the last code word inside part2_3_length is read as more quads by every decoder,
   *  and whether a LAST quad that overruns the part is kept differs -- P:2104 drops it, FFmpeg keeps it when it ends at
   *  line 576) */
  if (!c->iso_strict && rndn(4) == 0) { unsigned st = rndn(24); if ((w->bits - start) + st <= budget) bw_put(w, rndn(1u << 12), st > 12 ? 12 : st); }
  s->part2_3_length = (unsigned)(w->bits - start);
  if (s->part2_3_length > 4095) s->part2_3_length = 4095;
  *used = s->part2_3_length;
}

/* Generates n_frames frames; returns the number of bytes written (0 if cap is too small). */
size_t pk_generate(const pk_cfg* c, int n_frames, uint8_t* out, size_t cap) {
  build_enc();
  rng_state = c->seed;
  const int nch = c->mode == 3 ? 1 : 2;
  const unsigned side_bytes = c->version ? (nch == 1 ? 9 : 17) : (nch == 1 ? 17 : 32);
  const int ngr = c->version ? 1 : 2;
  if (c->version < 0 || c->version > 2) return 0;
  /* main-data byte stream of all frames, and where each frame's own area starts */
  const size_t md_cap = (size_t)n_frames * 1500 + 4096;
  uint8_t* md = (uint8_t*)calloc(md_cap, 1);
  uint8_t* sides = (uint8_t*)calloc((size_t)n_frames, 40);
  unsigned* fsize = (unsigned*)calloc((size_t)n_frames, sizeof(unsigned));
  unsigned* hdrs = (unsigned*)calloc((size_t)n_frames, sizeof(unsigned));
  if (!md || !sides || !fsize || !hdrs) return 0;
  bw w = {md, md_cap, 0};
  size_t area_start = 0;            /* bytes: sum of main-data sizes of earlier frames */
  unsigned pad_rest = 0;
  int prev_bt[2] = {0, 0}, run_mixed[2] = {0, 0};   /* iso_strict: the previous granule's block type per channel, the run's mixed flag */
  for (int f = 0; f < n_frames; f++) {
    unsigned bri = (unsigned)c->bitrate_index;
    if (c->vbr) bri = (unsigned)c->vbr_lo + rndn((unsigned)(c->vbr_hi - c->vbr_lo + 1));
    const unsigned br = c->version ? kLsfBitrates[bri] : kBitratesL3[bri];
    const unsigned sf = c->version ? kLsfSampleRates[3 * c->version + c->sfreq] : kSampleRates[c->sfreq];
    const unsigned spf = c->version ? 72u : 144u;       /* 576 samples a frame with LSF: 72 br / sf bytes (13818-3 2.4.3.1) */
    unsigned pad = 0;
    pad_rest += (spf * br) % sf;
    if (pad_rest >= sf) { pad = 1; pad_rest -= sf; }
    const unsigned fbytes = spf * br / sf + pad;
    const unsigned msize = fbytes - 4 - side_bytes - (c->crc ? 2 : 0);
    fsize[f] = msize;
    /* sync + ID: 0xFFF + 1 = MPEG-1; 0xFFF + 0 = MPEG-2 LSF; 0xFFE + 0 = "MPEG-2.5" (eleven sync bits, 00) */
    const unsigned sync_id = c->version == 0 ? (0xFFF00000u | (1u << 19)) : c->version == 1 ? 0xFFF00000u : 0xFFE00000u;
    hdrs[f] = sync_id | (1u << 17) | ((c->crc ? 0u : 1u) << 16) | (bri << 12) |
              ((unsigned)c->sfreq << 10) | (pad << 9) | ((unsigned)c->mode << 6) | ((unsigned)c->mode_ext << 4) | (1u << 2);
    /* where may this frame's data start? */
    size_t pos_bytes = (w.bits + 7) >> 3;
    size_t start = pos_bytes;
    if (!c->reservoir && start < area_start) start = area_start;
    const size_t back = c->version ? 255 : 511;         /* main_data_begin: 9 bits, 8 with LSF */
    if (area_start > back && start < area_start - back) start = area_start - back;
    if (f == 0) start = 0;
    const unsigned begin = (unsigned)(area_start - start);
    w.bits = start * 8;
    const size_t limit_bits = (area_start + msize) * 8;
    unsigned avail = (unsigned)(limit_bits - w.bits);
    avail = (unsigned)((uint64_t)avail * (unsigned)c->fill_pct / 100);
    gc_side gs[2][2];
    unsigned scfsi[2][4];
    for (int ch = 0; ch < nch; ch++) for (int b = 0; b < 4; b++) scfsi[ch][b] = rndn(2);
    int pre_bt[2][2] = {{-1, -1}, {-1, -1}}, pre_mixed[2][2] = {{-1, -1}, {-1, -1}};
    if (c->iso_strict) {
      /* ISO 11172-3 2.4.2.7: scfsi is 0 when a granule of the channel has block_type 2 (decoders disagree about
       * what a copy from a short-block granule means); intensity stereo: both channels share the block shape */
      for (int gr = 0; gr < ngr; gr++) {
        /* ... and the window sequence is the standard's (2.4.3.4.10.3: long -> start -> short ... -> stop -> long):
         * decoders may rely on it (FFmpeg's short-block overlap assumes the six zero samples a start window ends with) */
        for (int ch = 0; ch < nch; ch++) {
          const int in_short = prev_bt[ch] == 1 || prev_bt[ch] == 2;
          const unsigned p = rndn(100);
          if (!in_short) pre_bt[gr][ch] = p < (unsigned)(c->block_pct[1] + c->block_pct[2]) ? 1 : 0;
          else pre_bt[gr][ch] = p < (unsigned)(c->block_pct[0] + c->block_pct[3]) ? 3 : 2;
          /* one mixed_block_flag per run of short blocks: a mixed granule leaves a long window's tail in subbands
           * 0-1, which a pure short granule behind it would have to add under its first two windows */
          if (pre_bt[gr][ch] == 1) run_mixed[ch] = (int)rndn(100) < c->mixed_pct && !(c->version == 2 && c->sfreq == 2);   /* (8 kHz: no mixed blocks, lsf_tables.h) */
          pre_mixed[gr][ch] = run_mixed[ch];
        }
        /* joint stereo: one window shape for both channels (the reference rotates M/S after its reorder, the
         * standard before: they pair different lines when the shapes differ; encoders never let them) */
        if (nch == 2 && c->mode == 1 && c->mode_ext) { pre_bt[gr][1] = pre_bt[gr][0]; pre_mixed[gr][1] = pre_mixed[gr][0]; run_mixed[1] = run_mixed[0]; }
        for (int ch = 0; ch < nch; ch++) prev_bt[ch] = pre_bt[gr][ch];
      }
      for (int ch = 0; ch < nch; ch++)
        if (pre_bt[0][ch] == 2 || pre_bt[1][ch] == 2) for (int b = 0; b < 4; b++) scfsi[ch][b] = 0;
    }
    unsigned left = avail;
    for (int gr = 0; gr < ngr; gr++)
      for (int ch = 0; ch < nch; ch++) {
        const unsigned share = left / (unsigned)((ngr - gr) * nch - ch);
        unsigned budget = share > 4095 ? 4095 : share, used = 0;
        unsigned line_cap = 576;
        if (ch == 1 && c->is_cut_pct > 0 && (int)rndn(100) < c->is_cut_pct) line_cap = 2 * rndn(240);
        gen_gc(c, &w, &gs[gr][ch], budget, gr, scfsi[ch], &used, pre_bt[gr][ch], pre_mixed[gr][ch], line_cap,
               c->version && ch == 1 && c->mode == 1 && (c->mode_ext & 1));
        left -= used;
      }
    /* side info */
    bw sw = {sides + (size_t)f * 40, 40, 0};
    if (c->version) {                       /* 13818-3 2.4.1.7: 8 + 1 (mono) / 2 (stereo) private bits, no scfsi */
      bw_put(&sw, begin, 8);
      bw_put(&sw, 0, nch == 1 ? 1 : 2);
    } else {
      bw_put(&sw, begin, 9);
      bw_put(&sw, 0, nch == 1 ? 5 : 3);
      for (int ch = 0; ch < nch; ch++) for (int b = 0; b < 4; b++) bw_put(&sw, scfsi[ch][b], 1);
    }
    for (int gr = 0; gr < ngr; gr++)
      for (int ch = 0; ch < nch; ch++) {
        const gc_side* s = &gs[gr][ch];
        bw_put(&sw, s->part2_3_length, 12); bw_put(&sw, s->big_values, 9); bw_put(&sw, s->global_gain, 8);
        bw_put(&sw, s->scalefac_compress, c->version ? 9 : 4); bw_put(&sw, s->win_switch, 1);
        if (s->win_switch) {
          bw_put(&sw, s->block_type, 2); bw_put(&sw, s->mixed, 1);
          bw_put(&sw, s->table_select[0], 5); bw_put(&sw, s->table_select[1], 5);
          for (int k = 0; k < 3; k++) bw_put(&sw, s->subblock_gain[k], 3);
        } else {
          for (int k = 0; k < 3; k++) bw_put(&sw, s->table_select[k], 5);
          bw_put(&sw, s->region0, 4); bw_put(&sw, s->region1, 3);
        }
        if (!c->version) bw_put(&sw, s->preflag, 1);
        bw_put(&sw, s->scalefac_scale, 1); bw_put(&sw, s->count1table, 1);
      }
    area_start += msize;
  }
  /* assemble */
  size_t o = 0, area = 0;
  for (int f = 0; f < n_frames; f++) {
    const size_t need = 4 + (c->crc ? 2 : 0) + side_bytes + fsize[f];
    if (o + need > cap) { o = 0; break; }
    out[o++] = (uint8_t)(hdrs[f] >> 24); out[o++] = (uint8_t)(hdrs[f] >> 16);
    out[o++] = (uint8_t)(hdrs[f] >> 8); out[o++] = (uint8_t)hdrs[f];
    if (c->crc) { out[o++] = (uint8_t)rndn(256); out[o++] = (uint8_t)rndn(256); }
    memcpy(out + o, sides + (size_t)f * 40, side_bytes); o += side_bytes;
    memcpy(out + o, md + area, fsize[f]); o += fsize[f];
    area += fsize[f];
  }
  free(md); free(sides); free(fsize); free(hdrs);
  return o;
}

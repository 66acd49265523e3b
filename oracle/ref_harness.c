/*
 * ref_harness.c -- builds the REAL reference (technosaurus/PDMP3 pdmp3.c) into
 * oracle/_ref/libpdmp3_ref.so, from the sources where they lie.
 *
 * Only buildable where /root/reference exists (the build container).  The
 * reference is one C file with no dependency but libc/libm, so it is compiled
 * as-is by `#include`-ing it into this translation unit (which makes its
 * `static` functions reachable) -- no reference source is copied into the repo
 * and nothing is stubbed.  Flags follow the reference Makefile's feature
 * defines (Makefile:23: IMDCT_TABLES, IMDCT_NTABLES, POW34_TABLE) with
 * OUTPUT_RAW instead of OUTPUT_SOUND and plain -O2 (bit-identical PCM to the
 * Makefile flag set).
 *
 * TEST INFRASTRUCTURE: used to pin oracle/pdmp3_oracle.c (bit-exact) and to
 * generate tests/golden/.  May also serve bench.py's cpu_baseline leg
 * (kind "reference").
 */
#define OUTPUT_RAW
#define IMDCT_TABLES
#define IMDCT_NTABLES
#define POW34_TABLE
#define NDEBUG            /* silences the ERR()/DBG() stderr chatter (H15) */

#ifndef REF_SRC
#define REF_SRC "/root/reference/pdmp3.c"
#endif
#include REF_SRC

#include <time.h>
#include "../include/pdmp3_hip.h"

/* The reference handle is malloc'ed uninitialised (H13); use zeroed memory. */
pdmp3_handle* ref_new(void) {
  pdmp3_handle* id = calloc(1, sizeof *id);
  pdmp3_open_feed(id);
  return id;
}
void ref_delete(pdmp3_handle* id) { free(id); }
size_t ref_sizeof_handle(void) { return sizeof(pdmp3_handle); }

static void ref_load_frame(pdmp3_handle* id, const int16_t* spectra, const pdmp3_gc_side* sd) {
  unsigned fr = sd[0].frame;
  id->g_frame_header.sampling_frequency = fr & PDMP3_FR_SFREQ_MASK;
  id->g_frame_header.mode = (fr & PDMP3_FR_MODE_MASK) >> PDMP3_FR_MODE_SHIFT;
  id->g_frame_header.mode_extension = (fr & PDMP3_FR_MODEEXT_MASK) >> PDMP3_FR_MODEEXT_SHIFT;
  if (fr & PDMP3_FR_RESET) { id->hsynth_init = 1; id->synth_init = 1; }
  for (unsigned gr = 0; gr < 2; gr++)
    for (unsigned ch = 0; ch < 2; ch++) {
      const pdmp3_gc_side* s = &sd[gr * 2 + ch];
      const int16_t* sp = spectra + (gr * 2 + ch) * 576;
      for (unsigned i = 0; i < 576; i++) id->g_main_data.is[gr][ch][i] = (float)sp[i];
      id->g_side_info.count1[gr][ch] = s->count1;
      id->g_side_info.global_gain[gr][ch] = s->global_gain;
      id->g_side_info.scalefac_scale[gr][ch] = (s->flags & PDMP3_GC_SCALEFAC_SCALE) ? 1 : 0;
      id->g_side_info.preflag[gr][ch] = (s->flags & PDMP3_GC_PREFLAG) ? 1 : 0;
      id->g_side_info.win_switch_flag[gr][ch] = (s->flags & PDMP3_GC_WIN_SWITCH) ? 1 : 0;
      id->g_side_info.block_type[gr][ch] = (s->flags & PDMP3_GC_BLOCK_TYPE_MASK) >> PDMP3_GC_BLOCK_TYPE_SHIFT;
      id->g_side_info.mixed_block_flag[gr][ch] = (s->flags & PDMP3_GC_MIXED) ? 1 : 0;
      for (unsigned k = 0; k < 3; k++) id->g_side_info.subblock_gain[gr][ch][k] = s->subblock_gain[k];
      /* Only the REAL entries are written: [21] and [12][w] are left to the
       * reference's own out-of-bounds reads, i.e. to its memory layout. */
      for (unsigned k = 0; k < 21; k++) id->g_main_data.scalefac_l[gr][ch][k] = s->scalefac_l[k];
      for (unsigned k = 0; k < 12; k++)
        for (unsigned w = 0; w < 3; w++) id->g_main_data.scalefac_s[gr][ch][k][w] = s->scalefac_s[k][w];
    }
}

static void ref_store_pcm(pdmp3_handle* id, int16_t* pcm) {
  size_t done = 0;
  id->ostart = 0;
  Convert_Frame_S16(id, (unsigned char*)pcm, 4608, &done);
}

/* n frames through the reference's Decode_L3 + Convert_Frame_S16. */
int ref_decode_frames(pdmp3_handle* id, const int16_t* spectra, const pdmp3_gc_side* side,
                      int n_frames, int16_t* pcm, float* stages) {
  for (int f = 0; f < n_frames; f++) {
    const pdmp3_gc_side* sd = side + (size_t)f * 4;
    ref_load_frame(id, spectra + (size_t)f * 2304, sd);
    if (!stages) {
      Decode_L3(id);
    } else {
      /* same calls in the same order as Decode_L3 (P:1029-1047), with dumps */
      float* stg = stages + (size_t)f * 4 * 4 * 576;
      unsigned nch = (id->g_frame_header.mode == mpeg1_mode_single_channel ? 1 : 2);
      for (unsigned gr = 0; gr < 2; gr++) {
        for (unsigned ch = 0; ch < nch; ch++) {
          L3_Requantize(id, gr, ch);
          L3_Reorder(id, gr, ch);
          memcpy(stg + ((gr * 2 + ch) * 4 + 0) * 576, id->g_main_data.is[gr][ch], 576 * 4);
        }
        L3_Stereo(id, gr);
        for (unsigned ch = 0; ch < nch; ch++)
          memcpy(stg + ((gr * 2 + ch) * 4 + 1) * 576, id->g_main_data.is[gr][ch], 576 * 4);
        for (unsigned ch = 0; ch < nch; ch++) {
          L3_Antialias(id, gr, ch);
          memcpy(stg + ((gr * 2 + ch) * 4 + 2) * 576, id->g_main_data.is[gr][ch], 576 * 4);
          L3_Hybrid_Synthesis(id, gr, ch);
          L3_Frequency_Inversion(id, gr, ch);
          memcpy(stg + ((gr * 2 + ch) * 4 + 3) * 576, id->g_main_data.is[gr][ch], 576 * 4);
          L3_Subband_Synthesis(id, gr, ch, id->out[gr]);
        }
      }
    }
    ref_store_pcm(id, pcm + (size_t)f * 2304);
  }
  return 0;
}

/* Seconds for `reps` passes of Decode_L3 over the given frames (cpu_baseline). */
double ref_time_decode(pdmp3_handle* id, const int16_t* spectra, const pdmp3_gc_side* side,
                       int n_frames, int reps) {
  struct timespec a, b;
  clock_gettime(CLOCK_MONOTONIC, &a);
  for (int r = 0; r < reps; r++)
    for (int f = 0; f < n_frames; f++) {
      ref_load_frame(id, spectra + (size_t)f * 2304, side + (size_t)f * 4);
      Decode_L3(id);
    }
  clock_gettime(CLOCK_MONOTONIC, &b);
  return (b.tv_sec - a.tv_sec) + 1e-9 * (b.tv_nsec - a.tv_nsec);
}

/* Snapshot of the just-parsed frame in gc-record form: what the reference's
 * transforms are about to read, INCLUDING its out-of-bounds scalefactor
 * reads, taken straight from its memory (flat indexing of the arrays). */
static void ref_tap_frame(pdmp3_handle* id, int16_t* spectra, pdmp3_gc_side* sd) {
  memset(sd, 0, 4 * sizeof *sd);
  const unsigned* sfl = &id->g_main_data.scalefac_l[0][0][0];
  const unsigned* sfs = &id->g_main_data.scalefac_s[0][0][0][0];
  unsigned nch = (id->g_frame_header.mode == mpeg1_mode_single_channel ? 1 : 2);
  for (unsigned gr = 0; gr < 2; gr++)
    for (unsigned ch = 0; ch < 2; ch++) {
      pdmp3_gc_side* s = &sd[gr * 2 + ch];
      int16_t* sp = spectra + (gr * 2 + ch) * 576;
      s->frame = (uint8_t)((id->g_frame_header.sampling_frequency & 3) |
                           ((id->g_frame_header.mode & 3) << PDMP3_FR_MODE_SHIFT) |
                           ((id->g_frame_header.mode_extension & 3) << PDMP3_FR_MODEEXT_SHIFT) |
                           ((id->hsynth_init || id->synth_init) ? PDMP3_FR_RESET : 0));
      if (ch >= nch) continue;
      for (unsigned i = 0; i < 576; i++) sp[i] = (int16_t)id->g_main_data.is[gr][ch][i];
      s->count1 = (uint16_t)id->g_side_info.count1[gr][ch];
      s->global_gain = (uint8_t)id->g_side_info.global_gain[gr][ch];
      unsigned fl = 0;
      if (id->g_side_info.scalefac_scale[gr][ch]) fl |= PDMP3_GC_SCALEFAC_SCALE;
      if (id->g_side_info.preflag[gr][ch]) fl |= PDMP3_GC_PREFLAG;
      if (id->g_side_info.win_switch_flag[gr][ch]) {
        fl |= PDMP3_GC_WIN_SWITCH;
        if (id->g_side_info.mixed_block_flag[gr][ch]) fl |= PDMP3_GC_MIXED;
      }
      fl |= (id->g_side_info.block_type[gr][ch] & 3) << PDMP3_GC_BLOCK_TYPE_SHIFT;
      s->flags = (uint8_t)fl;
      for (unsigned k = 0; k < 3; k++) s->subblock_gain[k] = (uint8_t)id->g_side_info.subblock_gain[gr][ch][k];
      unsigned g = gr * 2 + ch;
      for (unsigned k = 0; k < 22; k++) s->scalefac_l[k] = (uint8_t)sfl[g * 21 + k];
      for (unsigned k = 0; k < 13; k++)
        for (unsigned w = 0; w < 3; w++) {
          if (g == 3 && k == 12) s->scalefac_s[k][w] = PDMP3_SF_PEEK;   /* lands in is[0][0][w] */
          else s->scalefac_s[k][w] = (uint8_t)sfs[g * 36 + k * 3 + w];
        }
    }
}

/* The CLI driver loop pdmp3() (P:2552-2587) over a memory buffer instead of a
 * FILE, with an optional tap between Read_Frame and Decode_L3.  Without a tap
 * it goes through the real pdmp3_read; with one, pdmp3_read's loop
 * (P:2431-2481) is re-walked here so the tap can sit at P:2452. */
size_t ref_decode_buffer_like_cli(const unsigned char* mp3, size_t n, unsigned char* pcm, size_t pcm_cap,
                                  int16_t* tap_spectra, pdmp3_gc_side* tap_side, int tap_cap, int* tap_n) {
  pdmp3_handle* id = ref_new();
  unsigned char out[INBUF_SIZE];
  size_t done, total = 0, pos = 0;
  int res, nt = 0;
  for (;;) {
    if (!tap_side) {
      res = pdmp3_read(id, out, INBUF_SIZE, &done);
    } else {
      unsigned char* om = out;
      size_t outsize = INBUF_SIZE;
      done = 0;
      res = PDMP3_ERR;
      if (id->ostart) {
        Convert_Frame_S16(id, om, outsize, &done);
        om += done; outsize -= done; res = PDMP3_OK;
      }
      while (outsize) {
        if (Get_Inbuf_Filled(id) >= (2 * 576)) {
          size_t p0 = id->processed;
          unsigned mark = id->istart;
          res = Read_Frame(id);
          if (res == PDMP3_OK || res == PDMP3_NEW_FORMAT) {
            size_t batch;
            if (nt < tap_cap) { ref_tap_frame(id, tap_spectra + (size_t)nt * 2304, tap_side + (size_t)nt * 4); }
            nt++;
            Decode_L3(id);
            Convert_Frame_S16(id, om, outsize, &batch);
            om += batch; outsize -= batch; done += batch;
          } else { id->processed = p0; id->istart = mark; break; }
        } else { res = PDMP3_NEED_MORE; break; }
      }
      if (id->new_header == 1 && res == PDMP3_OK) res = PDMP3_NEW_FORMAT;
    }
    if (res == PDMP3_ERR) break;
    if (total + done <= pcm_cap) memcpy(pcm + total, out, done);
    total += done;
    if (res == PDMP3_NEED_MORE) {
      size_t k = n - pos; if (k > 4096) k = 4096;
      if (!k) break;
      pdmp3_feed(id, mp3 + pos, k);
      pos += k;
    }
  }
  if (tap_n) *tap_n = nt;
  ref_delete(id);
  return total;
}

/* --- the tree the standard means by table 33 sits in the reference's own array, 31 words
 * from g_huffman_table + 2773 (P:512-515), where g_huffman_main[33] fails to point (P:569 says + 2261).  The two
 * accessors below let the tests pin the PDMP3_ISO_TABLE33 code book on the reference's memory and on the
 * reference's own tree walk. --- */

/* words [first, first + n) of g_huffman_table (P:235-515) as they sit in this library's memory; returns the
 * number of words in the array */
unsigned ref_huffman_table_words(unsigned first, unsigned n, unsigned short* out) {
  const unsigned total = (unsigned)(sizeof g_huffman_table / sizeof g_huffman_table[0]);
  for (unsigned i = 0; i < n && first + i < total; i++) out[i] = g_huffman_table[first + i];
  return total;
}

/* the reference's Huffman_Decode (P:1593-1643) for table number 33 with g_huffman_main[33] pointed, for the length
 * of this call, `offset` words into g_huffman_table; the bit stream is the `nbits` (<= 32) bits of `bits`, msb
 * first, then zeroes.  out = {v, w, x, y, bits consumed}; returns Huffman_Decode's status */
int ref_huffman_quad_at(unsigned offset, unsigned bits, int nbits, int* out) {
  pdmp3_handle* id = ref_new();
  for (int i = 0; i < 8; i++) id->g_main_data_vec[i] = 0;
  for (int i = 0; i < nbits && i < 32; i++)
    if ((bits >> (nbits - 1 - i)) & 1u) id->g_main_data_vec[i >> 3] |= 0x80u >> (i & 7);
  Set_Main_Pos(id, 0);
  const hufftables saved = g_huffman_main[33];
  g_huffman_main[33].hufftable = g_huffman_table + offset;
  int32_t x = 0, y = 0, v = 0, w = 0;
  const int res = Huffman_Decode(id, 33, &x, &y, &v, &w);
  g_huffman_main[33] = saved;
  out[0] = v; out[1] = w; out[2] = x; out[3] = y; out[4] = (int)Get_Main_Pos(id);
  ref_delete(id);
  return res;
}

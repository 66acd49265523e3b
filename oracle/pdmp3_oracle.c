/*
 * pdmp3_oracle.c -- CPU ORACLE (transform path).  TEST INFRASTRUCTURE ONLY.
 * See pdmp3_oracle.h for the rules and the parity status (PINNED against the
 * compiled reference in oracle/_ref and the fixtures in tests/golden/).
 *
 * Restates, operation by operation and in the reference's evaluation order,
 * what Decode_L3 (P:1024-1060) does with one parsed frame.  Every function
 * names the reference lines it follows.  Build with -O2 -ffp-contract=off and
 * WITHOUT -ffast-math: float expressions are binary32, double sub-expressions
 * binary64, exactly as in the reference's x86-64 build.
 *
 * "P:n" = /root/reference/pdmp3.c line n.
 */
#include "pdmp3_oracle.h"
#include "oracle_tables.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define O_PI          3.14159265358979323846   /* P:166 */
#define O_INV_SQRT_2  0.70710678118654752440   /* P:167 */

/* ------------------------------------------------------------------ */
/* libm-derived tables, built with the reference's own expressions     */
/* ------------------------------------------------------------------ */
static float g_pow43[8207];
static float g_nwin[64][32];
static int g_tables_ready = 0;

static void build_tables(void) {
  if (g_tables_ready) return;
  for (int i = 0; i < 8207; i++)                       /* P:978-979 */
    g_pow43[i] = pow((float)i, 4.0 / 3.0);
  for (int i = 0; i < 64; i++)                         /* P:1990-1993 */
    for (int j = 0; j < 32; j++)
      g_nwin[i][j] = cos(((float)(16 + i) * (2 * j + 1)) * (O_PI / 64.0));
  g_tables_ready = 1;
}

const float* orc_table_pow43(void) { build_tables(); return g_pow43; }
const float* orc_table_nwin(void) { build_tables(); return &g_nwin[0][0]; }

void orc_synth_reset(orc_synth* st) { memset(st, 0, sizeof *st); }

/* sfb boundary accessors over the contiguous l[23] s[14] layout (P:108-112) */
/* sfreq 3..8: MPEG-2 LSF / MPEG-2.5 (ISO/IEC 13818-3 table B.8) -- NOT the reference, which decodes MPEG-1 only
 * (P:1293); restated independently of pdmp3_amd/csrc/lsf_tables.h, both pinned by FFmpeg (tests/golden/lsf_*.npz) */
static const uint16_t ot_lsf_l[6][23] = {
  {0, 6, 12, 18, 24, 30, 36, 44, 54, 66, 80, 96, 116, 140, 168, 200, 238, 284, 336, 396, 464, 522, 576},   /* 22.05 kHz */
  {0, 6, 12, 18, 24, 30, 36, 44, 54, 66, 80, 96, 114, 136, 162, 194, 232, 278, 332, 394, 464, 540, 576},   /* 24 */
  {0, 6, 12, 18, 24, 30, 36, 44, 54, 66, 80, 96, 116, 140, 168, 200, 238, 284, 336, 396, 464, 522, 576},   /* 16 */
  {0, 6, 12, 18, 24, 30, 36, 44, 54, 66, 80, 96, 116, 140, 168, 200, 238, 284, 336, 396, 464, 522, 576},   /* 11.025 */
  {0, 6, 12, 18, 24, 30, 36, 44, 54, 66, 80, 96, 116, 140, 168, 200, 238, 284, 336, 396, 464, 522, 576},   /* 12 */
  {0, 12, 24, 36, 48, 60, 72, 88, 108, 132, 160, 192, 232, 280, 336, 400, 476, 566, 568, 570, 572, 574, 576},   /* 8 */
};
static const uint16_t ot_lsf_s[6][14] = {
  {0, 4, 8, 12, 18, 24, 32, 42, 56, 74, 100, 132, 174, 192},
  {0, 4, 8, 12, 18, 26, 36, 48, 62, 80, 104, 136, 180, 192},
  {0, 4, 8, 12, 18, 26, 36, 48, 62, 80, 104, 134, 174, 192},
  {0, 4, 8, 12, 18, 26, 36, 48, 62, 80, 104, 134, 174, 192},
  {0, 4, 8, 12, 18, 26, 36, 48, 62, 80, 104, 134, 174, 192},
  {0, 8, 16, 24, 36, 52, 72, 96, 124, 160, 162, 164, 166, 192},
};
/* orc_debug_24k_330 (tests only): band 18 of the 24 kHz long table starts at line 330, as in FFmpeg's and mpg123's
 * tables (the standard, LAME, libmad and minimp3 have 332): shows that the 24 kHz fixtures' residual is that one entry */
int g_orc_24k_330 = 0;
void orc_debug_24k_330(int on) { g_orc_24k_330 = on; }
static inline unsigned sfb_l(unsigned sfreq, unsigned i) {
  if (sfreq == 4 && i == 18 && g_orc_24k_330) return 330;
  return sfreq < 3 ? ot_sfb[sfreq * 37 + i] : ot_lsf_l[sfreq - 3][i];
}
static inline unsigned sfb_s(unsigned sfreq, unsigned i) { return sfreq < 3 ? ot_sfb[sfreq * 37 + 23 + i] : ot_lsf_s[sfreq - 3][i]; }

/* One frame's working set: the reference's id->g_main_data.is (P:99), in
 * place through all stages, plus unpacked side info. */
typedef struct frame_ws {
  float is[2][2][576];
  int16_t q[2][2][576];         /* the coded integers as they came (before P:1829 / P:1786): stage_stereo_std's zero test */
  unsigned count1[2][2], global_gain[2][2], scalefac_scale[2][2], preflag[2][2];
  unsigned wsf[2][2], block_type[2][2], mixed[2][2], subblock_gain[2][2][3];
  uint32_t sf_l[2][2][22];      /* value as the reference would read it */
  uint32_t sf_s[2][2][13][3];
  int sf_s_peek[2][2];          /* H5: [12][w] comes from the bits of is[0][0][w] */
  unsigned iso;                 /* PDMP3_GC_ISO_* of the frame's records: the standard's behaviour instead of H2 / H3 */
  unsigned sfreq, mode, mode_ext, nch;     /* sfreq: 0..2 MPEG-1, 3..8 LSF (3 * version + the header's field) */
  unsigned ver;                 /* 0 MPEG-1, 1 MPEG-2 LSF, 2 MPEG-2.5: one granule, the standard's behaviour throughout */
  unsigned lsf_is_scale, lsf_slen[4], lsf_nsfb[4];   /* LSF intensity stereo: channel 1's intensity_scale and partitions */
} frame_ws;

/* float value -> what `Requantize_Pow_43(is)` indexes with (P:2129-2131):
 * the float is converted to the `unsigned` parameter. */
static inline float pow43_signed(float v) {
  if (v < 0.0) return -g_pow43[(unsigned)(-v)];
  return g_pow43[(unsigned)v];
}

/* P:2121-2134 Requantize_Process_Long */
static void requant_long(frame_ws* w, unsigned gr, unsigned ch, unsigned i, unsigned sfb) {
  float sf_mult = w->scalefac_scale[gr][ch] ? 1.0 : 0.5;
  float pf_x_pt = w->preflag[gr][ch] * (float)ot_pretab[sfb];
  float tmp1 = pow(2.0, -(sf_mult * (w->sf_l[gr][ch][sfb] + pf_x_pt)));
  float tmp2 = pow(2.0, 0.25 * ((int32_t)w->global_gain[gr][ch] - 210));
  float tmp3 = pow43_signed(w->is[gr][ch][i]);
  w->is[gr][ch][i] = tmp1 * tmp2 * tmp3;
}

/* P:2140-2152 Requantize_Process_Short */
static void requant_short(frame_ws* w, unsigned gr, unsigned ch, unsigned i, unsigned sfb, unsigned win) {
  float sf_mult = w->scalefac_scale[gr][ch] ? 1.0f : 0.5f;
  uint32_t sf = w->sf_s[gr][ch][sfb][win];
  if (sfb == 12 && w->sf_s_peek[gr][ch]) {           /* H5: bits of is[0][0][win] */
    memcpy(&sf, &w->is[0][0][win], 4);
  }
  float tmp1 = pow(2.0f, -(sf_mult * sf));
  float tmp2 = pow(2.0f, 0.25f * ((float)w->global_gain[gr][ch] - 210.0f -
                                  8.0f * (float)w->subblock_gain[gr][ch][win]));
  float tmp3 = pow43_signed(w->is[gr][ch][i]);
  w->is[gr][ch][i] = tmp1 * tmp2 * tmp3;
}

/* P:1829-1905 L3_Requantize */
static void stage_requantize(frame_ws* w, unsigned gr, unsigned ch) {
  unsigned sfreq = w->sfreq, sfb, next_sfb, i, j, win, win_len;
  unsigned count1 = w->count1[gr][ch];
  if (w->wsf[gr][ch] == 1 && w->block_type[gr][ch] == 2) {
    if (w->mixed[gr][ch] != 0) {
      sfb = 0;
      next_sfb = sfb_l(sfreq, sfb + 1);
      for (i = 0; i < 36; i++) {                       /* P:1843-1849: no count1 bound */
        if (i == next_sfb) { sfb++; next_sfb = sfb_l(sfreq, sfb + 1); }
        requant_long(w, gr, ch, i, sfb);
      }
      sfb = 3;
      next_sfb = sfb_s(sfreq, sfb + 1) * 3;
      win_len = sfb_s(sfreq, sfb + 1) - sfb_s(sfreq, sfb);
      for (i = 36; i < count1; ) {
        if (i == next_sfb) {
          sfb++;
          next_sfb = sfb_s(sfreq, sfb + 1) * 3;
          win_len = sfb_s(sfreq, sfb + 1) - sfb_s(sfreq, sfb);
        }
        for (win = 0; win < 3; win++)
          for (j = 0; j < win_len; j++) { requant_short(w, gr, ch, i, sfb, win); i++; }
      }
    } else {
      sfb = 0;
      next_sfb = sfb_s(sfreq, sfb + 1) * 3;
      win_len = sfb_s(sfreq, sfb + 1) - sfb_s(sfreq, sfb);
      for (i = 0; i < count1; ) {
        if (i == next_sfb) {
          sfb++;
          next_sfb = sfb_s(sfreq, sfb + 1) * 3;
          win_len = sfb_s(sfreq, sfb + 1) - sfb_s(sfreq, sfb);
        }
        for (win = 0; win < 3; win++)
          for (j = 0; j < win_len; j++) { requant_short(w, gr, ch, i, sfb, win); i++; }
      }
    }
  } else {
    sfb = 0;
    next_sfb = sfb_l(sfreq, sfb + 1);
    for (i = 0; i < count1; i++) {
      if (i == next_sfb) { sfb++; next_sfb = sfb_l(sfreq, sfb + 1); }
      requant_long(w, gr, ch, i, sfb);
    }
  }
}

/* P:1786-1823 L3_Reorder */
static void stage_reorder(frame_ws* w, unsigned gr, unsigned ch) {
  unsigned sfreq = w->sfreq, i, j, next_sfb, sfb, win_len, win;
  float re[576];
  float* x = w->is[gr][ch];
  if (!(w->wsf[gr][ch] == 1 && w->block_type[gr][ch] == 2)) return;
  sfb = (w->mixed[gr][ch] != 0) ? 3 : 0;
  next_sfb = sfb_s(sfreq, sfb + 1) * 3;
  win_len = sfb_s(sfreq, sfb + 1) - sfb_s(sfreq, sfb);
  for (i = (sfb == 0) ? 0 : 36; i < 576; ) {
    if (i == next_sfb) {
      for (j = 0; j < 3 * win_len; j++) x[3 * sfb_s(sfreq, sfb) + j] = re[j];
      if (i >= w->count1[gr][ch]) return;              /* P:1806 early out */
      sfb++;
      next_sfb = sfb_s(sfreq, sfb + 1) * 3;
      win_len = sfb_s(sfreq, sfb + 1) - sfb_s(sfreq, sfb);
    }
    for (win = 0; win < 3; win++)
      for (j = 0; j < win_len; j++) { re[j * 3 + win] = x[i]; i++; }
  }
  for (j = 0; j < 3 * win_len; j++) x[3 * sfb_s(sfreq, 12) + j] = re[j];
}

/* P:2158-2183 Stereo_Process_Intensity_Long */
static void intensity_long(frame_ws* w, unsigned gr, unsigned sfb) {
  unsigned is_pos = w->sf_l[gr][0][sfb];
  if (is_pos == 7) return;
  unsigned a = sfb_l(w->sfreq, sfb), b = sfb_l(w->sfreq, sfb + 1);
  float rl, rr;
  if (is_pos == 6) { rl = 1.0f; rr = 0.0f; }
  else {
    /* is_pos > 5 reads past is_ratios[6] in the reference (H3): undefined,
     * excluded from parity corpora; the oracle reads 0 there. */
    float t = (is_pos < 6) ? ot_is_ratios[is_pos] : 0.0f;
    rl = t / (1.0f + t);
    rr = 1.0f / (1.0f + t);
  }
  for (unsigned i = a; i < b; i++) {
    float left = rl * w->is[gr][0][i];
    float right = rr * w->is[gr][0][i];
    w->is[gr][0][i] = left;
    w->is[gr][1][i] = right;
  }
}

/* P:2190-2220 Stereo_Process_Intensity_Short.  The reference assigns the
 * sample to an `unsigned` ratio variable and back (H3); for negative samples
 * that conversion is undefined behaviour -- restated here as what gcc/x86-64
 * emits (cvttss2si to 64 bit, low 32 bits kept).  Excluded from parity corpora. */
static void intensity_short(frame_ws* w, unsigned gr, unsigned sfb) {
  unsigned win_len = sfb_s(w->sfreq, sfb + 1) - sfb_s(w->sfreq, sfb);
  if (w->iso & PDMP3_GC_ISO_IS_SHORT) {
    /* NOT the reference ("ISO-correct switches", unpinned): the band's lines are in reordered order by
     * now (P:1786 ran before): line a + 3 j + win belongs to window win; left / right = the ratios of intensity_long
     * times the sample */
    unsigned a = sfb_s(w->sfreq, sfb) * 3;
    for (unsigned i = a; i < a + 3 * win_len; i++) {
      unsigned is_pos = w->sf_s[gr][0][sfb][(i - a) % 3];
      if (is_pos == 7) continue;
      float rl, rr;
      if (is_pos == 6) { rl = 1.0f; rr = 0.0f; }
      else {
        float t = (is_pos < 6) ? ot_is_ratios[is_pos] : 0.0f;
        rl = t / (1.0f + t);
        rr = 1.0f / (1.0f + t);
      }
      float left = rl * w->is[gr][0][i];
      float right = rr * w->is[gr][0][i];
      w->is[gr][0][i] = left;
      w->is[gr][1][i] = right;
    }
    return;
  }
  for (unsigned win = 0; win < 3; win++) {
    unsigned is_pos = w->sf_s[gr][0][sfb][win];
    if (is_pos == 7) continue;
    unsigned a = sfb_s(w->sfreq, sfb) * 3 + win_len * win, b = a + win_len;
    for (unsigned i = a; i < b; i++) {
      float x = w->is[gr][0][i];
      uint32_t u = (uint32_t)(int64_t)x;
      float v = (float)u;
      w->is[gr][0][i] = v;
      w->is[gr][1][i] = v;
    }
  }
}

/* LSF: the "not intensity coded" value of the right channel's scalefactor number bi (transmission order) */
static unsigned lsf_illegal(const frame_ws* w, unsigned bi) {
  unsigned acc = 0;

  for (unsigned k = 0; k < 4; k++) {
    acc += w->lsf_nsfb[k];
    if (bi < acc) return (1u << w->lsf_slen[k]) - 1u;
  }
  return 0;                      /* beyond the transmitted ones: scalefactor 0, slen 0 */
}

/* NOT the reference: joint stereo with intensity positions as ISO 11172-3 2.4.3.4.9.3 has them (PDMP3_GC_ISO_IS_STD;
 * pinned against FFmpeg's mpegaudiodec through tests/golden/iso_*.npz, tools/make_iso_golden.py).  Where the reference
 * departs (beyond SURVEY H3): it takes is_pos from the LEFT channel's scalefactors (P:2163, P:2200), bounds the
 * intensity region by count1 of the right channel (P:1946-1965) instead of by its last non-zero line -- per window in
 * short blocks --, leaves the last band (long 21 / short 12) out (P:1961, P:1953) where the standard gives it the
 * position of the band below, and rotates M/S over lines that are intensity coded.  Lines are in reordered order here
 * (P:1786 ran): in a short band starting at line a, line a + 3 j + win belongs to window win. */
static void stage_stereo_std(frame_ws* w, unsigned gr) {
  const unsigned sfreq = w->sfreq;
  const int shortb = (w->wsf[gr][1] == 1 && w->block_type[gr][1] == 2), mixed = shortb && w->mixed[gr][1] != 0;
  const unsigned nlong_mixed = w->ver ? 6 : 8;             /* a mixed block's long part: 36 lines = 8 MPEG-1 bands, 6 LSF bands */
  const unsigned NONE = 255;
  uint8_t is_pos_of[576];                                  /* NONE = not intensity coded */
  memset(is_pos_of, NONE, sizeof is_pos_of);
  /* the position that means "not intensity coded": 7 in MPEG-1; with LSF the largest value the scalefactor's slen holds
   * (13818-3 2.4.3.2), by partition of the right channel's scalefactors in transmission order */
#define ILLEGAL_OF(bi) (w->ver ? lsf_illegal(w, (bi)) : 7u)
  if (w->mode_ext & 0x1) {
    /* "zero" is said of the CODED value (2.4.3.4.9.3), so the test reads the integers, which are in bitstream order:
     * short band sfb, window win = lines 3 s[sfb] + win len .. + len */
    const int16_t* q = w->q[gr][1];
    int any_short = 0;
    if (shortb) {
      const unsigned first = mixed ? 3 : 0;
      for (unsigned win = 0; win < 3; win++) {
        int last = -1;                                     /* last band of this window with a non-zero line */
        for (unsigned sfb = first; sfb < 13; sfb++) {
          const unsigned a = 3 * sfb_s(sfreq, sfb), len = sfb_s(sfreq, sfb + 1) - sfb_s(sfreq, sfb);
          for (unsigned j = 0; j < len; j++) if (q[a + win * len + j] != 0) last = (int)sfb;
        }
        if (last >= 0) any_short = 1;
        for (unsigned sfb = first; sfb < 13; sfb++) {
          const unsigned a = 3 * sfb_s(sfreq, sfb), len = sfb_s(sfreq, sfb + 1) - sfb_s(sfreq, sfb);
          if ((int)sfb <= last) continue;                  /* a non-zero line in this band or above: not intensity */
          const unsigned sb = sfb < 12 ? sfb : 11;
          const unsigned pos = w->sf_s[gr][1][sb][win];
          const unsigned ill = ILLEGAL_OF((mixed ? nlong_mixed + (sb - 3) * 3 : sb * 3) + win);
          if (w->ver ? pos == ill : pos >= 7) continue;
          for (unsigned j = 0; j < len; j++) is_pos_of[a + 3 * j + win] = (uint8_t)pos;      /* reordered position */
        }
      }
    }
    if (!shortb || (mixed && !any_short)) {
      const unsigned nb = shortb ? nlong_mixed : 22, top = shortb ? 36 : 576;
      int last = -1;
      for (unsigned i = 0; i < top; i++) if (q[i] != 0) last = (int)i;
      for (unsigned sfb = 0; sfb < nb; sfb++) {
        const unsigned a = sfb_l(sfreq, sfb), b = sfb < 21 ? sfb_l(sfreq, sfb + 1) : 576;
        if ((int)a <= last) continue;
        const unsigned pos = w->sf_l[gr][1][sfb < 21 ? sfb : 20];
        if (w->ver ? pos == ILLEGAL_OF(sfb < 21 ? sfb : 20) : pos >= 7) continue;
        for (unsigned i = a; i < b; i++) is_pos_of[i] = (uint8_t)pos;
      }
    }
  }
  if (w->mode_ext & 0x2) {
    unsigned c0 = w->count1[gr][0], c1 = w->count1[gr][1];
    unsigned max_pos = (c0 > c1) ? c1 : c0;
    if (w->iso & PDMP3_GC_ISO_MS_ALL) max_pos = 576;
    for (unsigned i = 0; i < max_pos; i++) {
      if (is_pos_of[i] != NONE) continue;
      float left = (w->is[gr][0][i] + w->is[gr][1][i]) * (O_INV_SQRT_2);
      float right = (w->is[gr][0][i] - w->is[gr][1][i]) * (O_INV_SQRT_2);
      w->is[gr][0][i] = left;
      w->is[gr][1][i] = right;
    }
  }
#undef ILLEGAL_OF
  for (unsigned i = 0; i < 576; i++) {
    const unsigned pos = is_pos_of[i];
    if (pos == NONE) continue;
    float rl, rr;
    if (w->ver) {
      /* 13818-3 2.4.3.2: i0 = 2^(-1/4) (intensity_scale 0) or 2^(-1/2); p odd: left x i0^((p + 1) / 2), p even: right x i0^(p / 2) */
      const float f = (float)pow(2.0, -(double)((w->lsf_is_scale + 1) * ((pos + 1) >> 1)) / 4.0);
      rl = (pos & 1) ? f : 1.0f;
      rr = (pos & 1) ? 1.0f : f;
    } else if (pos == 6) { rl = 1.0f; rr = 0.0f; }
    else { float t = ot_is_ratios[pos]; rl = t / (1.0f + t); rr = 1.0f / (1.0f + t); }
    float left = rl * w->is[gr][0][i];
    float right = rr * w->is[gr][0][i];
    w->is[gr][0][i] = left;
    w->is[gr][1][i] = right;
  }
}

/* P:1911-1972 L3_Stereo */
static void stage_stereo(frame_ws* w, unsigned gr) {
  if (w->mode != 1 || w->mode_ext == 0) return;
  if ((w->iso & PDMP3_GC_ISO_IS_STD) && ((w->mode_ext & 0x1) || w->ver)) { stage_stereo_std(w, gr); return; }
  if (w->mode_ext & 0x2) {
    unsigned c0 = w->count1[gr][0], c1 = w->count1[gr][1];
    unsigned max_pos = (c0 > c1) ? c1 : c0;            /* P:1920 picks the smaller (H2) */
    /* the standard's: every line.  (Lines at and above the larger count1 are zero in both channels before the reorder,
     * but P:1786 ran: in a short band that count1 cuts through, lines of the coded windows sit ABOVE count1 in reordered
     * order -- "up to the larger count1", rounds 4-5, left those unrotated; found against FFmpeg in round 6) */
    if (w->iso & PDMP3_GC_ISO_MS_ALL) max_pos = 576;
    for (unsigned i = 0; i < max_pos; i++) {
      float left = (w->is[gr][0][i] + w->is[gr][1][i]) * (O_INV_SQRT_2);
      float right = (w->is[gr][0][i] - w->is[gr][1][i]) * (O_INV_SQRT_2);
      w->is[gr][0][i] = left;
      w->is[gr][1][i] = right;
    }
  }
  if (w->mode_ext & 0x1) {
    unsigned sfreq = w->sfreq, sfb, c1 = w->count1[gr][1];
    if (w->wsf[gr][0] == 1 && w->block_type[gr][0] == 2) {
      if (w->mixed[gr][0] != 0) {
        for (sfb = 0; sfb < 8; sfb++)
          if (sfb_l(sfreq, sfb) >= c1) intensity_long(w, gr, sfb);
        for (sfb = 3; sfb < 12; sfb++)
          if (sfb_s(sfreq, sfb) * 3 >= c1) intensity_short(w, gr, sfb);
      } else {
        for (sfb = 0; sfb < 12; sfb++)
          if (sfb_s(sfreq, sfb) * 3 >= c1) intensity_short(w, gr, sfb);
      }
    } else {
      for (sfb = 0; sfb < 21; sfb++)
        if (sfb_l(sfreq, sfb) >= c1) intensity_long(w, gr, sfb);
    }
  }
}

/* P:1706-1732 L3_Antialias */
static void stage_antialias(frame_ws* w, unsigned gr, unsigned ch) {
  int pure_short = (w->wsf[gr][ch] == 1 && w->block_type[gr][ch] == 2 && w->mixed[gr][ch] == 0);
  if (pure_short) return;
  unsigned sblim = (w->wsf[gr][ch] == 1 && w->block_type[gr][ch] == 2 && w->mixed[gr][ch] == 1) ? 2 : 32;
  float* x = w->is[gr][ch];
  for (unsigned sb = 1; sb < sblim; sb++)
    for (unsigned i = 0; i < 8; i++) {
      unsigned li = 18 * sb - 1 - i, ui = 18 * sb + i;
      float lb = x[li] * ot_cs[i] - x[ui] * ot_ca[i];
      float ub = x[ui] * ot_cs[i] + x[li] * ot_ca[i];
      x[li] = lb;
      x[ui] = ub;
    }
}

/* P:1649-1700 IMDCT_Win (table build: IMDCT_TABLES + IMDCT_NTABLES) */
static void imdct_win(const float in[18], float out[36], unsigned bt) {
  float tin[18], sum;
  unsigned i, m, p;
  for (i = 0; i < 36; i++) out[i] = 0.0;
  for (i = 0; i < 18; i++) tin[i] = in[i];
  if (bt == 2) {
    for (i = 0; i < 3; i++)
      for (p = 0; p < 12; p++) {
        sum = 0.0;
        for (m = 0; m < 6; m++) sum += tin[i + 3 * m] * ot_cos_n12[m * 12 + p];
        out[6 * i + p + 6] += sum * ot_imdct_win[2 * 36 + p];
      }
  } else {
    for (p = 0; p < 36; p++) {
      sum = 0.0;
      for (m = 0; m < 18; m++) sum += in[m] * ot_cos_n36[m * 36 + p];
      out[p] = sum * ot_imdct_win[bt * 36 + p];
    }
  }
}

/* P:1752-1780 L3_Hybrid_Synthesis */
static void stage_hybrid(frame_ws* w, orc_synth* st, unsigned gr, unsigned ch) {
  float rawout[36];
  float* x = w->is[gr][ch];
  for (unsigned sb = 0; sb < 32; sb++) {
    unsigned bt = (w->wsf[gr][ch] == 1 && w->mixed[gr][ch] == 1 && sb < 2) ? 0 : w->block_type[gr][ch];
    imdct_win(&x[sb * 18], rawout, bt);
    for (unsigned i = 0; i < 18; i++) {
      x[sb * 18 + i] = rawout[i] + st->store[ch][sb][i];
      st->store[ch][sb][i] = rawout[i + 18];
    }
  }
}

/* P:1738-1746 L3_Frequency_Inversion */
static void stage_freqinv(frame_ws* w, unsigned gr, unsigned ch) {
  float* x = w->is[gr][ch];
  for (unsigned sb = 1; sb < 32; sb += 2)
    for (unsigned i = 1; i < 18; i += 2) x[sb * 18 + i] = -x[sb * 18 + i];
}

/* float -> int32 of P:2028 as x86-64 performs it (cvttsd2si): out-of-range
 * and NaN give INT32_MIN ("integer indefinite"). */
static inline int32_t cvt_trunc_x86(double d) {
  if (!(d > -2147483649.0 && d < 2147483648.0)) return INT32_MIN;
  return (int32_t)d;
}

/* P:1978-2045 L3_Subband_Synthesis.  out[] is the reference's `outdata`
 * (id->out[gr]): ch0 in the high half, ch1 OR-ed into the low half. */
static void stage_subband(frame_ws* w, orc_synth* st, unsigned gr, unsigned ch, uint32_t out[576], float* fsum) {
  float u_vec[512], s_vec[32], sum;
  float* v = st->v_vec[ch];
  const float* x = w->is[gr][ch];
  unsigned i, j, ss;
  for (ss = 0; ss < 18; ss++) {
    for (i = 1023; i > 63; i--) v[i] = v[i - 64];
    for (i = 0; i < 32; i++) s_vec[i] = x[i * 18 + ss];
    for (i = 0; i < 64; i++) {
      sum = 0.0;
      for (j = 0; j < 32; j++) sum += g_nwin[i][j] * s_vec[j];
      v[i] = sum;
    }
    for (i = 0; i < 8; i++)
      for (j = 0; j < 32; j++) {
        u_vec[(i << 6) + j] = v[(i << 7) + j];
        u_vec[(i << 6) + j + 32] = v[(i << 7) + j + 96];
      }
    for (i = 0; i < 512; i++) u_vec[i] = u_vec[i] * ot_synth_dtbl[i];
    for (i = 0; i < 32; i++) {
      sum = 0.0;
      for (j = 0; j < 16; j++) sum += u_vec[(j << 5) + i];
      if (fsum) fsum[32 * ss + i] = sum;                    /* the binary32 `sum` of P:2028, before the scaling */
      int32_t samp = cvt_trunc_x86(sum * 32767.0);
      if (samp > 32767) samp = 32767;
      else if (samp < -32767) samp = -32767;
      samp &= 0xffff;
      if (ch == 0) {
        if (w->nch == 1) out[32 * ss + i] = ((uint32_t)samp << 16) | (uint32_t)samp;
        else out[32 * ss + i] = (uint32_t)samp << 16;
      } else {
        out[32 * ss + i] |= (uint32_t)samp;
      }
    }
  }
}

static void unpack_frame(frame_ws* w, const int16_t* spectra, const pdmp3_gc_side* side) {
  unsigned fr = side[0].frame;
  w->sfreq = fr & PDMP3_FR_SFREQ_MASK;
  w->mode = (fr & PDMP3_FR_MODE_MASK) >> PDMP3_FR_MODE_SHIFT;
  w->mode_ext = (fr & PDMP3_FR_MODEEXT_MASK) >> PDMP3_FR_MODEEXT_SHIFT;
  w->nch = (w->mode == 3) ? 1 : 2;
  w->iso = side[0].iso;
  w->ver = side[0].lsf & PDMP3_LSF_VERSION_MASK;
  if (w->ver > 2) w->ver = 2;
  if (w->ver) {
    w->sfreq = 3 * w->ver + (w->sfreq > 2 ? 2 : w->sfreq);
    w->iso |= PDMP3_GC_ISO_MS_ALL | PDMP3_GC_ISO_IS_SHORT | PDMP3_GC_ISO_IS_STD;      /* no reference behaviour exists for LSF */
    w->lsf_is_scale = (side[1].lsf & PDMP3_LSF_IS_SCALE) ? 1 : 0;
    for (unsigned k = 0; k < 4; k++) { w->lsf_slen[k] = side[1].lsf_slen[k]; w->lsf_nsfb[k] = side[1].lsf_nsfb[k]; }
  }
  for (unsigned gr = 0; gr < 2; gr++)
    for (unsigned ch = 0; ch < 2; ch++) {
      const pdmp3_gc_side* s = &side[gr * 2 + ch];
      const int16_t* sp = spectra + (gr * 2 + ch) * 576;
      for (unsigned i = 0; i < 576; i++) { w->is[gr][ch][i] = (float)sp[i]; w->q[gr][ch][i] = sp[i]; }
      w->count1[gr][ch] = s->count1;
      w->global_gain[gr][ch] = s->global_gain;
      w->scalefac_scale[gr][ch] = (s->flags & PDMP3_GC_SCALEFAC_SCALE) ? 1 : 0;
      w->preflag[gr][ch] = (s->flags & PDMP3_GC_PREFLAG) ? 1 : 0;
      w->wsf[gr][ch] = (s->flags & PDMP3_GC_WIN_SWITCH) ? 1 : 0;
      w->block_type[gr][ch] = (s->flags & PDMP3_GC_BLOCK_TYPE_MASK) >> PDMP3_GC_BLOCK_TYPE_SHIFT;
      w->mixed[gr][ch] = (s->flags & PDMP3_GC_MIXED) ? 1 : 0;
      for (unsigned k = 0; k < 3; k++) w->subblock_gain[gr][ch][k] = s->subblock_gain[k];
      for (unsigned k = 0; k < 22; k++) w->sf_l[gr][ch][k] = s->scalefac_l[k];
      w->sf_s_peek[gr][ch] = 0;
      for (unsigned k = 0; k < 13; k++)
        for (unsigned win = 0; win < 3; win++) w->sf_s[gr][ch][k][win] = s->scalefac_s[k][win];
      if (s->scalefac_s[12][0] == PDMP3_SF_PEEK) w->sf_s_peek[gr][ch] = 1;
    }
}

int orc_decode_frames(orc_synth* st, const int16_t* spectra, const pdmp3_gc_side* side,
                      int n_frames, int16_t* pcm, float* stages) {
  return orc_decode_frames_f32(st, spectra, side, n_frames, pcm, NULL, stages);
}

/* the same, also handing out the synthesis sums P:2028 turns into int16: pcm_f32 (nullable), interleaved like pcm,
 * 2304 floats per frame (mono: first 1152); pcm (nullable) */
int orc_decode_frames_f32(orc_synth* st, const int16_t* spectra, const pdmp3_gc_side* side,
                          int n_frames, int16_t* pcm, float* pcm_f32, float* stages) {
  build_tables();
  frame_ws* w = (frame_ws*)malloc(sizeof *w);
  uint32_t out[2][576];
  float fs[2][2][576];
  for (int f = 0; f < n_frames; f++) {
    const pdmp3_gc_side* sd = side + (size_t)f * 4;
    unpack_frame(w, spectra + (size_t)f * 2304, sd);
    if (sd[0].frame & PDMP3_FR_RESET) orc_synth_reset(st);   /* P:1757-1766, P:1996-2003 */
    float* stg = stages ? stages + (size_t)f * 4 * 4 * 576 : NULL;
    /* P:1029-1047 Decode_L3 (an LSF frame has one granule: the records [1][ch] do not exist) */
    const unsigned ngr = w->ver ? 1 : 2;
    for (unsigned gr = 0; gr < ngr; gr++) {
      for (unsigned ch = 0; ch < w->nch; ch++) {
        stage_requantize(w, gr, ch);
        stage_reorder(w, gr, ch);
        if (stg) memcpy(stg + ((gr * 2 + ch) * 4 + 0) * 576, w->is[gr][ch], 576 * 4);
      }
      stage_stereo(w, gr);
      if (stg)
        for (unsigned ch = 0; ch < w->nch; ch++)
          memcpy(stg + ((gr * 2 + ch) * 4 + 1) * 576, w->is[gr][ch], 576 * 4);
      for (unsigned ch = 0; ch < w->nch; ch++) {
        stage_antialias(w, gr, ch);
        if (stg) memcpy(stg + ((gr * 2 + ch) * 4 + 2) * 576, w->is[gr][ch], 576 * 4);
        stage_hybrid(w, st, gr, ch);
        stage_freqinv(w, gr, ch);
        if (stg) memcpy(stg + ((gr * 2 + ch) * 4 + 3) * 576, w->is[gr][ch], 576 * 4);
        stage_subband(w, st, gr, ch, out[gr], fs[gr][ch]);
      }
    }
    /* P:2307-2345 Convert_Frame_S16 for a whole frame */
    if (pcm_f32) {
      float* of = pcm_f32 + (size_t)f * 2304;
      for (unsigned gr = 0; gr < ngr; gr++)
        for (unsigned i = 0; i < 576; i++)
          for (unsigned ch = 0; ch < w->nch; ch++) of[(gr * 576 + i) * w->nch + ch] = fs[gr][ch][i];
    }
    int16_t* o = pcm + (size_t)f * 2304;
    if (pcm)
    for (unsigned gr = 0; gr < ngr; gr++)
      for (unsigned i = 0; i < 576; i++) {
        uint32_t v = out[gr][i];
        if (w->nch == 1) o[gr * 576 + i] = (int16_t)(v & 0xffff);
        else {
          o[2 * (gr * 576 + i)] = (int16_t)((v & 0xffff0000u) >> 16);
          o[2 * (gr * 576 + i) + 1] = (int16_t)(v & 0xffff);
        }
      }
  }
  free(w);
  return 0;
}

/* ------------------------------------------------------------------ */
/* SURVEY 8d synthetic generator (C2 / C5): integer-only splitmix64    */
/* ------------------------------------------------------------------ */
static inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

/* key layout: frame*4096 + gr*2048 + ch*1024 + slot; slots 0..575 are lines,
 * 576.. are side fields. */
static inline uint64_t gen_r(uint64_t seed, int64_t frame, unsigned gr, unsigned ch, unsigned slot) {
  return splitmix64(seed ^ ((uint64_t)frame * 4096u + gr * 2048u + ch * 1024u + slot));
}

void orc_generate_frames(uint64_t seed, int64_t first_frame, int n_frames,
                         int16_t* spectra, pdmp3_gc_side* side) {
  for (int f = 0; f < n_frames; f++) {
    int64_t frame = first_frame + f;
    pdmp3_gc_side* sd = side + (size_t)f * 4;
    memset(sd, 0, 4 * sizeof *sd);
    for (unsigned gr = 0; gr < 2; gr++)
      for (unsigned ch = 0; ch < 2; ch++) {
        pdmp3_gc_side* s = &sd[gr * 2 + ch];
        int16_t* sp = spectra + ((size_t)f * 4 + gr * 2 + ch) * 576;
        uint64_t r;
        r = gen_r(seed, frame, gr, ch, 576);
        unsigned count1 = 2 * (240 + (unsigned)(r % 49));
        s->count1 = (uint16_t)count1;
        r = gen_r(seed, frame, gr, ch, 577);
        s->global_gain = (uint8_t)(130 + r % 30);
        r = gen_r(seed, frame, gr, ch, 578);
        unsigned flags = 0;
        if (r & 1) flags |= PDMP3_GC_SCALEFAC_SCALE;
        if (r & 2) flags |= PDMP3_GC_PREFLAG;
        unsigned pct = (unsigned)((r >> 8) % 100), bt = 0, mixed = 0;
        if (pct < 85) bt = 0;
        else if (pct < 90) bt = 1;
        else if (pct < 95) { bt = 2; mixed = (unsigned)((r >> 20) & 1); }
        else bt = 3;
        if (bt != 0) flags |= PDMP3_GC_WIN_SWITCH;
        flags |= bt << PDMP3_GC_BLOCK_TYPE_SHIFT;
        if (mixed) flags |= PDMP3_GC_MIXED;
        s->flags = (uint8_t)flags;
        r = gen_r(seed, frame, gr, ch, 579);
        for (unsigned k = 0; k < 3; k++) s->subblock_gain[k] = (uint8_t)((r >> (8 * k)) % 4);
        /* 44.1 kHz, joint stereo, MS on, intensity off */
        s->frame = (uint8_t)(0u | (1u << PDMP3_FR_MODE_SHIFT) | (2u << PDMP3_FR_MODEEXT_SHIFT));
        for (unsigned k = 0; k < 21; k++) s->scalefac_l[k] = (uint8_t)(gen_r(seed, frame, gr, ch, 600 + k) % 8);
        for (unsigned k = 0; k < 12; k++)
          for (unsigned wn = 0; wn < 3; wn++)
            s->scalefac_s[k][wn] = (uint8_t)(gen_r(seed, frame, gr, ch, 640 + k * 3 + wn) % 8);
        for (unsigned line = 0; line < 576; line++) {
          int v = 0;
          if (line < count1) {
            uint64_t q = gen_r(seed, frame, gr, ch, line);
            unsigned A = 1 + 40 * (576 - line) / 576;
            unsigned r2 = (unsigned)(q >> 16);
            unsigned mag = ((q & 0xff) == 0) ? (r2 % 8207u) : (r2 % (A + 1));
            v = (q & 0x100) ? -(int)mag : (int)mag;
          }
          sp[line] = (int16_t)v;
        }
      }
    /* out-of-bounds scalefactor reads, resolved by the memory-layout rule of
     * (g,0)->[g][1][0]; (0,1)->[1][0][0]; (1,1)[21]->scalefac_s[0][0][0][0];
     * (1,1) short [12][w] -> PEEK. */
    for (unsigned g = 0; g < 2; g++) {
      sd[g * 2 + 0].scalefac_l[21] = sd[g * 2 + 1].scalefac_l[0];
      for (unsigned wn = 0; wn < 3; wn++) sd[g * 2 + 0].scalefac_s[12][wn] = sd[g * 2 + 1].scalefac_s[0][wn];
    }
    sd[0 * 2 + 1].scalefac_l[21] = sd[1 * 2 + 0].scalefac_l[0];
    for (unsigned wn = 0; wn < 3; wn++) sd[0 * 2 + 1].scalefac_s[12][wn] = sd[1 * 2 + 0].scalefac_s[0][wn];
    sd[1 * 2 + 1].scalefac_l[21] = sd[0].scalefac_s[0][0];
    for (unsigned wn = 0; wn < 3; wn++) sd[1 * 2 + 1].scalefac_s[12][wn] = PDMP3_SF_PEEK;
    if (frame == 0)
      for (unsigned k = 0; k < 4; k++) sd[k].frame |= PDMP3_FR_RESET;   /* frame byte is per frame */
  }
}

/* Seconds for `reps` passes over the frames (bench.py cpu_baseline, kind "port"). */
#include <time.h>
double orc_time_decode(const int16_t* spectra, const pdmp3_gc_side* side, int n_frames, int reps) {
  orc_synth* st = (orc_synth*)calloc(1, sizeof *st);
  int16_t* pcm = (int16_t*)malloc((size_t)n_frames * 4608);
  struct timespec a, b;
  build_tables();
  clock_gettime(CLOCK_MONOTONIC, &a);
  for (int r = 0; r < reps; r++) orc_decode_frames(st, spectra, side, n_frames, pcm, NULL);
  clock_gettime(CLOCK_MONOTONIC, &b);
  free(pcm);
  free(st);
  return (b.tv_sec - a.tv_sec) + 1e-9 * (b.tv_nsec - a.tv_nsec);
}

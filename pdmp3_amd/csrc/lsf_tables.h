// lsf_tables.h -- constants of the low-sampling-frequency extension (ISO/IEC 13818-3 "MPEG-2 LSF", and the "MPEG-2.5"
// extension of it for 8 / 11.025 / 12 kHz): SURVEY 8f #4, last third.
//
// NOT from the reference: technosaurus/PDMP3 rejects these streams (pdmp3.c:1293, `id != 1`) and carries MPEG-1 tables
// only (pdmp3.c:517-533 bit rates and sampling frequencies, 879-892 scalefactor bands).  The numbers below are the
// standard's -- 13818-3 table B.8 (band boundaries), 2.4.2.7 / table B.1 (scalefac_compress -> slen, nr_of_sfb_block),
// table B.2-ish bit rates -- written down here by hand and PINNED by an independent decoder: FFmpeg's mpegaudiodec decodes
// packer streams of every one of the six sampling frequencies, and this engine's PCM is within 2 LSB of it
// (tests/golden/lsf_*.npz, tests/test_lsf_pin.py).  A wrong boundary or partition count moves lines between bands and shows up
// there as hundreds of LSB.
//
// Index convention ("sfreq9"): 0..2 = MPEG-1 44.1 / 48 / 32 kHz (tables_data.h), 3..5 = MPEG-2 22.05 / 24 / 16 kHz,
// 6..8 = MPEG-2.5 11.025 / 12 / 8 kHz: sfreq9 = 3 * version + the header's sampling_frequency field, version 0 / 1 / 2.
// Plain C, shared by the host stage, the packer and (through host_tables.h) the kernels' constant bank.
#pragma once
#include <stdint.h>

static const uint32_t kLsfSampleRates[9] = {44100, 48000, 32000, 22050, 24000, 16000, 11025, 12000, 8000};
// Layer III bit rates of the LSF header field, kbit/s x 1000 (13818-3 2.4.2.3); index 0 = free format, 15 = forbidden
static const uint32_t kLsfBitrates[15] = {0, 8000, 16000, 24000, 32000, 40000, 48000, 56000, 64000, 80000, 96000, 112000, 128000, 144000, 160000};

// long-block band boundaries l[0..22] (13818-3 table B.8); rows: 22.05, 24, 16, 11.025, 12, 8 kHz
static const uint16_t kLsfSfbLong[6][23] = {
  {0, 6, 12, 18, 24, 30, 36, 44, 54, 66, 80, 96, 116, 140, 168, 200, 238, 284, 336, 396, 464, 522, 576},
  {0, 6, 12, 18, 24, 30, 36, 44, 54, 66, 80, 96, 114, 136, 162, 194, 232, 278, 332, 394, 464, 540, 576},
  {0, 6, 12, 18, 24, 30, 36, 44, 54, 66, 80, 96, 116, 140, 168, 200, 238, 284, 336, 396, 464, 522, 576},
  {0, 6, 12, 18, 24, 30, 36, 44, 54, 66, 80, 96, 116, 140, 168, 200, 238, 284, 336, 396, 464, 522, 576},
  {0, 6, 12, 18, 24, 30, 36, 44, 54, 66, 80, 96, 116, 140, 168, 200, 238, 284, 336, 396, 464, 522, 576},
  {0, 12, 24, 36, 48, 60, 72, 88, 108, 132, 160, 192, 232, 280, 336, 400, 476, 566, 568, 570, 572, 574, 576},
};
// short-block band boundaries s[0..13] (one window; x 3 for line numbers)
static const uint16_t kLsfSfbShort[6][14] = {
  {0, 4, 8, 12, 18, 24, 32, 42, 56, 74, 100, 132, 174, 192},
  {0, 4, 8, 12, 18, 26, 36, 48, 62, 80, 104, 136, 180, 192},
  {0, 4, 8, 12, 18, 26, 36, 48, 62, 80, 104, 134, 174, 192},
  {0, 4, 8, 12, 18, 26, 36, 48, 62, 80, 104, 134, 174, 192},
  {0, 4, 8, 12, 18, 26, 36, 48, 62, 80, 104, 134, 174, 192},
  {0, 8, 16, 24, 36, 52, 72, 96, 124, 160, 162, 164, 166, 192},
};

// nr_of_sfb_block[class][block shape][partition] (13818-3 2.4.3.2): class 0..2 = scalefac_compress < 400 / < 500 / >= 500
// (class 2 sets preflag), class 3..5 = the RIGHT channel of an intensity-stereo frame, (scalefac_compress >> 1) < 180 /
// < 244 / >= 244; block shape 0 = long (incl. start / stop), 1 = short, 2 = mixed.  Short and mixed counts are in
// scalefactors (three per band: the windows), in the order they are transmitted: band by band, window by window.
static const uint8_t kLsfNsfb[6][3][4] = {
  {{6, 5, 5, 5}, {9, 9, 9, 9}, {6, 9, 9, 9}},
  {{6, 5, 7, 3}, {9, 9, 12, 6}, {6, 9, 12, 6}},
  {{11, 10, 0, 0}, {18, 18, 0, 0}, {15, 18, 0, 0}},
  {{7, 7, 7, 0}, {12, 12, 12, 0}, {6, 15, 12, 0}},
  {{6, 6, 6, 3}, {12, 9, 9, 6}, {6, 12, 9, 6}},
  {{8, 8, 5, 0}, {15, 12, 9, 0}, {6, 18, 9, 0}},
};

// scalefac_compress (9 bits) -> class and the four slen (13818-3 2.4.3.2).  is_right = channel 1 of a frame with
// mode_extension bit 0 (intensity stereo) set.  Returns the class; *preflag as the standard derives it.
static inline int lsf_slen_of(unsigned sfc, int is_right, uint8_t slen[4], int* preflag) {
  *preflag = 0;
  if (is_right) {
    const unsigned h = sfc >> 1;
    if (h < 180) { slen[0] = (uint8_t)(h / 36); slen[1] = (uint8_t)((h % 36) / 6); slen[2] = (uint8_t)(h % 6); slen[3] = 0; return 3; }
    if (h < 244) { const unsigned k = h - 180; slen[0] = (uint8_t)((k % 64) >> 4); slen[1] = (uint8_t)((k % 16) >> 2); slen[2] = (uint8_t)(k % 4); slen[3] = 0; return 4; }
    { const unsigned k = h - 244; slen[0] = (uint8_t)(k / 3); slen[1] = (uint8_t)(k % 3); slen[2] = 0; slen[3] = 0; return 5; }
  }
  if (sfc < 400) { slen[0] = (uint8_t)((sfc >> 4) / 5); slen[1] = (uint8_t)((sfc >> 4) % 5); slen[2] = (uint8_t)((sfc % 16) >> 2); slen[3] = (uint8_t)(sfc % 4); return 0; }
  if (sfc < 500) { const unsigned k = sfc - 400; slen[0] = (uint8_t)((k >> 2) / 5); slen[1] = (uint8_t)((k >> 2) % 5); slen[2] = (uint8_t)(k % 4); slen[3] = 0; return 1; }
  { const unsigned k = sfc - 500; slen[0] = (uint8_t)(k / 3); slen[1] = (uint8_t)(k % 3); slen[2] = 0; slen[3] = 0; *preflag = 1; return 2; }
}

// frame bytes of a Layer III frame: 144 br / sf for MPEG-1 (pdmp3.c:1135-1138), 72 br / sf for LSF (576 samples a frame)
static inline unsigned lsf_frame_bytes(unsigned version, unsigned bitrate, unsigned sfreq9, unsigned padding) {
  return (version ? 72u : 144u) * bitrate / kLsfSampleRates[sfreq9] + padding;
}

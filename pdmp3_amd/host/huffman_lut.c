/* huffman_lut.c -- libpdmp3.so: the Huffman code books (csrc/tables_data.h) as two-level lookup tables, built once per
 * process; the table of MPEG-1 frame sizes.  See host_internal.h for the map of the library. */
#include "host_internal.h"

/* ------------------------------------------------------------------------ */
/* Huffman code books -> two-level lookup tables                             */
/* ------------------------------------------------------------------------ */

huff_lut g_lut[PDMP3_NUM_HUFF_BOOKS];
pthread_once_t g_lut_once = PTHREAD_ONCE_INIT;

uint16_t g_frame_q[15][3];       /* frame sizes without padding (frame_bytes); filled here: handles are created on any thread */
void build_luts(void) {
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


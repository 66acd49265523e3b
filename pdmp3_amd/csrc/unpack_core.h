// unpack_core.h -- main-data decoding on the device: scalefactors
// + Huffman of one granule-channel per LANE, then the frame-to-frame merge of
// the scalefactor / count1 state that the reference never clears.
//
// Replaces, for whole windows of frames at once, what the host stage does per
// frame (host/frame_parse.c decode_main / apply_main / emit_records, themselves
// restatements of pdmp3.c:1376-1437 Read_Main_L3, P:2051-2115 Read_Huffman,
// P:1593-1643 Huffman_Decode).  The input is what is left after the strictly
// sequential part of the bitstream (ring, header sync, side info, bit
// reservoir): per frame a pdmp3_frame_bits and a snapshot of the reservoir
// buffer.  The output is exactly the gc records of include/pdmp3_hip.h, which
// the transform kernel then consumes in place -- decoded spectra never cross
// PCIe.
//
//   unpack_gc    lane = (frame, gr, ch).  The start bit of a granule-channel
//                inside the reservoir follows from the side info alone
//                (part2_3_length of the ones before it, P:2110; the widths of
//                its scalefactors), so the four of a frame decode independently
//                and a granule-channel splits into independent jobs:
//                unpack_records (side fields, scalefactors), unpack_plan +
//                unpack_step (where each code word starts: the one sequential
//                chain), unpack_value (values and signs of a symbol, lines
//                [0, count1) as int16 into a zeroed buffer), unpack_tail (the
//                reference's overshoot rule, count1).  Two-level code-book
//                lookup (8-bit first level, per-prefix second level or leaf).
//                k_unpack (engine.hip) gives the jobs to four waves that share
//                books and reservoir rows in LDS; unpack_gc runs them in a row.
//   merge_slot   thread = one of the 232 values that survive frames
//                (scalefac_l[2][2][21], scalefac_s[2][2][12][3], count1[2][2];
//                SURVEY H4-H6): walks the window's frames in order, keeps
//                "the last value written", and stores it into the records --
//                including the one-past-the-end reads of the reference
//                (scalefac_l[21], scalefac_s[12][*]), which land on the first
//                element of the next block.
//
// Same file for hipcc (kernels in engine.hip) and g++ (tests/host_emul: CPU
// test of the logic against the host stage; not a product path).
#pragma once

#include "decode_core.h"

// A row's 32-bit word as the big-endian number it is in the stream.  The kernel keeps its rows byte-swapped in LDS
// (engine.hip k_unpack swaps while it copies them in): one v_perm per word and symbol less in the loops.
#if defined(__HIP_DEVICE_COMPILE__)
#define PD_ROW_BE(w) (w)
#else
#define PD_ROW_BE(w) __builtin_bswap32(w)
#endif

namespace pdmp3 {

constexpr int kHuffFirstBits = 8;
constexpr int kHuffLutMax = 9024;                 // entries: 19 first levels, the second levels and a leaf per short code word (host_tables.h checks)
constexpr unsigned kResBytes = PDMP3_RESERVOIR_BYTES;
constexpr unsigned kFastLimit = (kResBytes - 8) * 8u;   // bit positions from which an 8-byte load stays inside the row

// One contiguous blob; every workgroup copies it into LDS.
// lut entries.  EVERY symbol is two lookups -- 46 % of a 320 kbps stream's code words are longer than the 8 bits of the
// first level, so with 64 lanes per wave the second one is needed in practically every trip anyway, and making it
// unconditional takes the "is it a link" test and the select out of the loop's dependent chain:
//   first level (256 per book, index = the next 8 bits):   (31 - sub_bits) | byte offset of a second-level table << 8
//              sub_bits = 0: a code word of <= 8 bits; its "table" is the one leaf
//   leaf  = byte 0: (x << 4 | y) or v w x y
//           byte 1: clen, the bits of the whole code word (both levels)
//           byte 2: clen + the number of values != 0 (a sign bit follows each)
//           byte 3: the number of values == 15 of a pair (linbits follow each, if the table has any); <= 2
//   lut[0] = 0: the leaf of every prefix no code word begins with (the reference's error path: nothing is consumed,
//           the values are 0)
// book_base[kZeroBook] is a first level whose 256 entries all lead to lut[0]: what a table without code words (0, 4,
// 14) "decodes" to.
constexpr int kZeroBook = 19;
PD_HD unsigned leaf_clen(uint32_t e) { return (e >> 8) & 0xff; }
PD_HD unsigned leaf_adv(uint32_t e) { return (e >> 16) & 0xff; }
PD_HD unsigned leaf_nlin(uint32_t e) { return e >> 24; }
// index of the leaf that the first-level entry l1 and the bits behind the first 8 (t31 = those bits from bit 30 down,
// bit 31 clear) lead to
PD_HD unsigned leaf_index(uint32_t l1, uint32_t t31) { return (l1 >> 10) + (t31 >> (l1 & 31)); }
PD_HD uint32_t leaf_at(const uint32_t* lut, uint32_t l1, uint32_t t31) {
#if defined(__HIP_DEVICE_COMPILE__)
  // (bits 5-9 of l1 are clear: l1 >> 8 is the table's BYTE offset, and the address is one shift-and-add away)
  return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(lut) + (l1 >> 8) + ((t31 >> (l1 & 31)) << 2));
#else
  return lut[leaf_index(l1, t31)];
#endif
}
struct UnpackTables {
  uint16_t book_base[20];        // first-level table of book b starts at lut[book_base[b]]
  int8_t book_of_table[36];      // ISO table number (0..33) -> book, -1: no code words (tables 0, 4, 14)
  uint8_t linbits[36];
  uint8_t slen[32];              // [scalefac_compress][2], P:1378-1381
  uint16_t sfb_l[3][24];         // g_sf_band_indices[].l, P:879-892; [23] = s[0] (the arrays are contiguous, H7)
  uint16_t sfb_s[3][16];
  uint32_t n_lut, pad[3];
  uint32_t lut[kHuffLutMax];
};

// what one granule-channel's main data yields (cf. main_out in host/frame_parse.c)
struct alignas(16) GcRaw {
  uint16_t count1;
  uint8_t count1_set;            // 0 when part2_3_length == 0: count1 keeps its old value (H6)
  uint8_t sf_l_copy;             // granule 1: bit b = band group b is taken from granule 0 (scfsi)
  uint32_t sf_l_set;             // bit sfb: scalefac_l[sfb] was read from the stream
  uint16_t sf_s_set;             // bit sfb: scalefac_s[sfb][0..2] were read
  uint16_t pad0;
  uint8_t sf_l[24];
  uint8_t sf_s[36];
  uint8_t pad1[8];
};
static_assert(sizeof(GcRaw) == 80, "GcRaw layout");

constexpr int kMergeSlots = 84 + 144 + 4;          // scalefac_l, scalefac_s, count1

// ---------------------------------------------------------------------------
// reservoir rows from the pool (include/pdmp3_hip.h: pdmp3_row_desc).  d points at the frame's descriptor inside the
// window's array: d[-k] is the k-th frame before it.
// ---------------------------------------------------------------------------
PD_HD uint8_t row_byte(const pdmp3_row_desc* d, const uint8_t* pool, unsigned j) {
  while (d->top <= j) {                           // up the skyline of the segment: every hop has a larger top
    if (!d->up) return pool[d->s_off + j];
    d -= d->up;
  }
  return pool[d->row_off + j];
}
PD_HD uint32_t row_word(const pdmp3_row_desc* d, const uint8_t* pool, unsigned j) {   // bytes j .. j + 3, little-endian
  while (d->top <= j) {
    if (!d->up) { uint32_t v; __builtin_memcpy(&v, pool + d->s_off + j, 4); return v; }   // (the image is 2064 bytes: j + 3 is inside)
    d -= d->up;
  }
  if (d->top >= j + 4) { uint32_t v; __builtin_memcpy(&v, pool + d->row_off + j, 4); return v; }
  uint32_t v = 0;                                 // the word straddles a boundary: byte by byte
  for (int q = 0; q < 4; q++) v |= (uint32_t)row_byte(d, pool, j + q) << (8 * q);
  return v;
}

// bytes j .. j + 15 (j a multiple of 16) as four little-endian words: one 16-byte read where they come from one place --
// nearly always: a row changes its source at the tops of the frames on its skyline, a handful of places in 2064 bytes
PD_HD void row_chunk16(const pdmp3_row_desc* d, const uint8_t* pool, unsigned j, uint32_t out[4]) {
  const pdmp3_row_desc* e = d;
  const uint8_t* src = nullptr;
  for (;;) {
    if (e->top > j) { if (e->top >= j + 16) src = pool + e->row_off + j; break; }
    if (!e->up) { src = pool + e->s_off + j; break; }   // (the image is 2064 = 129 x 16 bytes: j + 15 is inside)
    e -= e->up;
  }
  if (src) { __builtin_memcpy(out, src, 16); return; }
  for (int q = 0; q < 4; q++) out[q] = row_word(d, pool, j + 4u * (unsigned)q);
}

// ---------------------------------------------------------------------------
// bit reader over one reservoir row (same windows as host/frame_parse.c peek32 / peek64)
// ---------------------------------------------------------------------------
struct BitPos {
  const uint8_t* buf;
  unsigned pos;
};

PD_HD uint32_t peek32(const BitPos& b) {            // next 25+ valid bits, MSB first; clamped at the row's end
  unsigned byte = b.pos >> 3;
  if (byte > kResBytes - 5) byte = kResBytes - 5;
  const uint32_t* q = reinterpret_cast<const uint32_t*>(b.buf) + (byte >> 2);              // the 8 bytes around: 5 from byte & 3 on
  const uint64_t two = ((uint64_t)PD_ROW_BE(q[0]) << 32) | PD_ROW_BE(q[1]);
  const uint64_t w = (two << (8 * (byte & 3))) >> 24;                                      // bytes byte .. byte + 4
  return (uint32_t)(w >> (8 - (b.pos & 7)));
}
PD_HD uint64_t peek64(const BitPos& b);
PD_HD unsigned get_bits(BitPos& b, unsigned n) {    // n <= 16
  if (!n) return 0;
  const unsigned v = b.pos <= kFastLimit ? (unsigned)(peek64(b) >> (64 - n)) : peek32(b) >> (32 - n);
  b.pos += n;
  return v;
}
// 64 valid bits from bit `pos` on, out of three ALIGNED 32-bit words (the row sits in LDS in the kernel: aligned
// ds_read_b32s, no dependence on global-memory latency -- with the row in HBM every symbol waited ~500 cycles for
// its load, behind the acknowledgement of the scattered stores before it: 365 us per 2048-frame window).
// Caller guarantees pos <= kFastLimit; the third word may lie up to 4 bytes past the row, none of its bits that
// reach the result do.
PD_HD uint64_t peek64(const BitPos& b) {
  const uint32_t* p = reinterpret_cast<const uint32_t*>(b.buf) + (b.pos >> 5);
  const uint32_t d0 = PD_ROW_BE(p[0]), d1 = PD_ROW_BE(p[1]), d2 = PD_ROW_BE(p[2]);
  const unsigned s = b.pos & 31;
  const uint64_t two = ((uint64_t)d0 << 32) | d1;
  return s ? (two << s) | (uint64_t)(d2 >> (32 - s)) : two;
}

// The same bits out of REGISTERS: d0, d1 are the row's big-endian words pos >> 5 and the next one, nx the word after
// them, already on its way from LDS.  A code word (<= 19 bits) or the linbits and signs that follow it (<= 28 bits) are
// one funnel shift of d0:d1, and rw_step() after each of the two moves the window by at most one word, taking the
// word asked for a step earlier -- its LDS latency passes under the table lookup instead of in front of every symbol
// (peek64: three loads and ~a hundred cycles of waiting per symbol, in a loop that is one dependent chain per lane).
// Valid while the iteration began at pos <= kFastLimit: the last word touched is then 8 bytes past the row at most.
struct RegWin {
  const uint32_t* row;
  uint32_t d0, d1, nx;
  unsigned wi;
};
PD_HD void rw_open(RegWin& r, const uint8_t* buf, unsigned pos) {
  r.row = reinterpret_cast<const uint32_t*>(buf);
  r.wi = pos >> 5;
  if (r.wi > kFastLimit / 32) r.wi = kFastLimit / 32;      // (a position out there is never read through the window)
  r.d0 = PD_ROW_BE(r.row[r.wi]);
  r.d1 = PD_ROW_BE(r.row[r.wi + 1]);
  r.nx = 0;
}
PD_HD uint32_t rw_peek(const RegWin& r, unsigned pos) {     // 32 bits from bit `pos` (pos >> 5 == r.wi)
  const unsigned s = pos & 31;
  return (uint32_t)(((((uint64_t)r.d0) << 32) | r.d1) >> (32 - s));   // one 64-bit shift (32 - s is 1..32: no branch for s = 0)
}
// ask for the word after the window; rw_step() takes it.  Kept apart so that no load is in flight across a loop's
// back edge (the compiler then waits for it at the loop head) and the lookup's wait covers this one too.
PD_HD void rw_ask(RegWin& r) { r.nx = r.row[r.wi + 2]; }
PD_HD void rw_step(RegWin& r, unsigned pos) {               // pos has moved by < 32 bits since rw_ask()
  const bool adv = (pos >> 5) != r.wi;
  const uint32_t sw = PD_ROW_BE(r.nx);
  r.d0 = adv ? r.d1 : r.d0;
  r.d1 = adv ? sw : r.d1;
  r.wi = pos >> 5;
}

// Short fields in a row (scalefactors).  Same per-call decision as get_bits: past kFastLimit the byte-wise clamped window.
PD_HD unsigned get_field(BitPos& b, RegWin& r, unsigned n) {   // n <= 16
  if (!n) return 0;
  unsigned v;
  if (b.pos > kFastLimit) v = peek32(b) >> (32 - n);
  else { rw_ask(r); v = rw_peek(r, b.pos) >> (32 - n); }
  b.pos += n;
  if (b.pos <= kFastLimit + 32) rw_step(r, b.pos);
  return v;
}

// one code word from a 64-bit window: returns leaf value, adds its length to `used`
PD_HD unsigned lut_symbol(const uint32_t* lut, unsigned base, uint64_t w, unsigned& used) {
  const uint32_t l1 = lut[base + (unsigned)(w >> (64 - kHuffFirstBits))];
  const uint32_t e = leaf_at(lut, l1, (uint32_t)((w << kHuffFirstBits) >> 33));
  used += leaf_clen(e);
  return e & 0xff;
}
PD_HD unsigned lut_symbol_slow(const uint32_t* lut, unsigned base, BitPos& b) {
  const uint64_t w = (uint64_t)peek32(b) << 32;
  unsigned used = 0;
  const unsigned leaf = lut_symbol(lut, base, w, used);     // codes are <= 19 bits: inside peek32's 25
  b.pos += used;
  return leaf;
}

// bits of part 2 (scalefactors) of one granule-channel, from the side info alone (P:1376-1437)
PD_HD unsigned part2_bits(const UnpackTables& U, const pdmp3_frame_bits& F, int gr, int ch) {
  const pdmp3_gc_bits& s = F.gc[gr * 2 + ch];
  const unsigned slen1 = U.slen[s.scalefac_compress * 2], slen2 = U.slen[s.scalefac_compress * 2 + 1];
  const bool shrt = (s.flags & PDMP3_GC_WIN_SWITCH) && ((s.flags & PDMP3_GC_BLOCK_TYPE_MASK) >> PDMP3_GC_BLOCK_TYPE_SHIFT) == 2;
  if (shrt) return (s.flags & PDMP3_GC_MIXED) ? 8 * slen1 + 9 * slen1 + 18 * slen2 : 18 * slen1 + 18 * slen2;
  unsigned n = 0;
  for (int g4 = 0; g4 < 4; g4++)
    if (!(gr == 1 && (F.scfsi[ch] >> g4 & 1))) n += (g4 == 0 ? 6u : 5u) * (g4 < 2 ? slen1 : slen2);
  return n;
}

#if defined(__HIPCC__)
#define PD_COLD __device__ __noinline__
#define PD_MUL24(a, b) __umul24((a), (b))        /* full-rate multiply of small values */
#else
#define PD_COLD static __attribute__((noinline))
#define PD_MUL24(a, b) ((a) * (b))
#endif

// store the pair at (even) line pos; big_values > 288 is not checked by the reference (H8): lines >= 576 are dropped
PD_HD void store_pair(int16_t* is, unsigned pos, int x, int y) {
  if (pos + 1 < 576) {
    const uint32_t v = (uint32_t)(uint16_t)(int16_t)x | ((uint32_t)(uint16_t)(int16_t)y << 16);
    __builtin_memcpy(is + pos, &v, 4);
  } else if (pos < 576) is[pos] = (int16_t)x;
}

// The rest of the pairs once the bit position has left the fast region (corrupt streams only): byte-wise windows,
// clamped at the row's end.  Out of line: the hot loop below stays small.
PD_COLD unsigned unpack_pairs_slow(const uint32_t* lut, BitPos& b, int base0, int base1, int base2, unsigned lin0,
                                   unsigned lin1, unsigned lin2, unsigned e0, unsigned e1, unsigned nbig, unsigned pos,
                                   int16_t* is) {
  for (; pos < nbig; pos += 2) {
    const int base = pos < e0 ? base0 : pos < e1 ? base1 : base2;
    const unsigned linbits = pos < e0 ? lin0 : pos < e1 ? lin1 : lin2;
    if (base < 0) continue;
    int x, y;
    if (b.pos <= kFastLimit) {                     // (a region without code words may have brought us back here)
      const uint64_t w = peek64(b);
      unsigned used = 0;
      const unsigned leaf = lut_symbol(lut, (unsigned)base, w, used);
      x = leaf >> 4; y = leaf & 15;
      if (linbits && x == 15) { x += (int)((w << used) >> (64 - linbits)); used += linbits; }
      if (x) { if ((w << used) >> 63) x = -x; used++; }
      if (linbits && y == 15) { y += (int)((w << used) >> (64 - linbits)); used += linbits; }
      if (y) { if ((w << used) >> 63) y = -y; used++; }
      b.pos += used;
    } else {
      const unsigned leaf = lut_symbol_slow(lut, (unsigned)base, b);
      x = leaf >> 4; y = leaf & 15;
      if (linbits && x == 15) x += (int)get_bits(b, linbits);
      if (x > 0 && get_bits(b, 1)) x = -x;
      if (linbits && y == 15) y += (int)get_bits(b, linbits);
      if (y > 0 && get_bits(b, 1)) y = -y;
    }
    store_pair(is, pos, x, y);
  }
  return pos;
}

// the rest of the count1 quads once the bit position has left the fast region (corrupt streams only)
PD_COLD unsigned unpack_quads_slow(const uint32_t* lut, BitPos& b, unsigned qbase, unsigned end, unsigned pos, int16_t* is) {
  while (pos <= 572 && b.pos <= end) {
    int q[4];
    if (b.pos <= kFastLimit) {
      const uint64_t w = peek64(b);
      unsigned used = 0;
      const unsigned leaf = lut_symbol(lut, qbase, w, used);
      for (int k = 0; k < 4; k++) {
        q[k] = (int)(leaf >> (3 - k)) & 1;
        if (q[k]) { if ((w << used) >> 63) q[k] = -1; used++; }
      }
      b.pos += used;
    } else {
      const unsigned leaf = lut_symbol_slow(lut, qbase, b);
      for (int k = 0; k < 4; k++) {
        q[k] = (int)(leaf >> (3 - k)) & 1;
        if (q[k] && get_bits(b, 1)) q[k] = -1;
      }
    }
    store_pair(is, pos, q[0], q[1]);
    store_pair(is, pos + 2, q[2], q[3]);
    pos += 4;
  }
  return pos;
}

// ---------------------------------------------------------------------------
// The symbols of one granule-channel (Read_Huffman P:2051-2115; cf. decode_pairs in host/frame_parse.c), in two stages.
//
// What makes the loop sequential is only WHERE the next code word starts.  unpack_step() does just that much per
// symbol -- the code book lookup and the bits the symbol takes in all: code word, linbits, sign bits, the last three
// out of the leaf entry without looking at them -- and hands a record to unpack_value(), which reads the values
// and signs at leisure and stores the lines: on the device in OTHER waves (engine.hip k_unpack: one wave walks 64
// bit streams, three take the records out of an LDS ring), in the host build right away.
//
// ONE loop for the big_values pairs of the three regions and the count1 quads: the lanes of a wave sit in different
// regions with different tables, and a loop per kind costs the longest lane of each kind (406 trips per wave of 64
// for a 320 kbps stream) instead of the longest lane (<= 288: 2 pairs + 4 quads <= 576 lines).  A region whose
// table has no code words (0, 4, 14) reads no bits and leaves its (pre-zeroed) lines alone.
// ---------------------------------------------------------------------------
// the 64 bits around the position in registers: d0 holds bit 32 (pos >> 5).  A symbol takes up to 19 + 2 + 2 x 13 = 47
// bits, so the window is simply read again where the symbol ends (one aligned 8-byte LDS read; sliding a register
// window by 0, 1 or 2 words took five times the instructions, and a lone wave pays for instructions, not for latency)
struct Win2 {
  const uint32_t* row;
  uint32_t d0, d1;
};
PD_HD void w2_load(Win2& r, unsigned pos) {                // (pos <= kFastLimit + 47: the second word is <= 4 bytes past the row)
  const uint32_t* p = r.row + (pos >> 5);
  r.d0 = PD_ROW_BE(p[0]);
  r.d1 = PD_ROW_BE(p[1]);
}
PD_HD void w2_open(Win2& r, const uint8_t* buf, unsigned pos) {
  r.row = reinterpret_cast<const uint32_t*>(buf);
  w2_load(r, pos <= kFastLimit ? pos : kFastLimit);        // (a position out there is never read through the window)
}

// what the loop needs of the side info, per lane
struct SymPlan {
  unsigned tab0, tab1, tab2;     // per region: first-level table of its book (kZeroBook's: no code words) | linbits << 16
  unsigned qbase;                // count1 book
  unsigned e0, e1, nbig;         // line where region 0 / 1 / the pairs end
  unsigned end;                  // last bit of the granule-channel
};
PD_HD int plan_base(unsigned tab, unsigned zero_base) { return (tab & 0xffffu) == zero_base ? -1 : (int)(tab & 0xffffu); }   // as the byte-wise forms want it
struct SymState {
  unsigned pos;                  // bit position
  unsigned line;
};

// record of one symbol:  x = pos | quad << 15 | line << 16 | linbits << 26 | (tag << 30: the ring's business)
//                        y = the leaf entry (0: values are zero) | kRecEnd
constexpr unsigned kRecNopLine = 1023;           // "nothing to store" (lines >= 576 are dropped, H8)
constexpr uint32_t kRecEnd = 0x80000000u;        // (a leaf never has the link bit)
struct SymRec {
  uint32_t x, y;
};

// does the reference's loop go on?  Pairs: to the last pair whatever the position (P:2071-2097); quads: while lines and
// bits are left (P:2099-2105).  Past kFastLimit the window is not valid: the byte-wise forms below take over.
PD_HD bool sym_active(unsigned nbig, unsigned end, const SymState& s) {
  const bool more = (s.line < nbig) | ((s.line <= 572) & (s.pos <= end));     // (no short cuts: plain compares, no branches)
  return more & (s.pos <= kFastLimit);
}

// one symbol: `w` at s.pos (wi == s.pos >> 5), active lane
// (the plan as separate values: handed over as a struct the device compiler keeps it in memory and turns the selects
// below into loads from a selected address, at the head of every trip's dependent chain)
// `act` false: the lane is through (or not there): it looks at the zero book, takes no bits, stays where it is and its
// record says "zeroes" -- no branch around the step, whose bookkeeping would cost more than the step's idle lanes
PD_HD SymRec unpack_step(const uint32_t* lut, unsigned tab0, unsigned tab1, unsigned tab2, unsigned qbase, unsigned ztab,
                         unsigned e0, unsigned e1, unsigned nbig, bool act, SymState& s, Win2& w) {
  const bool pair = s.line < nbig;
  const unsigned tab_p = s.line < e0 ? tab0 : s.line < e1 ? tab1 : tab2;
  const unsigned tab_a = pair ? tab_p : qbase;             // (linbits 0 in the upper half)
  const unsigned tab = act ? tab_a : ztab;
  const unsigned lin = tab >> 16;
  const uint32_t bits = (uint32_t)((((((uint64_t)w.d0) << 32) | w.d1) << (s.pos & 31)) >> 32);
  const unsigned i1 = (tab & 0xffffu) + (bits >> (32 - kHuffFirstBits));
  const uint32_t l1 = lut[i1];
  const uint32_t e = leaf_at(lut, l1, (bits << kHuffFirstBits) >> 1);
  SymRec rec;
  rec.x = s.pos | (pair ? 0u : 1u << 15) | (s.line << 16) | (lin << 26);
  rec.y = e;
  s.pos += leaf_adv(e) + PD_MUL24(leaf_nlin(e), lin);
  s.line += act ? (pair ? 2u : 4u) : 0u;
  w2_load(w, s.pos);
  return rec;
}

// values and signs of a record's symbol, stored to the lines of its granule-channel.  What follows the code word --
// linbits and signs, <= 28 bits -- is one 32-bit window (two aligned words of the row, one funnel shift); everything
// behind that is 32-bit arithmetic (64-bit shifts are quarter rate on the device, and this is the value waves' loop).
PD_HD void unpack_value(const uint8_t* row, const SymRec rec, int16_t* is) {
  const unsigned line = (rec.x >> 16) & 0x3ff;
  const uint32_t e = rec.y;
  if (line >= 576 || e == 0) return;                       // (e == 0: zeroes, and the lines are zero)
  const unsigned linbits = (rec.x >> 26) & 15;
  const unsigned pv = (rec.x & 0x7fffu) + leaf_clen(e);    // <= kFastLimit + 19: the second word is inside the row
  const uint32_t* p = reinterpret_cast<const uint32_t*>(row) + (pv >> 5);
  uint32_t w = (uint32_t)((((((uint64_t)PD_ROW_BE(p[0])) << 32) | PD_ROW_BE(p[1])) << (pv & 31)) >> 32);
  if (rec.x & (1u << 15)) {                                // v w x y: a sign bit follows each nonzero one
    int q[4];
    for (int k = 0; k < 4; k++) {
      const unsigned nz = (e >> (3 - k)) & 1;
      q[k] = (nz && (w >> 31)) ? -1 : (int)nz;
      w <<= nz;
    }
    // (the walker reads a quad only while line <= 572, P:2099: all four lines exist; line is even: one 8-byte store)
    const uint32_t v[2] = {(uint32_t)(uint16_t)(int16_t)q[0] | ((uint32_t)(uint16_t)(int16_t)q[1] << 16),
                           (uint32_t)(uint16_t)(int16_t)q[2] | ((uint32_t)(uint16_t)(int16_t)q[3] << 16)};
    __builtin_memcpy(__builtin_assume_aligned(is + line, 4), v, 8);
  } else {                                                 // (v >> 1) >> (31 - n) == v >> (32 - n) for n = 1..31 and 0 for n = 0
    int x = (int)((e >> 4) & 15), y = (int)(e & 15);
    const unsigned lbx = (x == 15) ? linbits : 0;
    x += (int)((w >> 1) >> (31 - lbx));
    w <<= lbx;
    const unsigned nzx = x != 0;
    x = (nzx && (w >> 31)) ? -x : x;
    w <<= nzx;
    const unsigned lby = (y == 15) ? linbits : 0;
    y += (int)((w >> 1) >> (31 - lby));
    w <<= lby;
    y = (y != 0 && (w >> 31)) ? -y : y;
    // (line is even and < 576: both lines exist)
    const uint32_t v = (uint32_t)(uint16_t)(int16_t)x | ((uint32_t)(uint16_t)(int16_t)y << 16);
    __builtin_memcpy(__builtin_assume_aligned(is + line, 4), &v, 4);
  }
}

// ---------------------------------------------------------------------------
// one granule-channel = unpack_records (side fields and scalefactors into the records)
//                     + unpack_plan    (where the Huffman data starts and what the symbol loop needs: side info only)
//                     + the symbol loop (host: unpack_gc below; device: k_unpack)
//                     + unpack_tail    (byte-wise rest on corrupt streams, the overshoot rule, count1)
// The first two do not depend on each other -- the scalefactors' widths are in the side info, so the position behind
// them is too -- and k_unpack gives them to different waves: the walker starts on the symbols at once.
// `spectra_gc` (576 int16) must be zero on entry.  `rec` and `raw` are fully written.
// ---------------------------------------------------------------------------
// bit position of granule-channel g's part 2: every one before it ends at start + part2_3_length -- or, when that is
// zero, right after its scalefactors (P:2062: Read_Huffman returns before touching the position)
PD_HD unsigned part2_start_of(const UnpackTables& U, const pdmp3_frame_bits& F, int g, int nch) {
  unsigned pos = 0;
  for (int q = 0; q < g; q++) {
    if ((q & 1) >= nch) continue;
    const unsigned p23 = F.gc[q].part2_3_length;
    pos += p23 ? p23 : part2_bits(U, F, q >> 1, q & 1);
  }
  return pos;
}

// The fields of a gc record that are the frame's own side info, into a record that is ZERO: everything but the
// scalefactors and count1, which the merge fills in (the values that survive frames).
PD_HD void side_fields(const pdmp3_frame_bits& F, int g, pdmp3_gc_side* rec) {
  const int ch = g & 1;
  const int nch = ((F.frame & PDMP3_FR_MODE_MASK) >> PDMP3_FR_MODE_SHIFT) == 3 ? 1 : 2;
  rec->frame = F.frame & (uint8_t)~PDMP3_FR_NEWSTREAM;
  // the ISO switches of the frame (include/pdmp3.h PDMP3_ISO_*: MS_BOUND = 2, IS_SHORT = 4) as the records' PDMP3_GC_ISO_* bits
  rec->iso = (uint8_t)(((F.iso & 0x02u) ? PDMP3_GC_ISO_MS_ALL : 0u) | ((F.iso & 0x04u) ? PDMP3_GC_ISO_IS_SHORT : 0u) |
                       ((F.iso & 0x20u) ? PDMP3_GC_ISO_IS_STD : 0u));
  if (ch >= nch) return;
  const pdmp3_gc_bits& s = F.gc[g];
  rec->global_gain = s.global_gain;
  rec->flags = s.flags;
  rec->subblock_gain[0] = s.subblock_gain[0]; rec->subblock_gain[1] = s.subblock_gain[1]; rec->subblock_gain[2] = s.subblock_gain[2];
  if (g == 3 && !(F.iso & 0x10u)) rec->scalefac_s[12][0] = rec->scalefac_s[12][1] = rec->scalefac_s[12][2] = PDMP3_SF_PEEK;   // (PDMP3_ISO_SF12: stays 0)
}

// the scalefactors the stream carries for this granule-channel (P:1383-1430), as the merge's input
PD_HD void unpack_scalefactors(const UnpackTables& U, const uint8_t* res, const pdmp3_frame_bits& F, int g, GcRaw* raw) {
  const int gr = g >> 1, ch = g & 1;
  const int nch = ((F.frame & PDMP3_FR_MODE_MASK) >> PDMP3_FR_MODE_SHIFT) == 3 ? 1 : 2;
  {
    uint32_t* w32 = reinterpret_cast<uint32_t*>(raw);
    for (int i = 0; i < 20; i++) w32[i] = 0;
  }
  if (ch >= nch) return;
  const pdmp3_gc_bits& s = F.gc[g];
  BitPos b{res, part2_start_of(U, F, g, nch)};
  const unsigned slen1 = U.slen[s.scalefac_compress * 2], slen2 = U.slen[s.scalefac_compress * 2 + 1];
  const bool wsf = (s.flags & PDMP3_GC_WIN_SWITCH) != 0;
  const unsigned bt = (s.flags & PDMP3_GC_BLOCK_TYPE_MASK) >> PDMP3_GC_BLOCK_TYPE_SHIFT;
  RegWin r;
  rw_open(r, res, b.pos);
  if (wsf && bt == 2) {
    unsigned first_short = 0;
    if (s.flags & PDMP3_GC_MIXED) {
      for (unsigned sfb = 0; sfb < 8; sfb++) raw->sf_l[sfb] = (uint8_t)get_field(b, r, slen1);
      raw->sf_l_set = 0xffu;
      first_short = 3;
    }
    unsigned set = 0;
    for (unsigned sfb = first_short; sfb < 12; sfb++) {
      for (unsigned w = 0; w < 3; w++) raw->sf_s[sfb * 3 + w] = (uint8_t)get_field(b, r, sfb < 6 ? slen1 : slen2);
      set |= 1u << sfb;
    }
    raw->sf_s_set = (uint16_t)set;
  } else {
    unsigned set = 0, copy = 0;
    for (unsigned g4 = 0; g4 < 4; g4++) {
      const unsigned lo = g4 ? 1 + 5 * g4 : 0, hi = 6 + 5 * g4, nb = g4 < 2 ? slen1 : slen2;
      if (gr == 1 && (F.scfsi[ch] >> g4 & 1)) copy |= 1u << g4;
      else for (unsigned sfb = lo; sfb < hi; sfb++) { raw->sf_l[sfb] = (uint8_t)get_field(b, r, nb); set |= 1u << sfb; }
    }
    raw->sf_l_set = set;
    raw->sf_l_copy = (uint8_t)copy;
  }
}

// both (the sequential form; k_unpack writes the merge's input only and k_merge_apply builds the whole record)
PD_HD void unpack_records(const UnpackTables& U, const uint8_t* res, const pdmp3_frame_bits& F, int g, pdmp3_gc_side* rec,
                          GcRaw* raw) {
  uint32_t* r32 = reinterpret_cast<uint32_t*>(rec);
  for (int i = 0; i < 32; i++) r32[i] = 0;
  side_fields(F, g, rec);
  unpack_scalefactors(U, res, F, g, raw);
}

// false: no Huffman data (channel absent, or part2_3_length == 0: spectra stay zero, count1 keeps its old value, H6)
PD_HD bool unpack_plan(const UnpackTables& U, const pdmp3_frame_bits& F, int g, SymPlan& P, SymState& st) {
  const int gr = g >> 1, ch = g & 1;
  const int nch = ((F.frame & PDMP3_FR_MODE_MASK) >> PDMP3_FR_MODE_SHIFT) == 3 ? 1 : 2;
  const int sfreq = (F.frame & PDMP3_FR_SFREQ_MASK) > 2 ? 2 : (F.frame & PDMP3_FR_SFREQ_MASK);
  if (ch >= nch) return false;
  const pdmp3_gc_bits& s = F.gc[g];
  if (s.part2_3_length == 0) return false;
  const unsigned part2_start = part2_start_of(U, F, g, nch);
  const bool wsf = (s.flags & PDMP3_GC_WIN_SWITCH) != 0;
  const unsigned bt = (s.flags & PDMP3_GC_BLOCK_TYPE_MASK) >> PDMP3_GC_BLOCK_TYPE_SHIFT;
  // ---- Huffman (P:2051-2115)
  P.end = part2_start + s.part2_3_length - 1;
  unsigned r1, r2;
  if (wsf && bt == 2) { r1 = 36; r2 = 576; }
  else {
    const unsigned i1 = s.region0_count + 1u, i2 = s.region0_count + s.region1_count + 2u;
    r1 = i1 < 23 ? U.sfb_l[sfreq][i1] : U.sfb_s[sfreq][i1 - 23];      // H7
    r2 = i2 < 23 ? U.sfb_l[sfreq][i2] : U.sfb_s[sfreq][i2 - 23];
  }
  P.nbig = s.big_values * 2u;
  P.e0 = (r1 + 1) & ~1u; P.e1 = (r2 + 1) & ~1u;
  if (P.e0 > P.nbig) P.e0 = P.nbig;
  if (P.e1 > P.nbig) P.e1 = P.nbig;
  if (P.e1 < P.e0) P.e1 = P.e0;
  {
    const int b0 = U.book_of_table[s.table_select[0]], b1 = U.book_of_table[s.table_select[1]], b2 = U.book_of_table[s.table_select[2]];
    P.tab0 = U.book_base[b0 < 0 ? kZeroBook : b0] | (unsigned)U.linbits[s.table_select[0]] << 16;
    P.tab1 = U.book_base[b1 < 0 ? kZeroBook : b1] | (unsigned)U.linbits[s.table_select[1]] << 16;
    P.tab2 = U.book_base[b2 < 0 ? kZeroBook : b2] | (unsigned)U.linbits[s.table_select[2]] << 16;
  }
  // count1 region: table 32 or the reference's mis-pointed table 33 (H1); both books are <= 8 bits deep
  P.qbase = U.book_base[U.book_of_table[32 + s.count1table_select]];
  st.pos = part2_start + part2_bits(U, F, gr, ch);
  st.line = 0;
  return true;
}

// after the symbol loop, once every record's lines are stored
PD_HD void unpack_tail(const UnpackTables& U, const uint32_t* lut, const uint8_t* res, const SymPlan& P, SymState st, int16_t* spectra_gc,
                       GcRaw* raw) {
  BitPos b{res, st.pos};
  unsigned pos = st.line;
  if (pos < P.nbig) {
    const unsigned z = U.book_base[kZeroBook];
    pos = unpack_pairs_slow(lut, b, plan_base(P.tab0, z), plan_base(P.tab1, z), plan_base(P.tab2, z), P.tab0 >> 16, P.tab1 >> 16,
                            P.tab2 >> 16, P.e0, P.e1, P.nbig, pos, spectra_gc);
  }
  if (pos <= 572 && b.pos <= P.end) pos = unpack_quads_slow(lut, b, P.qbase, P.end, pos, spectra_gc);
  // Overshoot: the reference takes the last four lines back (P:2106-2108) -- the last quad, or, when no quad
  // was read, the last two PAIRS -- and zero-fills from there.  (pos < 4 wraps like the reference's unsigned:
  // count1 becomes 576 and nothing is zeroed.)
  if (b.pos > P.end + 1) {
    pos -= 4;
    for (unsigned i = pos; i < 576 && i < pos + 4; i++) spectra_gc[i] = 0;
  }
  if (pos > 576) pos = 576;
  raw->count1 = (uint16_t)pos;
  raw->count1_set = 1;
}

// the three in a row (host test build; reference form of what k_unpack does with four waves)
PD_HD void unpack_gc(const UnpackTables& U, const uint32_t* lut, const uint8_t* res, const pdmp3_frame_bits& F, int g,
                     int16_t* spectra_gc, pdmp3_gc_side* rec, GcRaw* raw) {
  SymPlan P;
  SymState st;
  unpack_records(U, res, F, g, rec, raw);
  if (!unpack_plan(U, F, g, P, st)) return;
  Win2 w;
  w2_open(w, res, st.pos);
  while (sym_active(P.nbig, P.end, st))
    unpack_value(res, unpack_step(lut, P.tab0, P.tab1, P.tab2, P.qbase, U.book_base[kZeroBook], P.e0, P.e1, P.nbig, true, st, w), spectra_gc);
  unpack_tail(U, lut, res, P, st, spectra_gc, raw);
}

// ---------------------------------------------------------------------------
// frame-to-frame merge: slot t of the 232 surviving values
//   t <  84          scalefac_l[g][sfb]      g = t / 21
//   t < 228          scalefac_s[g][sfb][w]   g = (t - 84) / 36
//   t < 232          count1[g]
// state[kMergeSlots] (uint16) carries the values from one window to the next.
// ---------------------------------------------------------------------------
PD_HD bool gc_active(const pdmp3_frame_bits& F, int g) {
  return (g & 1) == 0 || ((F.frame & PDMP3_FR_MODE_MASK) >> PDMP3_FR_MODE_SHIFT) != 3;
}

// state index of the granule-0 twin that a granule-1 scalefac_l slot may copy from (scfsi), else -1
PD_HD int merge_twin(int t) {
  if (t >= 42 && t < 84) return t - 42;            // (gr 1, ch, sfb) -> (gr 0, ch, sfb)
  return -1;
}

// what frame f does to slot t: `set` = the stream carried a new value `val`; for granule-1 scalefac_l slots also
// the same for the twin (set0 / val0) and `copy` = this frame takes the twin's value instead
struct MergeIn {
  bool set, set0, copy;
  unsigned val, val0;
};

PD_HD MergeIn merge_load(int t, const GcRaw* raw_f /* raw + f * 4 */, bool newstream = false) {
  MergeIn m{false, false, false, 0, 0};
  if (newstream) { m.set = true; m.set0 = true; }   // PDMP3_FR_NEWSTREAM: the old value is gone -- 0 unless this frame writes one
  if (t < 84) {
    const int g = t / 21, sfb = t - 21 * g;
    const GcRaw& r = raw_f[g];
    const bool w = (r.sf_l_set >> sfb & 1) != 0;
    m.val = w ? r.sf_l[sfb] : 0;
    m.set = m.set || w;
    if (g >= 2) {
      const int grp = sfb < 6 ? 0 : sfb < 11 ? 1 : sfb < 16 ? 2 : 3;
      const GcRaw& r0 = raw_f[g & 1];
      const bool w0 = (r0.sf_l_set >> sfb & 1) != 0;
      m.val0 = w0 ? r0.sf_l[sfb] : 0;
      m.set0 = m.set0 || w0;
      m.copy = (r.sf_l_copy >> grp & 1) != 0;
    }
  } else if (t < 228) {
    const int u = t - 84, g = u / 36, k = u - 36 * g;
    const GcRaw& r = raw_f[g];
    const bool w = (r.sf_s_set >> (k / 3) & 1) != 0;
    m.val = w ? r.sf_s[k] : 0;
    m.set = m.set || w;
  } else {
    const GcRaw& r = raw_f[t - 228];
    const bool w = r.count1_set != 0;
    m.val = w ? r.count1 : 0;
    m.set = m.set || w;
  }
  return m;
}

// slot t's value after frame f goes into the frame's records: its own field, and where the reference's
// one-past-the-end reads land (the first element of the NEXT [gr][ch] block)
PD_HD void merge_store(int t, const pdmp3_frame_bits& F, pdmp3_gc_side* R /* rec + f * 4 */, unsigned val) {
  if (t < 84) {
    const int g = t / 21, sfb = t - 21 * g;
    if (gc_active(F, g)) R[g].scalefac_l[sfb] = (uint8_t)val;
    if (sfb == 0 && g >= 1 && gc_active(F, g - 1) && !(F.iso & 0x08u)) R[g - 1].scalefac_l[21] = (uint8_t)val;   // (PDMP3_ISO_SF21: stays 0)
  } else if (t < 228) {
    const int u = t - 84, g = u / 36, k = u - 36 * g, sfb = k / 3, w = k - 3 * sfb;
    if (gc_active(F, g)) R[g].scalefac_s[sfb][w] = (uint8_t)val;
    if (sfb == 0 && g >= 1 && gc_active(F, g - 1) && !(F.iso & 0x10u)) R[g - 1].scalefac_s[12][w] = (uint8_t)val;   // (PDMP3_ISO_SF12)
    if (k == 0 && g == 0 && gc_active(F, 3) && !(F.iso & 0x08u)) R[3].scalefac_l[21] = (uint8_t)val;   // last block: scalefac_s follows
  } else {
    const int g = t - 228;
    if (gc_active(F, g)) R[g].count1 = (uint16_t)val;
  }
}

// sequential form (host test build; the device kernel k_merge does the same with wave scans over 64 frames)
// state_in must not alias state_out: granule-1 slots also read their twin's incoming value
PD_HD void merge_slot(int t, const GcRaw* raw, const pdmp3_frame_bits* F, int n, const uint16_t* state_in, uint16_t* state,
                      pdmp3_gc_side* rec) {
  const int tw = merge_twin(t);
  unsigned val = state_in[t], val0 = tw >= 0 ? state_in[tw] : 0;
  for (int f = 0; f < n; f++) {
    const MergeIn m = merge_load(t, raw + (size_t)f * 4, (F[f].frame & PDMP3_FR_NEWSTREAM) != 0);
    if (m.set0) val0 = m.val0;
    if (m.set) val = m.val;
    if (m.copy) val = val0;
    merge_store(t, F[f], rec + (size_t)f * 4, val);
  }
  state[t] = (uint16_t)val;
}

// ---------------------------------------------------------------------------
// The same merge by BLOCKS of 32 frames (engine.hip: k_merge_outcome, k_merge_apply).  A slot's chain through a window
// is cut where the blocks meet: a first kernel finds what each block does to each slot without knowing what reaches it
// -- nothing, a value, or (granule-1 scalefac_l slots, scfsi) "the twin's value as it reached this block" -- and, per
// eight blocks, what the eight do together; the second walks those outcomes from the window's start up to its own
// block (a lane per slot: at most 31 super-blocks and 7 blocks in a window of 8192 frames), then through its 32 frames
// with the frames' inputs and the records under construction in LDS.  Round 1-4's kernel had a
// workgroup per slot read one byte out of every frame's 320 and write one into every frame's 512: 47 us a window of
// 8192 frames, all of it the texture path's one-lane-per-line accesses.
// ---------------------------------------------------------------------------
constexpr int kMergeBlk = 32;                     // frames of a block
constexpr int kMergeSuper = 8;                    // blocks whose outcomes are composed into one (a "super-block": 256 frames)
constexpr int kMergeLanes = 256;                  // threads of a workgroup and entries of a row of outcomes; lane t < kMergeSlots is slot t
constexpr int kMergeBatch = 8;                    // frames (rows of outcomes) whose LDS reads are under way together
constexpr unsigned kOutValue = 1u << 14, kOutTwin = 2u << 14, kOutKind = 3u << 14, kOutVal = 0x3fffu;

// merge_load / merge_store for one slot with everything that depends on the slot alone worked out once: where in a
// frame's 320 bytes of merge input (GcRaw [4]) its "set" bit, its value, its twin's and its copy bit are, where in the
// frame's 512 bytes of records its value goes and where the reference's one-past-the-end read of a neighbour finds it.
// The per-frame loop is then five 32-bit reads, shifts and selects, no branch on the slot.
struct MergeLane {
  uint16_t set_w, set0_w, val_w, val0_w, copy_w;     // byte offsets of aligned 32-bit words in the frame's merge input
  uint8_t set_bit, set0_bit, val_sh, val0_sh;
  uint32_t val_mask, copy_mask;                      // copy_mask 0: no twin (set0 / val0 then read the slot's own words, unused)
  uint16_t own_off, alias_off;                       // byte offsets in the frame's records; alias_off 0xffff: none
  uint8_t own_g, alias_g, alias_iso, own16;          // stores happen when gc own_g / alias_g is active and (iso & alias_iso) == 0
};
// per frame: bit 0 PDMP3_FR_NEWSTREAM, bits 8-11 gc g is active, bits 16-23 the frame's PDMP3_ISO_* switches
PD_HD uint32_t merge_frame_meta(unsigned frame, unsigned iso) {
  const bool mono = ((frame & PDMP3_FR_MODE_MASK) >> PDMP3_FR_MODE_SHIFT) == 3;
  return ((frame & PDMP3_FR_NEWSTREAM) ? 1u : 0u) | (mono ? 0x500u : 0xf00u) | (iso & 0xffu) << 16;
}
PD_HD uint32_t merge_frame_meta(const pdmp3_frame_bits& F) { return merge_frame_meta(F.frame, F.iso); }
static_assert(offsetof(pdmp3_frame_bits, frame) == 0 && offsetof(pdmp3_frame_bits, iso) == 3, "k_merge_apply reads both with the record's first word");
PD_HD MergeLane merge_lane(int t) {
  MergeLane L{};
  const unsigned kRaw = (unsigned)sizeof(GcRaw), kRec = (unsigned)sizeof(pdmp3_gc_side);
  const unsigned o_set_l = (unsigned)offsetof(GcRaw, sf_l_set), o_set_s = (unsigned)offsetof(GcRaw, sf_s_set);
  const unsigned o_sf_l = (unsigned)offsetof(GcRaw, sf_l), o_sf_s = (unsigned)offsetof(GcRaw, sf_s);
  const unsigned r_sf_l = (unsigned)offsetof(pdmp3_gc_side, scalefac_l), r_sf_s = (unsigned)offsetof(pdmp3_gc_side, scalefac_s);
  L.alias_off = 0xffff; L.val_mask = 0xffu;
  unsigned set_byte, val_byte, set0_byte, val0_byte;
  if (t < 84) {
    const unsigned g = (unsigned)t / 21, sfb = (unsigned)t - 21 * g;
    set_byte = g * kRaw + o_set_l; L.set_bit = (uint8_t)sfb;
    val_byte = g * kRaw + o_sf_l + sfb;
    set0_byte = set_byte; val0_byte = val_byte; L.set0_bit = L.set_bit;
    if (g >= 2) {
      const unsigned grp = sfb < 6 ? 0 : sfb < 11 ? 1 : sfb < 16 ? 2 : 3;
      set0_byte = (g & 1) * kRaw + o_set_l;
      val0_byte = (g & 1) * kRaw + o_sf_l + sfb;
      L.copy_w = (uint16_t)(g * kRaw + ((unsigned)offsetof(GcRaw, sf_l_copy) & ~3u));
      L.copy_mask = 1u << (8 * ((unsigned)offsetof(GcRaw, sf_l_copy) & 3u) + grp);
    }
    L.own_off = (uint16_t)(g * kRec + r_sf_l + sfb); L.own_g = (uint8_t)g;
    if (sfb == 0 && g >= 1) { L.alias_off = (uint16_t)((g - 1) * kRec + r_sf_l + 21); L.alias_g = (uint8_t)(g - 1); L.alias_iso = 0x08; }
  } else if (t < 228) {
    const unsigned u = (unsigned)t - 84, g = u / 36, k = u - 36 * g, sfb = k / 3, w = k - 3 * sfb;
    set_byte = g * kRaw + o_set_s; L.set_bit = (uint8_t)sfb;
    val_byte = g * kRaw + o_sf_s + k;
    set0_byte = set_byte; val0_byte = val_byte; L.set0_bit = L.set_bit;
    L.own_off = (uint16_t)(g * kRec + r_sf_s + k); L.own_g = (uint8_t)g;
    if (sfb == 0 && g >= 1) { L.alias_off = (uint16_t)((g - 1) * kRec + r_sf_s + 36 + w); L.alias_g = (uint8_t)(g - 1); L.alias_iso = 0x10; }
    if (k == 0 && g == 0) { L.alias_off = (uint16_t)(3 * kRec + r_sf_l + 21); L.alias_g = 3; L.alias_iso = 0x08; }   // last block: scalefac_s follows
  } else {
    const unsigned g = (unsigned)t - 228;
    set_byte = g * kRaw + (unsigned)offsetof(GcRaw, count1_set); L.set_bit = 0;      // (count1_set is 0 or 1)
    val_byte = g * kRaw + (unsigned)offsetof(GcRaw, count1); L.val_mask = 0xffffu;
    set0_byte = set_byte; val0_byte = val_byte; L.set0_bit = 0;
    L.own_off = (uint16_t)(g * kRec + (unsigned)offsetof(pdmp3_gc_side, count1)); L.own_g = (uint8_t)g; L.own16 = 1;
  }
  L.set_w = (uint16_t)(set_byte & ~3u); L.set_bit = (uint8_t)(L.set_bit + 8 * (set_byte & 3u));
  L.set0_w = (uint16_t)(set0_byte & ~3u); L.set0_bit = (uint8_t)(L.set0_bit + 8 * (set0_byte & 3u));
  L.val_w = (uint16_t)(val_byte & ~3u); L.val_sh = (uint8_t)(8 * (val_byte & 3u));
  L.val0_w = (uint16_t)(val0_byte & ~3u); L.val0_sh = (uint8_t)(8 * (val0_byte & 3u));
  return L;
}
static_assert(offsetof(GcRaw, count1) % 4 == 0 && offsetof(GcRaw, sf_l_set) % 4 == 0 && offsetof(GcRaw, sf_s_set) % 4 == 0 &&
              sizeof(GcRaw) % 4 == 0, "merge_lane reads aligned words");
// frame i of a block: rawf = its 320 bytes of merge input as words.  TWIN = false: a slot known to have no twin (set0 /
// val0 / copy are not looked at; what a wave's slots are is known per wave: merge_wave_kind)
template <bool TWIN>
PD_HD MergeIn merge_lane_load(const MergeLane& L, const uint32_t* rawf, bool newstream) {
  MergeIn m{false, false, false, 0, 0};
  const bool w = (rawf[L.set_w >> 2] >> L.set_bit & 1u) != 0;
  const unsigned v = (rawf[L.val_w >> 2] >> L.val_sh) & L.val_mask;
  m.val = w ? v : 0u; m.set = w || newstream;
  if (TWIN) {
    const bool w0 = (rawf[L.set0_w >> 2] >> L.set0_bit & 1u) != 0;
    const unsigned v0 = (rawf[L.val0_w >> 2] >> L.val0_sh) & 0xffu;
    m.val0 = w0 ? v0 : 0u; m.set0 = w0 || newstream;
    m.copy = (rawf[L.copy_w >> 2] & L.copy_mask) != 0;
  }
  return m;
}
// (no branches: a store that is not to happen goes to `trash`, a byte of the lane's own -- the loop over a block's frames
//  stays one basic block and the reads of the frames ahead are under way while these are issued.  WIDE = false: no slot
//  of the wave is count1's, the only 16-bit value)
template <bool WIDE>
PD_HD void merge_lane_store(const MergeLane& L, uint32_t meta, uint8_t* recf, unsigned val, uint8_t* trash) {
  const bool own = (meta >> (8 + L.own_g) & 1u) != 0;
  const bool alias = L.alias_off != 0xffff && (meta >> (8 + L.alias_g) & 1u) && !((meta >> 16) & L.alias_iso);
  *(own ? recf + L.own_off : trash) = (uint8_t)val;
  if (WIDE) *(own && L.own16 ? recf + L.own_off + 1 : trash) = (uint8_t)(val >> 8);
  *(alias ? recf + L.alias_off : trash) = (uint8_t)val;
}
// slots [t0, t0 + 64) of a wave: bit 0 = some have twins (granule-1 scalefac_l: 42 .. 83), bit 1 = count1's are among them
PD_HD unsigned merge_wave_kind(int t0) { return (t0 < 84 && t0 + 64 > 42 ? 1u : 0u) | (t0 + 64 > 228 ? 2u : 0u); }

// what the nb frames of a block do to slot t: 0 (nothing), kOutValue | v, or kOutTwin.  frame_flags: pdmp3_frame_bits.frame
// (Eight frames at a time, reads first: a lane's walk is a chain of LDS round trips otherwise.  TAIL: fewer than eight
//  are left -- past the last one it is read again and ignored.)
struct MergeOut { bool has, has0, twin; unsigned val, val0; };
template <bool TWIN, bool TAIL>
PD_HD void merge_outcome_batch(const MergeLane& L, const uint32_t* rw, const uint8_t* frame_flags, int i0, int nb, MergeOut& o) {
  MergeIn m[kMergeBatch];
  PD_UNROLL for (int k = 0; k < kMergeBatch; k++) {
    const int i = !TAIL || i0 + k < nb ? i0 + k : nb - 1;
    m[k] = merge_lane_load<TWIN>(L, rw + (size_t)i * (4 * sizeof(GcRaw) / 4), (frame_flags[i] & PDMP3_FR_NEWSTREAM) != 0);
  }
  PD_UNROLL for (int k = 0; k < kMergeBatch; k++) {
    const bool valid = !TAIL || i0 + k < nb, set = valid && m[k].set;
    o.val = set ? m[k].val : o.val; o.has = o.has || set;
    if (TWIN) {
      const bool set0 = valid && m[k].set0, copy = valid && m[k].copy;
      o.val0 = set0 ? m[k].val0 : o.val0; o.has0 = o.has0 || set0;
      o.twin = o.twin && !set;
      o.twin = copy ? !o.has0 : o.twin; o.has = o.has || copy; o.val = copy ? o.val0 : o.val;
    }
  }
}
template <bool TWIN>
PD_HD unsigned merge_block_outcome_t(int t, const GcRaw* raw_blk, const uint8_t* frame_flags, int nb) {
  const MergeLane L = merge_lane(t);
  const uint32_t* rw = reinterpret_cast<const uint32_t*>(raw_blk);
  MergeOut o{false, false, false, 0, 0};
  int i0 = 0;
  for (; i0 + kMergeBatch <= nb; i0 += kMergeBatch) merge_outcome_batch<TWIN, false>(L, rw, frame_flags, i0, nb, o);
  if (i0 < nb) merge_outcome_batch<TWIN, true>(L, rw, frame_flags, i0, nb, o);
  return !o.has ? 0u : o.twin ? kOutTwin : (kOutValue | (o.val & kOutVal));
}
// kind: merge_wave_kind of the wave the slot is in (the same for all its lanes), or 3 where that is not known
PD_HD unsigned merge_block_outcome(int t, const GcRaw* raw_blk, const uint8_t* frame_flags, int nb, unsigned kind = 3) {
  return (kind & 1u) ? merge_block_outcome_t<true>(t, raw_blk, frame_flags, nb) : merge_block_outcome_t<false>(t, raw_blk, frame_flags, nb);
}

// one block (or super-block) further: o / o0 = the outcomes of the slot and of its twin (0 for slots that have none)
PD_HD void merge_carry_step(unsigned o, unsigned o0, unsigned& val, unsigned& val0) {
  val = (o & kOutKind) == kOutValue ? (o & kOutVal) : (o & kOutKind) == kOutTwin ? val0 : val;      // (the twin's value in FRONT of the block)
  val0 = (o0 & kOutKind) == kOutValue ? (o0 & kOutVal) : val0;
}
// nr rows of outcomes (kMergeLanes entries each) in a row: eight at a time, reads first
template <bool TWIN>
PD_HD void merge_carry_rows(const uint32_t* rows, int nr, int t, int tw, unsigned& val, unsigned& val0) {
  for (int j0 = 0; j0 < nr; j0 += kMergeBatch) {
    unsigned o[kMergeBatch], o0[kMergeBatch];
    PD_UNROLL for (int k = 0; k < kMergeBatch; k++) {
      const int j = j0 + k < nr ? j0 + k : nr - 1;
      o[k] = rows[(size_t)j * kMergeLanes + t];
      o0[k] = TWIN && tw >= 0 ? rows[(size_t)j * kMergeLanes + tw] : 0u;
    }
    PD_UNROLL for (int k = 0; k < kMergeBatch; k++)
      if (j0 + k < nr) merge_carry_step(o[k], o0[k], val, val0);
  }
}
// the same on outcomes instead of values: acc / acc0 = what the blocks so far do together (0, kOutValue | v, kOutTwin = "the
// twin's value in front of the FIRST of them")
PD_HD void merge_compose_step(unsigned o, unsigned o0, unsigned& acc, unsigned& acc0) {
  const unsigned k = o & kOutKind;
  acc = k == kOutValue ? o : k == kOutTwin ? ((acc0 & kOutKind) == kOutValue ? acc0 : kOutTwin) : acc;
  acc0 = (o0 & kOutKind) == kOutValue ? o0 : acc0;
}

// the block's frames with what reaches it: merge_slot's loop; meta[i] = merge_frame_meta of frame i; returns the slot's
// value behind the block
template <bool TWIN, bool WIDE, bool TAIL>
PD_HD void merge_apply_batch(const MergeLane& L, const uint32_t* rw, const uint32_t* meta, int i0, int nb, uint8_t* rec, uint8_t* trash,
                             unsigned& val, unsigned& val0) {
  MergeIn m[kMergeBatch];
  uint32_t mt[kMergeBatch];
  PD_UNROLL for (int k = 0; k < kMergeBatch; k++) {
    const int i = !TAIL || i0 + k < nb ? i0 + k : nb - 1;
    mt[k] = meta[i];
    m[k] = merge_lane_load<TWIN>(L, rw + (size_t)i * (4 * sizeof(GcRaw) / 4), (mt[k] & 1u) != 0);
  }
  PD_UNROLL for (int k = 0; k < kMergeBatch; k++) {
    const bool valid = !TAIL || i0 + k < nb;
    val = valid && m[k].set ? m[k].val : val;
    if (TWIN) {
      val0 = valid && m[k].set0 ? m[k].val0 : val0;
      val = valid && m[k].copy ? val0 : val;
    }
    const int i = valid ? i0 + k : nb - 1;                     // (an ignored frame: every store to `trash`)
    merge_lane_store<WIDE>(L, valid ? mt[k] : 0u, rec + (size_t)i * (4 * sizeof(pdmp3_gc_side)), val, trash);
  }
}
template <bool TWIN, bool WIDE>
PD_HD unsigned merge_block_apply_t(int t, unsigned val, unsigned val0, const GcRaw* raw_blk, const uint32_t* meta, int nb,
                                   pdmp3_gc_side* rec_blk, uint8_t* trash) {
  const MergeLane L = merge_lane(t);
  const uint32_t* rw = reinterpret_cast<const uint32_t*>(raw_blk);
  uint8_t* rec = reinterpret_cast<uint8_t*>(rec_blk);
  int i0 = 0;
  for (; i0 + kMergeBatch <= nb; i0 += kMergeBatch) merge_apply_batch<TWIN, WIDE, false>(L, rw, meta, i0, nb, rec, trash, val, val0);
  if (i0 < nb) merge_apply_batch<TWIN, WIDE, true>(L, rw, meta, i0, nb, rec, trash, val, val0);
  return val;
}
PD_HD unsigned merge_block_apply(int t, unsigned val, unsigned val0, const GcRaw* raw_blk, const uint32_t* meta, int nb,
                                 pdmp3_gc_side* rec_blk, uint8_t* trash, unsigned kind = 3) {
  switch (kind & 3u) {
    case 0: return merge_block_apply_t<false, false>(t, val, val0, raw_blk, meta, nb, rec_blk, trash);
    case 1: return merge_block_apply_t<true, false>(t, val, val0, raw_blk, meta, nb, rec_blk, trash);
    case 2: return merge_block_apply_t<false, true>(t, val, val0, raw_blk, meta, nb, rec_blk, trash);
    default: return merge_block_apply_t<true, true>(t, val, val0, raw_blk, meta, nb, rec_blk, trash);
  }
}

// the two kernels one after the other on the host (test build): side must hold the records' side fields already
// (unpack_records); outc: merge_outcome_rows(n) x kMergeLanes entries of scratch -- a row per block, then a row per super-block
PD_HD int merge_outcome_rows(int n_frames) {
  const int nblk = (n_frames + kMergeBlk - 1) / kMergeBlk;
  return nblk + (nblk + kMergeSuper - 1) / kMergeSuper;
}
PD_HD void merge_blocks(const GcRaw* raw, const pdmp3_frame_bits* F, int n, const uint16_t* state_in, uint16_t* state,
                        pdmp3_gc_side* rec, uint32_t* outc) {
  const int nblk = (n + kMergeBlk - 1) / kMergeBlk;
  uint32_t* sup = outc + (size_t)nblk * kMergeLanes;
  for (int b = 0; b < nblk; b++) {
    const int f0 = b * kMergeBlk, nb = n - f0 < kMergeBlk ? n - f0 : kMergeBlk;
    uint8_t fl[kMergeBlk];
    for (int i = 0; i < nb; i++) fl[i] = F[f0 + i].frame;
    for (int t = 0; t < kMergeSlots; t++)
      outc[(size_t)b * kMergeLanes + t] = merge_block_outcome(t, raw + (size_t)f0 * 4, fl, nb, merge_wave_kind(t & ~63));
    if (b % kMergeSuper == kMergeSuper - 1 || b == nblk - 1)              // (the device: whichever of the eight workgroups is through last)
      for (int t = 0; t < kMergeSlots; t++) {
        const int tw = merge_twin(t), b0 = b - b % kMergeSuper;
        unsigned acc = 0, acc0 = 0;
        for (int j = b0; j <= b; j++) merge_compose_step(outc[(size_t)j * kMergeLanes + t], tw >= 0 ? outc[(size_t)j * kMergeLanes + tw] : 0u, acc, acc0);
        sup[(size_t)(b / kMergeSuper) * kMergeLanes + t] = acc;
      }
  }
  for (int t = 0; t < kMergeSlots; t++) state[t] = state_in[t];        // (a window without frames)
  for (int b = 0; b < nblk; b++) {
    const int f0 = b * kMergeBlk, nb = n - f0 < kMergeBlk ? n - f0 : kMergeBlk;
    uint32_t meta[kMergeBlk];
    for (int i = 0; i < nb; i++) meta[i] = merge_frame_meta(F[f0 + i]);
    for (int t = 0; t < kMergeSlots; t++) {
      const int tw = merge_twin(t);
      unsigned val = state_in[t], val0 = tw >= 0 ? state_in[tw] : 0;
      const uint32_t* own = outc + (size_t)(b - b % kMergeSuper) * kMergeLanes;
      if (merge_wave_kind(t & ~63) & 1u) {
        merge_carry_rows<true>(sup, b / kMergeSuper, t, tw, val, val0);
        merge_carry_rows<true>(own, b % kMergeSuper, t, tw, val, val0);
      } else {
        merge_carry_rows<false>(sup, b / kMergeSuper, t, tw, val, val0);
        merge_carry_rows<false>(own, b % kMergeSuper, t, tw, val, val0);
      }
      uint8_t trash;
      val = merge_block_apply(t, val, val0, raw + (size_t)f0 * 4, meta, nb, rec + (size_t)f0 * 4, &trash, merge_wave_kind(t & ~63));
      if (b == nblk - 1) state[t] = (uint16_t)val;
    }
  }
}

}  // namespace pdmp3

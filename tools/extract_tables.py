#!/usr/bin/env python3
"""Extract the reference decoder's constant DATA tables and re-emit them in this
repo's own formats.

Run in the build container only (needs /root/reference/pdmp3.c).  Outputs are
committed, so nothing here runs on the GPU box.

What is emitted (all values are data, bit-for-bit; no reference code is copied):

  oracle/oracle_tables.h          C arrays for the CPU oracle: float tables as
                                  C99 hex-float literals, the 16-bit Huffman
                                  node array, sfb boundaries, bitrates, slen.
  pdmp3_amd/csrc/tables_data.h    The same float tables for the product (device
                                  upload) + sfb boundaries + *derived* Huffman
                                  code books (codeword lists obtained by a DFS
                                  over the node array; table 33 twice: as the
                                  reference mis-points it (H1) and ISO-correct).

Provenance (pdmp3.c line numbers):
  cs/ca/is_ratios P:573-575, g_imdct_win P:577-603, cos_N12 P:606-619,
  cos_N36 P:620-729, g_synth_dtbl P:740-870, g_huffman_table P:235-515,
  g_huffman_main P:535-570, g_mpeg1_bitrates P:517-528, g_sampling_frequency
  P:529, mpeg1_scalefac_sizes P:530-533, g_sf_band_indices P:879-892,
  pretab P:2123.

The low-precision decimal literals (cos_N36 is off from the true cosine by up
to 6.8e-6) cannot be regenerated from a formula, which is why they are carried
as data.  CRC-32 of each emitted array is printed and checked against
SURVEY.md appendix C.
"""
import os
import re
import struct
import sys
import zlib

import numpy as np

REF = "/root/reference/pdmp3.c"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

EXPECT_CRC = {
    "cs": 0x40A1F1A8, "ca": 0x2B144553, "is_ratios": 0x0159E1E9,
    "imdct_win": 0x33A8CA30, "cos_n12": 0x9F16EF93, "cos_n36": 0xA72E18BD,
    "synth_dtbl": 0x39C3A499, "huff_nodes": 0x3756A2E8,
}


def strip_comments(s):
    s = re.sub(r"/\*.*?\*/", "", s, flags=re.S)
    s = re.sub(r"//[^\n]*", "", s)
    return s


def brace_body(src, start):
    """Return text of the {...} initializer that begins at/after `start`."""
    i = src.index("{", start)
    depth, j = 0, i
    while True:
        c = src[j]
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                return src[i:j + 1]
        j += 1


def float_table(src, name):
    m = re.search(r"\b" + name + r"\s*(\[[^\]]*\])+\s*=\s*\{", src)
    body = strip_comments(brace_body(src, m.start()))
    toks = re.findall(r"-?\d+\.\d+(?:[eE][-+]?\d+)?", body)
    # C semantics: literal with 'f' suffix is parsed to float directly; without
    # suffix it is a double converted to float.  Both equal round-to-nearest of
    # the decimal string for these magnitudes (double rounding cannot bite at
    # <=9 significant digits).
    return np.array([np.float32(float(t)) for t in toks], dtype=np.float32)


def int_table(src, name, hexa=False):
    m = re.search(r"\b" + name + r"\s*(\[[^\]]*\])*\s*=\s*\{", src)
    body = strip_comments(brace_body(src, m.start()))
    if hexa:
        return [int(t, 16) for t in re.findall(r"0x[0-9a-fA-F]+", body)]
    body = re.sub(r"\*\s*Hz", "", body)
    return [int(t) for t in re.findall(r"-?\d+", body)]


def crc(arr_bytes):
    return zlib.crc32(arr_bytes) & 0xFFFFFFFF


def hexf(v):
    """Exact C99 hex-float literal for a float32 value."""
    v = float(np.float32(v))
    if v == 0.0:
        return "0x0p+0f"
    return v.hex() + "f"


def emit_f32(name, arr, per_line=6):
    out = ["static const float %s[%d] = {" % (name, arr.size)]
    flat = arr.reshape(-1)
    for i in range(0, flat.size, per_line):
        out.append("  " + ", ".join(hexf(x) for x in flat[i:i + per_line]) + ",")
    out.append("};")
    return "\n".join(out)


def emit_int(name, ctype, vals, per_line=12, fmt="%d"):
    out = ["static const %s %s[%d] = {" % (ctype, name, len(vals))]
    for i in range(0, len(vals), per_line):
        out.append("  " + ", ".join(fmt % v for v in vals[i:i + per_line]) + ",")
    out.append("};")
    return "\n".join(out)


def huff_main(src):
    """Parse g_huffman_main: list of (offset or None, treelen, linbits)."""
    m = re.search(r"g_huffman_main\s*\[34\]\s*=\s*\{", src)
    body = strip_comments(brace_body(src, m.start()))
    rows = re.findall(r"\{\s*(NULL|g_huffman_table(?:\s*\+\s*\d+)?)\s*,\s*(\d+)\s*,\s*(\d+)\s*\}", body)
    res = []
    for ptr, tl, lb in rows:
        if ptr == "NULL":
            off = None
        else:
            mm = re.search(r"\+\s*(\d+)", ptr)
            off = int(mm.group(1)) if mm else 0
        res.append((off, int(tl), int(lb)))
    assert len(res) == 34
    return res


def walk_codes(nodes, off, treelen, maxbits=32):
    """Enumerate every bit string the reference's tree walk (P:1603-1621) can
    consume, with the value it yields.  Returns list of (bits, nbits, val,
    is_error).  A path ends when a leaf is reached *at the loop head*, or when
    the loop condition (--bitsleft>0 && point<treelen) fails (=> error, x=y=0).
    """
    out = []
    # state: (point, bits, nbits, bitsleft)
    stack = [(0, 0, 0, maxbits)]
    while stack:
        point, bits, nbits, bitsleft = stack.pop()
        node = nodes[off + point]
        if (node & 0xFF00) == 0:
            out.append((bits, nbits, node & 0xFF, False))
            continue
        for bit in (0, 1):
            p = point
            if bit:
                while (nodes[off + p] & 0xFF) >= 250:
                    p += nodes[off + p] & 0xFF
                p += nodes[off + p] & 0xFF
            else:
                while (nodes[off + p] >> 8) >= 250:
                    p += nodes[off + p] >> 8
                p += nodes[off + p] >> 8
            nb = nbits + 1
            b = (bits << 1) | bit
            bl = bitsleft - 1
            if bl > 0 and p < treelen:
                stack.append((p, b, nb, bl))
            else:
                out.append((b, nb, 0, True))  # error exit: x=y=0
    out.sort(key=lambda t: (t[1], t[0]))
    return out


def main():
    src = strip_comments(open(REF).read())
    f = {}
    f["cs"] = float_table(src, "cs")
    f["ca"] = float_table(src, "ca")
    f["is_ratios"] = float_table(src, "is_ratios")
    f["imdct_win"] = float_table(src, "g_imdct_win")
    f["cos_n12"] = float_table(src, "cos_N12")
    f["cos_n36"] = float_table(src, "cos_N36")
    f["synth_dtbl"] = float_table(src, "g_synth_dtbl")
    sizes = {"cs": 8, "ca": 8, "is_ratios": 6, "imdct_win": 144, "cos_n12": 72,
             "cos_n36": 648, "synth_dtbl": 512}
    for k, n in sizes.items():
        assert f[k].size == n, (k, f[k].size)
        c = crc(f[k].astype("<f4").tobytes())
        ok = (c == EXPECT_CRC[k])
        print("%-12s n=%4d crc32=%08x %s" % (k, n, c, "ok" if ok else "MISMATCH"))
        assert ok

    nodes = int_table(src, "g_huffman_table", hexa=True)
    assert len(nodes) == 2804
    c = crc(struct.pack("<%dH" % len(nodes), *nodes))
    print("huff_nodes   n=%4d crc32=%08x" % (len(nodes), c))
    assert c == EXPECT_CRC["huff_nodes"]
    hmain = huff_main(src)

    bitrates = int_table(src, "g_mpeg1_bitrates")
    assert len(bitrates) == 45
    sfreqs = [44100, 48000, 32000]
    slen = int_table(src, "mpeg1_scalefac_sizes")
    assert len(slen) == 32
    sfb = int_table(src, "g_sf_band_indices")
    assert len(sfb) == 3 * 37
    sfb_l = [sfb[i * 37:i * 37 + 23] for i in range(3)]
    sfb_s = [sfb[i * 37 + 23:i * 37 + 37] for i in range(3)]
    m = re.search(r"pretab\[21\]\s*=\s*\{([^}]*)\}", src)
    pretab = [int(t) for t in re.findall(r"\d+", m.group(1))]
    assert len(pretab) == 21

    hdr = ("/* GENERATED by tools/extract_tables.py -- constant DATA of the reference\n"
           " * decoder (technosaurus/PDMP3 pdmp3.c), carried bit-for-bit because the\n"
           " * low-precision literals cannot be regenerated by formula (SURVEY.md H19).\n"
           " * Do not edit by hand. */\n")

    # ---------------- oracle header ----------------
    o = [hdr, "#ifndef ORACLE_TABLES_H\n#define ORACLE_TABLES_H\n#include <stdint.h>\n"]
    o.append("/* P:573 */\n" + emit_f32("ot_cs", f["cs"]))
    o.append("/* P:574 */\n" + emit_f32("ot_ca", f["ca"]))
    o.append("/* P:575 */\n" + emit_f32("ot_is_ratios", f["is_ratios"]))
    o.append("/* P:577-603, [4][36] */\n" + emit_f32("ot_imdct_win", f["imdct_win"]))
    o.append("/* P:606-619, [6][12] */\n" + emit_f32("ot_cos_n12", f["cos_n12"]))
    o.append("/* P:620-729, [18][36] */\n" + emit_f32("ot_cos_n36", f["cos_n36"]))
    o.append("/* P:740-870 */\n" + emit_f32("ot_synth_dtbl", f["synth_dtbl"]))
    o.append("/* P:235-515: node = (left_off<<8)|right_off, hi byte 0 => leaf (x<<4|y) */\n" +
             emit_int("ot_huff_nodes", "uint16_t", nodes, 10, "0x%04x"))
    o.append("/* P:535-570: {offset into ot_huff_nodes or -1, treelen, linbits}; row 33 keeps the\n"
             " * reference's wrong offset 2261 (H1). */")
    rows = ["  {%d, %d, %d}," % (-1 if off is None else off, tl, lb) for off, tl, lb in hmain]
    o.append("static const struct { int off; int treelen; int linbits; } ot_huff_main[34] = {\n" +
             "\n".join(rows) + "\n};")
    o.append("/* P:517-528 [layer-1][bitrate_index] */\n" + emit_int("ot_bitrates", "uint32_t", bitrates, 8))
    o.append("/* P:529 */\n" + emit_int("ot_sfreq", "uint32_t", sfreqs))
    o.append("/* P:530-533 [16][2] */\n" + emit_int("ot_slen", "uint8_t", slen, 16))
    o.append("/* P:879-892: l[23] then s[14] per sampling-frequency index, CONTIGUOUS as in the\n"
             " * reference struct so that l[23], l[24] read s[0], s[1] (H7). */\n" +
             emit_int("ot_sfb", "uint32_t", sfb, 23))
    o.append("/* P:2123, index 21 reads past the array => 0 (H4) */\n" +
             emit_int("ot_pretab", "uint8_t", pretab + [0], 22))
    o.append("#endif\n")
    with open(os.path.join(ROOT, "oracle", "oracle_tables.h"), "w") as fh:
        fh.write("\n".join(o))

    # ---------------- product header ----------------
    p = [hdr, "#ifndef PDMP3_TABLES_DATA_H\n#define PDMP3_TABLES_DATA_H\n#include <stdint.h>\n"]
    p.append(emit_f32("kAliasCs", f["cs"]))
    p.append(emit_f32("kAliasCa", f["ca"]))
    p.append(emit_f32("kIsRatios", f["is_ratios"]))
    p.append("/* [bt][36] */\n" + emit_f32("kImdctWin", f["imdct_win"]))
    p.append("/* [m][p] 6x12 */\n" + emit_f32("kCosN12", f["cos_n12"]))
    p.append("/* [m][p] 18x36 */\n" + emit_f32("kCosN36", f["cos_n36"]))
    p.append(emit_f32("kSynthD", f["synth_dtbl"]))
    for i in range(3):
        p.append(emit_int("kSfbLong%d" % i, "uint16_t", sfb_l[i], 23))
        p.append(emit_int("kSfbShort%d" % i, "uint16_t", sfb_s[i], 14))
    p.append(emit_int("kPretab", "uint8_t", pretab + [0], 22))
    p.append(emit_int("kBitratesL3", "uint32_t", bitrates[30:45], 8))
    p.append(emit_int("kSampleRates", "uint32_t", sfreqs))
    p.append(emit_int("kSlen", "uint8_t", slen, 16))

    # Huffman code books (derived).  Entry = {code, len, val, err}.
    p.append("typedef struct { uint32_t code; uint8_t len; uint8_t val; uint8_t err; } pdmp3_hcode;")
    book_of_table = []
    books = {}
    for t, (off, tl, lb) in enumerate(hmain):
        if off is None:
            book_of_table.append(-1)
            continue
        key = (off, tl)
        if key not in books:
            books[key] = (len(books), walk_codes(nodes, off, tl))
        book_of_table.append(books[key][0])
    # ISO-correct table 33 (the array's last 31 nodes) as an extra book
    iso33 = (2773, 31)
    books[iso33] = (len(books), walk_codes(nodes, iso33[0], iso33[1]))
    order = sorted(books.items(), key=lambda kv: kv[1][0])
    counts = []
    for (off, tl), (idx, codes) in order:
        maxlen = max(c[1] for c in codes)
        nerr = sum(1 for c in codes if c[3])
        counts.append(len(codes))
        p.append("/* book %d: nodes[%d..%d), %d codes, max len %d, %d error exits */" %
                 (idx, off, off + tl, len(codes), maxlen, nerr))
        rows = ["  {0x%x, %d, 0x%02x, %d}," % (c[0], c[1], c[2], 1 if c[3] else 0) for c in codes]
        p.append("static const pdmp3_hcode kHuffBook%d[%d] = {\n%s\n};" % (idx, len(codes), "\n".join(rows)))
    p.append("#define PDMP3_NUM_HUFF_BOOKS %d" % len(order))
    p.append("static const pdmp3_hcode* const kHuffBooks[%d] = { %s };" %
             (len(order), ", ".join("kHuffBook%d" % i for i in range(len(order)))))
    p.append(emit_int("kHuffBookSize", "uint16_t", counts))
    p.append("/* table number -> book (-1 = empty table); table 33 -> the H1 mis-pointed book */\n" +
             emit_int("kHuffBookOfTable", "int8_t", book_of_table, 17))
    p.append("#define PDMP3_HUFF_BOOK_ISO33 %d" % books[iso33][0])
    p.append(emit_int("kHuffLinbits", "uint8_t", [lb for _, _, lb in hmain], 17))
    p.append("#endif\n")
    os.makedirs(os.path.join(ROOT, "pdmp3_amd", "csrc"), exist_ok=True)
    with open(os.path.join(ROOT, "pdmp3_amd", "csrc", "tables_data.h"), "w") as fh:
        fh.write("\n".join(p))
    print("books:", [(k, v[0], len(v[1])) for k, v in order])


if __name__ == "__main__":
    sys.exit(main())

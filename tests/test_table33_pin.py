"""SURVEY H1 / Appendix D, VERDICT r04 #7: the PDMP3_ISO_TABLE33 switch, half-pinned.

The reference's g_huffman_main[33] points at g_huffman_table + 2261 (inside table 24, P:569); the tree the standard
means by table 33 (count1 table B: 4-bit code words) sits in the SAME array, its last 31 words (P:512-515).  The
switch decodes with that tree.  Pinned here:
  * tests/golden/huff_table33.json = those 31 words read out of the compiled reference's memory (oracle/_ref), plus
    what the reference's OWN Huffman_Decode (P:1593-1643) returns for every 8-bit input when its table-33 pointer is
    set there (tools/make_golden.py);
  * build container (oracle/_ref present): the fixture is re-derived from the library and compared;
  * everywhere: the oracle's restatement with ORC_ISO_TABLE33 and the product's code book for count1table_select = 2
    (pdmp3_amd/csrc/tables_data.h kHuffBook17, the one book behind the host LUT and the device tables) give exactly
    the fixture's quadruples.
What stays unpinned is that the STANDARD means this tree (nothing of the standard is in the image) and the other four
switches."""
import json
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "huff_table33.json")))
ISO_TABLE33 = 0x01


def test_fixture_is_the_reference_memory(reference):
    words, total = reference.huffman_table_words(GOLD["first"], 31)
    assert total == GOLD["total_words"] == 2804 and GOLD["first"] == total - 31
    assert [int(w) for w in words] == GOLD["words"]
    for bits in range(256):
        (v, w, x, y), used, res = reference.huffman_quad_at(GOLD["first"], bits, 8)
        assert res == 0 and [v, w, x, y, used] == GOLD["quads"]["%02x" % bits], hex(bits)
    # and the pointer the reference actually uses (H1) is NOT this tree: it decodes something else
    assert any(reference.huffman_quad_at(2261, b, 8)[0] != tuple(GOLD["quads"]["%02x" % b][:4]) for b in range(256))


def test_oracle_nodes_and_switch(oracle):
    nodes = oracle.huffman_nodes()
    assert nodes.size == 2804 and [int(w) for w in nodes[2773:]] == GOLD["words"]
    for bits in range(256):
        (v, w, x, y), used, res = oracle.huffman_quad(ISO_TABLE33, bits, 8)
        assert res == 0 and [v, w, x, y, used] == GOLD["quads"]["%02x" % bits], hex(bits)


def test_oracle_without_the_switch_is_the_reference_h1(oracle, reference):
    for bits in range(256):
        assert oracle.huffman_quad(0, bits, 8) == reference.huffman_quad_at(2261, bits, 8), hex(bits)


def _product_book(name):
    src = open(os.path.join(ROOT, "pdmp3_amd", "csrc", "tables_data.h")).read()
    m = re.search(r"static const pdmp3_hcode %s\[(\d+)\] = \{(.*?)\};" % name, src, re.S)
    rows = re.findall(r"\{0x([0-9a-f]+), (\d+), 0x([0-9a-f]+), (\d+)\}", m.group(2))
    assert len(rows) == int(m.group(1))
    return [(int(c, 16), int(l), int(v, 16), int(e)) for c, l, v, e in rows]


def test_product_code_book_is_that_tree():
    src = open(os.path.join(ROOT, "pdmp3_amd", "csrc", "tables_data.h")).read()
    assert re.search(r"#define PDMP3_HUFF_BOOK_ISO33 17\b", src)
    book = _product_book("kHuffBook17")
    assert len(book) == 16 and all(l == 4 and e == 0 for _, l, _, e in book)
    assert sorted(c for c, _, _, _ in book) == list(range(16))          # a complete prefix code of 4-bit words
    for code, length, val, _ in book:
        # the quadruple's magnitudes are the value's four bits v w x y (P:1628-1631); signs follow for the non-zero ones
        for signs in range(16):
            bits = (code << 4) | signs
            v, w, x, y, used = GOLD["quads"]["%02x" % bits]
            assert [abs(v), abs(w), abs(x), abs(y)] == [(val >> 3) & 1, (val >> 2) & 1, (val >> 1) & 1, val & 1], hex(bits)
            assert used == 4 + bin(val).count("1")
            k = 0
            for q in (v, w, x, y):                                         # sign bits in the order v, w, x, y
                if q != 0:
                    assert (q < 0) == bool((signs >> (3 - k)) & 1)
                    k += 1

#!/usr/bin/env python3
"""Generate tests/golden/ from the REAL reference (oracle/_ref, compiled from
/root/reference by oracle/Makefile).  Run in the build container only.

Fixtures are data: expected outputs of the reference for inputs that the tests
re-derive from (case name, seed) -- tests/corpus.py -- or from the integer-only
C2 generator.  Nothing of the reference's source is stored.

  tests/golden/<case>.npz       64 frames per case, kept small: sha256 of the int16 PCM of all frames and
                                of each of the 4 stage dumps, PCM of the first 4 frames and of the last one, stage3
                                float32 of the first 2 frames, sha256 of the inputs
  tests/golden/c2_prefix.npz    PCM of the first 32 frames of the C2 stream + sha256 of the
                                PCM of all 2048 frames
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import corpus  # noqa: E402
from oracle import oracle as orc  # noqa: E402

N_FRAMES = corpus.N_GOLDEN
C2_SEED = 0x5EED0000C2


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    orc.build()
    ref = orc.Reference()
    o = orc.Oracle()
    gold = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gold, exist_ok=True)
    for name in corpus.ALL_CASES:
        sp, sd = corpus.case(name, n=N_FRAMES)
        pcm, stg = ref.decode(sp, sd, stages=True)
        nch = 1 if ((int(sd["frame"][0, 0, 0]) >> 2) & 3) == 3 else 2
        np.savez_compressed(
            os.path.join(gold, name + ".npz"),
            n_frames=np.array([N_FRAMES]), pcm_sha=np.array([sha(pcm)]), pcm_head=pcm[:4], pcm_last=pcm[-1:],
            stage3_head=stg[:2, :, :, 3],
            stage_sha=np.array([sha(stg[:, :, :nch, k]) for k in range(4)]),
            input_sha=np.array([sha(sp), sha(sd)]))
        print("%-24s pcm sha %s" % (name, sha(pcm)[:16]))
    sp, sd = o.generate(C2_SEED, 0, 2048)
    pcm = ref.decode(sp, sd)
    np.savez_compressed(os.path.join(gold, "c2_prefix.npz"), pcm_head=pcm[:32],
                        pcm_sha_2048=np.array([sha(pcm)]), input_sha=np.array([sha(sp), sha(sd)]))
    print("c2 2048-frame pcm sha", sha(pcm)[:16])
    # the tree of the REAL table 33 -- the last 31 words of the reference's g_huffman_table, read
    # out of the compiled reference's memory -- and what the reference's own Huffman_Decode makes of every 8-bit pattern
    # (4-bit code word + up to 4 sign bits) when its table-33 pointer is set there
    import json
    words, total = ref.huffman_table_words(2773, 31)
    assert total == 2804
    quads = {}
    for bits in range(256):
        (v, w, x, y), used, res = ref.huffman_quad_at(2773, bits, 8)
        assert res == 0
        quads["%02x" % bits] = [v, w, x, y, used]
    json.dump({"what": "g_huffman_table[2773..2804) of /root/reference/pdmp3.c as compiled into oracle/_ref, and the reference's "
                       "Huffman_Decode over that tree for every 8-bit input: [v, w, x, y, bits consumed]",
               "first": 2773, "total_words": total, "words": [int(t) for t in words], "quads": quads},
              open(os.path.join(gold, "huff_table33.json"), "w"), indent=0)
    print("table 33 tree:", " ".join("%04x" % t for t in words))


if __name__ == "__main__":
    main()

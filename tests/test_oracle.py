"""The oracle (oracle/pdmp3_oracle.c) pinned against the reference.

* vs committed golden fixtures (tests/golden/, produced by the compiled
  reference via tools/make_golden.py) -- runs everywhere;
* vs oracle/_ref itself -- runs where it is built (this container, or a box
  that received the prebuilt .so).
Bar: bit-exact (0 LSB, 0 ulp).
"""
import os

import numpy as np
import pytest

import corpus
from conftest import C2_SEED
from util import sha, nch_of

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", list(corpus.ALL_CASES))
def test_oracle_matches_golden(oracle, name):
    """64 frames per case from the compiled reference: PCM of all frames by hash, head and last frame by value"""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    n = int(g["n_frames"][0])
    assert n >= 64
    sp, sd = corpus.case(name, n=n)
    assert [sha(sp), sha(sd)] == list(g["input_sha"]), "corpus generator drifted from the fixtures"
    pcm, stg = oracle.decode(sp, sd, stages=True)
    assert sha(pcm) == str(g["pcm_sha"][0])
    assert np.array_equal(pcm[:4], g["pcm_head"]) and np.array_equal(pcm[-1:], g["pcm_last"])
    nch = nch_of(sd)
    assert [sha(stg[:, :, :nch, k]) for k in range(4)] == list(g["stage_sha"])
    assert np.array_equal(stg[:2, :, :nch, 3].view(np.uint32), g["stage3_head"][:, :, :nch].view(np.uint32))


def test_oracle_matches_golden_c2(oracle):
    g = np.load(os.path.join(GOLD, "c2_prefix.npz"))
    sp, sd = oracle.generate(C2_SEED, 0, 2048)
    assert [sha(sp), sha(sd)] == list(g["input_sha"])
    pcm = oracle.decode(sp, sd)
    assert np.array_equal(pcm[:32], g["pcm_head"])
    assert sha(pcm) == str(g["pcm_sha_2048"][0])


@pytest.mark.parametrize("name", list(corpus.ALL_CASES))
def test_oracle_bit_exact_vs_reference(oracle, reference, name):
    sp, sd = corpus.case(name, n=10, seed=1234)
    p1, s1 = oracle.decode(sp, sd, stages=True)
    p2, s2 = reference.decode(sp, sd, stages=True)
    assert np.array_equal(p1, p2)
    assert np.array_equal(s1.view(np.uint32), s2.view(np.uint32))


def test_oracle_generator_shards(oracle):
    """counter-based: any sub-range equals the same range of a longer run."""
    sp, sd = oracle.generate(C2_SEED, 0, 40)
    sp2, sd2 = oracle.generate(C2_SEED, 17, 9)
    assert np.array_equal(sp[17:26], sp2)
    assert np.array_equal(sd[17:26].view(np.uint8), sd2.view(np.uint8))


def test_oracle_empty(oracle):
    sp, sd = oracle.generate(C2_SEED, 0, 1)
    pcm = oracle.decode(sp[:0], sd[:0])
    assert pcm.shape == (0, 2304)

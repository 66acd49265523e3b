"""Bitstream front end of the oracle (oracle/pdmp3_oracle_stream.c) pinned
against the reference: the real 9 KB clip (22 frames, block types 0-3, bit
reservoir up to 511 bytes, stale count1 (H6), count1table_select=1 (H1)).

The golden md5 is the reference's own output for this clip.
Clip provenance: MathJax `extensions/a11y/invalid_keypress.mp3` (Apache-2.0),
shipped inside the python `kaleido` package of this image.
"""
import hashlib
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CLIP = os.path.join(GOLD, "clip_invalid_keypress.mp3")
CLIP_PCM_MD5 = "691b76164c105c1f1edc5f2fe7bb7c8f"   # reference CLI output, 20 frames, 92160 bytes


def test_clip_pcm_md5(oracle):
    mp3 = open(CLIP, "rb").read()
    pcm = oracle.decode_buffer_like_cli(mp3)
    assert len(pcm) == 92160                      # 22 frames in, 20 out (H10 tail drop)
    assert hashlib.md5(pcm).hexdigest() == CLIP_PCM_MD5


def test_clip_records_feed_transform_oracle(oracle):
    """tap -> records -> orc_decode_frames reproduces the stream PCM: the record
    boundary (include/pdmp3_hip.h) carries everything the transforms need."""
    mp3 = open(CLIP, "rb").read()
    pcm, sp, sd = oracle.decode_buffer_like_cli(mp3, tap_frames=64)
    assert sp.shape[0] == 20
    again = oracle.decode(sp, sd)
    assert again.tobytes() == pcm
    # the clip exercises what SURVEY says it does
    bt = (sd["flags"] >> 3) & 3
    assert set(np.unique(bt)) == {0, 1, 2, 3}
    assert (sd["count1"] == 576).sum() >= 2       # H1: table 33 garbage runs to 576


def test_clip_vs_reference(oracle, reference):
    mp3 = open(CLIP, "rb").read()
    p_ref, sp_r, sd_r = reference.decode_buffer_like_cli(mp3, tap_frames=64)
    p_orc, sp_o, sd_o = oracle.decode_buffer_like_cli(mp3, tap_frames=64)
    assert p_ref == p_orc
    assert np.array_equal(sp_r, sp_o)
    assert np.array_equal(sd_r.view(np.uint8), sd_o.view(np.uint8))
    assert reference.decode_buffer_like_cli(mp3) == p_ref   # through the real pdmp3_read

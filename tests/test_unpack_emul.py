"""Device-side main-data decoding (pdmp3_amd/csrc/unpack_core.h: scalefactors + Huffman per granule-channel,
frame-to-frame merge) compiled for the host (tests/host_emul) against the product's host stage, which is itself
pinned bit-for-bit to the oracle / reference (test_host_stage.py, test_bulk_host.py): from the same side info and
reservoir snapshots both must build IDENTICAL gc records, for any split of the stream into windows.  No GPU."""
import ctypes as C

import numpy as np
import pytest

from test_bulk_host import _streams
from test_host_stage import _records_equal
from pdmp3_amd.packer import packer


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def emul_unpack(emul, bits, res, cuts=None):
    from pdmp3_amd.hip import SIDE_DTYPE
    n = bits.shape[0]
    sp = np.zeros((n, 2, 2, 576), np.int16)
    sd = np.zeros((n, 2, 2), SIDE_DTYPE)
    state = np.zeros(256, np.uint16)
    cuts = [0, n] if cuts is None else cuts
    for a, b in zip(cuts[:-1], cuts[1:]):
        if b > a:
            rc = emul.emul_unpack_frames(_p(bits[a:b]), _p(res[a:b]), b - a, _p(state), _p(sp[a:b]), _p(sd[a:b]))
            assert 0 < rc <= 9728          # kHuffLutMax (unpack_core.h)
    return sp, sd


@pytest.fixture(scope="module")
def streams():
    s = _streams()
    # corrupt main data: part2_3_length / big_values that overrun the reservoir (slow-path windows, H8, count1 wrap)
    rs = np.random.RandomState(9)
    body = bytearray(packer.generate(n_frames=60, seed=41, bitrate_index=11, block_pct=(40, 10, 40, 10)))
    for k in rs.randint(200, len(body) - 200, size=120):
        body[k] ^= 1 << int(rs.randint(0, 8))
    s["bit_flips"] = bytes(body)
    return s


def test_unpack_matches_host_stage(emul, streams):
    from pdmp3_amd import api
    b = api.BulkDecoder(threads=2, window_frames=64, parse_only=True)
    try:
        for name, mp3 in streams.items():
            sp_h, sd_h, nbytes = b.parse(mp3)
            bits, res, nbytes2 = api.parse_bits(mp3)
            assert bits.shape[0] == sp_h.shape[0] and nbytes == nbytes2, name
            if not bits.shape[0]:
                continue
            got = emul_unpack(emul, bits, res)
            assert np.array_equal(got[0], sp_h), name
            assert np.array_equal(got[1].view(np.uint8), sd_h.view(np.uint8)), name
            n = bits.shape[0]
            for cuts in ([0, 1, n], [0, n // 3, n // 2, n - 1, n], list(range(0, n, 7)) + [n]):
                assert _records_equal(emul_unpack(emul, bits, res, cuts), got), (name, cuts)
    finally:
        b.close()


def test_unpack_table_blob_fits_lds(emul):
    bits = np.zeros(1, dtype=np.dtype([("b", "u1", (80,))]))
    res = np.zeros((1, 2064), np.uint8)
    from pdmp3_amd.hip import SIDE_DTYPE
    n_lut = emul.emul_unpack_frames(_p(bits), _p(res), 1, _p(np.zeros(256, np.uint16)), _p(np.zeros(2304, np.int16)),
                                    _p(np.zeros(4, SIDE_DTYPE)))
    assert 8000 < n_lut <= 9728           # first levels of the 18 books + the zero book, second levels, a leaf per short code word


@pytest.mark.parametrize("base", ["vbr_mixed", "mono_32k", "linbits_320k"])
def test_unpack_fuzz_matches_host_stage(emul, base):
    """random corruption (bit flips, byte splats, truncation): whatever the host stage builds from a broken stream --
    region overruns (H7, H8), part2_3_length running off the reservoir, count1 wrapping below zero -- the device
    logic builds the same records, with random window cuts.  (ASan/UBSan builds of both were fuzzed the same way.)"""
    from pdmp3_amd import api
    kw = {"vbr_mixed": dict(n_frames=90, seed=51, vbr=True, block_pct=(40, 10, 40, 10), mixed_pct=50),
          "mono_32k": dict(n_frames=90, seed=52, sfreq=2, mode=3, bitrate_index=7),
          "linbits_320k": dict(n_frames=70, seed=53, mode=1, mode_ext=2, bitrate_index=14, big_pct=200, gain=(100, 140))}[base]
    orig = np.frombuffer(packer.generate(**kw), dtype=np.uint8)
    rs = np.random.RandomState({"vbr_mixed": 7, "mono_32k": 8, "linbits_320k": 9}[base])
    b = api.BulkDecoder(threads=2, window_frames=32, parse_only=True)
    frames = 0
    try:
        for it in range(25):
            m = orig.copy()
            kind = it % 3
            for p in rs.randint(0, len(m), size=1 + rs.randint(0, 6 if kind == 0 else 150)):
                m[p] = rs.randint(0, 256) if kind == 2 else m[p] ^ (1 << rs.randint(0, 8))
            if it % 5 == 4:
                m = m[:rs.randint(1, len(m))]
            m = np.ascontiguousarray(m)
            try:
                sp_h, sd_h, _ = b.parse(m)
            except api.RingReplay:
                continue
            bits, res, _ = api.parse_bits(m)
            n = bits.shape[0]
            assert n == sp_h.shape[0]
            if not n:
                continue
            cuts = sorted(set([0, n] + rs.randint(0, n, size=rs.randint(0, 5)).tolist()))
            got = emul_unpack(emul, bits, res, cuts)
            assert np.array_equal(got[0], sp_h), (base, it)
            assert np.array_equal(got[1].view(np.uint8), sd_h.view(np.uint8)), (base, it)
            frames += n
    finally:
        b.close()
    assert frames > 500


def test_newstream_flag_restarts_the_parse_state(emul, streams):
    """PDMP3_FR_NEWSTREAM on a stream's first frame: streams unpacked back to back in ONE pass, with whatever
    scalefactor / count1 state the previous one left, come out as if each had been unpacked alone"""
    from pdmp3_amd import api
    names = ["short_heavy_dual", "mono_32k_96", "cbr128_js_441", "vbr_48k_stereo_crc_tab33", "stereo_then_mono_then_stereo"]
    parts = [api.parse_bits(streams[k])[:2] for k in names]
    for bits, _ in parts:
        assert bits["frame"][0] & 0x80 and not (bits["frame"][1:] & 0x80).any()
    alone = [emul_unpack(emul, b, r) for b, r in parts]
    bits = np.concatenate([b for b, _ in parts])
    res = np.concatenate([r for _, r in parts])
    n = bits.shape[0]
    for cuts in ([0, n], [0, 7, n // 2, n - 3, n]):
        sp, sd = emul_unpack(emul, bits, res, cuts)
        assert np.array_equal(sp, np.concatenate([a[0] for a in alone]))
        assert np.array_equal(sd.view(np.uint8), np.concatenate([a[1] for a in alone]).view(np.uint8))
    assert not (sd["frame"] & 0x80).any()              # the flag stays out of the gc records


def _corrupt(rs, base, kind):
    m = np.frombuffer(base, dtype=np.uint8).copy()
    for p in rs.randint(0, len(m), size=1 + rs.randint(0, 4 if kind == 0 else 120)):
        m[p] = rs.randint(0, 256) if kind == 2 else m[p] ^ (1 << rs.randint(0, 8))
    return np.ascontiguousarray(m)


def test_pool_rows_are_the_snapshot_rows(emul):
    """the compact bits input (pdmp3_row_desc + pool): rows rebuilt by the device's rule (row_word, here on the host)
    are byte for byte the 2064-byte reservoir snapshots -- stale bytes beyond main_top included -- on clean streams
    (CBR, VBR with small frames: long look-backs), truncated ones and corrupted ones (underflows H9, resyncs)"""
    from pdmp3_amd import api
    rs = np.random.RandomState(11)
    bases = [packer.generate(n_frames=300, seed=21, sfreq=0, mode=1, mode_ext=2, bitrate_index=14),
             packer.generate(n_frames=400, seed=22, sfreq=2, mode=3, vbr=True, vbr_lo=1, vbr_hi=6),
             packer.generate(n_frames=250, seed=23, sfreq=1, mode=0, mode_ext=0, vbr=True, block_pct=(40, 10, 40, 10), crc=True),
             packer.generate(n_frames=200, seed=24, sfreq=0, mode=1, mode_ext=2, bitrate_index=2)]
    streams = [np.frombuffer(b, dtype=np.uint8) for b in bases] + [np.frombuffer(bases[0][:-700], dtype=np.uint8)]
    for it in range(60):
        streams.append(_corrupt(rs, bases[rs.randint(len(bases))], rs.randint(3)))
    # reservoir underflows on purpose (H9): main_data_begin = 511 in every 7th frame of streams with small frames
    for base in (packer.generate(n_frames=300, seed=31, sfreq=0, mode=1, mode_ext=2, bitrate_index=5),
                 packer.generate(n_frames=300, seed=32, sfreq=1, mode=3, bitrate_index=3)):
        m = np.frombuffer(base, dtype=np.uint8).copy()
        pos, k = 0, 0
        while pos + 6 < len(m):
            h = int.from_bytes(m[pos:pos + 4].tobytes(), "big")
            if k % 7 == 3:
                m[pos + 4] = 0xFF
                m[pos + 5] |= 0x80
            pos += 144 * [0, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320][(h >> 12) & 15] * 1000 // \
                [44100, 48000, 32000][(h >> 10) & 3] + ((h >> 9) & 1)
            k += 1
        streams.append(m)
    checked = explicit = segments = parsed = 0
    for m in streams:
        try:
            bits, res, _ = api.parse_bits(m)
        except (api.RingReplay, RuntimeError):
            continue
        bits2, desc, pool = api.parse_pool(m)
        assert len(bits2) == len(bits) and np.array_equal(bits2.view(np.uint8), bits.view(np.uint8))
        if len(bits) == 0:
            continue
        rows = np.zeros((len(bits), 2064), dtype=np.uint8)
        emul.emul_rows(desc.ctypes.data_as(C.c_void_p), pool.ctypes.data_as(C.c_void_p), len(bits), rows.ctypes.data_as(C.c_void_p))
        rows_w = np.zeros_like(rows)                        # (the rule word by word: what k_rows' 16-byte chunks must equal)
        emul.emul_rows_by_word(desc.ctypes.data_as(C.c_void_p), pool.ctypes.data_as(C.c_void_p), len(bits), rows_w.ctypes.data_as(C.c_void_p))
        assert np.array_equal(rows, rows_w)
        bad = np.nonzero((rows != res).any(axis=1))[0]
        assert bad.size == 0, (len(bits), bad[:5], desc[bad[:2]])
        checked += len(bits)
        parsed += 1
        explicit += int((desc["top"] == 2064).sum())
        segments += int((desc["back"] == 0).sum())
    # segments restarted by reservoir underflows, not only one per stream (a frame that is decoded after an irregular
    # fill -- H18, an explicit image -- needs a frame longer than 1152 bytes cut short: those streams take the ring-replay path)
    assert checked > 5000 and segments >= parsed + 8
    print("rows %d, segments %d, explicit images %d" % (checked, segments, explicit))


def test_pool_window_longer_than_the_link_fields(emul):
    """`back` / `up` of pdmp3_row_desc are 16 bits: a window of more than 65000 frames (only the parse hook makes one; the
    engine's windows are capped at 32768) starts a new segment, rows unchanged"""
    from pdmp3_amd import api
    mp3 = np.frombuffer(packer.generate(n_frames=67000, seed=5, sfreq=2, mode=3, bitrate_index=1), dtype=np.uint8)
    bits, res, _ = api.parse_bits(mp3)
    bits2, desc, pool = api.parse_pool(mp3)
    assert len(bits) > 66000 and np.array_equal(bits2.view(np.uint8), bits.view(np.uint8))
    assert int((desc["back"] == 0).sum()) == 2 and int(desc["back"].max()) == 65000
    rows = np.zeros((len(bits), 2064), dtype=np.uint8)
    emul.emul_rows(desc.ctypes.data_as(C.c_void_p), pool.ctypes.data_as(C.c_void_p), len(bits), rows.ctypes.data_as(C.c_void_p))
    assert np.array_equal(rows, res)


def test_merge_by_blocks_over_long_windows(emul):
    """k_merge_outcome / k_merge_apply cut a window into blocks of 32 frames, compose the outcomes of eight blocks and chain both (unpack_core.h
    merge_blocks; emul_unpack_frames runs it beside merge_slot and returns -2 / -3 when a record byte or the carried state
    differs): windows of a dozen blocks with scfsi copies, short and mixed blocks, a mono part (channel 1's slots keep
    their values through it) and a second stream behind the first (PDMP3_FR_NEWSTREAM in the middle of a block)"""
    from pdmp3_amd import api
    a = packer.generate(n_frames=420, seed=61, vbr=True, block_pct=(55, 10, 25, 10), mixed_pct=40, mode=1, mode_ext=2)
    m = packer.generate(n_frames=150, seed=62, mode=3, bitrate_index=7)
    c = packer.generate(n_frames=333, seed=63, bitrate_index=11, block_pct=(97, 1, 1, 1))
    parts = [api.parse_bits(x)[:2] for x in (a + m, c)]
    bits = np.concatenate([b for b, _ in parts])
    res = np.concatenate([r for _, r in parts])
    n = bits.shape[0]
    assert n > 850 and int(((bits["frame"] & 0x80) != 0).sum()) == 2
    whole = emul_unpack(emul, bits, res)
    b = api.BulkDecoder(threads=2, window_frames=256, parse_only=True)
    try:
        sd_h = np.concatenate([b.parse(x)[1] for x in (a + m, c)])
    finally:
        b.close()
    assert np.array_equal(whole[1].view(np.uint8), sd_h.view(np.uint8))
    for cuts in ([0, 64, 128, n], [0, 63, 129, 500, n], [0, 1, n - 1, n]):
        assert _records_equal(emul_unpack(emul, bits, res, cuts), whole), cuts


@pytest.mark.parametrize("p_set,p_copy,p_mono,p_new", [(90, 30, 0, 0), (50, 60, 10, 5), (10, 90, 30, 20), (3, 50, 0, 1), (100, 0, 100, 0)])
def test_merge_by_blocks_on_random_merge_input(emul, p_set, p_copy, p_mono, p_new):
    """the two merge kernels' host form (merge_blocks: per-lane descriptors, batched walks, outcomes of blocks and super-blocks,
    specialised per wave) against the rule (merge_slot) on random merge input -- set and copy bits at every density (a slot that
    is written in 3 % of the frames is carried over dozens of blocks, one whose twin is never written before a copy takes the
    incoming twin's value), mono frames, new streams in the middle of a block, the ISO switches -- for windows of every
    length around the block and super-block sizes"""
    emul.emul_merge_fuzz.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    for k, n in enumerate([1, 31, 32, 33, 63, 64, 255, 256, 257, 300, 511, 1024, 1400, 2500]):
        rc = emul.emul_merge_fuzz(0xA5A5 + 977 * k + p_set, n, p_set, p_copy, p_mono, p_new)
        assert rc == 0, (n, rc)

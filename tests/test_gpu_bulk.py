"""Bulk pipeline end to end on the GPU box (include/pdmp3_bulk.h): threaded
host Huffman -> pinned 3-deep slots -> pdmp3_hip_stream_submit -> PCM, against
(a) the oracle's CLI-loop decode of the same bytes (<= 1 LSB; it is the
bit-exact restatement of the reference), and (b) the product's own streaming
API (pdmp3_feed / pdmp3_read), which must give the IDENTICAL bytes: batching,
slot rotation and chunking change nothing."""
import os

import numpy as np
import pytest

from tools.packer import packer
from util import assert_pcm_close
from test_bulk_host import _streams

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def streams():
    return _streams()


@pytest.mark.parametrize("threads,window", [(4, 2048), (3, 5), (8, 32)])
def test_bulk_decode_matches_oracle_and_streaming_api(oracle, streams, threads, window):
    from pdmp3_amd import api
    b = api.BulkDecoder(threads=threads, window_frames=window)
    try:
        for name, mp3 in streams.items():
            want = np.frombuffer(oracle.decode_buffer_like_cli(mp3), dtype=np.int16)
            got = b.decode(mp3)
            assert got.shape == want.shape, name
            if got.size:
                assert_pcm_close(got, want, 1, name)
            via_api = np.frombuffer(api.decode_like_cli(mp3), dtype=np.int16)
            assert np.array_equal(got, via_api), name
    finally:
        b.close()


def test_bulk_long_vbr_stream(oracle):
    """many windows in flight, mixed block types, VBR frame sizes"""
    from pdmp3_amd import api
    mp3 = packer.generate(n_frames=6000, seed=78, vbr=True, block_pct=(40, 10, 40, 10))
    want = np.frombuffer(oracle.decode_buffer_like_cli(mp3), dtype=np.int16)
    b = api.BulkDecoder(threads=8, window_frames=500)
    try:
        got = b.decode(mp3)
        again = b.decode(mp3)
    finally:
        b.close()
    assert got.shape == want.shape and got.size >= 5990 * 2304
    assert_pcm_close(got, want, 1, "vbr6000")
    assert np.array_equal(got, again)                   # a reused decoder starts fresh


def test_bulk_output_smaller_than_stream():
    """pcm_cap below the stream's size: the prefix is written, the full size returned"""
    from pdmp3_amd import api
    mp3 = packer.generate(n_frames=50, seed=79, bitrate_index=9)
    b = api.BulkDecoder(threads=2, window_frames=8)
    try:
        full = b.decode(mp3)
        small = np.zeros(10000, dtype=np.int16)
        total, rate, ch = b.decode_into(mp3, small)
    finally:
        b.close()
    assert total == full.nbytes and (rate, ch) == (44100, 2)
    assert np.array_equal(small, full[:10000])


def test_slots_decode_in_submit_order(engine):
    """pdmp3_hip_stream_submit on rotating slots == one pdmp3_hip_decode_frames over the whole stream"""
    import ctypes as C
    import torch
    from pdmp3_amd import hip
    lib = hip.load_library()
    n, per = 96, 16
    sp, sd = hip.host_generate(0x5EED0000C2, 0, n)
    _, _, dpcm = engine.alloc_frames(n)
    dsp, dsd = engine.upload(sp, sd)
    engine.decode(dsp, dsd, dpcm, state=engine.new_state())
    torch.cuda.synchronize()
    want = dpcm.cpu().numpy()
    vp = C.c_void_p
    lib.pdmp3_hip_stream_create_slots.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp)]
    for f in ("pdmp3_hip_stream_slot_spectra", "pdmp3_hip_stream_slot_side", "pdmp3_hip_stream_slot_pcm"):
        getattr(lib, f).restype = vp
        getattr(lib, f).argtypes = [vp, C.c_int]
    lib.pdmp3_hip_stream_submit.argtypes = [vp, C.c_int, C.c_int]
    lib.pdmp3_hip_stream_wait.argtypes = [vp, C.c_int]
    lib.pdmp3_hip_stream_destroy.argtypes = [vp]
    hs = vp()
    assert lib.pdmp3_hip_stream_create_slots(engine.h, per, 3, C.byref(hs)) == 0
    got = np.zeros((n, 2304), dtype=np.int16)
    try:
        for w in range(n // per):
            slot = w % 3
            if w >= 3:
                assert lib.pdmp3_hip_stream_wait(hs, slot) == 0
                C.memmove(got[(w - 3) * per:].ctypes.data, lib.pdmp3_hip_stream_slot_pcm(hs, slot), per * 4608)
            C.memmove(lib.pdmp3_hip_stream_slot_spectra(hs, slot), sp[w * per:].ctypes.data, per * 4608)
            C.memmove(lib.pdmp3_hip_stream_slot_side(hs, slot), sd[w * per:].ctypes.data, per * 512)
            assert lib.pdmp3_hip_stream_submit(hs, slot, per) == 0
            assert lib.pdmp3_hip_stream_submit(hs, slot, per) != 0        # in flight: refused
        for w in range(n // per - 3, n // per):
            assert lib.pdmp3_hip_stream_wait(hs, w % 3) == 0
            C.memmove(got[w * per:].ctypes.data, lib.pdmp3_hip_stream_slot_pcm(hs, w % 3), per * 4608)
    finally:
        lib.pdmp3_hip_stream_destroy(hs)
    assert np.array_equal(got.reshape(want.shape), want)

"""Bulk pipeline end to end on the GPU box (include/pdmp3_bulk.h): threaded
host Huffman -> pinned 3-deep slots -> pdmp3_hip_stream_submit -> PCM, against
(a) the oracle's CLI-loop decode of the same bytes (<= 1 LSB; it is the
bit-exact restatement of the reference), and (b) the product's own streaming
API (pdmp3_feed / pdmp3_read), which must give the IDENTICAL bytes: batching,
slot rotation and chunking change nothing."""
import os

import numpy as np
import pytest

from pdmp3_amd.packer import packer
from util import assert_pcm_close
from test_bulk_host import _streams

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def streams():
    return _streams()


@pytest.mark.parametrize("host_huffman", [False, True])
@pytest.mark.parametrize("threads,window", [(4, 2048), (3, 5), (8, 32), (1, 1)])
def test_bulk_decode_matches_oracle_and_streaming_api(oracle, streams, threads, window, host_huffman):
    """host_huffman=False: scalefactors + Huffman + state merge on the device (submit_bits); True: on the host pool"""
    from pdmp3_amd import api
    b = api.BulkDecoder(threads=threads, window_frames=window, host_huffman=host_huffman)
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
    bh = api.BulkDecoder(threads=8, window_frames=300, host_huffman=True)
    try:
        got = b.decode(mp3)
        again = b.decode(mp3)
        via_host_huffman = bh.decode(mp3)
    finally:
        b.close()
        bh.close()
    assert got.shape == want.shape and got.size >= 5990 * 2304
    assert_pcm_close(got, want, 1, "vbr6000")
    assert np.array_equal(got, again)                   # a reused decoder starts fresh
    assert np.array_equal(got, via_host_huffman)        # device Huffman == host Huffman, bit for bit


def test_windows_larger_than_a_merge_step():
    """windows of 5000, 7001 and 14999 frames: k_merge_apply walks the rows of outcomes in front of a block 40 at a time (a
    window of 14999 frames has 58 super-blocks: two goes for its last blocks; 31 + 7 rows at most up to 8192 frames), the last
    block of a window is a partial one (7001 = 218 x 32 + 25), k_unpack gets 313 / 438 / 938 workgroups -- same PCM, bit for
    bit, as with the default window and as with host Huffman"""
    from pdmp3_amd import api
    mp3 = packer.generate(n_frames=15000, seed=79, vbr=True, block_pct=(40, 10, 40, 10), mixed_pct=30)
    outs = []
    for window, hh in ((2048, False), (5000, False), (7001, False), (14999, False), (4096, True)):
        b = api.BulkDecoder(threads=4, window_frames=window, host_huffman=hh)
        try:
            outs.append(b.decode(mp3))
        finally:
            b.close()
    assert outs[0].size >= 14990 * 2304
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])


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


def test_device_unpack_builds_the_host_stage_records(engine, streams):
    """pdmp3_hip_stream_submit_bits: the gc records the device builds from side info + reservoir snapshots are the
    host stage's, byte for byte (which test_host_stage.py pins to the oracle / reference), across window cuts"""
    import ctypes as C
    from pdmp3_amd import api, hip
    lib = hip.load_library()
    vp = C.c_void_p
    lib.pdmp3_hip_stream_create_slots.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp)]
    for f in ("pdmp3_hip_stream_slot_bits", "pdmp3_hip_stream_slot_reservoir"):
        getattr(lib, f).restype = vp
        getattr(lib, f).argtypes = [vp, C.c_int]
    lib.pdmp3_hip_stream_submit_bits.argtypes = [vp, C.c_int, C.c_int]
    lib.pdmp3_hip_stream_wait.argtypes = [vp, C.c_int]
    lib.pdmp3_hip_stream_reset.argtypes = [vp]
    lib.pdmp3_hip_stream_fetch_records.argtypes = [vp, C.c_int, C.c_int, vp, vp]
    lib.pdmp3_hip_stream_destroy.argtypes = [vp]
    per = 37
    hs = vp()
    assert lib.pdmp3_hip_stream_create_slots(engine.h, per, 2, C.byref(hs)) == 0
    host = api.BulkDecoder(threads=2, window_frames=64, parse_only=True)
    try:
        for name, mp3 in streams.items():
            sp_h, sd_h, _ = host.parse(mp3)
            bits, res, _ = api.parse_bits(mp3)
            n = bits.shape[0]
            assert n == sp_h.shape[0], name
            assert lib.pdmp3_hip_stream_reset(hs) == 0
            sp = np.zeros_like(sp_h)
            sd = np.zeros_like(sd_h)
            for w, a in enumerate(range(0, n, per)):
                k = min(per, n - a)
                slot = w % 2
                C.memmove(lib.pdmp3_hip_stream_slot_bits(hs, slot), bits[a:].ctypes.data, k * 80)
                C.memmove(lib.pdmp3_hip_stream_slot_reservoir(hs, slot), res[a:].ctypes.data, k * 2064)
                assert lib.pdmp3_hip_stream_submit_bits(hs, slot, k) == 0
                assert lib.pdmp3_hip_stream_wait(hs, slot) == 0
                assert lib.pdmp3_hip_stream_fetch_records(hs, slot, k, sp[a:].ctypes.data, sd[a:].ctypes.data) == 0
            assert np.array_equal(sp, sp_h), name
            assert np.array_equal(sd.view(np.uint8), sd_h.view(np.uint8)), name
    finally:
        host.close()
        lib.pdmp3_hip_stream_destroy(hs)


def test_bulk_decoder_on_named_device(oracle):
    """pdmp3_amd_bulk_new_on: the decoder's engine context is per HIP device (device 0 here; a multi-GPU corpus run
    gives each host thread a decoder on its own GPU, tools/bulk_bench.py --c4 N --gpus G)"""
    from pdmp3_amd import api
    mp3 = packer.generate(n_frames=80, seed=91, bitrate_index=11)
    want = np.frombuffer(oracle.decode_buffer_like_cli(mp3), dtype=np.int16)
    b = api.BulkDecoder(threads=2, window_frames=32, device=0)
    try:
        got = b.decode(mp3)
    finally:
        b.close()
    assert got.shape == want.shape
    assert_pcm_close(got, want, 1, "device 0")
    with pytest.raises(RuntimeError):
        api.BulkDecoder(threads=1, device=99)        # no such device: fails loudly, no fallback


@pytest.mark.parametrize("window", [2048, 16])
def test_streams_back_to_back_without_draining(oracle, streams, window):
    """pdmp3_amd_bulk_decode_async: streams follow each other through the pipeline with no host-side reset between
    them (PDMP3_FR_RESET / PDMP3_FR_NEWSTREAM on each first frame): every one must come out as if decoded alone --
    also mono after stereo, different sampling rates, empty streams in between"""
    from pdmp3_amd import api
    names = list(streams)
    order = names + names[::-1]
    b = api.BulkDecoder(threads=3, window_frames=window)
    alone = api.BulkDecoder(threads=2, window_frames=64)
    try:
        got = b.decode_many([streams[k] for k in order])
        for k, g in zip(order, got):
            want = np.frombuffer(oracle.decode_buffer_like_cli(streams[k]), dtype=np.int16)
            assert g.shape == want.shape, k
            if g.size:
                assert_pcm_close(g, want, 1, k)
            assert np.array_equal(g, alone.decode(streams[k])), k       # bit-identical to a decoder of its own
    finally:
        b.close()
        alone.close()


def test_corrupted_streams_device_equals_host_huffman():
    """bit flips, byte splats, truncation: whatever a broken stream makes of the host stage, the device-Huffman path
    (k_unpack / k_merge, rows and lines in LDS) produces the very same PCM -- 120 mutated streams back to back"""
    from pdmp3_amd import api
    rs = np.random.RandomState(21)
    bases = [np.frombuffer(packer.generate(n_frames=90, seed=51, vbr=True, block_pct=(40, 10, 40, 10), mixed_pct=50), dtype=np.uint8),
             np.frombuffer(packer.generate(n_frames=90, seed=52, sfreq=2, mode=3, bitrate_index=7), dtype=np.uint8),
             np.frombuffer(packer.generate(n_frames=70, seed=53, mode=1, mode_ext=2, bitrate_index=14, big_pct=200, gain=(100, 140)), dtype=np.uint8)]
    muts = []
    for it in range(120):
        m = bases[it % 3].copy()
        kind = (it // 3) % 3
        for p in rs.randint(0, len(m), size=1 + rs.randint(0, 6 if kind == 0 else 150)):
            m[p] = rs.randint(0, 256) if kind == 2 else m[p] ^ (1 << rs.randint(0, 8))
        if it % 7 == 6:
            m = m[:rs.randint(1, len(m))]
        m = np.ascontiguousarray(m)
        try:
            api.scan_buffer(m)
        except api.RingReplay:
            continue
        muts.append(m)
    dev = api.BulkDecoder(threads=2, window_frames=40)
    host = api.BulkDecoder(threads=4, window_frames=24, host_huffman=True)
    try:
        got = dev.decode_many(muts)
        frames = 0
        for m, g in zip(muts, got):
            assert np.array_equal(g, host.decode(m))
            frames += g.size // 1152
    finally:
        dev.close()
        host.close()
    assert len(muts) > 100 and frames > 5000


def test_decoder_survives_a_declined_stream(oracle):
    """a stream the whole-stream path declines (PDMP3_BULK_REPLAY, detected after part of it was already queued)
    leaves the decoder usable: the streams before and after it decode as if alone"""
    from pdmp3_amd import api
    good = packer.generate(n_frames=120, seed=61, bitrate_index=11, block_pct=(40, 10, 40, 10))
    replay = packer.generate(n_frames=4147, seed=435, sfreq=2, mode=0, mode_ext=0, vbr=True, vbr_lo=4, vbr_hi=13, bitrate_index=12,
                             block_pct=(40, 20, 20, 20), mixed_pct=50)
    want = np.frombuffer(oracle.decode_buffer_like_cli(good), dtype=np.int16)
    b = api.BulkDecoder(threads=2, window_frames=256)
    try:
        first = b.decode(good)
        junk = np.zeros(4147 * 2304, dtype=np.int16)
        with pytest.raises(api.RingReplay):
            b.decode_into(replay, junk)
        with pytest.raises(api.RingReplay):
            b.decode_into_async(replay, junk)
        again = b.decode_many([good, good])
    finally:
        b.close()
    assert_pcm_close(first, want, 1, "before")
    assert np.array_equal(again[0], first) and np.array_equal(again[1], first)


@pytest.mark.parametrize("host_huffman", [False, True])
def test_pinned_destination_is_written_directly(streams, host_huffman):
    """PCM buffers from pdmp3_amd_pcm_alloc: windows are downloaded straight into them (mono packed by a 2-D copy,
    mixed windows still through staging) -- same bytes as into ordinary memory"""
    from pdmp3_amd import api
    b = api.BulkDecoder(threads=2, window_frames=16, host_huffman=host_huffman)
    ref = api.BulkDecoder(threads=2, window_frames=64)
    bufs = []
    try:
        names = [k for k in streams if len(streams[k])]
        want = {k: ref.decode(streams[k]) for k in names}
        for k in names:                                    # synchronous
            p = api.PinnedPCM(want[k].size + 64)
            bufs.append(p)
            p.array[:] = 0x5A5A
            total, _, _ = b.decode_into(streams[k], p.array)
            assert total == want[k].nbytes, k
            assert np.array_equal(p.array[:want[k].size], want[k]), k
            assert (p.array[want[k].size:] == 0x5A5A).all(), k          # nothing written past the stream's PCM
        if not host_huffman:                               # back to back into one big pinned buffer
            big = api.PinnedPCM(sum(w.size for w in want.values()))
            bufs.append(big)
            off = 0
            for k in names:
                b.decode_into_async(streams[k], big.array[off:off + max(want[k].size, 1)])
                off += want[k].size
            b.wait()
            off = 0
            for k in names:
                assert np.array_equal(big.array[off:off + want[k].size], want[k]), k
                off += want[k].size
    finally:
        b.close()
        ref.close()
        for p in bufs:
            p.free()


@pytest.mark.parametrize("host_huffman", [False, True])
def test_pcm_stays_on_the_device(streams, host_huffman):
    """a device pointer as destination (here a torch tensor): bitstream in host memory -> PCM in HBM, nothing comes back"""
    import torch
    from pdmp3_amd import api
    b = api.BulkDecoder(threads=2, window_frames=32, host_huffman=host_huffman)
    ref = api.BulkDecoder(threads=2, window_frames=64)
    try:
        for k in ("cbr320_js_441", "mono_32k_96", "vbr_48k_stereo_crc_tab33", "short_heavy_dual"):
            want = ref.decode(streams[k])
            out = torch.full((want.size + 32,), 0x5A5A, dtype=torch.int16, device="cuda")
            total, rate, ch = b.decode_into_device(streams[k], out)
            torch.cuda.synchronize()
            got = out.cpu().numpy()
            assert total == want.nbytes, k
            assert np.array_equal(got[:want.size], want), k
            assert (got[want.size:] == 0x5A5A).all(), k
        outs = []
        for k in ("cbr128_js_441", "js_32k_256"):          # queued back to back
            want = ref.decode(streams[k])
            o = torch.zeros(want.size, dtype=torch.int16, device="cuda")
            torch.cuda.synchronize()                        # (the fill must be through: the decoder's streams do not wait for torch's)
            b.decode_into_device(streams[k], o, wait=False)
            outs.append((k, o, want))
        b.wait()
        torch.cuda.synchronize()
        for k, o, want in outs:
            assert np.array_equal(o.cpu().numpy(), want), k
    finally:
        b.close()
        ref.close()


@pytest.mark.parametrize("host_huffman", [False, True])
def test_device_destination_takes_mixed_windows_and_clipped_tails(host_huffman):
    """windows that cannot be downloaded straight into a DEVICE destination -- mono and stereo frames in one window,
    or a destination that ends inside a window -- are staged and copied, never dropped"""
    import torch
    from pdmp3_amd import api
    parts = [dict(n_frames=9, seed=31, bitrate_index=9), dict(n_frames=11, seed=32, mode=3, bitrate_index=7),
             dict(n_frames=2, seed=33, mode=1, mode_ext=2, bitrate_index=11, block_pct=(10, 10, 70, 10)),
             dict(n_frames=27, seed=34, mode=3, bitrate_index=7),
             dict(n_frames=40, seed=35, mode=1, mode_ext=2, bitrate_index=11, block_pct=(10, 10, 70, 10))]
    mp3 = b"".join(packer.generate(**p) for p in parts)
    b = api.BulkDecoder(threads=2, window_frames=16, host_huffman=host_huffman)
    ref = api.BulkDecoder(threads=2, window_frames=64)
    try:
        want = ref.decode(mp3)
        out = torch.full((want.size + 32,), 0x5A5A, dtype=torch.int16, device="cuda")
        total, rate, ch = b.decode_into_device(mp3, out)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert total == want.nbytes
        assert np.array_equal(got[:want.size], want)
        assert (got[want.size:] == 0x5A5A).all()
        # a destination that ends in the middle of a window (and of a frame): the prefix that fits is delivered
        short = want.size - 16 * 2304 - 1000
        out = torch.full((short + 64,), 0x5A5A, dtype=torch.int16, device="cuda")
        b.decode_into_device(mp3, out[:short])
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert np.array_equal(got[:short], want[:short])
        assert (got[short:] == 0x5A5A).all()
    finally:
        b.close()
        ref.close()


def test_corrupted_streams_pcm_against_the_oracle(oracle):
    """corrupted streams (without H8 frames, which the oracle cannot be given) end to end against the oracle's
    transforms: +-1 LSB wherever the signal is within a few times full scale; where a flipped global_gain drives the
    synthesis to 100x .. 4000x full scale the bar is the north-star float tolerance, 1e-5 of that amplitude"""
    from pdmp3_amd import api
    from util import pcm_tolerance_scaled
    rs = np.random.RandomState(777)
    bases = [np.frombuffer(packer.generate(n_frames=60, seed=500 + k, **kw), dtype=np.uint8) for k, kw in enumerate([
        dict(vbr=True, block_pct=(40, 10, 40, 10), mixed_pct=50), dict(mode=1, mode_ext=2, bitrate_index=14, big_pct=200, gain=(100, 140)),
        dict(sfreq=1, mode=0, mode_ext=0, crc=True, table33_pct=20), dict(mode=2, bitrate_index=12, block_pct=(10, 10, 70, 10))])]
    dev = api.BulkDecoder(threads=2, window_frames=48)
    streams = loud = 0
    try:
        for it in range(150):
            m = bases[rs.randint(len(bases))].copy()
            kind = rs.randint(3)
            for p in rs.randint(0, len(m), size=1 + rs.randint(0, 4 if kind == 0 else 200)):
                m[p] = rs.randint(0, 256) if kind == 2 else m[p] ^ (1 << rs.randint(0, 8))
            m = np.ascontiguousarray(m)
            try:
                api.scan_buffer(m)
            except api.RingReplay:
                continue
            bits, _, _ = api.parse_bits(m)
            if (bits["gc"]["big_values"] > 288).any() or (((bits["frame"] >> 2) & 3) == 3).any():
                continue                                   # H8, or a flipped header bit made a frame mono (layout below is stereo)
            sp, sd = api.parse_like_cli(m.tobytes(), 200)
            got = dev.decode(m)
            n = got.size // 2304
            if n == 0:
                continue
            want, st = oracle.decode(sp[:n], sd[:n], stages=True)
            got = got.reshape(n, 2304)
            d = np.abs(got.astype(np.int32) - want.astype(np.int32)).max(axis=1)
            amp = np.abs(st[:, :, :, 3]).reshape(n, -1).max(axis=1)
            hist = np.maximum.reduce([amp, np.roll(amp, 1), np.roll(amp, 2)])      # the polyphase history reaches back
            hist[:2] = np.maximum(hist[:2], amp[:2].max())
            quiet = hist < 1.0
            assert (d[quiet] <= 1).all(), (it, np.nonzero(quiet & (d > 1))[0][:4])
            assert d.max() <= pcm_tolerance_scaled(st[:, :, :, 3]), it
            loud += int((~quiet).any())
            streams += 1
    finally:
        dev.close()
    assert streams > 80 and loud > 0


def test_async_decode_is_done_with_the_stream_when_it_returns(streams):
    """pdmp3_amd_bulk_decode_async's contract: `mp3` may be released when the call returns.  The scanning thread only
    notes where the frames' main data lies in the caller's buffer and the submitter thread copies it later -- so the
    call must not return before that has happened: overwrite every input right after its call, wait at the end"""
    from pdmp3_amd import api
    names = [k for k in streams if len(streams[k]) > 2000]
    ref = api.BulkDecoder(threads=2, window_frames=64)
    b = api.BulkDecoder(threads=2, window_frames=32)
    try:
        want = {k: ref.decode(streams[k]) for k in names}
        outs = {k: np.zeros(max(want[k].size, 1), dtype=np.int16) for k in names}
        for k in names:
            buf = np.frombuffer(streams[k], dtype=np.uint8).copy()
            b.decode_into_async(buf, outs[k])
            buf[:] = 0xA5                                   # the caller's buffer is gone
        b.wait()
        for k in names:
            assert np.array_equal(outs[k][:want[k].size], want[k]), k
    finally:
        b.close()
        ref.close()


def test_compact_upload_equals_snapshot_rows(streams, monkeypatch):
    """the two forms of the device-Huffman input -- main data once in a pool + row descriptors (k_rows rebuilds the
    reservoir rows), and a 2064-byte snapshot per frame (PDMP3_BULK_SNAPSHOT_ROWS=1) -- give the same PCM"""
    from pdmp3_amd import api
    a = api.BulkDecoder(threads=2, window_frames=48)
    monkeypatch.setenv("PDMP3_BULK_SNAPSHOT_ROWS", "1")
    s = api.BulkDecoder(threads=2, window_frames=48)
    monkeypatch.delenv("PDMP3_BULK_SNAPSHOT_ROWS")
    try:
        for k, mp3 in streams.items():
            assert np.array_equal(a.decode(mp3), s.decode(mp3)), k
    finally:
        a.close()
        s.close()


@pytest.mark.parametrize("scanners,window,sub", [(4, 16, 0), (3, 7, 0), (8, 64, 0), (4, 64, 8), (8, 256, 16), (3, 100, 7)])
def test_split_scan_decodes_what_the_one_thread_scan_decodes(streams, monkeypatch, scanners, window, sub):
    """host/split_scan.c par_drive (round 4): pre-pass + scanner threads + stitch in window order, forced on for host
    destinations too (by default it is taken for device destinations only).  Regular streams go the split way, irregular
    ones (resync, tags, truncation) are turned down before or half way -- then the windows that have gone up are let
    through and the stream is decoded again by the one-thread scan: either way the PCM is the one-thread decoder's, bit
    for bit, and a decoder is reusable after both.  sub > 0: the scanners' private windows have that many frames and the
    engine's windows are made of as many as are there ($PDMP3_BULK_SUB_FRAMES; by default 256, i.e. the whole window at
    these sizes): windows of every length up to `window`, the private windows' pools one behind the other"""
    from pdmp3_amd import api
    from test_split_scan import _regular_streams
    allst = dict(streams)
    allst.update(_regular_streams())
    monkeypatch.setenv("PDMP3_BULK_SCAN_THREADS", "0")
    ref = api.BulkDecoder(threads=2, window_frames=window)
    monkeypatch.setenv("PDMP3_BULK_SCAN_THREADS", str(scanners))
    par = api.BulkDecoder(threads=2, window_frames=window)
    monkeypatch.delenv("PDMP3_BULK_SCAN_THREADS")
    if sub:
        monkeypatch.setenv("PDMP3_BULK_SUB_FRAMES", str(sub))      # (read per stream)
    try:
        for name, mp3 in allst.items():
            want = ref.decode(mp3)
            got = par.decode(mp3)
            assert got.shape == want.shape and np.array_equal(got, want), name
        # ... and the regular ones really went the split way (a private window that does not fit the engine's open window
        # is held for the next one: until round 5 the stitcher then skipped the window behind it, the frame count came out
        # short and the stream was decoded again the one-thread way -- right PCM, a quarter of the rate)
        taken0, given0 = par.split_scans()
        regular = {k: v for k, v in _regular_streams().items() if k != "main_data_flipped"}
        for name, mp3 in regular.items():
            par.decode(mp3)
        taken, given = par.split_scans()
        assert given == given0 and taken - taken0 >= 5, (taken0, given0, taken, given)      # (the five constant-bitrate ones at least: 400-700 frames each)
    finally:
        ref.close()
        par.close()


def test_split_scan_is_the_default_for_device_destinations(monkeypatch):
    import torch
    from pdmp3_amd import api
    mp3 = packer.generate(n_frames=30000, seed=0xD5, sfreq=0, mode=1, mode_ext=2, bitrate_index=14)
    total, frames = api.scan_buffer(mp3)
    monkeypatch.setenv("PDMP3_BULK_SCAN_THREADS", "0")
    ref = api.BulkDecoder(threads=2)
    monkeypatch.delenv("PDMP3_BULK_SCAN_THREADS")
    dev = api.BulkDecoder(threads=2)
    try:
        want = ref.decode(mp3)
        out = torch.zeros(total // 2, dtype=torch.int16, device="cuda:0")
        torch.cuda.synchronize()                            # (the decoder's streams do not wait for torch's: the fill must be through)
        dev.decode_into_device(mp3, out)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), want)
    finally:
        ref.close()
        dev.close()


def test_decoders_side_by_side_into_device_memory():
    """four decoders on their own threads, each with a queue of files of a few minutes (4096-frame files: the split scan
    with 256-frame private windows, its threads kept from file to file and shared between the decoders that scan at the
    same time), PCM left in device memory, files queued back to back with one wait at the end: every file's PCM is the
    one-thread decoder's"""
    import threading
    import torch
    from pdmp3_amd import api
    rs = np.random.RandomState(77)
    files, sizes = [], []
    i = 0
    while len(files) < 16:
        i += 1
        kw = dict(n_frames=int(rs.randint(3500, 5000)), seed=400 + i, sfreq=int(rs.randint(0, 3)), mode=int(rs.choice([0, 1, 3])),
                  block_pct=(40, 10, 40, 10), mixed_pct=30, gain=(128, 140))
        if i % 2:
            kw.update(vbr=True, vbr_lo=5, vbr_hi=14)
        else:
            kw.update(bitrate_index=int(rs.randint(7, 15)))
        f = np.frombuffer(packer.generate(**kw), dtype=np.uint8)
        try:
            sizes.append(api.scan_buffer(f)[0])
        except api.RingReplay:                              # (1152-byte frames: the reference has no finite output)
            continue
        files.append(f)
    os.environ["PDMP3_BULK_SCAN_THREADS"] = "0"
    try:
        ref = api.BulkDecoder(threads=2)
    finally:
        del os.environ["PDMP3_BULK_SCAN_THREADS"]
    want = [ref.decode(f) for f in files]
    ref.close()
    outs = [torch.zeros(max(s, 2) // 2, dtype=torch.int16, device="cuda:0") for s in sizes]
    torch.cuda.synchronize()                                # (the decoders' streams do not wait for torch's: the fills must be through)
    errs = []

    def work(j):
        b = api.BulkDecoder(threads=2)
        try:
            for _ in range(2):                              # (twice: the second pass runs on the kept threads and memory)
                for i in range(j, len(files), 4):
                    got, _, _ = b.decode_into_device(files[i], outs[i], wait=False)
                    if got != sizes[i]:
                        errs.append((i, got))
                b.wait()
        except Exception as e:                              # noqa: BLE001
            errs.append(repr(e))
        finally:
            b.close()
    ts = [threading.Thread(target=work, args=(j,)) for j in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    assert not errs, errs
    for i in range(len(files)):
        assert np.array_equal(outs[i].cpu().numpy(), want[i]), i


@pytest.mark.parametrize("p_set,p_copy,p_mono,p_new", [(90, 30, 0, 0), (10, 90, 30, 20), (3, 50, 0, 1)])
def test_merge_kernels_on_random_merge_input(emul, p_set, p_copy, p_mono, p_new):
    """k_merge_outcome + k_merge_apply alone (pdmp3_hip_debug_merge) on the random merge input of
    test_unpack_emul.test_merge_by_blocks_on_random_merge_input, against what the rule (merge_slot, host build) makes of it:
    every byte of the records -- the kernel writes all 128 of each -- and the carried state; windows of one block, of one
    super-block and a frame, of 40 super-blocks and more (the rows of outcomes in front of a block are staged 40 at a time),
    last blocks of 1 and of 31 frames; the super-blocks' counters back at zero after every launch"""
    import ctypes as C
    from pdmp3_amd import hip
    from pdmp3_amd.hip import SIDE_DTYPE
    eng = hip.Engine()
    lib = eng.lib
    lib.pdmp3_hip_debug_merge.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pdmp3_hip_debug_merge.restype = C.c_int
    emul.emul_merge_case.argtypes = [C.c_uint64] + [C.c_int] * 5 + [C.c_void_p] * 5
    raw_bytes = emul.emul_merge_raw_bytes()
    assert raw_bytes == 80

    def p(a):
        return a.ctypes.data_as(C.c_void_p)
    try:
        for k, n in enumerate([1, 32, 33, 255, 257, 2049, 8192, 10271, 12289]):
            raw = np.zeros((n, 4, raw_bytes), np.uint8)
            bits = np.zeros((n, 80), np.uint8)
            st_in = np.zeros(256, np.uint16)
            side_ref = np.zeros((n, 4), SIDE_DTYPE)
            st_ref = np.zeros(256, np.uint16)
            rc = emul.emul_merge_case(0xBEEF + 131 * k + p_copy, n, p_set, p_copy, p_mono, p_new, p(raw), p(bits), p(st_in), p(side_ref), p(st_ref))
            assert rc == 0, (n, rc)                          # (the host form of the kernels agrees with the rule on this case)
            side = np.zeros((n, 4), SIDE_DTYPE)
            st_out = np.zeros(256, np.uint16)
            rc = lib.pdmp3_hip_debug_merge(eng.h, p(raw), p(bits), n, p(st_in), p(st_out), p(side))
            assert rc == 0, (n, lib.pdmp3_hip_last_error())
            bad = np.nonzero((side.view(np.uint8).reshape(n, -1) != side_ref.view(np.uint8).reshape(n, -1)).any(axis=1))[0]
            assert bad.size == 0, (n, bad[:5])
            assert np.array_equal(st_out[:232], st_ref[:232]), n
    finally:
        eng.close()


def test_device_destination_with_slots_of_a_few_frames():
    """a decoder whose windows hold 1 or 3 frames and a device pointer as destination: the split scan is taken (the PCM stays on
    the device) and finds that not even one of its private windows fits an empty window of the engine (a window's pool has no room
    for the reservoir image a private window starts with) -- it gives the stream up, not the engine, and the one-thread scan decodes
    it: same PCM as the default decoder's"""
    import torch
    from pdmp3_amd import api
    mp3 = packer.generate(n_frames=60, seed=0xD7, sfreq=0, mode=1, mode_ext=2, bitrate_index=11)
    ref = api.BulkDecoder(threads=2)
    try:
        want = ref.decode(mp3)
    finally:
        ref.close()
    for window in (1, 3):
        b = api.BulkDecoder(threads=2, window_frames=window)
        try:
            out = torch.zeros(want.size, dtype=torch.int16, device="cuda:0")
            torch.cuda.synchronize()
            b.decode_into_device(mp3, out)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), want), window
            taken, given = b.split_scans()
            assert taken + given == 1, (window, taken, given)
        finally:
            b.close()

"""The whole-stream decoder's split scan (host/split_scan.c par_drive; round 4): a pre-pass that hops from header to header,
K scanner threads that start at window boundaries from the state the pre-pass leaves there, a stitch in stream order.
The scanners run the unchanged stage-A code, so ONE scanner from frame 0 (K = 1) is the sequential scanner with private
windows; what has to hold is that K = 2, 3, 4, 8 give the same windows byte for byte -- side-info records (the fields
the reference leaves stale, H20, included), row descriptors, copy lists, the segments' images of the reservoir buffer
(the bytes beyond the reservoir's fill included: corrupt main data reads them) -- and that every stream the pre-pass
does not recognise as regular is turned down (the decoder then takes it the one-thread way).  No GPU."""
import ctypes as C

import numpy as np
import pytest

from pdmp3_amd.packer import packer
from tests.test_bulk_host import _streams as _host_streams

NOT_TAKEN, GIVEN_UP = -3, -4


def split_scan(mp3, window, k, iso=0):
    from pdmp3_amd import api
    lib = api.load_library()
    f = lib.pdmp3_amd_test_split_scan
    f.restype = C.c_longlong
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_uint, C.c_void_p, C.c_size_t, C.POINTER(C.c_longlong)]
    a = np.frombuffer(mp3, dtype=np.uint8) if len(mp3) else np.zeros(1, np.uint8)
    out = np.zeros(len(mp3) * 3 + (1 << 20), dtype=np.uint8)
    frames = C.c_longlong(0)
    n = f(a.ctypes.data_as(C.c_void_p), len(mp3), window, k, iso, out.ctypes.data_as(C.c_void_p), out.nbytes, C.byref(frames))
    return (n, frames.value, out[:max(n, 0)].tobytes())


def api_frames(mp3):
    from pdmp3_amd import api
    return api.scan_buffer(mp3)[1]


def _regular_streams():
    s = {
        "cbr_320": packer.generate(n_frames=700, seed=0x51, sfreq=0, mode=1, mode_ext=2, bitrate_index=14),
        "vbr_mixed_blocks": packer.generate(n_frames=600, seed=0x52, sfreq=1, mode=0, vbr=True, vbr_lo=3, vbr_hi=14,
                                            block_pct=(30, 15, 40, 15), mixed_pct=50),
        "mono_crc_320": packer.generate(n_frames=500, seed=0x53, sfreq=2, mode=3, bitrate_index=9, crc=True,
                                        block_pct=(20, 10, 60, 10)),
        "long_blocks_only": packer.generate(n_frames=400, seed=0x54, bitrate_index=11, block_pct=(100, 0, 0, 0)),      # subblock_gain never set (H20: stale = 0)
        "short_blocks_only": packer.generate(n_frames=400, seed=0x55, bitrate_index=11, block_pct=(0, 0, 100, 0)),    # table_select[2] never set
        "no_reservoir": packer.generate(n_frames=400, seed=0x56, bitrate_index=9, reservoir=False),
        "stereo_mono_stereo": (packer.generate(n_frames=150, seed=0x57, bitrate_index=9) +
                               packer.generate(n_frames=170, seed=0x58, mode=3, bitrate_index=7) +
                               packer.generate(n_frames=160, seed=0x59, mode=1, mode_ext=2, bitrate_index=11, block_pct=(10, 10, 70, 10))),
    }
    # corrupt MAIN DATA only (headers and side info intact up to main_data_begin): the pre-pass sees a regular stream, the
    # scanners' windows must still agree -- part2_3_length overruns read the reservoir buffer's stale bytes
    rs = np.random.RandomState(11)
    body = bytearray(packer.generate(n_frames=500, seed=0x5A, bitrate_index=12, block_pct=(40, 10, 40, 10)))
    pos = 0
    while pos + 1045 < len(body):
        for _ in range(3):
            k = pos + 40 + int(rs.randint(0, 900))
            body[k] ^= 1 << int(rs.randint(0, 8))
        pos += 1044 + (1 if body[pos + 2] & 2 else 0)
    s["main_data_flipped"] = bytes(body)
    return s


@pytest.fixture(scope="module")
def regular():
    return _regular_streams()


@pytest.mark.parametrize("window", [16, 37, 64])
def test_k_scanners_give_the_one_scanners_windows(regular, window):
    from pdmp3_amd import api
    for name, mp3 in regular.items():
        n1, f1, blob1 = split_scan(mp3, window, 1)
        _, frames = api.scan_buffer(mp3)
        if name == "main_data_flipped" and n1 in (NOT_TAKEN, GIVEN_UP):
            continue                                      # (a flip landed in a header after all: turned down, fine)
        assert n1 > 0 and f1 == frames, (name, n1, f1, frames)
        for k in (2, 3, 4, 8):
            nk, fk, blobk = split_scan(mp3, window, k)
            assert (nk, fk) == (n1, f1), (name, window, k)
            assert blobk == blob1, (name, window, k)


def test_private_windows_of_1024_frames(regular):
    """the size the whole-stream decoder uses (par_drive: the engine's windows are made of several): every private window
    but the stream's last has exactly that many frames -- no short first windows here, those are the one-thread scan's"""
    mp3 = packer.generate(n_frames=9000, seed=0x5C, sfreq=0, mode=1, mode_ext=2, bitrate_index=9, block_pct=(40, 10, 40, 10))
    n1, f1, blob1 = split_scan(mp3, 1024, 1)
    assert n1 > 0 and f1 == api_frames(mp3)
    first = np.frombuffer(blob1[:4], dtype=np.int32)[0]
    assert first == 1024
    for k in (2, 4, 8):
        assert split_scan(mp3, 1024, k) == (n1, f1, blob1), k
    short = mp3[:417 * 4500]                              # ends inside a window
    a = split_scan(short, 1024, 1)
    assert a[0] > 0 and split_scan(short, 1024, 3) == a


def test_iso_switches_reach_every_scanner(regular):
    mp3 = regular["vbr_mixed_blocks"]
    a = split_scan(mp3, 32, 1, iso=0x1f)
    assert a[0] > 0 and a[2] != split_scan(mp3, 32, 1)[2]
    assert split_scan(mp3, 32, 4, iso=0x1f) == a


def test_irregular_streams_are_turned_down():
    """resync in the middle, a leading tag, truncation inside a frame, an underflowing reservoir, too short a stream: the
    split scan says no (before or after it has started), the decoder then scans on one thread"""
    hs = _host_streams()
    body = packer.generate(n_frames=400, seed=0x5B, bitrate_index=9)
    cases = {
        "junk_resync": body[:80000] + b"\x00" * 333 + body[80000:],
        "long_tag": bytes(2000) + body,
        "cut_mid_frame": body[:100000] + body[100400:],
        "underflow": body[417 * 200:],                      # starts where main_data_begin > 0: the first frames underflow (H9)
        "tiny": hs["tiny"], "empty": hs["empty"], "clip": hs["clip"],
    }
    for name, mp3 in cases.items():
        for k in (1, 3):
            n, _, _ = split_scan(mp3, 16, k)
            assert n in (NOT_TAKEN, GIVEN_UP), (name, k, n)


@pytest.mark.parametrize("parts", [2, 3, 6, 8])
def test_hop_threads_give_the_one_pre_pass(regular, parts, monkeypatch):
    """the pre-pass in `parts` parts (hop threads from guessed places, $PDMP3_BULK_PREPASS_THREADS): same windows as the
    pre-pass that walks the whole stream itself, on constant and variable bitrate (the guesses land inside frames), with
    a change of the channel count, with corrupt main data (header-like bytes where a guess may land)"""
    for name, mp3 in regular.items():
        monkeypatch.setenv("PDMP3_BULK_PREPASS_THREADS", "1")
        one = split_scan(mp3, 37, 4)
        monkeypatch.setenv("PDMP3_BULK_PREPASS_THREADS", str(parts))
        got = split_scan(mp3, 37, 4)
        if got[0] == GIVEN_UP and one[0] > 0:
            continue                                      # (a guess that chained but was no boundary: turned down, allowed)
        assert got == one, (name, parts)


def test_hop_threads_turn_down_what_the_pre_pass_turns_down(monkeypatch):
    body = packer.generate(n_frames=900, seed=0x5C, bitrate_index=9)
    at = 0
    for _ in range(700):                                  # the 700th header (128 kbps at 44.1 kHz: 417 bytes + the padding bit)
        at += 417 + ((body[at + 2] >> 1) & 1)
    assert body[at] == 0xff and body[at + 1] == 0xfb
    cases = {"junk_resync_late": body[:300000] + b"\x00" * 333 + body[300000:],          # inside a hop thread's part
             "cut_mid_frame_late": body[:250000] + body[250400:],
             "bad_header_late": body[:at] + b"\xff\xfb\xf0\x00" + body[at + 4:]}   # bitrate index 15
    for parts in (1, 3, 5):
        monkeypatch.setenv("PDMP3_BULK_PREPASS_THREADS", str(parts))
        for name, mp3 in cases.items():
            n, _, _ = split_scan(mp3, 16, 3)
            assert n in (NOT_TAKEN, GIVEN_UP), (name, parts, n)


def test_random_streams_random_splits(monkeypatch):
    """40 streams with the packer's parameters drawn at random (rate, bitrate or VBR range, mode, CRC, reservoir on / off,
    block mix, fill), each cut at random: private windows of 5 .. 70 frames, 2 .. 8 scanners, the pre-pass in 1 .. 5 parts --
    the windows are the one scanner's, byte for byte"""
    rs = np.random.RandomState(0x5ca9)
    taken = 0
    for i in range(40):
        blocks = rs.randint(0, 100, 4) + 1
        mode = int(rs.choice([0, 1, 1, 2, 3]))
        kw = dict(n_frames=int(rs.randint(250, 900)), seed=1000 + i, sfreq=int(rs.randint(0, 3)), mode=mode, mode_ext=int(rs.randint(0, 4)),
                  crc=bool(rs.randint(0, 2)), reservoir=bool(rs.randint(0, 4)), block_pct=tuple(int(x) for x in blocks),
                  mixed_pct=int(rs.randint(0, 101)), fill_pct=int(rs.randint(60, 100)), big_pct=int(rs.randint(0, 12)))
        if rs.randint(0, 2):
            lo = int(rs.randint(2, 10))
            kw.update(vbr=True, vbr_lo=lo, vbr_hi=int(rs.randint(lo, 15)))
        else:
            kw.update(bitrate_index=int(rs.randint(4, 15)))
        mp3 = packer.generate(**kw)
        window = int(rs.randint(5, 71))
        monkeypatch.setenv("PDMP3_BULK_PREPASS_THREADS", "1")
        one = split_scan(mp3, window, 1)
        if one[0] in (NOT_TAKEN, GIVEN_UP):
            continue                                      # (e.g. the first frames underflow without a reservoir to start from)
        assert one[1] == api_frames(mp3), kw
        taken += 1
        for _ in range(2):
            k, parts = int(rs.randint(2, 9)), int(rs.randint(1, 6))
            monkeypatch.setenv("PDMP3_BULK_PREPASS_THREADS", str(parts))
            got = split_scan(mp3, window, k)
            if got[0] == GIVEN_UP and parts > 1:
                continue                                  # (a guess that chained but was no boundary)
            assert got == one, (kw, window, k, parts)
    assert taken >= 30

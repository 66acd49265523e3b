"""Random call sequences of the seven API functions, replayed on the product's handle and on the oracle's
restatement of the reference's streaming API (oracle/pdmp3_oracle_stream.c, P:2351-2535): every return code and
every byte count must agree call by call -- the read-ahead inside pdmp3_read (host/stream_api.c: read_ahead) must be
invisible.  Used by the CPU suite (parse-only handle: codes and counts) and the GPU suite (PCM as well)."""
import numpy as np

from pdmp3_amd.packer import packer


_BITRATES = [0, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320]
_RATES = [44100, 48000, 32000]


def _frame_starts(mp3):
    """byte offsets of the frames of a clean stream (P:1135-1138), plus the end"""
    out, pos = [], 0
    while pos + 4 <= len(mp3):
        h = int.from_bytes(mp3[pos:pos + 4], "big")
        assert h >> 20 == 0xFFF, pos
        out.append(pos)
        pos += 144 * _BITRATES[(h >> 12) & 15] * 1000 // _RATES[(h >> 10) & 3] + ((h >> 9) & 1)
    out.append(len(mp3))
    return out


def streams():
    """(name, bytes): clean CBR / VBR streams, mode switches, junk between frames, a truncated tail"""
    parts = [dict(n_frames=9, seed=31, bitrate_index=9), dict(n_frames=11, seed=32, mode=3, bitrate_index=7),
             dict(n_frames=6, seed=33, mode=1, mode_ext=2, bitrate_index=11, block_pct=(10, 10, 70, 10)),
             dict(n_frames=9, seed=34, sfreq=1, mode=3, bitrate_index=7)]
    switch = b"".join(packer.generate(**p) for p in parts)
    cbr = packer.generate(n_frames=70, seed=7, sfreq=0, mode=1, mode_ext=2, bitrate_index=14)
    vbr = packer.generate(n_frames=90, seed=8, sfreq=2, mode=0, mode_ext=0, vbr=True, block_pct=(40, 10, 40, 10), mixed_pct=50)
    small = packer.generate(n_frames=120, seed=9, sfreq=1, mode=3, bitrate_index=3)
    rs = np.random.RandomState(5)
    junk = bytearray(cbr)
    for cut in (5000, 23000, 41000):                      # garbage between frames: resync, failed attempts, rewinds
        junk[cut:cut] = bytes(rs.randint(0, 256, 700, dtype=np.uint8))
    # (no bit-flipped streams here: corrupt side info or main data makes the reference write past its arrays, H8 and
    # kin, and the oracle restates that faithfully -- the whole-stream tests cover them against the host stage)
    return [("cbr320", cbr), ("vbr32k", vbr), ("mono_small", small), ("switch", switch), ("junk", bytes(junk)),
            ("cut", cbr[:len(cbr) - 517])]


def replay(seed, mp3, dec, orc, compare_pcm, max_calls=4000):
    """-> (calls made, bytes fed).  `dec` / `orc`: objects with open_feed / feed / read / decode / getformat."""
    rs = np.random.RandomState(seed)
    try:
        restarts = _frame_starts(mp3)[:-1]
    except AssertionError:                                # junk inside: restart from the top only
        restarts = [0]
    dec.open_feed(); orc.open_feed()
    pos, calls, idle = 0, 0, 0
    style = rs.randint(0, 4)                              # how eager the caller is with its feeds
    while calls < max_calls and idle < 12:
        calls += 1
        r = rs.rand()
        if r < (0.30, 0.45, 0.15, 0.30)[style]:
            n = int(rs.choice([1, 7, 417, 1044, 1152, 3000, 4096, 8192, 16384, 20000]))
            chunk = mp3[pos:pos + n]
            if not chunk:
                idle += 1
                continue
            a, b = dec.feed(chunk), orc.feed(chunk)
            assert a == b, ("feed", calls, n, a, b)
            if a == 0:
                pos += len(chunk)
        elif r < 0.80:
            n = int(rs.choice([1, 2, 3, 100, 1151, 2304, 4608, 5000, 16384, 16384, 16384, 40000, 100000]))
            (a, pa), (b, pb) = dec.read(n), orc.read(n)
            assert a == b and len(pa) == len(pb), ("read", calls, n, a, b, len(pa), len(pb))
            if compare_pcm and pa:
                d = np.abs(np.frombuffer(pa, np.int16).astype(np.int32) - np.frombuffer(pb, np.int16).astype(np.int32))
                assert d.max() <= 1, ("read pcm", calls, int(d.max()))
            if pos >= len(mp3):
                idle += 1
        elif r < 0.92:
            n = int(rs.choice([0, 300, 1044, 4096, 9000, 30000]))
            o = int(rs.choice([0, 0, 4608, 16384]))
            chunk = mp3[pos:pos + n]
            (a, pa), (b, pb) = dec.decode(chunk, o), orc.decode(chunk, o)
            assert a == b and len(pa) == len(pb), ("decode", calls, n, o, a, b, len(pa), len(pb))
            pos += len(chunk)                              # (what did not fit is dropped by both, H16)
            if compare_pcm and pa:
                d = np.abs(np.frombuffer(pa, np.int16).astype(np.int32) - np.frombuffer(pb, np.int16).astype(np.int32))
                assert d.max() <= 1, ("decode pcm", calls, int(d.max()))
        elif r < 0.99:
            a, b = dec.getformat(), orc.getformat()
            assert a == b, ("getformat", calls, a, b)
        else:
            assert dec.open_feed() == orc.open_feed()
            pos = int(restarts[rs.randint(0, max(1, len(restarts) // 2))])   # start again at some frame of the first half
    return calls, pos

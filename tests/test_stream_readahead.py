"""pdmp3_read's read-ahead is invisible: random call sequences on a parse-only handle (no GPU: return codes and byte
counts) against the oracle's restatement of the reference's streaming API."""
import numpy as np
import pytest

import stream_replay
from oracle.oracle import OracleStream


@pytest.mark.parametrize("seed", range(12))
def test_call_sequences_match_the_reference_api(oracle, seed):
    from pdmp3_amd import api
    for name, mp3 in stream_replay.streams():
        dec = api.Decoder(parse_only=True)
        orc = OracleStream(oracle)
        try:
            stream_replay.replay(1000 * seed + len(name), mp3, dec, orc, compare_pcm=False)
        finally:
            dec.close()
            orc.close()


def _read_path_records(mp3, cap, read_bytes=65536):
    """records tapped from pdmp3_read's own path (read_ahead: batches decoded by the caller and the helper threads).
    After PDMP3_NEED_MORE the ring holds < 1152 bytes (H10): seven 2048-byte feeds then fill it to 14-15 KiB, never to
    the last byte (a ring filled exactly looks EMPTY to the reference, P:1062-1068)"""
    from pdmp3_amd import api
    dec = api.Decoder(parse_only=True)
    dec.set_tap(cap)
    pos, total = 0, 0
    while True:
        rc, pcm = dec.read(read_bytes)
        total += len(pcm)
        if rc == api.PDMP3_ERR:
            break
        if rc == api.PDMP3_NEED_MORE:
            if pos >= len(mp3):
                break
            for _ in range(7):
                if pos < len(mp3):
                    assert dec.feed(mp3[pos:pos + 2048]) == api.PDMP3_OK
                    pos += 2048
    sp, sd = dec.tap()
    out = sp.copy(), sd.copy(), total
    dec.close()
    return out


def test_batches_decoded_with_helpers_give_the_reference_parsers_records():
    """read-ahead batches of up to 16 frames, their main data decoded by several threads at once (hp_run): the records
    are those of the frame-by-frame parser (itself pinned to the oracle in test_host_stage.py) -- scalefactors that
    survive frames (scfsi, H4-H6), block switches and the reservoir included -- also with three handles reading at the
    same time -- six handles over the four slots of the shared helper pool: batches side by side in the helpers' hands, and
    a handle that finds every slot taken decodes alone"""
    import threading
    from pdmp3_amd import api
    from pdmp3_amd.packer import packer
    cases = [dict(n_frames=150, seed=61, sfreq=0, mode=1, mode_ext=2, bitrate_index=14, block_pct=(40, 10, 40, 10)),
             dict(n_frames=200, seed=62, sfreq=2, mode=3, vbr=True, vbr_lo=1, vbr_hi=9),
             dict(n_frames=150, seed=63, sfreq=1, mode=0, mode_ext=0, bitrate_index=9, mixed_pct=50, block_pct=(30, 10, 50, 10))]
    mp3s = [packer.generate(**kw) for kw in cases]
    want = [api.parse_like_cli(m, 400) for m in mp3s]
    for m, w in zip(mp3s, want):
        sp, sd, total = _read_path_records(m, 400)
        assert sp.shape[0] > 100 and total > 0
        n = min(sp.shape[0], w[0].shape[0])                   # (the two drivers stop at different tails, H10)
        assert n >= sp.shape[0] - 3
        assert np.array_equal(sp[:n], w[0][:n]) and np.array_equal(sd[:n].view(np.uint8), w[1][:n].view(np.uint8))
    got = [None] * 6
    def run(i):
        got[i] = _read_path_records(mp3s[i % 3], 400, read_bytes=16384 if i & 1 else 65536)
    ths = [threading.Thread(target=run, args=(i,)) for i in range(6)]
    for t in ths: t.start()
    for t in ths: t.join(120)
    for i in range(6):
        assert got[i] is not None
        w = want[i % 3]
        n = min(got[i][0].shape[0], w[0].shape[0])
        assert n >= got[i][0].shape[0] - 3
        assert np.array_equal(got[i][0][:n], w[0][:n]) and np.array_equal(got[i][1][:n].view(np.uint8), w[1][:n].view(np.uint8))

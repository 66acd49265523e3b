import hashlib

import numpy as np


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def nch_of(side):
    return 1 if ((int(side["frame"][0, 0, 0]) >> 2) & 3) == 3 else 2


def pcm_tolerance_scaled(stage3):
    """ONLY for inputs that are not corpus cases with a literal tolerance (tests/corpus.py PCM_TOL_LSB): bit-flipped
    streams whose global_gain drives the synthesis 100x .. 4000x past full scale (test_gpu_bulk.py).  +-1 LSB, or the
    north-star float tolerance 1e-5 relative to the synthesis amplitude (an LSB is then far below the binary32
    resolution of the sums)."""
    amp = float(np.abs(stage3).max()) * 32.0       # bound of one matrixing output
    return max(1, int(np.ceil(1e-5 * 32767.0 * amp)))


def assert_pcm_close(got, want, tol=1, what=""):
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert d.max() <= tol, "%s PCM max-abs-diff %d LSB > %d (at %s)" % (
        what, d.max(), tol, np.unravel_index(d.argmax(), d.shape))
    return int(d.max()), int((d > 0).sum())


def oracle_decode_many(oracle, streams, threads=None):
    """the oracle's CLI-style decode of several byte streams on a thread pool (the ctypes call releases the GIL and the
    oracle's stream state is per call): int16 arrays, in order.  Test infrastructure only."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    oracle.decode_buffer_like_cli(streams[0][:8192])         # the oracle's lazily built tables, once, on this thread
    if threads is None:
        threads = max(1, min(16, len(os.sched_getaffinity(0))))
    with ThreadPoolExecutor(threads) as ex:
        return [np.frombuffer(b, dtype=np.int16) for b in ex.map(oracle.decode_buffer_like_cli, streams)]


def level(pcm):
    """(median, 99th percentile, fraction clipped) of |PCM|, to print beside a parity result"""
    a = np.abs(np.asarray(pcm, dtype=np.int16).astype(np.int32)).ravel()
    if a.size == 0:
        return 0, 0, 0.0
    return int(np.median(a)), int(np.percentile(a, 99)), float((a >= 32767).mean())

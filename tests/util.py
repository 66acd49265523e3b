import hashlib

import numpy as np


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def nch_of(side):
    return 1 if ((int(side["frame"][0, 0, 0]) >> 2) & 3) == 3 else 2


def pcm_tolerance(stage3):
    """Tolerance in int16 LSB: +-1 LSB, or the north-star float tolerance 1e-5
    relative to the synthesis amplitude when the signal is driven far beyond
    full scale (|sum| >> 1: the int16 result is then mostly clipped and an LSB
    is far below binary32 resolution of the sums)."""
    amp = float(np.abs(stage3).max()) * 32.0       # bound of one matrixing output
    return max(1, int(np.ceil(1e-5 * 32767.0 * amp)))


def assert_pcm_close(got, want, tol=1, what=""):
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert d.max() <= tol, "%s PCM max-abs-diff %d LSB > %d (at %s)" % (
        what, d.max(), tol, np.unravel_index(d.argmax(), d.shape))
    return int(d.max()), int((d > 0).sum())

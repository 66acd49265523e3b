"""How the hot path shards across the GPUs of one node.

Units are frames.  A stream is cut into contiguous frame ranges, one per rank;
the only coupling between neighbouring ranges is the 2-granule synthesis
history (+ the H5 corner), so every rank but the first starts HALO_FRAMES
earlier, decodes-and-discards the halo, and no collective is needed inside the
decode loop.  Whole files (independent streams, config C4) are dealt
largest-first to the least loaded rank.  The one exchange of the path is the
final PCM gather to rank 0.
"""
# The fixed halo covers a stream whose channel count does not change around the cut (C5 is all stereo; a file of a
# corpus is decoded whole by one rank).  Across a run of MONO frames channel 1's state is the last stereo frame's,
# however far back that is: inside one launch the kernel looks for it itself (decode_core.h: pre-halo), but it cannot
# look in front of the records it is given.  So when the frame before a cut is mono, the shard starts further back:
# at the frame in front of the last stereo frame (what the kernel's pre-halo wants to see), or at that frame itself
# when it carries PDMP3_FR_RESET.  Pass the per-frame flag bytes (pdmp3_gc_side.frame of any record of the frame) for
# that; without them the halo is the fixed one.
HALO_FRAMES = 2

_FR_MODE_SHIFT, _FR_RESET = 2, 0x40           # include/pdmp3_hip.h: PDMP3_FR_MODE_SHIFT, PDMP3_FR_RESET


def frame_range(n_frames, rank, world):
    """[lo, hi) of rank's share of n_frames frames (contiguous, sizes differ by <= 1)."""
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def halo_start(lo, frame_flags=None, halo=HALO_FRAMES):
    """First frame a shard that emits frames from `lo` on has to decode (cf. last_stereo_or_reset in decode_core.h)."""
    first = max(0, lo - halo)
    if frame_flags is None or lo <= 0:
        return first

    def mono(f):
        return ((int(frame_flags[f]) >> _FR_MODE_SHIFT) & 3) == 3

    def reset(f):
        return bool(int(frame_flags[f]) & _FR_RESET)

    if not mono(lo - 1) or reset(lo - 1):
        return first
    f = lo - 2
    while f >= 0 and mono(f) and not reset(f):
        f -= 1
    if f < 0 or mono(f):                      # no stereo frame back to the start / a mono RESET frame: channel 1 is zero
        return first
    return min(first, f if (reset(f) or f == 0) else f - 1)


def shard_with_halo(n_frames, rank, world, halo=HALO_FRAMES, frame_flags=None):
    """(first_frame_to_decode, n_frames_to_decode, n_halo_frames_to_discard)."""
    lo, hi = frame_range(n_frames, rank, world)
    first = halo_start(lo, frame_flags, halo)
    return first, hi - first, lo - first


def frame_flags_of(side):
    """The per-frame flag bytes (pdmp3_gc_side.frame) of a batch of records -- what the host stage hands out with them
    (pdmp3_amd.api.parse_like_cli / the bulk decoder's record form): numpy structured array [n, 2, 2] or uint8 [n, 4, 128]."""
    import numpy as np
    a = np.asarray(side)
    if a.dtype.names:
        return np.ascontiguousarray(a["frame"].reshape(a.shape[0], -1)[:, 0])
    return np.ascontiguousarray(a.reshape(a.shape[0], -1, 128)[:, 0, 7])


def shard_records(side, rank, world, halo=HALO_FRAMES):
    """shard_with_halo for a stream that is at hand as records: the cut is moved back past mono runs as far as channel 1's
    state reaches (halo_start), read off the records' own flag bytes."""
    return shard_with_halo(len(side), rank, world, halo, frame_flags_of(side))


def assign_files(sizes, world):
    """Largest-first greedy: returns a list (per rank) of file indices."""
    load = [0] * world
    out = [[] for _ in range(world)]
    for i in sorted(range(len(sizes)), key=lambda k: -sizes[k]):
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += sizes[i]
    return out

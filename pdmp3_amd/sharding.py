"""How the hot path shards across the GPUs of one node.

Units are frames.  A stream is cut into contiguous frame ranges, one per rank;
the only coupling between neighbouring ranges is the 2-granule synthesis
history (+ the H5 corner), so every rank but the first starts HALO_FRAMES
earlier, decodes-and-discards the halo, and no collective is needed inside the
decode loop.  Whole files (independent streams, config C4) are dealt
largest-first to the least loaded rank.  The one exchange of the path is the
final PCM gather to rank 0.
"""
# The fixed halo covers a stream whose channel count does not change inside it (C5 is all stereo; a file of a corpus
# is decoded whole by one rank).  Across a run of MONO frames channel 1's state is the last stereo frame's, however
# far back that is: inside one launch the kernel looks for it itself (decode_core.h: pre-halo), but it cannot look
# in front of the records it is given -- a shard boundary right after mono frames of a stream that also has stereo
# frames would start channel 1 from zero.  Shard such streams at frames where the channel count is settled (or
# decode them through the whole-stream decoder, which carries the state from window to window).
HALO_FRAMES = 2


def frame_range(n_frames, rank, world):
    """[lo, hi) of rank's share of n_frames frames (contiguous, sizes differ by <= 1)."""
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def shard_with_halo(n_frames, rank, world, halo=HALO_FRAMES):
    """(first_frame_to_decode, n_frames_to_decode, n_halo_frames_to_discard)."""
    lo, hi = frame_range(n_frames, rank, world)
    h = min(halo, lo)
    return lo - h, (hi - lo) + h, h


def assign_files(sizes, world):
    """Largest-first greedy: returns a list (per rank) of file indices."""
    load = [0] * world
    out = [[] for _ in range(world)]
    for i in sorted(range(len(sizes)), key=lambda k: -sizes[k]):
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += sizes[i]
    return out

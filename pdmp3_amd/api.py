"""ctypes mirror of include/pdmp3.h (pdmp3_amd/libpdmp3.so): the reference's
libmpg123-style streaming API, names and semantics unchanged.  Used by tests
and examples; the library itself is plain C."""
import ctypes as C
import os

import numpy as np

from .hip import SIDE_DTYPE

_HERE = os.path.dirname(os.path.abspath(__file__))
PDMP3_OK, PDMP3_ERR, PDMP3_NEED_MORE, PDMP3_NEW_FORMAT, PDMP3_NO_SPACE = 0, -1, -10, -11, 7
PDMP3_ENC_SIGNED_16 = 0xD0
_LIB = None

BULK_EXPORTS = ["pdmp3_amd_bulk_new", "pdmp3_amd_bulk_new_ex", "pdmp3_amd_bulk_new_on", "pdmp3_amd_bulk_delete", "pdmp3_amd_bulk_threads", "pdmp3_amd_bulk_split_scans", "pdmp3_amd_bulk_set_quirks",
                "pdmp3_amd_scan_buffer", "pdmp3_amd_scan_buffer_iso", "pdmp3_amd_corpus_assign", "pdmp3_amd_corpus_decode", "pdmp3_amd_bulk_decode", "pdmp3_amd_bulk_decode_async", "pdmp3_amd_bulk_wait", "pdmp3_amd_bulk_new_parse_only", "pdmp3_amd_bulk_parse",
                "pdmp3_amd_bulk_new_parse_bits", "pdmp3_amd_bulk_parse_bits", "pdmp3_amd_bulk_parse_pool", "pdmp3_amd_pcm_alloc", "pdmp3_amd_pcm_free", "pdmp3_amd_stream_loop", "pdmp3_amd_write_wav"]

# include/pdmp3_hip.h: pdmp3_gc_bits / pdmp3_frame_bits
GC_BITS_DTYPE = np.dtype([("part2_3_length", "<u2"), ("big_values", "<u2"), ("global_gain", "u1"), ("scalefac_compress", "u1"),
                          ("flags", "u1"), ("table_select", "u1", (3,)), ("subblock_gain", "u1", (3,)),
                          ("region0_count", "u1"), ("region1_count", "u1"), ("count1table_select", "u1")])
FRAME_BITS_DTYPE = np.dtype([("frame", "u1"), ("scfsi", "u1", (2,)), ("iso", "u1"), ("reserved", "u1", (12,)), ("gc", GC_BITS_DTYPE, (4,))])
RESERVOIR_BYTES = 2064
API_EXPORTS = ["pdmp3_new", "pdmp3_delete", "pdmp3_open_feed", "pdmp3_feed", "pdmp3_read",
               "pdmp3_decode", "pdmp3_getformat", "pdmp3", "pdmp3_amd_set_encoding", "pdmp3_amd_set_quirks"]
# include/pdmp3.h: the ISO-correct switches
ISO_TABLE33, ISO_MS_BOUND, ISO_IS_SHORT, ISO_SF21, ISO_SF12, ISO_IS_BOUND, ISO_ALL = 0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x3f
ISO_LSF = 0x40            # include/pdmp3.h PDMP3_ISO_LSF: MPEG-2 LSF / MPEG-2.5 streams are decoded (the reference rejects them)
PDMP3_ENC_SIGNED_16, PDMP3_ENC_FLOAT_32 = 0xD0, 0x200


def library_path():
    # PDMP3_HOST_LIB: alternative build of the host library (A/B experiments; it brings the engine library it was linked with)
    return os.environ.get("PDMP3_HOST_LIB") or os.path.join(_HERE, "libpdmp3.so")


def load_library():
    global _LIB
    if _LIB is not None:
        return _LIB
    import torch  # noqa: F401  (one HIP runtime per process: torch's, loaded first)
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError("pdmp3_amd: %s is missing -- run __graft_entry__.build()" % path)
    lib = C.CDLL(path)
    vp = C.c_void_p
    lib.pdmp3_new.restype = vp
    lib.pdmp3_new.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
    lib.pdmp3_delete.argtypes = [vp]
    lib.pdmp3_open_feed.argtypes = [vp]
    lib.pdmp3_feed.argtypes = [vp, vp, C.c_size_t]
    lib.pdmp3_read.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.pdmp3_decode.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.pdmp3_getformat.argtypes = [vp, C.POINTER(C.c_long), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.pdmp3_amd_set_encoding.argtypes = [vp, C.c_int]
    lib.pdmp3_amd_new_parse_only.restype = vp
    lib.pdmp3_amd_set_tap.argtypes = [vp, vp, vp, C.c_int]
    lib.pdmp3_amd_tap_count.argtypes = [vp]
    lib.pdmp3_amd_parse_available.argtypes = [vp]
    # include/pdmp3_bulk.h
    lib.pdmp3_amd_bulk_new.restype = vp
    lib.pdmp3_amd_bulk_new.argtypes = [C.c_int, C.c_int]
    lib.pdmp3_amd_bulk_new_parse_only.restype = vp
    lib.pdmp3_amd_bulk_new_parse_only.argtypes = [C.c_int, C.c_int]
    lib.pdmp3_amd_bulk_delete.argtypes = [vp]
    lib.pdmp3_amd_bulk_threads.argtypes = [vp]
    if hasattr(lib, "pdmp3_amd_bulk_split_scans"):          # (absent from builds before round 5's end: PDMP3_HOST_LIB A/B runs)
        lib.pdmp3_amd_bulk_split_scans.argtypes = [vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
        lib.pdmp3_amd_bulk_split_scans.restype = None
    lib.pdmp3_amd_scan_buffer.restype = C.c_longlong
    lib.pdmp3_amd_scan_buffer.argtypes = [vp, C.c_size_t, C.POINTER(C.c_longlong)]
    lib.pdmp3_amd_bulk_decode.restype = C.c_longlong
    lib.pdmp3_amd_bulk_decode.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_long), C.POINTER(C.c_int)]
    lib.pdmp3_amd_bulk_new_ex.restype = vp
    lib.pdmp3_amd_bulk_new_ex.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.pdmp3_amd_bulk_new_on.restype = vp
    lib.pdmp3_amd_bulk_new_on.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    lib.pdmp3_amd_bulk_new_parse_bits.restype = vp
    lib.pdmp3_amd_bulk_new_parse_bits.argtypes = []
    lib.pdmp3_amd_bulk_parse_bits.restype = C.c_longlong
    lib.pdmp3_amd_bulk_parse_bits.argtypes = [vp, vp, C.c_size_t, vp, vp, C.c_size_t, C.POINTER(C.c_longlong)]
    lib.pdmp3_amd_bulk_decode_async.restype = C.c_longlong
    lib.pdmp3_amd_bulk_decode_async.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_long), C.POINTER(C.c_int)]
    lib.pdmp3_amd_bulk_wait.argtypes = [vp]
    lib.pdmp3_amd_pcm_alloc.restype = vp
    lib.pdmp3_amd_pcm_alloc.argtypes = [C.c_size_t]
    lib.pdmp3_amd_pcm_free.argtypes = [vp]
    lib.pdmp3_amd_bulk_parse.restype = C.c_longlong
    lib.pdmp3_amd_bulk_parse.argtypes = [vp, vp, C.c_size_t, vp, vp, C.c_size_t, C.POINTER(C.c_longlong)]
    _LIB = lib
    return lib


class Decoder:
    """pdmp3_handle wrapper.  parse_only=True gives the GPU-less test handle."""

    def __init__(self, parse_only=False):
        self.lib = load_library()
        if parse_only:
            self.h = self.lib.pdmp3_amd_new_parse_only()
        else:
            err = C.c_int(0)
            self.h = self.lib.pdmp3_new(None, C.byref(err))
        if not self.h:
            raise RuntimeError("pdmp3_new failed (no MI355X transform engine; there is no CPU fallback)")
        self.lib.pdmp3_open_feed(self.h)

    def close(self):
        if self.h:
            self.lib.pdmp3_delete(self.h)
            self.h = None

    def open_feed(self):
        return self.lib.pdmp3_open_feed(self.h)

    def set_quirks(self, iso_mask):
        """pdmp3_amd_set_quirks: PDMP3_ISO_* bits = the standard's behaviour instead of the reference's"""
        self.lib.pdmp3_amd_set_quirks.argtypes = [C.c_void_p, C.c_uint]
        if self.lib.pdmp3_amd_set_quirks(self.h, iso_mask) != 0:
            raise ValueError("pdmp3_amd_set_quirks: unknown bits in %#x" % iso_mask)

    def feed(self, data: bytes):
        buf = (C.c_ubyte * len(data)).from_buffer_copy(data)
        return self.lib.pdmp3_feed(self.h, buf, len(data))

    def read(self, outsize):
        out = (C.c_ubyte * outsize)()
        done = C.c_size_t(0)
        rc = self.lib.pdmp3_read(self.h, out, outsize, C.byref(done))
        return rc, bytes(out[:done.value])

    def decode(self, data: bytes, outsize):
        buf = (C.c_ubyte * max(1, len(data))).from_buffer_copy(data or b"\0")
        out = (C.c_ubyte * outsize)() if outsize else None
        done = C.c_size_t(0)
        rc = self.lib.pdmp3_decode(self.h, buf, len(data), out, outsize, C.byref(done))
        return rc, bytes(out[:done.value]) if outsize else b""

    def getformat(self):
        rate, ch, enc = C.c_long(0), C.c_int(0), C.c_int(0)
        rc = self.lib.pdmp3_getformat(self.h, C.byref(rate), C.byref(ch), C.byref(enc))
        return rc, rate.value, ch.value, enc.value

    def set_encoding(self, enc):
        return self.lib.pdmp3_amd_set_encoding(self.h, enc)

    def set_tap(self, cap_frames):
        self._tap_sp = np.zeros((cap_frames, 2, 2, 576), dtype=np.int16)
        self._tap_sd = np.zeros((cap_frames, 2, 2), dtype=SIDE_DTYPE)
        self.lib.pdmp3_amd_set_tap(self.h, self._tap_sp.ctypes.data_as(C.c_void_p),
                                   self._tap_sd.ctypes.data_as(C.c_void_p), cap_frames)

    def tap(self):
        n = min(self.lib.pdmp3_amd_tap_count(self.h), self._tap_sp.shape[0])
        return self._tap_sp[:n], self._tap_sd[:n]

    def parse_available(self):
        return self.lib.pdmp3_amd_parse_available(self.h)


def stream_loop(mp3, feed_bytes=4096, read_bytes=16384, eager=False, want_pcm=True):
    """pdmp3_amd_stream_loop: the feed / read loop of the reference's driver run in C over a memory buffer.
    -> (PCM bytes delivered, int16 array or None)"""
    lib = load_library()
    lib.pdmp3_amd_stream_loop.restype = C.c_longlong
    lib.pdmp3_amd_stream_loop.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]
    a = _as_u8(mp3)
    out = np.empty((len(a) // 96 + 8) * 2304, dtype=np.int16) if want_pcm else None
    n = lib.pdmp3_amd_stream_loop(a.ctypes.data_as(C.c_void_p), len(a), out.ctypes.data_as(C.c_void_p) if want_pcm else None,
                                  out.nbytes if want_pcm else 0, feed_bytes, read_bytes, int(eager))
    if n < 0:
        raise RuntimeError("pdmp3_amd_stream_loop: no transform engine")
    return n, (out[:n // 2] if want_pcm else None)


def decode_like_cli(mp3: bytes, dec: "Decoder" = None):
    """The CLI driver loop pdmp3() (pdmp3.c:2552-2587) over a memory buffer."""
    own = dec is None
    if own:
        dec = Decoder()
    dec.open_feed()
    out, pos = [], 0
    while True:
        rc, pcm = dec.read(16384)
        if rc == PDMP3_ERR:
            break
        out.append(pcm)
        if rc == PDMP3_NEED_MORE:
            chunk = mp3[pos:pos + 4096]
            if not chunk:
                break
            dec.feed(chunk)
            pos += len(chunk)
    if own:
        dec.close()
    return b"".join(out)


def parse_like_cli(mp3: bytes, cap_frames, iso=0):
    """Host stage only (no GPU): records the host parser emits when driven with
    the CLI's feed cadence."""
    dec = Decoder(parse_only=True)
    if iso:
        dec.set_quirks(iso)
    dec.set_tap(cap_frames)
    pos = 0
    while True:
        rc = dec.parse_available()
        if rc == PDMP3_ERR:
            break
        chunk = mp3[pos:pos + 4096]
        if not chunk:
            break
        dec.feed(chunk)
        pos += len(chunk)
    sp, sd = dec.tap()
    dec.close()
    return sp.copy(), sd.copy()


def _as_u8(data):
    a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    return a if a.size else np.zeros(1, dtype=np.uint8)


class RingReplay(RuntimeError):
    """PDMP3_BULK_REPLAY: the reference would replay its input ring on this stream (include/pdmp3_bulk.h)"""


def scan_buffer(mp3, iso=0):
    """(pcm_bytes, frames) the CLI driver would produce for this stream (include/pdmp3_bulk.h); iso: the decoder's switches
    (PDMP3_ISO_LSF changes what counts as a frame)"""
    lib = load_library()
    a = _as_u8(mp3)
    frames = C.c_longlong(0)
    lib.pdmp3_amd_scan_buffer_iso.restype = C.c_longlong
    lib.pdmp3_amd_scan_buffer_iso.argtypes = [C.c_void_p, C.c_size_t, C.c_uint, C.POINTER(C.c_longlong)]
    total = lib.pdmp3_amd_scan_buffer_iso(a.ctypes.data_as(C.c_void_p), len(mp3), iso, C.byref(frames))
    if total == -2:
        raise RingReplay("the reference replays its input ring on this stream (no finite output)")
    return total, frames.value


class BulkDecoder:
    """include/pdmp3_bulk.h: whole-stream decode, host Huffman on a thread pool + pipelined GPU batches.
    parse_only=True: host stages only (records out), for machines without a GPU."""

    def __init__(self, threads=0, window_frames=0, parse_only=False, host_huffman=False, device=None):
        """host_huffman=False: scalefactors + Huffman run on the device (pdmp3_hip_stream_submit_bits), the pool only
        copies PCM out; True: they run on the pool's threads (the engine gets decoded records)."""
        self.lib = load_library()
        self.parse_only = parse_only
        if parse_only:
            self.h = self.lib.pdmp3_amd_bulk_new_parse_only(threads, window_frames)
        elif device is not None:
            self.h = self.lib.pdmp3_amd_bulk_new_on(threads, window_frames, 1 if host_huffman else 0, int(device))
        else:
            self.h = self.lib.pdmp3_amd_bulk_new_ex(threads, window_frames, 1 if host_huffman else 0)
        if not self.h:
            raise RuntimeError("pdmp3_amd_bulk_new failed (no MI355X transform engine; there is no CPU fallback)")
        self.threads = self.lib.pdmp3_amd_bulk_threads(self.h)

    def close(self):
        if self.h:
            self.lib.pdmp3_amd_bulk_delete(self.h)
            self.h = None

    def set_quirks(self, iso_mask):
        self.lib.pdmp3_amd_bulk_set_quirks.argtypes = [C.c_void_p, C.c_uint]
        if self.lib.pdmp3_amd_bulk_set_quirks(self.h, iso_mask) != 0:
            raise ValueError("pdmp3_amd_bulk_set_quirks: unknown bits in %#x" % iso_mask)
        self.iso = iso_mask

    def decode_into(self, mp3, out: np.ndarray):
        a = _as_u8(mp3)
        rate, ch = C.c_long(0), C.c_int(0)
        total = self.lib.pdmp3_amd_bulk_decode(self.h, a.ctypes.data_as(C.c_void_p), len(mp3),
                                               out.ctypes.data_as(C.c_void_p), out.nbytes, C.byref(rate), C.byref(ch))
        if total == -2:
            raise RingReplay("the reference replays its input ring on this stream (no finite output)")
        if total < 0:
            raise RuntimeError("pdmp3_amd_bulk_decode: engine failure")
        return total, rate.value, ch.value

    def decode_into_async(self, mp3, out: np.ndarray):
        """queue one stream; `out` is complete after wait()"""
        a = _as_u8(mp3)
        rate, ch = C.c_long(0), C.c_int(0)
        total = self.lib.pdmp3_amd_bulk_decode_async(self.h, a.ctypes.data_as(C.c_void_p), len(mp3),
                                                     out.ctypes.data_as(C.c_void_p), out.nbytes, C.byref(rate), C.byref(ch))
        if total == -2:
            raise RingReplay("the reference replays its input ring on this stream (no finite output)")
        if total < 0:
            raise RuntimeError("pdmp3_amd_bulk_decode_async: engine failure")
        return total, rate.value, ch.value

    def decode_into_device(self, mp3, out_tensor, wait=True):
        """PCM into a torch int16 tensor on the GPU.  Windows of one channel count go from the engine's buffer to the
        tensor without leaving the device; a window that mixes mono and stereo frames, or the one the tensor ends in,
        is staged in pinned host memory and copied from there.  The decoder writes from its own HIP streams, which do not
        wait for torch's: work of the caller's that is still pending on the tensor (a fill, say) must be through first
        (torch.cuda.synchronize()), and with wait=False the tensor is the decoder's until wait() has returned."""
        a = _as_u8(mp3)
        rate, ch = C.c_long(0), C.c_int(0)
        f = self.lib.pdmp3_amd_bulk_decode if wait else self.lib.pdmp3_amd_bulk_decode_async
        total = f(self.h, a.ctypes.data_as(C.c_void_p), len(mp3), C.c_void_p(out_tensor.data_ptr()),
                  out_tensor.numel() * out_tensor.element_size(), C.byref(rate), C.byref(ch))
        if total == -2:
            raise RingReplay("the reference replays its input ring on this stream (no finite output)")
        if total < 0:
            raise RuntimeError("pdmp3_amd_bulk_decode: engine failure")
        return total, rate.value, ch.value

    def split_scans(self):
        """-> (streams the split scan decoded to their end, streams it gave up half way and handed to the one-thread scan)"""
        a, g = C.c_longlong(0), C.c_longlong(0)
        self.lib.pdmp3_amd_bulk_split_scans(self.h, C.byref(a), C.byref(g))
        return a.value, g.value

    def wait(self):
        if self.lib.pdmp3_amd_bulk_wait(self.h) != 0:
            raise RuntimeError("pdmp3_amd_bulk_wait: engine failure")

    def decode_many(self, mp3s):
        """-> list of int16 PCM arrays, one per stream, each exactly the CLI driver's output for its bytes; the
        streams go through the pipeline back to back (pdmp3_amd_bulk_decode_async)."""
        outs = []
        for m in mp3s:
            total, _ = scan_buffer(m, getattr(self, "iso", 0))
            out = np.empty(max(total, 2) // 2, dtype=np.int16)
            got, _, _ = self.decode_into_async(m, out)
            assert got == total, (got, total)
            outs.append(out[:total // 2])
        self.wait()
        return outs

    def decode(self, mp3):
        """-> interleaved int16 PCM (numpy), exactly the CLI driver's output for these bytes."""
        total, _ = scan_buffer(mp3, getattr(self, "iso", 0))
        out = np.empty(max(total, 2) // 2, dtype=np.int16)
        got, rate, ch = self.decode_into(mp3, out)
        assert got == total, (got, total)
        return out[:total // 2]

    def parse(self, mp3):
        _, frames = scan_buffer(mp3)
        cap = frames + 1                           # frames of a failed last read are parsed too
        sp = np.zeros((cap, 2, 2, 576), dtype=np.int16)
        sd = np.zeros((cap, 2, 2), dtype=SIDE_DTYPE)
        a = _as_u8(mp3)
        pcm_bytes = C.c_longlong(0)
        n = self.lib.pdmp3_amd_bulk_parse(self.h, a.ctypes.data_as(C.c_void_p), len(mp3), sp.ctypes.data_as(C.c_void_p),
                                          sd.ctypes.data_as(C.c_void_p), cap, C.byref(pcm_bytes))
        if n < 0:
            raise RuntimeError("pdmp3_amd_bulk_parse failed")
        return sp[:n], sd[:n], pcm_bytes.value


def parse_bits(mp3, iso=0):
    """Stage A of the bulk pipeline alone: per frame the side info (pdmp3_frame_bits) and the reservoir snapshot
    that pdmp3_hip_stream_submit_bits is given.  No GPU."""
    lib = load_library()
    lib.pdmp3_amd_bulk_set_quirks.argtypes = [C.c_void_p, C.c_uint]
    _, frames = scan_buffer(mp3)
    cap = frames + 1
    bits = np.zeros(cap, dtype=FRAME_BITS_DTYPE)
    res = np.zeros((cap, RESERVOIR_BYTES), dtype=np.uint8)
    assert FRAME_BITS_DTYPE.itemsize == 80
    a = _as_u8(mp3)
    h = lib.pdmp3_amd_bulk_new_parse_bits()
    lib.pdmp3_amd_bulk_set_quirks(h, iso)
    pcm_bytes = C.c_longlong(0)
    n = lib.pdmp3_amd_bulk_parse_bits(h, a.ctypes.data_as(C.c_void_p), len(mp3), bits.ctypes.data_as(C.c_void_p),
                                      res.ctypes.data_as(C.c_void_p), cap, C.byref(pcm_bytes))
    lib.pdmp3_amd_bulk_delete(h)
    if n < 0:
        raise RuntimeError("pdmp3_amd_bulk_parse_bits failed")
    return bits[:n], res[:n], pcm_bytes.value


ROW_DESC_DTYPE = np.dtype([("row_off", "<u4"), ("s_off", "<u4"), ("top", "<u2"), ("back", "<u2"), ("up", "<u2"), ("reserved", "<u2")])


def parse_pool(mp3):
    """Stage A alone in the compact form the engine is given: (bits, row descriptors, pool).  No GPU."""
    lib = load_library()
    _, frames = scan_buffer(mp3)
    cap = frames + 1
    bits = np.zeros(cap, dtype=FRAME_BITS_DTYPE)
    desc = np.zeros(cap, dtype=ROW_DESC_DTYPE)
    pool = np.zeros(cap * 2064 + 16384, dtype=np.uint8)
    a = _as_u8(mp3)
    h = lib.pdmp3_amd_bulk_new_parse_bits()
    used = C.c_size_t(0)
    lib.pdmp3_amd_bulk_parse_pool.restype = C.c_longlong
    lib.pdmp3_amd_bulk_parse_pool.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                              C.c_size_t, C.POINTER(C.c_size_t)]
    n = lib.pdmp3_amd_bulk_parse_pool(h, a.ctypes.data_as(C.c_void_p), len(mp3), bits.ctypes.data_as(C.c_void_p),
                                      desc.ctypes.data_as(C.c_void_p), pool.ctypes.data_as(C.c_void_p), pool.nbytes, cap, C.byref(used))
    lib.pdmp3_amd_bulk_delete(h)
    if n < 0:
        raise RuntimeError("pdmp3_amd_bulk_parse_pool failed")
    return bits[:n], desc[:n], pool[:used.value]


class PinnedPCM:
    """int16 numpy view of a pinned host buffer (pdmp3_amd_pcm_alloc): decode into it and the GPU writes it directly"""

    def __init__(self, n_samples):
        self.lib = load_library()
        self.nbytes = max(2, int(n_samples) * 2)
        self.ptr = self.lib.pdmp3_amd_pcm_alloc(self.nbytes)
        if not self.ptr:
            raise MemoryError("pdmp3_amd_pcm_alloc(%d)" % self.nbytes)
        self.array = np.ctypeslib.as_array((C.c_int16 * (self.nbytes // 2)).from_address(self.ptr))

    def free(self):
        if self.ptr:
            self.array = None
            self.lib.pdmp3_amd_pcm_free(self.ptr)
            self.ptr = None


def corpus_assign(sizes, world):
    """pdmp3_amd_corpus_assign (include/pdmp3_bulk.h): device slot per file, largest first -- == sharding.assign_files; no GPU needed"""
    lib = load_library()
    n = len(sizes)
    arr = (C.c_size_t * n)(*[int(x) for x in sizes])
    out = (C.c_int * n)()
    lib.pdmp3_amd_corpus_assign.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.pdmp3_amd_corpus_assign.restype = None
    lib.pdmp3_amd_corpus_assign(arr, n, world, out)
    return list(out)


def corpus_decode(devices, mp3s, iso=0, threads=2, window_frames=0, host_huffman=False):
    """pdmp3_amd_corpus_decode: whole files over the devices of a node from C (one decoder and host thread per entry of
    `devices`) -> list of int16 arrays"""
    lib = load_library()
    n = len(mp3s)
    bufs = [_as_u8(m) for m in mp3s]
    totals = [max(0, scan_buffer(m, iso)[0]) for m in mp3s]
    outs = [np.empty(max(t, 2) // 2, dtype=np.int16) for t in totals]
    PP = C.c_void_p * n
    srcs = PP(*[b.ctypes.data for b in bufs])
    dsts = PP(*[o.ctypes.data for o in outs])
    sizes = (C.c_size_t * n)(*[len(m) for m in mp3s])
    caps = (C.c_size_t * n)(*[o.nbytes for o in outs])
    got = (C.c_longlong * n)()
    dev = (C.c_int * len(devices))(*[int(d) for d in devices])
    lib.pdmp3_amd_corpus_decode.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_uint, C.c_int, C.c_int, C.c_int]
    rc = lib.pdmp3_amd_corpus_decode(dev, len(devices), srcs, sizes, n, dsts, caps, got, iso, threads, window_frames, int(host_huffman))
    if rc != 0:
        raise RuntimeError("pdmp3_amd_corpus_decode failed")
    assert list(got) == totals, (list(got), totals)
    return [o[:t // 2] for o, t in zip(outs, totals)]

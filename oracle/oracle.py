"""ctypes bindings for the CPU oracle (oracle/liboracle.so) and, when present,
the compiled reference (oracle/_ref/libpdmp3_ref.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (pdmp3_amd/) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SIDE_DTYPE = np.dtype([
    ("count1", "<u2"), ("global_gain", "u1"), ("flags", "u1"),
    ("subblock_gain", "u1", (3,)), ("frame", "u1"),
    ("scalefac_l", "u1", (22,)), ("scalefac_s", "u1", (13, 3)),
    ("iso", "u1"), ("lsf", "u1"), ("lsf_slen", "u1", (4,)), ("lsf_nsfb", "u1", (4,)), ("reserved", "u1", (49,)),
])
assert SIDE_DTYPE.itemsize == 128

GC_SCALEFAC_SCALE, GC_PREFLAG, GC_WIN_SWITCH, GC_MIXED = 0x01, 0x02, 0x04, 0x20
GC_BLOCK_TYPE_SHIFT = 3
FR_MODE_SHIFT, FR_MODEEXT_SHIFT, FR_RESET = 2, 4, 0x40
SF_PEEK = 0xFF


def build(force=False):
    """Compile liboracle.so (and _ref when /root/reference is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(so)
            for f in ("pdmp3_oracle.c", "pdmp3_oracle_stream.c", "pdmp3_oracle.h", "oracle_tables.h")):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/pdmp3.c"):
        ref = os.path.join(_HERE, "_ref", "libpdmp3_ref.so")
        if force or not os.path.exists(ref) or \
                os.path.getmtime(os.path.join(_HERE, "ref_harness.c")) > os.path.getmtime(ref):
            subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_p = C.c_void_p


def _ptr(a):
    return a.ctypes.data_as(_p) if a is not None else None


class Oracle:
    def __init__(self):
        build()
        self.lib = C.CDLL(os.path.join(_HERE, "liboracle.so"))
        L = self.lib
        L.orc_decode_frames.argtypes = [_p, _p, _p, C.c_int, _p, _p]
        L.orc_decode_frames_f32.argtypes = [_p, _p, _p, C.c_int, _p, _p, _p]
        L.orc_generate_frames.argtypes = [C.c_uint64, C.c_int64, C.c_int, _p, _p]
        L.orc_table_pow43.restype = C.POINTER(C.c_float)
        L.orc_table_nwin.restype = C.POINTER(C.c_float)
        L.orc_decode_buffer_like_cli.restype = C.c_size_t
        L.orc_decode_buffer_like_cli.argtypes = [_p, C.c_size_t, _p, C.c_size_t, _p]
        L.orc_time_decode.restype = C.c_double
        L.orc_time_decode.argtypes = [_p, _p, C.c_int, C.c_int]

    def new_state(self):
        return np.zeros(2 * 32 * 18 + 2 * 1024, dtype=np.float32)

    def generate(self, seed, first_frame, n_frames):
        spectra = np.zeros((n_frames, 2, 2, 576), dtype=np.int16)
        side = np.zeros((n_frames, 2, 2), dtype=SIDE_DTYPE)
        self.lib.orc_generate_frames(seed, first_frame, n_frames, _ptr(spectra), _ptr(side))
        return spectra, side

    def decode(self, spectra, side, state=None, stages=False):
        n = spectra.shape[0]
        spectra = np.ascontiguousarray(spectra, dtype=np.int16)
        side = np.ascontiguousarray(side)
        if state is None:
            state = self.new_state()
        pcm = np.zeros((n, 2304), dtype=np.int16)
        stg = np.zeros((n, 2, 2, 4, 576), dtype=np.float32) if stages else None
        self.lib.orc_decode_frames(_ptr(state), _ptr(spectra), _ptr(side), n, _ptr(pcm), _ptr(stg))
        return (pcm, stg) if stages else pcm

    def decode_f32(self, spectra, side, state=None):
        """-> (int16 PCM, float PCM): the float is the binary32 `sum` of P:2028, i.e. int16 = clip(trunc(sum * 32767))"""
        n = spectra.shape[0]
        spectra = np.ascontiguousarray(spectra, dtype=np.int16)
        side = np.ascontiguousarray(side)
        if state is None:
            state = self.new_state()
        pcm = np.zeros((n, 2304), dtype=np.int16)
        f32 = np.zeros((n, 2304), dtype=np.float32)
        self.lib.orc_decode_frames_f32(_ptr(state), _ptr(spectra), _ptr(side), n, _ptr(pcm), _ptr(f32), None)
        return pcm, f32

    def time_decode(self, spectra, side, reps=1):
        n = spectra.shape[0]
        return self.lib.orc_time_decode(_ptr(spectra), _ptr(side), n, reps)

    def huffman_quad(self, iso, bits, nbits):
        """one count1 quadruple through the restatement's huffman_decode with the quirk mask `iso` (table number 33):
        ((v, w, x, y), bits consumed, status)"""
        out = (C.c_int * 5)()
        res = self.lib.orc_huffman_quad(C.c_uint(iso), C.c_uint(bits), C.c_int(nbits), out)
        return tuple(out[:4]), out[4], res

    def huffman_nodes(self):
        n = C.c_uint(0)
        self.lib.orc_huffman_nodes.restype = C.POINTER(C.c_uint16)
        p = self.lib.orc_huffman_nodes(C.byref(n))
        return np.ctypeslib.as_array(p, shape=(n.value,)).copy()

    def pow43(self):
        return np.ctypeslib.as_array(self.lib.orc_table_pow43(), shape=(8207,)).copy()

    def nwin(self):
        return np.ctypeslib.as_array(self.lib.orc_table_nwin(), shape=(64, 32)).copy()

    def decode_buffer_like_cli_iso(self, mp3: bytes, iso, tap_frames=0):
        """the CLI loop with the ISO-correct switches `iso` (include/pdmp3.h PDMP3_ISO_*; NOT the reference: unpinned)
        -> (pcm bytes, spectra, side records of up to tap_frames frames)"""
        s = OracleStream(self)
        s.set_quirks(iso)
        try:
            return s.decode_like_cli(mp3, tap_frames)
        finally:
            self.last_undefined = s.undefined()      # the stream made the reference's line counter wrap: nothing it decodes to is defined
            s.close()

    def decode_buffer_like_cli(self, mp3: bytes, tap_frames=0):
        buf = np.frombuffer(mp3, dtype=np.uint8)
        cap = (len(mp3) // 96 + 8) * 4608
        pcm = np.zeros(cap, dtype=np.uint8)

        class Tap(C.Structure):
            _fields_ = [("spectra", _p), ("side", _p), ("cap", C.c_int), ("n", C.c_int)]
        tap = None
        if tap_frames:
            sp = np.zeros((tap_frames, 2, 2, 576), dtype=np.int16)
            sd = np.zeros((tap_frames, 2, 2), dtype=SIDE_DTYPE)
            tap = Tap(_ptr(sp), _ptr(sd), tap_frames, 0)
        n = self.lib.orc_decode_buffer_like_cli(_ptr(buf), len(mp3), _ptr(pcm), cap,
                                                C.byref(tap) if tap else None)
        out = pcm[:n].tobytes()
        if tap_frames:
            k = min(tap.n, tap_frames)
            return out, sp[:k], sd[:k]
        return out


class OracleStream:
    """The oracle's restatement of the streaming API (orc_stream_*), driven like
    pdmp3_amd.api.Decoder so that call sequences can be replayed on both."""

    def __init__(self, oracle):
        self.L = oracle.lib
        self.L.orc_stream_new.restype = C.c_void_p
        self.L.orc_stream_open_feed.argtypes = [_p]
        self.L.orc_stream_feed.argtypes = [_p, _p, C.c_size_t]
        self.L.orc_stream_read.argtypes = [_p, _p, C.c_size_t, C.POINTER(C.c_size_t)]
        self.L.orc_stream_delete.argtypes = [_p]
        self.s = self.L.orc_stream_new()
        self.L.orc_stream_open_feed(self.s)

    def close(self):
        self.L.orc_stream_delete(self.s)

    # the seven API calls, with pdmp3_amd.api.Decoder's signatures
    def open_feed(self):
        return self.L.orc_stream_open_feed(self.s)

    def feed(self, data: bytes):
        buf = (C.c_ubyte * len(data)).from_buffer_copy(data)
        return self.L.orc_stream_feed(self.s, buf, len(data))

    def read(self, outsize):
        out = (C.c_ubyte * outsize)()
        done = C.c_size_t(0)
        rc = self.L.orc_stream_read(self.s, out, outsize, C.byref(done))
        return rc, bytes(out[:done.value])

    def decode(self, data: bytes, outsize):
        self.L.orc_stream_decode.argtypes = [_p, _p, C.c_size_t, _p, C.c_size_t, C.POINTER(C.c_size_t)]
        buf = (C.c_ubyte * max(1, len(data))).from_buffer_copy(data or b"\0")
        out = (C.c_ubyte * outsize)() if outsize else None
        done = C.c_size_t(0)
        rc = self.L.orc_stream_decode(self.s, buf, len(data), out, outsize, C.byref(done))
        return rc, bytes(out[:done.value]) if outsize else b""

    def getformat(self):
        self.L.orc_stream_getformat.argtypes = [_p, C.POINTER(C.c_long), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        rate, ch, enc = C.c_long(0), C.c_int(0), C.c_int(0)
        rc = self.L.orc_stream_getformat(self.s, C.byref(rate), C.byref(ch), C.byref(enc))
        return rc, rate.value, ch.value, enc.value

    def set_quirks(self, iso_mask):
        self.L.orc_stream_set_quirks.argtypes = [_p, C.c_uint]
        self.L.orc_stream_set_quirks(self.s, iso_mask)

    def undefined(self):
        self.L.orc_stream_undefined.argtypes = [_p]
        self.L.orc_stream_undefined.restype = C.c_int
        return bool(self.L.orc_stream_undefined(self.s))

    def decode_like_cli(self, mp3: bytes, tap_frames=0):
        """tap_frames > 0: -> (pcm bytes, spectra, side) with the records of the first tap_frames frames"""
        if tap_frames:
            class Tap(C.Structure):
                _fields_ = [("spectra", _p), ("side", _p), ("cap", C.c_int), ("n", C.c_int)]
            sp = np.zeros((tap_frames, 2, 2, 576), dtype=np.int16)
            sd = np.zeros((tap_frames, 2, 2), dtype=SIDE_DTYPE)
            tap = Tap(_ptr(sp), _ptr(sd), tap_frames, 0)
            self.L.orc_stream_set_tap.argtypes = [_p, _p]
            self.L.orc_stream_set_tap(self.s, C.byref(tap))
            try:
                pcm = self.decode_like_cli(mp3)
            finally:
                self.L.orc_stream_set_tap(self.s, None)
            k = min(tap.n, tap_frames)
            return pcm, sp[:k], sd[:k]
        self.L.orc_stream_open_feed(self.s)
        out, pos = [], 0
        buf = (C.c_ubyte * 16384)()
        done = C.c_size_t(0)
        while True:
            rc = self.L.orc_stream_read(self.s, buf, 16384, C.byref(done))
            if rc == -1:
                break
            out.append(bytes(buf[:done.value]))
            if rc == -10:
                chunk = mp3[pos:pos + 4096]
                if not chunk:
                    break
                cb = (C.c_ubyte * len(chunk)).from_buffer_copy(chunk)
                self.L.orc_stream_feed(self.s, cb, len(chunk))
                pos += len(chunk)
        return b"".join(out)


def have_ref():
    return os.path.exists(os.path.join(_HERE, "_ref", "libpdmp3_ref.so"))


class Reference:
    """The real reference decoder (oracle/_ref/libpdmp3_ref.so).  Its synthesis
    state is process-global: one stream at a time."""

    def __init__(self):
        self.lib = C.CDLL(os.path.join(_HERE, "_ref", "libpdmp3_ref.so"))
        L = self.lib
        L.ref_new.restype = _p
        L.ref_delete.argtypes = [_p]
        L.ref_decode_frames.argtypes = [_p, _p, _p, C.c_int, _p, _p]
        L.ref_time_decode.restype = C.c_double
        L.ref_time_decode.argtypes = [_p, _p, _p, C.c_int, C.c_int]
        L.ref_decode_buffer_like_cli.restype = C.c_size_t
        L.ref_decode_buffer_like_cli.argtypes = [_p, C.c_size_t, _p, C.c_size_t, _p, _p, C.c_int, _p]
        self.h = L.ref_new()

    def decode(self, spectra, side, stages=False):
        n = spectra.shape[0]
        spectra = np.ascontiguousarray(spectra, dtype=np.int16)
        side = np.ascontiguousarray(side)
        pcm = np.zeros((n, 2304), dtype=np.int16)
        stg = np.zeros((n, 2, 2, 4, 576), dtype=np.float32) if stages else None
        self.lib.ref_decode_frames(self.h, _ptr(spectra), _ptr(side), n, _ptr(pcm), _ptr(stg))
        return (pcm, stg) if stages else pcm

    def time_decode(self, spectra, side, reps=1):
        return self.lib.ref_time_decode(self.h, _ptr(spectra), _ptr(side), spectra.shape[0], reps)

    def huffman_table_words(self, first, n):
        """words [first, first + n) of the reference's g_huffman_table (P:235-515), from its memory"""
        out = np.zeros(n, dtype=np.uint16)
        self.lib.ref_huffman_table_words.restype = C.c_uint
        total = self.lib.ref_huffman_table_words(C.c_uint(first), C.c_uint(n), _ptr(out))
        return out, int(total)

    def huffman_quad_at(self, offset, bits, nbits):
        """the reference's own Huffman_Decode (P:1593-1643) on table number 33 with its tree pointer set `offset`
        words into g_huffman_table: ((v, w, x, y), bits consumed, status)"""
        out = (C.c_int * 5)()
        res = self.lib.ref_huffman_quad_at(C.c_uint(offset), C.c_uint(bits), C.c_int(nbits), out)
        return tuple(out[:4]), out[4], res

    def decode_buffer_like_cli(self, mp3: bytes, tap_frames=0):
        buf = np.frombuffer(mp3, dtype=np.uint8)
        cap = (len(mp3) // 96 + 8) * 4608
        pcm = np.zeros(cap, dtype=np.uint8)
        if tap_frames:
            sp = np.zeros((tap_frames, 2, 2, 576), dtype=np.int16)
            sd = np.zeros((tap_frames, 2, 2), dtype=SIDE_DTYPE)
            nt = C.c_int(0)
            n = self.lib.ref_decode_buffer_like_cli(_ptr(buf), len(mp3), _ptr(pcm), cap,
                                                    _ptr(sp), _ptr(sd), tap_frames, C.byref(nt))
            k = min(nt.value, tap_frames)
            return pcm[:n].tobytes(), sp[:k], sd[:k]
        n = self.lib.ref_decode_buffer_like_cli(_ptr(buf), len(mp3), _ptr(pcm), cap, None, None, 0, None)
        return pcm[:n].tobytes()

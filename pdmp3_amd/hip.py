"""ctypes mirror of include/pdmp3_hip.h.

Every method maps 1:1 to a C-ABI entry point; tensors are used only as owners
of device memory (`tensor.data_ptr()`), and the launch goes onto torch's
current HIP stream so that torch.cuda.Event timing brackets it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SIDE_DTYPE = np.dtype([
    ("count1", "<u2"), ("global_gain", "u1"), ("flags", "u1"),
    ("subblock_gain", "u1", (3,)), ("frame", "u1"),
    ("scalefac_l", "u1", (22,)), ("scalefac_s", "u1", (13, 3)),
    ("iso", "u1"), ("lsf", "u1"), ("lsf_slen", "u1", (4,)), ("lsf_nsfb", "u1", (4,)), ("reserved", "u1", (49,)),
])
FRAME_SPECTRA_INT16 = 4 * 576
FRAME_PCM_INT16 = 2304
FRAME_SIDE_BYTES = 512

EXPORTS = [
    "pdmp3_hip_create", "pdmp3_hip_destroy", "pdmp3_hip_last_error", "pdmp3_hip_state_bytes", "pdmp3_hip_last_launch_kind", "pdmp3_hip_pci_bus_id",
    "pdmp3_hip_decode_frames", "pdmp3_hip_decode_frames_f32", "pdmp3_hip_decode_frames_stages", "pdmp3_hip_generate_frames",
    "pdmp3_hip_decode_lsf_frames", "pdmp3_hip_decode_lsf_frames_f32", "pdmp3_hip_stream_set_lsf",
    "pdmp3_host_generate_frames",
    "pdmp3_hip_stream_create", "pdmp3_hip_stream_destroy", "pdmp3_hip_stream_reset", "pdmp3_hip_stream_spectra",
    "pdmp3_hip_stream_side", "pdmp3_hip_stream_pcm", "pdmp3_hip_stream_decode",
]


def library_path():
    # PDMP3_HIP_LIB: alternative build of the engine library (A/B experiments)
    return os.environ.get("PDMP3_HIP_LIB") or os.path.join(_HERE, "libpdmp3_hip.so")


def build_library(force=False):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    if force and os.path.exists(library_path()):
        os.remove(library_path())
    subprocess.check_call(["make", "-C", src_dir], stdout=subprocess.DEVNULL)
    return library_path()


def load_library():
    global _LIB
    if _LIB is not None:
        return _LIB
    # torch bundles its own HIP runtime; import it first so that both share ONE
    # runtime instance in this process (a second copy does not see the device).
    import torch  # noqa: F401
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            "pdmp3_amd: %s is missing -- build it with __graft_entry__.build() or "
            "`make -C pdmp3_amd/csrc`; there is no CPU fallback" % path)
    lib = C.CDLL(path)
    vp, i32, i64, u64 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64
    lib.pdmp3_hip_create.argtypes = [i32, C.POINTER(vp)]
    lib.pdmp3_hip_destroy.argtypes = [vp]
    lib.pdmp3_hip_last_error.restype = C.c_char_p
    lib.pdmp3_hip_state_bytes.restype = C.c_size_t
    if hasattr(lib, "pdmp3_hip_last_launch_kind"):
        lib.pdmp3_hip_last_launch_kind.argtypes = [vp]
    lib.pdmp3_hip_decode_frames.argtypes = [vp, vp, vp, i32, vp, vp, i32, vp]
    lib.pdmp3_hip_decode_frames_f32.argtypes = [vp, vp, vp, i32, vp, vp, i32, vp]
    lib.pdmp3_hip_decode_frames_stages.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp]
    lib.pdmp3_hip_generate_frames.argtypes = [vp, u64, i64, i32, vp, vp, vp]
    lib.pdmp3_host_generate_frames.argtypes = [u64, i64, i32, vp, vp]
    _LIB = lib
    return lib


NODE_EXPORTS = ["pdmp3_node_create", "pdmp3_node_destroy", "pdmp3_node_ranks", "pdmp3_node_shard",
                "pdmp3_node_decode_records", "pdmp3_node_decode_generated"]
NODE_RCCL, NODE_COPY = 0, 1


class NodeTiming(C.Structure):
    _fields_ = [("prepare_ms", C.c_double), ("decode_ms", C.c_double), ("gather_ms", C.c_double),
                ("gather_bytes", C.c_longlong), ("rccl_ranks", C.c_int), ("slices", C.c_int), ("total_ms", C.c_double)]


def node_shard(n_frames, rank, world, frame_flags=None):
    """pdmp3_node_shard (include/pdmp3_node.h): (first frame to decode, frames to decode, frames to discard) -- the C side of
    pdmp3_amd.sharding.shard_with_halo; a pure function, no GPU needed"""
    lib = load_library()
    lib.pdmp3_node_shard.argtypes = [C.c_longlong, C.c_int, C.c_int, C.c_void_p] + [C.POINTER(C.c_longlong)] * 3
    lib.pdmp3_node_shard.restype = None
    a, b, c = C.c_longlong(), C.c_longlong(), C.c_longlong()
    fl = None
    if frame_flags is not None:
        fl = np.ascontiguousarray(frame_flags, dtype=np.uint8)
    lib.pdmp3_node_shard(n_frames, rank, world, fl.ctypes.data_as(C.c_void_p) if fl is not None else None,
                         C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


class NodeDecoder:
    """include/pdmp3_node.h: one stream decoded by several GPUs of this node from ONE process -- frame-range shards with
    halos, no collective inside the decode, the PCM gathered to devices[0] over RCCL (transport = NODE_RCCL) or, for tests
    that list one GPU several times, by device copies (NODE_COPY)."""

    def __init__(self, devices, transport=NODE_RCCL):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("pdmp3_amd.NodeDecoder needs HIP devices; there is no CPU fallback")
        self.torch = torch
        self.lib = load_library()
        L = self.lib
        L.pdmp3_node_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.pdmp3_node_destroy.argtypes = [C.c_void_p]
        L.pdmp3_node_ranks.argtypes = [C.c_void_p]
        L.pdmp3_node_decode_records.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.POINTER(NodeTiming)]
        L.pdmp3_node_decode_generated.argtypes = [C.c_void_p, C.c_uint64, C.c_longlong, C.c_void_p, C.POINTER(NodeTiming)]
        self.devices = [int(d) for d in devices]
        arr = (C.c_int * len(self.devices))(*self.devices)
        h = C.c_void_p()
        rc = L.pdmp3_node_create(arr, len(self.devices), int(transport), C.byref(h))
        if rc != 0:
            raise RuntimeError("pdmp3_node_create: error %d: %s" % (rc, L.pdmp3_hip_last_error().decode()))
        self.h = h
        self.tdev = torch.device("cuda", self.devices[0])

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s: error %d: %s" % (what, rc, self.lib.pdmp3_hip_last_error().decode()))

    def decode_generated(self, seed, n_frames):
        """-> (int16 tensor [n_frames, 2304] on devices[0], NodeTiming)"""
        t = self.torch
        pcm = t.empty((n_frames, FRAME_PCM_INT16), dtype=t.int16, device=self.tdev)
        tm = NodeTiming()
        t.cuda.synchronize(self.tdev)
        self._check(self.lib.pdmp3_node_decode_generated(self.h, seed, n_frames, pcm.data_ptr(), C.byref(tm)), "pdmp3_node_decode_generated")
        return pcm, tm

    def decode_records(self, spectra, side):
        """spectra / side: numpy arrays in host memory (the layout of Engine.upload) -> (PCM tensor on devices[0], NodeTiming)"""
        t = self.torch
        sp = np.ascontiguousarray(spectra, dtype=np.int16)
        sd = np.ascontiguousarray(side)
        n = sp.shape[0]
        pcm = t.empty((n, FRAME_PCM_INT16), dtype=t.int16, device=self.tdev)
        tm = NodeTiming()
        t.cuda.synchronize(self.tdev)
        self._check(self.lib.pdmp3_node_decode_records(self.h, sp.ctypes.data_as(C.c_void_p), sd.ctypes.data_as(C.c_void_p), n,
                                                       pcm.data_ptr(), C.byref(tm)), "pdmp3_node_decode_records")
        return pcm, tm

    def close(self):
        if getattr(self, "h", None):
            self.lib.pdmp3_node_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def host_generate(seed, first_frame, n_frames):
    """pdmp3_host_generate_frames: the SURVEY 8d generator on the host (no GPU needed)."""
    lib = load_library()
    spectra = np.zeros((n_frames, 2, 2, 576), dtype=np.int16)
    side = np.zeros((n_frames, 2, 2), dtype=SIDE_DTYPE)
    rc = lib.pdmp3_host_generate_frames(seed, first_frame, n_frames,
                                        spectra.ctypes.data_as(C.c_void_p), side.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise RuntimeError(lib.pdmp3_hip_last_error().decode())
    return spectra, side


class Engine:
    """One pdmp3_hip_ctx on one GPU."""

    def __init__(self, device=0):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("pdmp3_amd.Engine needs a HIP device; there is no CPU fallback")
        self.torch = torch
        self.lib = load_library()
        self.device = int(device)
        h = C.c_void_p()
        self._check(self.lib.pdmp3_hip_create(self.device, C.byref(h)))
        self.h = h
        self.tdev = torch.device("cuda", self.device)

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError("pdmp3_hip error %d: %s" % (rc, self.lib.pdmp3_hip_last_error().decode()))

    def close(self):
        if getattr(self, "h", None):
            self.lib.pdmp3_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.tdev).cuda_stream)

    def last_launch_kernel(self):
        """name of the kernel the latest decode launch ran (pdmp3_hip_last_launch_kind)"""
        k = self.lib.pdmp3_hip_last_launch_kind(self.h)
        return {1: "k_decode (independent chunks, halo)", 8: "k_decode_g<.., 8> (one granule per wave, 8 waves per workgroup)",
                16: "k_decode_g<.., 16> (one granule per wave, 16 waves per workgroup)",
                32: "k_decode_p (persistent: 16 waves per workgroup going round a range of frames)"}.get(k, "none")

    # -- device buffers ----------------------------------------------------
    def alloc_frames(self, n_frames):
        t = self.torch
        spectra = t.empty((n_frames, 2, 2, 576), dtype=t.int16, device=self.tdev)
        side = t.zeros((n_frames, 4, 128), dtype=t.uint8, device=self.tdev)
        pcm = t.empty((n_frames, FRAME_PCM_INT16), dtype=t.int16, device=self.tdev)
        return spectra, side, pcm

    def new_state(self):
        t = self.torch
        return t.zeros(self.lib.pdmp3_hip_state_bytes() // 4, dtype=t.float32, device=self.tdev)

    def upload(self, spectra_np, side_np):
        t = self.torch
        sp = t.from_numpy(np.ascontiguousarray(spectra_np, dtype=np.int16)).to(self.tdev)
        sd = t.from_numpy(np.ascontiguousarray(side_np).view(np.uint8).reshape(-1, 4, 128).copy()).to(self.tdev)
        return sp, sd

    # -- C-ABI calls ---------------------------------------------------------
    def generate(self, seed, first_frame, n_frames, spectra, side):
        self._check(self.lib.pdmp3_hip_generate_frames(self.h, seed, first_frame, n_frames,
                                                       spectra.data_ptr(), side.data_ptr(), self._stream()))

    def decode(self, spectra, side, pcm, n_frames=None, state=None, chunk_frames=0):
        n = int(spectra.shape[0]) if n_frames is None else int(n_frames)
        self._check(self.lib.pdmp3_hip_decode_frames(
            self.h, spectra.data_ptr(), side.data_ptr(), n,
            state.data_ptr() if state is not None else None, pcm.data_ptr(), int(chunk_frames), self._stream()))

    def decode_lsf(self, spectra, side, pcm, n_frames=None, state=None):
        """pdmp3_hip_decode_lsf_frames: n MPEG-2 LSF / 2.5 frames (records with lsf != 0, one channel count) -> PCM in stream
        order, 576 sample-frames per frame (layout: include/pdmp3_hip.h); pcm int16 or float32 tensor"""
        n = int(spectra.shape[0]) if n_frames is None else int(n_frames)
        f = self.lib.pdmp3_hip_decode_lsf_frames_f32 if pcm.dtype == self.torch.float32 else self.lib.pdmp3_hip_decode_lsf_frames
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        self._check(f(self.h, spectra.data_ptr(), side.data_ptr(), n, state.data_ptr() if state is not None else None, pcm.data_ptr(), self._stream()))

    def has_persistent_kernel(self):
        """was the library built with -DPDMP3_WITH_RING_KERNEL (k_decode_p, an opt-in since round 5)?  Asked by trying:
        a launch with chunk_frames = PDMP3_HIP_CHUNK_PERSISTENT fails cleanly in a build without it."""
        import torch
        sp, sd, pcm = self.alloc_frames(16)
        sp.zero_(); sd.zero_()
        rc = self.lib.pdmp3_hip_decode_frames(self.h, sp.data_ptr(), sd.data_ptr(), 16, None, pcm.data_ptr(), -3, self._stream())
        torch.cuda.synchronize()
        return rc == 0

    def decode_f32(self, spectra, side, pcm_f32, n_frames=None, state=None, chunk_frames=0):
        """float PCM (pdmp3_hip_decode_frames_f32): pcm_f32 = float32 tensor, 2304 floats per frame"""
        n = int(spectra.shape[0]) if n_frames is None else int(n_frames)
        self._check(self.lib.pdmp3_hip_decode_frames_f32(
            self.h, spectra.data_ptr(), side.data_ptr(), n,
            state.data_ptr() if state is not None else None, pcm_f32.data_ptr(), int(chunk_frames), self._stream()))

    def decode_stages(self, spectra, side, pcm, stages, state=None):
        n = int(spectra.shape[0])
        self._check(self.lib.pdmp3_hip_decode_frames_stages(
            self.h, spectra.data_ptr(), side.data_ptr(), n,
            state.data_ptr() if state is not None else None, pcm.data_ptr(), stages.data_ptr(), self._stream()))

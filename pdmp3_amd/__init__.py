"""pdmp3_amd -- MI355X (gfx950) transform engine for the PDMP3 Layer-III hot path.

The product is the C-ABI shared library `pdmp3_amd/libpdmp3_hip.so`
(include/pdmp3_hip.h).  This package is only the thin Python mirror used by
bench.py and the tests: ctypes bindings that pass raw device pointers (from
torch tensors: plumbing for device memory, streams and torch.distributed) into
the library.  There is NO CPU fallback here: if the library is missing or
there is no GPU, calls raise.
"""
from .hip import (Engine, NodeDecoder, SIDE_DTYPE, FRAME_PCM_INT16, FRAME_SPECTRA_INT16,  # noqa: F401
                  library_path, load_library, build_library)

__all__ = ["Engine", "NodeDecoder", "SIDE_DTYPE", "library_path", "load_library", "build_library"]

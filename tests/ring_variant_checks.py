"""GPU parity tests of the persistent granule kernel k_decode_p (decode_core.h run_granule_ring), which only the VARIANT
library pdmp3_amd/libpdmp3_hip_ring.so carries (pdmp3_amd/csrc/Makefile).  Not collected by a plain `pytest tests`
(the file name does not match): tests/test_gpu_ring_variant.py runs this file in a child process whose PDMP3_HIP_LIB
points at the variant, so that every -m gpu run exercises the kernel and nothing is skipped."""
import numpy as np
import pytest

import corpus
from test_gpu_parity import gpu_decode
from util import assert_pcm_close, nch_of

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["ms_mixed_blocks_441", "ms_short_heavy_480", "mono_441", "mono_320", "ms_resets", "dual_480_fs", "ms_is_short_480_fs"])
def test_gpu_persistent_granule_kernel_equals_independent_chunks(engine, name):
    """k_decode_p (chunk_frames = PDMP3_HIP_CHUNK_PERSISTENT: 16 waves going round a range of frames, ring of LDS mailboxes, one
    halo per range) on the device against independent chunks: PCM and carried state bit-identical -- H5-heavy, mono (every
    frame through run_chunk inside the loop), RESET frames, the stereo / mono / stereo stream cut by range boundaries; the
    engine makes ranges of >= 8 frames, so 64-frame corpora are 8 workgroups"""
    import torch
    from test_pipeline_emul import _mode_switch_records
    assert engine.has_persistent_kernel(), "run through tests/test_gpu_ring_variant.py (PDMP3_HIP_LIB = the library built with the kernel)"
    for sp, sd in (corpus.case(name), _mode_switch_records()):
        dsp, dsd = engine.upload(sp, sd)
        n = sp.shape[0]
        a = torch.zeros((n, 2304), dtype=torch.int16, device=engine.tdev)
        b = torch.zeros_like(a)
        sa, sb = engine.new_state(), engine.new_state()
        engine.decode(dsp, dsd, a, state=sa, chunk_frames=-3)
        assert "k_decode_p" in engine.last_launch_kernel()
        engine.decode(dsp, dsd, b, state=sb, chunk_frames=3)
        torch.cuda.synchronize()
        assert torch.equal(a, b) and torch.equal(sa, sb), name


@pytest.mark.parametrize("name", ["iso_ms_all_441", "iso_std_ms_is_short_480", "iso_std_ms_is_mixed_320"])
def test_gpu_persistent_kernel_on_iso_records(engine, oracle, name):
    sp, sd = corpus.case(name)
    want = oracle.decode(sp, sd)
    got = gpu_decode(engine, sp, sd, chunk=-3)
    assert "k_decode_p" in engine.last_launch_kernel()
    assert_pcm_close(got, want, 1, name)
    assert np.array_equal(got, gpu_decode(engine, sp, sd, chunk=3))

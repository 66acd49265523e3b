"""Host build of the device pipeline (pdmp3_amd/csrc/decode_core.h compiled by
g++, tests/host_emul: the SAME source the GPU runs, one wave = 64 fibers, the matrix
instruction evaluated with the hardware's fragment layout as k-ordered fmaf chains)
against the oracle: validates the kernel's indexing, tables, fragment layouts,
chunk/halo logic and state hand-off without a GPU.  Same bars as the GPU parity
tests (test_gpu_parity.py), which go through the C-ABI on the real device."""
import ctypes as C

import numpy as np
import pytest

import corpus
from conftest import C2_SEED
from util import assert_pcm_close, nch_of


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def emul_decode(emul, sp, sd, chunk=0, state=None, stages=False):
    n = sp.shape[0]
    pcm = np.zeros((n, 2304), np.int16)
    stg = np.zeros((n, 2, 2, 4, 576), np.float32) if stages else None
    emul.emul_decode_frames(_p(sp), _p(sd), n, _p(state), _p(pcm), _p(stg), chunk)
    return (pcm, stg) if stages else pcm


@pytest.mark.parametrize("name", list(corpus.ALL_CASES))
def test_emul_vs_oracle(oracle, emul, name):
    """the bars of test_gpu_parity.py, literally: corpus.PCM_TOL_LSB (1 LSB but for ms_loud_clip: 32), 64 frames"""
    sp, sd = corpus.case(name)
    want, ws = oracle.decode(sp, sd, stages=True)
    got, gs = emul_decode(emul, sp, sd, stages=True)
    nch = nch_of(sd)
    for k in range(3):   # separately rounded mul / add like the reference: bit-exact
        assert np.array_equal(ws[:, :, :nch, k].view(np.uint32), gs[:, :, :nch, k].view(np.uint32)), "stage %d" % k
    amp = max(1.0, float(np.abs(ws[:, :, :nch, 3]).max()))
    err = float(np.abs(ws[:, :, :nch, 3] - gs[:, :, :nch, 3]).max())
    assert err <= 1e-5 * amp, "hybrid output differs by %g (amplitude %g)" % (err, amp)
    assert_pcm_close(got, want, corpus.PCM_TOL_LSB[name], name)
    # the stage dumps come from the general formulation of every phase; without them the wave takes the fast paths
    # (ph_requant_long for granules of long blocks): the same arithmetic per line, so the same PCM bit for bit
    assert np.array_equal(emul_decode(emul, sp, sd), got), "fast paths differ from the general formulation"


@pytest.mark.parametrize("chunk", [1, 2, 3, 5, 16])
def test_emul_chunking_is_invisible(oracle, emul, chunk):
    sp, sd = oracle.generate(C2_SEED, 0, 40)
    whole = emul_decode(emul, sp, sd, 0)
    assert np.array_equal(emul_decode(emul, sp, sd, chunk), whole)


def test_emul_state_handoff(oracle, emul):
    sp, sd = oracle.generate(C2_SEED, 0, 30)
    whole = emul_decode(emul, sp, sd, 0)
    st = np.zeros(emul.emul_state_floats(), np.float32)
    a = emul_decode(emul, sp[:11], sd[:11], 0, st)
    b = emul_decode(emul, sp[11:], sd[11:], 4, st)
    assert np.array_equal(np.concatenate([a, b]), whole)


def test_emul_generator(oracle, emul):
    sp, sd = oracle.generate(C2_SEED, 3, 9)
    sp2 = np.zeros_like(sp)
    sd2 = np.zeros_like(sd)
    emul.emul_generate_frames(C.c_uint64(C2_SEED), C.c_int64(3), 9, _p(sp2), _p(sd2))
    assert np.array_equal(sp, sp2) and np.array_equal(sd.view(np.uint8), sd2.view(np.uint8))


def _mode_switch_records():
    """stereo / mono / stereo runs, incl. single stereo frames between mono runs (records from the host parser)"""
    from pdmp3_amd import api
    from pdmp3_amd.packer import packer
    parts = [dict(n_frames=9, seed=31, bitrate_index=9), dict(n_frames=11, seed=32, mode=3, bitrate_index=7),
             dict(n_frames=2, seed=33, mode=1, mode_ext=2, bitrate_index=11, block_pct=(10, 10, 70, 10)),
             dict(n_frames=7, seed=34, mode=3, bitrate_index=7),
             dict(n_frames=12, seed=35, mode=1, mode_ext=2, bitrate_index=11, block_pct=(10, 10, 70, 10))]
    mp3 = b"".join(packer.generate(**p) for p in parts)
    return api.parse_like_cli(mp3, 64)


@pytest.mark.parametrize("chunk", [1, 2, 3, 4, 7, 16])
def test_emul_channel1_state_survives_mono_runs(oracle, emul, chunk):
    """mono frames leave channel 1's overlap / polyphase history alone (P:1777, P:2126 are per channel): a chunk
    that starts after a mono run must find channel 1 as the last stereo frame left it"""
    sp, sd = _mode_switch_records()
    assert sp.shape[0] >= 38
    want = oracle.decode(sp, sd)
    assert_pcm_close(emul_decode(emul, sp, sd, chunk), want, 1, "chunk %d" % chunk)
    assert np.array_equal(emul_decode(emul, sp, sd, chunk), emul_decode(emul, sp, sd, 0))


def test_emul_channel1_state_across_batches(oracle, emul):
    """the carried state keeps channel 1 through an all-mono batch decoded in several chunks"""
    sp, sd = _mode_switch_records()
    whole = emul_decode(emul, sp, sd, 0)
    st = np.zeros(emul.emul_state_floats(), np.float32)
    cuts = [0, 9, 14, 20, 22, 29, sp.shape[0]]          # 9..20 mono (two batches), 22..29 mono
    out = [emul_decode(emul, sp[a:b], sd[a:b], 2, st) for a, b in zip(cuts[:-1], cuts[1:])]
    assert np.array_equal(np.concatenate(out), whole)
    # batches that end in mono right after stereo frames (pre-halo touching the ordinary halo), every chunk size
    for chunk in (1, 2, 3):
        for cuts in ([0, 7, 11, 19, 24, 31, sp.shape[0]], [0, 8, 10, 21, 23, 30, sp.shape[0]]):
            st[:] = 0
            out = [emul_decode(emul, sp[a:b], sd[a:b], chunk, st) for a, b in zip(cuts[:-1], cuts[1:])]
            assert np.array_equal(np.concatenate(out), whole), (chunk, cuts)


def test_shard_cut_after_mono_frames_reaches_back_to_the_last_stereo_frame(emul):
    """pdmp3_amd.sharding.halo_start: a shard whose cut follows mono frames starts in front of the last stereo frame,
    so that channel 1 is what that frame left (the kernel's pre-halo finds it inside the shard); with the fixed
    2-frame halo alone some of these cuts come out wrong -- the test must bite"""
    from pdmp3_amd.sharding import halo_start
    sp, sd = _mode_switch_records()
    n = sp.shape[0]
    flags = sd["frame"][:, 0, 0] if sd["frame"].ndim == 3 else sd["frame"].reshape(n, -1)[:, 0]
    whole = emul_decode(emul, sp, sd, 0)
    fixed_fails = 0
    for lo in range(1, n):
        first = halo_start(lo, flags)
        assert 0 <= first <= max(0, lo - 2) or first == 0
        part = emul_decode(emul, sp[first:], sd[first:], 3)[lo - first:]
        assert np.array_equal(part, whole[lo:]), (lo, first)
        f2 = max(0, lo - 2)
        if first != f2 and not np.array_equal(emul_decode(emul, sp[f2:], sd[f2:], 3)[lo - f2:], whole[lo:]):
            fixed_fails += 1
    assert fixed_fails > 0
    assert halo_start(lo, None) == lo - 2 and halo_start(0, flags) == 0


def test_oracle_float_pcm_is_what_the_int16_comes_from(oracle):
    """float output is pinned through its quantisation: the oracle's int16 PCM -- bit-exact against the compiled
    reference (test_oracle.py) -- is clip(trunc(float * 32767)) of the float it hands out (P:2028-2031)"""
    sp, sd = oracle.generate(C2_SEED, 0, 24)
    pcm, f32 = oracle.decode_f32(sp, sd)
    assert np.array_equal(pcm, oracle.decode(sp, sd))
    q = np.clip(np.trunc(f32.astype(np.float64) * 32767.0), -32767, 32767).astype(np.int16)
    assert np.array_equal(q, pcm)
    assert np.abs(f32).max() > 0.5


@pytest.mark.parametrize("name", list(corpus.ALL_CASES))
def test_emul_float_pcm(oracle, emul, name):
    """pdmp3_hip_decode_frames_f32 on the host build: 1e-5 absolute (north_star's float tolerance) on the `_fs` cases,
    corpus.F32_TOL_ABS's literal number on the loud ones; chunked == unchunked"""
    sp, sd = corpus.case(name)
    _, want = oracle.decode_f32(sp, sd)
    n = sp.shape[0]
    got = np.zeros((n, 2304), np.float32)
    emul.emul_decode_frames_f32(_p(sp), _p(sd), n, None, _p(got), 0)
    assert float(np.abs(got - want).max()) <= corpus.F32_TOL_ABS[name]
    got2 = np.zeros((n, 2304), np.float32)
    emul.emul_decode_frames_f32(_p(sp), _p(sd), n, None, _p(got2), 3)
    assert np.array_equal(got, got2)


def emul_decode_granules(emul, sp, sd, state=None, f32=False, debug=0, sf_hint=0):
    n = sp.shape[0]
    pcm = np.zeros((n, 2304), np.float32 if f32 else np.int16)
    emul.emul_decode_frames_granules(_p(sp), _p(sd), n, _p(state), None if f32 else _p(pcm), _p(pcm) if f32 else None, debug, sf_hint)
    return pcm


@pytest.mark.parametrize("name", list(corpus.CASES))
def test_emul_granule_waves_equal_independent_chunks(emul, name):
    """run_granule (one granule per wave, tails and matrixing rows handed on) is the same arithmetic in the same order as
    run_chunk: PCM bit-identical for every block-type mix (incl. the H5 corner, whose peek values travel with the rows),
    mono corpora (run_chunk inside the granule kernel) and the long-block fast path of the requantisation"""
    sp, sd = corpus.case(name, n=13)
    want = emul_decode(emul, sp, sd, 0)
    assert np.array_equal(emul_decode_granules(emul, sp, sd), want)


def test_emul_granule_waves_states_resets_mode_and_rate_switches(emul, oracle):
    sp, sd = _mode_switch_records()
    whole = emul_decode(emul, sp, sd, 0)
    assert np.array_equal(emul_decode_granules(emul, sp, sd), whole)
    # the carried state in and out, batch after batch (granule 0 of a batch takes the caller's state; H5 at frame 0)
    sp, sd = oracle.generate(C2_SEED, 0, 64)
    whole = emul_decode(emul, sp, sd, 0)
    st = np.zeros(emul.emul_state_floats(), np.float32)
    st_ref = np.zeros(emul.emul_state_floats(), np.float32)
    cuts = [0, 1, 2, 9, 10, 31, 47, 48, 64]
    out = [emul_decode_granules(emul, sp[a:b], sd[a:b], st) for a, b in zip(cuts[:-1], cuts[1:])]
    assert np.array_equal(np.concatenate(out), whole)
    emul_decode(emul, sp, sd, 0, st_ref)
    assert np.array_equal(st.view(np.uint32), st_ref.view(np.uint32))
    # a RESET frame in the middle: its input state is zero, no wait
    sd2 = sd.copy()
    sd2["frame"][20] |= 0x40
    assert np.array_equal(emul_decode_granules(emul, sp, sd2), emul_decode(emul, sp, sd2, 0))
    # float PCM
    got = emul_decode_granules(emul, sp[:24], sd[:24], f32=True)
    want = np.zeros((24, 2304), np.float32)
    emul.emul_decode_frames_f32(_p(sp[:24]), _p(sd[:24]), 24, None, _p(want), 0)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # every frame a short block in granule 1 / channel 1 (H5: the peek values come with the rows of granule 0)
    sp3, sd3 = corpus.case("ms_short_heavy_480", n=16)
    assert np.array_equal(emul_decode_granules(emul, sp3, sd3), emul_decode(emul, sp3, sd3, 0))
    # the sampling frequency changes inside the batch, and the workgroups' line tables are for another one altogether:
    # such granules read the global line table (ph_requant's TG)
    a, sa = corpus.case("ms_long_441", n=6)
    b, sb = corpus.case("ms_short_heavy_480", n=6)
    sp4 = np.concatenate([a, b, a]); sd4 = np.concatenate([sa, sb, sa])
    for hint in (0, 1, 2):
        assert np.array_equal(emul_decode_granules(emul, sp4, sd4, sf_hint=hint), emul_decode(emul, sp4, sd4, 0)), hint


@pytest.mark.parametrize("name", ["ms_long_441", "ms_short_heavy_480", "mono_441"])
def test_emul_granule_waves_give_up_waiting_for_other_workgroups(emul, name):
    """the bounded wait (gran_far_wait): with every wait for another workgroup giving up at once, the first wave of each
    workgroup derives the state at the start of its frame the independent way (gran_slow_chunk(..., state_only): run_chunk's
    halo as a called function) and goes on as a granule wave with it -- same PCM"""
    sp, sd = corpus.case(name, n=21)
    want = emul_decode(emul, sp, sd, 0)
    assert np.array_equal(emul_decode_granules(emul, sp, sd, debug=1), want)


def emul_decode_ring(emul, sp, sd, per, state=None, f32=False, sf_hint=0):
    n = sp.shape[0]
    pcm = np.zeros((n, 2304), np.float32 if f32 else np.int16)
    emul.emul_decode_frames_ring(_p(sp), _p(sd), n, _p(state), None if f32 else _p(pcm), _p(pcm) if f32 else None, per, sf_hint)
    return pcm


@pytest.mark.parametrize("name", list(corpus.ALL_CASES))
def test_emul_persistent_granule_kernel_equals_independent_chunks(emul, name):
    """run_granule_ring (k_decode_p: 16 waves going round a range of frames, ring of LDS mailboxes, one halo per range) is the
    same arithmetic in the same order as run_chunk: PCM bit-identical on every corpus -- ranges of 8 frames (one turn of
    the ring), 9 (odd: a range ends in the middle of a turn), 23 and the whole batch; H5 frames at range starts, mono
    corpora (every frame through run_chunk inside the loop), RESET frames, intensity stereo"""
    sp, sd = corpus.case(name, n=51)
    want = emul_decode(emul, sp, sd, 0)
    for per in (8, 9, 23, 51):
        assert np.array_equal(emul_decode_ring(emul, sp, sd, per), want), per


def test_emul_persistent_granule_kernel_states_and_mode_switches(emul, oracle):
    """stereo / mono / stereo runs cut by range boundaries everywhere (the halo frame in front of a range is mono, or
    stereo right after mono: the range's first wave derives the state with run_chunk's halo; a range that starts with a
    mono frame or a RESET frame needs none), the carried state in and out batch after batch, float PCM, another
    sampling frequency than the workgroup's tables"""
    sp, sd = _mode_switch_records()
    whole = emul_decode(emul, sp, sd, 0)
    for per in (8, 9, 10, 11, 13):
        assert np.array_equal(emul_decode_ring(emul, sp, sd, per), whole), per
    sp, sd = oracle.generate(C2_SEED, 0, 96)
    whole = emul_decode(emul, sp, sd, 0)
    st = np.zeros(emul.emul_state_floats(), np.float32)
    st_ref = np.zeros(emul.emul_state_floats(), np.float32)
    cuts = [0, 17, 33, 34, 60, 96]
    out = [emul_decode_ring(emul, sp[a:b], sd[a:b], 8, st) for a, b in zip(cuts[:-1], cuts[1:])]
    assert np.array_equal(np.concatenate(out), whole)
    emul_decode(emul, sp, sd, 0, st_ref)
    assert np.array_equal(st.view(np.uint32), st_ref.view(np.uint32))
    sd2 = sd.copy()
    for f in (16, 24, 41):                                  # RESET at a range start, right before one, in the middle
        sd2["frame"][f] |= 0x40
    assert np.array_equal(emul_decode_ring(emul, sp, sd2, 8), emul_decode(emul, sp, sd2, 0))
    got = emul_decode_ring(emul, sp[:40], sd[:40], 8, f32=True)
    want = np.zeros((40, 2304), np.float32)
    emul.emul_decode_frames_f32(_p(sp[:40]), _p(sd[:40]), 40, None, _p(want), 0)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    a, sa = corpus.case("ms_long_441", n=12)
    b, sb = corpus.case("ms_short_heavy_480", n=12)
    sp4 = np.concatenate([a, b, a]); sd4 = np.concatenate([sa, sb, sa])
    for hint in (0, 1):
        assert np.array_equal(emul_decode_ring(emul, sp4, sd4, 8, sf_hint=hint), emul_decode(emul, sp4, sd4, 0)), hint


def test_emul_int16_conversion_lane_form_equals_the_reference_conversion(emul):
    """pcm_convert18_lane -- binary32 product rounded toward zero, v_med3 clamp (NaN -> min3), the wrap-around beyond
    65538 -- against pcm_from_sum (P:2028-2031: binary64 product, cvttsd2si, clip to +-32767), per sample: the
    per-sample form (`wrap`) everywhere, the fast form wherever no sample of the wave is beyond the wrap point -- NaNs
    included, which the wave-wide test (an fmax) does not see: both forms give -32767 for them"""
    rng = np.random.default_rng(7)
    n = 18 * 4000
    s = np.concatenate([
        rng.standard_normal(n).astype(np.float32),                              # around full scale
        (rng.standard_normal(n) * 1e-3).astype(np.float32),
        (rng.integers(-40000, 40000, n) / np.float32(32767.0)).astype(np.float32),   # products next to integers
        np.nextafter((rng.integers(-40000, 40000, n) / np.float32(32767.0)).astype(np.float32), np.float32(0)),
        (rng.standard_normal(n) * 3e4).astype(np.float32),                      # up to and beyond the wrap point
        rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32).view(np.float32),  # any bit pattern: NaNs, infinities, denormals
    ])
    special = np.array([np.nan, np.inf, -np.inf, 65538.0, np.nextafter(np.float32(65538.0), np.float32(1e9)), -65538.0, -70000.0,
                        1.0, -1.0, 0.0, -0.0, 32767.0 / 32767.0, 3.4e38, -3.4e38, 1e-45, 65537.99, 2.0, -2.0], np.float32)
    s = np.concatenate([s, special]).astype(np.float32)
    assert s.size % 18 == 0
    g = s.size // 18
    fast = np.zeros(s.size, np.int32); wrap = np.zeros(s.size, np.int32); exact = np.zeros(s.size, np.int32)
    with np.errstate(all="ignore"):
        emul.emul_pcm_convert18(_p(s), g, _p(fast), _p(wrap), _p(exact))
    assert np.array_equal(wrap, exact)
    ok = ~(s > np.float32(65538.0))                     # NaNs stay in: the fast form must get them right by itself
    assert np.array_equal(fast[ok], exact[ok])
    assert (exact[np.isnan(s)] == -32767).all() and (fast[np.isnan(s)] == -32767).all()

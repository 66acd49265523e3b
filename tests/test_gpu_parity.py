"""Parity tests proper: the HIP engine, called through the C-ABI
(include/pdmp3_hip.h via pdmp3_amd.hip), against the CPU oracle on the same
seeded inputs, against the committed golden fixtures of the reference, and --
at BASELINE.json's full sizes -- through size-independent properties.

Bars, as literal numbers (tests/corpus.py PCM_TOL_LSB / F32_TOL_ABS):

  int16 PCM   +-1 LSB (north_star) on all 30 record corpora but one: the 15 loud cases (global_gain 120..170,
              6-37 % of the samples clipped), their 15 `_fs` twins at a realistic level (peaks 13-32 k, median
              100-670 LSB, nothing clipped); ms_loud_clip (4.5e5 x full scale, 99 % clipped): 32 LSB
  float PCM   1e-5 absolute on the 15 `_fs` twins; the loud cases' sums are 19..141 x full scale where one binary32
              ulp is 2e-6..1.5e-5: a literal number per case (2 x the measured difference = 1.5..3.5 ulp)
  stages 0-2  bit-exact (separately rounded mul/add like the reference); stage 3 (hybrid) 1e-5 x amplitude
  64 frames per case; the reference fixture holds the PCM's hash, head and last frame.
"""
import os
import sys

import numpy as np
import pytest

import corpus
from conftest import C2_SEED
from util import assert_pcm_close, nch_of, sha

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def gpu_decode(engine, sp, sd, chunk=0, state=None, stages=False):
    import torch
    dsp, dsd = engine.upload(sp, sd)
    n = sp.shape[0]
    pcm = torch.zeros((n, 2304), dtype=torch.int16, device=engine.tdev)
    if stages:
        stg = torch.zeros((n, 2, 2, 4, 576), dtype=torch.float32, device=engine.tdev)
        engine.decode_stages(dsp, dsd, pcm, stg, state=state)
        torch.cuda.synchronize()
        return pcm.cpu().numpy(), stg.cpu().numpy()
    engine.decode(dsp, dsd, pcm, state=state, chunk_frames=chunk)
    torch.cuda.synchronize()
    return pcm.cpu().numpy()


def test_native_library_is_loaded(engine):
    import pdmp3_amd
    maps = open("/proc/self/maps").read()
    assert os.path.basename(pdmp3_amd.library_path()) in maps


@pytest.mark.parametrize("name", list(corpus.ALL_CASES))
def test_gpu_vs_oracle_and_golden(engine, oracle, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    n = int(g["n_frames"][0])
    sp, sd = corpus.case(name, n=n)
    want, ws = oracle.decode(sp, sd, stages=True)
    assert sha(want) == str(g["pcm_sha"][0]), "oracle drifted from the reference fixture"
    got, gs = gpu_decode(engine, sp, sd, stages=True)
    nch = nch_of(sd)
    for k in range(3):
        assert np.array_equal(ws[:, :, :nch, k].view(np.uint32), gs[:, :, :nch, k].view(np.uint32)), \
            "stage %d not bit-exact" % k
    amp = max(1.0, float(np.abs(ws[:, :, :nch, 3]).max()))
    err = float(np.abs(ws[:, :, :nch, 3] - gs[:, :, :nch, 3]).max())
    assert err <= 1e-5 * amp, "hybrid output differs by %g (amplitude %g)" % (err, amp)
    tol = corpus.PCM_TOL_LSB[name]
    assert_pcm_close(got, want, tol, name + " vs oracle (= reference fixture by hash)")
    assert_pcm_close(got[:4], g["pcm_head"], tol, name + " vs reference fixture, head")
    assert_pcm_close(got[-1:], g["pcm_last"], tol, name + " vs reference fixture, last frame")
    # the normal (non-dump) kernels: chunk form with a halo per wave, granule form
    for chunk in (1, 5, 0):
        assert_pcm_close(gpu_decode(engine, sp, sd, chunk=chunk), want, tol, name + " chunk=%d" % chunk)


def test_gpu_generator_matches_oracle(engine, oracle):
    import torch
    n = 300
    spectra, side, _ = engine.alloc_frames(n)
    engine.generate(C2_SEED, 1000, n, spectra, side)
    torch.cuda.synchronize()
    sp, sd = oracle.generate(C2_SEED, 1000, n)
    assert np.array_equal(spectra.cpu().numpy(), sp)
    assert np.array_equal(side.cpu().numpy(), sd.view(np.uint8).reshape(n, 4, 128))


def test_gpu_c2_prefix_vs_reference_fixture(engine, oracle):
    g = np.load(os.path.join(GOLD, "c2_prefix.npz"))
    sp, sd = oracle.generate(C2_SEED, 0, 32)
    got = gpu_decode(engine, sp, sd, chunk=4)
    assert_pcm_close(got, g["pcm_head"], 1, "C2 prefix")


def test_gpu_c2_full_size(engine, oracle):
    """BASELINE configs[1]: 2048 frames = 4096 granules, one stream.  Oracle
    finishes this in <1 s, so compare everything; plus chunking invariance."""
    import torch
    n = 2048
    spectra, side, pcm = engine.alloc_frames(n)
    engine.generate(C2_SEED, 0, n, spectra, side)
    engine.decode(spectra, side, pcm, chunk_frames=0)
    torch.cuda.synchronize()
    got = pcm.cpu().numpy()
    sp, sd = oracle.generate(C2_SEED, 0, n)
    want = oracle.decode(sp, sd)
    g = np.load(os.path.join(GOLD, "c2_prefix.npz"))
    assert sha(want) == str(g["pcm_sha_2048"][0])
    dmax, ndiff = assert_pcm_close(got, want, 1, "C2 full")
    assert ndiff < 0.02 * got.size
    for chunk in (1, 2, 7, 64, 2048):
        pcm2 = torch.zeros_like(pcm)
        engine.decode(spectra, side, pcm2, chunk_frames=chunk)
        torch.cuda.synchronize()
        assert torch.equal(pcm, pcm2), "chunk=%d changes the PCM" % chunk


def test_gpu_state_handoff(engine, oracle):
    import torch
    sp, sd = oracle.generate(C2_SEED, 0, 50)
    whole = gpu_decode(engine, sp, sd, chunk=50)
    st = engine.new_state()
    a = gpu_decode(engine, sp[:13], sd[:13], chunk=0, state=st)
    b = gpu_decode(engine, sp[13:], sd[13:], chunk=5, state=st)
    assert np.array_equal(np.concatenate([a, b]), whole)


def test_gpu_shard_with_halo_equals_whole(engine, oracle):
    """Multi-GPU rule: a shard decoded from 2 halo frames earlier,
    halo output discarded, equals the same range of the whole stream."""
    sp, sd = oracle.generate(C2_SEED, 0, 96)
    whole = gpu_decode(engine, sp, sd, chunk=8)
    lo = 40
    part = gpu_decode(engine, sp[lo - 2:], sd[lo - 2:], chunk=8)[2:]
    assert np.array_equal(part, whole[lo:])


def test_gpu_ragged_and_tiny(engine, oracle):
    sp, sd = oracle.generate(C2_SEED, 0, 9)
    want = oracle.decode(sp, sd)
    for n in (1, 2, 3, 9):
        got = gpu_decode(engine, sp[:n], sd[:n], chunk=4)
        assert_pcm_close(got, want[:n], 1, "n=%d" % n)
    engine.decode(*engine.upload(sp[:0], sd[:0]), engine.alloc_frames(1)[2], n_frames=0)


def test_gpu_large_synthetic_properties(engine):
    """C5-style size (65536 frames on one GPU): determinism, chunk invariance,
    and equality of a prefix with an independent small run."""
    import torch
    n = 65536
    spectra, side, pcm = engine.alloc_frames(n)
    seed = 0x5EED0000C5
    engine.generate(seed, 0, n, spectra, side)
    engine.decode(spectra, side, pcm, chunk_frames=32)
    pcm2 = torch.empty_like(pcm)
    engine.decode(spectra, side, pcm2, chunk_frames=16)
    torch.cuda.synchronize()
    assert torch.equal(pcm, pcm2)
    sp3, sd3, pcm3 = engine.alloc_frames(128)
    engine.generate(seed, 0, 128, sp3, sd3)
    engine.decode(sp3, sd3, pcm3, chunk_frames=128)
    torch.cuda.synchronize()
    assert torch.equal(pcm[:128], pcm3)
    assert int(pcm.abs().max()) > 1000


def test_gpu_c5_shards(engine, oracle):
    """BASELINE configs[4] (C5) at its stated size on one GPU: the 1 000 000-frame synthetic stream is cut into 8
    frame-range shards of 125 000 frames (pdmp3_amd.sharding), each generated on device from its counter-based
    generator and decoded from a 2-frame halo; parity against the oracle on every frame within +-2 of
    a shard boundary and on a 4096-frame prefix of each shard.  The oracle decodes the prefixes in pieces of 512
    frames on a thread pool, each started cold 8 frames in front of the frames it is compared on (the synthesis
    state reaches 2 granules back)."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from pdmp3_amd.sharding import shard_with_halo, frame_range
    seed = 0x5EED0000C5
    per, prefix, piece = 125000, 4096, 512
    world, total = 8, 8 * per
    oracle.decode(*oracle.generate(seed, 0, 1))            # the oracle's lazily built tables, once, on this thread

    def want_of(ab):
        a, b = ab
        w0 = max(0, a - 8)
        sp, sd = oracle.generate(seed, w0, b - w0)
        if w0 > 0:
            sd["frame"][0] |= 0x40                          # oracle starts cold; its first 8 frames are discarded
        return oracle.decode(sp, sd)[a - w0:]

    checked = 0
    with ThreadPoolExecutor(max(1, min(16, len(os.sched_getaffinity(0))))) as pool:
        for rank in range(world):
            first, count, halo = shard_with_halo(total, rank, world)
            lo, hi = frame_range(total, rank, world)
            assert (count, halo) == ((per + 2, 2) if rank else (per, 0))
            spectra, side, pcm = engine.alloc_frames(count)
            engine.generate(seed, first, count, spectra, side)
            engine.decode(spectra, side, pcm)
            torch.cuda.synchronize()
            assert pcm.shape[0] - halo == hi - lo == per
            # shard start (the boundary's +2 side and the 4096-frame prefix) and shard end (the next boundary's -2 side)
            wins = [(a, min(a + piece, lo + prefix)) for a in range(lo, lo + prefix, piece)] + [(hi - 4, hi)]
            got = [pcm[halo + a - lo:halo + b - lo].cpu().numpy() for a, b in wins]
            del spectra, side, pcm
            for (a, b), g, w in zip(wins, got, pool.map(want_of, wins)):
                assert_pcm_close(g, w, 1, "shard %d frames %d..%d" % (rank, a, b))
                checked += b - a
    assert checked == world * (prefix + 4)


@pytest.mark.parametrize("chunk", [0, 1, 2, 3, 7])
def test_gpu_channel1_state_survives_mono_runs(engine, oracle, chunk):
    """stereo / mono / stereo: channel 1's overlap and polyphase history are what the last stereo frame left
    (P:1777, P:2126 are per channel), for any chunking, and across batches through the carried state"""
    from test_pipeline_emul import _mode_switch_records
    sp, sd = _mode_switch_records()
    want = oracle.decode(sp, sd)
    whole = gpu_decode(engine, sp, sd, chunk=sp.shape[0])
    assert_pcm_close(whole, want, 1, "one chunk")
    assert np.array_equal(gpu_decode(engine, sp, sd, chunk=chunk), whole)
    st = engine.new_state()
    for cuts in ([0, 9, 14, 20, 22, 29, sp.shape[0]], [0, 7, 11, 19, 24, 31, sp.shape[0]], [0, 8, 10, 21, 23, 30, sp.shape[0]]):
        st.zero_()
        parts = [gpu_decode(engine, sp[a:b], sd[a:b], chunk=chunk, state=st) for a, b in zip(cuts[:-1], cuts[1:])]
        assert np.array_equal(np.concatenate(parts), whole), cuts


@pytest.mark.parametrize("name", list(corpus.ALL_CASES))
def test_gpu_float_pcm(engine, oracle, name):
    """float PCM (pdmp3_hip_decode_frames_f32): the synthesis sums before P:2028's scaling, against the
    oracle's (whose int16 = clip(trunc(float * 32767)) is the reference's, bit for bit): 1e-5 absolute on the `_fs`
    cases, the literal per-case number of corpus.F32_TOL_ABS on the loud ones; int16 form and float form agree"""
    import torch
    n = corpus.N_GOLDEN
    sp, sd = corpus.case(name, n=n)
    _, want = oracle.decode_f32(sp, sd)
    dsp, dsd = engine.upload(sp, sd)
    out = torch.zeros((n, 2304), dtype=torch.float32, device=engine.tdev)
    engine.decode_f32(dsp, dsd, out, chunk_frames=3)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    err = float(np.abs(got - want).max())
    assert err <= corpus.F32_TOL_ABS[name], "%s: float PCM differs by %g > %g" % (name, err, corpus.F32_TOL_ABS[name])
    q = np.clip(np.trunc(got.astype(np.float64) * 32767.0), -32767, 32767)
    q[got > 65538.0] = -32767                      # P:2028-2031 on x86-64: the wrap-around of cvttsd2si
    nch = nch_of(sd)
    assert np.array_equal(q[:, :1152 * nch].astype(np.int16), gpu_decode(engine, sp, sd, chunk=3)[:, :1152 * nch]), name


def test_gpu_float_pcm_c2(engine, oracle):
    import torch
    n = 2048
    spectra, side, _ = engine.alloc_frames(n)
    engine.generate(C2_SEED, 0, n, spectra, side)
    out = torch.zeros((n, 2304), dtype=torch.float32, device=engine.tdev)
    engine.decode_f32(spectra, side, out)
    torch.cuda.synchronize()
    sp, sd = oracle.generate(C2_SEED, 0, 96)
    _, want = oracle.decode_f32(sp, sd)
    got = out[:96].cpu().numpy()
    assert float(np.abs(got - want).max()) <= 1e-5              # literal (C2's sums reach 21 x full scale in these frames)
    pcm = torch.zeros((n, 2304), dtype=torch.int16, device=engine.tdev)
    engine.decode(spectra, side, pcm)
    torch.cuda.synchronize()
    f = out.cpu().numpy().astype(np.float64)
    q = np.clip(np.trunc(f * 32767.0), -32767, 32767)
    q[f > 65538.0] = -32767
    assert np.array_equal(q.astype(np.int16), pcm.cpu().numpy())


def test_gpu_granule_launches_equal_independent_chunks(engine):
    """one granule per wave with tails and rows handed from wave to wave (run_granule; what every launch of up to
    12288 frames uses: PDMP3_HIP_GRAN_MAX) against independent 2-frame chunks with halos: PCM and carried state bit-identical, over many synthetic
    batches, sizes that leave the last workgroup partly filled, and host threads launching on their own streams"""
    import threading
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from soak_chain import one
    for r in range(48):
        n = 2048 if r % 4 else [1, 2, 7, 8, 9, 63, 65, 1000, 2047, 4099][(r // 4) % 10]
        assert one(engine, 0x5EED000000 + r, n), (r, n)
    out = {}

    def worker(k):
        s = torch.cuda.Stream()
        ok = True
        with torch.cuda.stream(s):
            for r in range(12):
                ok = one(engine, 0xABC000 + 1000 * k + r, 2048) and ok
        out[k] = ok
    th = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert all(out.get(k) for k in range(4)), out


def test_gpu_granule_kernel_gives_up_waiting_for_other_workgroups(monkeypatch, oracle):
    """the granule kernel's give-up path ON THE DEVICE.  An engine created with
    PDMP3_HIP_DEBUG_FAR_TIMEOUT=1 makes every wait for another workgroup time out at once: the first wave of each
    workgroup derives its frame's opening state the independent way (gran_slow_chunk(..., state_only): run_chunk's halo
    as a called function on generic LDS pointers).  PCM and carried state must equal the chunk kernel's bit for bit:
    launch sizes of one workgroup up to several rounds of them, the H5-heavy / mono / RESET corpora, float PCM."""
    import torch
    import pdmp3_amd
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from soak_chain import one
    monkeypatch.setenv("PDMP3_HIP_DEBUG_FAR_TIMEOUT", "1")
    eng = pdmp3_amd.Engine(0)
    monkeypatch.delenv("PDMP3_HIP_DEBUG_FAR_TIMEOUT")
    try:
        for k, n in enumerate((9, 65, 2048, 4099, 12288)):
            assert one(eng, 0x5EED00F000 + k, n), n
        for name in ("ms_short_heavy_480", "mono_441", "ms_resets", "ms_mixed_blocks_441_fs", "ms_is_short_480_fs"):
            sp, sd = corpus.case(name)
            a = gpu_decode(eng, sp, sd, chunk=1)                 # granule kernel, every far wait given up
            b = gpu_decode(eng, sp, sd, chunk=2)                 # independent chunks
            assert np.array_equal(a, b), name
            assert_pcm_close(a, oracle.decode(sp, sd), corpus.PCM_TOL_LSB[name], name)
        sp, sd = oracle.generate(C2_SEED, 0, 300)
        dsp, dsd = eng.upload(sp, sd)
        f1 = torch.zeros((300, 2304), dtype=torch.float32, device=eng.tdev)
        f2 = torch.zeros_like(f1)
        eng.decode_f32(dsp, dsd, f1, chunk_frames=1)
        eng.decode_f32(dsp, dsd, f2, chunk_frames=3)
        torch.cuda.synchronize()
        assert torch.equal(f1, f2)
    finally:
        eng.close()

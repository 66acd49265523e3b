#!/usr/bin/env python3
"""bench.py -- throughput of the Layer-III transform hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: one rank per GPU -- either launched by torch.distributed.run, or, as the plain command, bench.py starts
  `python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py ...` itself as a child and relays its line)

A "step" is one pass of the hot path (pdmp3_hip_decode_frames: requantize ->
reorder -> stereo -> antialias -> IMDCT/overlap -> frequency inversion ->
polyphase -> int16 PCM) over one batch of synthetic input that is already
resident in HBM.  Workload at N = 1 is BASELINE.json configs[1] (SURVEY 8d
"C2"): one stream of 2048 stereo 44.1 kHz frames = 4096 pre-Huffman-decoded
granules (x 2 channels), integer-only generator, seed 0x5EED0000C2.  At N > 1
the workload is BASELINE.json configs[4] (SURVEY 8d "C5") at N ranks: one stream
of N x 125 000 frames (seed 0x5EED0000C5, counter-based: every rank generates its
own shard on its GPU), rank r decodes frames [125000 r, 125000 (r+1)) from a
2-frame halo: no data-path collective, fixed work per GPU (weak
scaling).  The one exchange of the path -- the final PCM gather to rank 0 over
RCCL, 576 MB per rank -- runs once after the timed region and is reported as
`gather_ms` / `gather_GBps`.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# before HIP comes up (torch brings it up here): the whole-stream decoder of the `end_to_end` extra keeps six windows in
# flight on their own HIP streams, which overlap only across hardware queues (default 4); the library sets the same
# default when it initialises HIP itself (host/stream_api.c shared_ctx_on).  No effect on the hot-path metric (one stream).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

SEED_C2 = 0x5EED0000C2
FRAMES_PER_GPU = 2048                 # C2: 4096 granules
SEED_C5 = 0x5EED0000C5
SHARD_FRAMES = 125000                 # C5: 1 000 000 frames on 8 GPUs
ALGO_BYTES_PER_FRAME = 9728           # 4 gc x (1152 B spectra + 128 B side + 1152 B PCM)
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: 8.0 TB/s spec
ALGO_FLOP_PER_FRAME = 552e3           # direct-form flops of one stereo frame (41.5 k IMDCT + 91.6 k polyphase + ~5 k per gc)
FP32_PEAK_TFLOPS = 157.3              # MI355X fp32 vector = fp32 matrix peak (the IMDCT / matrixing run on v_mfma_f32_16x16x4_f32)
RT_FRAMES_PER_S = 44100.0 / 1152.0


def effective_cpus():
    """CPUs this process can actually keep busy: affinity mask capped by the cgroup CPU quota (the GPU boxes show
    256 CPUs but run the job under a quota of a few)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except Exception:
            pass
    return n


def cpu_baseline(sample_frames, target_seconds, all_cores_seconds=0.0):
    """The reference CPU path on this box's host cores (rank 0, N = 1 only).
    kind "reference": the real pdmp3.c Decode_L3 from oracle/_ref (prebuilt in
    the build container); falls back to the oracle port when _ref is absent."""
    from oracle import oracle as orc
    o = orc.Oracle()
    sp, sd = o.generate(SEED_C2, 0, sample_frames)
    if orc.have_ref():
        dec, kind = orc.Reference(), "reference"
    else:
        dec, kind = o, "port"
    t1 = dec.time_decode(sp, sd, 1)
    reps = max(1, int(target_seconds / max(t1, 1e-6)))
    t = dec.time_decode(sp, sd, reps)
    fps = sample_frames * reps / t
    out = {
        "value": round(fps, 1), "unit": "frames/s", "cores": 1, "kind": kind,
        "sample": "%d passes of Decode_L3 over the same %d C2 frames (%.1f s, 1 thread; "
                  "the reference is single-threaded with process-global state)" % (reps, sample_frames, t),
        "x_realtime": round(fps / RT_FRAMES_PER_S, 1),
        "host_cpus": os.cpu_count(),
    }
    # all host cores.  The reference keeps its synthesis state in function statics, so "all cores"
    # = one forked process per CPU, each decoding the same sample (what a per-file farm of the reference would do).
    out["effective_cpus"] = effective_cpus()
    if all_cores_seconds > 0 and out["effective_cpus"] > 1:
        import multiprocessing as mp
        procs = min(out["effective_cpus"], 256)
        reps_all = max(1, int(all_cores_seconds / max(t1, 1e-6)))
        global _MP_JOB
        _MP_JOB = (dec, sp, sd, reps_all)
        try:
            ctx = mp.get_context("fork")
            with ctx.Pool(procs) as pool:
                pool.map(_mp_warm, range(procs))                 # every worker up and paged in
                w0 = time.perf_counter()
                pool.map(_mp_run, range(procs), chunksize=1)
                wall = time.perf_counter() - w0
            agg = procs * sample_frames * reps_all / wall
            out["all_cores"] = {"value": round(agg, 1), "unit": "frames/s", "cores": procs, "kind": kind,
                                "sample": "%d processes x %d passes over the same %d frames (%.1f s wall)" %
                                          (procs, reps_all, sample_frames, wall),
                                "x_realtime": round(agg / RT_FRAMES_PER_S, 1)}
        except Exception as e:                                   # a reported extra, never fatal
            out["all_cores"] = {"error": repr(e)}
    return out


_MP_JOB = None


def _mp_warm(_):
    dec, sp, sd, _reps = _MP_JOB
    dec.time_decode(sp[:64], sd[:64], 1)
    return 0


def _mp_run(_):
    dec, sp, sd, reps = _MP_JOB
    return dec.time_decode(sp, sd, reps)


def parity_checker():
    """the CPU decoder the bench line's `parity` field is measured against: the compiled reference (oracle/_ref, kind
    "reference") when its prebuilt library travelled with the repo, else the oracle's restatement ("port").  Checker
    only: called after the timed region, never part of what is measured."""
    from oracle import oracle as orc
    o = orc.Oracle()
    return (orc.Reference(), "reference", o) if orc.have_ref() else (o, "port", o)


def parity_of(got, want, frames, against, what):
    import numpy as np
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    return {"max_abs_diff_lsb": int(d.max()) if d.size else 0, "samples_differing": int((d > 0).sum()), "samples": int(d.size),
            "frames": int(frames), "against": against, "what": what, "tolerance_lsb": 1, "ok": bool(d.size == 0 or d.max() <= 1)}


def boundary_parity(gathered, seed, n_per_rank, world):
    """N > 1, rank 0: the gathered PCM against the CPU decoder on every frame within +-2 of a shard boundary, the first
    4 frames of the stream and the last 2 (the CPU decoder starts cold 8 frames in front of each window; the
    synthesis state reaches 2 granules back).  gathered: int16 [world * n_per_rank][2304]"""
    import numpy as np
    dec, against, o = parity_checker()
    total = world * n_per_rank
    wins = [(0, 4)] + [(r * n_per_rank - 2, r * n_per_rank + 2) for r in range(1, world)] + [(total - 2, total)]
    got, want = [], []
    for a, b in wins:
        w0 = max(0, a - 8)
        sp, sd = o.generate(seed, w0, b - w0)
        sd["frame"][0] |= 0x40                                  # PDMP3_FR_RESET: the CPU decoder starts from zero here
        want.append(dec.decode(sp, sd)[a - w0:])
        got.append(gathered[a:b])
    return parity_of(np.concatenate(got), np.concatenate(want), sum(b - a for a, b in wins), against,
                     "gathered PCM, frames within +-2 of each of the %d shard boundaries + stream head and tail" % (world - 1))


MFMA_BUSY_FRAC = None      # from the same PMC summary as `traffic` (SQ_VALU_MFMA_BUSY_CYCLES), when that is of this kernel


def measured_traffic(frames, n_halo):
    """HBM bytes per launch of k_decode from the committed PMC passes
    (profiles/*_pmc_summary.json; separate rocprofv3 --pmc runs of this same
    command, tools/gpu_profile.sh).  gfx950 correction per MI355X_MICROARCH.md
    (HBM): FETCH_SIZE reports half of a wide coalesced read stream -> x2;
    WRITE_SIZE matched the known PCM byte count exactly in calibration."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")))
    if not files or frames != FRAMES_PER_GPU or n_halo:
        return None, None
    try:
        import hashlib
        d = json.load(open(files[-1]))
        h = hashlib.sha256()
        for f in ("pdmp3_amd/csrc/decode_core.h", "pdmp3_amd/csrc/engine.hip"):
            h.update(open(os.path.join(ROOT, f), "rb").read())
        if d.get("kernel_source_sha16") != h.hexdigest()[:16]:
            return None, "stale: %s was measured on another version of the kernel" % os.path.basename(files[-1])
        fetch_kb = d["fetch"]["per_dispatch"]["FETCH_SIZE"]
        write_kb = d["write"]["per_dispatch"]["WRITE_SIZE"]
        global MFMA_BUSY_FRAC
        MFMA_BUSY_FRAC = d.get("mfma_busy_frac")
        return int((2.0 * fetch_kb + write_kb) * 1024), os.path.basename(files[-1])
    except Exception:
        return None, None


def self_launch(n_ranks):
    """`python bench.py --gpus N` without a launcher around it: run `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N bench.py <the same arguments>` as a child process (one rank per GPU, rendezvous on 127.0.0.1 at a
    free port) and relay what it prints.  Called before anything of this process has imported torch or initialised HIP:
    the children are ordinary fresh processes, nothing is exec'ed from a process that holds the GPU."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")            # dmabuf IPC: what RCCL needs on these hosts
    env.setdefault("OMP_NUM_THREADS", "1")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    for l in r.stdout.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)                              # the launcher's chatter, kept off the one-line contract
    if lines:
        print(lines[-1])
    sys.stdout.flush()
    sys.exit(r.returncode if r.returncode else (0 if lines else 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--chunk", type=int, default=0, help="frames per workgroup chunk (0 = engine default)")
    ap.add_argument("--frames", type=int, default=0, help="frames per GPU per step (0: 2048 = C2 at 1 GPU, 125000 = the C5 shard at N > 1)")
    ap.add_argument("--dump-gathered", default="", help="N > 1, rank 0: write the gathered PCM (int16 .npy) here (tests)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-all-seconds", type=float, default=6.0, help="all-host-cores leg of the CPU baseline (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the bitstream-to-PCM extra")
    ap.add_argument("--from-idle", dest="from_idle", action="store_true", default=True,
                    help="N = 1: also time the W + K launches once before any other GPU work (from_idle_gpu; the default)")
    ap.add_argument("--no-from-idle", dest="from_idle", action="store_false",
                    help="skip from_idle_gpu (a rocprofv3 summary of the command then holds the headline's launches only)")
    ap.add_argument("--shard", type=int, default=SHARD_FRAMES, help="N = 1: frames of the extra C5-shard figure (0 = skip)")
    ap.add_argument("--big", type=int, default=131072, help="frames of the extra large-batch roofline probe (0 = skip)")
    ap.add_argument("--slices", type=int, default=8, help="N > 1: pieces a rank's shard is decoded and sent in per step (never under 4096 frames each)")
    ap.add_argument("--strong-frames", type=int, default=1000000, help="N > 1: frames of the fixed stream of the strong-scaling leg (0 = skip)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the plain command (`python bench.py --gpus N ...`): start the N ranks here, as a FRESH child -- this process has
        # not imported torch nor touched the GPU, and never will -- and hand on its one JSON line and its exit code
        return self_launch(args.gpus)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # the CPU legs run first: they fork workers, which must happen before this process touches the GPU
    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu:
        cpu = cpu_baseline(FRAMES_PER_GPU, args.cpu_seconds, args.cpu_all_seconds)

    import torch
    import torch.distributed as dist
    import pdmp3_amd

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and rank == 0:
        print("bench.py: --gpus %d under a launcher with WORLD_SIZE=%d: running %d ranks" % (args.gpus, world, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # PDMP3_BENCH_BACKEND=gloo + several ranks on one GPU is only for exercising this code path on a
    # single-GPU box; the real multi-GPU run is one rank per GPU over RCCL ("nccl").
    backend = os.environ.get("PDMP3_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=backend)
    coll_dev = "cuda" if backend == "nccl" else "cpu"

    eng = pdmp3_amd.Engine(dev_index)
    n = args.frames or (FRAMES_PER_GPU if world == 1 else SHARD_FRAMES)
    seed = SEED_C2 if world == 1 else SEED_C5
    halo = 2 if rank > 0 else 0
    first = rank * n - halo
    spectra, side, pcm = eng.alloc_frames(n + halo)
    eng.generate(seed, first, n + halo, spectra, side)
    torch.cuda.synchronize()

    def step():
        eng.decode(spectra, side, pcm, chunk_frames=args.chunk)

    def larger_launches():
        """the legs on launches of the C5-shard size.  They run BEFORE the warm-up and the timed region of the headline (one
        GPU only): a decoder that is kept busy runs at the clocks these 60 ms of launches bring the GPU to, and the W + K
        launches of the headline -- 0.5 ms with the driver's W = 5, K = 20 -- do not get there by themselves (the same 25
        launches measured first, from an idle GPU: 21.4-21.9 us each; after these legs: see `clocks` in the line)."""
        extra = {}
        if args.shard and world == 1:
            # what ONE rank of an N > 1 run does, on this GPU alone: the C5 shard of rank 1 -- 125 000 frames from a 2-frame
            # halo, same seed, same engine call, hence the same kernel -- so that the driver's 1 -> N curve compares like
            # with like (the N = 1 headline is C2, another workload and another kernel).  Outside the timed region (before it: see larger_launches).
            ns = args.shard
            sp3, sd3, pcm3 = eng.alloc_frames(ns + 2)
            eng.generate(SEED_C5, ns - 2, ns + 2, sp3, sd3)
            for _ in range(5):
                eng.decode(sp3, sd3, pcm3, chunk_frames=args.chunk)
            torch.cuda.synchronize()
            reps = 20
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            w0 = time.perf_counter()
            a.record()
            for _ in range(reps):
                eng.decode(sp3, sd3, pcm3, chunk_frames=args.chunk)
            b.record()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - w0) / reps
            ms = a.elapsed_time(b) / reps
            ach = (ns + 2) * ALGO_BYTES_PER_FRAME / (ms * 1e-3) / 1e9
            extra["c5_shard_1gpu"] = {
                "workload": "C5 shard of rank 1 (BASELINE configs[4]): frames [%d, %d) of the 1M-frame stream decoded from a 2-frame "
                            "halo, one GPU, what each rank of an N > 1 run does per step" % (ns, 2 * ns),
                "frames": ns, "halo_frames": 2, "kernel": eng.last_launch_kernel(), "avg_launch_ms": round(ms, 4),
                "ms_per_step": round(wall * 1e3, 4), "frames_per_s": round(ns / wall, 1),
                "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5)}
            # kept for the parity check that runs after the headline's timed region: the PCM of the shard's first 256 frames
            # (the 2 halo frames in front of them are decode-and-discard) and of its last 2
            extra["_c5_pcm"] = (ns, pcm3[2:2 + min(256, ns)].cpu().numpy(), pcm3[-2:].cpu().numpy())
            del sp3, sd3, pcm3

        if args.big and world == 1:
            # kernel quality at throughput size (SURVEY 8d C5 shard scale), outside the timed region (before it)
            nb = args.big
            sp2, sd2, pcm2 = eng.alloc_frames(nb)
            eng.generate(0x5EED0000C5, 0, nb, sp2, sd2)
            for _ in range(10):                             # (sustained-throughput figure: the first launches after the 28 us
                eng.decode(sp2, sd2, pcm2, chunk_frames=args.chunk)   #  C2 launches run 5 % slower than the steady state)
            torch.cuda.synchronize()
            reps = 20
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                eng.decode(sp2, sd2, pcm2, chunk_frames=args.chunk)
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / reps
            ach = nb * ALGO_BYTES_PER_FRAME / (ms * 1e-3) / 1e9
            extra["roofline_large_batch"] = {
                "frames": nb, "avg_launch_ms": round(ms, 4), "frames_per_s": round(nb / (ms * 1e-3), 1),
                "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                "fp32_tflops": round(nb * ALGO_FLOP_PER_FRAME / (ms * 1e-3) / 1e12, 2),
                "fp32_frac": round(nb * ALGO_FLOP_PER_FRAME / (ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 5),
            }
            # float PCM (pdmp3_hip_decode_frames_f32): 3584 algorithmic bytes per granule-channel
            # (1152 spectra + 128 side + 2304 float PCM) = 14336 per stereo frame
            pcmf = torch.empty((nb, 2304), dtype=torch.float32, device=eng.tdev)
            for _ in range(2):
                eng.decode_f32(sp2, sd2, pcmf, chunk_frames=args.chunk)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                eng.decode_f32(sp2, sd2, pcmf, chunk_frames=args.chunk)
            b.record()
            torch.cuda.synchronize()
            msf = a.elapsed_time(b) / reps
            achf = nb * 14336 / (msf * 1e-3) / 1e9
            extra["roofline_float_pcm"] = {"frames": nb, "avg_launch_ms": round(msf, 4), "frames_per_s": round(nb / (msf * 1e-3), 1),
                                         "algorithmic_bytes_per_frame": 14336, "achieved": round(achf, 2), "peak": HBM_PEAK_GBS,
                                         "unit": "GB/s", "frac": round(achf / HBM_PEAK_GBS, 5), "kernel": "k_decode<false, true, 1>"}
            del sp2, sd2, pcm2, pcmf


        return extra

    # from_idle_gpu: the same W + K launches once from an idle GPU, reported beside the headline -- the figure that compares
    # like with like across rounds (rounds 1-3 timed the headline this way); --no-from-idle leaves it out, so that a
    # rocprofv3 summary of the command holds the headline's launches only
    from_idle = None
    if world == 1 and args.from_idle and (args.shard or args.big):
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        w0 = time.perf_counter()
        e0.record()
        for _ in range(args.steps):
            step()
        e1.record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - w0
        ms_i = e0.elapsed_time(e1) / args.steps
        ach_i = n * ALGO_BYTES_PER_FRAME / (ms_i * 1e-3) / 1e9
        from_idle = {"avg_launch_ms": round(ms_i, 5), "ms_per_step": round(wall / args.steps * 1e3, 5),
                     "frames_per_s": round(n * args.steps / wall, 1),
                     "achieved": round(ach_i, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach_i / HBM_PEAK_GBS, 5),
                     "what": "the same %d warm-up + %d timed launches as the headline, run first, before any other GPU work of this process" % (args.warmup, args.steps)}
    extra_legs = larger_launches() if world == 1 else {}

    # ---- N > 1: the step is decode + the path's one exchange, pipelined (BASELINE configs[4]: "... with RCCL PCM gather") ----
    # A rank's shard is decoded in S consecutive pieces (the synthesis state goes from piece to piece through a state
    # block: no extra halos) and piece j's PCM is sent to rank 0 on a second stream as soon as its kernel is done, under the
    # kernels of the pieces behind it and -- the PCM buffer is double -- of the next step.  `value` is this steady state;
    # `value_decode_only` is the same K steps without the exchange (rounds 1-5's `value`), `gather_ms` the exchange alone.
    pipe = None
    if world > 1:
        import numpy as np
        S = max(1, min(args.slices, (n + halo) // 4096 or 1))
        nccl = backend == "nccl"
        state = eng.new_state()
        comm_stream = torch.cuda.Stream(device=dev_index) if nccl else None
        halo_of = lambda m: 2 if m > 0 else 0

        def pieces(m, frames):
            """[(lo, hi, a)]: piece j of rank m decodes frames [lo, hi) of its buffer and sends [a, hi) (a: behind the halo)"""
            tot = frames + halo_of(m)
            q = (tot + S - 1) // S
            out = []
            for j in range(S):
                lo, hi = j * q, min((j + 1) * q, tot)
                out.append((lo, max(lo, hi), max(lo, halo_of(m))))
            return out

        class Pipe:
            """the sharded step over a given shard size (the weak leg: n per rank; the strong leg: strong_frames / world)"""

            def __init__(self, frames, sp_, sd_):
                self.frames, self.sp, self.sd = frames, sp_, sd_
                self.bufs = [torch.empty((frames + halo, 2304), dtype=torch.int16, device=eng.tdev) for _ in range(2)]
                self.free_ev = [None, None]                 # "the exchange that read this buffer is done"
                self.step_no = 0
                self.gathered = None
                if rank == 0:
                    self.gathered = torch.empty((world * frames, 4608), dtype=torch.uint8, device=eng.tdev if nccl else "cpu")

            def step(self, exchange=True):
                b = self.step_no & 1
                self.step_no += 1
                pcm_b = self.bufs[b]
                pcm_u8 = pcm_b.view(torch.uint8)
                cur = torch.cuda.current_stream()
                if self.free_ev[b] is not None:
                    cur.wait_event(self.free_ev[b])         # the decode two steps on must not overwrite PCM still being sent
                state.zero_()
                for j, (lo, hi, a) in enumerate(pieces(rank, self.frames)):
                    if hi > lo:
                        eng.decode(self.sp[lo:hi], self.sd[lo:hi], pcm_b[lo:hi], state=state, chunk_frames=args.chunk)
                    if not exchange:
                        continue
                    if rank == 1 and os.environ.get("PDMP3_BENCH_TEST_HANG"):   # (tests/test_gpu_multi.py: a peer that never sends)
                        time.sleep(1e6)
                    if nccl:
                        ev = torch.cuda.Event()
                        ev.record(cur)
                        comm_stream.wait_event(ev)
                        with torch.cuda.stream(comm_stream):
                            if rank == 0:
                                if hi > a:
                                    self.gathered[a:hi].copy_(pcm_u8[a:hi], non_blocking=True)
                                ops = []
                                for m in range(1, world):
                                    mlo, mhi, ma = pieces(m, self.frames)[j]
                                    if mhi > ma:
                                        ops.append(dist.P2POp(dist.irecv, self.gathered[m * self.frames + ma - 2:m * self.frames + mhi - 2], m))
                            else:
                                ops = [dist.P2POp(dist.isend, pcm_u8[a:hi], 0)] if hi > a else []
                            for r_ in (dist.batch_isend_irecv(ops) if ops else []):
                                r_.wait()                   # (stream-ordered for RCCL: comm_stream waits, the host does not)
                    else:                                   # gloo on the one GPU of the test box: through host memory, blocking
                        torch.cuda.synchronize()
                        if rank == 0:
                            if hi > a:
                                self.gathered[a:hi] = pcm_u8[a:hi].cpu()
                            for m in range(1, world):
                                mlo, mhi, ma = pieces(m, self.frames)[j]
                                if mhi > ma:
                                    dist.recv(self.gathered[m * self.frames + ma - 2:m * self.frames + mhi - 2], m)
                        elif hi > a:
                            dist.send(pcm_u8[a:hi].cpu(), 0)
                if exchange and nccl:
                    self.free_ev[b] = torch.cuda.Event()
                    self.free_ev[b].record(comm_stream)

            def timed(self, warm, steps, exchange):
                for _ in range(warm):
                    self.step(exchange)
                torch.cuda.synchronize()
                dist.barrier()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                t0_ = time.perf_counter()
                e0.record()
                for _ in range(steps):
                    self.step(exchange)
                e1.record()
                torch.cuda.synchronize()
                dist.barrier()
                torch.cuda.synchronize()
                dt_ = time.perf_counter() - t0_
                tt_ = torch.tensor([dt_], dtype=torch.float64, device=coll_dev)
                dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
                return float(tt_.item()), e0.elapsed_time(e1) / steps

        pipe = Pipe(n, spectra, side)
        pcm = pipe.bufs[0]
        tiny = torch.zeros(16, dtype=torch.uint8, device=coll_dev)     # communicator set-up is not part of any timed exchange
        dist.gather(tiny, [torch.empty_like(tiny) for _ in range(world)] if rank == 0 else None, dst=0)
        dt_dec, kern_ms = pipe.timed(args.warmup, args.steps, exchange=False)
        pipeline_error = None
        # Everything from here to the end of the N > 1 legs exchanges PCM between ranks (point-to-point sends the
        # rounds before this one never issued on hardware).  Should that ever hang instead of raising, the run still ends
        # with a valid line: after PDMP3_BENCH_WATCHDOG_S seconds (default 150; the legs take a few) rank 0 prints the
        # decode-only measurement above -- labelled as such -- and every rank exits.
        import threading
        legs_done = threading.Event()

        def watchdog():
            limit = float(os.environ.get("PDMP3_BENCH_WATCHDOG_S", "150"))
            if legs_done.wait(limit):
                return
            if rank == 0:
                fps_ = n * world * args.steps / dt_dec
                bytes_ = (n + halo) * ALGO_BYTES_PER_FRAME
                ach_ = bytes_ / (kern_ms * 1e-3) / 1e9
                print(json.dumps({
                    "metric": "MP3 frames/sec (44.1 kHz stereo 320 kbps), transforms-only hot path", "value": round(fps_, 1),
                    "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                    "ms_per_step": round(dt_dec / args.steps * 1e3, 7), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                    "dtype": "f32", "data": "synthetic",
                    "config": {"workload": "C5 (BASELINE configs[4]) at %d GPUs: one stream of %d stereo frames, %d per GPU per step (shards by "
                                           "frame range with a 2-frame halo)" % (world, n * world, n), "frames_per_gpu": n, "seed": hex(seed)},
                    "roofline": {"bound": "hbm", "achieved": round(ach_, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach_ / HBM_PEAK_GBS, 5),
                                 "traffic": None, "avg_launch_ms": round(kern_ms, 5), "algorithmic_bytes_per_launch": bytes_},
                    "pipelined_exchange_failed": "watchdog: the exchange legs made no progress for %.0f s -- `value` is the decode-only step "
                                                 "(max over ranks), nothing was gathered, no parity check ran" % limit,
                    "value_decode_only": round(fps_, 1)}), flush=True)
            os._exit(0)

        threading.Thread(target=watchdog, daemon=True).start()
        try:
            if os.environ.get("PDMP3_BENCH_NO_PIPELINE"):
                raise RuntimeError("PDMP3_BENCH_NO_PIPELINE is set")
            dt, _ = pipe.timed(args.warmup, args.steps, exchange=True)  # <- the measured steps: decode + exchange, overlapped
        except Exception as e:                                           # (never seen on hardware: the line then says so and falls
            pipeline_error = repr(e)                                     #  back to rounds 1-5's decode-only step, the gather after it)
            dt = dt_dec
        kernel_name = eng.last_launch_kernel()
    else:
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        # HIP events on the launch stream (torch's current stream = the one handed to the C-ABI) bracket the K
        # back-to-back launches: mean launch duration = elapsed / K (the stream never idles: the host enqueues a
        # step in a few microseconds, a step runs ~50).
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(args.steps):
            step()
        ev1.record()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        kern_ms = ev0.elapsed_time(ev1) / args.steps
        kernel_name = eng.last_launch_kernel()

    parity = None
    per_rank = None
    strong = None
    if world > 1:
        # what every rank ran, for the reader of a scaling curve: the kernel (the engine's choice, read back), the time of
        # one step's kernels from the rank's own HIP events (decode-only loop), the device
        mine_info = {"rank": rank, "device": dev_index, "kernel": kernel_name, "decode_ms_per_step": round(kern_ms, 5),
                     "frames": n, "halo_frames": halo, "first_frame": first, "slices": S}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine_info)
        # the exchange ALONE, once, un-overlapped and in one piece (what rounds 1-5 reported as gather_ms)
        mine = pipe.bufs[(pipe.step_no - 1) & 1][halo:].contiguous().view(torch.uint8).to(coll_dev)   # RCCL has no int16: bytes
        bufs = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
        torch.cuda.synchronize()
        dist.barrier()
        g0 = time.perf_counter()
        dist.gather(mine, bufs, dst=0)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3
        gather_bytes = int(mine.numel()) * (world - 1)                  # what crosses links into rank 0
        if rank == 0:
            # the PCM the LAST MEASURED STEP's pipeline put together on rank 0 (not the one-piece gather's): parity is of the
            # thing that was timed
            plain = torch.cat([b_.cpu() for b_ in bufs]).numpy().view(np.int16).reshape(-1, 2304)
            gathered = plain if pipeline_error else pipe.gathered.cpu().numpy().view(np.int16).reshape(-1, 2304)
            same_as_plain = bool(np.array_equal(gathered, plain))
            if args.dump_gathered:
                np.save(args.dump_gathered, gathered)
            try:
                parity = boundary_parity(gathered, seed, n, world)
                parity["pipelined_gather_equals_one_piece_gather"] = same_as_plain
            except Exception as e:                                       # a check beside the number, never fatal
                parity = {"error": repr(e)}
            del gathered, plain
        del bufs, mine
        # strong scaling: the FIXED stream of BASELINE configs[4] (1 M frames) over this many GPUs, the same pipelined step
        ns = args.strong_frames // world if args.strong_frames else 0
        if ns and ns != n and not pipeline_error:
            try:
                pipe = None
                spectra = side = None
                torch.cuda.empty_cache()
                sp_s = torch.empty((ns + halo, 2, 2, 576), dtype=torch.int16, device=eng.tdev)
                sd_s = torch.zeros((ns + halo, 4, 128), dtype=torch.uint8, device=eng.tdev)
                eng.generate(seed, rank * ns - halo, ns + halo, sp_s, sd_s)
                torch.cuda.synchronize()
                ps = Pipe(ns, sp_s, sd_s)
                dts, _ = ps.timed(1, 3, exchange=True)
                strong = {"stream_frames": ns * world, "frames_per_gpu": ns, "steps": 3, "ms_per_step": round(dts / 3 * 1e3, 4),
                          "frames_per_s": round(ns * world * 3 / dts, 1), "scaling": "strong",
                          "what": "the fixed %d-frame stream cut into %d frame ranges, decode + exchange pipelined as in the headline" % (ns * world, world)}
                del ps, sp_s, sd_s
            except Exception as e:
                strong = {"error": repr(e)}
        elif ns:
            strong = {"stream_frames": ns * world, "frames_per_gpu": ns, "scaling": "strong", "same_as": "the headline: at this N the weak-scaling shard IS the 1 M-frame stream / N"}
        legs_done.set()
    else:
        gather_ms = None

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    total_frames = n * world * args.steps
    fps = total_frames / dt
    launch_bytes = (n + halo) * ALGO_BYTES_PER_FRAME
    traffic, traffic_src = measured_traffic(n, halo)
    achieved = launch_bytes / (kern_ms * 1e-3) / 1e9
    out = {
        "metric": "MP3 frames/sec (44.1 kHz stereo 320 kbps), transforms-only hot path",
        "value": round(fps, 1),
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 7),        # (seven decimals: 0.0189212, not 0.02 -- the consistency check deserves more than one digit)
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": ("C2 (BASELINE configs[1]): %d stereo frames = %d granules per GPU per step, "
                         "pre-Huffman-decoded spectra resident in HBM, one stream" % (n, 2 * n)) if world == 1 else
                        ("C5 (BASELINE configs[4]) at %d GPUs: one stream of %d stereo frames, %d per GPU per step (shards by "
                         "frame range with a 2-frame halo), spectra generated on each GPU and resident in HBM; PCM gathered "
                         "to rank 0 over RCCL after the timed region" % (world, n * world, n)),
            "frames_per_gpu": n, "granules_per_gpu": 2 * n, "seed": hex(seed),
            "chunk_frames": args.chunk or "auto", "sharding": "frame-range x%d, no data-path collective" % world,
        },
        "x_realtime": round(fps / RT_FRAMES_PER_S, 1),
        "roofline": {
            "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
            # (which kernel that was is the engine's decision -- the granule kernel for launches of up to
            #  PDMP3_HIP_GRAN_MAX frames, 12288 on an MI355X, else independent chunks -- and is read back from it:
            #  pdmp3_hip_last_launch_kind)
            "kernel": kernel_name, "avg_launch_ms": round(kern_ms, 5),
            "algorithmic_bytes_per_launch": launch_bytes,
        },
    }
    # "MFMA roof is cited ... if IMDCT/matrixing are run as dense fp32 contractions": they are, so the same
    # launch against the fp32 roof, in direct-form-equivalent (algorithmic) flops -- the bound that actually binds:
    # 57 flop/B against a machine balance of ~20 flop/B
    tf = (n + halo) * ALGO_FLOP_PER_FRAME / (kern_ms * 1e-3) / 1e12
    out["roofline_fp32"] = {"bound": "mfma", "achieved": round(tf, 2), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(tf / FP32_PEAK_TFLOPS, 5), "algorithmic_flop_per_frame": ALGO_FLOP_PER_FRAME,
                            "dtype": "f32 (v_mfma_f32_16x16x4_f32 + VALU)",
                            # `frac` divides DIRECT-FORM flops (what the reference's loops would execute) by time: the kernel runs the
                            # folded transforms, a quarter of those flops -- it is an algorithmic rate, not a utilisation.  What the
                            # matrix pipe really does, from the PMC pass of this kernel (profiles/*_pmc_summary.json):
                            "what_frac_is": "direct-form-equivalent flops / time / peak: an algorithmic rate, NOT the matrix pipe's utilisation",
                            "mfma_busy_frac": round(MFMA_BUSY_FRAC, 4) if (MFMA_BUSY_FRAC and traffic) else None}
    if world == 1:
        # BASELINE.json's metric, second half ("PCM max-abs-diff vs ref"): the PCM the last timed step left in HBM, all
        # of it, against the CPU decoder on the same generated frames.  After the timed region; the checker is never timed.
        try:
            dec, against, o = parity_checker()
            sp_h, sd_h = o.generate(seed, first, n)
            parity = parity_of(pcm.cpu().numpy(), dec.decode(sp_h, sd_h), n, against,
                               "the whole batch of the last timed step (int16 PCM, P:2028-2031)")
        except Exception as e:
            parity = {"error": repr(e)}
    out["parity"] = parity
    c5_pcm = extra_legs.pop("_c5_pcm", None)
    if c5_pcm is not None:
        # the C5 shard's own parity (after every timed region): its first 256 frames -- the shard boundary, decoded from the
        # 2-frame halo -- and its last 2 against the CPU decoder started cold 8 frames earlier (the state is 2 deep)
        try:
            import numpy as np
            dec, against, o = parity_checker()
            ns, head, tail = c5_pcm
            got, want = [], []
            for a, b, g in ((ns, ns + head.shape[0], head), (2 * ns - 2, 2 * ns, tail)):
                w0 = max(0, a - 8)
                sp_h, sd_h = o.generate(SEED_C5, w0, b - w0)
                sd_h["frame"][0] |= 0x40                            # PDMP3_FR_RESET
                want.append(dec.decode(sp_h, sd_h)[a - w0:])
                got.append(g)
            extra_legs["c5_shard_1gpu"]["parity"] = parity_of(
                np.concatenate(got), np.concatenate(want), head.shape[0] + 2, against,
                "the shard's first %d frames (behind the 2-frame halo) and its last 2" % head.shape[0])
        except Exception as e:
            extra_legs["c5_shard_1gpu"]["parity"] = {"error": repr(e)}
    if per_rank is not None:
        out["ranks"] = per_rank
        out["rccl_ranks"] = dist.get_world_size() if backend == "nccl" else 0
        out["collective_backend"] = "rccl" if backend == "nccl" else backend
    if gather_ms is not None:
        step_ms, dec_ms = dt / args.steps * 1e3, dt_dec / args.steps * 1e3
        if pipeline_error:
            out["pipelined_exchange_failed"] = pipeline_error + " -- `value` is the decode-only step, the gather ran after it"
        out["value_decode_only"] = round(n * world * args.steps / dt_dec, 1)
        out["ms_per_step_decode_only"] = round(dec_ms, 5)
        out["slices"] = S
        # 1 = the shorter of the two is completely hidden under the longer one, 0 = they ran one after the other
        out["overlap_frac"] = round(max(0.0, min(1.0, (dec_ms + gather_ms - step_ms) / max(1e-9, min(dec_ms, gather_ms)))), 3)
        out["gather_model"] = ("DESIGN.md section 5: (N - 1) x shard bytes into one GPU over N - 1 xGMI links of ~153 GB/s peak each; at N = 8, "
                               "7 x 576 MB in ~3.8 ms at link peak against ~0.93 ms of decode per shard: the exchange, not the kernel, bounds the step")
        if strong is not None:
            out["strong_scaling"] = strong
        out["gather_ms"] = round(gather_ms, 3)
        out["gather_bytes"] = gather_bytes
        out["gather_GBps"] = round(gather_bytes / (gather_ms * 1e-3) / 1e9, 2)
        out["gather_backend"] = "rccl" if backend == "nccl" else backend

    out.update(extra_legs)
    if from_idle:
        out["from_idle_gpu"] = from_idle
    if extra_legs:
        out["clocks"] = ("busy: the legs on launches of 125 000 / 131 072 frames (c5_shard_1gpu, roofline_large_batch, roofline_float_pcm: "
                         "about 60 ms of launches) ran before the warm-up and the timed region; `from_idle_gpu` = the same W + K launches measured first, "
                         "from an idle GPU: the figure to compare across rounds")

    if world == 1 and not args.no_e2e:
        # beside the hot-path metric: the same path fed from a bitstream in host memory to PCM in host memory (host scan ->
        # device Huffman -> transforms; PCIe both ways and the host stages included).  A reported extra, never `value`.
        try:
            import numpy as np
            from pdmp3_amd.packer import packer
            from pdmp3_amd import api
            nf = 137813                                      # BASELINE configs[2] (C3): one hour of audio
            mp3 = np.frombuffer(packer.generate(n_frames=nf, seed=0xC3, sfreq=0, mode=1, mode_ext=2, bitrate_index=14), dtype=np.uint8)
            total, frames = api.scan_buffer(mp3)
            pcm_out = np.empty(total // 2, dtype=np.int16)
            bd = api.BulkDecoder(threads=0)
            best = None
            for _ in range(4):
                t0 = time.perf_counter()
                got, _, _ = bd.decode_into(mp3, pcm_out)
                dt_e = time.perf_counter() - t0
                best = dt_e if best is None else min(best, dt_e)
            assert got == total
            # the same stream with the PCM (i) into pinned host memory -- no host-side copy, what is left is PCIe: 4608 B per
            # frame down at 50-odd GB/s is the ceiling of ANY destination in host memory, ~11 M frames/s -- and (ii) left in HBM
            # (a device pointer as destination), where the host's sequential scan of the stream is the bound
            pin = api.PinnedPCM(total // 2)
            best_pin = None
            for _ in range(4):
                t0 = time.perf_counter()
                bd.decode_into(mp3, pin.array)
                dt_e = time.perf_counter() - t0
                best_pin = dt_e if best_pin is None else min(best_pin, dt_e)
            dout = torch.empty(total // 2, dtype=torch.int16, device=eng.tdev)
            best_dev, all_dev = None, []
            for _ in range(8):                               # (4 ms each; the first ones find the decoder's scan threads asleep)
                t0 = time.perf_counter()
                bd.decode_into_device(mp3, dout, wait=True)
                dt_e = time.perf_counter() - t0
                all_dev.append(dt_e)
                best_dev = dt_e if best_dev is None else min(best_dev, dt_e)
            same = bool(np.array_equal(dout.cpu().numpy(), pcm_out)) and bool(np.array_equal(pin.array[:total // 2], pcm_out))
            pin.free()
            del dout
            scans = list(bd.split_scans())                   # [streams the split scan decoded, streams it gave up]
            bd.close()
            out["end_to_end"] = {"workload": "C3-style stream (44.1 kHz joint stereo 320 kbps CBR), %d frames, include/pdmp3_bulk.h" % frames,
                                 "frames_per_s": round(frames / best, 1), "x_realtime": round(frames / best / RT_FRAMES_PER_S, 1),
                                 "seconds": round(best, 4), "host_threads": bd.threads + 2, "huffman": "device",
                                 "includes": "host header/side-info/reservoir scan, H2D, k_rows, k_unpack, k_merge_outcome, k_merge_apply, k_decode_g, D2H, copy to pageable memory; best of 4 runs (of 8 with the PCM left in HBM)",
                                 "pcm_to_pinned_host": {"frames_per_s": round(frames / best_pin, 1), "seconds": round(best_pin, 4),
                                                        "pcie_GBps": round(frames * 4608 / best_pin / 1e9, 1)},
                                 "pcm_left_in_hbm": {"frames_per_s": round(frames / best_dev, 1), "seconds": round(best_dev, 4),
                                                     "median_frames_per_s": round(frames / sorted(all_dev)[len(all_dev) // 2], 1), "runs": len(all_dev),
                                                     "split_scans": scans},
                                 "three_destinations_same_pcm": same}
        except Exception as e:
            out["end_to_end"] = {"error": repr(e)}
        # BASELINE configs[2] (C3) through the drop-in API itself: pdmp3_feed / pdmp3_read driven by a C loop
        # (pdmp3_amd_stream_loop) at the reference driver's cadence (4096-byte feeds on PDMP3_NEED_MORE, 16 KiB reads),
        # and with a caller that keeps the 16 KiB ring full.  One host thread, synchronous calls: a reported extra.
        try:
            nf = 20000
            mp3s = np.frombuffer(packer.generate(n_frames=nf, seed=0xC3, sfreq=0, mode=1, mode_ext=2, bitrate_index=14), dtype=np.uint8)
            res = {}
            for key, (feed, read, eager) in (("reference_cadence", (4096, 16384, False)), ("ring_kept_full", (4096, 65536, True))):
                api.stream_loop(mp3s[:200000], feed, read, eager, want_pcm=False)      # warm-up (engine, streams)
                t0 = time.perf_counter()
                nbytes, _ = api.stream_loop(mp3s, feed, read, eager, want_pcm=False)
                dt_s = time.perf_counter() - t0
                fr = nbytes / 4608.0
                res[key] = {"frames_per_s": round(fr / dt_s, 1), "x_realtime": round(fr / dt_s / RT_FRAMES_PER_S, 1),
                            "frames": int(fr), "seconds": round(dt_s, 4), "feed_bytes": feed, "read_bytes": read}
            out["streaming_api"] = {"workload": "C3-style stream (44.1 kHz joint stereo 320 kbps CBR) through pdmp3_feed / pdmp3_read "
                                                "(include/pdmp3.h), host Huffman by the calling thread and the library's helper threads "
                                                "(PDMP3_STREAM_THREADS, default 3), read-ahead batches of up to 16 frames", **res}
        except Exception as e:
            out["streaming_api"] = {"error": repr(e)}
    if cpu is not None:
        out["cpu_baseline"] = cpu
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

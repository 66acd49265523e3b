#!/usr/bin/env python3
"""Throughput of the bulk pipeline (include/pdmp3_bulk.h) on a long synthetic
stream (44.1 kHz joint stereo 320 kbps).  Host stages alone
(--parse-only, runs anywhere) or end to end on the GPU box.

  python tools/bulk_bench.py --frames 137813 --threads 1,8,32,64
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")     # before HIP starts (torch brings it up): see host/stream_api.c shared_ctx_on


def c4(args, api):
    """files dealt largest-first to JOBS decoders (pdmp3_amd.sharding.assign_files), each on its own host thread
    with its own HIP streams; decoders and output buffers exist before the clock starts"""
    import threading
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from test_gpu_corpus import _c4_files
    from pdmp3_amd.sharding import assign_files
    files = _c4_files(4096, 64, 12)
    files = [np.frombuffer(f, dtype=np.uint8) for f in files + files[:10]]
    sizes = [api.scan_buffer(f) for f in files]
    pins = [api.PinnedPCM(max(b, 2) // 2) for b, _ in sizes] if args.pinned else []
    outs = [p.array for p in pins] if args.pinned else [np.zeros(max(b, 2) // 2, dtype=np.int16) for b, _ in sizes]
    plan = assign_files([len(f) for f in files], args.c4)
    import torch
    ngpu = max(1, min(args.gpus, torch.cuda.device_count()))
    decs = [api.BulkDecoder(threads=2, window_frames=args.window, host_huffman=args.host_huffman, device=j % ngpu)
            for j in range(args.c4)]

    douts = [torch.empty(max(b, 2) // 2, dtype=torch.int16, device="cuda:%d" % (0)) for b, _ in sizes] if args.device_out else None

    calls = {}

    def work(j):
        t_calls = []
        for i in plan[j]:                                  # back to back, one wait at the end
            t0 = time.perf_counter()
            if douts is not None:
                got, _, _ = decs[j].decode_into_device(files[i], douts[i], wait=False)
            else:
                got, _, _ = decs[j].decode_into_async(files[i], outs[i])
            t_calls.append(time.perf_counter() - t0)
            assert got == sizes[i][0]
        t0 = time.perf_counter()
        decs[j].wait()
        calls[j] = (t_calls, time.perf_counter() - t0)
    best = None
    for _ in range(args.reps + 1):                      # first pass = warm-up
        ts = [threading.Thread(target=work, args=(j,)) for j in range(args.c4)]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    for d in decs:
        d.close()
    if os.environ.get("BULK_BENCH_VERBOSE"):               # the last pass: where a decoder's time went
        for j, (tc, tw) in sorted(calls.items()):
            print("decoder %d: %d files, calls %.2f ms in all (median %.0f us, longest %.0f us), final wait %.2f ms" %
                  (j, len(tc), sum(tc) * 1e3, sorted(tc)[len(tc) // 2] * 1e6, max(tc) * 1e6, tw * 1e3), file=sys.stderr)
    frames = sum(fr for _, fr in sizes)
    print(json.dumps({"workload": "C4: %d files, %d frames, mono/stereo/joint x 32/44.1/48 kHz x CBR/VBR x block mixes" % (len(files), frames),
                      "decoders": args.c4, "gpus": ngpu, "seconds": round(best, 4), "frames_per_s": round(frames / best, 1),
                      "mp3_bytes": int(sum(len(f) for f in files)), "pcm_bytes": int(sum(b for b, _ in sizes)),
                      "mode": "host Huffman" if args.host_huffman else "device Huffman", "pcm": "device" if args.device_out else "pinned" if args.pinned else "pageable",
                      "host_cpus": os.cpu_count()}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=20000)
    ap.add_argument("--threads", default="1,4,8")
    ap.add_argument("--window", type=int, default=0)
    ap.add_argument("--parse-only", action="store_true")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--c4", type=int, default=0, metavar="JOBS",
                    help="SURVEY 8d C4 instead: the mixed corpus (64 files x >= 4096 frames), JOBS decoders in parallel")
    ap.add_argument("--device-out", action="store_true", help="PCM into device memory (torch tensors): it never leaves the GPU")
    ap.add_argument("--pinned", action="store_true", help="PCM into pinned host buffers (pdmp3_amd_pcm_alloc): no host copy")
    ap.add_argument("--gpus", type=int, default=1, help="--c4: decoder j runs on GPU j %% GPUS")
    ap.add_argument("--host-huffman", action="store_true", help="scalefactors + Huffman on the host pool instead of the device")
    args = ap.parse_args()
    from pdmp3_amd.packer import packer
    from pdmp3_amd import api
    if args.c4:
        return c4(args, api)
    t0 = time.perf_counter()
    mp3 = packer.generate(n_frames=args.frames, seed=0xC3, sfreq=0, mode=1, mode_ext=2, bitrate_index=14)
    a = np.frombuffer(mp3, dtype=np.uint8)
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    total, frames = api.scan_buffer(a)
    t_scan = time.perf_counter() - t0
    rt = 44100.0 / 1152.0
    out = {"frames": frames, "mp3_bytes": len(mp3), "pcm_bytes": total, "packer_s": round(t_gen, 2),
           "scan_ms": round(t_scan * 1e3, 2), "scan_frames_per_s": round(frames / t_scan, 1), "host_cpus": os.cpu_count(),
           "runs": []}
    pin = api.PinnedPCM(total // 2) if args.pinned else None
    pcm = pin.array if pin else np.empty(total // 2, dtype=np.int16)
    dout = None
    if args.device_out:                        # PCM stays in HBM (a torch tensor as the destination)
        import torch
        dout = torch.empty(max(total, 2) // 2, dtype=torch.int16, device="cuda:0")
    for th in [int(x) for x in args.threads.split(",")]:
        b = api.BulkDecoder(threads=th, window_frames=args.window, parse_only=args.parse_only, host_huffman=args.host_huffman)
        best = None
        for _ in range(args.reps):
            t0 = time.perf_counter()
            if args.parse_only:
                b.parse(a)
            elif dout is not None:
                got, _, _ = b.decode_into_device(a, dout, wait=True)
                assert got == total
            else:
                got, _, _ = b.decode_into(a, pcm)
                assert got == total
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        b.close()
        out["runs"].append({"threads": th, "seconds": round(best, 4), "frames_per_s": round(frames / best, 1),
                            "x_realtime": round(frames / best / rt, 1), "mode": "parse" if args.parse_only else ("decode, host Huffman" if args.host_huffman else "decode, device Huffman"),
                            "pcm": "device" if dout is not None else "pinned" if args.pinned else "pageable"})
    print(json.dumps(out))


if __name__ == "__main__":
    main()

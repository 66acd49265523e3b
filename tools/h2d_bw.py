"""Host-to-device copy rate from pinned memory against the copy's size, one copy at a time and two side by side on two
streams (the whole-stream decoder uploads ~9 MB per window of 8192 frames): what the link gives this process.
  python3 tools/h2d_bw.py"""
import json
import time

import torch


def main():
    dev = torch.device("cuda:0")
    out = []
    for mb in (1, 2, 4, 9, 18, 36, 72):
        n = mb << 20
        h = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(2)]
        d = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(2)]
        s = [torch.cuda.Stream(), torch.cuda.Stream()]
        for k in (1, 2):
            best = None
            for rep in range(12):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(k):
                    with torch.cuda.stream(s[i]):
                        d[i].copy_(h[i], non_blocking=True)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            out.append({"MB_each": mb, "copies": k, "us": round(best * 1e6, 1), "GBps": round(k * n / best / 1e9, 1)})
    # the same amount as one copy or as four quarter-size copies on one stream
    n = 9 << 20
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    for parts in (1, 2, 4):
        best = None
        for rep in range(12):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            q = n // parts
            for i in range(parts):
                d[i * q:(i + 1) * q].copy_(h[i * q:(i + 1) * q], non_blocking=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out.append({"MB_total": 9, "parts_one_stream": parts, "us": round(best * 1e6, 1), "GBps": round(n / best / 1e9, 1)})
    print(json.dumps(out))


if __name__ == "__main__":
    main()

"""GPU timeline of the whole-stream decoder from a rocprofv3 --kernel-trace --memory-copy-trace run of tools/bulk_bench.py:
per decode (runs of kernels separated by idle gaps > 0.4 ms) the span, the time some kernel was running, the sum per
kernel, and where the GPU had nothing to do.
  python3 tools/bulk_timeline.py gpurun_out/prof_bulk_tl2/runc/230   [--list]"""
import csv
import sys


def main():
    pre = sys.argv[1]
    k = list(csv.DictReader(open(pre + "_kernel_trace.csv")))
    m = list(csv.DictReader(open(pre + "_memory_copy_trace.csv")))
    ker = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0].split()[-1][-14:],
                  int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) for r in k)
    cop = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in m)
    cuts = [0]
    for i in range(1, len(ker)):
        if ker[i][0] - max(e[1] for e in ker[max(0, i - 8):i]) > 400_000:
            cuts.append(i)
    cuts.append(len(ker))
    for a, b in zip(cuts[:-1], cuts[1:]):
        seg = ker[a:b]
        if len(seg) < 40:
            continue
        t0, t1 = seg[0][0], max(e[1] for e in seg)
        busy, gaps = 0, []
        cs, ce = seg[0][0], seg[0][1]
        for s, e, n, g in seg[1:]:
            if s > ce:
                busy += ce - cs
                gaps.append((cs, ce, s))
                cs, ce = s, e
            else:
                ce = max(ce, e)
        busy += ce - cs
        per = {}
        for s, e, n, g in seg:
            per[n] = per.get(n, 0) + e - s
        h2d = sum(e - s for s, e in cop if t0 - 300_000 <= s <= t1)
        print("decode: %d kernels, span %.2f ms, some kernel running %.2f ms, H2D copies %.2f ms | %s" % (
            len(seg), (t1 - t0) / 1e6, busy / 1e6, h2d / 1e6, ", ".join("%s %.2f" % (n, v / 1e6) for n, v in sorted(per.items(), key=lambda x: -x[1]))))
        idle = [(s2 - e1, (e1 - t0) / 1e6) for _, e1, s2 in gaps]
        big = sorted(idle, reverse=True)[:8]
        print("   idle %.2f ms in %d gaps; the longest (us at ms): %s" % (sum(i for i, _ in idle) / 1e6, len(idle), ", ".join("%.0f@%.2f" % (i / 1e3, at) for i, at in big)))
        # idle by quarter of the span
        q = [0, 0, 0, 0]
        for i, at in idle:
            q[min(3, int(at / ((t1 - t0) / 1e6) * 4))] += i
        print("   idle by quarter of the span: %s ms" % ", ".join("%.2f" % (x / 1e6) for x in q))
        if "--list" in sys.argv:
            for s, e, n, g in seg:
                print("   %8.3f %8.3f %7.1f us %-14s %d wgs" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e3, n, g))


if __name__ == "__main__":
    main()

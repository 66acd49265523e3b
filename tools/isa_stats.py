"""Instruction-class counts per basic block of one kernel in a hipcc -save-temps .s file.

    python tools/isa_stats.py file.s _Z8k_decodeILb0EE   [--blocks]

Counts are static (per block, not weighted by trip count); --blocks lists every block so that the hot granule
loop can be read off by its labels.  Development tool, not part of the product.
"""
import re
import sys
from collections import Counter, OrderedDict


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, kern = sys.argv[1], sys.argv[2]
    show_blocks = "--blocks" in sys.argv
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(kern) and l.rstrip().endswith(":") or
                 (l.startswith(kern) and ":" in l.split(";")[0]))
    blocks = OrderedDict()
    cur = "entry"
    blocks[cur] = Counter()
    ops = Counter()
    for l in lines[start + 1:]:
        s = l.split(";")[0].strip()
        if not s:
            continue
        if s.startswith(".Lfunc_end") or s.startswith("s_endpgm"):
            if s.startswith("s_endpgm"):
                blocks[cur]["salu"] += 1
                continue
            break
        m = re.match(r"^(\.LBB[0-9_]+):", s)
        if m:
            cur = m.group(1)
            blocks[cur] = Counter()
            continue
        if s.startswith("."):
            continue
        op = s.split()[0]
        c = classify(op)
        blocks[cur][c] += 1
        ops[op] += 1
        if c == "branch":
            blocks[cur]["->" + s.split()[-1]] += 0
    tot = Counter()
    for b, c in blocks.items():
        for k, v in c.items():
            if not k.startswith("->"):
                tot[k] += v
    print("total", dict(tot))
    if show_blocks:
        for b, c in blocks.items():
            n = sum(v for k, v in c.items() if not k.startswith("->"))
            tg = [k for k in c if k.startswith("->")]
            print("%-12s n=%4d  valu=%4d mfma=%3d lds=%3d vmem=%3d salu=%3d smem=%3d wait=%3d nop=%3d  %s" % (
                b, n, c["valu"], c["mfma"], c["lds"], c["vmem"], c["salu"], c["smem"], c["wait"], c["nop"], " ".join(tg)))
    if "--ops" in sys.argv:
        for op, n in ops.most_common(60):
            print("%6d %s" % (n, op))


if __name__ == "__main__":
    main()

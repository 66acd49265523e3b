"""Static instruction counts of a kernel's LOOP bodies (the blocks clang marks "in Loop: Header=...") and of the whole kernel,
from a hipcc -save-temps .s file: VALU (of them v_mov, v_readlane / v_writelane = SGPR spill traffic), MFMA, LDS, VMEM, SMEM,
SALU, waits -- and the kernel's register figures (.vgpr_count, spilled SGPRs / VGPRs) from its metadata.  Before / after tables
of a round's kernels: profiles/r05_isa_stats.txt.

    python tools/isa_loop_stats.py file.s KERNEL_SUBSTRING [more kernels ...]
"""
import re
import sys
from collections import Counter, OrderedDict

CLASSES = ["valu", "v_mov", "v_lane", "mfma", "lds", "vmem", "smem", "salu", "wait", "branch"]


def classify(op):
    c = []
    if op.startswith(("v_mfma", "v_smfmac")):
        return ["mfma"]
    if op.startswith("v_"):
        c.append("valu")
        if op.startswith("v_mov_b") or op.startswith("v_accvgpr") or op.startswith("v_pk_mov"):
            c.append("v_mov")
        if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            c.append("v_lane")
        return c
    if op.startswith("ds_"):
        return ["lds"]
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return ["vmem"]
    if op.startswith(("s_load", "s_buffer_load")):
        return ["smem"]
    if op.startswith("s_waitcnt"):
        return ["wait"]
    if op.startswith(("s_cbranch", "s_branch")):
        return ["branch"]
    if op.startswith("s_nop"):
        return []
    if op.startswith("s_"):
        return ["salu"]
    return []


def kernel_text(lines, sub):
    start = next(i for i, l in enumerate(lines) if l.startswith(sub) and ":" in l.split(";")[0])
    name = lines[start].split(":")[0]
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    return name, lines[start + 1:end]


def stats(body):
    total, loops = Counter(), OrderedDict()
    in_loop = None
    for l in body:
        m = re.match(r"^(\.LBB[0-9_]+):\s*(;.*)?$", l)
        if m:
            c = m.group(2) or ""
            h = re.search(r"Header=(BB[0-9_]+)", c)
            if "Loop Header" in c and "Depth=1" in c:
                in_loop = m.group(1)[2:]          # this block IS a depth-1 header
            elif h:
                pass                              # stays in the loop it names (inner blocks name their own header)
            else:
                in_loop = None
            continue
        if re.match(r"^\s*;.*in Loop: Header=(BB[0-9_]+) Depth=1", l):
            in_loop = re.search(r"Header=(BB[0-9_]+)", l).group(1)
            continue
        s = l.split(";")[0].strip()
        if not s or s.startswith("."):
            continue
        op = s.split()[0]
        cl = classify(op)
        for k in cl:
            total[k] += 1
        total["all"] += 1
        if in_loop:
            d = loops.setdefault(in_loop, Counter())
            for k in cl:
                d[k] += 1
            d["all"] += 1
    return total, loops


def meta(lines, name):
    """the kernel's entry in the amdhsa.kernels metadata: from the list item that holds its .name to the next item"""
    out = {}
    idx = [k for k, l in enumerate(lines) if re.match(r"\s*\.name:\s+" + re.escape(name) + r"\s*$", l)]
    if not idx:
        return out
    i = idx[0]
    lo = max(k for k in range(i, -1, -1) if lines[k].lstrip().startswith("- ."))
    hi = next((k for k in range(i + 1, len(lines)) if lines[k].lstrip().startswith("- .") or lines[k].startswith("amdhsa.")), len(lines))
    for l in lines[lo:hi]:
        m = re.match(r"\s*-?\s*\.(sgpr_count|sgpr_spill_count|vgpr_count|vgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size):\s*(\d+)", l)
        if m:
            out[m.group(1)] = int(m.group(2))
    return out


def main():
    lines = open(sys.argv[1]).read().split("\n")
    for sub in sys.argv[2:]:
        name, body = kernel_text(lines, sub)
        total, loops = stats(body)
        print("%s" % name)
        print("  registers: %s" % ", ".join("%s %d" % kv for kv in sorted(meta(lines, name).items())))
        print("  %-26s %s" % ("", " ".join("%7s" % c for c in ["all"] + CLASSES)))
        print("  %-26s %s" % ("whole kernel", " ".join("%7d" % total[c] for c in ["all"] + CLASSES)))
        for h, d in loops.items():
            if d["all"] >= 200:
                print("  %-26s %s" % ("loop at " + h, " ".join("%7d" % d[c] for c in ["all"] + CLASSES)))


if __name__ == "__main__":
    main()

"""GPU diagnostic: where does the HIP path differ from the oracle?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import pdmp3_amd
import corpus
from oracle.oracle import Oracle

eng = pdmp3_amd.Engine(0)
o = Oracle()
names = sys.argv[1:] or ["ms_long_441", "ms_mixed_blocks_441"]
for name in names:
    sp, sd = corpus.case(name, n=6)
    want, ws = o.decode(sp, sd, stages=True)
    dsp, dsd = eng.upload(sp, sd)
    n = sp.shape[0]
    pcm = torch.zeros((n, 2304), dtype=torch.int16, device=eng.tdev)
    stg = torch.zeros((n, 2, 2, 4, 576), dtype=torch.float32, device=eng.tdev)
    eng.decode_stages(dsp, dsd, pcm, stg)
    torch.cuda.synchronize()
    gs = stg.cpu().numpy(); got = pcm.cpu().numpy()
    print("==", name, "pcm maxdiff", np.abs(got.astype(int) - want).max())
    for k in range(4):
        a = ws[:, :, :, k]; b = gs[:, :, :, k]
        bad = np.argwhere(a.view(np.uint32) != b.view(np.uint32))
        print(" stage", k, "mismatches", len(bad), "max abs", np.abs(a - b).max(), "max rel", (np.abs(a - b) / (np.abs(a) + 1e-30)).max())
        for idx in bad[:6]:
            f, g, c, l = idx
            s = sd[f, g, c]
            print("   f%d g%d c%d line %d want %r (%08x) got %r (%08x) is=%d count1=%d gg=%d flags=%02x" % (
                f, g, c, l, a[f, g, c, l], a[f, g, c, l].view(np.uint32), b[f, g, c, l], b[f, g, c, l].view(np.uint32),
                sp[f, g, c, l], s["count1"], s["global_gain"], s["flags"]))

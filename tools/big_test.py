"""Launch time of the large batch for a few chunk sizes.  usage: python tools/big_test.py [n_frames]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pdmp3_amd
eng = pdmp3_amd.Engine(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
sp, sd, pcm = eng.alloc_frames(n)
eng.generate(0x5EED0000C5, 0, n, sp, sd)
for chunk in (0, 32, 64, 16):
    for _ in range(10): eng.decode(sp, sd, pcm, chunk_frames=chunk)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): eng.decode(sp, sd, pcm, chunk_frames=chunk)
    b.record(); torch.cuda.synchronize()
    try:
        kind = eng.last_launch_kernel()
    except Exception:
        kind = "?"
    print("n %d chunk %2d  %.4f ms per launch  (%s)" % (n, chunk, a.elapsed_time(b) / 20, kind))

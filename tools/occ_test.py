"""Launch time of pdmp3_hip_decode_frames against the batch size.  usage: python tools/occ_test.py [sizes...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pdmp3_amd
eng = pdmp3_amd.Engine(0)
sizes = [int(x) for x in sys.argv[1:]] or [1, 4, 16, 128, 256, 512, 1024, 1536, 2048, 3072, 4096]
for n in sizes:
    sp, sd, pcm = eng.alloc_frames(n)
    eng.generate(0x5EED0000C2, 0, n, sp, sd)
    for _ in range(20): eng.decode(sp, sd, pcm)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200): eng.decode(sp, sd, pcm)
    b.record(); torch.cuda.synchronize()
    print("n %5d  %.2f us per launch" % (n, a.elapsed_time(b) * 1e3 / 200))

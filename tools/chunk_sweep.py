import sys, torch
sys.path.insert(0, '.')
from pdmp3_amd.hip import Engine
eng = Engine()
n = 131072
sp, sd, pcm = eng.alloc_frames(n)
eng.generate(0x5EED0000C5, 0, n, sp, sd)
for chunk in (0, 16, 24, 32, 48, 64, 96, 128):
    for _ in range(3): eng.decode(sp, sd, pcm, chunk_frames=chunk)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): eng.decode(sp, sd, pcm, chunk_frames=chunk)
    b.record(); torch.cuda.synchronize()
    print("chunk %3d: %.4f ms" % (chunk, a.elapsed_time(b) / 20))

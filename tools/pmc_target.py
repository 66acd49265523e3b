"""Small fixed workload for rocprofv3 --pmc runs: a few launches of k_decode on
131072 frames (chunk 32) and on the C2 batch (2048 frames, auto chunk)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pdmp3_amd
eng = pdmp3_amd.Engine(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sp, sd, pcm = eng.alloc_frames(n)
eng.generate(0x5EED0000C5, 0, n, sp, sd)
for _ in range(4):
    eng.decode(sp, sd, pcm, chunk_frames=chunk)
torch.cuda.synchronize()

#!/usr/bin/env python3
"""PCIe-inclusive rate of the byte-stream layer on its own: `python profiles/stream_rate.py [streams] [rounds]`.
Encodes `streams` 5.1 frames on the GPU, then times ac3mi_stream_convert_many (bench.py's stream_layer leg, more rounds)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pkg = bench.importlib_pkg()
dev = torch.device("cuda:0")
eng = pkg.Engine(0)
enc = pkg.EncodeDesc(48000, 384000, 6)
g = torch.Generator(device=dev).manual_seed(5)
pcm = ((torch.rand((n, 1, 1536, 6), device=dev, generator=g) - 0.5) * 20000).round().to(torch.int16)
last = torch.zeros((n, 6, 256), dtype=torch.int16, device=dev)
csnr = torch.full((n,), 40, dtype=torch.int32, device=dev)
frames = torch.zeros((n, 1, enc.frame_bytes()), dtype=torch.uint8, device=dev)
eng.encode_batch(enc, pcm, (0, 2, 1, 4, 5, 3), last, csnr, out=frames)
torch.cuda.synchronize()
for _ in range(3):
    r = bench.stream_layer_timing(pkg, eng, frames.cpu().numpy(), rounds=rounds)
    print("%d streams: %.2f ms per round, %.3f M frames/s, all bytes used %s" % (n, r["ms_per_round"], r["frames_per_s"] / 1e6, r["all_bytes_used"]))
# the two legs a round cannot avoid, on their own: PCM device -> pinned host, and one host copy of it (single thread)
import time
import numpy as np
nbytes = n * 6 * 256 * 6 * 2
d = torch.empty(nbytes, dtype=torch.uint8, device=dev)
h = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
for _ in range(2):
    h.copy_(d, non_blocking=True); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    h.copy_(d, non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print("D2H %.1f MB pinned: %.2f ms = %.1f GB/s" % (nbytes / 1e6, dt * 1e3, nbytes / dt / 1e9))
a = h.numpy(); b = np.empty_like(a)
b[:] = a
t0 = time.perf_counter()
for _ in range(5):
    b[:] = a
dt = (time.perf_counter() - t0) / 5
print("host copy of the same bytes, one thread: %.2f ms = %.1f GB/s; %d host threads visible" % (dt * 1e3, nbytes / dt / 1e9, os.cpu_count()))

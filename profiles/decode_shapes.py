#!/usr/bin/env python3
"""Decode-to-s16 time per 65 536 frames by batch shape: `[AC3MI_LIB=...] python profiles/decode_shapes.py` (frames: the
engine's encoder on bench.py's content; few long streams take the frame-parallel front end)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

pkg = bench.importlib_pkg()
dev = torch.device("cuda:0")
eng = pkg.Engine(0)
enc = pkg.EncodeDesc(48000, 384000, 6)
fb = enc.frame_bytes()
N = 65536
g = torch.Generator(device=dev).manual_seed(99)
t = torch.arange(1536, device=dev, dtype=torch.float32)
ph = torch.rand((N, 1, 6), device=dev, generator=g) * 6.28
fr = 0.01 * torch.arange(1, 7, device=dev, dtype=torch.float32)
pcm = 8000.0 * torch.sin(ph + fr * t[None, :, None]) + (torch.rand((N, 1536, 6), device=dev, generator=g) - 0.5) * 4096
env = torch.where(torch.rand((N, 3, 1, 6), device=dev, generator=g) < 0.5, 1.0, 1.0 / 32)
pcm = (pcm.reshape(N, 3, 512, 6) * env).reshape(N, 1, 1536, 6).round().clamp(-32768, 32767).to(torch.int16).contiguous()
last = torch.zeros((N, 6, 256), dtype=torch.int16, device=dev)
csnr = torch.full((N,), 40, dtype=torch.int32, device=dev)
frames = torch.zeros((N, 1, fb), dtype=torch.uint8, device=dev)
eng.encode_batch(enc, pcm, (0, 2, 1, 4, 5, 3), last, csnr, out=frames)
torch.cuda.synchronize()
dec = pkg.DecodeDesc(flags=7 | 16 | 32, level=1.0, bias=384.0, dynrng=1, acmod=7, lfeon=1, frame_bytes=fb)
out = []
for S in (65536, 8192, 1024, 128, 16):
    F = N // S
    x = frames.reshape(S, F, fb)
    delay = torch.zeros((S, 6, 128), dtype=torch.float32, device=dev)
    lfsr = torch.ones((S,), dtype=torch.int16, device=dev)
    o16 = torch.empty((S, F, 6, 256, 6), dtype=torch.int16, device=dev)
    status = torch.zeros((S, F), dtype=torch.int32, device=dev)
    best = 1e9
    for it in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.decode_s16_batch(dec, x, delay, lfsr, out=o16, status=status)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    out.append("%d x %d: %.2f ms" % (S, F, best * 1e3))
print(os.environ.get("AC3MI_LIB", "default").rsplit("/", 1)[-1], " | ".join(out))

#!/usr/bin/env python3
"""Where decode_kernel<0> spends a frame's time: `AC3MI_LIB=.../libac3mi_stamps.so python profiles/decode_stamps.py [streams]`
(library built with `make EXTRA="-DDEC_STAMPS"`).  Content = bench.py's decode leg (frames made by the engine's encoder)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
pkg = bench.importlib_pkg()
dev = torch.device("cuda:0")
eng = pkg.Engine(0)
eng.set_decode_mode(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
enc = pkg.EncodeDesc(48000, 384000, 6)
fb = enc.frame_bytes()
g = torch.Generator(device=dev).manual_seed(99)
t = torch.arange(1536, device=dev, dtype=torch.float32)
ph = torch.rand((S, 1, 6), device=dev, generator=g) * 6.28
fr = 0.01 * torch.arange(1, 7, device=dev, dtype=torch.float32)
pcm = 8000.0 * torch.sin(ph + fr * t[None, :, None]) + (torch.rand((S, 1536, 6), device=dev, generator=g) - 0.5) * 4096
env = torch.where(torch.rand((S, 3, 1, 6), device=dev, generator=g) < 0.5, 1.0, 1.0 / 32)
pcm = (pcm.reshape(S, 3, 512, 6) * env).reshape(S, 1536, 6).round().clamp(-32768, 32767).to(torch.int16).reshape(S, 1, 1536, 6).contiguous()
last = torch.zeros((S, 6, 256), dtype=torch.int16, device=dev)
csnr = torch.full((S,), 40, dtype=torch.int32, device=dev)
frames = torch.zeros((S, 1, fb), dtype=torch.uint8, device=dev)
eng.encode_batch(enc, pcm, (0, 2, 1, 4, 5, 3), last, csnr, out=frames)
dec = pkg.DecodeDesc(flags=7 | 16 | 32, level=1.0, bias=384.0, dynrng=1, acmod=7, lfeon=1, frame_bytes=fb)
delay = torch.zeros((S, 6, 128), dtype=torch.float32, device=dev)
lfsr = torch.ones((S,), dtype=torch.int16, device=dev)
out16 = torch.empty((S, 1, 6, 256, 6), dtype=torch.int16, device=dev)
status = torch.zeros((S, 1), dtype=torch.int32, device=dev)
lib = eng.lib
lib.ac3mi_debug_dec_cycles.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
out = (ctypes.c_ulonglong * 8)()
torch.cuda.synchronize()
assert lib.ac3mi_debug_dec_cycles(out, 1) == 0
names = ("staging + header", "side information", "exponents", "bit allocation (+ its parameters)", "mantissas, coupling, stores", "rest")
for it in range(3):
    eng.decode_s16_batch(dec, frames, delay, lfsr, out=out16, status=status)
    torch.cuda.synchronize()
    assert lib.ac3mi_debug_dec_cycles(out, 1) == 0
    tot = sum(out[i] for i in range(6))
    print("pass %d: %d frames" % (it, S))
    for i, nm in enumerate(names):
        print("   %-36s %9.0f ticks/frame  %5.1f %%" % (nm, out[i] / S, 100.0 * out[i] / max(tot, 1)))

#!/usr/bin/env python3
"""Where enc_pack_kernel spends a frame's time: `AC3MI_LIB=.../libac3mi_stamps.so python profiles/pack_stamps.py [streams]`
(library built with `make EXTRA="-DPACK_STAMPS"`).  Content = bench.py's encode leg.  Prints s_memtime cycles per frame
and wavefront for the kernel's sections (100 MHz counter: x21 for shader cycles at 2.1 GHz)."""
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
enc = pkg.EncodeDesc(48000, 384000, 6)
g = torch.Generator(device=dev).manual_seed(99)
t = torch.arange(1536, device=dev, dtype=torch.float32)
ph = torch.rand((S, 1, 6), device=dev, generator=g) * 6.28
fr = 0.01 * torch.arange(1, 7, device=dev, dtype=torch.float32)
pcm = 8000.0 * torch.sin(ph + fr * t[None, :, None]) + (torch.rand((S, 1536, 6), device=dev, generator=g) - 0.5) * 4096
env = torch.where(torch.rand((S, 3, 1, 6), device=dev, generator=g) < 0.5, 1.0, 1.0 / 32)
pcm = (pcm.reshape(S, 3, 512, 6) * env).reshape(S, 1536, 6).round().clamp(-32768, 32767).to(torch.int16).reshape(S, 1, 1536, 6).contiguous()
last = torch.zeros((S, 6, 256), dtype=torch.int16, device=dev)
csnr = torch.full((S,), 40, dtype=torch.int32, device=dev)
frames = torch.zeros((S, 1, enc.frame_bytes()), dtype=torch.uint8, device=dev)
lib = eng.lib
lib.ac3mi_debug_pack_cycles.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
out = (ctypes.c_ulonglong * 16)()
for it in range(3):
    eng.encode_batch(enc, pcm, (0, 2, 1, 4, 5, 3), last, csnr, out=frames)
    torch.cuda.synchronize()
    assert lib.ac3mi_debug_pack_cycles(out, 1) == 0
    n = max(out[5], 1)
    names = ("search kernel: set-up + search", "packer: header, side info, exponents", "packer: mantissas (whole)", "packer: CRC + store", None, None, "packer: set-up", None,
             "  mantissas: bap addresses", "  mantissas: stage 1", "  mantissas: stage 2")
    tot = sum(out[i] for i in (0, 1, 2, 3, 6))
    print("pass %d: %d + %d frames, %.2f searches/frame" % (it, out[5], out[7], out[4] / n))
    for i, nm in enumerate(names):
        if nm:
            print("   %-40s %9.0f ticks/frame  %5.1f %%" % (nm, out[i] / n, 100.0 * out[i] / max(tot, 1)))
# second-generation content (the transcode's encoder half): the frames above decoded to s16, encoded from fresh state
dec = pkg.DecodeDesc(flags=7 | 16 | 32, level=1.0, bias=384.0, dynrng=1, acmod=7, lfeon=1, frame_bytes=enc.frame_bytes())
delay = torch.zeros((S, 6, 128), dtype=torch.float32, device=dev)
lfsr = torch.ones((S,), dtype=torch.int16, device=dev)
s16, _ = eng.decode_s16_batch(dec, frames, delay, lfsr)
eng.sync()
last.zero_(); csnr.fill_(40)
torch.cuda.synchronize()
assert lib.ac3mi_debug_pack_cycles(out, 1) == 0
eng.encode_batch(enc, s16.reshape(S, 1, 1536, 6), (0, 2, 1, 4, 5, 3), last, csnr, out=frames)
torch.cuda.synchronize()
assert lib.ac3mi_debug_pack_cycles(out, 1) == 0
n = max(out[5], 1)
tot = sum(out[i] for i in (0, 1, 2, 3, 6))
print("decoded content, cold: %d frame visits, %.2f searches/frame visit" % (out[5], out[4] / n))
for i, nm in enumerate(names):
    if nm:
        print("   %-40s %9.0f ticks/frame  %5.1f %%" % (nm, out[i] / n, 100.0 * out[i] / max(tot, 1)))

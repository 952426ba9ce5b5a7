#!/usr/bin/env python3
"""Cold / warm encoder passes on bench.py's encode content: `[AC3MI_LIB=...] [AC3MI_ENCODE_MODE=1|2] python profiles/encode_cold.py [S] [passes]`.
cold = encoder history 0 and csnroffst 40 before every pass (BASELINE configs[2]); warm = the passes run on."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 5
pkg = bench.importlib_pkg()
dev = torch.device("cuda:0")
eng = pkg.Engine(0)
MODE = int(os.environ.get("AC3MI_ENCODE_MODE", "0"))
eng.set_encode_mode(MODE)
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
csnr40 = csnr.clone()
frames = torch.zeros((S, 1, enc.frame_bytes()), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()


def go():
    eng.encode_batch(enc, pcm, (0, 2, 1, 4, 5, 3), last, csnr, out=frames, wait_torch=False)


go()
eng.sync()
cold = 0.0
for _ in range(passes):
    eng.memset(last)
    eng.copy(csnr, csnr40)
    eng.timer_start()
    go()
    cold += eng.timer_stop()
csum = int(frames.to(torch.int64).sum().item())
eng.timer_start()
for _ in range(passes):
    go()
warm = eng.timer_stop()
print("%s mode=%s: cold %.3f ms  warm %.3f ms per %d frames (byte sum of the last cold pass %d)" % (
    os.environ.get("AC3MI_LIB", "libac3mi.so").rsplit("/", 1)[-1], MODE, cold / passes, warm / passes, S, csum), flush=True)

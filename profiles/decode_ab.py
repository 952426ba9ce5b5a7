#!/usr/bin/env python3
"""A/B timing aid: ms per pass of ac3mi_decode_s16_batch over 65 536 bench-like one-frame streams for the decode modes
given on the command line (default 3 1), with the library AC3MI_LIB points at.   python profiles/decode_ab.py [S] [modes...]"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

pkg = importlib.import_module("ac-3-acm-codec_amd")
eng = pkg.Engine(0)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
modes = [int(m) for m in sys.argv[2:]] or [3, 1]
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(99)
t = torch.arange(1536, device=dev, dtype=torch.float32)
ph = torch.rand((S, 1, 6), device=dev, generator=g) * 6.28
fr = 0.01 * torch.arange(1, 7, device=dev, dtype=torch.float32)
pcm = 8000.0 * torch.sin(ph + fr * t[None, :, None]) + (torch.rand((S, 1536, 6), device=dev, generator=g) - 0.5) * 4096
env = torch.where(torch.rand((S, 3, 1, 6), device=dev, generator=g) < 0.5, 1.0, 1.0 / 32)
pcm = (pcm.reshape(S, 3, 512, 6) * env).reshape(S, 1536, 6)
pcm = pcm.round().clamp(-32768, 32767).to(torch.int16).reshape(S, 1, 1536, 6).contiguous()
enc = pkg.EncodeDesc(48000, 384000, 6)
last = torch.zeros((S, 6, 256), dtype=torch.int16, device=dev)
csnr = torch.full((S,), 40, dtype=torch.int32, device=dev)
frames = eng.encode_batch(enc, pcm, (0, 2, 1, 4, 5, 3), last, csnr)
eng.sync()
dec = pkg.DecodeDesc(flags=7 | 16 | 32, level=1.0, bias=384.0, dynrng=1, acmod=7, lfeon=1, frame_bytes=enc.frame_bytes())
delay = torch.zeros((S, 6, 128), dtype=torch.float32, device=dev)
lfsr = torch.ones((S,), dtype=torch.int16, device=dev)
out = torch.empty((S, 1, 6, 256, 6), dtype=torch.int16, device=dev)
status = torch.zeros((S, 1), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for mode in modes:
    eng.set_decode_mode(mode)
    for _ in range(2):
        eng.decode_s16_batch(dec, frames, delay, lfsr, out=out, status=status, wait_torch=False)
    eng.sync()
    eng.timer_start()
    for _ in range(5):
        eng.decode_s16_batch(dec, frames, delay, lfsr, out=out, status=status, wait_torch=False)
    ms = eng.timer_stop() / 5
    ok = int((status & 0x1ff).max().item()) == 0
    print("%s mode %d: %.3f ms per %d frames (%.2f M frames/s) ok=%s" % (os.path.basename(os.environ.get("AC3MI_LIB", "libac3mi.so")), mode, ms, S, S / ms / 1e3, ok), flush=True)

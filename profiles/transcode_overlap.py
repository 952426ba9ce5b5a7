#!/usr/bin/env python3
"""Does ac3mi_transcode_batch's chunk pipeline buy anything?  One process, two engine contexts (AC3MI_NO_OVERLAP is read when a
context is created): the same 65 536 cold one-frame transcodes with the pipeline (two chunks: front end of chunk 1 beside
the transform of chunk 0, encoder of chunk 0 beside ...) and without (kernels back to back on one stream), next to the sum of
the separate decode-to-s16 and encode calls.   python profiles/transcode_overlap.py [S] [passes]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 10
pkg = bench.importlib_pkg()
dev = torch.device("cuda:0")
enc = pkg.EncodeDesc(48000, 384000, 6)
fb = enc.frame_bytes()
g = torch.Generator(device=dev).manual_seed(99)
t = torch.arange(1536, device=dev, dtype=torch.float32)
ph = torch.rand((S, 1, 6), device=dev, generator=g) * 6.28
fr = 0.01 * torch.arange(1, 7, device=dev, dtype=torch.float32)
pcm = 8000.0 * torch.sin(ph + fr * t[None, :, None]) + (torch.rand((S, 1536, 6), device=dev, generator=g) - 0.5) * 4096
env = torch.where(torch.rand((S, 3, 1, 6), device=dev, generator=g) < 0.5, 1.0, 1.0 / 32)
pcm = (pcm.reshape(S, 3, 512, 6) * env).reshape(S, 1536, 6).round().clamp(-32768, 32767).to(torch.int16).reshape(S, 1, 1536, 6).contiguous()
chmap = (0, 2, 1, 4, 5, 3)
dec = pkg.DecodeDesc(flags=7 | 16 | 32, level=1.0, bias=384.0, dynrng=1, acmod=7, lfeon=1, frame_bytes=fb)
csnr40 = torch.full((S,), 40, dtype=torch.int32, device=dev)
lfsr1 = torch.ones((S,), dtype=torch.int16, device=dev)
res = {}
for label, no_overlap in (("pipeline", False), ("back to back", True), ("pipeline", False), ("back to back", True)):
    if no_overlap:
        os.environ["AC3MI_NO_OVERLAP"] = "1"
    else:
        os.environ.pop("AC3MI_NO_OVERLAP", None)
    eng = pkg.Engine(0)
    last = torch.zeros((S, 6, 256), dtype=torch.int16, device=dev)
    csnr = csnr40.clone()
    frames = eng.encode_batch(enc, pcm, chmap, last, csnr)
    eng.sync()
    delay = torch.zeros((S, 6, 128), dtype=torch.float32, device=dev)
    lfsr = lfsr1.clone()
    out = torch.zeros((S, 1, fb), dtype=torch.uint8, device=dev)
    status = torch.zeros((S, 1), dtype=torch.int32, device=dev)
    s16 = torch.empty((S, 1, 6, 256, 6), dtype=torch.int16, device=dev)
    torch.cuda.synchronize()

    def reset():
        eng.memset(last); eng.copy(csnr, csnr40); eng.memset(delay); eng.copy(lfsr, lfsr1)

    def timed(fn):
        reset(); fn(); eng.sync()
        ms = 0.0
        for _ in range(passes):
            reset()
            eng.timer_start()
            fn()
            ms += eng.timer_stop()
        return ms / passes

    tc = timed(lambda: eng.transcode_batch(dec, enc, frames, delay, lfsr, chmap, last, csnr, out=out, status=status, wait_torch=False))
    d16 = timed(lambda: eng.decode_s16_batch(dec, frames, delay, lfsr, out=s16, status=status, wait_torch=False))
    en = timed(lambda: eng.encode_batch(enc, s16.reshape(S, 1, 1536, 6), chmap, last, csnr, out=out, wait_torch=False))
    print("%-13s transcode %.3f ms | decode_s16 %.3f + encode (of the decoded samples) %.3f = %.3f ms   per %d cold one-frame streams" %
          (label, tc, d16, en, d16 + en, S), flush=True)
    eng.close()

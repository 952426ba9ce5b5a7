#!/usr/bin/env python3
"""Where a workgroup of decode_wg_kernel spends its time: runs one decode_s16 batch through the WG_STAMPS build
(make -C ac-3-acm-codec_amd/csrc OUT=.../libac3mi_stamps.so BUILD=.../build_stamps EXTRA=-DWG_STAMPS) and prints, per
wavefront and phase of every block, cycles of work and cycles waiting at the barrier (s_memtime).
   AC3MI_LIB=ac-3-acm-codec_amd/libac3mi_stamps.so AC3MI_WG_STAMPS=/tmp/stamps.txt python profiles/wg_stamps.py"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

path = os.environ.setdefault("AC3MI_WG_STAMPS", "/tmp/wg_stamps.txt")
if os.path.exists(path):
    os.remove(path)
pkg = importlib.import_module("ac-3-acm-codec_amd")
eng = pkg.Engine(0)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
kind = sys.argv[2] if len(sys.argv) > 2 else "steps"
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
t = torch.arange(1536, device=dev, dtype=torch.float32)
ph = torch.rand((S, 1, 6), device=dev, generator=g) * 6.28
fr = 0.01 * torch.arange(1, 7, device=dev, dtype=torch.float32)
pcm = 8000.0 * torch.sin(ph + fr * t[None, :, None]) + (torch.rand((S, 1536, 6), device=dev, generator=g) - 0.5) * 4096
if kind == "steps":
    env = torch.where(torch.rand((S, 3, 1, 6), device=dev, generator=g) < 0.5, 1.0, 1.0 / 32)
    pcm = (pcm.reshape(S, 3, 512, 6) * env).reshape(S, 1536, 6)
pcm = pcm.round().clamp(-32768, 32767).to(torch.int16).reshape(S, 1, 1536, 6).contiguous()
enc = pkg.EncodeDesc(48000, 384000, 6)
last = torch.zeros((S, 6, 256), dtype=torch.int16, device=dev)
csnr = torch.full((S,), 40, dtype=torch.int32, device=dev)
frames = eng.encode_batch(enc, pcm, (0, 2, 1, 4, 5, 3), last, csnr)
eng.sync()
dec = pkg.DecodeDesc(flags=7 | 16 | 32, level=1.0, bias=384.0, dynrng=1, acmod=7, lfeon=1, frame_bytes=enc.frame_bytes())
eng.set_decode_mode(3)
delay = torch.zeros((S, 6, 128), dtype=torch.float32, device=dev)
lfsr = torch.ones((S,), dtype=torch.int16, device=dev)
for _ in range(2):
    out, st = eng.decode_s16_batch(dec, frames, delay, lfsr)
    eng.sync()
rows = [ln.split(":")[1].split() for ln in open(path).read().strip().split("\n\n")[-1].splitlines()]
T = [[int(x) for x in r] for r in rows]
names = ["ch0", "ch1", "ch2", "ch3", "ch4", "lfe", "parse", "xform"]
print("stamps: 0 staged | 1 after barrier | 2 header+block0 parsed | per block b: 4+8b after B1, 5+8b phase-1 work done, "
      "6+8b after B2, 7+8b phase-2 work done, 8+8b after B3, 9+8b phase-3 work done | 52 after drain barrier | 53 drain done | 54 frame end")
print("frame: %d cycles (stamp 0 of the first wave -> stamp 54)" % (max(t[54] for t in T) - min(t[0] for t in T)))
print("prologue (stage -> B1 of block 0): %d" % (T[0][4] - min(t[0] for t in T)))
for b in range(6):
    o = 4 + 8 * b
    print("block %d: period %d" % (b, (T[0][o + 8] if b < 5 else T[0][52]) - T[0][o]))
    for w in range(8):
        t = T[w]
        print("   %-5s phase1 work %6d wait %6d | phase2 work %6d wait %6d | phase3 work %6d" %
              (names[w], t[o + 1] - t[o], t[o + 2] - t[o + 1], t[o + 3] - t[o + 2], t[o + 4] - t[o + 3], t[o + 5] - t[o + 4]))
print("drain: xform %d" % (T[7][53] - T[7][52]))
ps = T[6][56:63]
print("parser, first half of block 3: state loads + cplco copy %d | blksw/dith/dynrng %d | coupling+remat %d | strategies+bandwidths %d | exponent bookkeeping %d | publish %d" % tuple(ps[i + 1] - ps[i] for i in range(6)))


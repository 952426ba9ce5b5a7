#!/usr/bin/env python3
"""Where a wavefront of mantx_kernel (decode_mx.hip) spends its time, per audio block:
`AC3MI_LIB=.../libac3mi_mxstamps.so python profiles/mx_stamps.py [streams]` (library built with `make EXTRA="-DMX_STAMPS"`).
Content = bench.py's frames.  s_memtime ticks (100 MHz) per wavefront."""
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
eng.set_decode_mode(6)
C = bench.Content(pkg, eng, dev, S, 0)
delay = torch.zeros((S, 6, 128), dtype=torch.float32, device=dev)
lfsr = torch.ones((S,), dtype=torch.int16, device=dev)
out16 = torch.empty((S, 1, 6, 256, 6), dtype=torch.int16, device=dev)
status = torch.zeros((S, 1), dtype=torch.int32, device=dev)
lib = eng.lib
lib.ac3mi_debug_mx_cycles.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
out = (ctypes.c_ulonglong * 48)()
torch.cuda.synchronize()
assert lib.ac3mi_debug_mx_cycles(out, 1) == 0
names = ("staging + barrier", "mantissas", "transform", "wait for block b-1", "window + output")
for it in range(2):
    eng.decode_s16_batch(C.dec, C.frames, delay, lfsr, out=out16, status=status)
    torch.cuda.synchronize()
    assert lib.ac3mi_debug_mx_cycles(out, 1) == 0
    print("pass %d: ticks per wavefront (block 0 .. 5 | mean)" % it)
    for i, nm in enumerate(names):
        v = [out[b * 8 + i] / S for b in range(6)]
        print("   %-20s %s | %7.1f" % (nm, " ".join("%7.1f" % x for x in v), sum(v) / 6))
    tot = [sum(out[b * 8 + i] for i in range(5)) / S for b in range(6)]
    print("   %-20s %s | %7.1f" % ("total", " ".join("%7.1f" % x for x in tot), sum(tot) / 6))

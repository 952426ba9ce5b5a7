#!/usr/bin/env python3
"""Rounds of few one-frame streams through ac3mi_transcode_batch from fresh state (the small end of bench.py's per-stream
curve), for a kernel trace: `[AC3MI_LIB=...] python profiles/small_rounds.py [sizes...]`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

sizes = [int(a) for a in sys.argv[1:]] or [1, 64, 512]
pkg = bench.importlib_pkg()
dev = torch.device("cuda:0")
eng = pkg.Engine(0)
for S in sizes:
    C = bench.Content(pkg, eng, dev, S, 0)
    C.reset_transcode(); C.transcode(); eng.sync()
    best, tot = 1e9, 0.0
    for _ in range(20):
        C.reset_transcode()
        eng.timer_start()
        C.transcode()
        ms = eng.timer_stop()
        best = min(best, ms); tot += ms
    print("%s: %5d streams  %.4f ms per round (best %.4f)" % (os.path.basename(os.environ.get("AC3MI_LIB", "libac3mi.so")), S, tot / 20, best), flush=True)

#!/usr/bin/env python3
"""A/B of decode modes inside the bench's headline call: ms per ac3mi_transcode_batch over S one-frame 5.1 streams from fresh
state, and per ac3mi_decode_batch (float PCM), for the decode modes given (default 4 6).
`python profiles/transcode_ab.py [S] [modes...]`"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
modes = [int(m) for m in sys.argv[2:]] or [4, 6]
pkg = bench.importlib_pkg()
dev = torch.device("cuda:0")
eng = pkg.Engine(0)
C = bench.Content(pkg, eng, dev, S, 0)
delay = torch.zeros((S, 6, 128), dtype=torch.float32, device=dev)
lfsr = torch.ones((S,), dtype=torch.int16, device=dev)
outf = torch.empty((S, 1, 6, 6, 256), dtype=torch.float32, device=dev)
status = torch.zeros((S, 1), dtype=torch.int32, device=dev)
descf = pkg.DecodeDesc(flags=7 | 16 | 32, level=1.0, bias=0.0, dynrng=1, acmod=7, lfeon=1, frame_bytes=C.fb)
sums = {}
for rnd in range(2):
    for mode in modes:
        eng.set_decode_mode(mode)
        C.reset_transcode(); C.transcode(); eng.sync()
        tot = 0.0
        for _ in range(5):
            C.reset_transcode()
            eng.timer_start()
            C.transcode()
            tot += eng.timer_stop()
        sums[mode] = int(C.frames2.to(torch.int64).sum().item())
        eng.decode_batch(descf, C.frames, delay, lfsr, out=outf, status=status, wait_torch=False); eng.sync()
        eng.timer_start()
        for _ in range(5):
            eng.decode_batch(descf, C.frames, delay, lfsr, out=outf, status=status, wait_torch=False)
        dec = eng.timer_stop() / 5
        print("mode %d: transcode %.3f ms, decode to float %.3f ms per %d frames; workspace %.2f GB" % (mode, tot / 5, dec, S, eng.workspace_bytes() / 1e9), flush=True)
print("transcoded frames equal across modes:", len(set(sums.values())) == 1)

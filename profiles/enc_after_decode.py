#!/usr/bin/env python3
"""Does the encoder run slower inside a transcode than on its own?  Legs on bench.py's content, each `passes` times from fresh
state: A = encode the synthetic PCM, B = encode the s16 the decoder makes of A's frames (the transcode's encoder input, as a
call of its own), C = the transcode call.  Run under a kernel trace (run_pmc_groups.sh / rocprofv3 --kernel-trace) and read the
encoder kernels' durations per leg: `python profiles/enc_after_decode.py [S] [passes]`."""
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
C = bench.Content(pkg, eng, dev, S, 0)
out16 = torch.empty((S, 1, 6, 256, 6), dtype=torch.int16, device=dev)
delay16 = torch.zeros((S, 6, 128), dtype=torch.float32, device=dev)
lfsr16 = torch.ones((S,), dtype=torch.int16, device=dev)
status16 = torch.zeros((S, 1), dtype=torch.int32, device=dev)
eng.decode_s16_batch(C.dec, C.frames, delay16, lfsr16, out=out16, status=status16, wait_torch=False)
eng.sync()
pcm2 = out16.reshape(S, 1, 1536, 6)
frames_b = torch.zeros_like(C.frames)


def leg(name, fn, reset):
    fn(); eng.sync()
    tot = 0.0
    for _ in range(passes):
        reset()
        eng.timer_start()
        fn()
        tot += eng.timer_stop()
    print("%-34s %.3f ms per %d frames" % (name, tot / passes, S), flush=True)


def reset_enc():
    eng.memset(C.last)
    eng.copy(C.csnr, C.csnr40)


leg("A encode, synthetic PCM", lambda: eng.encode_batch(C.enc, C.pcm, C.chmap, C.last, C.csnr, out=frames_b, wait_torch=False), reset_enc)
leg("B encode, decoded s16", lambda: eng.encode_batch(C.enc, pcm2, C.chmap, C.last, C.csnr, out=frames_b, wait_torch=False), reset_enc)
leg("C transcode", C.transcode, C.reset_transcode)
same = bool((frames_b == C.frames2).all().item())
print("B's frames == C's frames:", same)

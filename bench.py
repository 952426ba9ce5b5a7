#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X AC-3 block-transform engine.

Metric (BASELINE.json): AC-3 5.1 @ 48 kHz frames/sec/GPU, decode + encode.
Workload (BASELINE configs[4]'s per-GPU step at the batch size of configs[1]/[2]): 65536 independent 5.1 / 48 kHz /
384 kbps one-frame streams per GPU, each decoded (a52_frame + 6 x a52_block to s16 PCM, what the ACM driver's decode
loop does) and re-encoded (AC3_encode_frame) in ONE ac3mi_transcode_batch call, every step from FRESH stream state
(decoder overlap 0 / dither seed 1, encoder history 0 / csnroffst 40: BASELINE's definition of configs[2] / [4]); frames
in and frames out resident in HBM.  One "step" = state reset + one such call; `value` = frames of all ranks / wall time.

  python bench.py --gpus N --steps K --warmup W
  N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
  (RANK / LOCAL_RANK / WORLD_SIZE in the environment), or started plainly, in which case this process only
  spawns N child ranks of itself (before touching the GPU), one per device, and relays rank 0's line.
  Every rank owns its own 65536 streams - independent streams shard with no data-path collective: weak scaling.

Prints ONE JSON line on rank 0 (see DESIGN.md §5 for every field).  The transform-only number that headed the line in
rounds 1-3 (BASELINE configs[1]: IMDCT-512 + overlap-add, the HBM-bound kernel) is `extra.transform_imdct512`.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "AC-3 5.1@48kHz frames/sec/GPU (decode+encode); achieved HBM GB/s vs peak"      # BASELINE.json, verbatim
FRAMES_PER_GPU = 65536
N_CH = 6
# algorithmic HBM bytes per transcoded frame, unfused definition: decode 1 536 in + 36 864 out, encode
# 18 432 in + 1 536 out (a fused transcoder's minimum would be 3 072)
TRANSCODE_BYTES_PER_FRAME = 38400 + 19968
# algorithmic HBM bytes per frame of this workload:
# 36 planes x 1 KiB coefficients in + 36 x 1 KiB PCM out + 6 ch x 128 floats overlap state r+w
BYTES_PER_FRAME = 36 * 1024 + 36 * 1024 + 2 * N_CH * 128 * 4
HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(seconds_target=12.0):
    """The reference's CPU path for the same workload, on the host cores of this box.

    kind "reference": liba52's own a52_imdct_512 (oracle/_ref/liba52_ref.so, compiled from the
    reference sources in the build container) driven by a C loop; kind "port": oracle/liborc.so
    when the reference build is absent.  Bounded sample, one worker thread per host core."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from tests import _harness as H

    # a 1-GPU box grants this job 16 host cores whatever the affinity mask says
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 16))
    frames_per_call = 512
    rng = np.random.default_rng(1)
    coef = (rng.standard_normal((frames_per_call, 6, N_CH, 256)) * 0.05).astype(np.float32)

    if H.have_ref():
        kind = "reference"
        L = H.ref()
        L.refglue_imdct512_batch.argtypes = [H.fp, H.fp, ctypes.c_long, ctypes.c_long, H.cf]
        L.refglue_imdct512_batch.restype = None

        def work(_):
            data = coef.copy()
            delay = np.zeros((N_CH, 256), np.float32)
            # plane order [frame][blk][ch]: plane k belongs to chain k % 6 -> block-sequential per channel
            L.refglue_imdct512_batch(H.P(data, H.fp), H.P(delay, H.fp), frames_per_call * 36, N_CH, 0.0)
            return frames_per_call
    else:
        kind = "port"
        H.orc()

        def work(_):
            H.orc_xform(coef.reshape(1, frames_per_call, 6, N_CH, 256), None, 7, 1, 7 | 16)
            return frames_per_call

    work(0)                                            # warm caches / page in
    t0 = time.perf_counter()
    work(0)
    per_call = time.perf_counter() - t0
    calls_per_thread = max(1, min(int(seconds_target / max(per_call, 1e-6)), 2000))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        done = sum(ex.map(work, range(cores * calls_per_thread)))
    dt = time.perf_counter() - t0
    return {
        "value": done / dt,
        "unit": "frames/s",
        "cores": cores,
        "kind": kind,
        "sample": "%d frames (36 x a52_imdct_512 each) on %d threads, %.1f s wall; 1-thread rate %.0f frames/s"
                  % (done, cores, dt, frames_per_call / per_call),
    }


def cpu_baseline_codec(frames, seconds_each=4.0):
    """CPU rates beside the secondary legs, same bitstreams, whole loops in C: full frame decode with the reference's
    own liba52 (oracle/_ref, kind "reference"; the oracle port if it is absent) and encode with the encoder oracle
    (kind "port": the reference's ac3enc does not build here).  One stream per worker thread, a few seconds each."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from tests import _harness as H
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 16))
    n, fb = frames.shape[0], frames.shape[2]
    stream = np.ascontiguousarray(frames[:, 0, :])                   # n frames played as one stream per worker
    buf = np.zeros(stream.size + 64, np.uint8)
    buf[:stream.size] = stream.reshape(-1)
    O = H.orc()
    O.orc_a52_decode_frames.argtypes = [H.u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, H.fp]
    O.orc_ac3enc_encode_frames.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, H.i16p, ctypes.c_int, H.u8p, H.u8p]
    if H.have_ref():
        kind = "reference"
        R = H.ref()
        R.refglue_decode_frames.argtypes = [H.u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, H.fp]

        def dec():
            assert R.refglue_decode_frames(H.P(buf, H.u8p), n, fb, 7 | 16, 1.0, 0.0, None) == 0
    else:
        kind = "port"

        def dec():
            assert O.orc_a52_decode_frames(H.P(buf, H.u8p), n, fb, 7 | 16, 1.0, 0.0, None) == 0
    nenc = 16
    pcm16 = np.ascontiguousarray(H.gen_pcm(nenc, 6, seed=3, kind="bursts"))
    cm = (ctypes.c_uint8 * 8)(*H.CHMAP6)

    def enc():
        assert O.orc_ac3enc_encode_frames(48000, 384000, 6, H.P(pcm16, H.i16p), nenc, cm, None) == 0

    def rate(fn, unit):
        fn()
        t0 = time.perf_counter()
        fn()
        per = time.perf_counter() - t0
        reps = max(1, min(int(seconds_each / max(per, 1e-6)), 500))

        def worker(_):
            for _ in range(reps):
                fn()
            return unit * reps
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            done = sum(ex.map(worker, range(cores)))
        return done / (time.perf_counter() - t0), unit / per

    d_all, d_one = rate(dec, n)
    e_all, e_one = rate(enc, nenc)
    return {"decode": {"value": d_all, "unit": "frames/s", "cores": cores, "kind": kind, "one_thread": d_one},
            "encode": {"value": e_all, "unit": "frames/s", "cores": cores, "kind": "port", "one_thread": e_one}}


def _newest_profile(suffix):
    pdir = os.path.join(ROOT, "profiles")
    names = sorted(fn for fn in (os.listdir(pdir) if os.path.isdir(pdir) else []) if fn.endswith(suffix))
    for fn in reversed(names):
        try:
            d = json.load(open(os.path.join(pdir, fn)))
            d["file"] = "profiles/" + fn
            return d
        except (OSError, ValueError):
            continue
    return None


def measured_traffic(frames, kernels=("xform_kernel<false, 4, false>",)):
    """HBM bytes per launch of the named kernels, summed, from the newest committed rocprofv3 PMC summary
    (profiles/*_hbm_traffic.json: FETCH_SIZE doubled per the guide's gfx950 correction + WRITE_SIZE, separate --pmc passes,
    recipe profiles/run_r04.sh) - only if it was taken at this batch size and holds every kernel; else None."""
    d = _newest_profile("_hbm_traffic.json")
    if not d or d.get("frames_per_launch") != frames:
        return None, None
    try:
        return sum(d["kernels"][k]["hbm_bytes_per_launch"] for k in kernels), d["file"]
    except KeyError:
        return None, d["file"]


def instruction_mix():
    """Per-frame instruction counts of the engine kernels from the newest committed rocprofv3 PMC summary
    (profiles/*_instruction_mix.json; recipe profiles/run_r04.sh).  None if absent."""
    return _newest_profile("_instruction_mix.json")


N_SIMD = 256 * 4           # MI355X: 256 CUs x 4 SIMDs (MI355X_MICROARCH.md, chip-level parameters)


def leg_rooflines(name, nbytes, fps, probe_ginst, mix, probe_salu=None):
    """The two ceilings of a secondary leg.  hbm: algorithmic bytes x frames/s against 8 TB/s.
    valu_issue: VALU instructions per frame (committed PMC summary, named in `source`) x frames/s against what all SIMDs
    issue, measured by ac3mi_probe_valu_rate in this very run."""
    out = {"hbm": {"bound": "hbm", "achieved": nbytes * fps / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": nbytes * fps / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_frame": nbytes}}
    leg = (mix or {}).get("legs", {}).get(name)
    if leg and probe_ginst:
        valu = sum(mix["kernels"][k]["valu_per_frame"] for k in leg)
        salu = sum(mix["kernels"][k].get("salu_per_frame", 0) for k in leg)
        peak = probe_ginst * N_SIMD                  # 10^9 instructions/s, whole chip
        out["valu_issue"] = {"bound": "valu_issue", "achieved": valu * fps / 1e9, "peak": peak, "unit": "Ginst/s",
                             "frac": valu * fps / 1e9 / peak, "valu_per_frame": valu, "salu_per_frame": salu,
                             "kernels": leg, "source": mix.get("file"),
                             "peak_source": "ac3mi_probe_valu_rate of this run x %d SIMDs" % N_SIMD}
        if probe_salu:
            speak = probe_salu * N_SIMD
            out["salu_issue"] = {"bound": "salu_issue", "achieved": salu * fps / 1e9, "peak": speak, "unit": "Ginst/s",
                                 "frac": salu * fps / 1e9 / speak, "salu_per_frame": salu, "source": mix.get("file"),
                                 "peak_source": "ac3mi_probe_salu_rate of this run x %d SIMDs" % N_SIMD}
            # one wavefront issues its vector and scalar instructions in order: the two shares add up
            out["issue"] = {"bound": "valu+salu issue", "frac": out["valu_issue"]["frac"] + out["salu_issue"]["frac"],
                            "note": "sum of the two fractions: share of the issue slots the leg's instruction counts need at its measured rate"}
    else:
        out["valu_issue"] = None
    return out


_CRC_TAB = None


def ac3_crc_ok(frames):
    """Both CRCs of every AC-3 frame (numpy, host): crc1 covers the first 5/8 of the frame, crc2 the whole frame
    (ENC/ac3enc.cpp:1599-1638); a frame is intact when the CRC-16 (poly 0x8005) of bytes [2, 5/8) and of bytes [2, end)
    are both zero.  frames: [n][frame_bytes] uint8.  Returns the number of frames failing either check."""
    import numpy as np
    global _CRC_TAB
    if _CRC_TAB is None:
        t = np.zeros(256, np.uint32)
        for n in range(256):
            c = n << 8
            for _ in range(8):
                c = ((c << 1) ^ 0x8005) & 0xffff if c & 0x8000 else (c << 1) & 0xffff
            t[n] = c
        _CRC_TAB = t
    n, fb = frames.shape
    words = fb // 2
    fs58 = ((words >> 1) + (words >> 3)) * 2          # bytes covered by crc1 (incl. the sync word, which is skipped)
    crc = np.zeros(n, np.uint32)
    bad1 = None
    cols = np.ascontiguousarray(frames.T)              # [fb][n]: one contiguous row per byte position
    for i in range(2, fb):
        crc = (_CRC_TAB[(cols[i] ^ (crc >> 8)) & 0xff] ^ (crc << 8)) & 0xffff
        if i == fs58 - 1:
            bad1 = crc != 0
    return int(np.count_nonzero(bad1 | (crc != 0)))


class Content:
    """The synthetic batch every leg works on: seeded 5.1 PCM with level steps, its AC-3 frames (encoded once by the engine,
    untimed, from fresh state), the two descriptors and the state arrays of the transcode legs."""

    def __init__(self, pkg, eng, dev, S, rank):
        import torch
        self.S = S
        self.enc = pkg.EncodeDesc(48000, 384000, 6)
        self.fb = self.enc.frame_bytes()
        self.dec = pkg.DecodeDesc(flags=7 | 16 | 32, level=1.0, bias=384.0, dynrng=1, acmod=7, lfeon=1, frame_bytes=self.fb)
        self.chmap = (0, 2, 1, 4, 5, 3)
        g = torch.Generator(device=dev).manual_seed(99 + rank)
        self.gen = g
        t = torch.arange(1536, device=dev, dtype=torch.float32)
        ph = torch.rand((S, 1, 6), device=dev, generator=g) * 6.28
        fr = 0.01 * torch.arange(1, 7, device=dev, dtype=torch.float32)
        pcm = 8000.0 * torch.sin(ph + fr * t[None, :, None]) + (torch.rand((S, 1536, 6), device=dev, generator=g) - 0.5) * 4096
        # level steps (x1 / x1/32 per 512-sample segment and channel) so that frames carry a realistic mix of new and
        # reused exponent sets: a stationary signal would reuse block 0's exponents five times in every channel
        env = torch.where(torch.rand((S, 3, 1, 6), device=dev, generator=g) < 0.5, 1.0, 1.0 / 32)
        pcm = (pcm.reshape(S, 3, 512, 6) * env).reshape(S, 1536, 6)
        self.pcm = pcm.round().clamp(-32768, 32767).to(torch.int16).reshape(S, 1, 1536, 6).contiguous()
        self.last = torch.zeros((S, 6, 256), dtype=torch.int16, device=dev)
        self.csnr = torch.full((S,), 40, dtype=torch.int32, device=dev)
        self.csnr40 = torch.full((S,), 40, dtype=torch.int32, device=dev)
        self.lfsr1 = torch.ones((S,), dtype=torch.int16, device=dev)
        self.frames = torch.zeros((S, 1, self.fb), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)
        eng.encode_batch(self.enc, self.pcm, self.chmap, self.last, self.csnr, out=self.frames, wait_torch=False)
        eng.sync()
        # the transcode legs' own state and output
        self.delay2 = torch.zeros((S, 6, 128), dtype=torch.float32, device=dev)
        self.lfsr2 = torch.ones((S,), dtype=torch.int16, device=dev)
        self.last2 = torch.zeros((S, 6, 256), dtype=torch.int16, device=dev)
        self.csnr2 = torch.full((S,), 40, dtype=torch.int32, device=dev)
        self.frames2 = torch.zeros((S, 1, self.fb), dtype=torch.uint8, device=dev)
        self.status_tc = torch.zeros((S, 1), dtype=torch.int32, device=dev)
        self.eng = eng

    def reset_transcode(self):
        """fresh stream state, on the engine's stream"""
        e = self.eng
        e.memset(self.last2)
        e.copy(self.csnr2, self.csnr40)
        e.memset(self.delay2)
        e.copy(self.lfsr2, self.lfsr1)

    def transcode(self):
        self.eng.transcode_batch(self.dec, self.enc, self.frames, self.delay2, self.lfsr2, self.chmap, self.last2, self.csnr2,
                                 out=self.frames2, status=self.status_tc, wait_torch=False)


def secondary_timings(pkg, eng, dev, C, rank, dist, barrier, steps=20, checks=True, warm=True):
    """The other BASELINE configs on the same batch (frames resident in HBM): configs[2] encode (s16 PCM -> frames),
    bitstream decode (frames -> float PCM / s16), configs[3] (mixed block sizes + downmix), and the transcode with its
    stream state carried on ("warm").  Each: frames/s of this rank's shard (MAX-reduced time), algorithmic GB/s per
    SURVEY.md 8d, x realtime, and its ceilings (leg_rooflines)."""
    import torch
    S = C.S
    enc, dec, fb, chmap, pcm, frames, last, csnr = C.enc, C.dec, C.fb, C.chmap, C.pcm, C.frames, C.last, C.csnr
    g = C.gen
    delay = torch.zeros((S, 6, 128), dtype=torch.float32, device=dev)
    lfsr = torch.ones((S,), dtype=torch.int16, device=dev)
    out = torch.empty((S, 1, 6, 6, 256), dtype=torch.float32, device=dev)
    # one status array per leg: each leg's frame verdicts are read after all legs have run
    status = torch.zeros((S, 1), dtype=torch.int32, device=dev)
    status16 = torch.zeros((S, 1), dtype=torch.int32, device=dev)
    frames_e = torch.zeros((S, 1, fb), dtype=torch.uint8, device=dev)

    def do_enc():
        eng.encode_batch(enc, pcm, chmap, last, csnr, out=frames_e, wait_torch=False)

    def do_dec():
        eng.decode_batch(dec, frames, delay, lfsr, out=out, status=status, wait_torch=False)

    out16 = torch.empty((S, 1, 6, 256, 6), dtype=torch.int16, device=dev)
    delay16 = torch.zeros((S, 6, 128), dtype=torch.float32, device=dev)
    lfsr16 = torch.ones((S,), dtype=torch.int16, device=dev)

    def do_dec16():
        eng.decode_s16_batch(dec, frames, delay16, lfsr16, out=out16, status=status16, wait_torch=False)

    # BASELINE configs[3]: mixed short/long blocks (a quarter of the channel-blocks switched) with the 5.1 -> 2.0
    # downmix folded into the transform: 5 of the 6 planes in (liba52 drops the LFE), 2 planes out, 2 overlap tails
    mixdesc = pkg.XformDesc(7, 1, 2, 0.0)
    coef_mix = torch.randn((S, 1, 6, 6, 256), device=dev, generator=g) * 0.05
    blksw_mix = (torch.rand((S, 1, 6, 5), device=dev, generator=g) < 0.25).to(torch.uint8)
    delay_mix = torch.zeros((S, 2, 128), dtype=torch.float32, device=dev)
    out_mix = torch.empty((S, 1, 6, 2, 256), dtype=torch.float32, device=dev)

    def do_mix():
        eng.imdct_batch(mixdesc, coef_mix, delay_mix, blksw=blksw_mix, out=out_mix, wait_torch=False)

    # BASELINE configs[2] is defined on FRESH encoder state: every frame an independent stream with last_samples = 0 and
    # csnroffst = 40 (ENC/ac3enc.cpp:921, 969, 1092 - the SNR-offset search starts from the stream's previous result, so a
    # stream that re-encodes the same content starts at its own optimum from pass 2 on).  "cold" = state put back to those
    # values on the engine's stream before every timed pass, outside the timer (the leg's figure); "warm" = the passes run
    # on, each continuing the streams of the one before (the search's best case).
    def reset_enc():
        eng.memset(last)
        eng.copy(csnr, C.csnr40)

    def timed(fn, reset=None):
        torch.cuda.synchronize(dev)
        if reset:
            reset()
        fn()
        barrier()
        if reset is None:
            eng.timer_start()
            for _ in range(steps):
                fn()
            ms = eng.timer_stop() / steps
        else:
            ms = 0.0
            for _ in range(steps):
                reset()
                eng.timer_start()
                fn()
                ms += eng.timer_stop()
            ms /= steps
        barrier()
        if dist is not None:
            tt = torch.tensor([ms], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            ms = float(tt[0])
        return ms

    res = {}
    for name, fn, nbytes, reset in (("transform_downmix_mixed_blocks", do_mix, 30720 + 12288 + 2 * 1024, None),   # the LFE plane is not mixed in
                                    ("encode", do_enc, 18432 + 1536 + 2 * 3072, reset_enc),
                                    ("decode", do_dec, 1536 + 36864 + 2 * 3072, None),
                                    ("decode_s16", do_dec16, 1536 + 18432 + 2 * 3072, None)):
        ms = timed(fn, reset)
        fps = S / (ms * 1e-3)
        res[name] = {"frames_per_s_per_gpu": fps, "ms_per_pass": ms, "algorithmic_GBps": nbytes * fps / 1e9,
                     "hbm_frac": nbytes * fps / 1e9 / HBM_PEAK_GBS, "realtime_x": fps * 0.032,
                     "algorithmic_bytes_per_frame": nbytes}
        if reset is not None:
            res[name]["state"] = "cold: encoder history 0 and csnroffst 40 before every timed pass (BASELINE's definition)"
            if warm:
                wms = timed(fn, None)
                res[name]["warm"] = {"frames_per_s_per_gpu": S / (wms * 1e-3), "ms_per_pass": wms,
                                     "state": "warm: every pass continues the streams of the pass before (same content: the search starts at its optimum)"}
    # the same decode with the mantissa kernel and the transform as two kernels (ac3mi_set_decode_mode 4; auto fuses them for
    # one-frame streams, decode_mx.hip): the standing A/B of the fusion, and what keeps both kernels in the profiles
    eng.set_decode_mode(4)
    try:
        ms2 = timed(do_dec16, None)
    finally:
        eng.set_decode_mode(int(os.environ.get("AC3MI_DECODE_MODE", "0")))
    res["decode_s16"]["two_kernels"] = {"ms_per_pass": ms2, "frames_per_s_per_gpu": S / (ms2 * 1e-3),
                                        "what": "ac3mi_set_decode_mode 4: coefficient planes through HBM between mant_kernel and xform_kernel (36.9 KB written and read per frame)"}
    if warm:
        wms = timed(C.transcode, None)
        res["transcode_warm"] = {"frames_per_s_per_gpu": S / (wms * 1e-3), "ms_per_pass": wms,
                                 "state": "the headline's call with the stream state carried on from pass to pass (the search's best case)"}
    # ---- untimed epilogue: every leg's verdict on the whole batch, from its own buffers ----
    eng.sync()
    torch.cuda.synchronize(dev)
    res["decode"]["all_frames_ok"] = int((status & 0x1ff).max().item()) == 0
    res["decode_s16"]["all_frames_ok"] = int((status16 & 0x1ff).max().item()) == 0
    enc_host = frames_e.cpu().numpy().reshape(S, -1)[:, :fb]
    res["encode"]["frames_failing_crc"] = ac3_crc_ok(enc_host)
    res["encode"]["all_frames_ok"] = res["encode"]["frames_failing_crc"] == 0
    if rank == 0 and checks:
        res["encode"]["bit_exact_vs_oracle"] = check_against_oracle(pkg, eng, dev, enc, dec, chmap, pcm)
    res["note"] = ("secondary timings on %d frames/GPU (5.1, 48 kHz, 384 kbps); encode and the decode front end are "
                   "instruction-bound, not HBM-bound: hbm_frac is reported for completeness" % S)
    return res


def per_stream_curve(eng, C, sizes=(1, 64, 512, 2048, 4096, 8192, 65536), passes=8):
    """north_star: ">= 50x realtime per stream".  One frame is 32 ms of audio, so a stream runs at 32 ms / (time of the round
    that advances it by a frame): rounds of S one-frame streams through ac3mi_transcode_batch, each round from fresh stream
    state (reset outside the timer), HIP events on the engine's stream.  Small rounds take the fused decoder and the block
    packer (ac3mi.h), large ones the split kernels - whatever the engine picks for the batch."""
    rows = []
    for S in sizes:
        if S > C.S:
            continue
        fr, st = C.frames[:S], C.status_tc[:S]
        d, l, h, c, o = C.delay2[:S], C.lfsr2[:S], C.last2[:S], C.csnr2[:S], C.frames2[:S]

        def go():
            eng.transcode_batch(C.dec, C.enc, fr, d, l, C.chmap, h, c, out=o, status=st, wait_torch=False)

        def reset():
            eng.memset(h)
            eng.copy(c, C.csnr40[:S])
            eng.memset(d)
            eng.copy(l, C.lfsr1[:S])
        reset()
        go()
        eng.sync()
        best, tot = 1e9, 0.0
        for _ in range(passes):
            reset()
            eng.timer_start()
            go()
            ms = eng.timer_stop()
            best, tot = min(best, ms), tot + ms
        ms = tot / passes
        rows.append({"streams": S, "ms_per_round": ms, "best_ms": best, "realtime_x_per_stream": 32.0 / ms,
                     "aggregate_frames_per_s": S / (ms * 1e-3)})
    ok = [r for r in rows if r["realtime_x_per_stream"] >= 50.0]
    best = max(ok, key=lambda r: r["streams"]) if ok else None
    return {"rounds": rows,
            "largest_round_at_50x_realtime_per_stream": best,
            "note": "decode + encode of one frame per stream and round (ac3mi_transcode_batch, fresh state, frames resident in HBM); "
                    "a stream's realtime factor = 32 ms / round time; rounds measured: the operating point lies between the last "
                    "round at >= 50x and the next one"}


def dropin_single_stream(frames_host):
    """What ONE stream gets through the drop-in loop the ACM driver runs (src/AC3ACM.cpp:1498-1581, 1762): a plain-C host
    (tools/ac3mi_loop.c, built by tools/Makefile against include/ac3mi_dropin.h) calls a52_syncinfo / a52_frame / 6 x a52_block
    + the MapTab converter and AC3_encode_frame frame by frame - one launch sequence per frame, PCIe both ways."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "tools", "ac3mi_loop")
    if not os.path.exists(exe):
        return {"skipped": "tools/ac3mi_loop not built"}
    with tempfile.NamedTemporaryFile(suffix=".ac3", delete=False) as f:
        f.write(frames_host.tobytes())
        name = f.name
    try:
        p = subprocess.run([exe, name, "3"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        if p.returncode != 0:
            return {"skipped": "ac3mi_loop rc %d: %s" % (p.returncode, p.stderr.decode()[-200:])}
        return json.loads(p.stdout.decode().strip().splitlines()[-1])
    except (subprocess.TimeoutExpired, ValueError) as e:
        return {"skipped": str(e)[:200]}
    finally:
        os.unlink(name)


def check_against_oracle(pkg, eng, dev, enc, dec, chmap, pcm, n_enc=4096, n_dec=256):
    """BASELINE configs[2]'s "bit-exact check" at bench size, untimed.  The timed legs carry stream state from pass to
    pass, so this runs ONE fresh-state pass over the first n_enc one-frame streams of the batch (encoder history 0,
    csnroffst 40, as AC3_encode_init leaves a stream) and compares every frame byte for byte with the CPU oracle's
    encoding of the same PCM; the first n_dec of those frames are then decoded to s16 by the engine (fresh decoder state)
    and by the oracle (<= 1 step apart: the float PCM may differ by one ulp at bias 384).  The oracle is the checker only."""
    import ctypes
    import numpy as np
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from tests import _harness as H
    O = H.orc()
    O.orc_ac3enc_encode_frames.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, H.i16p, ctypes.c_int, H.u8p, H.u8p]
    S = pcm.shape[0]
    n_enc, n_dec = min(n_enc, S), min(n_dec, n_enc, S)
    fb = enc.frame_bytes()
    sub = pcm[:n_enc].contiguous()
    last = torch.zeros((n_enc, 6, 256), dtype=torch.int16, device=dev)
    csnr = torch.full((n_enc,), 40, dtype=torch.int32, device=dev)
    got = eng.encode_batch(enc, sub, chmap, last, csnr)
    eng.sync()
    got_host = got.cpu().numpy().reshape(n_enc, -1)[:, :fb]
    src = np.ascontiguousarray(sub.cpu().numpy().reshape(n_enc, 1536 * 6))
    cm = (ctypes.c_uint8 * 8)(*H.CHMAP6)
    want = np.zeros((n_enc, fb), np.uint8)

    def one(i):
        return O.orc_ac3enc_encode_frames(48000, 384000, 6, H.P(src[i], H.i16p), 1, cm, H.P(want[i], H.u8p))
    with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as ex:
        rcs = list(ex.map(one, range(n_enc)))
    enc_mismatch = int(np.count_nonzero((want != got_host).any(axis=1))) + sum(1 for r in rcs if r)
    delay = torch.zeros((n_dec, 6, 128), dtype=torch.float32, device=dev)
    lfsr = torch.ones((n_dec,), dtype=torch.int16, device=dev)
    out16, st = eng.decode_s16_batch(dec, got[:n_dec].contiguous(), delay, lfsr)
    eng.sync()
    got16 = out16.cpu().numpy().reshape(n_dec, 6, 256, 6)
    dec_ok = int((st & 0x1ff).max().item()) == 0
    worst = 0
    ref16 = np.zeros((256, 6), np.int16)
    for i in range(n_dec):
        pcmf, errs, oflags = H.orc_decode(got_host[i:i + 1], 7 | 16 | 32, 1.0, 384.0)
        dec_ok = dec_ok and errs == 0
        for b in range(6):
            O.orc_convert_s16(H.P(np.ascontiguousarray(pcmf[0, b]), H.fp), H.P(ref16, H.i16p), oflags)
            worst = max(worst, int(np.abs(got16[i, b].astype(np.int32) - ref16.astype(np.int32)).max()))
    return {"encode_frames_checked": n_enc, "encode_frames_differing": enc_mismatch,
            "decode_s16_frames_checked": n_dec, "decode_s16_max_abs_step": worst,
            "ok": enc_mismatch == 0 and worst <= 1 and dec_ok,
            "note": "fresh-state pass over the first streams of the batch vs oracle/liborc.so (untimed epilogue)"}


def million_stream_transcode(pkg, eng, dev, frames, n_streams=1 << 20, passes=2):
    """BASELINE configs[4], one GPU's share (8M streams / 8 GPUs): decode -> s16 -> re-encode of 2^20 independent streams,
    one frame each per pass, every pass from fresh stream state (configs[4]'s definition; state reset outside the timer).
    AC3MI_BENCH_MILLION=0 skips it (takes ~1 s of GPU time and ~30 GB of HBM: frames in/out 3 GB, carry-over state 6.5 GB,
    the engine's tile-bounded workspaces ~20 GB: `engine_workspace_GB` is what the context really holds, ac3mi_workspace_bytes)."""
    import torch
    S0, _, fb = frames.shape
    big = frames.repeat((n_streams + S0 - 1) // S0, 1, 1)[:n_streams].contiguous()
    enc = pkg.EncodeDesc(48000, 384000, 6)
    dec = pkg.DecodeDesc(flags=7 | 16 | 32, level=1.0, bias=384.0, dynrng=1, acmod=7, lfeon=1, frame_bytes=fb)
    delay = torch.zeros((n_streams, 6, 128), dtype=torch.float32, device=dev)
    lfsr = torch.ones((n_streams,), dtype=torch.int16, device=dev)
    last = torch.zeros((n_streams, 6, 256), dtype=torch.int16, device=dev)
    csnr = torch.full((n_streams,), 40, dtype=torch.int32, device=dev)
    out = torch.zeros((n_streams, 1, fb), dtype=torch.uint8, device=dev)
    status = torch.zeros((n_streams, 1), dtype=torch.int32, device=dev)
    chmap = (0, 2, 1, 4, 5, 3)
    csnr40 = torch.full((n_streams,), 40, dtype=torch.int32, device=dev)
    lfsr1 = torch.ones((n_streams,), dtype=torch.int16, device=dev)
    _, total = torch.cuda.mem_get_info(dev)
    torch.cuda.synchronize(dev)
    eng.transcode_batch(dec, enc, big, delay, lfsr, chmap, last, csnr, out=out, status=status, wait_torch=False)
    ms = 0.0
    for _ in range(passes):
        eng.memset(last)
        eng.memset(delay)
        eng.copy(csnr, csnr40)
        eng.copy(lfsr, lfsr1)
        eng.timer_start()
        eng.transcode_batch(dec, enc, big, delay, lfsr, chmap, last, csnr, out=out, status=status, wait_torch=False)
        ms += eng.timer_stop()
    ms /= passes
    ok = int((status & 0x1ff).max().item()) == 0
    same = bool(torch.equal(out[:S0], out[S0:2 * S0])) if n_streams >= 2 * S0 else None
    plan = pkg.sharding.plan_transcode_bytes(n_streams)
    return {"streams": n_streams, "ms_per_pass": ms, "frames_per_s_per_gpu": n_streams / (ms * 1e-3), "all_frames_ok": ok, "state": "cold (fresh stream state every pass)",
            "replicas_agree_across_tiles": same, "engine_workspace_GB": eng.workspace_bytes() / 1e9,
            "planned_GB": {"workspace": plan["workspace"] / 1e9, "state_and_io": plan["state_and_io"] / 1e9}, "hbm_total_GB": total / 1e9}


def stream_layer_timing(pkg, eng, frames, rounds=20):
    """PCIe-inclusive rate of the byte-stream layer (include/ac3mi_stream.h): n live AC-3 streams, host buffers in,
    s16 host buffers out, one ac3mi_stream_convert_many call per frame time (one batched launch per round)."""
    import ctypes, importlib, time
    import numpy as np
    S = importlib.import_module(pkg.__name__ + ".stream")
    n, fb = frames.shape[0], frames.shape[2]
    pool = S.Pool(eng, n)
    src_fmt, dst_fmt = S.ac3_format(6, 48000, 384, block_align=fb), S.pcm_format(6, 48000)
    streams = [pool.open(src_fmt, dst_fmt)[1] for _ in range(n)]
    src = np.ascontiguousarray(frames.reshape(n, fb))
    dst = np.zeros((n, 6 * 256 * 6 * 2), np.uint8)
    hs = [S.StreamHeader(src[i].ctypes.data, fb, 0, dst[i].ctypes.data, dst.shape[1], 0, S.STREAMCONVERTF_START) for i in range(n)]
    sarr = (ctypes.c_void_p * n)(*[s.handle for s in streams])
    harr = (ctypes.POINTER(S.StreamHeader) * n)(*[ctypes.pointer(h) for h in hs])
    lib = pool.lib
    assert lib.ac3mi_stream_convert_many(sarr, harr, n) == 0        # warm-up (first touch of the pinned staging)
    for h in hs:
        h.flags = 0
    for _ in range(3):                                              # worker threads, page tables and clocks settle
        assert lib.ac3mi_stream_convert_many(sarr, harr, n) == 0
    t0 = time.perf_counter()
    for _ in range(rounds):
        assert lib.ac3mi_stream_convert_many(sarr, harr, n) == 0
    dt = (time.perf_counter() - t0) / rounds
    used = all(h.src_used == fb and h.dst_used == dst.shape[1] for h in hs)
    for s in streams:
        s.close()
    pool.close()
    return {"streams": n, "frames_per_s": n / dt, "ms_per_round": dt * 1e3, "all_bytes_used": bool(used),
            "note": "host AC-3 bytes -> host s16 PCM through ac3mi_stream_convert_many: buffering state machines, "
                    "H2D, decode + transform + s16 kernels, D2H, all inside the timed call (PCIe-inclusive)"}


def launch_ranks(args):
    """`python bench.py --gpus N` started without a launcher: spawn N ranks of this script, one per device, and relay
    rank 0's line.  Runs BEFORE anything touches the GPU in this process (the children are fresh processes, never an exec
    from a process that initialised HIP); the parent only waits and collects."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def launch_check(job_streams=0):
    """--launch-check: the rendezvous, barrier and MAX / SUM reductions of the N-rank path over gloo, no GPU and no
    engine (tests/test_bench_launcher.py runs it on CPU with world size 2, and with world size 8 on BASELINE configs[4]'s
    8M streams: every rank's shard and the HBM it plans for it)."""
    import torch
    import torch.distributed as dist
    sh = importlib_pkg().sharding
    rank, local_rank, world = sh.env_ranks()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
        dist.barrier()
    lo, hi = sh.shard(job_streams or FRAMES_PER_GPU * world, world, rank)
    plan = sh.plan_transcode_bytes(hi - lo)
    tmax, pmax = sh.reduce_max([0.001 * (rank + 1), float(plan["total"])], dist if world > 1 else None)
    total, fits = sh.reduce_sum([hi - lo, 1.0 if plan["fits"] else 0.0], dist if world > 1 else None)
    per_rank = [float(hi - lo)]
    if world > 1:                   # what main() does with every rank's own rate: gathered, not only MAX-reduced
        t = torch.tensor([float(hi - lo)], dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank = [float(x[0]) for x in allt]
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "streams_total": int(total), "max_time": tmax,
                          "local_rank": local_rank, "max_rank_hbm_plan_bytes": pmax, "ranks_that_fit_288GB": int(fits),
                          "per_rank_streams": per_rank}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def importlib_pkg():
    import importlib
    return importlib.import_module("ac-3-acm-codec_amd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=FRAMES_PER_GPU, help="frames (independent streams) per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary timings (transform, encode, decode, curves)")
    ap.add_argument("--no-checks", action="store_true", help="skip the untimed oracle check and the host-side legs (profiling runs)")
    ap.add_argument("--no-warm", action="store_true", help="skip the warm (state carried on) passes of the encode / transcode legs (PMC runs)")
    ap.add_argument("--launch-check", action="store_true", help="N-rank rendezvous and reductions only (gloo, no GPU)")
    ap.add_argument("--job-streams", type=int, default=0, help="--launch-check: streams of the whole job (default 65536 per rank)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    if args.launch_check:
        launch_check(args.job_streams)
        return

    import importlib
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the multi-rank code path on a one-GPU box: AC3MI_BENCH_REHEARSE=1 puts every rank on device 0 and
    # uses gloo (RCCL refuses two ranks on one device); never set by the driver
    rehearse = os.environ.get("AC3MI_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rdev = None if rehearse else dev                  # where the reductions' tensors live

    pkg = importlib.import_module("ac-3-acm-codec_amd")
    eng = pkg.Engine(local_rank)                      # fails loudly without libac3mi.so / a GPU

    S = args.frames
    # this rank's contiguous shard of the job's S x world independent streams (no data-path collective)
    lo, hi = pkg.sharding.shard(S * world, world, rank)
    assert hi - lo == S
    C = Content(pkg, eng, dev, S, rank)               # seeded PCM -> this rank's AC-3 frames (untimed)

    def step():
        C.reset_transcode()                           # fresh streams: part of the step
        C.transcode()

    def barrier():
        torch.cuda.synchronize(dev)
        eng.sync()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt_own = time.perf_counter() - t0
    dt = dt_own
    per_rank = [S * args.steps / dt_own]
    if dist is not None:
        t = torch.tensor([dt_own], device=rdev, dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank = [S * args.steps / float(x[0]) for x in allt]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    # the call alone (no state reset), HIP events on the engine's own stream, one pair per call: what the rooflines divide by
    n_ev = max(1, min(args.steps, 20))
    call_ms = 0.0
    for _ in range(n_ev):
        C.reset_transcode()
        eng.timer_start()
        C.transcode()
        call_ms += eng.timer_stop()
    call_ms /= n_ev
    if dist is not None:
        t = torch.tensor([call_ms], device=rdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        call_ms = float(t[0])
    eng.sync()
    torch.cuda.synchronize(dev)
    tc_ok_decode = int((C.status_tc & 0x1ff).max().item()) == 0
    tc_host = C.frames2.cpu().numpy().reshape(S, -1)[:, :C.fb]
    tc_bad_crc = ac3_crc_ok(tc_host)

    # ---- the ceilings this run is priced against: probes of this very run + the committed PMC summaries ----
    rate = eng.probe_valu_rate()
    srate = eng.probe_salu_rate()
    mix = instruction_mix()
    fps_call = S / (call_ms * 1e-3)
    rl = leg_rooflines("transcode", TRANSCODE_BYTES_PER_FRAME, fps_call, rate, mix, srate)
    tc_kernels = (mix or {}).get("legs", {}).get("transcode", [])
    traffic, traffic_file = measured_traffic(S, tc_kernels) if tc_kernels else (None, None)
    hbm = dict(rl["hbm"])
    hbm["traffic"] = traffic
    hbm["traffic_source"] = ("sum over the call's kernels of 2 x FETCH_SIZE + WRITE_SIZE per launch, committed rocprofv3 --pmc passes at this batch "
                             "size (%s), not collected in this run" % traffic_file)
    hbm["fused_minimum_bytes_per_frame"] = 3072
    issue = rl.get("issue")
    roofline = {
        "bound": "issue",
        "achieved": rl["valu_issue"]["achieved"] if issue else None,
        "peak": rl["valu_issue"]["peak"] if issue else None,
        "unit": "Ginst/s",
        "frac": rl["valu_issue"]["frac"] if issue else None,
        "traffic": traffic,
        "note": "decode + encode is bound by instruction issue, not by HBM (hbm.frac).  achieved / peak / frac = the VECTOR unit, the binding "
                "one: the call's vector instructions per second (committed per-frame counts x this run's rate) against "
                "ac3mi_probe_valu_rate of this run x 1 024 SIMDs (8 wavefronts per SIMD issuing nothing else; the kernels run 4 - 8 per SIMD with "
                "a scalar stream beside, where the vector stream tops out at 0.75 - 0.89 of that: extra.mixed_probe).  issue.salu = the same "
                "for the CUs' scalar units, issue.sum = both shares added (they are different units and overlap across wavefronts: it can "
                "exceed 1)",
        "hbm": hbm,
        "issue": {"valu": rl.get("valu_issue"), "salu": rl.get("salu_issue"), "frac": rl["valu_issue"]["frac"] if issue else None,
                  "sum": issue["frac"] if issue else None},
        "kernels": tc_kernels,
        "call_ms": call_ms,
        "valu_probe_ginst_per_s_per_simd": rate,
        "salu_probe_ginst_per_s_per_simd": srate,
    }

    # ---- secondary numbers (not the headline) ----
    extra = None
    if not args.no_extra:
        extra = {}
        # BASELINE configs[1], the HBM-bound kernel (the headline of rounds 1-3): IMDCT-512 + window + overlap-add
        desc = pkg.XformDesc(acmod=7, lfeon=1, output=7 | 16, bias=0.0)
        g = torch.Generator(device=dev).manual_seed(1234 + rank)
        coef = torch.randn((S, 1, 6, N_CH, 256), device=dev, generator=g, dtype=torch.float32) * (2.0 ** -8)
        delay = torch.zeros((S, N_CH, 128), device=dev, dtype=torch.float32)
        out = torch.empty((S, 1, 6, N_CH, 256), device=dev, dtype=torch.float32)
        torch.cuda.synchronize(dev)
        for _ in range(3):
            eng.imdct_batch(desc, coef, delay, None, out=out, wait_torch=False)
        barrier()
        eng.timer_start()
        for _ in range(50):
            eng.imdct_batch(desc, coef, delay, None, out=out, wait_torch=False)
        kernel_ms = eng.timer_stop() / 50
        barrier()
        if dist is not None:
            t = torch.tensor([kernel_ms], device=rdev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            kernel_ms = float(t[0])
        del coef, delay, out
        achieved = BYTES_PER_FRAME * S / (kernel_ms * 1e-3) / 1e9
        copy_gbs = None
        if rank == 0:
            try:
                copy_gbs = eng.probe_copy_rate(1 << 31)     # the guide's float4 copy on this box in this run (untimed)
            except Exception:
                copy_gbs = None
        xt, xt_file = measured_traffic(S)
        extra["transform_imdct512"] = {
            "workload": "BASELINE configs[1]: batched decode transform, %d independent 5.1/48kHz frames per GPU, IMDCT-512 + KBD window + "
                        "overlap-add, coefficients and PCM resident in HBM" % S,
            "frames_per_s_per_gpu": S / (kernel_ms * 1e-3), "ms_per_pass": kernel_ms, "dtype": "f32",
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": xt, "traffic_source": "committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel at this batch size (%s)" % xt_file,
                         "kernel": "ac3mi::xform_kernel<false, 4, false>", "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": BYTES_PER_FRAME * S,
                         "copy_probe": None if not copy_gbs else {
                             "GBps": copy_gbs, "frac_of_copy": achieved / copy_gbs,
                             "note": "bare float4 copy, one element per lane, 2 GiB per array, measured in this run (ac3mi_probe_copy_rate): "
                                     "the practical ceiling of a read+write stream on this box; `peak` stays the spec number"}}}
        extra.update(secondary_timings(pkg, eng, dev, C, rank, dist, barrier, checks=not args.no_checks, warm=not args.no_warm))
        for name in ("encode", "decode", "decode_s16", "transform_downmix_mixed_blocks"):
            extra[name]["roofline"] = leg_rooflines(name, extra[name]["algorithmic_bytes_per_frame"], extra[name]["frames_per_s_per_gpu"], rate, mix, srate)
        extra["mixed_probe"] = {"note": "10^9 (vector, scalar) instructions per second and SIMD, every wavefront issuing 3 vector per scalar instruction "
                                        "on independent registers (ac3mi_probe_mixed_rate), by wavefronts per SIMD",
                                "by_wavefronts_per_simd": {str(w): list(eng.probe_mixed_rate(w)) for w in (4, 5, 6, 7, 8)}}
        if rank == 0:
            extra["per_stream_curve"] = per_stream_curve(eng, C)
        million_skip = None
        if os.environ.get("AC3MI_BENCH_MILLION", "1") != "1":
            million_skip = "AC3MI_BENCH_MILLION=0"
        elif S < 65536:
            million_skip = "--frames below 65536"
        else:
            try:
                extra["transcode_million_streams"] = million_stream_transcode(pkg, eng, dev, C.frames)
            except RuntimeError as e:                       # (a GPU whose memory other jobs hold: the leg is left out, the line says why)
                million_skip = str(e)[:200]
        extra["transcode_million_streams_skipped"] = million_skip
        if dist is None and not args.no_checks:             # host-side legs: single-process runs only
            extra["stream_layer"] = stream_layer_timing(pkg, eng, C.frames[:8192].cpu().numpy())
            extra["dropin_single_stream"] = dropin_single_stream(tc_host[:48])

    if rank == 0:
        value = S * world * args.steps / dt
        line = {
            "metric": METRIC,
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32+i32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[4]'s per-GPU step at configs[1]/[2]'s batch size: %d independent 5.1/48kHz/384kbps one-frame "
                            "streams per GPU, decode (frames -> s16 PCM) + re-encode (AC3_encode_frame) in one ac3mi_transcode_batch call, "
                            "every step from fresh stream state (reset inside the step), frames in and out resident in HBM" % S,
                "frames_per_gpu": S,
                "channels": N_CH,
                "blocks_per_frame": 6,
                "parallelism": "streams sharded over %d GPU(s), no collective" % world,
                "frames_per_s_per_gpu": value / world,
                "realtime_x_per_gpu": value / world * 0.032,
            },
            "per_rank_frames_per_s": per_rank,
            "roofline": roofline,
            "all_frames_ok": tc_ok_decode and tc_bad_crc == 0,
            "frames_failing_crc": tc_bad_crc,
            # (kept from rounds 2-3 for readers of older lines: the same number as `value` per GPU, from the event-timed call)
            "decode_plus_encode_frames_per_s": fps_call * world,
            "decode_plus_encode_hbm_frac": hbm["frac"],
        }
        if extra is not None:
            line["extra"] = extra
        if not args.no_cpu_baseline:
            # (N > 1: rank 0 measures it after the barrier, the other ranks are done)
            cpu = cpu_baseline_codec(tc_host[:64].reshape(64, 1, C.fb).copy())
            d, e = cpu["decode"]["value"], cpu["encode"]["value"]
            line["cpu_baseline"] = {
                "value": 1.0 / (1.0 / d + 1.0 / e), "unit": "frames/s", "cores": cpu["decode"]["cores"], "kind": "port",
                "sample": "the same frames through the CPU path, decode then encode, whole loops in C on %d threads, ~4 s each: decode = the "
                          "reference's own liba52 (kind %s, %.0f frames/s), encode = the encoder oracle (kind port: ac3enc.cpp does not "
                          "build here, %.0f frames/s); value = 1 / (1/decode + 1/encode)" % (cpu["decode"]["cores"], cpu["decode"]["kind"], d, e),
                "parts": cpu}
            if extra is not None and world == 1:
                extra["transform_imdct512"]["cpu_baseline"] = cpu_baseline()
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)

    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

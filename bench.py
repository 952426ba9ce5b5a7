#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X AC-3 block-transform engine.

Workload (BASELINE.json configs[1]): batched decode transform — 65536 independent
5.1 / 48 kHz frames per GPU, each 6 blocks x 6 channels of 256 dequantised coefficients,
through IMDCT-512 + KBD window + overlap-add (the synthesis stage of a52_block).
One "step" = one pass of ac3mi_imdct_batch over that batch, inputs resident in HBM.

  python bench.py --gpus N --steps K --warmup W
  N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
  (RANK / LOCAL_RANK / WORLD_SIZE in the environment), or started plainly, in which case this process only
  spawns N child ranks of itself (before touching the GPU), one per device, and relays rank 0's line.
  Every rank owns its own 65536 streams - independent streams shard with no data-path collective: weak scaling.

Prints ONE JSON line on rank 0 (see DESIGN.md §6 for every field).
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

FRAMES_PER_GPU = 65536
N_CH = 6
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


def measured_traffic(frames):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/run_profile.sh),
    only if they were taken at this batch size; else None."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for fn in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if fn.endswith("_hbm_traffic.json"):
            try:
                d = json.load(open(os.path.join(pdir, fn)))
                if d.get("frames_per_launch") == frames and "xform_kernel<false" in d.get("kernel", ""):
                    best = d["hbm_bytes_per_launch"]
            except (OSError, ValueError, KeyError):
                pass
    return best


def instruction_mix():
    """Per-frame instruction counts of the engine kernels from the committed rocprofv3 PMC summary
    (profiles/*_instruction_mix.json, newest by name; recipe profiles/run_r02.sh).  None if absent."""
    pdir = os.path.join(ROOT, "profiles")
    names = sorted(fn for fn in (os.listdir(pdir) if os.path.isdir(pdir) else []) if fn.endswith("_instruction_mix.json"))
    if not names:
        return None
    try:
        d = json.load(open(os.path.join(pdir, names[-1])))
        d["file"] = "profiles/" + names[-1]
        return d
    except (OSError, ValueError):
        return None


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


def secondary_timings(pkg, eng, dev, S, rank, dist, barrier, steps=20, checks=True, warm=True):
    """Whole-path numbers for the other BASELINE configs on the same batch size (frames resident in HBM):
    configs[2] encode (s16 PCM -> frames), bitstream decode (frames -> float PCM, both kernels) and
    decode -> s16 -> re-encode (configs[4]'s per-GPU transcode step).  Each: frames/s of this rank's shard
    (MAX-reduced time), algorithmic GB/s per SURVEY.md §8d, x realtime."""
    import torch
    S = min(S, 65536)
    enc = pkg.EncodeDesc(48000, 384000, 6)
    fb = enc.frame_bytes()
    g = torch.Generator(device=dev).manual_seed(99 + rank)
    t = torch.arange(1536, device=dev, dtype=torch.float32)
    ph = torch.rand((S, 1, 6), device=dev, generator=g) * 6.28
    fr = 0.01 * torch.arange(1, 7, device=dev, dtype=torch.float32)
    pcm = 8000.0 * torch.sin(ph + fr * t[None, :, None]) + (torch.rand((S, 1536, 6), device=dev, generator=g) - 0.5) * 4096
    # level steps (x1 / x1/32 per 512-sample segment and channel) so that frames carry a realistic mix of new and
    # reused exponent sets: a stationary signal would reuse block 0's exponents five times in every channel
    env = torch.where(torch.rand((S, 3, 1, 6), device=dev, generator=g) < 0.5, 1.0, 1.0 / 32)
    pcm = (pcm.reshape(S, 3, 512, 6) * env).reshape(S, 1536, 6)
    pcm = pcm.round().clamp(-32768, 32767).to(torch.int16).reshape(S, 1, 1536, 6).contiguous()
    last = torch.zeros((S, 6, 256), dtype=torch.int16, device=dev)
    csnr = torch.full((S,), 40, dtype=torch.int32, device=dev)
    frames = torch.zeros((S, 1, fb), dtype=torch.uint8, device=dev)
    dec = pkg.DecodeDesc(flags=7 | 16 | 32, level=1.0, bias=384.0, dynrng=1, acmod=7, lfeon=1, frame_bytes=fb)
    delay = torch.zeros((S, 6, 128), dtype=torch.float32, device=dev)
    lfsr = torch.ones((S,), dtype=torch.int16, device=dev)
    out = torch.empty((S, 1, 6, 6, 256), dtype=torch.float32, device=dev)
    # one status array per leg: each leg's frame verdicts are read after all legs have run
    status = torch.zeros((S, 1), dtype=torch.int32, device=dev)
    status16 = torch.zeros((S, 1), dtype=torch.int32, device=dev)
    status_tc = torch.zeros((S, 1), dtype=torch.int32, device=dev)
    s16 = torch.empty((S, 6 * 256, 6), dtype=torch.int16, device=dev)
    frames2 = torch.zeros((S, 1, fb), dtype=torch.uint8, device=dev)
    chmap = (0, 2, 1, 4, 5, 3)
    import ctypes

    def do_enc(src=pcm, dst=frames):
        eng.encode_batch(enc, src, chmap, last, csnr, out=dst, wait_torch=False)

    def do_dec():
        eng.decode_batch(dec, frames, delay, lfsr, out=out, status=status, wait_torch=False)

    out16 = torch.empty((S, 1, 6, 256, 6), dtype=torch.int16, device=dev)
    delay16 = torch.zeros((S, 6, 128), dtype=torch.float32, device=dev)
    lfsr16 = torch.ones((S,), dtype=torch.int16, device=dev)

    def do_dec16():
        eng.decode_s16_batch(dec, frames, delay16, lfsr16, out=out16, status=status16, wait_torch=False)

    def do_cvt():
        eng._check(eng.lib.ac3mi_convert_s16_batch(ctypes.c_void_p(eng.ctx), ctypes.c_void_p(out.data_ptr()),
                                                  ctypes.c_void_p(s16.data_ptr()), 7 | 16, ctypes.c_size_t(S * 6)))

    delay2 = torch.zeros((S, 6, 128), dtype=torch.float32, device=dev)
    lfsr2 = torch.ones((S,), dtype=torch.int16, device=dev)
    last2 = torch.zeros((S, 6, 256), dtype=torch.int16, device=dev)
    csnr2 = torch.full((S,), 40, dtype=torch.int32, device=dev)

    def do_transcode():
        eng.transcode_batch(dec, enc, frames, delay2, lfsr2, chmap, last2, csnr2, out=frames2, status=status_tc, wait_torch=False)

    # BASELINE configs[3]: mixed short/long blocks (a quarter of the channel-blocks switched) with the 5.1 -> 2.0
    # downmix folded into the transform: 5 of the 6 planes in (liba52 drops the LFE), 2 planes out, 2 overlap tails
    mixdesc = pkg.XformDesc(7, 1, 2, 0.0)
    coef_mix = torch.randn((S, 1, 6, 6, 256), device=dev, generator=g) * 0.05
    blksw_mix = (torch.rand((S, 1, 6, 5), device=dev, generator=g) < 0.25).to(torch.uint8)
    delay_mix = torch.zeros((S, 2, 128), dtype=torch.float32, device=dev)
    out_mix = torch.empty((S, 1, 6, 2, 256), dtype=torch.float32, device=dev)

    def do_mix():
        eng.imdct_batch(mixdesc, coef_mix, delay_mix, blksw=blksw_mix, out=out_mix, wait_torch=False)

    # BASELINE configs[2] / configs[4] are defined on FRESH encoder state: every frame an independent stream with
    # last_samples = 0 and csnroffst = 40 (ENC/ac3enc.cpp:921, 969, 1092 - the SNR-offset search starts from the stream's
    # previous result, so a stream that re-encodes the same content starts at its own optimum from pass 2 on).  "cold" =
    # state put back to those values on the engine's stream before every timed pass, outside the timer (the leg's headline);
    # "warm" = the passes run on, each continuing the streams of the one before (the search's best case).
    csnr40 = torch.full((S,), 40, dtype=torch.int32, device=dev)
    lfsr1 = torch.ones((S,), dtype=torch.int16, device=dev)

    def reset_enc():
        eng.memset(last)
        eng.copy(csnr, csnr40)

    def reset_transcode():
        eng.memset(last2)
        eng.copy(csnr2, csnr40)
        eng.memset(delay2)
        eng.copy(lfsr2, lfsr1)

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
                                    ("decode_s16", do_dec16, 1536 + 18432 + 2 * 3072, None),
                                    ("transcode", do_transcode, 38400 + 19968, reset_transcode)):
        ms = timed(fn, reset)
        fps = S / (ms * 1e-3)
        res[name] = {"frames_per_s_per_gpu": fps, "ms_per_pass": ms, "algorithmic_GBps": nbytes * fps / 1e9,
                     "hbm_frac": nbytes * fps / 1e9 / HBM_PEAK_GBS, "realtime_x": fps * 0.032,
                     "algorithmic_bytes_per_frame": nbytes}
        if reset is not None:
            res[name]["state"] = "cold: encoder history 0 and csnroffst 40 before every timed pass (BASELINE's definition)"
            if not warm:
                continue
            wms = timed(fn, None)
            res[name]["warm"] = {"frames_per_s_per_gpu": S / (wms * 1e-3), "ms_per_pass": wms,
                                 "state": "warm: every pass continues the streams of the pass before (same content: the search starts at its optimum)"}
    # the issue ceiling of the instruction-bound kernels: plain VALU instructions per second and SIMD, chip-wide load
    rate = eng.probe_valu_rate()
    srate = eng.probe_salu_rate()
    res["salu_probe"] = {"ginst_per_s_per_simd": srate,
                         "note": "scalar instructions per second and SIMD (one scalar unit per CU, shared by its four SIMDs), chip-wide load"}
    res["valu_probe"] = {"ginst_per_s_per_simd": rate,
                         "note": "plain 32-bit VALU instructions one SIMD sustains with all SIMDs busy (8 wavefronts each); "
                                 "the valu_issue rooflines of the legs are priced against this x %d SIMDs" % N_SIMD}
    # ... and what the two pipes sustain TOGETHER at the integer kernels' own mix (three vector instructions per scalar one) and
    # occupancies: the vector pipe reaches its own ceiling only at 8 wavefronts per SIMD; the scalar stream rides along
    res["mixed_probe"] = {"note": "10^9 (vector, scalar) instructions per second and SIMD, every wavefront issuing 3 vector per scalar instruction "
                                  "on independent registers (ac3mi_probe_mixed_rate); the packers run 4 wavefronts per SIMD, the parse kernel 5, "
                                  "enc_mdct_kernel 7, mant_kernel 8",
                          "by_wavefronts_per_simd": {str(w): list(eng.probe_mixed_rate(w)) for w in (4, 5, 6, 7, 8)}}
    mix = instruction_mix()
    for name in ("encode", "decode", "decode_s16", "transcode", "transform_downmix_mixed_blocks"):
        res[name]["roofline"] = leg_rooflines(name, res[name]["algorithmic_bytes_per_frame"], res[name]["frames_per_s_per_gpu"], rate, mix, srate)
    # ---- untimed epilogue: every leg's verdict on the whole batch, from its own buffers ----
    eng.sync()
    torch.cuda.synchronize(dev)
    res["decode"]["all_frames_ok"] = int((status & 0x1ff).max().item()) == 0
    res["decode_s16"]["all_frames_ok"] = int((status16 & 0x1ff).max().item()) == 0
    res["transcode"]["all_frames_decoded_ok"] = int((status_tc & 0x1ff).max().item()) == 0
    enc_host = frames.cpu().numpy().reshape(S, -1)[:, :fb]
    tc_host = frames2.cpu().numpy().reshape(S, -1)[:, :fb]
    res["encode"]["frames_failing_crc"] = ac3_crc_ok(enc_host)
    res["transcode"]["frames_failing_crc"] = ac3_crc_ok(tc_host)
    res["encode"]["all_frames_ok"] = res["encode"]["frames_failing_crc"] == 0
    res["transcode"]["all_frames_ok"] = res["transcode"]["frames_failing_crc"] == 0 and res["transcode"]["all_frames_decoded_ok"]
    if rank == 0 and checks:
        res["encode"]["bit_exact_vs_oracle"] = check_against_oracle(pkg, eng, dev, enc, dec, chmap, pcm)
    res["_frames"] = enc_host[:64].reshape(64, 1, fb).copy()      # for the CPU rates beside these legs (dropped from the line)
    if os.environ.get("AC3MI_BENCH_MILLION", "1") == "1" and S >= 65536:
        try:
            res["transcode_million_streams"] = million_stream_transcode(pkg, eng, dev, frames)
        except RuntimeError as e:                       # (a GPU whose memory other jobs hold: the leg is left out, the line says why)
            res["transcode_million_streams"] = {"skipped": str(e)[:200]}
    if dist is None and checks:           # host-side work on up to 16 threads: single-process runs only
        res["stream_layer"] = stream_layer_timing(pkg, eng, frames[:8192].cpu().numpy())
    res["note"] = ("secondary timings on %d frames/GPU (5.1, 48 kHz, 384 kbps); encode and the decode front end are "
                   "integer/latency-bound, not HBM-bound: hbm_frac is reported for completeness" % S)
    return res


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
    the engine's tiled workspace ~20 GB)."""
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
    free0, total = torch.cuda.mem_get_info(dev)
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
    free1, _ = torch.cuda.mem_get_info(dev)
    ok = int((status & 0x1ff).max().item()) == 0
    same = bool(torch.equal(out[:S0], out[S0:2 * S0])) if n_streams >= 2 * S0 else None
    return {"streams": n_streams, "ms_per_pass": ms, "frames_per_s_per_gpu": n_streams / (ms * 1e-3), "all_frames_ok": ok, "state": "cold (fresh stream state every pass)",
            "replicas_agree_across_tiles": same, "engine_workspace_GB": (free0 - free1) / 1e9, "hbm_total_GB": total / 1e9}


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
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "streams_total": int(total), "max_time": tmax,
                          "local_rank": local_rank, "max_rank_hbm_plan_bytes": pmax, "ranks_that_fit_288GB": int(fits)}), flush=True)
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
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary decode/encode/transcode timings")
    ap.add_argument("--no-checks", action="store_true", help="skip the untimed oracle check and the stream-layer leg (profiling runs)")
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

    pkg = importlib.import_module("ac-3-acm-codec_amd")
    eng = pkg.Engine(local_rank)                      # fails loudly without libac3mi.so / a GPU
    desc = pkg.XformDesc(acmod=7, lfeon=1, output=7 | 16, bias=0.0)

    S = args.frames
    # this rank's contiguous shard of the job's S x world independent streams (no data-path collective)
    lo, hi = pkg.sharding.shard(S * world, world, rank)
    assert hi - lo == S
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    coef = torch.randn((S, 1, 6, N_CH, 256), device=dev, generator=g, dtype=torch.float32) * (2.0 ** -8)
    delay = torch.zeros((S, N_CH, 128), device=dev, dtype=torch.float32)
    out = torch.empty((S, 1, 6, N_CH, 256), device=dev, dtype=torch.float32)

    def step():
        eng.imdct_batch(desc, coef, delay, None, out=out, wait_torch=False)

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
    eng.timer_start()
    for _ in range(args.steps):
        step()
    kernel_ms = eng.timer_stop() / args.steps          # HIP events on the engine's own stream
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt, kernel_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, kernel_ms = float(t[0]), float(t[1])

    # the guide's float4 copy on this box in this run: what a read+write stream reaches at best (rank 0, untimed)
    copy_gbs = None
    if rank == 0:
        try:
            copy_gbs = eng.probe_copy_rate(1 << 31)
        except Exception:                                   # (an older library: the line simply lacks the field)
            copy_gbs = None

    # ---- secondary timings (not the headline): full frame decode, encode, transcode ----
    extra = None
    if not args.no_extra:
        extra = secondary_timings(pkg, eng, dev, S, rank, dist, barrier, checks=not args.no_checks, warm=not args.no_warm)

    if rank == 0:
        total_frames = S * world * args.steps
        value = total_frames / dt
        achieved = BYTES_PER_FRAME * S / (kernel_ms * 1e-3) / 1e9
        line = {
            "metric": "AC-3 5.1@48kHz frames/sec (batched decode transform: IMDCT-512 + window + overlap-add)",
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: batched decode, %d independent 5.1/48kHz frames per GPU, "
                            "IMDCT-512 + KBD window + overlap-add, coefficients and PCM resident in HBM" % S,
                "frames_per_gpu": S,
                "channels": N_CH,
                "blocks_per_frame": 6,
                "parallelism": "streams sharded over %d GPU(s), no collective" % world,
                "realtime_x_per_gpu": value / world * 0.032,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic(S),
                "traffic_source": "committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel at this batch size "
                                  "(profiles/*_hbm_traffic.json), not collected in this run",
                "kernel": "ac3mi::xform_kernel<false, 4, false>",
                "kernel_ms": kernel_ms,
                "algorithmic_bytes_per_launch": BYTES_PER_FRAME * S,
                "copy_probe": None if not copy_gbs else {
                    "GBps": copy_gbs, "frac_of_copy": achieved / copy_gbs,
                    "note": "bare float4 copy, one element per lane, 2 GiB per array, measured in this run "
                            "(ac3mi_probe_copy_rate): the practical ceiling of a read+write stream on this box; `peak` stays the spec number"},
            },
        }
        if extra is not None:
            line["extra"] = extra
            # BASELINE's metric wording, "frames/sec/GPU (decode+encode)": the one-call transcode, fresh stream state
            line["decode_plus_encode_frames_per_s"] = extra["transcode"]["frames_per_s_per_gpu"] * world
            line["decode_plus_encode_hbm_frac"] = extra["transcode"]["hbm_frac"]
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline()
            if extra is not None and "_frames" in extra:
                extra["cpu"] = cpu_baseline_codec(extra.pop("_frames"))
        else:
            line["cpu_baseline"] = None
        if extra is not None:
            extra.pop("_frames", None)
        print(json.dumps(line), flush=True)

    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

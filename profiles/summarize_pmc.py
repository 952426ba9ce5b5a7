#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of profiles/run_r04.sh (run_r03.sh, run_r02.sh) into the summaries committed under profiles/:
   <tag>_kernel_stats.csv      per-kernel durations (kernel trace of the whole bench, secondary legs included)
   <tag>_hbm_traffic.json      per kernel: FETCH_SIZE (doubled, MI355X_MICROARCH.md 'HBM') + WRITE_SIZE per launch vs the
                               algorithmic bytes of SURVEY.md 8(d)
   <tag>_instruction_mix.json  per kernel: VALU / SALU / LDS instructions and wave-cycle breakdown per frame; bench.py's
                               valu_issue rooflines read this file
Counters come from separate --pmc passes; only dispatches of the full batch (largest grid of each kernel) are averaged."""
import collections
import csv
import glob
import json
import os
import re
import sys

out, tag, frames = sys.argv[1], sys.argv[2], int(sys.argv[3])


def rows(pattern):
    for fn in glob.glob(os.path.join(out, pattern)):
        yield from csv.DictReader(open(fn))


def short(name):
    name = name.replace("void ", "").replace("ac3mi::", "")
    name = name.split("(")[0].strip()
    # xform_kernel<MIX, WPS, S16, MS>: the plain kernels keep their three-parameter names
    return re.sub(r"(xform_kernel<\w+, \d+, \w+), false>", r"\1>", name)


def per_kernel(pattern, counters):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
    for r in rows(pattern):
        if "ac3mi" not in r["Kernel_Name"] or r["Counter_Name"] not in counters:
            continue
        acc[short(r["Kernel_Name"])][int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, by_grid in acc.items():
        g = max(by_grid)
        res[k] = {c: sum(v) / len(v) for c, v in by_grid[g].items()}
        res[k]["grid_size"] = g
        res[k]["launches_averaged"] = len(next(iter(by_grid[g].values())))
    return res


# ---- kernel stats: rocprofv3's own per-kernel table averages EVERY launch (the bench's small rounds included), so the
#      committed summary is recomputed from the kernel trace over the launches of each kernel's LARGEST grid (the full batch)
import statistics
by = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows(os.path.join("trace", "*kernel_trace.csv")):
    k = r["Kernel_Name"]
    if "at::native" in k or "rocclr" in k:
        continue
    g = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1) if "Grid_Size_X" in r else int(r["Grid_Size"])
    by[k][g].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(os.path.join(out, "summary_kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev", "GridSize", "LaunchesOfOtherGridSizes"])
    tab = []
    for k, bg in by.items():
        g = max(bg)
        d = bg[g]
        tab.append([k[:110], len(d), sum(d), sum(d) / len(d), min(d), max(d), statistics.pstdev(d) if len(d) > 1 else 0.0, g,
                    sum(len(v) for gg, v in bg.items() if gg != g)])
    for row in sorted(tab, key=lambda t: -t[2]):
        w.writerow(row)

# ---- HBM traffic
# algorithmic bytes per frame by kernel: what the kernel must move when nothing is re-read
ALG = {
    "xform_kernel<false, 4, false>": 36864 + 36864 + 6144,
    "xform_kernel<false, 3, true>": 36864 + 18432 + 6144,
    "xform_kernel<true, 2, false>": 30720 + 12288 + 2048,
    "xform_kernel<true, 3, false>": 30720 + 12288 + 2048,
    "decode_kernel<0>": 1536 + 36864,
    "decode_kernel<4>": 1536 + 480,             # frame in, six block descriptors out (+ the rows that changed: data-dependent)
    "mant_kernel": 1536 + 480 + 36864,          # frame + descriptors in (+ the rows of every segment), planes out
    "mantx_kernel<true>": 1536 + 480 + 18432 + 6144,    # ... s16 PCM out instead, overlap state in and out
    "mantx_kernel<false>": 1536 + 480 + 36864 + 6144,   # ... float PCM out
    "decode_wg_kernel<0>": 1536 + 36864,
    "decode_wg_kernel<1>": 1536 + 36864 + 6144,
    "decode_wg_kernel<2>": 1536 + 18432 + 6144,
    "enc_mdct_kernel": 18432 + 6144,
    "enc_pack_kernel<0>": 1536,
    "enc_pack_kernel<2>": 1536,
    "enc_packf_kernel<true>": 1536,
    "enc_packf_kernel<false>": 1536,
    "enc_packb_kernel": 1536,
}
fetch = per_kernel("pmc_fetch/*counter_collection.csv", {"FETCH_SIZE"})
write = per_kernel("pmc_write/*counter_collection.csv", {"WRITE_SIZE"})
traffic = {}
for k in sorted(set(fetch) | set(write)):
    f = fetch.get(k, {}).get("FETCH_SIZE", 0.0) * 1024
    w = write.get(k, {}).get("WRITE_SIZE", 0.0) * 1024
    d = {"fetch_bytes_raw": f, "fetch_bytes_corrected_x2": 2 * f, "write_bytes": w, "hbm_bytes_per_launch": 2 * f + w,
         "hbm_bytes_per_frame": (2 * f + w) / frames, "grid_size": fetch.get(k, write.get(k))["grid_size"]}
    if k in ALG:
        d["algorithmic_bytes_per_frame"] = ALG[k]
        d["ratio_to_algorithmic"] = d["hbm_bytes_per_frame"] / ALG[k]
    traffic[k] = d
json.dump({"frames_per_launch": frames, "kernels": traffic,
           "kernel": "ac3mi::xform_kernel<false, 4, false>",
           "hbm_bytes_per_launch": traffic.get("xform_kernel<false, 4, false>", {}).get("hbm_bytes_per_launch"),
           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE as "
                   "read; separate --pmc passes; AC3MI_NO_OVERLAP=1 so that one launch of a kernel covers the whole batch"},
          open(os.path.join(out, "summary_hbm_traffic.json"), "w"), indent=1)

# ---- instruction mix
mixa = per_kernel("pmc_mix_a/*counter_collection.csv", {"SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
                                                         "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"})
kern = {}
for k, d in mixa.items():
    if "SQ_INSTS_VALU" not in d:
        continue
    kern[k] = {"valu_per_frame": d["SQ_INSTS_VALU"] / frames, "salu_per_frame": d.get("SQ_INSTS_SALU", 0) / frames,
               "lds_per_frame": d.get("SQ_INSTS_LDS", 0) / frames, "waves": d.get("SQ_WAVES"),
               "wave_cycles_per_frame": d.get("SQ_WAVE_CYCLES", 0) / frames,
               "wait_any_share": d.get("SQ_WAIT_ANY", 0) / max(d.get("SQ_WAVE_CYCLES", 1), 1),
               "wait_inst_any_share": d.get("SQ_WAIT_INST_ANY", 0) / max(d.get("SQ_WAVE_CYCLES", 1), 1),
               "active_inst_any_share": d.get("SQ_ACTIVE_INST_ANY", 0) / max(d.get("SQ_WAVE_CYCLES", 1), 1),
               "grid_size": d["grid_size"]}


def pick(*prefixes):
    # (only kernels whose largest launch covered the whole batch - at least a wavefront per frame: the small-batch kernels of
    # the per-stream curve would otherwise enter a leg with counts divided by the wrong number of frames; the transform kernels run
    # eight chains per wavefront)
    return [k for k in kern if any(k.startswith(p) for p in prefixes) and (kern[k].get("waves") or 0) >= frames // 8]


# what one pass of a bench leg launches (round 4: to s16 the mantissa kernel transforms too, decode_mx.hip; the s16 transform
# kernel runs in the `two_kernels` A/B of the decode_s16 leg only)
legs = {
    "decode": pick("decode_kernel<4>", "mant_kernel", "xform_kernel<false, 4, false>"),
    "decode_s16": pick("decode_kernel<4>", "mantx_kernel<true>"),
    "decode_s16_two_kernels": pick("decode_kernel<4>", "mant_kernel", "xform_kernel<false, 3, true>"),
    "encode": pick("enc_mdct_kernel", "enc_search_kernel<1>", "enc_packf_kernel"),
    "transform_downmix_mixed_blocks": pick("xform_kernel<true, 2, false>", "xform_kernel<true, 3, false>"),
}
legs["transcode"] = sorted(set(legs["decode_s16"]) | set(legs["encode"]))
json.dump({"frames_per_launch": frames, "kernels": kern, "legs": legs,
           "note": "rocprofv3 --pmc pass (SQ counters summed over the chip, divided by the frames of one launch); wave-cycle "
                   "counters are in units of 4 cycles; legs list the kernels one pass of a bench leg launches"},
          open(os.path.join(out, "summary_instruction_mix.json"), "w"), indent=1)
for k, d in sorted(kern.items()):
    print("%-34s valu %7.0f salu %7.0f lds %6.0f /frame  wait_any %.2f wait_inst %.2f active %.2f" %
          (k, d["valu_per_frame"], d["salu_per_frame"], d["lds_per_frame"], d["wait_any_share"], d["wait_inst_any_share"], d["active_inst_any_share"]))
for k, d in sorted(traffic.items()):
    print("%-34s hbm %8.0f B/frame  %s" % (k, d["hbm_bytes_per_frame"], "x%.2f of algorithmic" % d["ratio_to_algorithmic"] if "ratio_to_algorithmic" in d else ""))

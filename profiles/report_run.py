#!/usr/bin/env python3
"""The figures the docs quote, from one run of profiles/run_r04.sh + bench.py: `python profiles/report_run.py [gpurun_out/prof_TAG]
[bench.json]` - kernel table, the encoder's kernels per leg (encode / transcode) from the kernel trace, the bench line's legs."""
import collections
import csv
import json
import statistics
import sys

prof = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r04"
bench = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/bench_r04.json"
for row in list(csv.reader(open(prof + "/summary_kernel_stats.csv")))[1:12]:
    print("%-62s n=%3s avg %8.1f us min %8.1f" % (row[0][:62], row[1], float(row[3]) / 1e3, float(row[4]) / 1e3))
rows = [r for r in csv.DictReader(open(prof + "/trace/trace_kernel_trace.csv")) if "ac3mi" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
acc = collections.defaultdict(list)
for i, r in enumerate(rows):
    n = r["Kernel_Name"]
    if "enc_" in n and r["Grid_Size_X"] in ("25165824", "4194304"):
        j = i
        while "enc_" in rows[j]["Kernel_Name"]:
            j -= 1
        leg = "transcode" if "mantx" in rows[j]["Kernel_Name"] and int(rows[i]["Start_Timestamp"]) - int(rows[j]["End_Timestamp"]) < 5e6 else "encode"
        acc[(n.split("(")[0][-28:], leg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items()):
    print("%-30s %-9s n=%2d mean %7.0f us min %7.0f" % (k[0], k[1], len(v), statistics.mean(v), min(v)))
for f in (bench, prof + "/trace_bench.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r, e = d["roofline"], d["extra"]
    print(f)
    print("  value %.0f ms/step %.3f call_ms %.3f | roofline.frac %.3f (valu) salu %.3f sum %.3f | hbm.frac %.4f traffic %.2f GB" % (
        d["value"], d["ms_per_step"], r["call_ms"], r["frac"], r["issue"]["salu"]["frac"], r["issue"]["sum"], r["hbm"]["frac"], r["traffic"] / 1e9))
    print("  decode_s16 %.3f (two kernels %.3f) decode %.3f encode %.3f mixed %.3f imdct %.2f M frames/s" % (
        e["decode_s16"]["ms_per_pass"], e["decode_s16"]["two_kernels"]["ms_per_pass"], e["decode"]["ms_per_pass"], e["encode"]["ms_per_pass"],
        e["transform_downmix_mixed_blocks"]["ms_per_pass"], e["transform_imdct512"]["frames_per_s_per_gpu"] / 1e6))
    if "warm" in e["encode"]:
        print("  warm: encode %.3f transcode %.3f" % (e["encode"]["warm"]["ms_per_pass"], e["transcode_warm"]["ms_per_pass"]))
    print("  curve", [(x["streams"], round(x["ms_per_round"], 3), round(x["realtime_x_per_stream"], 1), round(x["aggregate_frames_per_s"] / 1e6, 2)) for x in e["per_stream_curve"]["rounds"]])
    if e.get("dropin_single_stream"):
        print("  dropin", {k: e["dropin_single_stream"][k] for k in ("decode_frames_per_s", "encode_frames_per_s", "decode_plus_encode_frames_per_s", "realtime_x")})
    if e.get("transcode_million_streams"):
        print("  million", {k: e["transcode_million_streams"][k] for k in ("ms_per_pass", "frames_per_s_per_gpu", "engine_workspace_GB", "planned_GB")})
    if e.get("stream_layer"):
        print("  stream layer %.0f frames/s" % e["stream_layer"]["frames_per_s"])
    if d.get("cpu_baseline"):
        print("  cpu", round(d["cpu_baseline"]["value"]), {k: round(v["value"]) for k, v in d["cpu_baseline"]["parts"].items()})

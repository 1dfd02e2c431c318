#!/usr/bin/env python
"""Condense gpurun_out/prof_<tag>/ (scripts/profile_bench.sh) into profiles/<tag>_*: the rocprofv3 kernel stats as they
are, and per-kernel HBM traffic from the two PMC passes (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
gfx950; counter unit KiB)."""
import collections
import csv
import os
import shutil
import sys

tag = sys.argv[1]
src = os.path.join("gpurun_out", "prof_" + tag)
os.makedirs("profiles", exist_ok=True)
shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join("profiles", tag + "_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "trace_serial", "trace_kernel_stats.csv")):
    shutil.copy(os.path.join(src, "trace_serial", "trace_kernel_stats.csv"), os.path.join("profiles", tag + "_kernel_stats_serial_launch.csv"))
with open(os.path.join(src, "bench_trace.log")) as f:
    lines = [l for l in f if l.startswith("{")]
with open(os.path.join("profiles", tag + "_bench_under_rocprof.json"), "w") as f:
    f.writelines(lines)
agg = collections.defaultdict(lambda: dict(n=0, fetch=0.0, write=0.0, vgpr=0, lds=0, scratch=0))
for name in ("fetch", "write"):
    p = os.path.join(src, "pmc_" + name, "pmc_counter_collection.csv")
    if not os.path.exists(p):
        continue
    for r in csv.DictReader(open(p)):
        a = agg[r["Kernel_Name"]]
        if name == "fetch":
            a["n"] += 1
        a[name] += float(r["Counter_Value"])
        a["vgpr"], a["lds"], a["scratch"] = r["VGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"]
with open(os.path.join("profiles", tag + "_pmc_hbm_traffic.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches", "FETCH_SIZE_KiB_per_launch_raw", "fetch_MB_per_launch_x2_corrected", "WRITE_SIZE_KiB_per_launch",
                "write_MB_per_launch", "VGPR", "LDS_bytes", "scratch_bytes"])
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["write"]):
        n = max(1, a["n"])
        w.writerow([k, a["n"], round(a["fetch"] / n, 2), round(2 * a["fetch"] / n * 1024 / 1e6, 3), round(a["write"] / n, 2),
                    round(a["write"] / n * 1024 / 1e6, 3), a["vgpr"], a["lds"], a["scratch"]])
print(open(os.path.join("profiles", tag + "_pmc_hbm_traffic.csv")).read())

# per-launch HBM bytes (fetch x2 + write) of the buildSystem kernels, for bench.py's roofline.traffic
import json
short = {"k_reproj_build": "k_reproj", "k_stream<mc::ImuBatchView, true": "k_stream<Imu>", "k_stream<mc::LidarBatchView, true": "k_stream<Lidar>",
         "k_motion(": "k_motion", "k_static<true>": "k_static"}
out = {"tag": tag, "command": "python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline", "unit": "bytes per launch", "kernels": {}}
for k, a in agg.items():
    for pat, name in short.items():
        if pat in k:
            n = max(1, a["n"])
            out["kernels"][name] = dict(fetch=2 * a["fetch"] / n * 1024, write=a["write"] / n * 1024)
with open(os.path.join("profiles", "hbm_traffic_latest.json"), "w") as f:
    json.dump(out, f, indent=1)

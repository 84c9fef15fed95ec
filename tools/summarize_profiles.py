#!/usr/bin/env python3
"""Summarise gpurun_out/prof (written by tools/collect_profiles.sh on the GPU box) into profiles/<tag>_*:
   <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of `python bench.py --steps 20 --warmup 3 --no-cpu`
   <tag>_bench.json         the bench line printed by that same (profiled) run and by an unprofiled run
   <tag>_pmc_traffic.json   FETCH_SIZE / WRITE_SIZE per launch of the dominant kernel, with the gfx950 calibration
                            (tools/ubench/hbm_calib: 2 GiB streamed with this solver's 8 B/lane accesses)
"""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/prof"
dst = "profiles"
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "stats", "bench_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))


def counter(path, kernel_sub, name):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if kernel_sub in r["Kernel_Name"] and r["Counter_Name"] == name]
    return sum(v) / len(v), len(v)


calib_bytes = float(1 << 31)  # hbm_calib streams 2 GiB per kernel
cf, _ = counter(os.path.join(src, "calib_fetch", "calib_counter_collection.csv"), "read_kernel", "FETCH_SIZE")
cw, _ = counter(os.path.join(src, "calib_write", "calib_counter_collection.csv"), "write_kernel", "WRITE_SIZE")
fetch_factor = calib_bytes / (cf * 1024.0)   # bytes per reported KiB unit / 1024
write_factor = calib_bytes / (cw * 1024.0)
f, nf = counter(os.path.join(src, "fetch", "bench_counter_collection.csv"), "nmpc_ipm_", "FETCH_SIZE")
w, nw = counter(os.path.join(src, "write", "bench_counter_collection.csv"), "nmpc_ipm_", "WRITE_SIZE")
stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(src, "stats", "bench_kernel_stats.csv")))}
kname = [k for k in stats if "nmpc_ipm_" in k][0]
bench_prof = json.loads(open(os.path.join(src, "bench_stats.json")).read().strip().splitlines()[-1])
bench_plain = json.loads(open(os.path.join(src, "bench_plain.json")).read().strip().splitlines()[-1])
traffic = {
    "kernel": kname, "batch": bench_plain["config"]["batch_per_gpu"],
    "FETCH_SIZE_raw_KiB_per_launch": f, "WRITE_SIZE_raw_KiB_per_launch": w, "launches_averaged": [nf, nw],
    "calibration": {"what": "tools/ubench/hbm_calib: 2 GiB streamed once with coalesced 8 B/lane loads / stores",
                    "FETCH_SIZE_raw_KiB": cf, "WRITE_SIZE_raw_KiB": cw,
                    "fetch_factor": fetch_factor, "write_factor": write_factor,
                    "note": "gfx950 FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section): factor 2; WRITE_SIZE exact"},
    "fetch_bytes_per_launch": f * 1024.0 * fetch_factor, "write_bytes_per_launch": w * 1024.0 * write_factor,
    "rocprof_avg_kernel_ns": float(stats[kname]["AverageNs"]), "rocprof_calls": int(stats[kname]["Calls"]),
}
traffic["bytes_per_launch"] = traffic["fetch_bytes_per_launch"] + traffic["write_bytes_per_launch"]
traffic["achieved_TBps_fabric"] = traffic["bytes_per_launch"] / traffic["rocprof_avg_kernel_ns"] / 1e3
json.dump(traffic, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
json.dump({"profiled_run": bench_prof, "unprofiled_run": bench_plain}, open(os.path.join(dst, f"{tag}_bench.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
print("bench kernel_ms (HIP events):", bench_plain["roofline"]["kernel_ms"], " rocprof avg ms:", traffic["rocprof_avg_kernel_ns"] / 1e6)

#!/usr/bin/env python3
"""Mean / median / min duration of the solver kernel from a rocprofv3 --kernel-trace csv (no --stats): tools/collect_r06.sh.
VERDICT r05 item 9a: the --stats AVERAGE of round 5 (0.861 ms) was above the unprofiled step of the same box (0.852 ms); the average is pulled up by the
dispatches the profiler's own bookkeeping delays -- the median is what a launch takes.   python tools/r06/kernel_trace_median.py <dir>"""
import csv, glob, json, os, sys
import statistics as st
d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        if "nmpc_ipm_" in r.get("Kernel_Name", ""):
            rows.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
bench = None
p = os.path.join(d, "bench_under_trace.json")
if os.path.exists(p):
    try: bench = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception: pass
out = {"what": "rocprofv3 --kernel-trace (no --stats) around `python bench.py --steps 20 --warmup 3 --no-cpu --streams 1`: durations of the solver kernel's dispatches, ms",
       "dispatches": len(rows), "mean_ms": st.mean(rows) if rows else None, "median_ms": st.median(rows) if rows else None, "min_ms": min(rows) if rows else None,
       "p90_ms": (sorted(rows)[int(0.9 * len(rows))] if rows else None),
       "bench_line_of_the_traced_run": ({"ms_per_step": bench["ms_per_step"], "kernel_ms_hip_events": bench["roofline"]["kernel_ms"]} if bench else None)}
print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""Per-kernel roofline accounting of the device tick: every kernel of DeviceFleet.full_tick with its bound, its
algorithmic work per launch, the achieved rate and the fraction of the MI355X peak -- from the rocprofv3 --kernel-trace --stats summary of
`python tools/full_tick_bench.py 4096 10 20000 0.5 0` (tools/collect_r05.sh) and that run's own JSON line; the A* from tests/tools/astar_bench.py.

   python tools/tick_rooflines.py <kernel_stats.csv> <full_tick.json> [astar_bench.jsonl] > profiles/r05_tick_rooflines.json

Work definitions (per planner and tick; B planners, N = 20 stages) -- stated here so that the fractions can be recomputed:
  solve      SURVEY 8d: mean_it * N * F_stage flop with F_stage = 31.5 k + 18 (m - 6) k... i.e. 31.5e3 + 18 * (m - 6) flop for m live corridor rows
             per stage (m = mean rows of the fleet's polytopes); FP64 issue bound, priced against 78.6 TFLOP/s
  corridor   per decomposition: candidates * 51 flop (local-box test in the box frame 30 + seed-ellipsoid distance 21) over the grid rows under
             the box's hull + in_box * 21 (distances in the final ellipsoid) + planes * in_box / 2 * 8 (plane tests; half the set is still alive on
             average); bytes: candidates * 28 (cell-sorted point + index) + in_box * 24 (gather into the register tile).  Latency-bound chain of
             ~21 dependent rounds per decomposition; reported against the FP64 vector peak AND as L2-side bytes/s
  tube       1.26 Mflop per planner (Gramian quadrature + Taylor steps + Jacobi sqrtm per stage and channel); FP64 VALU bound
  pack       HBM: N * (10 + 4 M) * 8 B parameters + N * 17 * 8 + 72 B written, N * (72 + 24 + 8) + polytopes 2 * rows * 32 B read
  reference / update / mode: launch-bound (tens of KB per launch): reported as time only
  astar      one expansion of ONE search = ~900 collision samples (survivors * check_num) of ~500 FP64-heavy instructions (~800 before the cell box and the
             shared reciprocals of round 4) on one CU (the FP64 issue
             rate of a CU: 64 lanes * 4 SIMDs / 4 cycles = 64 lane-instructions per cycle) + the serial pop / commit; reported per expansion of
             the longest search, which is what a batch waits for"""
import csv, json, sys

FP64_PEAK, HBM_PEAK = 78.6e12, 8.0e12
stats = {r["Name"]: r for r in csv.DictReader(open(sys.argv[1]))}
tick = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
B = int(tick["workload"].split()[0]); N, M = 20, 30
rows = float(tick.get("mean_rows", 25.0)); dec = float(tick["polytopes_per_planner"])
cand, inbox = float(tick.get("grid_candidates_per_decomposition", 2000.0)), float(tick.get("in_box_points_per_decomposition", 1000.0))


def avg_ms(sub):
    ks = [k for k in stats if sub in k]
    if not ks:
        return None, 0, None
    tot = sum(float(stats[k]["TotalDurationNs"]) for k in ks); calls = sum(int(stats[k]["Calls"]) for k in ks)
    return tot / calls / 1e6, calls, ks


out = {"workload": tick["workload"], "ms_per_step_hip_events": tick["ms_per_step"], "ms_per_tick": tick["ms_per_tick"], "kernels": {}}
# solve
ms, calls, ks = avg_ms("nmpc_ipm_lds_kernel")
if ms:
    f_stage = 31.5e3 + 18.0 * (rows - 6.0)
    fl = B * tick["mean_iters"] * N * f_stage
    out["kernels"]["solve"] = {"kernel": ks, "avg_ms": ms, "calls": calls, "bound": "fp64-issue", "algorithmic_flop_per_launch": fl,
                               "achieved_TFLOPs": fl / ms / 1e9, "frac": fl / (ms * 1e-3) / FP64_PEAK,
                               "note": f"mean {tick['mean_iters']:.2f} iterations (warm ticks), {rows:.0f} live rows per stage: the (20, 10) variant that re-reads the rows"}
# corridor
ms, calls, ks = avg_ms("corridor_wave_kernel")
ms2, calls2, ks2 = avg_ms("corridor_kernel")
if ms:
    planes = max(rows - 6.0, 0.0)
    fl = B * dec * (cand * 51 + inbox * 21 + planes * inbox * 0.5 * 8)
    by = B * dec * (cand * 28 + inbox * 24)
    tot = ms + (ms2 or 0.0) * (calls2 / max(calls, 1)) if ms2 else ms
    out["kernels"]["corridor"] = {"kernel": ks + (ks2 or []), "avg_ms_wave_kernel": ms, "avg_ms_fallback_kernels_per_tick": (ms2 or 0.0) * (calls2 / max(calls, 1)) if ms2 else 0.0,
                                  "calls": calls, "bound": "latency (chain of ~21 dependent rounds per decomposition), FP64 VALU",
                                  "algorithmic_flop_per_launch": fl, "achieved_TFLOPs": fl / tot / 1e9, "frac": fl / (tot * 1e-3) / FP64_PEAK,
                                  "l2_side_bytes_per_launch": by, "achieved_TBps_l2_side": by / tot / 1e9,
                                  "decompositions_per_planner": dec, "note": "the two workgroup kernels behind it only run for planners it flags (none on this workload): their time is launch overhead"}
ms, calls, ks = avg_ms("tube")
if ms:
    fl = B * 1.26e6
    out["kernels"]["tube"] = {"kernel": ks, "avg_ms": ms, "calls": calls, "bound": "fp64-valu", "algorithmic_flop_per_launch": fl, "achieved_TFLOPs": fl / ms / 1e9,
                              "frac": fl / (ms * 1e-3) / FP64_PEAK}
ms, calls, ks = avg_ms("pack")
if ms:
    by = B * (N * (10 + 4 * M) * 8 + N * 17 * 8 + 72 + N * (72 + 24 + 8) + dec * rows * 32)
    out["kernels"]["pack"] = {"kernel": ks, "avg_ms": ms, "calls": calls, "bound": "hbm", "algorithmic_bytes_per_launch": by, "achieved_GBps": by / ms / 1e6,
                              "frac": by / (ms * 1e-3) / HBM_PEAK}
for name, sub in (("reference", "reference"), ("update", "update"), ("coldstart", "coldstart")):
    ms, calls, ks = avg_ms(sub)
    if ms:
        out["kernels"][name] = {"kernel": ks, "avg_ms": ms, "calls": calls, "bound": "launch latency (a few hundred KB per launch)"}
if len(sys.argv) > 3:
    for line in open(sys.argv[3]):
        a = json.loads(line)
        if a.get("world") != "pillars":
            continue
        us = a["gpu_us_per_expansion_of_the_longest_search"]
        lane_instr = 900 * 500.0
        out["kernels"]["astar"] = {"kernel": "astar_kernel (1024 threads per planner)", "bound": "latency of the serial expansion loop; inside an expansion the FP64 issue rate of ONE CU",
                                   "searches_per_s": a["gpu_searches_per_s"], "us_per_expansion_of_the_longest_search": us, "expansions_of_the_longest_search": a["expansions_max"],
                                   "algorithmic_lane_instructions_per_expansion": lane_instr,
                                   "achieved_lane_instructions_per_cycle_at_2.4GHz": lane_instr / (us * 1e-6 * 2.4e9), "peak_lane_instructions_per_cycle_one_cu": 64.0,
                                   "frac": lane_instr / (us * 1e-6 * 2.4e9) / 64.0,
                                   "note": "a batch ends with its longest search; its expansions run one after the other on one CU"}
print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""gpurun_out/prof_sq (tools/collect_sq_counters.sh) -> profiles/<tag>_sq_counters.json: averages per launch of the dominant
kernel and the derived shares.  python tools/summarize_sq.py <tag> [iterations_per_launch]"""
import collections, csv, glob, json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
its = float(sys.argv[2]) if len(sys.argv) > 2 else 20107.0
tot = collections.defaultdict(list)
for f in glob.glob("gpurun_out/prof_sq/pass*/bench_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "nmpc_ipm_" in r["Kernel_Name"]:
            tot[r["Counter_Name"]].append(float(r["Counter_Value"]))
d = {k: sum(v) / len(v) for k, v in tot.items()}
wc = d["SQ_WAVE_CYCLES"]
out = {"command": "tools/collect_sq_counters.sh (rocprofv3 --pmc <8 SQ counters per pass> --kernel-trace -- python bench.py --steps 6 --warmup 2 --no-cpu), averages per launch of the dominant kernel, B = 4096",
       "raw": d,
       "derived": {
           "instructions_per_ipm_iteration": d["SQ_INSTS"] / its,
           "share_valu": d["SQ_INSTS_VALU"] / d["SQ_INSTS"], "share_salu": d["SQ_INSTS_SALU"] / d["SQ_INSTS"],
           "share_lds": d["SQ_INSTS_LDS"] / d["SQ_INSTS"], "share_vmem": (d["SQ_INSTS_VMEM_RD"] + d["SQ_INSTS_VMEM_WR"]) / d["SQ_INSTS"],
           "vmem_instructions_per_iteration": (d["SQ_INSTS_VMEM_RD"] + d["SQ_INSTS_VMEM_WR"]) / its,
           "fp64_valu_per_iteration": (d["SQ_INSTS_VALU_FMA_F64"] + d["SQ_INSTS_VALU_ADD_F64"] + d["SQ_INSTS_VALU_MUL_F64"] + d["SQ_INSTS_VALU_TRANS_F64"]) / its,
           "mfma_per_iteration": d["SQ_INSTS_MFMA"] / its,
           "wave_time_parked_on_waitcnt_or_barrier": d["SQ_WAIT_ANY"] / wc,
           "wave_time_issue_stalled": d["SQ_WAIT_INST_ANY"] / wc,
           "wave_time_issuing": d["SQ_ACTIVE_INST_ANY"] / wc,
           "issue_quad_cycles_per_launch": d["SQ_ACTIVE_INST_ANY"],
           "lds_bank_conflict_share_of_lds_active": d["SQ_LDS_BANK_CONFLICT"] / d["SQ_ACTIVE_INST_LDS"],
           "note": "four waves per problem, three of them parked at a workgroup barrier while the fourth works: SQ_WAIT_ANY / SQ_WAVE_CYCLES "
                   "counts that parking and is not comparable with a one-wave-per-problem kernel; compare issue_quad_cycles_per_launch / kernel time"}}
json.dump(out, open(f"profiles/{tag}_sq_counters.json", "w"), indent=1)
print(json.dumps(out["derived"], indent=1))

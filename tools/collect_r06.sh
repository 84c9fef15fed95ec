#!/bin/bash
# GPU box: everything profiles/r06_* (final state of round 6) is made of.  Output under gpurun_out/r06c/ (+ gpurun_out/prof, prof_sq), summarised by
# tools/summarize_profiles.py r06_final gpurun_out/prof, tools/summarize_sq.py r06_final, tools/r06/kernel_trace_median.py in the build container.
#    gpurun -- 'bash tools/collect_r06.sh [quick]'
export TMPDIR=/tmp
ROOT=$PWD
R=$PWD/gpurun_out/r06c; rm -rf $R; mkdir -p $R
python bench.py --steps 20 --warmup 3 > $R/bench_default.json 2> $R/bench_default.err
python bench.py --steps 20 --warmup 3 --no-cpu --batch 16384 > $R/bench_16k.json 2>/dev/null
FRP_Q4=0 python bench.py --steps 20 --warmup 3 --no-cpu > $R/bench_three_per_cu.json 2>/dev/null
for c in "--config 3" "--config 3 --scaling strong" "--config 4" "--config 2 --scaling strong"; do
  FRP_BENCH_FORCE_DIST=$([[ "$c" == *strong* ]] && echo 1) python bench.py --steps 10 --warmup 2 --repeats 3 --no-cpu $c 2>/dev/null | tail -1 >> $R/other_configs.jsonl
done
python tools/bench_configs.py > $R/bench_configs.txt 2>&1
if [ -f forces_resilient_planner_amd/lib_prof.so ]; then
  for b in 1 4096; do FRP_Q4=0 FRP_LIB=$PWD/forces_resilient_planner_amd/lib_prof.so python tools/prof_lds.py $b 2 >> $R/wave_phases.txt 2>&1; done
  FRP_Q4_MIN_B=0 FRP_LIB=$PWD/forces_resilient_planner_amd/lib_prof.so python tools/prof_lds.py 1 2 >> $R/wave_phases.txt 2>&1
  FRP_LIB=$PWD/forces_resilient_planner_amd/lib_prof.so python tools/prof_lds.py 4096 2 >> $R/wave_phases.txt 2>&1
  FRP_LIB=$PWD/forces_resilient_planner_amd/lib_prof.so python tools/prof_lds.py 4096 3 >> $R/wave_phases.txt 2>&1
  FRP_LIB=$PWD/forces_resilient_planner_amd/lib_prof.so python tools/dbg/prof_rows.py >> $R/wave_phases.txt 2>&1
  FRP_LIB=$PWD/forces_resilient_planner_amd/lib_prof.so python tools/timeline.py 4096 > $R/timeline.txt 2>&1
fi
python tools/r06/dropin_lat.py > $R/dropin.txt 2>/dev/null
FRP_NMPC_TWIST=-1 python tools/r06/dropin_lat.py >> $R/dropin.txt 2>/dev/null
python tools/full_tick_bench.py 4096 10 20000 0.5 2 > $R/full_tick.json 2> $R/full_tick.err
python tools/dbg/tick_hist.py > $R/tick_rows.txt 2>/dev/null
python tools/receding_bench.py 65536 20 > $R/configs4_receding.json 2>/dev/null
if [ "$1" != "quick" ]; then
  python tests/tools/sweep_check.py > $R/sweep_check.txt 2>&1
  python tests/tools/soak.py 240 > $R/soak.txt 2>&1
  # where every flag mismatch of THIS soak parts (kind,B,seed,problem out of its own list)
  python - > $R/soak_specs.txt <<'PY'
import ast, sys
for l in open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r06c/soak.txt"):
    if l.startswith("exit-flag mismatch:"):
        d = ast.literal_eval(l.split(":", 1)[1].strip()); print("%d,%d,%d,%d" % (d["kind"], d["B"], d["seed"], d["problem"]))
PY
  python tests/tools/soak_diverge.py $(cat $R/soak_specs.txt) > $R/soak_diverge.txt 2>&1
fi
bash tools/collect_profiles.sh > $R/collect_profiles.log 2>&1
# the same command under --kernel-trace alone (no --stats post-processing): every dispatch's begin / end stamp -> median and mean of the solver kernel (tools/r06/kernel_trace_median.py)
P=$PWD/gpurun_out/prof/trace; rm -rf $P; mkdir -p $P
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $P -o bench -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu --streams 1 > $P/bench_under_trace.json 2> $P/trace.log)
python tools/r06/kernel_trace_median.py $P > $R/kernel_trace_median.json 2> $R/kernel_trace_median.err
bash tools/collect_sq_counters.sh > $R/collect_sq.log 2>&1
ls $R

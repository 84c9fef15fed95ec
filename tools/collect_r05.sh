#!/bin/bash
# GPU box: everything profiles/r05_* is made of.  Output under gpurun_out/r05c/ (+ gpurun_out/prof, prof_sq, prof_tick), summarised by
# tools/summarize_profiles.py / tools/summarize_sq.py / tools/tick_rooflines.py in the build container.
#    gpurun -- 'bash tools/collect_r05.sh [quick]'
export TMPDIR=/tmp
ROOT=$PWD
R=$PWD/gpurun_out/r05c; rm -rf $R; mkdir -p $R
python bench.py --steps 20 --warmup 3 > $R/bench_default.json 2> $R/bench_default.err
python bench.py --steps 20 --warmup 3 --no-cpu --batch 16384 > $R/bench_16k.json 2>/dev/null
FRP_Q4=0 python bench.py --steps 20 --warmup 3 --no-cpu > $R/bench_three_per_cu.json 2>/dev/null
FRP_Q4=0 python bench.py --steps 20 --warmup 3 --no-cpu --batch 16384 > $R/bench_three_per_cu_16k.json 2>/dev/null
FRP_ISO_IT=0 python bench.py --steps 20 --warmup 3 --no-cpu > $R/bench_no_isolation.json 2>/dev/null
FRP_LIB=$PWD/forces_resilient_planner_amd/lib_defaultflags.so python bench.py --steps 20 --warmup 3 --no-cpu > $R/bench_defaultflags.json 2>/dev/null
for c in "--config 3" "--config 3 --scaling strong" "--config 4" "--config 2 --scaling strong"; do
  FRP_BENCH_FORCE_DIST=$([[ "$c" == *strong* ]] && echo 1) python bench.py --steps 10 --warmup 2 --repeats 3 --no-cpu $c 2>/dev/null | tail -1 >> $R/other_configs.jsonl
done
python tools/bench_configs.py > $R/bench_configs.txt 2>&1
if [ -f forces_resilient_planner_amd/lib_prof.so ]; then
  for b in 1 4096; do FRP_Q4=0 FRP_LIB=$PWD/forces_resilient_planner_amd/lib_prof.so python tools/prof_lds.py $b 2 >> $R/wave_phases.txt 2>&1; done
  FRP_Q4_MIN_B=0 FRP_LIB=$PWD/forces_resilient_planner_amd/lib_prof.so python tools/prof_lds.py 1 2 >> $R/wave_phases.txt 2>&1
  FRP_LIB=$PWD/forces_resilient_planner_amd/lib_prof.so python tools/prof_lds.py 4096 2 >> $R/wave_phases.txt 2>&1
  FRP_LIB=$PWD/forces_resilient_planner_amd/lib_prof.so python tools/timeline.py 4096 > $R/timeline.txt 2>&1
  FRP_ISO_IT=0 FRP_LIB=$PWD/forces_resilient_planner_amd/lib_prof.so python tools/timeline.py 4096 >> $R/timeline.txt 2>&1
fi
python tools/twist_latency.py 2>&1 | grep -v amdgpu.ids > $R/twist_latency.txt
python tests/tools/e2e_bench.py > $R/e2e.txt 2>/dev/null
python tools/full_tick_bench.py 4096 10 20000 0.5 2 > $R/full_tick.json 2> $R/full_tick.err
python tools/receding_bench.py 65536 20 > $R/configs4_receding.json 2>/dev/null
python tests/tools/astar_bench.py 1024 pillars 20000 2>/dev/null | tail -1 > $R/astar_bench.jsonl
python tests/tools/astar_bench.py 1024 wall_gap 20000 2>/dev/null | tail -1 >> $R/astar_bench.jsonl
python tests/tools/corridor_bench.py 4096 20000 0.5 2>/dev/null | tail -1 > $R/corridor_bench.jsonl
python tests/tools/corridor_bench.py 4096 62000 0.5 2>/dev/null | tail -1 >> $R/corridor_bench.jsonl
if [ "$1" != "quick" ]; then
  python tests/tools/sweep_check.py > $R/sweep_check.txt 2>&1
  python tests/tools/soak.py 240 > $R/soak.txt 2>&1
fi
# the whole tick under rocprofv3 (kernel trace + stats): per-kernel durations for tools/tick_rooflines.py
P=$PWD/gpurun_out/prof_tick; rm -rf $P; mkdir -p $P
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -o tick -- python $ROOT/tools/full_tick_bench.py 4096 10 20000 0.5 0 > $P/full_tick_under_rocprof.json 2> $P/stats.log)
bash tools/collect_profiles.sh > $R/collect_profiles.log 2>&1
bash tools/collect_sq_counters.sh > $R/collect_sq.log 2>&1
ls $R

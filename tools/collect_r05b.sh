#!/bin/bash
# GPU box: what changed after tools/collect_r05.sh ran (the corridor kernel of round 5, the A* default): corridor bench, tick, its kernel
# stats under rocprofv3, the corridor soak, the A* bench, the corridor kernel's phase profile, the default bench line.
#    gpurun -- 'bash tools/collect_r05b.sh'        -> gpurun_out/r05b/
export TMPDIR=/tmp
ROOT=$PWD
R=$PWD/gpurun_out/r05b; rm -rf $R; mkdir -p $R
python tests/tools/corridor_bench.py 4096 20000 0.5 2>/dev/null | tail -1 > $R/corridor_bench.jsonl
python tests/tools/corridor_bench.py 4096 62000 0.5 2>/dev/null | tail -1 >> $R/corridor_bench.jsonl
FRP_CORRIDOR_WAVE=0 python tests/tools/corridor_bench.py 4096 20000 0.5 2>/dev/null | tail -1 >> $R/corridor_bench.jsonl
FRP_CORRIDOR_WAVE=0 python tests/tools/corridor_bench.py 4096 62000 0.5 2>/dev/null | tail -1 >> $R/corridor_bench.jsonl
python tools/full_tick_bench.py 4096 10 20000 0.5 2 > $R/full_tick.json 2> $R/full_tick.err
python tests/tools/astar_bench.py 1024 pillars 20000 2>/dev/null | tail -1 > $R/astar_bench.jsonl
python tests/tools/astar_bench.py 1024 wall_gap 20000 2>/dev/null | tail -1 >> $R/astar_bench.jsonl
python tests/tools/soak_corridor.py 100 2>/dev/null | tail -1 > $R/soak_corridor.txt
if [ -f forces_resilient_planner_amd/lib_crprof.so ]; then
  for P in 20000 62000; do echo "# corridor_bench.py 4096 $P 0.5, -DFRP_CORRIDOR_PROFILE build" >> $R/corridor_phases.txt
    FRP_LIB=$PWD/forces_resilient_planner_amd/lib_crprof.so python tests/tools/corridor_bench.py 4096 $P 0.5 2>&1 | grep "^wave" | head -2 >> $R/corridor_phases.txt; done
  echo "# full_tick_bench.py 4096 3 20000 0.5 0 (last tick)" >> $R/corridor_phases.txt
  FRP_LIB=$PWD/forces_resilient_planner_amd/lib_crprof.so python tools/full_tick_bench.py 4096 3 20000 0.5 0 2>&1 | grep "^wave" | tail -2 >> $R/corridor_phases.txt
fi
python tools/graph_tick.py 1 200 5000 2>/dev/null | tail -1 > $R/graph_tick.jsonl; python tools/graph_tick.py 64 200 5000 2>/dev/null | tail -1 >> $R/graph_tick.jsonl
python tools/twist_latency.py dropin 2>/dev/null | tail -1 > $R/dropin.txt; FRP_NMPC_DROPIN_SPIN=0 python tools/twist_latency.py dropin 2>/dev/null | tail -1 >> $R/dropin.txt; FRP_NMPC_TWIST=-1 python tools/twist_latency.py dropin 2>/dev/null | tail -1 >> $R/dropin.txt
python bench.py --steps 20 --warmup 3 > $R/bench_default.json 2> $R/bench_default.err
P=$PWD/gpurun_out/prof_tick; rm -rf $P; mkdir -p $P
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -o tick -- python $ROOT/tools/full_tick_bench.py 4096 10 20000 0.5 0 > $P/full_tick_under_rocprof.json 2> $P/stats.log)
cp $(find $P/stats -name "*kernel_stats.csv" | head -1) $R/full_tick_kernel_stats.csv 2>/dev/null
P2=$PWD/gpurun_out/prof_corridor; rm -rf $P2; mkdir -p $P2
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $P2/stats -o cor -- python $ROOT/tests/tools/corridor_bench.py 4096 20000 0.5 > $P2/out.json 2> $P2/stats.log)
cp $(find $P2/stats -name "*kernel_stats.csv" | head -1) $R/corridor_bench_kernel_stats.csv 2>/dev/null
ls -la $R

#!/bin/bash
# GPU box, round 6 final with the Q30 variant: -m gpu suite + smoke, configs[3] / other configs, wave phases of configs[3] on both variants, sweep check, a short soak
export TMPDIR=/tmp
mkdir -p gpurun_out
R=gpurun_out/r06d; rm -rf $R; mkdir -p $R
P=$PWD/forces_resilient_planner_amd
( timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -3; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ) > $R/gputests.txt
for c in "--config 3" "--config 3 --scaling strong" "--config 4" "--config 2 --scaling strong"; do
  FRP_BENCH_FORCE_DIST=$([[ "$c" == *strong* ]] && echo 1) timeout 300 python bench.py --steps 10 --warmup 2 --repeats 3 --no-cpu $c 2>/dev/null | tail -1 >> $R/other_configs.jsonl
done
FRP_Q30=0 timeout 300 python bench.py --config 3 --steps 10 --warmup 2 --repeats 3 --no-cpu 2>/dev/null | tail -1 > $R/config3_two_per_cu.json
timeout 300 python tools/bench_configs.py > $R/bench_configs.txt 2>&1
FRP_LIB=$P/lib_prof.so timeout 300 python tools/prof_lds.py 4096 3 > $R/wave_phases_config3.txt 2>&1
FRP_LIB=$P/lib_prof.so FRP_Q30_MIN_B=0 timeout 300 python tools/prof_lds.py 1 3 >> $R/wave_phases_config3.txt 2>&1
FRP_LIB=$P/lib_prof.so FRP_Q30=0 timeout 300 python tools/prof_lds.py 4096 3 >> $R/wave_phases_config3.txt 2>&1
timeout 900 python tests/tools/sweep_check.py > $R/sweep_check.txt 2>&1
timeout 900 python tests/tools/soak.py 180 > $R/soak.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 3 > $R/bench_default.json 2> $R/bench_default.err
cat $R/gputests.txt; tail -n 3 $R/soak.txt | cut -c1-300

#!/bin/bash
# GPU box: everything profiles/r02_* is made of.  Output under gpurun_out/r02/ (+ gpurun_out/prof, prof_sq), summarised by
# tools/summarize_profiles.py / tools/summarize_sq.py here in the build container.
export TMPDIR=/tmp
R=$PWD/gpurun_out/r02; rm -rf $R; mkdir -p $R
python bench.py --steps 20 --warmup 3 > $R/bench_default.json 2> $R/bench_default.err
# same-box A/B against the round-1 kernel (single wave per problem, HBM workspace)
python bench.py --steps 20 --warmup 3 --no-cpu > $R/ab_lds.json 2>/dev/null
FRP_KERNEL=r01 python bench.py --steps 20 --warmup 3 --no-cpu > $R/ab_r01.json 2>/dev/null
python bench.py --steps 20 --warmup 3 --no-cpu --batch 16384 > $R/ab_lds_16k.json 2>/dev/null
FRP_KERNEL=r01 python bench.py --steps 20 --warmup 3 --no-cpu --batch 16384 > $R/ab_r01_16k.json 2>/dev/null
# the other BASELINE configs and the strong-scaling form
for c in "--config 3" "--config 3 --scaling strong" "--config 4" "--config 2 --scaling strong"; do
  python bench.py --steps 10 --warmup 2 --repeats 3 --no-cpu $c 2>/dev/null | tail -1 >> $R/other_configs.jsonl
done
FRP_KERNEL=r01 python bench.py --steps 10 --warmup 2 --repeats 3 --no-cpu --config 3 2>/dev/null | tail -1 > $R/r01_config3.json
FRP_KERNEL=r01 python bench.py --steps 10 --warmup 2 --repeats 3 --no-cpu --config 4 2>/dev/null | tail -1 > $R/r01_config4.json
python tools/bench_configs.py > $R/bench_configs.txt 2>&1
# per-wave phase cycles
if [ -f forces_resilient_planner_amd/lib_prof.so ]; then
  for b in 1 4096; do FRP_LIB=forces_resilient_planner_amd/lib_prof.so python tools/prof_lds.py $b 2 >> $R/wave_phases.txt 2>&1; done
fi
timeout 60 tools/ubench/sweep_timing > $R/sweep_timing.txt 2>&1
tools/ubench/rcp_f64 > $R/rcp_f64.txt 2>&1
(echo "== chain_lat"; timeout 60 tools/ubench/chain_lat; echo "== issue_rate"; timeout 60 tools/ubench/issue_rate; echo "== fwd_model"; timeout 60 tools/ubench/fwd_model; for n in 4 20; do echo "== sweep_timing N=$n"; tools/ubench/sweep_timing $n; done) > $R/issue_model.txt 2>&1
python tools/stage_eval_bench.py > $R/stage_eval.json 2>/dev/null
python tools/full_tick_bench.py 4096 10 20000 > $R/full_tick.json 2> $R/full_tick.err
python tools/receding_bench.py 65536 20 > $R/configs4_receding.json 2>/dev/null
bash tools/collect_profiles.sh > $R/collect_profiles.log 2>&1
bash tools/collect_sq_counters.sh > $R/collect_sq.log 2>&1
# rocprof kernel stats of the round-1 kernel on the same box
cd /tmp && FRP_KERNEL=r01 rocprofv3 --kernel-trace --stats --output-format csv -d $R/r01_stats -o bench -- python $OLDPWD/bench.py --steps 20 --warmup 3 --no-cpu > /dev/null 2> $R/r01_stats.log; cd $OLDPWD
ls $R

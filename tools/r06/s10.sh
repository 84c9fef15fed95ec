#!/bin/bash
# GPU box, round 6: the four-per-CU form for stages with up to 30 corridor rows (the tick's solver variant): -DFRP_Q4_MORE_ROWS build of the Q4 unit, FRP_Q4_MAXF=30
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r06_s10.txt; : > $O
P=$PWD/forces_resilient_planner_amd
echo "== parity with the rows on the four-per-CU variants (every covered launch)" >> $O
( FRP_LIB=$P/lib_q4rows.so FRP_Q4_MAXF=30 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -3 ) >> $O
for rep in 1 2; do
for mf in 6 30; do echo -n "FRP_Q4_MAXF=$mf tick: " >> $O; FRP_LIB=$P/lib_q4rows.so FRP_Q4_MAXF=$mf timeout 300 python tools/full_tick_bench.py 4096 10 20000 0.5 0 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_tick'],4), {k:round(v,4) for k,v in j['ms_per_step'].items()}, j['mean_iters'], j['converged_frac'])" >> $O; done
done
cat $O

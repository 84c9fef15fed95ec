#!/bin/bash
# GPU box, round 6 session 3: head start (the longest expected solves get a CU each from the start) on/off/K; FRP_EARLY_FACTOR with LDS polls;
# the time line of the launch (profile build).
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r06_s3.txt; : > $O
P=$PWD/forces_resilient_planner_amd
echo "== parity: -m gpu subset on the product (head start on)" >> $O
( timeout 900 python -m pytest tests -q -m gpu -x -k "oracle or order or queue or fixtures or hard or soak or variant" 2>&1 | tail -3 ) >> $O
bl() { python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ms/step %.4f kernel_ms %.4f value %.0f frac %.4f its %.3f conv %.4f' % (j['ms_per_step'], j['roofline']['kernel_ms'], j['value'], j['roofline']['frac'], j['config']['mean_ipm_iterations'], j['config']['converged_frac']))"; }
for rep in 1 2; do
for hs in 0 8 16 32; do
  for B in 4096 16384; do
    echo -n "head_start=$hs B=$B: " >> $O
    FRP_HEAD_START=$hs timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --repeats 5 --batch $B 2>/dev/null | tail -1 | bl >> $O
  done
done
for B in 4096 16384; do
  echo -n "early(+head 16) B=$B: " >> $O
  FRP_LIB=$P/lib_early.so timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --repeats 5 --batch $B 2>/dev/null | tail -1 | bl >> $O
  echo -n "early head_start=0 B=$B: " >> $O
  FRP_HEAD_START=0 FRP_LIB=$P/lib_early.so timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --repeats 5 --batch $B 2>/dev/null | tail -1 | bl >> $O
done
done
echo "== other seeds (config2 with another seed through tools/bench_configs?)" >> $O
echo "== config 4 / config 1 with head start on / off" >> $O
for hs in 0 16; do
  echo -n "head_start=$hs config 4: " >> $O
  FRP_HEAD_START=$hs timeout 300 python bench.py --config 4 --steps 10 --warmup 2 --no-cpu --repeats 3 2>/dev/null | tail -1 | bl >> $O
done
echo "== time line (profile build), head start on / off" >> $O
for hs in 16 0; do echo "head_start=$hs" >> $O; FRP_HEAD_START=$hs FRP_LIB=$P/lib_prof.so timeout 300 python tools/timeline.py 2>/dev/null | head -14 >> $O; done
cat $O

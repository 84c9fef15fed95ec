#!/bin/bash
# GPU box, round 6: the Q30 variant's knobs (rows re-read, bound rounds split, inlined factorisation) on configs[3]; the full -m gpu suite on the product; wave phases
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r06_s8.txt; : > $O
P=$PWD/forces_resilient_planner_amd
( timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 ) >> $O
bl() { python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ms/step %.4f kernel_ms %.4f value %.0f frac %.4f its %.3f conv %.4f' % (j['ms_per_step'], j['roofline']['kernel_ms'], j['value'], j['roofline']['frac'], j['config']['mean_ipm_iterations'], j['config']['converged_frac']))"; }
for rep in 1 2; do
for lib in libfrp_nmpc_amd lib_q30a lib_q30b lib_q30c lib_q30d lib_q30e lib_q30f; do
  echo -n "$lib config 3: " >> $O
  FRP_LIB=$P/$lib.so timeout 300 python bench.py --config 3 --steps 5 --warmup 1 --no-cpu --repeats 3 2>/dev/null | tail -1 | bl >> $O
done
done
echo -n "config 3 strong (RCCL world 1): " >> $O
FRP_BENCH_FORCE_DIST=1 timeout 300 python bench.py --config 3 --scaling strong --steps 5 --warmup 1 --no-cpu --repeats 3 2>/dev/null | tail -1 | bl >> $O
for B in 512 1024 2048 4096; do for q in 0 1; do echo -n "FRP_Q30=$q config 3 B=$B: " >> $O; FRP_Q30_MIN_B=0 FRP_Q30=$q timeout 300 python bench.py --config 3 --batch $B --steps 10 --warmup 2 --no-cpu --repeats 3 2>/dev/null | tail -1 | bl >> $O; done; done
cat $O

#!/bin/bash
# GPU box, round 6 session 2: FRP_EARLY_FACTOR (the Riccati wave starts the factorisation before the termination test is known) --
# the -m gpu suite on that build, then same-box A/B: B = 4096 / 16384 (four per CU), FRP_Q4=0 (three per CU), configs[3], the drop-in call.
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r06_s2.txt; : > $O
P=$PWD/forces_resilient_planner_amd
echo "== -m gpu suite on lib_early" >> $O
( FRP_LIB=$P/lib_early.so timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 ) >> $O
bl() { python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ms/step %.4f kernel_ms %.4f value %.0f frac %.4f its %.3f conv %.4f' % (j['ms_per_step'], j['roofline']['kernel_ms'], j['value'], j['roofline']['frac'], j['config']['mean_ipm_iterations'], j['config']['converged_frac']))"; }
for rep in 1 2; do
for lib in libfrp_nmpc_amd lib_early; do
  for B in 4096 16384; do
    echo -n "$lib B=$B: " >> $O
    FRP_LIB=$P/$lib.so timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --repeats 5 --batch $B 2>/dev/null | tail -1 | bl >> $O
  done
  echo -n "$lib FRP_Q4=0 B=4096: " >> $O
  FRP_Q4=0 FRP_LIB=$P/$lib.so timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --repeats 5 2>/dev/null | tail -1 | bl >> $O
  echo -n "$lib config 3: " >> $O
  FRP_LIB=$P/$lib.so timeout 300 python bench.py --config 3 --steps 5 --warmup 1 --no-cpu --repeats 3 2>/dev/null | tail -1 | bl >> $O
  echo -n "$lib " >> $O; FRP_LIB=$P/$lib.so timeout 120 python tools/r06/dropin_lat.py 2>/dev/null >> $O
  echo -n "$lib twist " >> $O; FRP_NMPC_TWIST=-1 FRP_LIB=$P/$lib.so timeout 120 python tools/r06/dropin_lat.py 2>/dev/null >> $O
done
done
echo "== full tick" >> $O
for lib in libfrp_nmpc_amd lib_early; do echo -n "$lib " >> $O; FRP_LIB=$P/$lib.so timeout 300 python tools/full_tick_bench.py 4096 10 20000 0.5 0 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_tick'], j['ms_per_step'], j['mean_iters'])" >> $O; done
cat $O

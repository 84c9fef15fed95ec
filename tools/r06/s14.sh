#!/bin/bash
# GPU box, round 6: what one trip to the L2 workspace costs under load (the Riccati wave's S_xx fetch at the head of the step phase, timed in the profile build) and
# whether a touch of the same lines by the idle bounds wave (-DFRP_QP_TOUCH=1) shortens it; then the unprofiled A/B.
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
O=gpurun_out/r06/touch.txt; : > $O
P=$PWD/forces_resilient_planner_amd
for lib in lib_prof lib_proftouch; do  # (both with FRP_Q4_PARK)
  echo "== $lib" >> $O
  FRP_Q4_MIN_B=0 FRP_LIB=$P/$lib.so timeout 300 python tools/prof_lds.py 1 2 >> $O 2>&1
  FRP_LIB=$P/$lib.so timeout 300 python tools/prof_lds.py 4096 2 >> $O 2>&1
done
bl() { python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ms/step %.4f kernel_ms %.4f value %.0f frac %.4f' % (j['ms_per_step'], j['roofline']['kernel_ms'], j['value'], j['roofline']['frac']))"; }
for rep in 1 2; do
for lib in libfrp_nmpc_amd lib_touch; do
  for B in 4096 16384; do
    echo -n "$lib B=$B: " >> $O
    FRP_LIB=$P/$lib.so timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --repeats 5 --batch $B 2>/dev/null | tail -1 | bl >> $O
  done
done
done
cat $O | cut -c1-330

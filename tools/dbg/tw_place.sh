#!/bin/bash
export TMPDIR=/tmp
for L in main twplace; do
  if [ $L = main ]; then LIB=$PWD/forces_resilient_planner_amd/libfrp_nmpc_amd.so; else LIB=$PWD/forces_resilient_planner_amd/lib_$L.so; fi
  for t in 0 3 4 5 6 8 9; do
    FRP_LIB=$LIB python bench.py --steps 20 --warmup 3 --no-cpu --repeats 3 --twist $t 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$L twist $t', 'ms/step %.4f kernel_ms %.4f' % (j['ms_per_step'], j['roofline']['kernel_ms']))"
  done
done

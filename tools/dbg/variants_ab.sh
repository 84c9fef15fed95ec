#!/bin/bash
# GPU box: bench product vs variant libs at B = 4096 and 16384 (two rounds)
export TMPDIR=/tmp
for r in 1 2; do
for v in main "$@"; do
  if [ $v = main ]; then L=$PWD/forces_resilient_planner_amd/libfrp_nmpc_amd.so; else L=$PWD/forces_resilient_planner_amd/lib_$v.so; fi
  for B in 4096 16384; do
    echo -n "$v B=$B: "
    FRP_LIB=$L timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --repeats 3 --batch $B 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('ms/step %.4f kernel_ms %.4f value %.0f' % (j['ms_per_step'], j['roofline']['kernel_ms'], j['value']))"
  done
done
done

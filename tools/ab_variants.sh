#!/bin/bash
# GPU box: bench every forces_resilient_planner_amd/lib_<name>.so given on the command line against the product build (two rounds)
export TMPDIR=/tmp
for i in 1 2; do
  for n in main "$@"; do
    if [ $n = main ]; then L=$PWD/forces_resilient_planner_amd/libfrp_nmpc_amd.so; else L=$PWD/forces_resilient_planner_amd/lib_$n.so; fi
    FRP_LIB=$L python bench.py --steps 20 --warmup 3 --no-cpu --repeats 3 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$n', 'ms/step %.4f kernel_ms %.4f pipelined %.0f' % (j['ms_per_step'], j['roofline']['kernel_ms'], j['config']['pipelined_solves_per_s'] or 0))"
  done
done

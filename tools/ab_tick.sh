#!/bin/bash
# GPU box: full-tick bench with every forces_resilient_planner_amd/lib_<name>.so given on the command line and the product build
export TMPDIR=/tmp
for n in main "$@"; do
  if [ $n = main ]; then L=$PWD/forces_resilient_planner_amd/libfrp_nmpc_amd.so; else L=$PWD/forces_resilient_planner_amd/lib_$n.so; fi
  FRP_LIB=$L python tools/full_tick_bench.py 4096 10 20000 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n', 'tick %.3f ms' % j['ms_per_tick'], {k: round(v,3) for k,v in j['ms_per_step'].items()})"
done

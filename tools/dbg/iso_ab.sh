#!/bin/bash
# GPU box: the long-solve isolation of the Q4 variants: off / thresholds, B = 4096 and 16384; time line
export TMPDIR=/tmp
for iso in 0 6 8 10 12; do
  for B in 4096 16384; do
    echo -n "FRP_ISO_IT=$iso B=$B: "
    FRP_ISO_IT=$iso timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --repeats 3 --batch $B 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('ms/step %.4f kernel_ms %.4f value %.0f' % (j['ms_per_step'], j['roofline']['kernel_ms'], j['value']))"
  done
done
FRP_LIB=$PWD/forces_resilient_planner_amd/lib_prof.so python tools/timeline.py 4096 2>/dev/null
python tools/dbg/q4_iter.py 2 64 2>/dev/null | grep "maxit 200"

#!/bin/bash
# GPU box: A/B of the three-per-CU variants (FRP_Q4=0) against the four-per-CU ones at B = 4096 and 16384, then the wave phases of the latter
export TMPDIR=/tmp
for q in 0 1; do
  for B in 4096 16384; do
    echo -n "FRP_Q4=$q B=$B: "
    FRP_Q4=$q timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --repeats 3 --batch $B 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('ms/step %.4f kernel_ms %.4f value %.0f' % (j['ms_per_step'], j['roofline']['kernel_ms'], j['value']))"
  done
done
for B in 1 4096; do FRP_LIB=$PWD/forces_resilient_planner_amd/lib_prof.so python tools/prof_lds.py $B 2 2>/dev/null | grep -v "segments\|whole sweeps"; done

#!/bin/bash
# GPU box, round 6: A/B of experiment libraries against the product build: bash tools/r06/s16.sh lib_a lib_b ...   (two rounds, B = 4096 and 16384)
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
O=gpurun_out/r06/ab_$(echo "$@" | tr ' ' '_').txt; : > $O
P=$PWD/forces_resilient_planner_amd
bl() { python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ms/step %.4f kernel_ms %.4f value %.0f frac %.4f' % (j['ms_per_step'], j['roofline']['kernel_ms'], j['value'], j['roofline']['frac']))"; }
for rep in 1 2; do
for lib in "$@"; do
  for B in 4096 16384; do
    echo -n "$lib B=$B: " >> $O
    FRP_LIB=$P/$lib.so timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --repeats 5 --batch $B 2>/dev/null | tail -1 | bl >> $O
  done
done
done
cat $O

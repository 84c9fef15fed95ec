#!/bin/bash
# GPU box, round 6: the corridor lanes' parking in LDS (FRP_Q4_PARK, default 1) against the build without it (lib_nopark / lib_profnopark): wave phases, then the
# unprofiled A/B, then the parity tests that run on the four-per-CU variant.
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
O=gpurun_out/r06/park.txt; : > $O
P=$PWD/forces_resilient_planner_amd
for lib in lib_profnopark lib_prof; do
  echo "== $lib" >> $O
  FRP_Q4_MIN_B=0 FRP_LIB=$P/$lib.so timeout 300 python tools/prof_lds.py 1 2 >> $O 2>&1
  FRP_LIB=$P/$lib.so timeout 300 python tools/prof_lds.py 4096 2 >> $O 2>&1
done
bl() { python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ms/step %.4f kernel_ms %.4f value %.0f frac %.4f' % (j['ms_per_step'], j['roofline']['kernel_ms'], j['value'], j['roofline']['frac']))"; }
for rep in 1 2; do
for lib in lib_nopark libfrp_nmpc_amd; do
  for B in 4096 16384; do
    echo -n "$lib B=$B: " >> $O
    FRP_LIB=$P/$lib.so timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --repeats 5 --batch $B 2>/dev/null | tail -1 | bl >> $O
  done
done
done
( timeout 1200 python -m pytest tests -q -m gpu -x -k "oracle or order or queue or fixtures or hard or variant or barrier" 2>&1 | tail -3 ) >> $O
cat $O | cut -c1-330

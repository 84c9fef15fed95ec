#!/bin/bash
# GPU box, round 6 session 4: staggered first round (FRP_STAGGER = 1, 2, 3 x 8 k cycles per arrival index) against the product, head start off
export TMPDIR=/tmp FRP_HEAD_START=0
mkdir -p gpurun_out
O=gpurun_out/r06_s4.txt; : > $O
P=$PWD/forces_resilient_planner_amd
bl() { python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ms/step %.4f kernel_ms %.4f value %.0f frac %.4f' % (j['ms_per_step'], j['roofline']['kernel_ms'], j['value'], j['roofline']['frac']))"; }
for rep in 1 2; do
for lib in libfrp_nmpc_amd lib_stag1 lib_stag2 lib_stag3; do
  for B in 4096 16384; do
    echo -n "$lib B=$B: " >> $O
    FRP_LIB=$P/$lib.so timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --repeats 5 --batch $B 2>/dev/null | tail -1 | bl >> $O
  done
done
done
for iso in 6 8 16; do
  echo -n "FRP_ISO_IT=$iso B=4096: " >> $O
  FRP_ISO_IT=$iso timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --repeats 5 2>/dev/null | tail -1 | bl >> $O
done
cat $O

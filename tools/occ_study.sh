#!/bin/bash
run() { python bench.py --steps 10 --warmup 2 --no-cpu --batch $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  B=$1 slots=${FRP_RESIDENT_SLOTS:-default}: solves/s %.0f kernel_ms %.3f' % (d['value'], d['roofline']['kernel_ms']))"; }
cp forces_resilient_planner_amd/lib_main.so forces_resilient_planner_amd/libfrp_nmpc_amd.so
for B in 1024 2048 4096 8192 16384 32768; do run $B; done
for S in 1280 1792; do FRP_RESIDENT_SLOTS=$S run 32768; done
for S in 1024 1152 1280 1408; do FRP_RESIDENT_SLOTS=$S run 4096; done

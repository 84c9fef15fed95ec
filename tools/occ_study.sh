#!/bin/bash
run() { python bench.py --steps 10 --warmup 2 --no-cpu --batch $1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  B=$1 slots=${FRP_RESIDENT_SLOTS:-default}: solves/s %.0f kernel_ms %.3f' % (d['value'], d['roofline']['kernel_ms']))"; }
cp forces_resilient_planner_amd/lib_main.so forces_resilient_planner_amd/libfrp_nmpc_amd.so
echo main; for B in 256 512 1024 2048 4096; do run $B; done
for S in 512 1024 1536 2048 3072; do FRP_RESIDENT_SLOTS=$S run 4096; done
echo w1; cp forces_resilient_planner_amd/lib_w1.so forces_resilient_planner_amd/libfrp_nmpc_amd.so
for S in 512 1024; do FRP_RESIDENT_SLOTS=$S run 4096; done
cp forces_resilient_planner_amd/lib_main.so forces_resilient_planner_amd/libfrp_nmpc_amd.so
rocm-smi --showclocks 2>/dev/null | grep -iE "sclk|mclk" | head -4

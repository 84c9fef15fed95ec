#!/bin/bash
# resident-slot sweep for a library variant: tools/occ_study.sh lib_x.so
cp forces_resilient_planner_amd/$1 forces_resilient_planner_amd/libfrp_nmpc_amd.so
run() { python bench.py --steps 20 --warmup 3 --no-cpu --batch $1 --streams 2 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  B=$1 slots=${FRP_RESIDENT_SLOTS:-default}: %.0f solves/s (serial %.0f)' % (d['value'], d['config']['single_stream_solves_per_s']))"; }
for S in 1536 1792 2048; do FRP_RESIDENT_SLOTS=$S run 16384; done
cp forces_resilient_planner_amd/lib_main.so forces_resilient_planner_amd/libfrp_nmpc_amd.so

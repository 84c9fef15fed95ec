#!/bin/bash
# resident-slot sweep (2 streams like bench.py): tools/occ_study.sh
run() { python bench.py --steps 20 --warmup 3 --no-cpu --batch $1 --streams 2 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  B=$1 slots=${FRP_RESIDENT_SLOTS:-default}: %.0f solves/s (serial %.0f)' % (d['value'], d['config']['single_stream_solves_per_s']))"; }
for S in 1024 1280 1536 1792; do FRP_RESIDENT_SLOTS=$S run 4096; done
for S in 1280 1536 1792; do FRP_RESIDENT_SLOTS=$S run 16384; done

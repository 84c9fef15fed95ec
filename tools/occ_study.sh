#!/bin/bash
# resident-slot sweep, serial (1 stream) and pipelined (2 streams) launches
run() { python bench.py --steps 20 --warmup 3 --no-cpu --batch $1 --streams $2 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  B=$1 streams=$2 slots=${FRP_RESIDENT_SLOTS:-default}: %.0f solves/s (serial %.0f)' % (d['value'], d['config']['single_stream_solves_per_s']))"; }
for S in 768 1024 1280 1536 2048; do FRP_RESIDENT_SLOTS=$S run 4096 2; done
for S in 1024 1536; do FRP_RESIDENT_SLOTS=$S run 16384 2; done

#!/bin/bash
for v in "$@"; do
  cp forces_resilient_planner_amd/$v forces_resilient_planner_amd/libfrp_nmpc_amd.so
  echo "=== $v"
  python bench.py --steps 10 --warmup 2 --no-cpu | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('solves/s %.0f  ms %.3f  kernel_ms %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"
  python bench.py --steps 10 --warmup 2 --no-cpu --batch 16384 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B=16384: solves/s %.0f  ms %.3f' % (d['value'], d['ms_per_step']))"
done

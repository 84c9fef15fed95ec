#!/bin/bash
export TMPDIR=/tmp
for i in 1 2; do
for t in 0 -1 8 9 11 12; do
  python bench.py --steps 20 --warmup 3 --no-cpu --repeats 3 --twist $t 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('twist $t', 'ms/step %.4f kernel_ms %.4f conv %.4f mean_it %.3f' % (j['ms_per_step'], j['roofline']['kernel_ms'], j['config']['converged_frac'], j['config']['mean_ipm_iterations']))"
done; done

#!/bin/bash
export TMPDIR=/tmp
python tools/dbg/twist_check.py 64 -1 2>&1 | grep -v amdgpu.ids | grep -v "per-"
FRP_LIB=$PWD/forces_resilient_planner_amd/lib_prof.so python tools/prof_lds.py 4096 2 10 2>&1 | grep -v amdgpu.ids
for i in 1 2; do
for t in 0 -1 8 12; do
  python bench.py --steps 20 --warmup 3 --no-cpu --repeats 3 --twist $t 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('twist $t', 'ms/step %.4f kernel_ms %.4f conv %.4f mean_it %.3f' % (j['ms_per_step'], j['roofline']['kernel_ms'], j['config']['converged_frac'], j['config']['mean_ipm_iterations']))"
done; done

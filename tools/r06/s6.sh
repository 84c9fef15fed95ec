#!/bin/bash
# GPU box, round 6: the proxy for configs[3] at three problems per CU -- the N <= 32 variant compiled at the 168-register cap of three waves per SIMD
# (dynamic LDS so that the compiler keeps the cap), still run at two per CU: what does the register cap cost an iteration?
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r06_s6.txt; : > $O
P=$PWD/forces_resilient_planner_amd
bl() { python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ms/step %.4f kernel_ms %.4f value %.0f frac %.4f its %.3f conv %.4f' % (j['ms_per_step'], j['roofline']['kernel_ms'], j['value'], j['roofline']['frac'], j['config']['mean_ipm_iterations'], j['config']['converged_frac']))"; }
for rep in 1 2; do
for lib in libfrp_nmpc_amd lib_n32e lib_n32a lib_n32b lib_n32c lib_n32d; do
  echo -n "$lib config 3: " >> $O
  FRP_LIB=$P/$lib.so timeout 300 python bench.py --config 3 --steps 5 --warmup 1 --no-cpu --repeats 3 2>/dev/null | tail -1 | bl >> $O
done
done
echo "== parity of the 168-register builds (N = 30 tests)" >> $O
for lib in lib_n32c; do ( FRP_LIB=$P/$lib.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "variant or batch_matches_oracle" 2>&1 | tail -2 ) >> $O; done
cat $O

#!/bin/bash
# GPU box, round 6: the Q30 variant (N <= 30 at three problems per CU) -- iterate-level agreement with the oracle, parity subset, bench against the two-per-CU variants
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r06_s7.txt; : > $O
echo "== per-iteration agreement, configs[3], B = 64, everything on the Q30 variant" >> $O
FRP_Q30_MIN_B=0 timeout 300 python tools/dbg/q4_iter.py 3 64 >> $O 2>&1
echo "== the same on the two-per-CU variants" >> $O
FRP_Q30=0 timeout 300 python tools/dbg/q4_iter.py 3 64 >> $O 2>&1
echo "== parity subset with every covered launch on Q30" >> $O
( FRP_Q30_MIN_B=0 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "variant or batch_matches_oracle or fixtures or horizon" 2>&1 | tail -4 ) >> $O
bl() { python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ms/step %.4f kernel_ms %.4f value %.0f frac %.4f its %.3f conv %.4f' % (j['ms_per_step'], j['roofline']['kernel_ms'], j['value'], j['roofline']['frac'], j['config']['mean_ipm_iterations'], j['config']['converged_frac']))"; }
for rep in 1 2; do
for q in 0 1; do
  echo -n "FRP_Q30=$q config 3: " >> $O
  FRP_Q30=$q timeout 300 python bench.py --config 3 --steps 5 --warmup 1 --no-cpu --repeats 3 2>/dev/null | tail -1 | bl >> $O
done
done
cat $O

#!/bin/bash
# GPU box, round 5 first contact of the Q4 variants: parity subset, then A/B against the three-per-CU variants (FRP_Q4=0)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batch_matches or hard_family or full_size_properties or horizon_lengths or every_kernel_variant or gauss_newton or iteration_limit or indefinite_cost or receding_horizon_warm or queue_order_hint" 2>&1 | tail -15
for q in 0 1; do
  for B in 4096 16384; do
    echo -n "FRP_Q4=$q B=$B: "
    FRP_Q4=$q timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --repeats 3 --batch $B 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('ms/step %.4f kernel_ms %.4f value %.0f mean_iters %s' % (j['ms_per_step'], j['roofline']['kernel_ms'], j['value'], j['config'].get('mean_iterations')))"
  done
done

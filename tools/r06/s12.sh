#!/bin/bash
# config 4 vs config 2 at the same batch: Gauss-Newton redos per solve (bench JSON key mean_gauss_newton_redos) and the kernel time
mkdir -p gpurun_out/r06; O=gpurun_out/r06/config4_redos.txt; : > $O
for c in "--config 4" "--config 4 --mu0 1.0" "--config 4 --mu0 0.1" "--config 2 --batch 65536"; do
  echo "== bench.py $c" >> $O
  timeout 300 python bench.py $c --steps 10 --warmup 3 --no-cpu 2>>$O | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']
print({k:(j.get(k) if k in j else c.get(k)) for k in ('value','ms_per_step','mean_ipm_iterations','mean_gauss_newton_redos','p95_ipm_iterations','max_ipm_iterations','converged_fraction')}, j['roofline'].get('kernel_ms'))" >> $O 2>&1
done
cat $O

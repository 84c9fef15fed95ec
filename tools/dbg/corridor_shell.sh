#!/bin/bash
# corridor tests + dense bench of lib_shell.so (the working tree's corridor kernels on the product's other objects) + profile build
cd "$(dirname "$0")/../.."
export FRP_LIB=$PWD/forces_resilient_planner_amd/lib_shell.so
python tools/dbg/corridor_shell_cases.py 2>&1 | grep -v amdgpu.ids | head -12
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "corridor or tick" 2>&1 | grep -B30 "short test summary" | grep "^E\|^>\|Error\|passed\|failed" | cut -c1-300 | head -20; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "corridor or tick" 2>&1 | tail -2
for P in 20000 62000; do
  timeout 300 python tests/tools/corridor_bench.py 4096 $P 0.5 2>&1 | tail -1 | cut -c1-160
  FRP_LIB=$PWD/forces_resilient_planner_amd/lib_shellprof.so timeout 300 python tests/tools/corridor_bench.py 4096 $P 0.5 2>&1 | grep "shell wave" | head -2
done
timeout 300 python tools/full_tick_bench.py 4096 10 20000 2>&1 | tail -1 | cut -c1-400

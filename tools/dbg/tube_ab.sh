#!/bin/bash
# tube kernel variants: tests + time in the tick
cd "$(dirname "$0")/../.."
P=$PWD/forces_resilient_planner_amd
for n in "$@"; do
  [ "$n" = product ] && lib="" || lib=$P/lib_$n.so
  echo "== $n"
  FRP_LIB=$lib timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "tube or tick" 2>&1 | tail -1
  for i in 1 2; do FRP_LIB=$lib timeout 300 python tools/full_tick_bench.py 4096 10 20000 2>&1 | tail -1 | cut -c100-330; done
done

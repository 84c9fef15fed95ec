#!/bin/bash
# dense-cloud corridor bench of the product and of lib_<name>.so variants: tools/dbg/corridor_dense.sh [names...]
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for n in "" "$@"; do
  for P in 20000 62000; do
    echo "== lib=${n:-product} P=$P"
    FRP_LIB=${n:+$PWD/forces_resilient_planner_amd/lib_$n.so} timeout 300 python tests/tools/corridor_bench.py 4096 $P 0.5 2>&1 | tail -4 | cut -c1-420
  done
done

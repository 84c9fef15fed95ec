#!/bin/bash
# A*: product (cached multi-push commit) vs lib_aslow.so (round-4 commit) on the pillar world, and the phase profile of both
cd "$(dirname "$0")/../.."
P=$PWD/forces_resilient_planner_amd
for n in "" aslow; do
  echo "== ${n:-product}"
  FRP_LIB=${n:+$P/lib_$n.so} timeout 600 python tests/tools/astar_bench.py 1024 pillars 20000 2>/dev/null | tail -1 | python -c "
import sys, json; j = json.loads(sys.stdin.read()); print({k: j[k] for k in ('gpu_ms', 'gpu_searches_per_s', 'expansions_max', 'gpu_us_per_expansion_of_the_longest_search', 'same_results_on_the_cpu_sample')})"
done
for n in aprof aprofslow; do echo "== $n"; for B in 1 1024; do FRP_LIB=$P/lib_$n.so timeout 600 python tests/tools/astar_prof.py $B 2>&1 | grep -v amdgpu | tail -3; done; done

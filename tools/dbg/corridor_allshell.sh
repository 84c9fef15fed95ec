#!/bin/bash
# every planner through the shell form (FRP_CORRIDOR_SHELL=all) for tile / occupancy variants: dense bench + tick corridor
cd "$(dirname "$0")/../.."
P=$PWD/forces_resilient_planner_amd
for n in "$@"; do
  for mode in ${MODES:-"" all}; do
    [ "$n" = product ] && lib="" || lib=$P/lib_$n.so
    echo "== $n shell=${mode:-flagged}"
    for C in 20000 62000; do FRP_CORRIDOR_SHELL=$mode FRP_LIB=$lib timeout 300 python tests/tools/corridor_bench.py 4096 $C 0.5 2>&1 | tail -1 | cut -c60-125; done
    FRP_CORRIDOR_SHELL=$mode FRP_LIB=$lib timeout 300 python tools/full_tick_bench.py 4096 10 20000 2>&1 | tail -1 | cut -c250-330
  done
done

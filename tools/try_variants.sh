#!/bin/bash
# run on the GPU box: for each variant library, parity quick check + phase profile + timing
for v in "$@"; do
  cp forces_resilient_planner_amd/$v forces_resilient_planner_amd/libfrp_nmpc_amd.so
  echo "=== $v"
  python tests/tools/dbg_iter.py 0 | tail -1 | sed 's/.*|dz|/|dz|/'
  python tools/prof_phases.py 4096 | tail -8
done

#!/bin/bash
# tools/try_libs.sh "<python args>" lib_a.so lib_b.so ... : run a python tool against several library variants
cmd=$1; shift
for v in "$@"; do
  cp forces_resilient_planner_amd/$v forces_resilient_planner_amd/libfrp_nmpc_amd.so
  echo "=== $v"; python $cmd 2>&1 | tail -12
done
cp forces_resilient_planner_amd/lib_main.so forces_resilient_planner_amd/libfrp_nmpc_amd.so

#!/bin/bash
for v in "$@"; do cp forces_resilient_planner_amd/$v forces_resilient_planner_amd/libfrp_nmpc_amd.so; echo "== $v"; python tools/prof_phases.py 4096 | grep -E "eval|factor|fwd|affine|backvec|total"; done
cp forces_resilient_planner_amd/lib_main.so forces_resilient_planner_amd/libfrp_nmpc_amd.so

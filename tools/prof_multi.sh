#!/bin/bash
# GPU box: phase cycle profile at several batch sizes (occupancy study), then restore the main library
cp forces_resilient_planner_amd/lib_prof.so forces_resilient_planner_amd/libfrp_nmpc_amd.so
for B in 256 1024 2048 4096 8192; do python tools/prof_phases.py $B | grep -E "^B|total|eval|factor|fwd|affine|backvec"; done
cp forces_resilient_planner_amd/lib_main.so forces_resilient_planner_amd/libfrp_nmpc_amd.so

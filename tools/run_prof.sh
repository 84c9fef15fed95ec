#!/bin/bash
# run a python tool against the profiling library, then restore the main one
cp forces_resilient_planner_amd/lib_prof.so forces_resilient_planner_amd/libfrp_nmpc_amd.so
python "$@"
cp forces_resilient_planner_amd/lib_main.so forces_resilient_planner_amd/libfrp_nmpc_amd.so

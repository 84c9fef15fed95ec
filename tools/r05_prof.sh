#!/bin/bash
export TMPDIR=/tmp
export FRP_LIB=$PWD/forces_resilient_planner_amd/lib_prof.so
for B in 1 4096; do
FRP_Q4=0 python tools/prof_lds.py $B 2 2>/dev/null
python tools/prof_lds.py $B 2 2>/dev/null
done

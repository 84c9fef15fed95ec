#!/bin/bash
export TMPDIR=/tmp
echo "== Q4"; python tools/dbg/q4_iter.py 2 32 2>/dev/null | grep maxit
bash tools/r05_first.sh
for B in 1 4096; do FRP_LIB=$PWD/forces_resilient_planner_amd/lib_prof.so python tools/prof_lds.py $B 2 2>/dev/null | grep -v "segments\|whole sweeps"; done

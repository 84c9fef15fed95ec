#!/bin/bash
export TMPDIR=/tmp
export FRP_LIB=$PWD/forces_resilient_planner_amd/lib_prof.so
FRP_Q4=0 python tools/timeline.py 4096 2>/dev/null
python tools/timeline.py 4096 2>/dev/null

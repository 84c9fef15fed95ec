#!/bin/bash
for m in -1 5 15; do python tools/dbg/twist_check.py 64 $m; done
for r in 1e10 1e11 1e13; do echo "== rho $r"; FRP_LIB=$PWD/forces_resilient_planner_amd/lib_rho$r.so python tools/dbg/twist_check.py 64 -1; done

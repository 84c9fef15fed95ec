#!/bin/bash
# tools/dbg/corridor_variant.sh <name> [flags...] -> forces_resilient_planner_amd/lib_<name>.so: the product objects of _build/ with
# frp_corridor.hip recompiled with the extra flags (experiments on the corridor kernels only; select with FRP_LIB=<name>)
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
P=forces_resilient_planner_amd
o=$P/_build/frp_corridor.$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Iinclude -mllvm -disable-machine-licm "$@" -c $P/csrc/frp_corridor.hip -o $o
objs=$(ls $P/_build/*.hip.o | grep -v frp_corridor.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $o -o $P/lib_$name.so
echo $P/lib_$name.so

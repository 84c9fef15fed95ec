#!/bin/bash
# tools/build_variant.sh <name> [extra hipcc flags...]  ->  forces_resilient_planner_amd/lib_<name>.so  (experiments only)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -c forces_resilient_planner_amd/csrc/frp_astar.hip -o forces_resilient_planner_amd/csrc/frp_astar.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function -Iinclude "$@" \
  forces_resilient_planner_amd/csrc/frp_kernels.hip forces_resilient_planner_amd/csrc/frp_ipm_lds.hip forces_resilient_planner_amd/csrc/frp_capi.hip forces_resilient_planner_amd/csrc/frp_pack.hip forces_resilient_planner_amd/csrc/frp_tube.hip forces_resilient_planner_amd/csrc/frp_corridor.hip forces_resilient_planner_amd/csrc/frp_reference.hip forces_resilient_planner_amd/csrc/frp_astar.o \
  -o forces_resilient_planner_amd/lib_$name.so

#!/bin/bash
# tools/dbg/tu_variant.sh <source.hip> <name> [flags...] -> forces_resilient_planner_amd/lib_<name>.so: the product objects of _build/ with ONE
# translation unit recompiled with the extra flags on top of its per-source flags (select with FRP_LIB=.../lib_<name>.so)
set -e
cd "$(dirname "$0")/../.."
src=$1; name=$2; shift 2
P=forces_resilient_planner_amd
o=$P/_build/${src%.hip}.$name.o
per=$(python - "$src" <<'PY'
import sys
from forces_resilient_planner_amd import build
print(" ".join(build.PER_SOURCE_FLAGS.get(sys.argv[1], [])))
PY
)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Iinclude $per "$@" -c $P/csrc/$src -o $o
objs=$(ls $P/_build/*.hip.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $o -o $P/lib_$name.so
echo $P/lib_$name.so

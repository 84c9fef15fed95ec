#!/bin/bash
# build tools/ubench/sweep_timing from the working tree and run it on the GPU box
cd "$(dirname "$0")/ubench" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value "$@" sweep_timing.hip -o sweep_timing 2>&1 | grep -E "error|Scratch" ; cd ../..
/usr/local/graft/bin/gpurun --timeout 300 -- 'tools/ubench/sweep_timing | head -5' 2>&1 | tail -5

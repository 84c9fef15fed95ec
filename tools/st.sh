#!/bin/bash
# build tools/ubench/sweep_timing from the working tree (with the solver's code-generation flags) and run it on the GPU box:
#   tools/st.sh <tag>   -> cycles per sweep / stage, and gpurun_out/st_<tag>.txt = the results of the last repetition, field by field
cd "$(dirname "$0")/.."
tag=${1:-cur}
CG="-mllvm -amdgpu-use-amdgpu-trackers=1 -mllvm -disable-machine-licm -mllvm -disable-machine-cse -mllvm -amdgpu-enable-rewrite-partial-reg-uses=0 -mllvm -amdgpu-load-store-vectorizer=0"
(cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value $CG sweep_timing.hip -o sweep_timing 2>&1 | grep -E "error|Scratch") 
/usr/local/graft/bin/gpurun --timeout 300 -- "mkdir -p gpurun_out; tools/ubench/sweep_timing 20 gpurun_out/st_$tag.txt | head -6" 2>&1 | tail -6

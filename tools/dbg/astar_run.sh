#!/bin/bash
# GPU box: the A* parity tests, then the two bench worlds
export TMPDIR=/tmp
mkdir -p gpurun_out/r05a
timeout 900 python -m pytest tests/test_gpu_astar.py -x -q -m gpu 2>&1 | tail -5
for k in pillars wall_gap; do timeout 300 python tests/tools/astar_bench.py 1024 $k 20000 2>/dev/null | tail -1; done | tee gpurun_out/r05a/astar_bench.jsonl

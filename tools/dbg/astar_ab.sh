#!/bin/bash
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_astar.py -q -x 2>&1 | tail -3
python tests/tools/astar_bench.py 1024 pillars 20000 2>/dev/null | tail -1 | cut -c1-420
python tests/tools/astar_bench.py 1024 wall_gap 20000 2>/dev/null | tail -1 | cut -c1-420
python tests/tools/astar_check.py 12 16 2>&1 | tail -3

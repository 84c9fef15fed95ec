#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -s -k "twist" 2>&1 | tail -8
timeout 3000 python tools/twist_soak.py 12 > gpurun_out/r05_twist_soak.txt 2>&1; tail -8 gpurun_out/r05_twist_soak.txt
python tools/twist_latency.py 2>/dev/null | tail -12

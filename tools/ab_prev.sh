#!/bin/bash
# GPU box: same-box A/B of the working tree against forces_resilient_planner_amd/lib_prev.so (the previous commit's build).
export TMPDIR=/tmp
R=$PWD/gpurun_out/ab; rm -rf $R; mkdir -p $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "horizon or batch or fixture or oracle" > $R/pytest.txt 2>&1; tail -3 $R/pytest.txt
tools/ubench/sweep_timing > $R/sweep_timing.txt 2>&1; tail -8 $R/sweep_timing.txt
for i in 1 2; do
  python bench.py --steps 20 --warmup 3 --no-cpu 2>/dev/null | tail -1 > $R/new_$i.json
  FRP_LIB=$PWD/forces_resilient_planner_amd/lib_prev.so python bench.py --steps 20 --warmup 3 --no-cpu 2>/dev/null | tail -1 > $R/prev_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab/*_?.json")):
    try:
        j=json.load(open(f)); print(f.split('/')[-1], "value %.0f ms/step %.4f kernel_ms %.4f pipelined %.0f" % (j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], j["config"]["pipelined_solves_per_s"] or 0))
    except Exception as e: print(f, "ERR", e)
PY

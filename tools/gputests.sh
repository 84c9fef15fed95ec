#!/bin/bash
# GPU box: the whole -m gpu suite + smoke + the default bench line (what the driver runs at round end)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_default.json; python - <<'PY'
import json
j = json.load(open("gpurun_out/bench_default.json"))
print({k: j[k] for k in ("metric", "value", "ms_per_step")}, j["roofline"], j.get("cpu_baseline"))
PY

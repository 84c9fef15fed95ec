#!/bin/bash
# GPU box, round 6 session 1: (a) the -m gpu suite on the working tree, (b) same-box baseline vs the faces-first order of the Q4 model +
# corridor wave at B = 4096 / 16384, (c) wave phases + the model wave's evaluation segments of both (profile builds), (d) the tick's
# live-row histogram, (e) where the soak's flag mismatches part, (f) the host path incl. two batches in flight.
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r06_s1.txt; : > $O
P=$PWD/forces_resilient_planner_amd
( timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 ) >> $O
bl() { python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ms/step %.4f kernel_ms %.4f value %.0f frac %.4f' % (j['ms_per_step'], j['roofline']['kernel_ms'], j['value'], j['roofline']['frac']))"; }
for rep in 1 2; do
for lib in libfrp_nmpc_amd lib_ff; do
  for B in 4096 16384; do
    echo -n "$lib B=$B: " >> $O
    FRP_LIB=$P/$lib.so timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --repeats 5 --batch $B 2>/dev/null | tail -1 | bl >> $O
  done
done
done
for lib in lib_prof lib_ffprof; do
  for B in 1 4096; do echo "== $lib" >> $O; FRP_LIB=$P/$lib.so timeout 300 python tools/prof_lds.py $B 2 2>/dev/null | grep -v "factor sweep segments\|^  model phase segments" >> $O; done
done
echo "== tick histogram" >> $O
timeout 300 python tools/dbg/tick_hist.py 2>/dev/null >> $O
echo "== soak divergence" >> $O
timeout 900 python tests/tools/soak_diverge.py >> $O 2>&1
echo "== default bench (host path incl. pipelined)" >> $O
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/r06_s1_bench_default.json
python - >> $O <<'PY'
import json
j = json.load(open("gpurun_out/r06_s1_bench_default.json"))
print({k: j[k] for k in ("value", "ms_per_step")}, j["roofline"]["frac"], j["roofline"]["kernel_ms"])
print(json.dumps(j.get("end_to_end"), indent=0)[:3000])
print(j.get("dropin_latency_ms"), j.get("full_tick", {}).get("ms_per_step"), j.get("full_tick", {}).get("ms_per_tick"))
PY
cat $O

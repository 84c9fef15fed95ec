#!/bin/bash
# the plain one-wavefront kernel's first scan through stream_hull (lib_stream.so) against the product: corridor tests, dense bench, tick
cd "$(dirname "$0")/../.."
for n in "" ${VARIANTS:-stream}; do
  export FRP_LIB=${n:+$PWD/forces_resilient_planner_amd/lib_$n.so}
  echo "== ${n:-product}"
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "corridor or tick" 2>&1 | tail -2
  for P in 20000 62000; do timeout 300 python tests/tools/corridor_bench.py 4096 $P 0.5 2>&1 | tail -1 | cut -c1-130; done
  for i in 1 2; do timeout 300 python tools/full_tick_bench.py 4096 10 20000 2>&1 | tail -1 | cut -c100-400; done
done

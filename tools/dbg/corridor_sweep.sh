#!/bin/bash
# dense bench (19 k / 62 k clouds) + the tick's corridor for lib_<name>.so builds of the corridor translation unit
cd "$(dirname "$0")/../.."
P=$PWD/forces_resilient_planner_amd
for n in "$@"; do
  [ "$n" = product ] && lib="" || lib=$P/lib_$n.so
  a=$(FRP_LIB=$lib timeout 300 python tests/tools/corridor_bench.py 4096 20000 0.5 2>&1 | tail -1 | python -c "import sys,json; print('%.3f' % (1e3*json.loads(sys.stdin.read())['seconds']))")
  b=$(FRP_LIB=$lib timeout 300 python tests/tools/corridor_bench.py 4096 62000 0.5 2>&1 | tail -1 | python -c "import sys,json; print('%.3f' % (1e3*json.loads(sys.stdin.read())['seconds']))")
  c=$(FRP_LIB=$lib timeout 300 python tools/full_tick_bench.py 4096 10 20000 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.4f tick %.4f' % (j['ms_per_step']['corridor'], j['ms_per_tick']))")
  echo "$n: 19k $a ms  62k $b ms  tick corridor $c"
done

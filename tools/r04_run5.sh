cd /root/repo; export TMPDIR=/tmp
for n in cpa cpd; do echo "== $n"; FRP_LIB=$PWD/forces_resilient_planner_amd/lib_$n.so python tools/full_tick_bench.py 4096 1 20000 0.5 0 2>&1 | grep -v "^{" | tail -8; done

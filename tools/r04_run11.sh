cd /root/repo; export TMPDIR=/tmp
for n in main a512 a1024; do
  if [ $n = main ]; then L=$PWD/forces_resilient_planner_amd/libfrp_nmpc_amd.so; else L=$PWD/forces_resilient_planner_amd/lib_$n.so; fi
  echo "== $n"; FRP_LIB=$L python tests/tools/astar_bench.py 1024 pillars 20000 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['gpu_ms'], j['gpu_searches_per_s'], j['gpu_us_per_expansion_of_the_longest_search'], j['same_results_on_the_cpu_sample'])"
  FRP_LIB=$L timeout 900 python -m pytest tests/test_gpu_astar.py -q -x 2>&1 | tail -1
done

cd /root/repo; export TMPDIR=/tmp
for n in a640 a768 a1024; do
  L=$PWD/forces_resilient_planner_amd/lib_$n.so
  echo "== $n"; FRP_LIB=$L timeout 600 python tests/tools/astar_bench.py 1024 pillars 20000 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['gpu_ms'], j['gpu_searches_per_s'], j['gpu_us_per_expansion_of_the_longest_search'], j['same_results_on_the_cpu_sample'])"
  FRP_LIB=$L timeout 600 python tests/tools/astar_bench.py 1024 wall_gap 20000 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['gpu_ms'], j['gpu_searches_per_s'], j['gpu_us_per_expansion_of_the_longest_search'], j['same_results_on_the_cpu_sample'])"
  FRP_LIB=$L timeout 600 python tests/tools/astar_bench.py 4096 empty 4000 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['gpu_ms'], j['gpu_searches_per_s'], j['same_results_on_the_cpu_sample'])"
done

cd /root/repo; export TMPDIR=/tmp
for n in main w3a w3b w2r4 w2t24; do
  if [ $n = main ]; then L=$PWD/forces_resilient_planner_amd/libfrp_nmpc_amd.so; else L=$PWD/forces_resilient_planner_amd/lib_$n.so; fi
  echo "== $n"; for i in 1 2; do FRP_LIB=$L python tools/full_tick_bench.py 4096 10 20000 0.5 0 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_tick'], j['ms_per_step']['corridor'])"; done
  FRP_LIB=$L timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "corridor" 2>&1 | tail -1
done

for v in main norb; do
  cp forces_resilient_planner_amd/lib_$v.so forces_resilient_planner_amd/libfrp_nmpc_amd.so
  for s in 1536 1792 2048; do
    echo -n "$v slots=$s: "
    FRP_RESIDENT_SLOTS=$s timeout 120 python bench.py --steps 10 --warmup 2 --no-cpu | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('2-stream %.0f  serial %.0f  kernel_ms %.3f' % (d['value'], d['config']['single_stream_solves_per_s'], d['roofline']['kernel_ms']))"
    echo -n "   B=16384: "
    FRP_RESIDENT_SLOTS=$s timeout 120 python bench.py --steps 6 --warmup 2 --no-cpu --batch 16384 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('2-stream %.0f  serial %.0f' % (d['value'], d['config']['single_stream_solves_per_s']))"
  done
done
cp forces_resilient_planner_amd/lib_main.so forces_resilient_planner_amd/libfrp_nmpc_amd.so

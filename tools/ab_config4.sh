for v in main prev nocyc; do
 if [ $v = main ]; then L=$PWD/forces_resilient_planner_amd/libfrp_nmpc_amd.so; else L=$PWD/forces_resilient_planner_amd/lib_$v.so; fi
 for f in "" "--no-order-hint"; do
  FRP_LIB=$L python bench.py --config 4 --steps 10 --warmup 3 --repeats 3 --no-cpu $f 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v','$f', round(j['value']), round(j['ms_per_step'],3), j['config'].get('mean_ipm_iterations'))"
 done
done

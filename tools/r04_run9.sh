cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bench_gpu.py -q -x 2>&1 | tail -3
export FRP_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29571 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
for cfg in 2 3; do for ch in 1 2 4; do
python bench.py --no-cpu --config $cfg --scaling strong --chunks $ch --steps 10 --warmup 2 --repeats 3 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('cfg $cfg chunks $ch ms/step %.4f value %.0f'%(j['ms_per_step'], j['value']), j['config']['strong_scaling_phases'], 'kernel_ms', j['roofline']['kernel_ms'])"
done; done
unset FRP_BENCH_FORCE_DIST WORLD_SIZE RANK LOCAL_RANK
python bench.py --no-cpu --steps 20 --warmup 3 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('main ms/step', j['ms_per_step'], 'kernel_ms', j['roofline']['kernel_ms'], j['roofline']['kernel_launches_timed'])"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "hard or batch_matches_oracle or statically" 2>&1 | tail -3

# GPU box: bench default line (with the kernel-event hook), the same without the full legs, bench + info tests
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r04
python bench.py > gpurun_out/r04/bench_default.json 2> gpurun_out/r04/bench_default.err; tail -c 3000 gpurun_out/r04/bench_default.json
python bench.py --no-cpu --steps 20 --warmup 3 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('nocpu ms/step', j['ms_per_step'], 'kernel_ms', j['roofline']['kernel_ms'], j['roofline']['kernel_launches_timed'])"
timeout 900 python -m pytest tests/test_bench_gpu.py tests/test_gpu_parity.py -q -x -k "bench or default_line or strong or monte or statically or batch_matches_oracle or face_count or dropin" 2>&1 | tail -5

cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "corridor or full_tick or fleet" 2>&1 | tail -3
for i in 1 2; do python tools/full_tick_bench.py 4096 10 20000 0.5 0 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_tick'], j['ms_per_step'])"; done
FRP_CORRIDOR_WAVE=0 python tools/full_tick_bench.py 4096 10 20000 0.5 0 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('wave off', j['ms_per_tick'], j['ms_per_step']['corridor'])"
python tests/tools/corridor_bench.py 4096 20000 0.5 | tail -1
python tests/tools/corridor_bench.py 4096 62000 0.5 | tail -1

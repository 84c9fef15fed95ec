cd /root/repo; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_astar.py -q -x 2>&1 | tail -5
python tests/tools/astar_bench.py 1024 pillars 20000 > gpurun_out/r04_astar_pillars.json; cat gpurun_out/r04_astar_pillars.json | cut -c1-400

set -x
cd /root/repo
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
(FRP_LIB=$PWD/forces_resilient_planner_amd/lib_pc3.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "test_batch_matches_oracle or test_every_kernel_variant or face_count_beyond" 2>&1 | tail -5) > gpurun_out/r04/ab1_parity.txt
cat gpurun_out/r04/ab1_parity.txt
tools/ab_variants.sh p3 c3 pc3 p1 2>&1 | tee gpurun_out/r04/ab1.txt

#!/bin/bash
# GPU box, round 6: soak with every covered launch on the high-residency variants (FRP_Q30_MIN_B=0: the soak's N = 21..30 launches hold at most 1500 problems, below the
# three-per-CU variant's default threshold), rocprofv3 kernel stats of configs[3], the final -m gpu suite + smoke + default bench line
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD/gpurun_out/r06e; rm -rf $R; mkdir -p $R
FRP_Q30_MIN_B=0 timeout 900 python tests/tools/soak.py 240 > $R/soak_q30.txt 2>&1
ROOT=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof3 -o c3 -- python $ROOT/bench.py --config 3 --steps 10 --warmup 2 --repeats 3 --no-cpu > $R/config3_under_rocprof.json 2> $R/prof3.log)
find $R/prof3 -name "*kernel_stats.csv" -exec cp {} $R/config3_kernel_stats.csv \;
bash tools/gputests.sh > $R/gputests.txt 2>&1
grep "^soak:" $R/soak_q30.txt | cut -c1-250; tail -3 $R/gputests.txt | cut -c1-200; head -5 $R/config3_kernel_stats.csv

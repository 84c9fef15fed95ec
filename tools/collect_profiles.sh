#!/bin/bash
# GPU box: rocprofv3 evidence for bench.py's roofline numbers -> gpurun_out/prof/ (summarised by tools/summarize_profiles.py)
# Counters are collected in their own runs, one per counter (FETCH_SIZE and WRITE_SIZE do not fit one pass), with
# --kernel-trace only.
set -x
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof; rm -rf $OUT; mkdir -p $OUT
BENCH="python $PWD/bench.py --steps 20 --warmup 3 --no-cpu --streams 1"   # one stream: launches do not overlap, so per-kernel durations are those of an isolated launch
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $BENCH > $OUT/bench_stats.json 2> $OUT/stats.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o bench -- $BENCH > /dev/null 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o bench -- $BENCH > /dev/null 2> $OUT/write.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/calib_fetch -o calib -- $OLDPWD/tools/ubench/hbm_calib > /dev/null 2> $OUT/calib_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/calib_write -o calib -- $OLDPWD/tools/ubench/hbm_calib > /dev/null 2> $OUT/calib_write.log
cd $OLDPWD
$BENCH > $OUT/bench_plain.json
find $OUT -name "*.csv" | head -30

#!/bin/bash
# GPU box: rocprofv3 per-kernel statistics of the whole on-device tick -> gpurun_out/prof_tick/ (copy the two small files to profiles/)
export TMPDIR=/tmp
R=$PWD; OUT=$R/gpurun_out/prof_tick; rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o tick -- python $R/tools/full_tick_bench.py 4096 10 20000 > $OUT/tick.json 2> $OUT/log.txt
cd $R
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $OUT/kernel_stats.csv; head -12 "$f"; else echo "no kernel_stats.csv"; tail -5 $OUT/log.txt; fi
find $OUT -name "*kernel_trace.csv" -delete

#!/bin/bash
# FETCH_SIZE of the dominant kernel for several resident-slot counts (is the L2 holding the state at low counts?)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_slots; rm -rf $OUT; mkdir -p $OUT
cd /tmp
for S in 384 512 768 1024 1536; do
  FRP_RESIDENT_SLOTS=$S rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/s$S -o b -- python $OLDPWD/bench.py --steps 6 --warmup 2 --no-cpu --streams 1 > $OUT/s$S.json 2> $OUT/s$S.log
done

#!/bin/bash
# GPU box, round 6: soak on FRESH draws (seeds other than the 2026 every earlier soak used), once on the default variant choice and once with
# every covered launch forced onto the high-residency variants (FRP_Q30_MIN_B=0, FRP_Q4_MIN_B=0)
R=gpurun_out/r06; mkdir -p $R
timeout 1000 python tests/tools/soak.py 780 2027 > $R/soak_seed2027.txt 2>&1
FRP_Q30_MIN_B=0 FRP_Q4_MIN_B=0 timeout 1000 python tests/tools/soak.py 780 2028 > $R/soak_seed2028_high_residency.txt 2>&1
tail -n 4 $R/soak_seed2027.txt | cut -c1-400; tail -n 4 $R/soak_seed2028_high_residency.txt | cut -c1-400

#!/bin/bash
# GPU box: SQ performance counters of the dominant kernel (instruction mix, stall breakdown), 8 SQ counters per pass,
# counters only with --kernel-trace.  -> gpurun_out/prof_sq/passN/
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_sq; rm -rf $OUT; mkdir -p $OUT
BENCH="python $PWD/bench.py --steps 6 --warmup 2 --no-cpu --streams 1"
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pass$i -o bench -- $BENCH > /dev/null 2> $OUT/pass$i.log
done
ls $OUT/*/

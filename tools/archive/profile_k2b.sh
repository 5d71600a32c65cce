#!/bin/bash
# K2b against the FP4 strips under rocprofv3 on the headline shape (gpurun, from the repo root):
# kernel trace + stats, then two SQ counter passes per form, each in its own run.  $1 = tag
set -e
R=$PWD
OUT=$R/gpurun_out/prof_$1
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
for ops in 4 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace$ops -o t -- python3 $R/tools/midsize_pass.py --rows 10000 --passes 200 --opt variant=4 --opt k2_strip_operands=$ops --opt k2_fold_inline=0 > $OUT/pass$ops.json 2> $OUT/trace$ops.err
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $OUT/pmc1_$ops -o p -- python3 $R/tools/midsize_pass.py --rows 10000 --passes 5 --warm-ms 5 --opt variant=4 --opt k2_strip_operands=$ops --opt k2_fold_inline=0 > $OUT/pmc1_$ops.json 2> $OUT/pmc1_$ops.err
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL --kernel-trace --output-format csv -d $OUT/pmc2_$ops -o p -- python3 $R/tools/midsize_pass.py --rows 10000 --passes 5 --warm-ms 5 --opt variant=4 --opt k2_strip_operands=$ops --opt k2_fold_inline=0 > $OUT/pmc2_$ops.json 2> $OUT/pmc2_$ops.err
  rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc3_$ops -o p -- python3 $R/tools/midsize_pass.py --rows 10000 --passes 5 --warm-ms 5 --opt variant=4 --opt k2_strip_operands=$ops --opt k2_fold_inline=0 > $OUT/pmc3_$ops.json 2> $OUT/pmc3_$ops.err || true
  python3 $R/tools/pmc_summary.py $OUT/pmc_summary_$ops.csv $(find $OUT/pmc1_$ops $OUT/pmc2_$ops $OUT/pmc3_$ops -name "*counter_collection.csv")
done
cat $OUT/pass4.json $OUT/pass5.json
find $OUT -name "*kernel_stats.csv" | xargs -I{} sh -c 'echo {}; head -6 {}'
cat $OUT/pmc_summary_4.csv $OUT/pmc_summary_5.csv

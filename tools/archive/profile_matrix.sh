#!/bin/bash
# Profiles of the materialised-output path (tools/bench_matrix.py, headline shape, AND only) on
# the GPU box: kernel trace + stats, then SQ / LDS / TCC counter passes, each in its own run.
set -e
R=$PWD
OUT=$R/gpurun_out/prof_matrix_$1
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/tools/bench_matrix.py --ops and --reps 30 > $OUT/bench.json 2> $OUT/trace.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $OUT/pmc1 -o p -- python3 $R/tools/bench_matrix.py --ops and --reps 5 > $OUT/pmc1.json 2> $OUT/pmc1.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD --kernel-trace --output-format csv -d $OUT/pmc2 -o p -- python3 $R/tools/bench_matrix.py --ops and --reps 5 > $OUT/pmc2.json 2> $OUT/pmc2.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o p -- python3 $R/tools/bench_matrix.py --ops and --reps 5 > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o p -- python3 $R/tools/bench_matrix.py --ops and --reps 5 > $OUT/write.json 2> $OUT/write.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/tcc -o p -- python3 $R/tools/bench_matrix.py --ops and --reps 5 > $OUT/tcc.json 2> $OUT/tcc.err
cat $OUT/bench.json
cat $OUT/trace/t_kernel_stats.csv

#!/bin/bash
# Profiles of the default bench path on the GPU box (run through gpurun from the repo root):
# kernel trace + stats, then SQ counter passes, each in its own rocprofv3 run.
set -e
R=$PWD
OUT=$R/gpurun_out/prof_$1
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/trace.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $OUT/pmc1 -o p -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/pmc1.json 2> $OUT/pmc1.err
rocprofv3 --pmc SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/pmc2 -o p -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/pmc2.json 2> $OUT/pmc2.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o p -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o p -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/write.json 2> $OUT/write.err
find $OUT -name "*.csv" | head -20
cat $OUT/bench.json

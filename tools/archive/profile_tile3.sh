#!/bin/bash
# tile16_bits_kernel (k2_tile_shape = 3) against tilebits8_kernel (2) under rocprofv3 at the headline shape.  $1 = tag
set -e
R=$PWD; OUT=$R/gpurun_out/prof_tile_$1; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for ts in 4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace$ts -o t -- python3 $R/tools/bench_matrix.py --ops and --reps 30 --opt k2_tile_shape=$ts > $OUT/trace$ts.json 2> $OUT/trace$ts.err
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/pmc1_$ts -o p -- python3 $R/tools/bench_matrix.py --ops and --reps 5 --opt k2_tile_shape=$ts > $OUT/pmc1_$ts.json 2> $OUT/pmc1_$ts.err
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc2_$ts -o p -- python3 $R/tools/bench_matrix.py --ops and --reps 5 --opt k2_tile_shape=$ts > $OUT/pmc2_$ts.json 2> $OUT/pmc2_$ts.err
done
cd $R
python3 tools/pmc_summary.py $OUT/pmc_summary.csv $(find $OUT -name "*counter_collection.csv")
cat $OUT/trace4.json
grep -h -i "tile" $OUT/trace4/t_kernel_stats.csv
grep "tile" $OUT/pmc_summary.csv

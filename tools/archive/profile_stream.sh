#!/bin/bash
# Counters of the one-launch stream kernel (K2q, bitstream_kernel) at mid-size N (M = 65536), each counter set in
# its own rocprofv3 pass, the program directly after `--`. usage (repo root, through gpurun): tools/profile_stream.sh
set -e
R=$PWD; OUT=$R/gpurun_out/prof_stream; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for N in 1024 2048 4096 8192; do
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/sq_$N -o p -- python3 $R/tools/midsize_pass.py --rows $N --passes 20 --warm-ms 20 > $OUT/sq_$N.json 2> $OUT/sq_$N.err
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d $OUT/lds_$N -o p -- python3 $R/tools/midsize_pass.py --rows $N --passes 20 --warm-ms 20 > $OUT/lds_$N.json 2> $OUT/lds_$N.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch_$N -o p -- python3 $R/tools/midsize_pass.py --rows $N --passes 20 --warm-ms 20 > $OUT/fetch_$N.json 2> $OUT/fetch_$N.err
done
cd $R
for N in 1024 2048 4096 8192; do
  python3 tools/pmc_summary.py $OUT/summary_$N.csv $OUT/sq_$N/*counter_collection.csv $OUT/lds_$N/*counter_collection.csv $OUT/fetch_$N/*counter_collection.csv
  echo "== N = $N"; grep bitstream $OUT/summary_$N.csv
done

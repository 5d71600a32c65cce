#!/bin/bash
# Evidence for the materialised-output kernels at the headline shape (run through gpurun from the repo
# root): wall time per call of the four kernels (interleaved, 3 rounds), then kernel durations and SQ /
# TCC counters of the default (bit-operand, two waves per SIMD) kernel, each counter set in its own pass.
set -e
R=$PWD; OUT=$R/gpurun_out/prof_matrix_bits; mkdir -p $OUT
{
  echo "# wall ms per call (tools/bench_matrix.py, AND, 20 reps, min), option k2_tile_shape: 2 = bit operands, two waves per SIMD (default); 1 = bit operands, one wave per SIMD; 16 / 32 = FP4 shadow kernels (expansion pass included)"
  for round in 1 2 3; do
    for ts in 2 1 16 32; do
      printf "k2_tile_shape=%d round %d: " $ts $round
      timeout -k 10 120 python3 tools/bench_matrix.py --ops and --reps 20 --opt k2_tile_shape=$ts 2>/dev/null | grep -o '"ms_per_call": [0-9.]*'
    done
  done
  echo "# all ops, default kernel"
  timeout -k 10 120 python3 tools/bench_matrix.py --reps 20 2>/dev/null
} > $OUT/wall.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/tools/bench_matrix.py --ops and --reps 20 > $OUT/trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o p -- python3 $R/tools/bench_matrix.py --ops and --reps 5 > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o p -- python3 $R/tools/bench_matrix.py --ops and --reps 5 > $OUT/write.json 2> $OUT/write.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/tcc -o p -- python3 $R/tools/bench_matrix.py --ops and --reps 5 > $OUT/tcc.json 2> $OUT/tcc.err
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/pmc1 -o p -- python3 $R/tools/bench_matrix.py --ops and --reps 5 > $OUT/pmc1.json 2> $OUT/pmc1.err
cd $R
python3 tools/pmc_summary.py $OUT/pmc_summary.csv $OUT/fetch/p_counter_collection.csv $OUT/write/p_counter_collection.csv $OUT/tcc/p_counter_collection.csv $OUT/pmc1/p_counter_collection.csv
cat $OUT/wall.txt
grep -i "tilebits\|zero_tiles" $OUT/trace/t_kernel_stats.csv
grep tilebits $OUT/pmc_summary.csv

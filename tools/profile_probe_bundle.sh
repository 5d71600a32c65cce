#!/bin/bash
# [r6] probe_lists_kernel (one group of 128 rows per workgroup) against probe_lists_fat_kernel (bundles of four) at c4
# (STORM_t, N = 10000 x M = 524288): kernel durations and LDS counters per load, each counter set in its own rocprofv3
# pass (the program directly after `--`). Run through gpurun from the repo root.
set -e
R=$PWD; OUT=$R/gpurun_out/prof_bundle; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for B in 1 4; do
  for L in 104 524 20971; do
    export STORM_PROBE_BUNDLE=$B
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_b${B}_$L -o t -- python3 $R/tools/bench_sparse_probe.py $L > $OUT/trace_b${B}_$L.json 2> $OUT/trace_b${B}_$L.err
    rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/sq_b${B}_$L -o p -- python3 $R/tools/bench_sparse_probe.py $L > $OUT/sq_b${B}_$L.json 2> $OUT/sq_b${B}_$L.err
  done
done
cd $R
python3 tools/probe_bundle_summary.py $OUT > $OUT/summary.txt; cat $OUT/summary.txt; exit 0
for B in 1 4; do for L in 104 524 20971; do
  echo "== bundle $B load $L"
  grep -h "probe_lists" $OUT/trace_b${B}_$L/*kernel_stats.csv | head -2
  python3 tools/pmc_summary.py $OUT/sq_b${B}_$L.csv $(find $OUT/sq_b${B}_$L -name '*counter_collection.csv') && grep "probe_lists" $OUT/sq_b${B}_$L.csv
done; done > $OUT/summary.txt 2>&1
cat $OUT/summary.txt

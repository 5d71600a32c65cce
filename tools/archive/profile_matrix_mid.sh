#!/bin/bash
# kernel durations of the materialised output at LD-window row counts (rocprofv3 --kernel-trace --stats), from the repo root
set -e
R=$PWD
OUT=$R/gpurun_out/prof_mid
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
for n in 1024 2048 4096; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n$n -o t -- python3 $R/tools/bench_matrix_sizes.py $n > $OUT/n$n.log 2> $OUT/n$n.err
  f=$(find $OUT/n$n -name "*kernel_stats.csv" | head -1)
  echo "== N=$n"; cut -d, -f1-8 $f | head -8 | cut -c1-230
done

#!/bin/bash
# Evidence for the list-probe kernel (K4, probe_lists_kernel) at c4 (STORM_t, N = 10000 x M = 524288), run
# through gpurun from the repo root: wall time per load against the dense path, then per load kernel durations
# and LDS / L2 counters, each counter set in its own rocprofv3 pass (the program directly after `--`).
set -e
R=$PWD; OUT=$R/gpurun_out/prof_sparse; mkdir -p $OUT
python3 tools/bench_sparse_probe.py 5,104,524,2097,5242,10485,20971,30000 > $OUT/wall.jsonl 2> $OUT/wall.err
cd /tmp; export TMPDIR=/tmp
for L in 104 524 5242 20971; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$L -o t -- python3 $R/tools/bench_sparse_probe.py $L > $OUT/trace_$L.json 2> $OUT/trace_$L.err
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/sq_$L -o p -- python3 $R/tools/bench_sparse_probe.py $L > $OUT/sq_$L.json 2> $OUT/sq_$L.err
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/tcc_$L -o p -- python3 $R/tools/bench_sparse_probe.py $L > $OUT/tcc_$L.json 2> $OUT/tcc_$L.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch_$L -o p -- python3 $R/tools/bench_sparse_probe.py $L > $OUT/fetch_$L.json 2> $OUT/fetch_$L.err
done
cd $R
python3 tools/probe_roof.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt

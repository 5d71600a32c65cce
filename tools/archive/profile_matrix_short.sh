set -e
R=$PWD; OUT=$R/gpurun_out/prof_matrix_$1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o p -- python3 $R/tools/bench_matrix.py --ops and --reps 5 > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/tcc -o p -- python3 $R/tools/bench_matrix.py --ops and --reps 5 > $OUT/tcc.json 2> $OUT/tcc.err
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/pmc1 -o p -- python3 $R/tools/bench_matrix.py --ops and --reps 5 > $OUT/pmc1.json 2> $OUT/pmc1.err

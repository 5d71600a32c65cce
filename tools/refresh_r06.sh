#!/bin/bash
# Round-6 refresh on the GPU box (gpurun, from the repo root): bench (unprofiled + rocprofv3 passes), harness TSVs at c2 and c4,
# the output kernels over row counts, K2h under rocprofv3, first-call costs, K5, the wrappers, the soak, the whole GPU suite.
# $1 = 1: first half (bench, profiles, TSVs, first calls, K2h), 2: second half (tools, soak, suite) - two gpurun calls.
# Everything under gpurun_out/refresh_r06/, gpurun_out/prof_r06/ and gpurun_out/prof_tile128_r06/.
set -e
R=$PWD
O=$R/gpurun_out/refresh_r06
mkdir -p $O
if [ "${1:-1}" = "1" ]; then
python3 bench.py > $O/bench_unprofiled.json 2> $O/bench_unprofiled.err
tail -c 600 $O/bench_unprofiled.json; echo
bash tools/profile_default.sh r06 > $O/profile.log 2>&1 || tail -5 $O/profile.log
echo "profile done"
./stormbitmaps_amd/storm_benchmark 65536 10000 32768,6553,655,65,5 --cpu-seconds 0.5 > $O/storm_benchmark_c2.tsv 2> $O/storm_benchmark_c2.err
echo "c2 tsv done"
./stormbitmaps_amd/storm_benchmark 524288 10000 262144,131072,52428,20971,5242,524,104 --cpu-seconds 0.5 > $O/storm_benchmark_c4.tsv 2> $O/storm_benchmark_c4.err
echo "c4 tsv done"
python3 tools/bench_cold.py > $O/cold.jsonl 2> $O/cold.err || true
echo "cold done"
python3 tools/bench_matrix_sizes.py > $O/matrix_sizes.jsonl 2> $O/matrix_sizes.err || true
bash tools/profile_tile128.sh r06 > $O/profile_tile128.log 2>&1 || tail -5 $O/profile_tile128.log
echo "tile128 done"
exit 0
fi
python3 tools/check_tile5.py --quick > $O/check_tile5.txt 2>&1 || true
python3 tools/bench_wrapper.py > $O/wrapper.jsonl 2>&1 || true
python3 tools/bench_pass_sizes.py > $O/pass_sizes.jsonl 2>&1 || true
python3 tools/check_lists_matrix.py --draws 5,52,104,190,262,524,1048,2096,3145,3670 > $O/lists_matrix.txt 2>&1 || true
python3 tools/bench_storm_matrix.py --draws 104,524,20971 > $O/storm_matrix.jsonl 2>&1 || true
python3 tools/bench_sparse_small.py > $O/sparse_small.jsonl 2>&1 || true
echo "tools done"
python3 tools/soak_parity.py --seconds 240 --seed 66 > $O/soak.txt 2>&1 || true
tail -1 $O/soak.txt
python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1 || true
tail -3 $O/pytest_gpu.txt

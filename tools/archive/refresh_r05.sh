#!/bin/bash
# Round-5 refresh on the GPU box (gpurun, from the repo root): bench (unprofiled + rocprofv3 passes), harness TSVs at c2 and c4,
# the output kernels, the wrappers, K5, the soak, the whole GPU suite. Everything under gpurun_out/refresh_r05/ and
# gpurun_out/prof_r05/.
set -e
R=$PWD
O=$R/gpurun_out/refresh_r05
mkdir -p $O
python3 bench.py > $O/bench_unprofiled.json 2> $O/bench_unprofiled.err
tail -c 600 $O/bench_unprofiled.json; echo
bash tools/profile_default.sh r05 > $O/profile.log 2>&1 || tail -5 $O/profile.log
echo "profile done"
./stormbitmaps_amd/storm_benchmark 65536 10000 32768,6553,655,65,5 --cpu-seconds 0.5 > $O/storm_benchmark_c2.tsv 2> $O/storm_benchmark_c2.err
echo "c2 tsv done"
./stormbitmaps_amd/storm_benchmark 524288 10000 262144,131072,52428,20971,5242,524,104 --cpu-seconds 0.5 > $O/storm_benchmark_c4.tsv 2> $O/storm_benchmark_c4.err
echo "c4 tsv done"
python3 tools/check_tile5.py --quick > $O/check_tile5.txt 2>&1 || true
python3 tools/bench_tile_round.py > $O/tile_round.jsonl 2>&1 || true
python3 tools/bench_wrapper.py > $O/wrapper.jsonl 2>&1 || true
python3 tools/bench_matrix.py --reps 30 > $O/bench_matrix.jsonl 2>&1 || true
python3 tools/check_lists_matrix.py --draws 5,52,104,190,262,524,1048,2096,3145,3670 > $O/lists_matrix.txt 2>&1 || true
python3 tools/bench_storm_matrix.py --draws 104,524,20971 > $O/storm_matrix.jsonl 2>&1 || true
python3 tools/bench_sparse_small.py > $O/sparse_small.jsonl 2>&1 || true
echo "tools done"
python3 tools/soak_parity.py --seconds 300 --seed 55 > $O/soak.txt 2>&1 || true
tail -1 $O/soak.txt
python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1 || true
tail -3 $O/pytest_gpu.txt

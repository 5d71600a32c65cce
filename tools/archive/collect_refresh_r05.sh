#!/bin/bash
# gpurun_out/refresh_r05 + prof_r05 (tools/refresh_r05.sh on the GPU box) -> profiles/r05_i_* ; from the repo root, on the CPU side
set -e
P=gpurun_out/prof_r05; O=gpurun_out/refresh_r05
python3 tools/pmc_traffic.py $P/fetch/p_counter_collection.csv $P/write/p_counter_collection.csv 4 10000 1024 profiles/pmc_hbm_bytes_per_launch.json --merge | head -4
python3 tools/pmc_summary.py profiles/r05_i_pmc_summary_bench_c2.csv $P/pmc1/p_counter_collection.csv $P/pmc2/p_counter_collection.csv $P/fetch/p_counter_collection.csv $P/write/p_counter_collection.csv
cp $P/trace/t_kernel_stats.csv profiles/r05_i_kernel_stats_bench_c2.csv
cp $P/bench.json profiles/r05_i_bench_c2_profiled.json
cp $O/bench_unprofiled.json profiles/r05_i_bench_c2.json
cp $O/storm_benchmark_c2.tsv profiles/r05_i_storm_benchmark_c2.tsv
cp $O/storm_benchmark_c4.tsv profiles/r05_i_storm_benchmark_c4.tsv
cp $O/pytest_gpu.txt profiles/r05_i_pytest_gpu.txt
grep "^{" $O/tile_round.jsonl > profiles/r05_i_tile_round.jsonl
grep "^{" $O/wrapper.jsonl > profiles/r05_i_wrapper_streamed.jsonl
grep "^{" $O/bench_matrix.jsonl > profiles/r05_i_bench_matrix_c2.jsonl
grep "^{" $O/lists_matrix.txt > profiles/r05_i_storm_matrix_lists.jsonl
grep "^{" $O/storm_matrix.jsonl > profiles/r05_i_storm_matrix_c4.jsonl
grep "^{" $O/sparse_small.jsonl > profiles/r05_i_sparse_small_calls.jsonl
tail -1 $O/soak.txt > profiles/r05_i_soak_parity.jsonl
grep -v "^/opt" $O/check_tile5.txt | tail -8 > profiles/r05_i_tile_kernels_c2.jsonl
python3 - <<'PY'
import json,csv
d=json.load(open('profiles/r05_i_bench_c2.json'))
print("bench ms/step", d['ms_per_step'], "frac", d['roofline']['frac'], "kernel_ms", d['roofline']['kernel_ms'], "traffic", d['roofline']['traffic'])
for r in csv.DictReader(open('profiles/r05_i_kernel_stats_bench_c2.csv')):
    if 'strip16_bits' in r['Name']: print("rocprof", r['Calls'], r['AverageNs'], r['MinNs'])
PY
grep "strip16_bits_kernel,derived" profiles/r05_i_pmc_summary_bench_c2.csv | cut -c1-160
cat profiles/r05_i_sparse_small_calls.jsonl | cut -c1-120
cat profiles/r05_i_soak_parity.jsonl | cut -c1-300
tail -1 profiles/r05_i_pytest_gpu.txt

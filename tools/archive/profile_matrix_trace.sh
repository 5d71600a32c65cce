# kernel durations of the materialised-output call (no counters): tools/profile_matrix_trace.sh <tag> [bench_matrix args]
set -e
R=$PWD; TAG=$1; shift; OUT=$R/gpurun_out/trace_matrix_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $R/tools/bench_matrix.py --ops and --reps 20 "$@" > $OUT/bench.json 2> $OUT/err.txt
cat $OUT/bench.json
python3 - <<PY
import csv, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open("$OUT/t_kernel_trace.csv")):
    d[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    v2 = sorted(v)
    print(f"{k:60s} n={len(v):4d} mean={sum(v)/len(v):9.1f} us  min={v2[0]:9.1f}  median={v2[len(v)//2]:9.1f}")
PY

#!/bin/bash
# Per-kernel split of a mid-size pass (M = 65536): rocprofv3 kernel trace + stats per N and path.
# usage (from the repo root, through gpurun): tools/profile_midsize.sh TAG
set -e
R=$PWD
TAG=$1
OUT=$R/gpurun_out/midsize_$TAG
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
for N in 1024 2048 4096; do
  for OP in 0 4; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n${N}_op$OP -o t -- python3 $R/tools/midsize_pass.py --rows $N --opt k2_strip_operands=$OP >> $OUT/passes.jsonl 2> $OUT/n${N}_op$OP.err
    python3 - $OUT/n${N}_op$OP <<'PY' >> $OUT/kernels.txt
import csv, glob, sys
d = sys.argv[1]
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    print("==", d.split("/")[-1])
    for r in csv.DictReader(open(f)):
        if float(r['Percentage']) > 0.5:
            print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg_ns {float(r['AverageNs']):10.0f} pct {r['Percentage']}")
PY
  done
done
cat $OUT/passes.jsonl $OUT/kernels.txt

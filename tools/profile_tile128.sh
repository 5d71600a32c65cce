#!/bin/bash
# tile128_kernel (K2h) under rocprofv3 on the GPU box (gpurun, from the repo root): per row count a kernel trace with stats and
# two counter passes, each in its own run; one summary text.  $1 = tag
set -e
R=$PWD
OUT=$R/gpurun_out/prof_tile128_$1
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
for N in 1024 2048 4096; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$N -o t -- python3 $R/tools/bench_matrix_sizes.py $N --only 6:0 > $OUT/trace_$N.json 2> $OUT/trace_$N.err
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $OUT/pmc1_$N -o p -- python3 $R/tools/bench_matrix_sizes.py $N --only 6:0 > $OUT/pmc1_$N.json 2> $OUT/pmc1_$N.err
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc2_$N -o p -- python3 $R/tools/bench_matrix_sizes.py $N --only 6:0 > $OUT/pmc2_$N.json 2> $OUT/pmc2_$N.err
done
python3 - $OUT > $OUT/summary.txt <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
print("tile128_kernel (K2h), M = 65536 dense, rows x rows triangle into device memory; rocprofv3, one run per counter set")
for N in (1024, 2048, 4096):
    line = [l for l in open(f"{out}/trace_{N}.json") if l.startswith("{")]
    print(f"\nrows {N}: unprofiled-call view from the same run: {line[-1].strip() if line else '-'}")
    for f in glob.glob(f"{out}/trace_{N}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "tile128" in r["Name"] or "zero" in r["Name"]:
                print(f"  kernel stats: {r['Name'][:60]} calls {r['Calls']} avg_ns {r['AverageNs']} min_ns {r['MinNs']} max_ns {r['MaxNs']}")
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/pmc?_{N}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "tile128" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    c = {k: sum(v) / len(v) for k, v in acc.items()}
    for k in sorted(c):
        print(f"  {k:28s} {c[k]:.5g}")
    if "SQ_BUSY_CYCLES" in c:
        print("  matrix pipe busy (SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 * 1024))", round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["SQ_BUSY_CYCLES"] / 32 * 1024), 3))
    if "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"]:
        print("  bank conflict / idx active", round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4))
    if "SQ_WAVE_CYCLES" in c:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY"):
            print(f"  {k} / wave cycles", round(c[k] / c["SQ_WAVE_CYCLES"], 3))
PY
cat $OUT/summary.txt

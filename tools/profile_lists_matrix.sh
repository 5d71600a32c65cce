#!/bin/bash
# K5's window kernel under rocprofv3 counters at the c4 shape (gpurun, from the repo root): $1 = tag, $2 = positions per row
set -e
R=$PWD
OUT=$R/gpurun_out/prof_lists_$1
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
D=${2:-2096}
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc1 -o p -- python3 $R/tools/probes/lists_ablate.py $D > $OUT/pmc1.json 2> $OUT/pmc1.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d $OUT/pmc2 -o p -- python3 $R/tools/probes/lists_ablate.py $D > $OUT/pmc2.json 2> $OUT/pmc2.err || true
python3 - $OUT $D <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "lists_matrix" in r["Kernel_Name"]]
    # the probe runs debug 0 first (2 warm-up + 8 timed launches): the first 10 dispatches are the real kernel
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})[:10]
    for r in rows:
        if int(r["Dispatch_Id"]) in ids: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in acc.items()}
print(f"lists_matrix_kernel at {sys.argv[2]} positions per row (c4 shape), averages per launch")
for k in sorted(c): print(f"  {k:28s} {c[k]:.5g}")
if c.get("SQ_LDS_IDX_ACTIVE"): print("  bank conflict cycles / LDS index-active cycles", round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 3))
if c.get("SQ_BUSY_CYCLES"): print("  LDS index-active cycles per CU-cycle", round(c["SQ_LDS_IDX_ACTIVE"] / (c["SQ_BUSY_CYCLES"] / 32 * 256), 3))
if c.get("SQ_WAVE_CYCLES"):
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY"): print(f"  {k} / wave cycles", round(c[k] / c["SQ_WAVE_CYCLES"], 3))
PY

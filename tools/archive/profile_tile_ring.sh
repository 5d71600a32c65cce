#!/bin/bash
# tools/probes/tile_ring under rocprofv3 (gpurun, from the repo root): SQ counter passes, each in its own run. $1 = tag, $2 = ablate, $3 = sched
set -e
R=$PWD
OUT=$R/gpurun_out/prof_ring_$1
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
B=$R/tools/probes/tile_ring
A=${2:-0}
S=${3:-0}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $OUT/pmc1 -o p -- $B 10000 65536 3 $A 0 $S > $OUT/pmc1.json 2> $OUT/pmc1.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL --kernel-trace --output-format csv -d $OUT/pmc2 -o p -- $B 10000 65536 3 $A 0 $S > $OUT/pmc2.json 2> $OUT/pmc2.err
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc3 -o p -- $B 10000 65536 3 $A 0 $S > $OUT/pmc3.json 2> $OUT/pmc3.err || true
python3 - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "tile_ring" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in acc.items()}
for k in sorted(c):
    print(f"{k:32s} {c[k]:.5g}")
if "SQ_BUSY_CYCLES" in c:
    print("matrix pipe busy", c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["SQ_BUSY_CYCLES"] / 32 * 1024))
if "SQ_LDS_IDX_ACTIVE" in c:
    print("bank conflict / idx active", c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"])
    print("LDS idx active per CU-cycle", c["SQ_LDS_IDX_ACTIVE"] / (c.get("SQ_BUSY_CYCLES", 0) / 32 * 256 or 1))
if "SQ_WAVE_CYCLES" in c:
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY"):
        print(k, "/ wave cycles", c[k] / c["SQ_WAVE_CYCLES"])
PY

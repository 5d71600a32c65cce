#!/bin/bash
# kernel and memory-copy trace of the streamed raw-buffer wrapper at the headline shape (from the repo root, on the GPU box)
set -e
R=$PWD
OUT=$R/gpurun_out/prof_wrapper
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/t -o w -- python3 $R/tools/wrapper_loop.py 10 > $OUT/run.log 2> $OUT/run.err
ls $OUT/t
python3 - <<PY
import csv, glob
k = list(csv.DictReader(open(glob.glob("$OUT/t/*kernel_trace.csv")[0])))
m = list(csv.DictReader(open(glob.glob("$OUT/t/*memory_copy_trace.csv")[0])))
print("kernel cols", list(k[0].keys())[:12])
print("copy cols", list(m[0].keys()))
ev = []
for r in k:
    if "strip16_bits" in r["Kernel_Name"] or "fold_slots" in r["Kernel_Name"]:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0][-22:]))
for r in m:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
# the last call: the last 8 strip kernels and everything from the copy in front of the first of them
ks = [i for i, e in enumerate(ev) if "strip16_bits" in e[2]]
i0 = max(0, ks[-8] - 2)
t0 = ev[i0][0]
for s, e, n in ev[i0:]:
    print("%9.1f us  +%8.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
PY

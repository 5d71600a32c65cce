#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per kernel (one row per kernel x counter) from one or
more p_counter_collection.csv files:  python tools/pmc_summary.py out.csv pass1.csv [pass2.csv ...]"""
import collections
import csv
import sys


def main():
    out, paths = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(list)
    for path in paths:
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if name.startswith("storm::") or "probe_lists" in name or "tile128" in name:
                acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "launches", "mean", "min", "max"])
        for (name, counter), v in sorted(acc.items()):
            w.writerow([name, counter, len(v), f"{sum(v) / len(v):.6g}", f"{min(v):.6g}", f"{max(v):.6g}"])
        by_kernel = collections.defaultdict(dict)
        for (name, counter), v in acc.items():
            by_kernel[name][counter] = sum(v) / len(v)
        for name, c in sorted(by_kernel.items()):
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c and c["SQ_BUSY_CYCLES"]:
                # SQ_BUSY_CYCLES sums the 32 shader engines' busy cycles (= 32 x the kernel's duration
                # in cycles); SQ_VALU_MFMA_BUSY_CYCLES sums the matrix-pipe cycles of all 1024 SIMDs
                frac = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["SQ_BUSY_CYCLES"] / 32.0 * 1024.0)
                w.writerow([name, "derived: matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 * 1024 SIMDs)",
                            "", f"{frac:.4f}", "", ""])
            if "SQ_LDS_BANK_CONFLICT" in c and "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"]:
                w.writerow([name, "derived: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE", "", f"{c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.4f}", "", ""])


if __name__ == "__main__":
    main()

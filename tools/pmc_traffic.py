#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/profile_default.sh into
profiles/pmc_hbm_bytes_per_launch.json (read by bench.py for roofline.traffic).

FETCH_SIZE / WRITE_SIZE are in KiB (MI355X_MICROARCH.md, HBM section). On gfx950 FETCH_SIZE
reports half of the bytes of a 16 B/lane stream; every load of these kernels is one, so fetch is
doubled. The correction is checked on expand_fp4_kernel, which reads the bit matrix exactly once
and writes the 4x larger FP4 shadow exactly once."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stormbitmaps_amd._lib import kernel_source_hash  # noqa: E402


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and r["Kernel_Name"].startswith(("storm::", "void storm::")):
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fetch_csv, write_csv, variant, rows, words, out = sys.argv[1:7]
    rows, words = int(rows), int(words)
    fetch = per_kernel(fetch_csv, "FETCH_SIZE")
    write = per_kernel(write_csv, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        if any(s in k for s in ("synth_fill", "column_identity", "fold_slots")):
            continue
        kernels[k] = {"fetch_bytes_corrected": fetch.get(k, 0.0) * 1024 * 2,
                      "write_bytes": write.get(k, 0.0) * 1024}
    dominant = max(kernels, key=lambda k: kernels[k]["fetch_bytes_corrected"])
    expand = next((k for k in kernels if "expand_fp4" in k), None)
    doc = json.load(open(out)) if len(sys.argv) > 7 and sys.argv[7] == "--merge" else {}
    doc["_about"] = ("HBM-side bytes per launch from rocprofv3 PMC (FETCH_SIZE, WRITE_SIZE; separate "
                     "passes; tools/profile_default.sh + tools/pmc_traffic.py), headline shape. FETCH_SIZE "
                     "is KiB and on gfx950 reports half of a 16 B/lane stream: doubled here, checked on "
                     "expand_fp4_kernel (reads the bit matrix once, writes the 4x FP4 shadow once).")
    entry = {"source_hash": kernel_source_hash(),  # bench.py ignores traffic measured on other sources
             "dominant_kernel": dominant,
             "hbm_bytes_per_launch": sum(kernels[dominant].values()),
             "all_kernels_bytes_per_launch": sum(sum(v.values()) for v in kernels.values()),
             "kernels": kernels}
    if expand:
        entry["calibration"] = {"expand_reads_expected": rows * words * 8,
                                "expand_reads_measured": kernels[expand]["fetch_bytes_corrected"],
                                "expand_writes_expected_min": rows * words * 32,
                                "expand_writes_measured": kernels[expand]["write_bytes"]}
    doc[f"variant{variant}"] = entry
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(entry, indent=1))


if __name__ == "__main__":
    main()

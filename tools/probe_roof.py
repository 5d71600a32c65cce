#!/usr/bin/env python3
"""Roofline summary of the list-probe kernel (K4) from the outputs of tools/profile_sparse.sh.

Unit of work: one streamed position = one 2-byte read from L2 (the items that read one chunk of the stream run on
one XCD at the same time) + one 2-byte LDS lookup of "how many of the item's 128 A rows list this position" + one
add (round 3, late: the popcount of the 16-byte entry is taken once per item, not once per lookup). Roofs per
MI355X_MICROARCH.md: 2- and 4-byte LDS reads are served 32 lanes per clock = 32 lookups/clk/CU x 256 CUs when the
32 lanes meet 32 banks; VALU: ~2 full-rate instructions per lookup (one address, half a three-input add, loop share) = 32 lookups/clk/CU; clock taken as
2.4 GHz (spec peak). Algorithmic lookups for the synthetic c4
container: per block column E = N x u listed positions (u = 65536 (1 - exp(-d / 65536)) unique positions per
block of d = load / 8 draws), N / 128 groups of A rows, every group meets the positions of the rows behind it
(far) and its own rows' positions (near)."""
import csv
import glob
import json
import math
import os
import sys

N, BLOCKS, ROWS_PER_ITEM = 10000, 8, 128
LDS_LOOKUPS_PER_CLK_CU, CUS, CLK = 32, 256, 2.4e9
# what the chip delivers for conflict-free 2-byte gathers from a 16 KiB table, 1024-thread workgroups, two per CU
# (tools/probes/lds_gather_roof.hip, profiles/r04_h_lds_gather_roof.jsonl); uniformly random positions: 5.5e12
MEASURED_GATHER_PEAK = 1.665e13


def counters(path):
    """Per pass: the kernel runs in two forms (far-only items, own-row items), one launch each — their means add."""
    acc = {}
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "probe_lists_kernel" in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], {}).setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(sum(v) / len(v) for v in forms.values()) for k, forms in acc.items()}


def kernel_us(path):
    total, calls = 0.0, 0
    for f in glob.glob(path + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "probe_lists_kernel" in r["Name"]:
                total += float(r["AverageNs"]) / 1e3
                calls = max(calls, int(r["Calls"]))
    return (total or None), calls


def main():
    out = sys.argv[1]
    wall = {}
    for l in open(os.path.join(out, "wall.jsonl")):
        if l.startswith("{"):
            d = json.loads(l)
            wall[d["load"]] = d
    print("load  mean_len  dense_ms  probe_ms  | kernel_us  lookups      lookups/s   frac of LDS roof | L2 hit  "
          "L2 req B/launch  alg B/launch  HBM fetch B | LDS insts  bank-conflict/active  wait_lds/wave")
    roof = LDS_LOOKUPS_PER_CLK_CU * CUS * CLK
    for load in sorted(wall):
        d = load / BLOCKS
        u = 65536.0 * (1.0 - math.exp(-d / 65536.0))
        e_col = N * u
        groups = (N + ROWS_PER_ITEM - 1) // ROWS_PER_ITEM
        far = sum(e_col * max(0, N - (g + 1) * ROWS_PER_ITEM) / N for g in range(groups))
        lookups = BLOCKS * (far + e_col)
        us, calls = kernel_us(os.path.join(out, f"trace_{load}"))
        w = wall[load]
        line = f"{load:6d} {u:8.0f} {w['dense_ms']:9.3f} {w['probe_ms']:9.3f}  |"
        if us:
            rate = lookups / (us * 1e-6)
            line += f" {us:9.1f}  {lookups:11.3e}  {rate:10.3e}  {rate / roof:8.3f} ({rate / MEASURED_GATHER_PEAK:5.3f})  |"
            sq = counters(os.path.join(out, f"sq_{load}"))
            tcc = counters(os.path.join(out, f"tcc_{load}"))
            fetch = counters(os.path.join(out, f"fetch_{load}"))
            if tcc:
                hit, miss, req = tcc.get("TCC_HIT_sum", 0), tcc.get("TCC_MISS_sum", 0), tcc.get("TCC_REQ_sum", 0)
                line += f" {hit / max(hit + miss, 1):6.3f}  {req * 128:14.3e}  {BLOCKS * far * 2:11.3e}"
            if fetch:
                line += f"  {fetch.get('FETCH_SIZE', 0) * 1024 * 2:10.3e} |"   # KiB units; x2 on gfx950 (16 B/lane streams)
            if sq:
                line += (f" {sq.get('SQ_INSTS_LDS', 0):9.3e}  {sq.get('SQ_LDS_BANK_CONFLICT', 0) / max(sq.get('SQ_LDS_IDX_ACTIVE', 1), 1):8.3f}"
                         f"  {sq.get('SQ_WAIT_INST_LDS', 0) / max(sq.get('SQ_WAVE_CYCLES', 1), 1):8.3f}")
        print(line)
    print(f"# measured conflict-free 2-byte gather peak (tools/probes/lds_gather_roof.hip): {MEASURED_GATHER_PEAK:.3e} lookups/s; uniformly random positions: 5.5e12")
    print(f"# LDS roof: {LDS_LOOKUPS_PER_CLK_CU} two-byte lookups/clk/CU x {CUS} CUs x {CLK / 1e9} GHz = {roof:.3e} lookups/s; VALU roof (2 instructions per lookup): {64 / 2 * CUS * CLK:.3e}")


if __name__ == "__main__":
    main()

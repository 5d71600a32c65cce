import csv, glob, collections, sys
d = sys.argv[1]
print("# probe_lists_kernel (bundle 1: one group of 128 rows per workgroup) against probe_lists_fat_kernel (bundle 4) at c4")
print("# (STORM_t, N = 10000, M = 524288), tools/profile_probe_bundle.sh: rocprofv3 --kernel-trace --stats for the durations,")
print("# a separate --pmc pass for the counters; averages over the launches of tools/bench_sparse_probe.py")
for B in (1, 4):
    for L in (104, 524, 20971):
        ks = glob.glob(f"{d}/trace_b{B}_{L}/*kernel_stats.csv")
        dur = None
        for r in csv.DictReader(open(ks[0])):
            if "probe_lists" in r["Name"]:
                dur = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
        acc = collections.defaultdict(list)
        for f in glob.glob(f"{d}/sq_b{B}_{L}/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if "probe_lists" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        c = {k: sum(v) / len(v) for k, v in acc.items()}
        print(f"bundle {B} draws {L:6d}: kernel {dur[0]:8.1f} us (x{dur[1]})  SQ_INSTS_LDS {c.get('SQ_INSTS_LDS', 0):.3g}  "
              f"SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE {c.get('SQ_LDS_BANK_CONFLICT', 0) / max(1, c.get('SQ_LDS_IDX_ACTIVE', 1)):.3f}  "
              f"SQ_LDS_IDX_ACTIVE {c.get('SQ_LDS_IDX_ACTIVE', 0):.3g}  SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES {c.get('SQ_WAIT_INST_LDS', 0) / max(1, c.get('SQ_WAVE_CYCLES', 1)):.3f}  "
              f"SQ_INSTS_VALU {c.get('SQ_INSTS_VALU', 0):.3g}  SQ_INSTS_VMEM_RD {c.get('SQ_INSTS_VMEM_RD', 0):.3g}  SQ_BUSY_CYCLES {c.get('SQ_BUSY_CYCLES', 0):.3g}")

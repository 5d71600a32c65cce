#!/usr/bin/env python3
"""Per-kernel durations out of a rocprofv3 rocpd database (t_results.db): kernel_times.py <db> [name filter]
-> one line per (kernel, grid): launches, min / median / mean us."""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
g = collections.defaultdict(list)
for name, grid, dt in db.execute("select name, grid_x, (end - start) from kernels order by start"):
    if flt in name:
        g[(name.split("(")[0][-48:], grid)].append(dt / 1e3)
for (name, grid), v in g.items():
    v.sort()
    print(f"{name:50s} grid {grid:8d} n {len(v):4d} min {v[0]:8.1f} med {v[len(v) // 2]:8.1f} mean {sum(v) / len(v):8.1f} us")

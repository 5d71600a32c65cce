#!/usr/bin/env python3
"""The longest HIP API calls of a rocprofv3 --hip-trace run (rocpd database): api_top.py <db> [n]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = "regions" if "regions" in tabs else [t for t in tabs if "region" in t][0]
cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
print(view, cols)
rows = list(db.execute(f"select name, start, end from {view} order by start"))
t_first = rows[0][1]
big = sorted(rows, key=lambda r: r[1] - r[2])[:n]
for name, s, e in sorted(big, key=lambda r: r[1]):
    print(f"{(s - t_first) / 1e6:10.3f} ms  +{(e - s) / 1e3:9.1f} us  {name}")

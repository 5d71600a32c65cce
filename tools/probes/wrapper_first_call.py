"""First / second / third STORM_wrapper_diag_blocked call (the caller's buffer travels every call) in a fresh process."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import stormbitmaps_amd as sb
from stormbitmaps_amd import synth
sb.load()
N, M = 10000, 65536
mat = synth.dense_matrix_c(M, N, M // 2, seed=42)
rec = {}
for k in range(4):
    t0 = time.perf_counter(); v = sb.wrapper_diag_blocked(mat, 31); rec[f"call{k}_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
rec["total"] = int(v)
print(json.dumps(rec))

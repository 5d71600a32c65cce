#!/usr/bin/env python3
"""Row pitch of the dense matrix against K2b's pass time: option k2_matrix_pad = number of 512-byte chunks added to a
pitch that is a multiple of 1 KiB. One JSON line per (shape, pad)."""
import json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stormbitmaps_amd as sb

pads = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,1,2,3,5,7").split(",")]
# shapes: "rows x bits x reps, ..." (default: the headline shape and c3)
shapes = [tuple(int(v) for v in t.split("x")) for t in (sys.argv[2] if len(sys.argv) > 2 else "10000x65536x40,10000x524288x10").split(",")]
for N, M, reps in shapes:
    for pad in pads:
        ctx = sb.HipContext(0)
        ctx.set_option("k2_matrix_pad", pad)
        m = ctx.matrix(N, M // 64)
        m.fill_synthetic(M, M // 2, seed=42)
        want = m.pairw()
        for _ in range(3):
            m.pairw()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); got = m.pairw(); ts.append(time.perf_counter() - t0)
        print(json.dumps({"rows": N, "bits": M, "pad_chunks": pad, "pitch_bytes": int(m.stride_words) * 8 if hasattr(m, "stride_words") else None,
                          "ms_best": round(min(ts) * 1e3, 4), "ms_median": round(sorted(ts)[len(ts) // 2] * 1e3, 4), "ok": got == want}), flush=True)
        m.close(); ctx.close()

import sys, json, itertools
sys.path.insert(0, "/root/repo")
import numpy as np, torch, stormbitmaps_amd as sb
ctx = sb.HipContext(0)
ctx.set_option("k2_strip_operands", 2)
bad = 0
shapes = [(2, 64), (3, 512), (63, 4096), (64, 4096), (65, 640), (200, 4096), (256, 4096), (257, 1000), (300, 65536),
          (511, 8192), (512, 65536), (513, 4160), (700, 65536), (1000, 30000), (1024, 65536), (1100, 4096), (1500, 12345),
          (2048, 65536), (2300, 20000), (3000, 65536), (4096, 16384)]
for N, M in shapes:
    W = (M + 63) // 64
    for draws in (M // 2, max(1, M // 50)):
        m = ctx.matrix(N, W); m.fill_synthetic(M, draws, seed=N + M)
        want = m.column_identity()
        got = m.pairw()
        got2 = m.pairw()
        parts = []
        for G in (2, 3, 5):
            parts.append(sum(m.pairw(r, G) for r in range(G)))
        info = ctx.last_launch_info()
        ok = got == want and got2 == want and all(p == want for p in parts)
        bad += not ok
        print(N, M, draws, "OK" if ok else f"FAIL got {got} {got2} parts {parts} want {want}", info, flush=True)
        m.close()
for gpc in (1, 2, 3):
    for mp in (1, 8, 40):
        ctx.set_option("k2_stream_groups_per_cu", gpc); ctx.set_option("k2_stream_min_piece", mp)
        for N, M in ((1024, 65536), (2300, 20000), (777, 9999)):
            m = ctx.matrix(N, (M + 63) // 64); m.fill_synthetic(M, M // 3, seed=5)
            want = m.column_identity(); got = m.pairw()
            ok = got == want; bad += not ok
            print("opts", gpc, mp, N, M, "OK" if ok else f"FAIL {got} {want}", ctx.last_launch_info(), flush=True)
            m.close()
print("BAD", bad)

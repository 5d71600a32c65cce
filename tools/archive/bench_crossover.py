#!/usr/bin/env python3
"""Where does the matrix-core path overtake the popcount kernel? Pass time (HIP events, data
resident) of variants 2 and 4 over a grid of small shapes; basis of the `auto` rule."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import stormbitmaps_amd as sb
    stream = torch.cuda.current_stream()
    ctx = sb.HipContext(0, stream.cuda_stream)
    total_t = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    for M in (4096, 65536, 524288):
        for N in (64, 128, 256, 384, 512, 768, 1024, 1536, 2048):
            m = ctx.matrix(N, M // 64)
            m.fill_synthetic(M, M // 2, seed=42)
            want = m.column_identity()
            row = {"bits": M, "rows": N}
            for variant in (2, 4):
                ctx.set_option("variant", variant)
                for _ in range(3):
                    m.pairw_launch(total_t.data_ptr(), 0, 1)
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
                for _ in range(50):
                    m.pairw_launch(total_t.data_ptr(), 0, 1)
                b.record(stream)
                torch.cuda.synchronize()
                assert int(total_t.item()) == want, (M, N, variant)
                row[f"v{variant}_us"] = round(a.elapsed_time(b) / 50 * 1e3, 1)
            ctx.set_option("variant", -1)
            print(json.dumps(row), flush=True)
            m.close()


if __name__ == "__main__":
    main()

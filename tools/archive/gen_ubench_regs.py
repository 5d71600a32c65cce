#!/usr/bin/env python3
"""Generates tools/ubench_regs.hip: hand-numbered VGPR streams that mimic the dense kernel's
inner loop (32 A rows x {and, bcnt, and, bcnt}) to find out what the VALU charges for
register-bank placement, temp reuse and instruction grouping on gfx950."""

A0 = 32          # a_lo[r] = v[A0+2r], a_hi[r] = v[A0+2r+1]
ACC = [4, 5, 6, 7]


def stream(b_lo, b_hi, temps, grouped, a_swap=False, n_acc=4, split_hi=False, n_a=64, a0=None):
    """One B word against 32 A rows. temps: list of temp regs cycled; grouped: issue `grouped`
    ands back-to-back before their bcnts (1 = alternate like hipcc emits)."""
    ops = []
    for r in range(32):
        base = A0 if a0 is None else a0
        lo, hi = base + (2 * r) % n_a, base + (2 * r + 1) % n_a
        if a_swap:
            lo, hi = hi, lo
        accs = list(range(4, 4 + n_acc)) if n_acc <= 4 else [4, 5, 6, 7] + list(range(12, 12 + n_acc - 4))
        if split_hi:   # lo and hi halves feed different accumulators
            ops.append((lo, b_lo, accs[(2 * r) % n_acc]))
            ops.append((hi, b_hi, accs[(2 * r + 1) % n_acc]))
        else:
            ops.append((lo, b_lo, accs[r % n_acc]))
            ops.append((hi, b_hi, accs[r % n_acc]))
    out = []
    for g in range(0, len(ops), grouped):
        grp = ops[g:g + grouped]
        for k, (a, b, acc) in enumerate(grp):
            out.append(f"v_and_b32 v{temps[(g + k) % len(temps)]}, v{a}, v{b}")
        for k, (a, b, acc) in enumerate(grp):
            out.append(f"v_bcnt_u32_b32 v{acc}, v{temps[(g + k) % len(temps)]}, v{acc}")
    return out


def and_only():
    return [f"v_and_b32 v{100 + (r & 7)}, v{A0 + r}, v{8 + (r & 1)}" for r in range(64)] * 2


def bcnt_only(n_acc):
    accs = [4, 5, 6, 7] + list(range(12, 28))
    return [f"v_bcnt_u32_b32 v{accs[r % n_acc]}, v{A0 + (r % 64)}, v{accs[r % n_acc]}" for r in range(128)]


VARIANTS = {
    # name: (b_lo, b_hi, temps, grouped, a_swap)
    "kernel_like":      (8, 9, [99], 1, False),          # even&even / odd&odd operands, one temp
    "cross_parity":     (8, 9, [99], 1, True),           # even&odd operands
    "b_bank_1_3":       (9, 11, [99], 1, False),         # a even (banks 0,2) with b odd (1,3)
    "temps4":           (8, 9, [100, 101, 102, 103], 1, False),
    "grouped4":         (8, 9, [100, 101, 102, 103], 4, False),
    "grouped8":         (8, 9, [100, 101, 102, 103, 104, 105, 106, 107], 8, False),
    "grouped8_cross":   (9, 8, [100, 101, 102, 103, 104, 105, 106, 107], 8, False),
    "acc8_split":       (8, 9, [100, 101, 102, 103], 1, False, 8, True),
    "acc16_split":      (8, 9, [100, 101, 102, 103], 1, False, 16, True),
    "acc16_grouped8":   (8, 9, [100, 101, 102, 103, 104, 105, 106, 107], 8, False, 16, True),
    "acc16_grouped16":  (8, 9, list(range(100, 116)), 16, False, 16, True),
}
VARIANTS.update({
    "a_regs_8":   (8, 9, [100, 101, 102, 103], 1, False, 4, False, 8),
    "a_regs_16":  (8, 9, [100, 101, 102, 103], 1, False, 4, False, 16),
    "a_regs_32":  (8, 9, [100, 101, 102, 103], 1, False, 4, False, 32),
    "a_regs_2":   (8, 9, [100, 101, 102, 103], 1, False, 4, False, 2),
})
def shifted(lines):
    return ["s_nop 0"] + lines


def e64(lines):
    return [l.replace("v_and_b32 ", "v_and_b32_e64 ") for l in lines]


def nop_between(lines):
    out = []
    for l in lines:
        if l.startswith("v_bcnt"):
            out.append("s_nop 0")
        out.append(l)
    return out


def nop_after_bcnt(lines):
    out = []
    for l in lines:
        out.append(l)
        if l.startswith("v_bcnt"):
            out.append("s_nop 0")
    return out


def nop_every(lines, n, what="s_nop 0"):
    out = []
    for i, l in enumerate(lines):
        if i % n == 0:
            out.append(what)
        out.append(l)
    return out


def nop_before_and(lines):
    out = []
    for l in lines:
        if l.startswith("v_and"):
            out.append("s_nop 0")
        out.append(l)
    return out


T8 = [100, 101, 102, 103, 104, 105, 106, 107]
RAW0 = {
    "kl_nop_after_bcnt": nop_after_bcnt(stream(8, 9, [99], 1)),
    "kl_nop_every4":     nop_every(stream(8, 9, [99], 1), 4),
    "kl_nop_every8":     nop_every(stream(8, 9, [99], 1), 8),
    "kl_nop_both":       nop_before_and(nop_between(stream(8, 9, [99], 1))),
    "g2_nop_every2":     nop_every(stream(8, 9, T8, 2), 2),
    "g2_nop_every4":     nop_every(stream(8, 9, T8, 2), 4),
    "g8_nop_every8":     nop_every(stream(8, 9, T8, 8), 8),
    "kl_e64_nop_betw":   nop_between(e64(stream(8, 9, [99], 1))),
    "kl_snop1_between":  [l.replace("s_nop 0", "s_nop 1") for l in nop_between(stream(8, 9, [99], 1))],
    "kl_sleep_every8":   nop_every(stream(8, 9, [99], 1), 8, "s_sleep 0"),
    "kl_shift4":       shifted(stream(8, 9, [99], 1)),
    "kl_and_e64":      e64(stream(8, 9, [99], 1)),
    "kl_nop_between":  nop_between(stream(8, 9, [99], 1)),
    "g2":              stream(8, 9, T8, 2),
    "g2_shift4":       shifted(stream(8, 9, T8, 2)),
    "g8_shift4":       shifted(stream(8, 9, T8, 8)),
    "g8_and_e64":      e64(stream(8, 9, T8, 8)),
    "g16_shift4":      shifted(stream(8, 9, list(range(100, 116)), 16, False, 16, True)),
}
RAW = {"and_only_64src": and_only(), "bcnt_only_acc4": bcnt_only(4), "bcnt_only_acc16": bcnt_only(16)}

HDR = r'''// GENERATED by tools/gen_ubench_regs.py — do not edit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
'''

CLOB = ", ".join(f'"v{i}"' for i in list(range(4, 28)) + list(range(32, 116)))


def kernel(name, lines):
    init = [f"v_mov_b32 v{i}, 0" for i in ACC + list(range(12, 28))] + \
           [f"v_mov_b32 v{i}, 0x{(0x9E3779B9 * (i + 1)) & 0xFFFFFFFF:x}" for i in list(range(8, 12)) + list(range(32, 96))]
    body = "\\n\\t".join(lines)
    ini = "\\n\\t".join(init)
    return f'''
__global__ __launch_bounds__(256, 4) void k_{name}(unsigned* out, int iters) {{
    asm volatile("{ini}" ::: {CLOB});
    for (int it = 0; it < iters; ++it) {{
        asm volatile("{body}" ::: {CLOB});
    }}
    unsigned s;
    asm volatile("v_add3_u32 %0, v4, v5, v6\\n\\tv_add_u32 %0, %0, v7" : "=v"(s) :: {CLOB});
    if (s == 0xDEADBEEF) out[0] = s;
}}
'''


def main():
    src = HDR
    for name, args in VARIANTS.items():
        src += kernel(name, stream(*args))
    RAW.update(RAW0)
    for name, lines in RAW.items():
        src += kernel(name, lines)
    src += r'''
typedef void (*kfn)(unsigned*, int);
static int run(const char* name, kfn f, int blocks, int iters, unsigned* d) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, d, iters / 10 + 1);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, d, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double instr = (double)blocks * 4 * (double)iters * 128.0;  // VALU wave-instructions
    printf("%-16s %.3f ms  %.3e word-pairs/s  %.3f ns per wave-instr per SIMD\n", name, ms,
           instr * 64 / 4 / (ms * 1e-3), ms * 1e6 / (instr / 1024.0));
    return 0;
}
int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 4;
    const int iters = argc > 2 ? atoi(argv[2]) : 4000;
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int blocks = p.multiProcessorCount * wps;
    unsigned* d; CHECK(hipMalloc(&d, 64));
    printf("waves/SIMD=%d iters=%d\n", wps, iters);
'''
    for name in list(VARIANTS) + list(RAW):
        src += f'    if (run("{name}", k_{name}, blocks, iters, d)) return 1;\n'
    src += "    return 0;\n}\n"
    open(__file__.replace("gen_ubench_regs.py", "ubench_regs.hip"), "w").write(src)


if __name__ == "__main__":
    main()

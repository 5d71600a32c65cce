#!/usr/bin/env python3
"""Clock and socket-power telemetry behind the "power-limited" diagnosis.

Three witnesses per kernel, over loops of >= 2 s of back-to-back launches each:
  * sysfs of the card the workers run on (found by PCI bus id): hwmon freq1_input (sclk) and power1_input (socket
    power), power1_cap — sampled every ~2 ms by a child process that starts BEFORE anything touches the GPU and
    never touches it itself;
  * the in-kernel clock: shader-clock ticks / 100 MHz ticks over every workgroup's lifetime (s_memtime /
    s_memrealtime; MI355X_MICROARCH.md "DVFS give-back" (6): board power and pp_dpm_sclk are not the test) — from
    the tools build of the library (STORM_HIP_LIB=stormbitmaps_amd/libstorm_hip_probes.so: in the shipped library no
    stamp executes) and from tools/probes/mfma_power_roof;
  * wall time per launch in the first and in the last quarter of the loop (a clock that sinks shows here).
The driver process starts the sampler and one worker process per kernel, in turn, with a pause between them; it
never initialises the GPU. Output: one JSON object per kernel (stdout), e.g. profiles/r05_*_clock_power.jsonl.

  python tools/clock_power.py [--seconds 3] [--kernels k2b,tile,mfma16_onehot,mfma16_random,mfma32_onehot,mfma32_random]
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------- sampler (child, no GPU)
def sampler(path, period):
    cards = []
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        hw = glob.glob(dev + "/hwmon/hwmon*")
        if not hw:
            continue
        try:
            bus = os.path.basename(os.path.realpath(dev))
            cap = int(open(hw[0] + "/power1_cap").read())
        except OSError:
            continue
        cards.append((bus, hw[0] + "/freq1_input", hw[0] + "/power1_input", cap))
    with open(path, "w") as f:
        f.write(json.dumps({"cards": [{"bus": c[0], "power_cap_w": c[3] / 1e6} for c in cards]}) + "\n")
        f.flush()
        while True:
            t = time.monotonic()
            row = [round(t, 6)]
            for _, fq, pw, _ in cards:
                try:
                    row.append(int(open(fq).read()) // 1000000)
                    row.append(int(open(pw).read()) // 1000000)
                except (OSError, ValueError):
                    row.extend((-1, -1))
            f.write(json.dumps(row) + "\n")
            f.flush()
            time.sleep(period)


# ---------------------------------------------------------------- workers (children, GPU)
def worker(kind, seconds):
    sys.path.insert(0, ROOT)
    import ctypes as C

    import torch

    import stormbitmaps_amd as sb
    from stormbitmaps_amd import _lib
    lib = _lib.load()
    ctx = sb.HipContext(0)
    N, M = 10000, 65536
    m = ctx.matrix(N, M // 64)
    m.fill_synthetic(M, M // 2, seed=42)
    hip = C.CDLL("libamdhip64.so")
    buf = C.create_string_buffer(64)
    hip.hipDeviceGetPCIBusId(buf, 64, 0)
    bus = buf.value.decode().lower()
    probe = getattr(lib, "storm_hip_probe_clock", None) if ctx.get_option("probes_build") == 1 else None
    out = (C.c_uint64 * 3)()
    if kind == "k2b":
        d_total = torch.zeros(1, dtype=torch.int64, device="cuda:0")
        call = lambda: m.pairw_launch(d_total.data_ptr())
        name = "storm::strip16_bits_kernel (+ fold_slots_kernel), whole pass"
        flop = N * (N - 1) // 2 * (M // 64) * 128.0
    elif kind == "k2h":      # [r6] the per-pair matrix of 4096 rows: tile128_kernel
        m.close()
        N = 4096
        m = ctx.matrix(N, M // 64)
        m.fill_synthetic(M, M // 2, seed=42)
        ctx.set_option("k2_tile_shape", 6)
        dst = torch.zeros((N, N), dtype=torch.int32, device="cuda:0")
        call = lambda: m.pairw_matrix_device(dst.data_ptr(), N, "and")
        name = "storm::tile128_kernel (per-pair matrix of 4096 rows), whole call"
        flop = N * (N - 1) // 2 * (M // 64) * 128.0
    elif kind == "ring":     # [r6] the c2 triangle on the ring kernel
        ctx.set_option("k2_tile_shape", 5)
        dst = torch.zeros((N, N), dtype=torch.int32, device="cuda:0")
        call = lambda: m.pairw_matrix_device(dst.data_ptr(), N, "and")
        name = "storm::tilering_kernel (materialised upper triangle at c2), whole call"
        flop = N * (N - 1) // 2 * (M // 64) * 128.0
    elif kind == "k5":       # [r6] the window kernel of the row lists at the c4 shape, 2096 positions per row
        s = sb.Storm()
        s.add_synthetic(524288, N, 2096, seed=42)
        lib.STORM_hip_set_option(b"matrix_lists", 1)
        lib.STORM_hip_set_option(b"matrix_lists_kernel", 1)
        dst = torch.zeros((N, N), dtype=torch.int32, device="cuda:0")
        call = lambda: s.pairw_matrix_device(dst.data_ptr(), N, N)
        name = "lists_matrix_kernel (K5, c4 shape, 2096 positions per row), whole call"
        flop = 0.0
    else:
        dst = torch.zeros((N, N), dtype=torch.int32, device="cuda:0")
        call = lambda: m.pairw_matrix_device(dst.data_ptr(), N, "and")
        name = "storm::tilebits8_kernel (materialised upper triangle), whole call"
        flop = N * (N - 1) // 2 * (M // 64) * 128.0
    for _ in range(3):
        call()
    ctx.synchronize()
    torch.cuda.synchronize()
    if probe:
        probe(out)
    ts = []
    t_begin = time.monotonic()
    while time.monotonic() - t_begin < seconds:
        t0 = time.perf_counter()
        for _ in range(16):
            call()
        ctx.synchronize()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 16 * 1e3)
    t_end = time.monotonic()
    clock = None
    wgs = 0
    if probe and probe(out) == 0 and out[1]:
        clock = 100.0 * out[0] / out[1]
        wgs = int(out[2])
    q = max(1, len(ts) // 4)
    last = sum(ts[-q:]) / q
    print(json.dumps({"kernel": name, "pci_bus": bus, "mono_begin": t_begin, "mono_end": t_end, "launches": len(ts) * 16,
                      "ms_first_quarter": round(sum(ts[:q]) / q, 4), "ms_last_quarter": round(last, 4),
                      "frac_of_10_pflops_last_quarter": round(flop / (last * 1e-3) / 1e16, 4),
                      "in_kernel_clock_mhz": round(clock, 1) if clock else None, "workgroups_stamped": wgs,
                      "library": os.environ.get("STORM_HIP_LIB", "stormbitmaps_amd/libstorm_hip.so")}))
    m.close()
    ctx.close()


# ---------------------------------------------------------------- driver (no GPU)
def summarise(rec, samples, cards):
    bus = (rec.get("pci_bus") or "").lower()
    idx = next((i for i, c in enumerate(cards) if c["bus"].lower() == bus), None)
    if idx is None:
        rec["sysfs"] = "card %s not among %s" % (bus, [c["bus"] for c in cards])
        return rec
    # the settled part of the loop: its second half
    t0 = rec["mono_begin"] + 0.5 * (rec["mono_end"] - rec["mono_begin"])
    sel = [(r[1 + 2 * idx], r[2 + 2 * idx]) for r in samples if t0 <= r[0] <= rec["mono_end"] and r[1 + 2 * idx] >= 0]
    idle = [(r[1 + 2 * idx], r[2 + 2 * idx]) for r in samples
            if rec["mono_begin"] - 1.2 <= r[0] <= rec["mono_begin"] - 0.7 and r[1 + 2 * idx] >= 0]
    if sel:
        f = sorted(s[0] for s in sel)
        p = sorted(s[1] for s in sel)
        rec["sysfs"] = {"samples": len(sel), "sclk_mhz_median": f[len(f) // 2], "sclk_mhz_min": f[0], "sclk_mhz_max": f[-1],
                        "socket_power_w_median": p[len(p) // 2], "socket_power_w_max": p[-1],
                        "power_cap_w": cards[idx]["power_cap_w"],
                        "idle_before_w": (sorted(s[1] for s in idle)[len(idle) // 2] if idle else None)}
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--kernels", default="k2b,k2b_shipped,tile,tile_shipped,mfma16_zeros,mfma16_onehot,mfma16_random,mfma32_onehot,mfma32_random")
    ap.add_argument("--samples", default=os.path.join(ROOT, "gpurun_out", "clock_power_samples.jsonl"))
    ap.add_argument("--worker", default=None)
    ap.add_argument("--sampler", default=None)
    args = ap.parse_args()
    if args.sampler:
        return sampler(args.sampler, 0.002)
    if args.worker:
        return worker(args.worker, args.seconds)

    os.makedirs(os.path.dirname(args.samples), exist_ok=True)
    samp = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--sampler", args.samples])
    time.sleep(1.0)
    recs = []
    probes_lib = os.path.join(ROOT, "stormbitmaps_amd", "libstorm_hip_probes.so")
    roof = os.path.join(ROOT, "tools", "probes", "mfma_power_roof")
    try:
        for k in args.kernels.split(","):
            env = dict(os.environ)
            if k in ("k2h", "ring", "k5"):            # [r6] shipped library: tile128_kernel at 4096 rows, tilering_kernel at c2, K5 at 2096 per row
                cmd = [sys.executable, os.path.abspath(__file__), "--worker", k, "--seconds", str(args.seconds)]
            elif k.split("_")[0] in ("k2b", "tile"):   # k2b_shipped / tile_shipped: the shipped library (no stamps, no in-kernel clock)
                if os.path.exists(probes_lib) and not k.endswith("_shipped"):
                    env["STORM_HIP_LIB"] = probes_lib
                cmd = [sys.executable, os.path.abspath(__file__), "--worker", k.split("_")[0], "--seconds", str(args.seconds)]
            elif k.startswith("ring_"):   # tools/probes/tile_ring, ablation mask behind the underscore
                cmd = [os.path.join(ROOT, "tools", "probes", "tile_ring"), "10000", "65536", "3", k[5:], str(args.seconds)]
            else:
                shape, data = k[4:6], {"onehot": "2", "random": "4", "zeros": "0"}[k.split("_")[1]]
                cmd = [roof, data, shape, str(args.seconds)]
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
            line = next((l for l in r.stdout.splitlines() if l.startswith("{")), None)
            if r.returncode != 0 or not line:
                print(json.dumps({"kernel": k, "error": (r.stderr or r.stdout)[-400:]}), flush=True)
                continue
            rec = json.loads(line)
            rec["label"] = k
            recs.append(rec)
            time.sleep(2.0)   # the chip idles between kernels: each loop starts from a cool socket
    finally:
        samp.terminate()
        samp.wait()
    rows = [json.loads(l) for l in open(args.samples)]
    cards, samples = rows[0]["cards"], rows[1:]
    for rec in recs:
        print(json.dumps(summarise(rec, samples, cards)), flush=True)


if __name__ == "__main__":
    main()

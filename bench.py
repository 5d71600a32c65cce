#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native StormBitmaps hot path.

Metric (BASELINE.json): 64-bit bitmap words/s for the XX^T upper-triangle pairwise
AND+popcount on STORM_contiguous_t (N=10000 rows x M=65536 bits, dense: 32768 draws per row),
the reference's README headline `benchmark 65536 10000` (benchmark.cpp:906-918).

    words/s = N(N-1)/2 * 2 * ceil(M/64) / t          (benchmark.cpp:128-129)

One step = one full all-pairs pass with the matrix already resident in HBM: the dense kernel
over this rank's shard of the work + the 8-byte all-reduce of the partial totals (issued async on
RCCL's stream, so it overlaps the next step's pass; all of them complete inside the timed region).
At --gpus N>1 (launched by torch.distributed.run, one rank per GPU, RCCL) the SAME total work
is sharded over the ranks (strong scaling); `value` is whole-job words/s.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_WORDPAIRS = 256 * 4 * 32 * 2.4e9 / 4  # 4 VALU lane-ops per 64-bit word pair
FP4_PEAK_TFLOPS = 10000.0  # MI355X_MICROARCH.md: FP6/FP4 MFMA ~10 PF dense
FP4_MEASURED_WORDPAIRS = 7.48e13  # tools/ubench_shape: bare v_mfma_scale_f32_16x16x128_f8f6f4 loop, random 0/1 operands (9.57 PFLOP/s)


def cpu_baseline(head_rows_fn, n_words, budget_s=12.0):
    """The CPU restatement of the reference path (oracle/, kind "port"), one thread like the
    reference, blocked loop with the harness's block size (benchmark.cpp:823-824), best SIMD
    leaf the host has — timed on a bounded row sample of the SAME matrix."""
    from tests._orc import Oracle  # oracle use is confined to this baseline leg
    orc = Oracle()
    kind = orc.lib.orc_best_leaf_kind()
    bsize = max(5, int(256e3 // (n_words * 8)))
    probe = head_rows_fn(600)
    secs, _ = orc.time_blocked(probe, kind, bsize)
    rate = (600 * 599 // 2) * 2 * n_words / max(secs, 1e-9)
    # rows whose pair count fits the time budget
    n = int(min(10000, max(600, (2 * budget_s * rate / (2 * n_words)) ** 0.5)))
    sample = head_rows_fn(n)
    n = int(sample.shape[0])  # the matrix may have fewer rows than the budget allows
    secs, total = orc.time_blocked(sample, kind, bsize)
    words = (n * (n - 1) // 2) * 2 * n_words
    # SURVEY §8d also asks for the plain scalar leaf (popcnt per word): a smaller row sample
    ns = min(n, 2500)
    s_secs, _ = orc.time_blocked(sample[:ns], 0, bsize)
    scalar = {"value": (ns * (ns - 1) // 2) * 2 * n_words / max(s_secs, 1e-9), "leaf": orc.leaf_name(0),
              "seconds": round(s_secs, 3), "sample": f"first {ns} rows"}
    return {"value": words / secs, "unit": "words/s", "cores": 1, "kind": "port", "scalar_leaf": scalar,
            "leaf": orc.leaf_name(kind), "seconds": round(secs, 3),
            "sample": f"first {n} rows of the benchmark matrix ({n * (n - 1) // 2} pairs), "
                      f"orc_wrapper_diag_blocked bsize={bsize}",
            "host_cpus": os.cpu_count(), "sample_total": total, "sample_rows": n}


def mat_head_total(ctx, mat, n, n_words):
    """GPU all-pairs total over the first n rows of the benchmark matrix (device-to-device copy of
    the head rows into a second matrix): what a CPU sample smaller than the matrix is compared with."""
    head = ctx.matrix(n, n_words)
    try:
        head.import_device(mat.device_ptr, n, mat.stride_words)
        return head.pairw()
    finally:
        head.close()


def free_port():
    import socket
    with socket.socket() as sock:   # fixed ports collide with concurrent runs and TIME_WAIT
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def launch_ranks(args, real_stdout):
    """`python3 bench.py --gpus N` without RANK in the environment: run
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>`
    as a child (never os.exec*), one rank per GPU, rendezvous on 127.0.0.1 at a free port. Rank 0's
    JSON line is the child's whole stdout; it is passed on unchanged. Returns the exit code."""
    import subprocess
    if os.environ.get("ROCP_TOOL_LIBRARIES") or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        # a profiler's preloaded library has initialised the GPU in THIS process already
        print("bench.py: refusing to start ranks from a profiled process; profile at --gpus 1 or put "
              "torch.distributed.run outside the profiler", file=sys.stderr)
        return 2
    port = args.master_port or free_port()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: starting", args.gpus, "ranks:", " ".join(cmd), file=sys.stderr)
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, env=env, cwd=ROOT)
    out, _ = child.communicate()
    lines = [l for l in out.decode(errors="replace").splitlines() if l.lstrip().startswith("{")]
    rc = child.returncode
    if lines:
        os.write(real_stdout, (lines[-1] + "\n").encode())
    elif rc == 0:
        print("bench.py: the ranks exited 0 without a JSON line", file=sys.stderr)
        rc = 1
    return rc


def main():
    # The contract is ONE JSON line on stdout. RCCL prints a version banner on stdout at
    # communicator creation, so everything else is sent to stderr and the JSON line is written
    # to the saved stdout at the end.
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=10000)
    ap.add_argument("--bits", type=int, default=65536)
    ap.add_argument("--draws", type=int, default=0, help="0 = bits/2 (dense)")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--variant", type=int, default=-1)
    ap.add_argument("--prewarm-ms", type=float, default=150.0,
                    help="untimed passes for this long before the warmup steps (clock ramp)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-shadow-resident", action="store_true")
    ap.add_argument("--opt", action="append", default=[], help="ctx option key=value (tuning)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, default) | gloo (rehearsal)")
    ap.add_argument("--all-on-device0", action="store_true",
                    help="rehearsal on a 1-GPU box: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--master-port", type=int, default=0,
                    help="rendezvous port when bench.py starts its own ranks (0 = a free one)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # Called bare (`python3 bench.py --gpus N`), not under a launcher: start the N ranks as CHILD
        # processes — decided here, before torch is imported or anything touches a GPU (a process
        # that has initialised HIP must never be replaced or forked) — and relay the job's single
        # JSON line and exit code.
        sys.exit(launch_ranks(args, real_stdout))

    import torch
    import torch.distributed as dist
    import stormbitmaps_amd as sb
    from stormbitmaps_amd import dist as sdist

    rank, world, local_rank = sdist.rank_world()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run "
                             "(one rank per GPU); see the module docstring")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("no GPU visible: the hot path has no CPU fallback")
    if args.all_on_device0:
        if args.backend == "nccl":
            raise SystemExit("--all-on-device0 needs --backend gloo (RCCL wants one GPU per rank)")
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    under_launcher = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    if world > 1 or under_launcher:
        sdist.init_process_group(args.backend)  # "nccl" = RCCL over xGMI

    N, M = args.rows, args.bits
    draws = args.draws or M // 2
    W = (M + 63) // 64
    stream = torch.cuda.current_stream(dev)
    ctx = sb.HipContext(local_rank, stream.cuda_stream)
    if args.variant >= 0:
        ctx.set_option("variant", args.variant)
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    mat = ctx.matrix(N, W)
    mat.fill_synthetic(M, draws, seed=args.seed)  # resident in HBM before any timing
    # Two result words used in turn: the 8-byte all-reduce of step i runs on RCCL's own stream while
    # the pass of step i+1 is already being computed (async_op); a word is only rewritten after the
    # all-reduce that last used it has been ordered in front of the compute stream. Every step still
    # carries its collective, and all of them have completed before the timed region's closing fence.
    totals = torch.zeros(2, dtype=torch.int64, device=dev)
    barrier_word = torch.zeros(1, dtype=torch.int64, device=dev)
    pending = [None, None]
    state = {"n": 0}

    collective = dist.is_initialized()

    def launch_pass():
        k = state["n"] & 1
        if pending[k] is not None:
            pending[k].wait()   # stream-level: orders the compute stream behind that collective
            pending[k] = None
        mat.pairw_launch(totals[k:].data_ptr(), rank, world)
        return k

    def reduce_total(k):
        if not collective:
            return
        if args.backend == "gloo":  # CPU transport (rehearsal only)
            host = totals[k:k + 1].cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            totals[k:k + 1].copy_(host)
        else:
            pending[k] = dist.all_reduce(totals[k:k + 1], op=dist.ReduceOp.SUM, async_op=True)

    def step():
        k = launch_pass()
        reduce_total(k)
        state["n"] += 1

    def last_total():
        return int(totals[(state["n"] - 1) & 1].item())

    def fence():
        for k in (0, 1):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None
        if collective:
            # barrier: no rank gets past it before every rank has reached it. With RCCL it is an
            # all-reduce on a resident word followed by the ONE host wait below (dist.barrier()
            # makes its own tensor and its own host wait: a longer idle gap in front of the timed
            # steps, and idle time is what costs clock: profiles/r02_k_bench_nccl_ramp.txt)
            if args.backend == "gloo":
                dist.barrier()
            else:
                dist.all_reduce(barrier_word, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize(dev)

    # The chip needs ~30 passes (~30 ms) from idle to settle (tools/archive/bench_ramp.py: passes 3..25
    # run 1.17 -> 0.92 ms, steady 0.88): an untimed pre-warm in front of the W warmup steps makes
    # the figure independent of how small W is. Same work as a step, results discarded.
    # Every rank must run the SAME number of steps (each step carries a collective).
    # One calibration batch, then ONE agreement on how many more passes make up the pre-warm (the
    # slowest rank's estimate), then those passes back to back with no host round trip in between: a
    # synchronise + all-reduce + .item() after every batch leaves the GPU idle each time, and a few
    # milliseconds of idle cost ~10 % on the next 10-20 passes (profiles/r02_k_bench_nccl_ramp.txt).
    # (the very first step pays for the RCCL communicator — seconds —, the shadow allocation and the
    #  work list: it must not be part of the calibration, or the pre-warm collapses to nothing; that is
    #  what made every N > 1 run of round 1's bench.py start its timed steps on a cold chip)
    step()
    fence()
    t_pre = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize(dev)
    per_step_ms = (time.perf_counter() - t_pre) * 1e3 / 10
    n_pre = max(0, int(args.prewarm_ms / max(per_step_ms, 1e-3)) - 10)
    if collective:
        w = torch.tensor([n_pre], dtype=torch.int64, device=dev if args.backend != "gloo" else "cpu")
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        n_pre = int(w.item())
    for _ in range(n_pre):
        step()
    for _ in range(args.warmup):
        step()
    fence()
    # dominant kernel alone: the library brackets it with HIP events on the launch stream
    ctx.set_option("time_kernels", 1)
    # The timed region carries exactly the contract's instrumentation: the library's two HIP events
    # around the dominant kernel on the launch stream (roofline.achieved). An event record costs the
    # stream ~4 us (measured: 8.7 us per step for a second pair around the whole pass), so the
    # per-pass durations are taken in a separate, untimed loop below.
    # At N > 1 a rank's pass is 1/N as long and the two event records (~8 us) are not: there the kernel is
    # bracketed on every 4th step only (still live, still inside the timed region; `kernel_ms_samples`).
    sample_every = 1 if world == 1 else 4
    t0 = time.perf_counter()
    host_us = []
    for i in range(args.steps):
        th = time.perf_counter()
        if sample_every > 1 and i % sample_every < 2:
            ctx.set_option("time_kernels", 0 if i % sample_every else (2 if i else 1))  # 2: resume the series
        k = launch_pass()
        reduce_total(k)
        state["n"] += 1
        host_us.append((time.perf_counter() - th) * 1e6)
    fence()
    elapsed = time.perf_counter() - t0
    my_elapsed = elapsed
    if collective:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend != "gloo" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total = last_total()
    dom_sum_ms, dom_n = ctx.kernel_time()
    ctx.set_option("time_kernels", 0)
    # untimed diagnostic: expand + dominant kernel + fold of single passes, bracketed by stream events
    # (same collective count on every rank: min(steps, 24) more steps)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(min(args.steps, 24))]
    for a, b in ev:
        a.record(stream)
        k = launch_pass()
        b.record(stream)
        reduce_total(k)
        state["n"] += 1
    fence()
    per_step_pass_ms = [a.elapsed_time(b) for a, b in ev]
    launch_ms = sum(per_step_pass_ms) / len(ev)
    kernel_ms = dom_sum_ms / dom_n if dom_n else launch_ms

    # N > 1 diagnostics (outside the timed region): the 8-byte all-reduce alone, and every rank's
    # own kernel / pass / wall time, so that a scaling run can be read rank by rank
    per_rank = None
    if collective:
        n_ar = 20
        scratch = torch.zeros(1, dtype=torch.int64, device=dev if args.backend != "gloo" else "cpu")
        dist.all_reduce(scratch); fence()
        t_ar = time.perf_counter()
        for _ in range(n_ar):
            dist.all_reduce(scratch, op=dist.ReduceOp.SUM)   # blocking form: latency of one collective
        torch.cuda.synchronize(dev)
        allreduce_us = (time.perf_counter() - t_ar) * 1e6 / n_ar
        # the same steps WITHOUT their collective (untimed diagnostic; every rank runs the same count): what the
        # all-reduce adds to a step is wall_ms_per_step - wall_ms_per_step_no_collective, rank by rank
        fence()
        t_nc = time.perf_counter()
        for _ in range(args.steps):
            launch_pass()
            state["n"] += 1
        torch.cuda.synchronize(dev)
        no_coll_ms = (time.perf_counter() - t_nc) * 1e3 / args.steps
        fence()
        mine = torch.tensor([kernel_ms, launch_ms, my_elapsed * 1e3 / args.steps, allreduce_us,
                             float(ctx.last_launch_info()["items"]), no_coll_ms], dtype=torch.float64,
                            device=dev if args.backend != "gloo" else "cpu")
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        # which physical GPU every rank drove (ordinal + PCI address), through the collective itself:
        # "the communicator saw N ranks on N different GPUs" is then checkable from the one line
        import ctypes
        pci = ctypes.create_string_buffer(32)
        sb.load().storm_hip_device_pci_bus_id(local_rank, pci, 32)
        ident = [None] * world
        dist.all_gather_object(ident, {"device": local_rank, "pci_bus_id": pci.value.decode(),
                                       "pid": os.getpid()})
        per_rank = [{"rank": r, "kernel_ms": round(float(g[0]), 4), "pass_ms": round(float(g[1]), 4),
                     "wall_ms_per_step": round(float(g[2]), 4),
                     "wall_ms_per_step_no_collective": round(float(g[5]), 4), "allreduce_us": round(float(g[3]), 1),
                     "work_items": int(g[4]), **ident[r]} for r, g in enumerate(gathered)]

    # Secondary figure, never `value`: the same pass when the FP4 re-encoding of the (unchanged)
    # matrix is kept in HBM between calls (library option keep_shadow, what the storm.h handles
    # do) — i.e. without the O(N*M) expansion at the head of every pass.
    shadow_resident = None
    operands_used = ctx.get_option("k2_operands_used")
    if not args.no_shadow_resident and ctx.get_option("variant_used") >= 4 and operands_used == 4:
        ctx.set_option("keep_shadow", 1)
        step(); fence()
        n2 = max(1, args.steps // 2)
        t1 = time.perf_counter()
        for _ in range(n2):
            step()
        fence()
        e2 = time.perf_counter() - t1
        if collective:
            t = torch.tensor([e2], dtype=torch.float64, device=dev if args.backend != "gloo" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e2 = float(t.item())
        ctx.set_option("keep_shadow", 0)
        shadow_resident = {"ms_per_step": e2 * 1e3 / n2, "steps": n2,
                           "total_matches": last_total() == total}

    info = ctx.last_launch_info()
    pairs = N * (N - 1) // 2
    words = pairs * 2 * W
    value = words * args.steps / elapsed

    # correctness of what was timed: size-independent identity sum_c C(n_c, 2), on the device
    identity = mat.column_identity()
    ok = (total == identity)

    if rank == 0:
        used = ctx.get_option("variant_used")
        alg_bytes_launch = pairs * W * 16 / world                # 16 B / word pair
        alg_flop_launch = pairs * W * 64 * 2 / world             # 64 bit-MACs per word pair
        hbm_gbs = alg_bytes_launch / (kernel_ms * 1e-3) / 1e9
        # HBM-side bytes of the dominant kernel per launch: measured with rocprofv3 PMC passes
        # (tools/profile_default.sh -> tools/pmc_traffic.py) and only quoted while the device
        # sources are the ones it was measured on
        traffic, traffic_note = None, "no PMC summary for this kernel variant under profiles/"
        pmc = os.path.join(ROOT, "profiles", "pmc_hbm_bytes_per_launch.json")
        if os.path.exists(pmc):
            try:
                from stormbitmaps_amd._lib import kernel_source_hash
                entry = json.load(open(pmc)).get(f"variant{used}", {})
                if entry.get("source_hash") == kernel_source_hash():
                    traffic = entry.get("hbm_bytes_per_launch")
                    traffic_note = (f"rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE of {entry.get('dominant_kernel')} "
                                    f"at 1 GPU, sources {entry.get('source_hash')}")
                elif entry:
                    traffic_note = ("stale: the PMC summary was measured on other device sources "
                                    f"({entry.get('source_hash')}); re-run tools/profile_default.sh")
            except Exception as e:  # noqa: BLE001
                traffic_note = f"unreadable PMC summary: {e}"
        step_ms = elapsed * 1e3 / args.steps
        if used >= 3:
            # K2/K2s: the bits are multiplied as FP4 on the matrix cores -> MFMA-bound
            achieved = alg_flop_launch / (kernel_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": achieved, "peak": FP4_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / FP4_PEAK_TFLOPS, "traffic": traffic if world == 1 else None,
                    "traffic_note": traffic_note,
                    # the same FLOPs over everything one step contains (fold, launch gaps, all-reduce;
                    # the FP4-shadow form: its expansion pass too): wall time of the timed loop / steps
                    "frac_whole_pass": alg_flop_launch / (step_ms * 1e-3) / 1e12 / FP4_PEAK_TFLOPS,
                    "kernel": ({5: "storm::strip16_bits_kernel", 2: "storm::bitstream_kernel<false>",
                                1: "storm::stripbits_kernel"}.get(operands_used,
                                "storm::strip16_fp4_kernel<4>" if ctx.get_option("k2_shape") == 16 else
                                "storm::strip_fp4_kernel")) if used >= 4 else "storm::pairw_fp4_kernel",
                    "operand_form": ({5: "bit rows read as they are; the FP4 image of every 64-row B stage is built in "
                                         "the LDS by the workgroup (no shadow matrix, no expansion pass)",
                                      2: "bit rows inflated in registers, one stage stream per workgroup",
                                      4: "FP4 shadow matrix rebuilt by expand_fp4_kernel in front of every pass"}
                                     .get(operands_used, str(operands_used))) if used >= 4 else "FP4 shadow",
                    "kernel_ms": kernel_ms, "kernel_ms_samples": dom_n, "pass_ms_untimed": launch_ms,
                    "algorithmic_flop_per_launch": alg_flop_launch,
                    "measured_fp4_mfma_peak_frac": (pairs * W / world / (kernel_ms * 1e-3)) / FP4_MEASURED_WORDPAIRS,
                    "hbm_algorithmic_gb_s": hbm_gbs, "hbm_algorithmic_frac": hbm_gbs / HBM_PEAK_GBS,
                    "note": "1 word pair = 64 bit-MACs = 128 FLOP on v_mfma_f32_16x16x128_f8f6f4 (FP4 operands). The "
                            "reference's no-reuse byte accounting (16 B per word pair, benchmark.cpp:131) is "
                            "kept as hbm_algorithmic_*; on-chip reuse puts it far above the HBM peak."}
        else:
            roof = {"bound": "hbm", "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": hbm_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                    "kernel": f"storm::pairw_dense_kernel<{used}>", "kernel_ms": kernel_ms,
                    "pass_ms_untimed": launch_ms,
                    "algorithmic_bytes_per_launch": alg_bytes_launch,
                    "valu_popcount_frac": (info["word_pairs_executed"] / (kernel_ms * 1e-3)) / VALU_PEAK_WORDPAIRS,
                    "note": "algorithmic bytes use the reference's no-reuse accounting (16 B per word pair, "
                            "benchmark.cpp:131); on-chip reuse lets it exceed the HBM peak — the binding "
                            "resource is VALU popcount issue (valu_popcount_frac)"}
        out = {
            "metric": f"64-bit bitmap words/s, XX^T upper-tri pairwise AND+popcount ({N}x{M})",
            "value": value, "unit": "words/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            # integer bit counts; on the matrix-core path the 0/1 bits are FP4 operands with exact
            # f32 accumulation (< 2^24 per accumulator), totals in uint64 — bit-exact either way
            "dtype": "u64" if used < 3 else "u64 (bits as exact FP4 0/1 MFMA operands, f32 accumulate < 2^24)",
            "data": "synthetic",
            "config": {"workload": f"STORM_contiguous_t N={N} M={M} dense draws={draws} seed={args.seed} "
                                   "(BASELINE configs[1], README `benchmark 65536 10000`)",
                       "entry_point": "storm_hip_pairw_dense_launch == STORM_contig_pairw_intersect_cardinality_blocked",
                       "parallelism": f"work shard x{world} (whole k-slices per rank + leftover slices cut along the "
                                      "pair space), X replicated, one uint64 all-reduce per step",
                       "kernel_variant": used, "work_items": info["items"]},
            "gb_per_s_algorithmic": value * 8 / 1e9,
            "total": total, "verified_against_column_identity": ok,
            "roofline": roof,
        }
        out["pass_ms_untimed_passes"] = [round(x, 4) for x in per_step_pass_ms[:24]]  # rank 0, after the timed steps
        out["host_enqueue_us_first_steps"] = [round(x, 1) for x in host_us[:24]]
        if per_rank is not None:
            out["per_rank"] = per_rank
            out["rccl_ranks"] = dist.get_world_size()     # ranks the process group (backend below) holds
            out["collective_backend"] = dist.get_backend()
            out["distinct_gpus"] = len({r["pci_bus_id"] for r in per_rank})
            out["slowest_rank_kernel_ms"] = max(r["kernel_ms"] for r in per_rank)
        if shadow_resident is not None:
            shadow_resident["value"] = words / (shadow_resident["ms_per_step"] * 1e-3)
            shadow_resident["note"] = ("same pass with the FP4 shadow of the unchanged matrix kept resident "
                                       "(option keep_shadow); informational, `value` always re-expands")
            out["shadow_resident"] = shadow_resident
        if world == 1 and not args.no_cpu_baseline:
            base = cpu_baseline(lambda n: mat.download(0, min(n, N)), W)
            out["cpu_baseline"] = base
            # the checker's word on what was timed: when the CPU sample was the whole matrix its
            # pair-by-pair total must be the GPU's (a smaller sample is checked on its own rows)
            n_cpu = base.pop("sample_rows")
            want = total if n_cpu == N else mat_head_total(ctx, mat, n_cpu, W)
            base["sample_total_matches_gpu"] = (base["sample_total"] == want)
            if not base["sample_total_matches_gpu"]:
                print(f"VERIFICATION FAILED: CPU oracle total over the first {n_cpu} rows "
                      f"{base['sample_total']} != GPU {want}", file=sys.stderr)
                ok = False
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
        if not ok:
            print(f"VERIFICATION FAILED: total {total} != column identity {identity}", file=sys.stderr)
    if collective:
        dist.barrier()
        dist.destroy_process_group()
    mat.close()
    ctx.close()
    if not ok:
        sys.exit(1)


if __name__ == "__main__":
    main()

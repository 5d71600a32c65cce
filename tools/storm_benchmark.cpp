// storm_benchmark.cpp — the reference's benchmark CLI on the MI355X path.
//
// Counterpart of benchmark.cpp (intersect_test :644-1059, benchmark_large :505-642, main
// :1085-1125), written from scratch against include/storm.h + libstorm_hip.so:
//     storm_benchmark <M> <N> [load1,load2,...] [--gpus G | --ranks R] [--seed S] [--reps R] [--cpu-seconds T]
// --gpus G : one process drives G GPUs (STORM_hip_set_devices; partials added on the host).
// --ranks R: R processes, one per GPU — forked HERE, before anything touches HIP — each computes its shard
//            (STORM_hip_set_shard) and the storm.h entry points return the RCCL all-reduced total
//            (STORM_hip_comm_init; the id travels from rank 0 through a pipe). Rank 0 prints the rows.
// Same positional arguments (samples first, benchmark.cpp:1067), same default loads and
// zero/duplicate rules (:695, :715-730), same routing (M < 256000 -> both containers, else
// STORM_t only, :1117-1121; STORM_t rows only when M >= 65536, :832), same optimal block size
// (:823-824), same TSV row shape: name \t load \t [size] \t + the 11 bench_t fields (:74-87).
// What a row means here:
//   * time_ms is ONE call, the first of the row — the reference times exactly one call per row right after
//     construction (benchmark.cpp:605-613, :896-918) — and so includes whatever that call has to bring to the
//     device (the rest of the rows, the sparse arena). The best of the following --reps - 1 calls is the extra
//     column steady_ms; the reference-shaped fields (throughput, cycles) are of the first call, the appended
//     rate columns of the steady one.
//   * cycles / cycles_word: the reference reads the CPU's cycle counter; a GPU row prints wall time x 2.4 GHz
//     (the MI355X's peak engine clock) x GPUs, a CPU row the host's time stamp counter. Instructions, branch and
//     cache misses stay 0: nothing counts them on the device.
//   * kernel / roof / roof_frac: the kernels the call ran (STORM_hip_last_pass) and the fraction of THEIR roof
//     the steady call reached: dense word pairs x 128 FLOP against 10 PFLOP/s FP4 per GPU (matrix-core
//     kernels), word pairs against the VALU popcount ceiling 1.97e13/s (popcount kernel), list-probe lookups
//     against the LDS ceiling 1.97e13 lookups/s (32 two-byte lookups per clock and CU); a call that ran two
//     kinds is priced as the sum of both parts' minimum times. One lookup of the list-probe kernel stands for
//     `rows_per_lookup` (128) of the reference's per-pair list tests (storm.c:4-73): lookups_per_s x 128 is the
//     pair-test rate, the roof fraction is on the lookups the kernel really does.
//   * CPU rows (bitmap-<leaf>-blocked-<b>-cpu): the reference's fwrapper_blocked<leaf> rows (benchmark.cpp:256-316,
//     :949-1045) — this harness's own blocked loop over the raw buffer calling the library's exported one-pair
//     leaf (STORM_intersect_count_scalar / _sse4 / _avx2 / _avx512, stormbitmaps_amd/csrc/storm_leaves.c) on ONE
//     host thread, over the first R rows only (R chosen for --cpu-seconds, default 0.5 s per leaf) and
//     EXTRAPOLATED to N rows by the pair count (note column). GPUs = 0 marks them.
// Other differences, forced by the platform: inputs come from the repo's deterministic generator
// (storm_synth.h) instead of std::random_device (:756-757); time has 3 decimals (a pass takes ~1 ms, the
// reference prints whole ms); no CRoaring rows. `--describe` prints one line per column.
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <sys/wait.h>
#include <unistd.h>
#if defined(__x86_64__)
#include <x86intrin.h>
#endif

#include "storm.h"
#include "storm_hip.h"
#include "storm_synth.h"

static const double kGpuClockHz = 2.4e9;          // MI355X peak engine clock (MI355X_MICROARCH.md)
static const double kFp4FlopPerS = 1e16;          // dense FP4 matrix-core peak per GPU
static const double kValuWordPairsPerS = 1.97e13; // VALU popcount ceiling per GPU
static const double kLdsLookupsPerS = 1.97e13;    // 32 two-byte LDS lookups per clock and CU x 256 CUs x 2.4 GHz

static int g_ranks = 0;  // --ranks: every process reports its own share of the work

struct Row {
    uint64_t total = 0;
    double first_ms = 0, steady_ms = 0;
    double cycles = 0;          // CPU rows: time stamp counter of the timed call
    int gpus = 1;               // 0: a CPU row
    uint64_t pass[4] = {0, 0, 0, 0};
    std::string note;
};

template <class F>
static Row timed(F&& f, int reps, int gpus) {
    Row r;
    r.gpus = gpus;
    r.steady_ms = 1e300;
    for (int k = 0; k < reps; ++k) {
        const auto t0 = std::chrono::high_resolution_clock::now();
        const uint64_t total = f();
        const auto t1 = std::chrono::high_resolution_clock::now();
        const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        if (k == 0) { r.first_ms = ms; r.total = total; }
        else if (ms < r.steady_ms) r.steady_ms = ms;
        if (total != r.total) r.note = "TOTALS DIFFER BETWEEN CALLS";
    }
    if (reps < 2) r.steady_ms = r.first_ms;
    STORM_hip_last_pass(r.pass);
    if (g_ranks > 1) {  // this rank's share of a sharded call: the job's work is (close to) `ranks` times that
        r.pass[1] *= (uint64_t)g_ranks;
        r.pass[2] *= (uint64_t)g_ranks;
    }
    return r;
}

static std::string kernel_names(uint64_t mask) {
    static const struct { uint64_t bit; const char* name; } k[] = {
        {STORM_HIP_RAN_POPCOUNT, "pairw_dense_kernel"}, {STORM_HIP_RAN_FP4_TILES, "pairw_fp4_kernel"},
        {STORM_HIP_RAN_FP4_STRIPS, "strip16_fp4_kernel"}, {STORM_HIP_RAN_BITSTREAM, "bitstream_kernel"},
        {STORM_HIP_RAN_BIT_STRIPS, "strip16_bits_kernel"}, {STORM_HIP_RAN_LIST_PROBE, "probe_lists_kernel"}};
    std::string out;
    for (const auto& e : k)
        if (mask & e.bit) out += (out.empty() ? "" : "+") + std::string(e.name);
    return out.empty() ? "-" : out;
}

static void print_row(const std::string& name, uint32_t load, const char* extra, const Row& r,
                      uint64_t n_variants, uint64_t n_ints) {
    // throughput as benchmark.cpp:128-131: pairs * 2 * W * 8 bytes / 2^20 per second, of the (first) timed call
    const double n_comps = (double)n_variants * (n_variants - 1) / 2.0;
    const double words = n_comps * 2.0 * (double)n_ints;
    const double mbs = words * 8.0 / (1024.0 * 1024.0) / (r.first_ms / 1000.0);
    const double cycles = r.gpus > 0 ? r.first_ms / 1000.0 * kGpuClockHz * r.gpus : r.cycles;
    const double secs = r.steady_ms / 1000.0;
    const double gbs = words * 8.0 / 1e9 / secs;                      // algorithmic (no reuse credit)
    const int g = r.gpus > 0 ? r.gpus : 1;
    // minimum time of what the call really did, part by part, at each kernel's own roof
    double t_min = 0;
    std::string roof = "-";
    if (r.gpus > 0) {
        const bool matrix = r.pass[0] & (STORM_HIP_RAN_FP4_TILES | STORM_HIP_RAN_FP4_STRIPS | STORM_HIP_RAN_BITSTREAM | STORM_HIP_RAN_BIT_STRIPS);
        const bool valu = (r.pass[0] & STORM_HIP_RAN_POPCOUNT) != 0, probe = (r.pass[0] & STORM_HIP_RAN_LIST_PROBE) != 0;
        if (matrix) t_min += (double)r.pass[1] * 128.0 / (kFp4FlopPerS * g);
        else if (valu) t_min += (double)r.pass[1] / (kValuWordPairsPerS * g);
        if (probe) t_min += (double)r.pass[2] / (kLdsLookupsPerS * g);
        roof = std::string(matrix ? "fp4_mfma" : valu ? "valu_popcount" : "") + ((matrix || valu) && probe ? "+" : "") + (probe ? "lds_lookups" : "");
        if (roof.empty()) roof = "-";
    }
    printf("%s\t%u\t%s%llu\t%.2f\t%.4e\t%.2f\t%.0f\t%llu\t%llu\t%llu\t%llu\t%.2f\t%.3f\t%.3f\t%.3f\t%d\t%.4e\t%.1f\t%.3f\t%s\t%s\t%.4f\t%.3e\t%llu\t%s\n",
           name.c_str(), load, extra, (unsigned long long)r.total, 0.0, cycles / words, 0.0, cycles, 0ull, 0ull, 0ull,
           0ull, mbs, r.first_ms, r.first_ms, r.steady_ms, r.gpus, words / secs, gbs, gbs / (8000.0 * g),
           r.gpus > 0 ? kernel_names(r.pass[0]).c_str() : "host", roof.c_str(), r.gpus > 0 ? t_min / secs : 0.0,
           r.gpus > 0 ? (double)r.pass[2] / secs : 0.0, (unsigned long long)r.pass[3], r.note.empty() ? "-" : r.note.c_str());
    fflush(stdout);
}

static void describe_columns() {
    fprintf(stderr,
            "columns of a result row (tab separated; reference row = name, load, [size], then bench_t::PrintPretty, benchmark.cpp:74-87):\n"
            "  1 Method                 row name of the reference (storm, storm-blocked, STORM-contig, STORM-contig-<b>); bitmap-hip-blocked-<b> = STORM_wrapper_diag_blocked on the raw buffer;\n"
            "                           bitmap-<leaf>-blocked-<b>-cpu = the harness's blocked loop over the library's host leaf, one thread, row sample, extrapolated;\n"
            "                           storm-blocked-cpu = the reference's STORM_t host path (benchmark.cpp:605-613, storm.c:897-961) over the library's one-pair helper, one thread, row sample;\n"
            "                           bitmap-scalar-skip-list = flwrapper<STORM_intersect_count_scalar_list> (benchmark.cpp:1039-1045, loads <= 300), one thread, row sample\n"
            "  2 Alts                   values drawn per row (the load)\n"
            "  [3 size]                 STORM_serialized_size, only in the M >= 256000 form (benchmark.cpp:609)\n"
            "  + total                  bench_t.total: sum over row pairs of popcount(A & B) (CPU rows: of the row sample)\n"
            "  + instructions_cycle     0 (no instruction counter on the device)\n"
            "  + cycles_word            cycles / (pairs * 2 * W)\n"
            "  + instructions_word      0\n"
            "  + cycles                 GPU rows: time_ms x 2.4 GHz (peak engine clock) x GPUs; CPU rows: the host's time stamp counter, extrapolated like the time\n"
            "  + instructions, MinBranchMiss, MinCacheRef, MinCacheMiss   0\n"
            "  + throughput             bench_t.throughput of the timed (first) call: pairs * 2 * W * 8 B / 2^20 / s (:129-131), MiB/s\n"
            "  + time_ms                bench_t.time_ms: ONE call, the first of the row, as the reference times it (3 decimals)\n"
            "  + first_call_ms          the same number again, named\n"
            "  + steady_ms              best of the following --reps - 1 calls (device state cached)\n"
            "  + GPUs                   devices the call was sharded over (--gpus / --ranks); 0 = a CPU row\n"
            "  + words_per_s            pairs * 2 * W / steady s: the BASELINE metric\n"
            "  + GB_per_s_algorithmic   words_per_s * 8 / 1e9 (no-reuse accounting of the reference)\n"
            "  + hbm_frac_algorithmic   that over 8 TB/s per GPU (exceeds 1 by design: operands are reused on chip, and the sparse paths skip absent blocks)\n"
            "  + kernel                 the kernels the call ran (STORM_hip_last_pass)\n"
            "  + roof                   what bounds them: fp4_mfma (10 PFLOP/s per GPU, 128 FLOP per word pair), valu_popcount (1.97e13 word pairs/s), lds_lookups (1.97e13/s)\n"
            "  + roof_frac              minimum time of the work the call really did at those roofs / steady time (never above 1)\n"
            "  + lookups_per_s          list-probe lookups per steady second (0 when that kernel did not run)\n"
            "  + rows_per_lookup        rows of the group one lookup stands for (128): lookups x 128 = the reference's per-pair list tests\n"
            "  + note                   CPU rows: the sample and that the time is extrapolated\n");
}

// ---- CPU rows: this harness's blocked upper-triangle loop (the shape of fwrapper_blocked, benchmark.cpp:256-316:
// diagonal blocks, then each block against the blocks behind it) over a one-pair host leaf of the library
typedef uint64_t (*leaf_t)(const uint64_t*, const uint64_t*, size_t);
static uint64_t cpu_blocked(leaf_t leaf, const uint64_t* vals, uint64_t rows, uint32_t n_ints, uint32_t bsize) {
    uint64_t total = 0;
    for (uint64_t i0 = 0; i0 < rows; i0 += bsize) {
        const uint64_t i1 = i0 + bsize < rows ? i0 + bsize : rows;
        for (uint64_t i = i0; i < i1; ++i)
            for (uint64_t j = i + 1; j < i1; ++j) total += leaf(vals + i * n_ints, vals + j * n_ints, n_ints);
        for (uint64_t j0 = i1; j0 < rows; j0 += bsize) {
            const uint64_t j1 = j0 + bsize < rows ? j0 + bsize : rows;
            for (uint64_t i = i0; i < i1; ++i)
                for (uint64_t j = j0; j < j1; ++j) total += leaf(vals + i * n_ints, vals + j * n_ints, n_ints);
        }
    }
    return total;
}
static inline uint64_t tsc() {
#if defined(__x86_64__)
    return __rdtsc();
#else
    return 0;
#endif
}
// vals holds the first `have` rows of an N-row matrix
static void cpu_rows(const uint64_t* vals, uint64_t have, uint64_t N, uint32_t n_ints, uint32_t load, uint32_t bsize,
                     double seconds, const char* extra) {
    struct Leaf { const char* name; leaf_t f; int bit; };
    std::vector<Leaf> leaves = {{"scalar", STORM_intersect_count_scalar, 0}};
#if defined(STORM_HAVE_SSE42)
    leaves.push_back({"sse4", STORM_intersect_count_sse4, STORM_CPUID_runtime_bit_SSE42});
#endif
#if defined(STORM_HAVE_AVX2)
    leaves.push_back({"avx2", STORM_intersect_count_avx2, STORM_CPUID_runtime_bit_AVX2});
#endif
#if defined(STORM_HAVE_AVX512)
    leaves.push_back({"avx512", STORM_intersect_count_avx512, STORM_CPUID_runtime_bit_AVX512BW});
#endif
    const int cpuid = STORM_get_cpuid();
    const uint64_t R0 = have < 64 ? have : 64;   // calibration and cross-check sample
    const uint64_t want0 = cpu_blocked(STORM_intersect_count_scalar, vals, R0, n_ints, bsize);
    for (const Leaf& l : leaves) {
        if (l.bit && !(cpuid & l.bit)) continue;
        // calibrate on 64 rows (and check the leaf against the scalar one there), then the sample whose pair count fits the budget
        uint64_t R = R0;
        auto t0 = std::chrono::high_resolution_clock::now();
        const bool agrees = cpu_blocked(l.f, vals, R, n_ints, bsize) == want0;
        double s = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
        const double per_pair = s / ((double)R * (R - 1) / 2.0 + 1.0);
        uint64_t fit = (uint64_t)std::sqrt(2.0 * seconds / (per_pair > 0 ? per_pair : 1e-9));
        R = fit < 64 ? 64 : fit;
        if (R > have) R = have;
        t0 = std::chrono::high_resolution_clock::now();
        const uint64_t c0 = tsc();
        const uint64_t total = cpu_blocked(l.f, vals, R, n_ints, bsize);
        const uint64_t c1 = tsc();
        s = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
        const double scale = ((double)N * (N - 1) / 2.0) / ((double)R * (R - 1) / 2.0);
        Row r;
        r.gpus = 0;
        r.total = total;
        r.first_ms = r.steady_ms = s * 1e3 * scale;
        r.cycles = (double)(c1 - c0) * scale;
        char note[320];
        snprintf(note, sizeof(note), "cpu 1 thread; first %llu of %llu rows timed (%.3f s), time and cycles extrapolated x%.1f by pair count; total is the sample's; leaf %s the scalar leaf on the first %llu rows",
                 (unsigned long long)R, (unsigned long long)N, s, scale, agrees ? "==" : "!=", (unsigned long long)R0);
        r.note = note;
        print_row(std::string("bitmap-") + l.name + "-blocked-" + std::to_string(bsize) + "-cpu", load, extra, r, N, n_ints);
    }
}

// ---- "storm-blocked-cpu": the reference's STORM_t CPU path itself (benchmark_large times STORM_pairw_intersect_cardinality_blocked
// on the host, benchmark.cpp:605-613; the loop is storm.c:897-961) over the product's OWN exported one-pair helper
// STORM_bitmap_cont_intersect_cardinality_premade (storm.c:790-814: merge of the two rows' block ids, then the 4-way kind
// dispatch per matching block): one thread, the first R rows of the container, extrapolated by pair count.
static uint64_t cpu_storm_blocked(const STORM_t* h, uint64_t rows, uint32_t bsize, uint32_t* scratch) {
    const STORM_compute_func leaf = STORM_get_intersect_count_func(1024);
    uint64_t total = 0;
    for (uint64_t i0 = 0; i0 < rows; i0 += bsize) {
        const uint64_t i1 = i0 + bsize < rows ? i0 + bsize : rows;
        for (uint64_t i = i0; i < i1; ++i)
            for (uint64_t j = i + 1; j < i1; ++j)
                total += STORM_bitmap_cont_intersect_cardinality_premade(&h->conts[i], &h->conts[j], leaf, scratch);
        for (uint64_t j0 = i1; j0 < rows; j0 += bsize) {
            const uint64_t j1 = j0 + bsize < rows ? j0 + bsize : rows;
            for (uint64_t i = i0; i < i1; ++i)
                for (uint64_t j = j0; j < j1; ++j)
                    total += STORM_bitmap_cont_intersect_cardinality_premade(&h->conts[i], &h->conts[j], leaf, scratch);
        }
    }
    return total;
}
static void cpu_storm_row(const STORM_t* h, uint64_t N, uint32_t n_ints, uint32_t load, double seconds, const char* extra,
                          const uint64_t* dense_rows, uint64_t dense_have) {
    // block size as STORM_pairw_intersect_cardinality_blocked(h, 0) derives it (storm.c:903-914): 256e3 / average serialized row
    const uint64_t bytes = STORM_serialized_size(h);
    uint32_t bsize = (uint32_t)std::ceil(256e3 / ((double)bytes / (double)(N ? N : 1) + 1.0));
    if (bsize < 5) bsize = 5;
    std::vector<uint32_t> scratch(2 * 4096);   // storm.c:900
    uint64_t R = N < 64 ? N : 64;
    if (dense_rows && dense_have < R) R = dense_have;
    auto t0 = std::chrono::high_resolution_clock::now();
    const uint64_t t64 = cpu_storm_blocked(h, R, bsize, scratch.data());
    double s = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    // (the same rows as a dense bit matrix under the scalar leaf: the container path must count the same)
    const bool agrees = !dense_rows || t64 == cpu_blocked(STORM_intersect_count_scalar, dense_rows, R, n_ints, 31);
    const uint64_t R0 = R;
    const double per_pair = s / ((double)R * (R - 1) / 2.0 + 1.0);
    const uint64_t fit = (uint64_t)std::sqrt(2.0 * seconds / (per_pair > 0 ? per_pair : 1e-9));
    R = fit < 64 ? 64 : fit;
    if (R > N) R = N;
    t0 = std::chrono::high_resolution_clock::now();
    const uint64_t c0 = tsc();
    const uint64_t total = cpu_storm_blocked(h, R, bsize, scratch.data());
    const uint64_t c1 = tsc();
    s = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    const double scale = ((double)N * (N - 1) / 2.0) / ((double)R * (R - 1) / 2.0 + 1e-9);
    Row r;
    r.gpus = 0;
    r.total = total;
    r.first_ms = r.steady_ms = s * 1e3 * scale;
    r.cycles = (double)(c1 - c0) * scale;
    char note[360];
    snprintf(note, sizeof(note), "cpu 1 thread; the STORM_t host path (storm.c:897-961) over the library's one-pair helper STORM_bitmap_cont_intersect_cardinality_premade, bsize %u; first %llu of %llu rows timed (%.3f s), time and cycles extrapolated x%.1f by pair count; total is the sample's; %s the dense scalar leaf on the first %llu rows",
             bsize, (unsigned long long)R, (unsigned long long)N, s, scale, agrees ? "==" : "!=", (unsigned long long)R0);
    r.note = note;
    print_row("storm-blocked-cpu", load, extra, r, N, n_ints);
}

// ---- "bitmap-scalar-skip-list" (benchmark.cpp:1039-1045, loads <= 300): flwrapper's plain all-pairs loop (:318-335) over
// STORM_intersect_count_scalar_list — the shorter row's positions probed in the other row's bitmap — one thread, first R rows.
static void cpu_skip_list_row(const uint64_t* vals, uint64_t N, uint32_t n_ints, uint32_t load, double seconds) {
    const uint64_t have = N < 4096 ? N : 4096;
    std::vector<std::vector<uint32_t>> pos(have);
    for (uint64_t i = 0; i < have; ++i)
        for (uint32_t w = 0; w < n_ints; ++w)
            for (uint64_t x = vals[i * n_ints + w]; x; x &= x - 1) pos[i].push_back(w * 64u + (uint32_t)__builtin_ctzll(x));
    auto run = [&](uint64_t rows) {
        uint64_t total = 0;
        for (uint64_t i = 0; i < rows; ++i)
            for (uint64_t j = i + 1; j < rows; ++j)
                total += STORM_intersect_count_scalar_list(vals + i * n_ints, vals + j * n_ints, pos[i].data(), pos[j].data(),
                                                           pos[i].size(), pos[j].size());
        return total;
    };
    uint64_t R = have < 64 ? have : 64;
    auto t0 = std::chrono::high_resolution_clock::now();
    const bool agrees = run(R) == cpu_blocked(STORM_intersect_count_scalar, vals, R, n_ints, 31);
    double s = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    const double per_pair = s / ((double)R * (R - 1) / 2.0 + 1.0);
    const uint64_t fit = (uint64_t)std::sqrt(2.0 * seconds / (per_pair > 0 ? per_pair : 1e-9));
    R = fit < 64 ? 64 : fit;
    if (R > have) R = have;
    t0 = std::chrono::high_resolution_clock::now();
    const uint64_t c0 = tsc();
    const uint64_t total = run(R);
    const uint64_t c1 = tsc();
    s = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    const double scale = ((double)N * (N - 1) / 2.0) / ((double)R * (R - 1) / 2.0 + 1e-9);
    Row r;
    r.gpus = 0;
    r.total = total;
    r.first_ms = r.steady_ms = s * 1e3 * scale;
    r.cycles = (double)(c1 - c0) * scale;
    char note[320];
    snprintf(note, sizeof(note), "cpu 1 thread; flwrapper's all-pairs loop over STORM_intersect_count_scalar_list; first %llu of %llu rows timed (%.3f s), time and cycles extrapolated x%.1f by pair count; total is the sample's; %s the dense scalar leaf on 64 rows",
             (unsigned long long)R, (unsigned long long)N, s, scale, agrees ? "==" : "!=");
    r.note = note;
    print_row("bitmap-scalar-skip-list", load, "", r, N, n_ints);
}

static std::vector<uint32_t> default_loads(uint32_t M) {
    return {M / 2, M / 4, M / 10, M / 25, M / 50, M / 100, M / 250, M / 1000, M / 5000, 5, 1};
}

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr,
                "\nAbout:   Computes sum(popcnt(A & B)) for the all-vs-all comparison of N integer\n"
                "         lists bounded by [0, M) on the MI355X.\n"
                "Usage:   storm_benchmark <M> <N> [v1[,v2]] [--gpus G | --ranks R] [--seed S] [--reps R] [--cpu-seconds T (0: no CPU rows)] [--describe]\n\n");
        return EXIT_FAILURE;
    }
    int64_t n_samples = 0, n_vals = 10000;  // one-argument form uses N = 10000 (benchmark.cpp:1102)
    std::vector<uint32_t> loads;
    // 8 calls per row: the first is the reference's one timed call; a ~1 ms kernel right after seconds of host-side
    // construction runs below the clocks back-to-back calls reach (0.93 against 0.77 ms at c2 with only two followers)
    int gpus = 1, reps = 8, positional = 0, ranks = 0;
    double cpu_seconds = 0.5;
    uint64_t seed = 42;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--gpus") && i + 1 < argc) gpus = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--ranks") && i + 1 < argc) ranks = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--seed") && i + 1 < argc) seed = strtoull(argv[++i], nullptr, 10);
        else if (!strcmp(argv[i], "--reps") && i + 1 < argc) reps = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--cpu-seconds") && i + 1 < argc) cpu_seconds = atof(argv[++i]);
        else if (!strcmp(argv[i], "--describe")) { describe_columns(); return EXIT_SUCCESS; }
        else if (positional == 0) { n_samples = atoll(argv[i]); ++positional; }
        else if (positional == 1) { n_vals = atoll(argv[i]); ++positional; }
        else {
            for (char* tok = strtok(argv[i], ","); tok; tok = strtok(nullptr, ",")) loads.push_back((uint32_t)atoi(tok));
            ++positional;
        }
    }
    if (n_samples <= 0) { fprintf(stderr, "Cannot have non-positive number of samples...\n"); return EXIT_FAILURE; }
    if (n_vals <= 0) { fprintf(stderr, "Cannot have non-positive number of vectors...\n"); return EXIT_FAILURE; }
    const uint32_t M = (uint32_t)n_samples;
    const uint64_t N = (uint64_t)n_vals;
    const bool large = n_samples >= 256000;  // benchmark.cpp:1117-1121
    if (loads.empty()) loads = default_loads(M);

    int rank = 0;
    if (ranks > 0) {
        // One process per GPU. The fork comes BEFORE any HIP call (a process that has initialised the GPU must
        // not be forked), so the device count is not looked at here: every child checks its own device.
        if (ranks > 16) { fprintf(stderr, "--ranks %d: at most 16\n", ranks); return EXIT_FAILURE; }
        // Under a profiler the preloaded tool library has brought HIP up before main(): the children would inherit
        // a forked HIP/HSA runtime (undefined behaviour). bench.py refuses the same case.
        const char* preload = getenv("LD_PRELOAD");
        if (getenv("ROCP_TOOL_LIBRARIES") || getenv("ROCPROFILER_REGISTER_FORCE_LOAD") ||
            (preload && (strstr(preload, "rocprof") || strstr(preload, "roctracer")))) {
            fprintf(stderr, "--ranks under a profiler: the GPU runtime is already initialised in this process and must "
                            "not be forked; profile with --gpus N, or one rank per profiler invocation\n");
            return EXIT_FAILURE;
        }
        int id_pipe[2];
        if (pipe(id_pipe) != 0) { perror("pipe"); return EXIT_FAILURE; }
        std::vector<pid_t> kids;
        bool child = false;
        for (int r = 0; r < ranks; ++r) {
            const pid_t pid = fork();
            if (pid < 0) { perror("fork"); return EXIT_FAILURE; }
            if (pid == 0) { rank = r; child = true; break; }
            kids.push_back(pid);
        }
        if (!child) {  // the launcher: no HIP here; wait for the ranks, report the worst exit code
            close(id_pipe[0]);
            close(id_pipe[1]);
            int worst = 0;
            for (pid_t k : kids) {
                int st = 0;
                waitpid(k, &st, 0);
                const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128;
                if (code > worst) worst = code;
            }
            return worst;
        }
        const int dev = rank;
        if (storm_hip_device_count() < ranks) {  // every rank sees the same count: all leave before the collective
            if (rank == 0) fprintf(stderr, "--ranks %d but only %d device(s) visible\n", ranks, storm_hip_device_count());
            return EXIT_FAILURE;
        }
        if (STORM_hip_set_devices(1, &dev) != 0 || STORM_hip_set_shard((uint32_t)rank, (uint32_t)ranks) != 0) return EXIT_FAILURE;
        uint8_t id[128];
        if (rank == 0) {
            if (STORM_hip_comm_unique_id(id) != 0) return EXIT_FAILURE;
            for (int r = 1; r < ranks; ++r)
                if (write(id_pipe[1], id, sizeof(id)) != (ssize_t)sizeof(id)) return EXIT_FAILURE;  // 128 B: atomic
        } else if (read(id_pipe[0], id, sizeof(id)) != (ssize_t)sizeof(id)) {
            return EXIT_FAILURE;
        }
        close(id_pipe[0]);
        close(id_pipe[1]);
        if (STORM_hip_comm_init(id) != 0) { fprintf(stderr, "rank %d: %s\n", rank, STORM_hip_error()); return EXIT_FAILURE; }
        gpus = ranks;  // the rows report the GPUs the job ran on
        g_ranks = ranks;
        if (rank != 0 && !freopen("/dev/null", "w", stdout)) return EXIT_FAILURE;  // rank 0 prints
    } else {
        const int visible = storm_hip_device_count();
        if (visible < 1) { fprintf(stderr, "no HIP device visible (no CPU fallback)\n"); return EXIT_FAILURE; }
        if (gpus > visible) { fprintf(stderr, "--gpus %d but only %d device(s) visible; using %d\n", gpus, visible, visible); gpus = visible; }
        std::vector<int> ids(gpus);
        for (int g = 0; g < gpus; ++g) ids[g] = g;
        STORM_hip_set_devices(gpus, ids.data());
    }

    // the reference's header line as it stands (benchmark.cpp:506, :671; it does not match its own rows),
    // then the names of the columns actually printed
    printf("Samples\tAlts\tMethod\tTime(ms)\tCPUCycles\tCount\tThroughput(MB/s)\tInts/s(1e6)\tIntersect/s(1e6)\tActualThroughput(MB/s)\tCycles/int\tCycles/intersect\n");
    printf("#Method\tAlts\t%stotal\tinstructions_cycle\tcycles_word\tinstructions_word\tcycles\tinstructions\tMinBranchMiss\tMinCacheRef\tMinCacheMiss\tthroughput(MiB/s)\ttime_ms\tfirst_call_ms\tsteady_ms\tGPUs\twords_per_s\tGB_per_s_algorithmic\thbm_frac_algorithmic\tkernel\troof\troof_frac\tlookups_per_s\trows_per_lookup\tnote\n",
           n_samples >= 256000 ? "size\t" : "");
    const uint32_t n_ints = (uint32_t)std::ceil(M / 64.0);
    uint32_t optimal_b = (uint32_t)(STORM_CACHE_BLOCK_SIZE / (n_ints * 8));  // :823-824
    if (optimal_b < 5) optimal_b = 5;

    STORM_t* twk2 = STORM_new();
    STORM_contiguous_t* twk_cont = large ? nullptr : STORM_contig_new(M);
    std::vector<uint64_t> vals;
    if (!large) vals.resize((size_t)n_ints * N);

    for (size_t a = 0; a < loads.size(); ++a) {
        if (loads[a] == 0) {  // :715-724: always finish with n_alts = 1
            if (a != 0 && loads[a - 1] != 1) loads[a] = 1; else if (a == 0) break;
        }
        if (a != 0 && loads[a] == loads[a - 1]) break;  // :727-730
        STORM_clear(twk2);
        storm_synth_fill_storm(twk2, M, 0, N, loads[a], seed);
        const uint64_t storm_size = STORM_serialized_size(twk2);
        if (large) {
            char extra[64];
            snprintf(extra, sizeof(extra), "%llu\t", (unsigned long long)storm_size);
            print_row("storm-blocked", loads[a], extra,
                      timed([&] { return STORM_pairw_intersect_cardinality_blocked(twk2, 0); }, reps, gpus), N, n_ints);  // :605-613
            if (cpu_seconds > 0 && rank == 0) {  // the dense leaf on the host beside it: a row sample of the same shape
                const uint64_t rows = N < 512 ? N : 512;
                std::vector<uint64_t> sample((size_t)n_ints * rows);
                storm_synth_fill_dense(sample.data(), n_ints, M, 0, rows, loads[a], seed);
                cpu_storm_row(twk2, N, n_ints, loads[a], cpu_seconds, extra, sample.data(), rows);   // what :605-613 times: the host path
                cpu_rows(sample.data(), rows, N, n_ints, loads[a], optimal_b, cpu_seconds, extra);
            }
            continue;
        }
        STORM_contig_clear(twk_cont);
        storm_synth_fill_contig(twk_cont, M, 0, N, loads[a], seed);
        storm_synth_fill_dense(vals.data(), n_ints, M, 0, N, loads[a], seed);
        if (n_samples >= 65536) {  // :832-852
            print_row("storm", loads[a], "", timed([&] { return STORM_pairw_intersect_cardinality(twk2); }, reps, gpus), N, n_ints);
            print_row("storm-blocked", loads[a], "", timed([&] { return STORM_pairw_intersect_cardinality_blocked(twk2, 0); }, reps, gpus), N, n_ints);
        }
        print_row("STORM-contig", loads[a], "", timed([&] { return STORM_contig_pairw_intersect_cardinality(twk_cont); }, reps, gpus), N, n_ints);  // :896-904
        print_row("STORM-contig-" + std::to_string(optimal_b), loads[a], "",
                  timed([&] { return STORM_contig_pairw_intersect_cardinality_blocked(twk_cont, optimal_b); }, reps, gpus), N, n_ints);  // :906-918
        // the reference's fwrapper_blocked<leaf> rows (:961,:1013,:1031) on the raw buffer: here the
        // raw-buffer wrapper, which copies `vals` to the device on every call ...
        print_row("bitmap-hip-blocked-" + std::to_string(optimal_b), loads[a], "",
                  timed([&] { return STORM_wrapper_diag_blocked((uint32_t)N, vals.data(), n_ints, nullptr, optimal_b); }, reps, gpus), N, n_ints);
        // ... and the same loop on the host over the library's SIMD leaves (one thread, row sample, extrapolated)
        if (cpu_seconds > 0 && rank == 0) cpu_rows(vals.data(), N, N, n_ints, loads[a], optimal_b, cpu_seconds, "");
        if (cpu_seconds > 0 && rank == 0 && n_samples >= 65536)   // the STORM_t host path beside the "storm" rows
            cpu_storm_row(twk2, N, n_ints, loads[a], cpu_seconds, "", vals.data(), N);
        if (cpu_seconds > 0 && rank == 0 && loads[a] <= 300) cpu_skip_list_row(vals.data(), N, n_ints, loads[a], cpu_seconds);   // :1039-1045
    }
    STORM_free(twk2);
    if (twk_cont) STORM_contig_free(twk_cont);
    if (ranks > 0) STORM_hip_shutdown();  // communicator first, then the contexts
    return EXIT_SUCCESS;
}

// mt19937_inputs.cpp — TEST INFRASTRUCTURE (oracle side).
// Regenerates the inputs the survey used when it recorded totals from the unmodified
// reference (SURVEY.md §8 a-note / §8c: "libstdc++ mt19937(42) +
// uniform_int_distribution<uint32_t>(0, M-1), draws with replacement per row"), following
// the reference harness's draw loop (benchmark.cpp:750-772): one engine for the whole matrix,
// `draws` values per row, a value is kept the first time its bit is seen, the kept values are
// sorted. Output is CSR: offsets[N+1], positions[offsets[N]].
#include <algorithm>
#include <cstdint>
#include <random>
#include <vector>

extern "C" uint64_t mtgen_positions(uint32_t M, uint32_t N, uint32_t draws, uint32_t seed,
                                    uint64_t* offsets, uint32_t* positions,
                                    uint64_t capacity) {
    std::mt19937 eng(seed);
    std::uniform_int_distribution<uint32_t> distr(0, M - 1);
    std::vector<uint64_t> seen((M + 63) / 64);
    std::vector<uint32_t> row;
    uint64_t used = 0;
    offsets[0] = 0;
    for (uint32_t j = 0; j < N; ++j) {
        std::fill(seen.begin(), seen.end(), 0);
        row.clear();
        for (uint32_t i = 0; i < draws; ++i) {
            const uint32_t v = distr(eng);
            if (((seen[v >> 6] >> (v & 63)) & 1) == 0) row.push_back(v);
            seen[v >> 6] |= 1ULL << (v & 63);
        }
        std::sort(row.begin(), row.end());
        if (used + row.size() > capacity) return (uint64_t)-1;
        std::copy(row.begin(), row.end(), positions + used);
        used += row.size();
        offsets[j + 1] = used;
    }
    return used;
}

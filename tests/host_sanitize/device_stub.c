/* device_stub.c — TEST ONLY. Stands in for libstorm_hip's device entry points so that the host
 * side (stormbitmaps_amd/csrc/storm_host.c: the storm.h containers, their growth paths and the
 * marshalling towards the device) can run under AddressSanitizer / UBSan on a machine without a
 * GPU. It computes no pair counts: the "totals" it returns are the number of set bits that
 * reached it, which the driver compares with what it fed in. Never linked into the product. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "storm_hip.h"

struct storm_hip_ctx_s { int device; uint64_t pending; };
struct storm_hip_matrix_s { uint64_t n_rows; uint32_t n_words; uint64_t* rows; };
struct storm_hip_sparse_s { uint64_t set_bits; };

static const char* g_err = "";
const char* storm_hip_last_error(void) { return g_err; }
int storm_hip_device_count(void) { return 1; }

int storm_hip_ctx_create(int device, void* stream, storm_hip_ctx_t** out) {
    (void)stream;
    *out = (storm_hip_ctx_t*)calloc(1, sizeof(**out));
    if (!*out) return STORM_HIP_ENOMEM;
    (*out)->device = device;
    return STORM_HIP_OK;
}
void storm_hip_ctx_destroy(storm_hip_ctx_t* ctx) { free(ctx); }
int storm_hip_ctx_set_option(storm_hip_ctx_t* ctx, const char* key, int64_t value) {
    (void)ctx; (void)key; (void)value;
    return STORM_HIP_OK;
}
int storm_hip_ctx_reserve_staging(storm_hip_ctx_t* ctx) { (void)ctx; return STORM_HIP_OK; }
int storm_hip_option_check(const char* key, int64_t value) {
    (void)value;
    return key && key[0] && key[0] != '?' ? STORM_HIP_OK : STORM_HIP_EINVAL;   /* ("?..." stands for a typo) */
}

int storm_hip_matrix_create(storm_hip_ctx_t* ctx, uint64_t n_rows, uint32_t n_words,
                            storm_hip_matrix_t** out) {
    (void)ctx;
    storm_hip_matrix_t* m = (storm_hip_matrix_t*)calloc(1, sizeof(*m));
    if (!m) return STORM_HIP_ENOMEM;
    m->n_rows = n_rows;
    m->n_words = n_words;
    m->rows = (uint64_t*)calloc((size_t)n_rows * n_words + 1, sizeof(uint64_t));
    *out = m;
    return m->rows ? STORM_HIP_OK : STORM_HIP_ENOMEM;
}
void storm_hip_matrix_destroy(storm_hip_ctx_t* ctx, storm_hip_matrix_t* m) {
    (void)ctx;
    if (m) free(m->rows);
    free(m);
}
int storm_hip_matrix_resize(storm_hip_ctx_t* ctx, storm_hip_matrix_t* m, uint64_t n_rows) {
    (void)ctx;
    uint64_t* nr = (uint64_t*)calloc((size_t)n_rows * m->n_words + 1, sizeof(uint64_t));
    if (!nr) return STORM_HIP_ENOMEM;
    const uint64_t keep = n_rows < m->n_rows ? n_rows : m->n_rows;
    memcpy(nr, m->rows, (size_t)keep * m->n_words * sizeof(uint64_t));
    free(m->rows);
    m->rows = nr;
    m->n_rows = n_rows;
    return STORM_HIP_OK;
}
int storm_hip_matrix_upload(storm_hip_ctx_t* ctx, storm_hip_matrix_t* m, uint64_t row0,
                            uint64_t n_rows, const uint64_t* host, uint64_t stride_words) {
    (void)ctx;
    if (row0 + n_rows > m->n_rows) return STORM_HIP_EINVAL;
    for (uint64_t r = 0; r < n_rows; ++r) /* reads exactly what the real upload reads */
        memcpy(m->rows + (row0 + r) * m->n_words, host + r * stride_words,
               (size_t)m->n_words * sizeof(uint64_t));
    return STORM_HIP_OK;
}

int storm_hip_matrix_set_rows_from_positions(storm_hip_ctx_t* ctx, storm_hip_matrix_t* m, uint64_t row0,
                                             uint64_t n_rows, const uint64_t* offsets,
                                             const uint32_t* positions) {
    (void)ctx;
    if (row0 + n_rows > m->n_rows) return STORM_HIP_EINVAL;
    for (uint64_t r = 0; r < n_rows; ++r) /* reads exactly what the real call reads */
        for (uint64_t k = offsets[r]; k < offsets[r + 1]; ++k) {
            if (positions[k] >= (uint64_t)m->n_words * 64) return STORM_HIP_EINVAL;
            m->rows[(row0 + r) * m->n_words + positions[k] / 64] |= 1ULL << (positions[k] % 64);
        }
    return STORM_HIP_OK;
}

static uint64_t set_bits_of(const storm_hip_matrix_t* m) {
    uint64_t n = 0;
    for (uint64_t k = 0; k < m->n_rows * m->n_words; ++k) n += (uint64_t)__builtin_popcountll(m->rows[k]);
    return n;
}
int storm_hip_pairw_dense_begin(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* m,
                                uint32_t shard_rank, uint32_t shard_count) {
    if (shard_rank >= shard_count) return STORM_HIP_EINVAL;
    ctx->pending = shard_rank == 0 ? set_bits_of(m) : 0;
    return STORM_HIP_OK;
}
int storm_hip_pairw_dense_end(storm_hip_ctx_t* ctx, uint64_t* h_total) {
    *h_total = ctx->pending;
    return STORM_HIP_OK;
}
int storm_hip_pairw_dense_upload(storm_hip_ctx_t* ctx, storm_hip_matrix_t* m, const uint64_t* host,
                                 uint64_t stride_words, uint64_t* h_total) {
    if (storm_hip_matrix_upload(ctx, m, 0, m->n_rows, host, stride_words) != STORM_HIP_OK) return STORM_HIP_EINVAL;
    *h_total = set_bits_of(m);
    return STORM_HIP_OK;
}
int storm_hip_square_dense(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* a,
                           const storm_hip_matrix_t* b, uint64_t* h_total) {
    (void)ctx;
    *h_total = set_bits_of(a) + set_bits_of(b);
    return STORM_HIP_OK;
}
int storm_hip_pairw_matrix_device(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* m, int op, uint32_t* d_out,
                                  uint64_t ld) {
    (void)ctx; (void)op;
    if (ld < m->n_rows) return STORM_HIP_EINVAL;
    for (uint64_t i = 0; i < m->n_rows; ++i) /* ("device" memory is host memory here: touches the whole window) */
        for (uint64_t j = 0; j < m->n_rows; ++j) d_out[i * ld + j] = 0;
    return STORM_HIP_OK;
}
int storm_hip_pairw_matrix_band_begin(storm_hip_ctx_t* ctx, const storm_hip_matrix_t* m, int op,
                                      uint64_t row0, uint64_t n_band_rows, uint32_t* h_out, uint64_t ld) {
    (void)ctx; (void)op;
    if (ld < m->n_rows || row0 + n_band_rows > m->n_rows) return STORM_HIP_EINVAL;
    for (uint64_t i = 0; i < n_band_rows; ++i) /* touches the whole band of the output */
        for (uint64_t j = 0; j < m->n_rows; ++j) h_out[i * ld + j] = 0;
    return STORM_HIP_OK;
}
int storm_hip_pairw_matrix_band_end(storm_hip_ctx_t* ctx) { (void)ctx; return STORM_HIP_OK; }

int storm_hip_sparse_create(storm_hip_ctx_t* ctx, uint64_t n_rows, uint64_t n_blocks,
                            const uint64_t* row_block_offset, const uint32_t* block_id,
                            const uint8_t* block_kind, const uint64_t* block_data_offset,
                            const uint32_t* block_n, const uint16_t* list_pool,
                            uint64_t list_pool_len, const uint64_t* bitmap_pool,
                            uint64_t bitmap_pool_words, storm_hip_sparse_t** out) {
    (void)ctx;
    if (row_block_offset[0] != 0 || row_block_offset[n_rows] != n_blocks) return STORM_HIP_EINVAL;
    uint64_t bits = 0;
    for (uint64_t r = 0; r < n_rows; ++r)
        for (uint64_t b = row_block_offset[r]; b < row_block_offset[r + 1]; ++b) {
            if (b > row_block_offset[r] && block_id[b] <= block_id[b - 1]) return STORM_HIP_EINVAL;
            if (block_kind[b] == 0) {
                if (block_data_offset[b] + block_n[b] > list_pool_len) return STORM_HIP_EINVAL;
                for (uint32_t k = 0; k < block_n[b]; ++k) {
                    const uint16_t v = list_pool[block_data_offset[b] + k];
                    if (k && v <= list_pool[block_data_offset[b] + k - 1]) return STORM_HIP_EINVAL;
                }
                bits += block_n[b];
            } else {
                if (block_data_offset[b] + 1024 > bitmap_pool_words) return STORM_HIP_EINVAL;
                for (int k = 0; k < 1024; ++k)
                    bits += (uint64_t)__builtin_popcountll(bitmap_pool[block_data_offset[b] + k]);
            }
        }
    *out = (storm_hip_sparse_t*)calloc(1, sizeof(**out));
    if (!*out) return STORM_HIP_ENOMEM;
    (*out)->set_bits = bits;
    return STORM_HIP_OK;
}
/* the block stage: a host copy of every staged block (ASan checks the 8 KiB read), tokens in order */
struct storm_hip_stage_s { uint64_t n; uint64_t sum; uint64_t list_bytes; };
int storm_hip_stage_create(storm_hip_ctx_t* ctx, storm_hip_stage_t** out) {
    (void)ctx;
    *out = (storm_hip_stage_t*)calloc(1, sizeof(**out));
    return *out ? STORM_HIP_OK : STORM_HIP_ENOMEM;
}
int storm_hip_stage_add(storm_hip_ctx_t* ctx, storm_hip_stage_t* st, const uint64_t* words, uint64_t* token) {
    (void)ctx;
    for (int k = 0; k < 1024; ++k) st->sum += (uint64_t)__builtin_popcountll(words[k]);
    *token = st->n++;
    return STORM_HIP_OK;
}
int storm_hip_stage_add_list(storm_hip_ctx_t* ctx, storm_hip_stage_t* st, const uint16_t* list, uint32_t n, uint64_t* token) {
    (void)ctx;
    if (n == 0 || n > 65536u) return STORM_HIP_EINVAL;
    for (uint32_t k = 0; k < n; ++k) st->sum += list[k];   /* (ASan checks the list's extent) */
    *token = st->list_bytes;
    st->list_bytes += 2ull * n;
    return STORM_HIP_OK;
}
uint64_t storm_hip_stage_count(const storm_hip_stage_t* st) { return st ? st->n : 0; }
void storm_hip_stage_destroy(storm_hip_ctx_t* ctx, storm_hip_stage_t* st) { (void)ctx; free(st); }
int storm_hip_sparse_create_blocks(storm_hip_ctx_t* ctx, uint64_t n_rows, uint64_t n_blocks,
                                   const uint64_t* row_block_offset, const uint32_t* block_id,
                                   const uint8_t* block_kind, const uint32_t* block_n,
                                   const void* const* block_ptr, storm_hip_sparse_t** out);
int storm_hip_sparse_create_blocks_staged(storm_hip_ctx_t* ctx, uint64_t n_rows, uint64_t n_blocks,
                                          const uint64_t* row_block_offset, const uint32_t* block_id,
                                          const uint8_t* block_kind, const uint32_t* block_n, const void* const* block_ptr,
                                          storm_hip_stage_t* stage, const uint64_t* token, storm_hip_sparse_t** out) {
    /* every bitmap block must carry the token of a staged block, in staging order; a list block a place inside the staged
     * lists, ascending, or ~0 */
    uint64_t next = 0, lnext = 0;
    for (uint64_t b = 0; b < n_blocks; ++b) {
        if (block_kind[b] && token[b] != next++) return STORM_HIP_EINVAL;
        if (!block_kind[b] && block_n[b] && token[b] != ~0ull) {
            if (token[b] != lnext) return STORM_HIP_EINVAL;
            lnext += 2ull * block_n[b];
        }
    }
    if (next != stage->n || lnext > stage->list_bytes) return STORM_HIP_EINVAL;
    return storm_hip_sparse_create_blocks(ctx, n_rows, n_blocks, row_block_offset, block_id, block_kind, block_n, block_ptr, out);
}
/* reads every list entry and bitmap word through the pointers handed over (ASan checks their extents) */
int storm_hip_sparse_create_blocks(storm_hip_ctx_t* ctx, uint64_t n_rows, uint64_t n_blocks,
                                   const uint64_t* row_block_offset, const uint32_t* block_id,
                                   const uint8_t* block_kind, const uint32_t* block_n,
                                   const void* const* block_ptr, storm_hip_sparse_t** out) {
    (void)ctx;
    if (row_block_offset[0] != 0 || row_block_offset[n_rows] != n_blocks) return STORM_HIP_EINVAL;
    uint64_t bits = 0;
    for (uint64_t r = 0; r < n_rows; ++r)
        for (uint64_t b = row_block_offset[r]; b < row_block_offset[r + 1]; ++b) {
            if (b > row_block_offset[r] && block_id[b] <= block_id[b - 1]) return STORM_HIP_EINVAL;
            if (block_kind[b] == 0) {
                const uint16_t* l = (const uint16_t*)block_ptr[b];
                for (uint32_t k = 0; k < block_n[b]; ++k)
                    if (k && l[k] <= l[k - 1]) return STORM_HIP_EINVAL;
                bits += block_n[b];
            } else {
                const uint64_t* w = (const uint64_t*)block_ptr[b];
                for (int k = 0; k < 1024; ++k) bits += (uint64_t)__builtin_popcountll(w[k]);
            }
        }
    *out = (storm_hip_sparse_t*)calloc(1, sizeof(**out));
    if (!*out) return STORM_HIP_ENOMEM;
    (*out)->set_bits = bits;
    return STORM_HIP_OK;
}
/* reads every block handed over (what ASan checks) and keeps only the shape */
int storm_hip_matrix_create_from_blocks(storm_hip_ctx_t* ctx, uint64_t n_rows, uint64_t n_blocks,
                                        const uint64_t* row_block_offset, const uint32_t* block_id,
                                        const uint8_t* block_kind, const uint32_t* block_n,
                                        const void* const* block_ptr, storm_hip_matrix_t** out) {
    if (n_rows == 0 || row_block_offset[0] != 0 || row_block_offset[n_rows] != n_blocks) return STORM_HIP_EINVAL;
    uint64_t touched = 0;
    uint32_t max_id = 0;
    for (uint64_t b = 0; b < n_blocks; ++b) {
        if (block_id[b] > max_id) max_id = block_id[b];
        if (block_kind[b] == 0) {
            const uint16_t* l = (const uint16_t*)block_ptr[b];
            for (uint32_t k = 0; k < block_n[b]; ++k) touched += l[k];
        } else {
            const uint64_t* w = (const uint64_t*)block_ptr[b];
            for (int k = 0; k < 1024; ++k) touched += w[k];
        }
    }
    if (max_id >= 512) return STORM_HIP_EINVAL;
    const int rc = storm_hip_matrix_create(ctx, n_rows, 1, out); /* (one word per row: the stub's band writer only needs n_rows) */
    if (rc == STORM_HIP_OK) (*out)->rows[0] = touched;
    return rc;
}
int storm_hip_sparse_create_serialized(storm_hip_ctx_t* ctx, const void* buf, uint64_t n_bytes,
                                       storm_hip_sparse_t** out) {
    (void)ctx;
    const uint8_t* p = (const uint8_t*)buf;
    uint64_t touched = 0;
    for (uint64_t i = 0; i < n_bytes; ++i) touched += p[i]; /* reads every byte handed over */
    *out = (storm_hip_sparse_t*)calloc(1, sizeof(**out));
    if (!*out) return STORM_HIP_ENOMEM;
    (*out)->set_bits = touched;
    return STORM_HIP_OK;
}
void storm_hip_sparse_destroy(storm_hip_ctx_t* ctx, storm_hip_sparse_t* s) { (void)ctx; free(s); }
int storm_hip_pairw_sparse_begin(storm_hip_ctx_t* ctx, const storm_hip_sparse_t* s, uint32_t shard_rank,
                                 uint32_t shard_count) {
    if (shard_rank >= shard_count) return STORM_HIP_EINVAL;
    ctx->pending = shard_rank == 0 ? s->set_bits : 0;
    return STORM_HIP_OK;
}
int storm_hip_pairw_sparse_end(storm_hip_ctx_t* ctx, uint64_t* h_total) {
    *h_total = ctx->pending;
    return STORM_HIP_OK;
}

/* the multi-process exchange (storm_hip_comm_*): one rank, the sum of one value is the value */
struct storm_hip_comm_s { uint32_t rank, world; };
int storm_hip_last_pass_report(storm_hip_ctx_t* ctx, uint64_t out[4]) {
    (void)ctx;
    memset(out, 0, 4 * sizeof(uint64_t));
    return STORM_HIP_OK;
}
int storm_hip_comm_unique_id(uint8_t id[STORM_HIP_COMM_ID_BYTES]) {
    memset(id, 7, STORM_HIP_COMM_ID_BYTES);
    return STORM_HIP_OK;
}
int storm_hip_comm_init_rank(storm_hip_ctx_t* ctx, const uint8_t id[STORM_HIP_COMM_ID_BYTES], uint32_t rank,
                             uint32_t world, storm_hip_comm_t** out) {
    (void)ctx; (void)id;
    *out = (storm_hip_comm_t*)calloc(1, sizeof(**out));
    if (!*out) return STORM_HIP_ENOMEM;
    (*out)->rank = rank;
    (*out)->world = world;
    return STORM_HIP_OK;
}
int storm_hip_comm_allreduce_u64(storm_hip_ctx_t* ctx, storm_hip_comm_t* comm, uint64_t* value) {
    (void)ctx; (void)comm; (void)value;
    return STORM_HIP_OK;
}
int storm_hip_comm_allreduce_u64s(storm_hip_ctx_t* ctx, storm_hip_comm_t* comm, uint64_t* values, uint32_t n) {
    (void)ctx; (void)comm; (void)values; (void)n;   /* a one-rank world: the sum is the value */
    return STORM_HIP_OK;
}
void storm_hip_comm_destroy(storm_hip_comm_t* comm) { free(comm); }

/* K5 (storm_hip_lists.hip): the stub never finds a container eligible, so the host takes the dense replica's path */
int storm_hip_rowlists_create_blocks(storm_hip_ctx_t* ctx, uint64_t n_rows, uint64_t n_blocks,
                                     const uint64_t* row_block_offset, const uint32_t* block_id,
                                     const uint8_t* block_kind, const uint32_t* block_n,
                                     const void* const* block_ptr, storm_hip_rowlists_t** out) {
    (void)ctx; (void)n_rows; (void)n_blocks; (void)row_block_offset; (void)block_id; (void)block_kind; (void)block_n; (void)block_ptr;
    *out = NULL;
    return STORM_HIP_OK;
}
int storm_hip_rowlists_create_blocks_staged(storm_hip_ctx_t* ctx, uint64_t n_rows, uint64_t n_blocks,
                                            const uint64_t* row_block_offset, const uint32_t* block_id,
                                            const uint8_t* block_kind, const uint32_t* block_n, const void* const* block_ptr,
                                            storm_hip_stage_t* stage, const uint64_t* token, storm_hip_rowlists_t** out) {
    (void)stage;
    for (uint64_t b = 0; token && b < n_blocks; ++b) (void)token[b];   /* (ASan checks the token array's extent) */
    return storm_hip_rowlists_create_blocks(ctx, n_rows, n_blocks, row_block_offset, block_id, block_kind, block_n, block_ptr, out);
}
void storm_hip_rowlists_destroy(storm_hip_ctx_t* ctx, storm_hip_rowlists_t* l) { (void)ctx; (void)l; }
int storm_hip_rowlists_worthwhile(storm_hip_ctx_t* ctx, const storm_hip_rowlists_t* l) { (void)ctx; return l == NULL; }
int storm_hip_rowlists_worthwhile_counts(storm_hip_ctx_t* ctx, uint64_t n_rows, uint64_t n_elems, uint64_t n_bits) {
    (void)ctx; (void)n_rows; (void)n_elems; (void)n_bits;
    return 0;
}
int storm_hip_rowlists_pairw_matrix_device(storm_hip_ctx_t* ctx, const storm_hip_rowlists_t* l, int op, uint32_t* d_out, uint64_t ld) {
    (void)ctx; (void)l; (void)op; (void)d_out; (void)ld;
    return STORM_HIP_EINVAL;
}
int storm_hip_rowlists_pairw_matrix(storm_hip_ctx_t* ctx, const storm_hip_rowlists_t* l, int op, uint32_t* h_out, uint64_t ld) {
    (void)ctx; (void)l; (void)op; (void)h_out; (void)ld;
    return STORM_HIP_EINVAL;
}
uint64_t storm_hip_rowlists_n_elems(const storm_hip_rowlists_t* l) { (void)l; return 0; }

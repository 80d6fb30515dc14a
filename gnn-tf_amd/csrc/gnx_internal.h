// Internal declarations shared by the libgnx.so translation units (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/gnx.h"

#define GNX_VERSION_NUM GNX_ABI_VERSION /* 0.6.0: the header's number */

namespace gnx {

void set_error(const char *fmt, ...);

#define GNX_HIP(expr)                                                                   \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            gnx::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return GNX_ERR_HIP;                                                         \
        }                                                                               \
    } while (0)

#define GNX_CHECK_ARG(cond, ...)                                                        \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            gnx::set_error(__VA_ARGS__);                                                \
            return GNX_ERR_INVALID;                                                     \
        }                                                                               \
    } while (0)

// Rows whose entry count exceeds LONG_ROW are cut into chunks of LONG_CHUNK entries that
// separate waves sum into a partial slab; a second kernel adds a row's partials in chunk
// order (fixed order => bitwise reproducible) and applies the epilogue.
#ifndef GNX_LONG_ROW
#define GNX_LONG_ROW 512
#endif
#ifndef GNX_LONG_CHUNK
#define GNX_LONG_CHUNK GNX_LONG_ROW
#endif
constexpr int LONG_ROW = GNX_LONG_ROW;
constexpr int LONG_CHUNK = GNX_LONG_CHUNK;
// Small structures (a citation graph: 10^5 rows) are bound by the LATENCY of their longest walk, not by bandwidth: one
// 512-entry row is 128 dependent gather rounds of a 16-lane group (0.13 ms -- the whole launch).  Between 2^15 and
// 2^20 rows the plan therefore cuts rows at 128 entries into 128-entry chunks, and the sub-wave kernels take chunks and short rows in one launch.
constexpr int SMALL_ROWS = 1 << 20;
constexpr int TINY_ROWS = 1 << 15;      // below this everything is cache-resident and launch-bound: an extra reduce launch costs more than it saves
// widest feature rows (floats) the K loop runs on the degree-relabelled copy of a large square graph.  Measured round 4 (RMAT
// 10M / 100M, K = 10) with the threshold at 32: C = 32 gains 7 % from the order and pays it back for the value permutation, the
// H0 permutation and the scattered last iteration (22.77 vs 22.78 ms); C = 64 loses.  16 stays.
constexpr int RELABEL_MAX_C = 16;
constexpr int SMALL_LONG_ROW = GNX_LONG_ROW < 128 ? GNX_LONG_ROW : 128;

// One CSR-like structure (the matrix itself, or its transpose).
struct Csr {
    int64_t n_rows = 0, n_cols = 0, nnz = 0;
    int64_t *rowptr = nullptr;  // [n_rows+1]
    int32_t *colidx = nullptr;  // [nnz]
    // long-row split plan
    int64_t n_long = 0, n_chunks = 0;
    int long_row = LONG_ROW, long_chunk = LONG_CHUNK;   // this structure's threshold / chunk length (set by build_long_plan)
    int32_t *long_rows = nullptr;       // [n_long] row ids
    int64_t *long_chunk_ptr = nullptr;  // [n_long+1] first chunk of each long row
    int32_t *chunk_long = nullptr;      // [n_chunks] index into long_rows
    int32_t *chunk_order = nullptr;     // [n_chunks] chunk ids sorted by the first column they touch (see gnx_graph.hip)
    // rows in stable order of descending (clamped) entry count: the rows that share a wave in the
    // sub-wave kernels then have similar lengths (power-law graphs otherwise leave most lanes idle)
    int32_t *row_order = nullptr;       // [n_rows]
    // gnx_graph_set_row_window: > 0 = the caller's numbering carries locality; rows are then taken in windows of this many consecutive
    // ids (degree-binned INSIDE a window, rows without entries still trailing everything) so that the rows in flight together are
    // neighbours in the caller's order and gather from one neighbourhood
    int64_t order_window = 0;
    int64_t *slot_beg = nullptr;        // [n_rows] first entry of row row_order[slot] ...
    int32_t *slot_cnt = nullptr;        // [n_rows] ... and its entry count: what the sub-wave kernels read instead of rowptr[row_order[slot]]
                                        // (coalesced, and no load that depends on another load before the row's entries are known)
    int64_t n_nonempty = 0;             // rows with at least one entry: the first n_nonempty slots of row_order (the rest are the empty rows)
    int32_t *nonempty_rows = nullptr;   // [n_nonempty] the same rows in ASCENDING order (only when some row is empty): what the one-wave-per-row
                                        // kernels walk while rows without entries are skipped
    // square structures: no row WITHOUT entries is referenced as a column by any entry (always so for a symmetric pattern).  Such rows
    // are alpha * H0 after every iteration and nobody gathers them: loops write them into their result only, never into work buffers
    bool empty_rows_unreferenced = false;
};

}  // namespace gnx

struct gnx_graph {
    gnx::Csr a;            // coalesced matrix
    float *raw_vals = nullptr;   // [a.nnz] summed duplicate values
    int32_t *rowidx = nullptr;   // [a.nnz] row of every coalesced entry
    // un-coalesced entries (only when duplicates exist; otherwise entries == coalesced)
    int64_t nnz_entries = 0;
    bool has_dups = false;
    float *e_vals = nullptr;     // [nnz_entries] entry values, sorted by (row, col), input order among dups
    int64_t *slot_ptr = nullptr; // [a.nnz+1] entry range of every coalesced slot
    // transposed structure (lazy)
    bool has_t = false;
    gnx::Csr t;                  // t.n_rows = a.n_cols
    int32_t *t_perm = nullptr;   // [a.nnz] coalesced slot of every transposed entry
    float *t_vals = nullptr;     // [a.nnz] scratch: values gathered into transposed order
    float *t_raw = nullptr;      // [a.nnz] raw values in transposed order (streaming column sums)
    int32_t *t_rowidx = nullptr; // [a.nnz] row of the transposed structure (= column of A) per transposed position
    uint16_t *t_mask = nullptr;  // [a.nnz] scratch: keep bits of up to 16 dropout streams per transposed position (gnx_graph_colsum_streams)
    // partial slab for long rows (grown on demand)
    float *partial = nullptr;
    size_t partial_bytes = 0;
    float *deg = nullptr;        // [a.n_cols] scratch for column sums / degree scales (lazy)
    const uint64_t *stream_offset = nullptr;   // optional device counter added to every dropout stream id of this handle
    // vertex block of a larger graph (gnx_graph_set_block): dropout draws are keyed by GLOBAL (row, col), and the degree
    // scale of local row r sits at position blk_row0_buf + r of the per-column scale vector
    int64_t blk_row0_global = 0, blk_row0_buf = 0;
    int32_t *blk_col_gid = nullptr;            // [a.n_cols] global vertex id of every column (owned), or null
    // degree-relabelled copy of a square matrix (lazy; narrow feature widths): vertex r_order[i] becomes vertex i, so the
    // rows of the hubs -- which most gathers hit -- are neighbours in memory and share cache lines.  r_order = the degree bins
    // of a.row_order (heaviest first) and, inside a bin, the vertices by the degree rank of their most popular neighbour: the
    // leaves of one hub become neighbours too, so the hub's row gathers them from consecutive lines
    bool has_r = false;
    gnx::Csr r;
    int32_t *r_order = nullptr;  // [n] new id -> old id
    int32_t *r_perm = nullptr;   // [a.nnz] coalesced slot of every relabelled entry
    float *r_vals = nullptr;     // [a.nnz] scratch: values gathered into relabelled order
    float *r_feat = nullptr;     // scratch: H0 in relabelled row order
    size_t r_feat_bytes = 0;
    const char *last_kernel = "";
};

namespace gnx {

int build_long_plan(Csr &m, hipStream_t s);
void free_csr(Csr &m);
void free_plan(Csr &m);
int ensure_transpose(gnx_graph *g, hipStream_t s);
int ensure_partial(gnx_graph *g, size_t bytes, hipStream_t s);
bool stream_is_capturing(hipStream_t s);
int ensure_relabel(gnx_graph *g, hipStream_t s);
int ensure_relabel_features(gnx_graph *g, size_t bytes, hipStream_t s);

// ---- counter RNG of the edge dropout: the same integer arithmetic as oracle/gnntf_oracle.py:hash_u24 ----------
__device__ __forceinline__ uint64_t rng_fin(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}
__device__ __forceinline__ uint32_t hash_u24(uint64_t seed, uint64_t stream, uint64_t row, uint64_t col, uint64_t dup) {
    const uint64_t k = seed ^ (stream * 0xD1342543DE82EF95ull);
    uint64_t x = rng_fin(k + row * 0x9E3779B97F4A7C15ull);
    x ^= col * 0xC2B2AE3D27D4EB4Full;
    x = rng_fin(x + dup * 0x165667B19E3779F9ull);
    return (uint32_t)(x >> 40);
}

// gnx_spmm_dropped: the values of one training iteration's dropped + re-normalised adjacency are produced inside the
// SpMM, per entry, with exactly the arithmetic of k_scale_values (gnx_prep.hip): (D[row] * drop(raw)) * D[col].
struct DropFuse {
    const float *D;        // degree scales of this (seed, stream); null = not fused
    uint64_t seed, stream;
    const uint64_t *offset; // optional device counter added to the stream id (gnx_graph_set_dropout_counter)
    uint32_t thr;          // keep iff hash >= thr
    float scale;           // 1 / (1 - p)
    int transposed;        // the structure walked is the transpose: its entry (r, c) is A[c][r]
    int col_prescaled;     // the gathered rows already carry their column's scale (written by the previous iteration's epilogue)
    int64_t row0_key, row0_D;   // vertex block: global id of row 0 / position of row 0's scale in D (0, 0 otherwise)
    const int32_t *gid;         // vertex block: global id of every column (null: the column index itself)
};

__device__ __forceinline__ float dropped_weight(const DropFuse &f, float raw, int64_t r, int64_t c) {
    const int64_t ar = f.transposed ? c : r, ac = f.transposed ? r : c;      // the entry's (row, col) in A
    const uint64_t stream = f.stream + (f.offset ? *f.offset : 0);
    const uint64_t kc = f.gid ? (uint64_t)f.gid[ac] : (uint64_t)ac;
    if (hash_u24(f.seed, stream, (uint64_t)(ar + f.row0_key), kc, 0) < f.thr) return 0.f;   // dropped: (D * 0) * D = 0 for finite scales
    if (f.col_prescaled && f.transposed)          // the gathered row (vertex ar) carries D[ar] already: what is left is the OUTPUT row's scale
        return (raw * f.scale) * f.D[ac];
    const float w = f.D[ar + f.row0_D] * (raw * f.scale);
    return f.col_prescaled ? w : w * f.D[ac];
}

struct SpmmArgs {
    const int64_t *rowptr;
    const int32_t *colidx;
    const float *vals;
    const float *diag;
    const float *X;
    int64_t ldx;
    const float *H0;
    int64_t ldh0;
    float beta, alpha;
    int act;
    float *out;
    int64_t ldo;
    const int32_t *out_rows;   // optional destination row of every result row (gnx_spmm_scatter / gnx_spmm_rows)
    bool map_h0;               // H0 rows are indexed through out_rows as well (gnx_spmm_rows)
    const float *out_scale;    // optional per-row factor applied to the finished row (gnx_spmm_dropped_chained: the NEXT iteration's column scale)
    // optional SECOND result of the same sums (gnx_spmm_dropped_back): out2[row] = acc * beta2 * (out2_scale ? out2_scale[row] : 1) --
    // no mix term, no activation; `out` then carries the running gradient sum and out2 the pre-scaled operand of the next step
    float *out2;
    int64_t ldo2;
    float beta2;
    const float *out2_scale;
    int64_t n_rows;
    int64_t n_nonempty;        // rows with entries = leading slots of row_order (Csr::n_nonempty)
    int64_t slot0;             // first row slot of this launch (a launch holds at most 2^32 work-items: huge graphs are dealt in pieces)
    int C;
    // long rows
    const int32_t *long_rows;
    const int64_t *long_chunk_ptr;
    const int32_t *chunk_long;
    const int32_t *chunk_order;
    const int32_t *row_order;
    const int64_t *slot_beg;        // Csr::slot_beg / slot_cnt (or null)
    const int32_t *slot_cnt;
    const int32_t *nonempty_rows;   // Csr::nonempty_rows (or null)
    const int32_t *row_list;        // set by the launcher: the wave-per-row kernels take row = row_list[slot] (null: row = slot)
    float *partial;
    int64_t n_long, n_chunks;
    int long_row, long_chunk;
    int tune;
    bool skip_empty;           // GNX_ACT_SKIP_EMPTY: rows without entries are left untouched
    int64_t xcd_rows;          // > 0: the row order carries locality (= Csr::order_window): the blocks that share an XCD take whole CHUNKS
    uint32_t xcd_chunk;        // of xcd_chunk consecutive blocks = one window's worth of slots (xcd_block; set per launch from xcd_rows)
    DropFuse fuse;
};

// "Done once" per DEVICE, not per process: a process may drive several GPUs (gnntf's nat.on_device, vertex blocks as threads), and a
// kernel attribute set on one device says nothing about the next.  Concurrent first calls may both do the (idempotent) work.
struct PerDeviceOnce {
    std::atomic<uint64_t> done{0};
    static int device() { int d = 0; if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); d = -1; } return d; }
    bool need(int dev) const { return dev < 0 || dev >= 64 || !((done.load(std::memory_order_acquire) >> dev) & 1u); }
    void set(int dev) { if (dev >= 0 && dev < 64) done.fetch_or(uint64_t(1) << dev, std::memory_order_release); }
};

int launch_spmm(gnx_graph *g, const Csr &m, SpmmArgs &p, hipStream_t s);                      // gnx_spmm.hip
void launch_long_rows(const SpmmArgs &p, hipStream_t s);                                       // gnx_spmm.hip: long rows only
const char *launch_spmm_dropped(const SpmmArgs &p, int vec, bool has_long, hipStream_t s);    // gnx_spmm_train.hip
// out[out_rows[r]] = act(X[in_rows[r]] . W + bias) on the matrix cores (gnx_dense.hip); row maps optional
int dense_rows(const float *X, int64_t ldx, int64_t n, int64_t F, const float *W, int64_t ldw, int64_t O, const float *bias, int act,
               const int32_t *in_rows, const int32_t *out_rows, float *out, int64_t ldo, hipStream_t s);
#ifdef GNX_TUNING
extern int tune_override;
#endif

}  // namespace gnx

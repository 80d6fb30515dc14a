// Normalisation of the adjacency on gfx950: GNN.get_adjacency of the reference
// (gnntf/core/gnn/gnn.py:36-50) and the edge dropout in front of it
// (gnntf/core/nn/layered.py:47-50).  All of this is HBM-bound integer/float streaming over
// nnz-sized arrays; no LDS tiling is needed, only coalesced slot-parallel passes.
#include "gnx_internal.h"

using namespace gnx;

namespace {

inline unsigned blocks_for(int64_t n, int bs = 256) { return (unsigned)((n + bs - 1) / bs); }

struct Drop {
    uint64_t seed, stream;
    const uint64_t *offset;    // optional device counter added to the stream id (graph-captured training steps)
    uint32_t thr;    // keep iff hash >= thr
    float scale;     // 1/(1-p)
    const float *e_vals;       // entry values (null when no duplicates)
    const int64_t *slot_ptr;   // entry range per slot (null when no duplicates)
    int64_t row0;              // vertex block (gnx_graph_set_block): global id of row 0, else 0
    const int32_t *gid;        // vertex block: global id of every column, else null
};

// the (row, col) the dropout draw of an entry is keyed by: global ids when the handle is a vertex block
__device__ __forceinline__ uint64_t key_row(const Drop &d, int64_t row) { return (uint64_t)(row + d.row0); }
__device__ __forceinline__ uint64_t key_col(const Drop &d, int64_t col) { return d.gid ? (uint64_t)d.gid[col] : (uint64_t)col; }

__device__ __forceinline__ uint64_t stream_of(const Drop &d) { return d.stream + (d.offset ? *d.offset : 0); }

// value of coalesced slot k after per-entry dropout (layered.py:50: kept * 1/(1-p), dropped -> 0)
template <bool DROPOUT>
__device__ __forceinline__ float slot_value(const float *__restrict__ raw, const Drop &d, int64_t k, int32_t row, int32_t col) {
    if (!DROPOUT) return raw[k];
    if (d.slot_ptr == nullptr)
        return hash_u24(d.seed, stream_of(d), key_row(d, row), key_col(d, col), 0) >= d.thr ? raw[k] * d.scale : 0.f;
    float acc = 0.f;
    const int64_t b = d.slot_ptr[k], e = d.slot_ptr[k + 1];
    const uint64_t kr = key_row(d, row), kc = key_col(d, col);
    for (int64_t i = b; i < e; ++i)
        if (hash_u24(d.seed, stream_of(d), kr, kc, (uint64_t)(i - b)) >= d.thr) acc += d.e_vals[i] * d.scale;
    return acc;
}

// the same for position p of the transposed structure; without duplicates the raw value is read from
// the transposed-order copy (a streaming read instead of a gather through t_perm)
template <bool DROPOUT>
__device__ __forceinline__ float t_value(const float *__restrict__ raw, const float *__restrict__ t_raw,
                                         const int32_t *__restrict__ t_perm, const Drop &d, int64_t p, int32_t row, int32_t col) {
    if (d.slot_ptr != nullptr) return slot_value<DROPOUT>(raw, d, t_perm[p], row, col);
    const float v = t_raw[p];
    if (!DROPOUT) return v;
    return hash_u24(d.seed, stream_of(d), key_row(d, row), key_col(d, col), 0) >= d.thr ? v * d.scale : 0.f;
}

// column sums over the transposed structure: 8 lanes per column, fixed reduction tree.
template <bool DROPOUT>
__global__ void k_colsum_short(const int64_t *__restrict__ t_rowptr, const int32_t *__restrict__ t_colidx,
                               const int32_t *__restrict__ t_perm, const float *__restrict__ raw,
                               const float *__restrict__ t_raw, Drop d, int64_t n_cols, int long_row, float *__restrict__ out) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t j = gid >> 3;
    const int sub = (int)(gid & 7);
    float acc = 0.f;
    bool is_long = false;
    if (j < n_cols) {
        const int64_t b = t_rowptr[j], e = t_rowptr[j + 1];
        is_long = (e - b) > long_row;
        if (!is_long)
            for (int64_t p = b + sub; p < e; p += 8) acc += t_value<DROPOUT>(raw, t_raw, t_perm, d, p, t_colidx[p], (int32_t)j);
    }
    acc += __shfl_xor(acc, 4);
    acc += __shfl_xor(acc, 2);
    acc += __shfl_xor(acc, 1);
    if (j < n_cols && sub == 0 && !is_long) out[j] = acc;
}

// one 256-thread block per long column; strided partial sums, then a fixed LDS tree.
template <bool DROPOUT>
__global__ __launch_bounds__(256) void k_colsum_long(const int64_t *__restrict__ t_rowptr, const int32_t *__restrict__ t_colidx,
                                                     const int32_t *__restrict__ t_perm, const float *__restrict__ raw,
                                                     const float *__restrict__ t_raw, Drop d,
                                                     const int32_t *__restrict__ long_rows, float *__restrict__ out) {
    __shared__ float red[256];
    const int32_t j = long_rows[blockIdx.x];
    const int64_t b = t_rowptr[j], e = t_rowptr[j + 1];
    float acc = 0.f;
    for (int64_t p = b + threadIdx.x; p < e; p += 256) acc += t_value<DROPOUT>(raw, t_raw, t_perm, d, p, t_colidx[p], j);
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[j] = red[0];
}

// Column sums of NS consecutive dropout streams in ONE pass over the transposed structure (no duplicate entries): the K
// iterations of a training step each drop the edges independently, so their column sums differ only in the hash -- the
// structure and the raw values are read once for all of them.  Same lane mapping and reduction tree as k_colsum_short / _long,
// so every stream's sums are bit for bit what the single-stream kernels give.  out[s * n_cols + j].
template <int NS>
__global__ void k_colsum_short_multi(const int64_t *__restrict__ t_rowptr, const int32_t *__restrict__ t_colidx,
                                     const float *__restrict__ t_raw, Drop d, int64_t n_cols, int long_row, float *__restrict__ out) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t j = gid >> 3;
    const int sub = (int)(gid & 7);
    float acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = 0.f;
    bool is_long = false;
    if (j < n_cols) {
        const int64_t b = t_rowptr[j], e = t_rowptr[j + 1];
        is_long = (e - b) > long_row;
        if (!is_long) {
            const uint64_t kc = key_col(d, j);
            for (int64_t p = b + sub; p < e; p += 8) {
                const float v = t_raw[p] * d.scale;
                const uint64_t row = key_row(d, t_colidx[p]);
#pragma unroll
                for (int s = 0; s < NS; ++s) acc[s] += hash_u24(d.seed, stream_of(d) + s, row, kc, 0) >= d.thr ? v : 0.f;
            }
        }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        float a = acc[s];
        a += __shfl_xor(a, 4);
        a += __shfl_xor(a, 2);
        a += __shfl_xor(a, 1);
        if (j < n_cols && sub == 0 && !is_long) out[(int64_t)s * n_cols + j] = a;
    }
}

template <int NS>
__global__ __launch_bounds__(256) void k_colsum_long_multi(const int64_t *__restrict__ t_rowptr, const int32_t *__restrict__ t_colidx,
                                                           const float *__restrict__ t_raw, Drop d, const int32_t *__restrict__ long_rows,
                                                           int64_t n_cols, float *__restrict__ out) {
    __shared__ float red[256];
    const int32_t j = long_rows[blockIdx.x];
    const int64_t b = t_rowptr[j], e = t_rowptr[j + 1];
    float acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = 0.f;
    const uint64_t kc = key_col(d, j);
    for (int64_t p = b + threadIdx.x; p < e; p += 256) {
        const float v = t_raw[p] * d.scale;
        const uint64_t row = key_row(d, t_colidx[p]);
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s] += hash_u24(d.seed, stream_of(d) + s, row, kc, 0) >= d.thr ? v : 0.f;
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        __syncthreads();
        red[threadIdx.x] = acc[s];
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[(int64_t)s * n_cols + j] = red[0];
    }
}

// The same sums in TWO passes (training steps: K streams at once, gnx_graph_colsum_streams).  In the kernels above the hashes sit
// inside a loop whose trip count differs from lane to lane (8 lanes per column, columns of every length in one wave): most of the
// VALU time goes to lanes that have run out of entries (measured round 4: 5.5 ms for 8 streams over 10^8 entries, hash-bound).
// Pass 1 hashes with EVERY lane busy -- one lane per transposed position, all streams of the batch, the keep bits packed into one
// 16-bit word per entry; pass 2 is the column walk of k_colsum_short_multi with a 2-byte read where the hashes were.  Same lane
// mapping, same order of additions, same reduction tree: bit for bit the sums of the kernels above.
__global__ __launch_bounds__(256) void k_keep_masks(const int32_t *__restrict__ t_rowidx /* column of A */, const int32_t *__restrict__ t_colidx /* row of A */,
                                                    int64_t nnz, Drop d, int ns, uint16_t *__restrict__ mask) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nnz) return;
    const uint64_t row = key_row(d, t_colidx[p]), kc = key_col(d, t_rowidx[p]);
    const uint64_t stream = stream_of(d);
    uint32_t bits = 0;
#pragma unroll 2
    for (int s = 0; s < ns; ++s) bits |= (hash_u24(d.seed, stream + s, row, kc, 0) >= d.thr ? 1u : 0u) << s;
    mask[p] = (uint16_t)bits;
}

// ``list`` / ``n_slots``: the columns WITH entries in ascending order (the transposed structure's nonempty_rows; the sums of the others
// are zeroed by the caller and their lanes never launched), or null / n_cols.
template <int NS>
__global__ void k_colsum_short_masked(const int64_t *__restrict__ t_rowptr, const float *__restrict__ t_raw, const uint16_t *__restrict__ mask,
                                      float scale, int ns, int64_t n_cols, int long_row, const int32_t *__restrict__ list, int64_t n_slots,
                                      float *__restrict__ out) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t slot = gid >> 3;
    const int64_t j = slot < n_slots ? (list ? (int64_t)list[slot] : slot) : n_cols;
    const int sub = (int)(gid & 7);
    float acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = 0.f;
    bool is_long = false;
    if (j < n_cols) {
        const int64_t b = t_rowptr[j], e = t_rowptr[j + 1];
        is_long = (e - b) > long_row;
        if (!is_long) {
            for (int64_t p = b + sub; p < e; p += 8) {
                const float v = t_raw[p] * scale;
                const uint32_t m = mask[p];
#pragma unroll
                for (int s = 0; s < NS; ++s) acc[s] += ((m >> s) & 1u) ? v : 0.f;
            }
        }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        float a = acc[s];
        a += __shfl_xor(a, 4);
        a += __shfl_xor(a, 2);
        a += __shfl_xor(a, 1);
        if (j < n_cols && sub == 0 && !is_long && s < ns) out[(int64_t)s * n_cols + j] = a;
    }
}

template <int NS>
__global__ __launch_bounds__(256) void k_colsum_long_masked(const int64_t *__restrict__ t_rowptr, const float *__restrict__ t_raw,
                                                            const uint16_t *__restrict__ mask, float scale, int ns,
                                                            const int32_t *__restrict__ long_rows, int64_t n_cols, float *__restrict__ out) {
    __shared__ float red[256];
    const int32_t j = long_rows[blockIdx.x];
    const int64_t b = t_rowptr[j], e = t_rowptr[j + 1];
    float acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = 0.f;
    for (int64_t p = b + threadIdx.x; p < e; p += 256) {
        const float v = t_raw[p] * scale;
        const uint32_t m = mask[p];
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s] += ((m >> s) & 1u) ? v : 0.f;
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (s < ns) {                                                // (block-uniform)
            __syncthreads();
            red[threadIdx.x] = acc[s];
            __syncthreads();
            for (int w = 128; w > 0; w >>= 1) {
                if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
                __syncthreads();
            }
            if (threadIdx.x == 0) out[(int64_t)s * n_cols + j] = red[0];
        }
    }
}

// gnn.py:41 / :44 with optional "+I before" folded in as +1 on every column sum
__global__ void k_degree_scale(float *__restrict__ d, int64_t n, int normalized, float eye) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    float x = d[j] + eye;
    if (normalized == GNX_NORM_SYMMETRIC) x = sqrtf(x);
    d[j] = (x != 0.f) ? 1.0f / x : 0.f;   // tf.math.divide_no_nan(1., x)
}

// gnn.py:42 / :45: v_ij <- (rs[i] * v_ij) * cs[j]
template <bool DROPOUT>
__global__ void k_scale_values(const int32_t *__restrict__ rowidx, const int32_t *__restrict__ colidx,
                               const float *__restrict__ raw, Drop d, const float *__restrict__ rs,
                               const float *__restrict__ cs, int64_t nnz, float *__restrict__ out) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nnz) return;
    const int32_t r = rowidx[k], c = colidx[k];
    float v = slot_value<DROPOUT>(raw, d, k, r, c);
    if (rs) v = rs[r] * v;
    if (cs) v = v * cs[c];
    out[k] = v;
}

// the same values written in the order of the transposed structure: position p holds entry
// (row = t_colidx[p], col = t_rowidx[p]) of A
template <bool DROPOUT>
__global__ void k_scale_values_t(const int32_t *__restrict__ t_rowidx, const int32_t *__restrict__ t_colidx,
                                 const int32_t *__restrict__ t_perm, const float *__restrict__ raw,
                                 const float *__restrict__ t_raw, Drop d, const float *__restrict__ rs,
                                 const float *__restrict__ cs, int64_t nnz, float *__restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nnz) return;
    const int32_t r = t_colidx[p], c = t_rowidx[p];
    float v = t_value<DROPOUT>(raw, t_raw, t_perm, d, p, r, c);
    if (rs) v = rs[r] * v;
    if (cs) v = v * cs[c];
    out[p] = v;
}

// diagonal weight of the identity added by add_eye (gnn.py:39,49)
__global__ void k_diag(const float *__restrict__ deg, int64_t n, int mode /*0: ones, 1: deg^2, 2: deg*/, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = mode == 0 ? 1.0f : (mode == 1 ? deg[i] * 1.0f * deg[i] : deg[i]);
}

int make_drop(gnx_graph *g, float p, uint64_t seed, uint64_t stream_id, Drop &d) {
    GNX_CHECK_ARG(p >= 0.f && p < 1.f, "dropout rate %g outside [0, 1)", (double)p);
    d.seed = seed; d.stream = stream_id; d.offset = g->stream_offset;
    d.thr = (uint32_t)((double)p * 16777216.0);
    d.scale = 1.0f / (1.0f - p);
    d.e_vals = g->has_dups ? g->e_vals : nullptr;
    d.slot_ptr = g->has_dups ? g->slot_ptr : nullptr;
    d.row0 = g->blk_row0_global; d.gid = g->blk_col_gid;
    return GNX_OK;
}

int ensure_deg(gnx_graph *g) {
    if (g->deg) return GNX_OK;
    GNX_HIP(hipMalloc((void **)&g->deg, (g->a.n_cols ? g->a.n_cols : 1) * sizeof(float)));
    return GNX_OK;
}

}  // namespace

extern "C" {

int gnx_graph_set_dropout_counter(gnx_graph_t g, const uint64_t *d_counter) {
    GNX_CHECK_ARG(g != nullptr, "gnx_graph_set_dropout_counter: NULL handle");
    g->stream_offset = d_counter;
    return GNX_OK;
}

int gnx_graph_set_block(gnx_graph_t g, int64_t row0_global, int64_t row0_buf, const int32_t *d_col_gid, void *stream) {
    GNX_CHECK_ARG(g != nullptr, "gnx_graph_set_block: NULL handle");
    if (d_col_gid == nullptr) {                              // back to a stand-alone graph
        if (g->blk_col_gid) (void)hipFree(g->blk_col_gid);
        g->blk_col_gid = nullptr; g->blk_row0_global = 0; g->blk_row0_buf = 0;
        return GNX_OK;
    }
    GNX_CHECK_ARG(row0_global >= 0 && row0_buf >= 0 && row0_buf + g->a.n_rows <= g->a.n_cols,
                  "gnx_graph_set_block: the %lld rows do not fit behind column %lld of %lld", (long long)g->a.n_rows,
                  (long long)row0_buf, (long long)g->a.n_cols);
    if (!g->blk_col_gid) GNX_HIP(hipMalloc((void **)&g->blk_col_gid, (g->a.n_cols ? g->a.n_cols : 1) * sizeof(int32_t)));
    GNX_HIP(hipMemcpyAsync(g->blk_col_gid, d_col_gid, g->a.n_cols * sizeof(int32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    g->blk_row0_global = row0_global; g->blk_row0_buf = row0_buf;
    return GNX_OK;
}

int gnx_graph_colsum(gnx_graph_t g, float dropout_p, uint64_t seed, uint64_t stream_id, float *d_colsum_out, void *stream) {
    GNX_CHECK_ARG(g != nullptr && d_colsum_out != nullptr, "gnx_graph_colsum: NULL argument");
    hipStream_t s = (hipStream_t)stream;
    int rc = ensure_transpose(g, s);
    if (rc != GNX_OK) return rc;
    Drop d;
    rc = make_drop(g, dropout_p, seed, stream_id, d);
    if (rc != GNX_OK) return rc;
    const Csr &t = g->t;
    if (t.n_rows == 0) return GNX_OK;
    const bool drop = dropout_p > 0.f;
    const unsigned nb = blocks_for(t.n_rows * 8);
    if (drop) hipLaunchKernelGGL(k_colsum_short<true>, dim3(nb), dim3(256), 0, s, t.rowptr, t.colidx, g->t_perm, g->raw_vals, g->t_raw, d,
                                 t.n_rows, t.long_row, d_colsum_out);
    else      hipLaunchKernelGGL(k_colsum_short<false>, dim3(nb), dim3(256), 0, s, t.rowptr, t.colidx, g->t_perm, g->raw_vals, g->t_raw, d,
                                 t.n_rows, t.long_row, d_colsum_out);
    if (t.n_long > 0) {
        if (drop) hipLaunchKernelGGL(k_colsum_long<true>, dim3((unsigned)t.n_long), dim3(256), 0, s, t.rowptr, t.colidx, g->t_perm, g->raw_vals,
                                     g->t_raw, d, t.long_rows, d_colsum_out);
        else      hipLaunchKernelGGL(k_colsum_long<false>, dim3((unsigned)t.n_long), dim3(256), 0, s, t.rowptr, t.colidx, g->t_perm, g->raw_vals,
                                     g->t_raw, d, t.long_rows, d_colsum_out);
    }
    GNX_HIP(hipGetLastError());
    return GNX_OK;
}

int gnx_graph_colsum_streams(gnx_graph_t g, float dropout_p, uint64_t seed, uint64_t first_stream, int n_streams, float *d_colsum_out,
                             void *stream) {
    GNX_CHECK_ARG(g != nullptr && d_colsum_out != nullptr, "gnx_graph_colsum_streams: NULL argument");
    GNX_CHECK_ARG(n_streams >= 1 && n_streams <= 4096, "gnx_graph_colsum_streams: bad stream count %d", n_streams);
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = g->a.n_cols;
    if (g->has_dups || dropout_p <= 0.f) {              // entry lists / no dropout: one stream at a time through the general kernels
        for (int k = 0; k < n_streams; ++k) {
            int rc = gnx_graph_colsum(g, dropout_p, seed, first_stream + k, d_colsum_out + (int64_t)k * n, stream);
            if (rc != GNX_OK) return rc;
        }
        return GNX_OK;
    }
    int rc = ensure_transpose(g, s);
    if (rc != GNX_OK) return rc;
    const Csr &t = g->t;
    if (t.n_rows == 0) return GNX_OK;
    const unsigned nb = blocks_for(t.n_rows * 8);
    if (n_streams >= 2 && t.nnz > 0) {
        // two passes per batch of up to 16 streams: keep bits with every lane busy, then the column walk over 2-byte masks
        if (!g->t_mask) {
            if (stream_is_capturing(s)) {
                set_error("gnx_graph_colsum_streams: the keep-bit scratch of this handle would have to be allocated while the stream is being "
                          "captured: call gnx_graph_reserve(handle, C, GNX_RESERVE_TRANSPOSED) or run the call once eagerly before capturing");
                return GNX_ERR_UNSUPPORTED;
            }
            GNX_HIP(hipMalloc((void **)&g->t_mask, (size_t)t.nnz * sizeof(uint16_t)));
        }
        for (int k0 = 0; k0 < n_streams; k0 += 16) {
            const int ns = n_streams - k0 < 16 ? n_streams - k0 : 16;
            Drop d;
            rc = make_drop(g, dropout_p, seed, first_stream + k0, d);
            if (rc != GNX_OK) return rc;
            float *out = d_colsum_out + (int64_t)k0 * n;
            hipLaunchKernelGGL(k_keep_masks, dim3(blocks_for(t.nnz)), dim3(256), 0, s, g->t_rowidx, t.colidx, t.nnz, d, ns, g->t_mask);
            // only the columns that have entries (ascending: the stores stay in order); the sums of the others are zero
            const bool trim = t.nonempty_rows != nullptr && t.n_nonempty < t.n_rows;
            const int64_t n_slots = trim ? t.n_nonempty : t.n_rows;
            if (trim) GNX_HIP(hipMemsetAsync(out, 0, (size_t)ns * (size_t)n * sizeof(float), s));
            const unsigned nbm = blocks_for(n_slots * 8);
#define GNX_MASKED(NS)                                                                                                                 \
            do {                                                                                                                       \
                hipLaunchKernelGGL(k_colsum_short_masked<NS>, dim3(nbm), dim3(256), 0, s, t.rowptr, g->t_raw, g->t_mask, d.scale, ns, \
                                   t.n_rows, t.long_row, trim ? t.nonempty_rows : nullptr, n_slots, out);                              \
                if (t.n_long > 0)                                                                                                      \
                    hipLaunchKernelGGL(k_colsum_long_masked<NS>, dim3((unsigned)t.n_long), dim3(256), 0, s, t.rowptr, g->t_raw,        \
                                       g->t_mask, d.scale, ns, t.long_rows, t.n_rows, out);                                            \
            } while (0)
            if (ns > 8) GNX_MASKED(16);
            else if (ns > 4) GNX_MASKED(8);
            else if (ns > 2) GNX_MASKED(4);
            else GNX_MASKED(2);
#undef GNX_MASKED
        }
        GNX_HIP(hipGetLastError());
        return GNX_OK;
    }
    for (int k0 = 0; k0 < n_streams;) {                 // one stream: a single pass
        Drop d;
        rc = make_drop(g, dropout_p, seed, first_stream + k0, d);
        if (rc != GNX_OK) return rc;
        float *out = d_colsum_out + (int64_t)k0 * n;
        hipLaunchKernelGGL(k_colsum_short_multi<1>, dim3(nb), dim3(256), 0, s, t.rowptr, t.colidx, g->t_raw, d, t.n_rows, t.long_row, out);
        if (t.n_long > 0)
            hipLaunchKernelGGL(k_colsum_long_multi<1>, dim3((unsigned)t.n_long), dim3(256), 0, s, t.rowptr, t.colidx, g->t_raw, d,
                               t.long_rows, t.n_rows, out);
        k0 += 1;
    }
    GNX_HIP(hipGetLastError());
    return GNX_OK;
}

int gnx_degree_scale(float *d_deg, int64_t n, int normalized, int add_eye_before, void *stream) {
    GNX_CHECK_ARG(n >= 0 && (n == 0 || d_deg != nullptr), "gnx_degree_scale: NULL array");
    GNX_CHECK_ARG(normalized == GNX_NORM_SYMMETRIC || normalized == GNX_NORM_BIPARTITE,
                  "Invalid matrix normalization");
    if (n == 0) return GNX_OK;
    hipLaunchKernelGGL(k_degree_scale, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, d_deg, n, normalized,
                       add_eye_before ? 1.0f : 0.0f);
    GNX_HIP(hipGetLastError());
    return GNX_OK;
}

int gnx_graph_scale_values(gnx_graph_t g, float dropout_p, uint64_t seed, uint64_t stream_id, const float *d_row_scale,
                           const float *d_col_scale, float *d_vals_out, void *stream) {
    GNX_CHECK_ARG(g != nullptr, "gnx_graph_scale_values: NULL handle");
    GNX_CHECK_ARG(g->a.nnz == 0 || d_vals_out != nullptr, "gnx_graph_scale_values: NULL output");
    Drop d;
    int rc = make_drop(g, dropout_p, seed, stream_id, d);
    if (rc != GNX_OK) return rc;
    if (g->a.nnz == 0) return GNX_OK;
    const unsigned nb = blocks_for(g->a.nnz);
    hipStream_t s = (hipStream_t)stream;
    if (dropout_p > 0.f) hipLaunchKernelGGL(k_scale_values<true>, dim3(nb), dim3(256), 0, s, g->rowidx, g->a.colidx, g->raw_vals, d, d_row_scale,
                                            d_col_scale, g->a.nnz, d_vals_out);
    else                 hipLaunchKernelGGL(k_scale_values<false>, dim3(nb), dim3(256), 0, s, g->rowidx, g->a.colidx, g->raw_vals, d, d_row_scale,
                                            d_col_scale, g->a.nnz, d_vals_out);
    GNX_HIP(hipGetLastError());
    return GNX_OK;
}

static int scale_values_any(gnx_graph_t g, bool transposed, float dropout_p, uint64_t seed, uint64_t stream_id,
                            const float *rs, const float *cs, float *out, void *stream) {
    if (!transposed) return gnx_graph_scale_values(g, dropout_p, seed, stream_id, rs, cs, out, stream);
    hipStream_t s = (hipStream_t)stream;
    int rc = ensure_transpose(g, s);
    if (rc != GNX_OK) return rc;
    Drop d;
    rc = make_drop(g, dropout_p, seed, stream_id, d);
    if (rc != GNX_OK) return rc;
    if (g->a.nnz == 0) return GNX_OK;
    const unsigned nb = blocks_for(g->a.nnz);
    if (dropout_p > 0.f) hipLaunchKernelGGL(k_scale_values_t<true>, dim3(nb), dim3(256), 0, s, g->t_rowidx, g->t.colidx, g->t_perm, g->raw_vals,
                                            g->t_raw, d, rs, cs, g->a.nnz, out);
    else                 hipLaunchKernelGGL(k_scale_values_t<false>, dim3(nb), dim3(256), 0, s, g->t_rowidx, g->t.colidx, g->t_perm, g->raw_vals,
                                            g->t_raw, d, rs, cs, g->a.nnz, out);
    GNX_HIP(hipGetLastError());
    return GNX_OK;
}

static int normalize_impl(const char *fn, gnx_graph_t g, bool transposed, int normalized, int add_eye, float dropout_p,
                          uint64_t seed, uint64_t stream_id, float *d_vals_out, float *d_diag_out, void *stream) {
    GNX_CHECK_ARG(g != nullptr, "%s: NULL handle", fn);
    GNX_CHECK_ARG(normalized == GNX_NORM_NONE || normalized == GNX_NORM_SYMMETRIC || normalized == GNX_NORM_BIPARTITE,
                  "Invalid matrix normalization");
    GNX_CHECK_ARG(add_eye == GNX_EYE_NONE || add_eye == GNX_EYE_BEFORE || add_eye == GNX_EYE_AFTER,
                  "%s: invalid add_eye %d", fn, add_eye);
    GNX_CHECK_ARG(g->a.nnz == 0 || d_vals_out != nullptr, "%s: NULL output", fn);
    GNX_CHECK_ARG(add_eye == GNX_EYE_NONE || d_diag_out != nullptr, "%s: add_eye needs d_diag_out", fn);
    GNX_CHECK_ARG((normalized == GNX_NORM_NONE && add_eye == GNX_EYE_NONE) || g->a.n_rows == g->a.n_cols,
                  "%s: normalisation / add_eye need a square graph (%lld x %lld)", fn,
                  (long long)g->a.n_rows, (long long)g->a.n_cols);
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = g->a.n_rows;
    int rc;
    if (normalized == GNX_NORM_NONE) {
        rc = scale_values_any(g, transposed, dropout_p, seed, stream_id, nullptr, nullptr, d_vals_out, stream);
        if (rc != GNX_OK) return rc;
        if (add_eye != GNX_EYE_NONE && n > 0)
            hipLaunchKernelGGL(k_diag, dim3(blocks_for(n)), dim3(256), 0, s, (const float *)nullptr, n, 0, d_diag_out);
        GNX_HIP(hipGetLastError());
        return GNX_OK;
    }
    rc = ensure_deg(g);
    if (rc != GNX_OK) return rc;
    rc = gnx_graph_colsum(g, dropout_p, seed, stream_id, g->deg, stream);
    if (rc != GNX_OK) return rc;
    rc = gnx_degree_scale(g->deg, g->a.n_cols, normalized, add_eye == GNX_EYE_BEFORE, stream);
    if (rc != GNX_OK) return rc;
    rc = scale_values_any(g, transposed, dropout_p, seed, stream_id, g->deg,
                          normalized == GNX_NORM_SYMMETRIC ? g->deg : nullptr, d_vals_out, stream);
    if (rc != GNX_OK) return rc;
    if (add_eye != GNX_EYE_NONE && n > 0) {
        const int mode = add_eye == GNX_EYE_AFTER ? 0 : (normalized == GNX_NORM_SYMMETRIC ? 1 : 2);
        hipLaunchKernelGGL(k_diag, dim3(blocks_for(n)), dim3(256), 0, s, g->deg, n, mode, d_diag_out);
    }
    GNX_HIP(hipGetLastError());
    return GNX_OK;
}

int gnx_graph_normalize(gnx_graph_t g, int normalized, int add_eye, float dropout_p, uint64_t seed, uint64_t stream_id,
                        float *d_vals_out, float *d_diag_out, void *stream) {
    return normalize_impl("gnx_graph_normalize", g, false, normalized, add_eye, dropout_p, seed, stream_id, d_vals_out,
                          d_diag_out, stream);
}

int gnx_graph_normalize_t(gnx_graph_t g, int normalized, int add_eye, float dropout_p, uint64_t seed, uint64_t stream_id,
                          float *d_vals_t_out, float *d_diag_out, void *stream) {
    return normalize_impl("gnx_graph_normalize_t", g, true, normalized, add_eye, dropout_p, seed, stream_id, d_vals_t_out,
                          d_diag_out, stream);
}

}  // extern "C"

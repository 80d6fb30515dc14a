// Graph handle: COO -> sorted entries -> coalesced CSR (+ lazy transposed structure) on gfx950.
// Replaces the tf.sparse.SparseTensor that gnntf's graph2adj returns
// (reference gnntf/core/gnn/graph_manipulation.py:24-31) as the container of the adjacency.
#include <cstring>
#include <string.h>

#include "gnx_internal.h"
#include <algorithm>

#include <rocprim/rocprim.hpp>

namespace gnx {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// small RAII holder so early returns do not leak temporaries
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    template <class T> T *as() { return static_cast<T *>(p); }
    void *release() { void *q = p; p = nullptr; return q; }
};

static inline unsigned blocks_for(int64_t n, int bs = 256) { return (unsigned)((n + bs - 1) / bs); }

static unsigned bits_for(uint64_t max_key_exclusive) {
    unsigned b = 1;
    while (b < 64 && (max_key_exclusive >> b) != 0) ++b;
    return b;
}

// ---- kernels ---------------------------------------------------------------------------
__global__ void k_make_keys(const int64_t *__restrict__ indices, int64_t nnz, int64_t n_rows, int64_t n_cols,
                            uint64_t *__restrict__ keys, int *__restrict__ bad) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    int64_t r = indices[2 * e], c = indices[2 * e + 1];
    if (r < 0 || r >= n_rows || c < 0 || c >= n_cols) {
        atomicExch(bad, 1);
        r = 0; c = 0;
    }
    keys[e] = (uint64_t)r * (uint64_t)n_cols + (uint64_t)c;
}

__global__ void k_heads(const uint64_t *__restrict__ keys, int64_t nnz, int32_t *__restrict__ head) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    head[e] = (e == 0 || keys[e] != keys[e - 1]) ? 1 : 0;
}

// slot = inclusive_scan(head) - 1
__global__ void k_fill_slots(const uint64_t *__restrict__ keys, const int32_t *__restrict__ scan, int64_t nnz,
                             int64_t n_cols, int32_t *__restrict__ colidx, int32_t *__restrict__ rowidx,
                             int64_t *__restrict__ slot_ptr /* may be null */) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const bool is_head = (e == 0) || (scan[e] != scan[e - 1]);
    if (!is_head) return;
    const int32_t s = scan[e] - 1;
    const uint64_t k = keys[e];
    rowidx[s] = (int32_t)(k / (uint64_t)n_cols);
    colidx[s] = (int32_t)(k % (uint64_t)n_cols);
    if (slot_ptr) slot_ptr[s] = e;
}

__global__ void k_sum_slots(const float *__restrict__ e_vals, const int64_t *__restrict__ slot_ptr, int64_t nslots,
                            float *__restrict__ out) {
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslots) return;
    float acc = 0.f;
    for (int64_t e = slot_ptr[s]; e < slot_ptr[s + 1]; ++e) acc += e_vals[e];  // input order
    out[s] = acc;
}

// ptr[r] = first position p with sorted_rows[p] >= r  (r in [0, n_rows]); sorted_rows int32 ascending
__global__ void k_lower_bound_rows(const int32_t *__restrict__ sorted_rows, int64_t nnz, int64_t n_rows,
                                   int64_t *__restrict__ ptr) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_rows) return;
    int64_t lo = 0, hi = nnz;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((int64_t)sorted_rows[mid] < r) lo = mid + 1; else hi = mid;
    }
    ptr[r] = lo;
}

__global__ void k_rows_from_ptr(const int64_t *__restrict__ rowptr, int64_t n_rows, int32_t *__restrict__ rowidx) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    for (int64_t e = rowptr[r]; e < rowptr[r + 1]; ++e) rowidx[e] = (int32_t)r;
}

__global__ void k_check_csr(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, int64_t n_rows,
                            int64_t n_cols, int64_t nnz, int *__restrict__ bad) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    int64_t b = rowptr[r], e = rowptr[r + 1];
    if (b > e || b < 0 || e > nnz || (r == 0 && b != 0) || (r == n_rows - 1 && e != nnz)) { atomicExch(bad, 1); return; }
    for (int64_t k = b; k < e; ++k) {
        int32_t c = colidx[k];
        if (c < 0 || c >= n_cols || (k > b && colidx[k - 1] >= c)) { atomicExch(bad, 1); return; }
    }
}

__global__ void k_flag_long(const int64_t *__restrict__ rowptr, int64_t n_rows, int long_row, int long_chunk,
                            int32_t *__restrict__ flag, int64_t *__restrict__ cnt) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    int64_t d = rowptr[r + 1] - rowptr[r];
    bool lg = d > long_row;
    flag[r] = lg ? 1 : 0;
    cnt[r] = lg ? (d + long_chunk - 1) / long_chunk : 0;
}

// sort key of the degree-binned row order: clamp - min(entries, clamp), so an ascending stable sort puts the
// heaviest rows first and keeps ascending row ids inside a bin.  The clamp covers every row the sub-wave kernels take
// (up to the threshold entries): the rows that share a wave then have EQUAL lengths, also in the 256..512 range
__global__ void k_order_keys(const int64_t *__restrict__ rowptr, int64_t n_rows, int clamp, uint16_t *__restrict__ keys,
                             int32_t *__restrict__ ids) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    int64_t d = rowptr[r + 1] - rowptr[r];
    keys[r] = (uint16_t)(clamp - (d < clamp ? d : clamp));
    ids[r] = (int32_t)r;
}

// The same bins inside WINDOWS of the caller's numbering (Csr::order_window): key = (window, bin); rows without entries get the
// window past the last one, so that they still trail the whole order (the launchers cut the order there)
__global__ void k_order_keys_windowed(const int64_t *__restrict__ rowptr, int64_t n_rows, int clamp, int64_t window, uint32_t n_windows,
                                      unsigned bin_bits, uint32_t *__restrict__ keys, int32_t *__restrict__ ids) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    int64_t d = rowptr[r + 1] - rowptr[r];
    const uint32_t w = d == 0 ? n_windows : (uint32_t)(r / window);
    keys[r] = (w << bin_bits) | (uint32_t)(clamp - (d < clamp ? d : clamp));
    ids[r] = (int32_t)r;
}

__global__ void k_slot_ptrs(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ row_order, int64_t n_rows,
                            int64_t *__restrict__ slot_beg, int32_t *__restrict__ slot_cnt) {
    int64_t sidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (sidx >= n_rows) return;
    const int64_t r = row_order[sidx], b = rowptr[r], e = rowptr[r + 1];
    slot_beg[sidx] = b;
    slot_cnt[sidx] = (int32_t)(e - b < 0x7fffffff ? e - b : 0x7fffffff);
}

__global__ void k_count_nonempty(const int64_t *__restrict__ rowptr, int64_t n_rows, unsigned long long *__restrict__ count) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool has = r < n_rows && rowptr[r + 1] > rowptr[r];
    const unsigned long long votes = __popcll(__ballot(has));
    if ((threadIdx.x & 63) == 0 && votes) atomicAdd(count, votes);
}

__global__ void k_mark_referenced(const int32_t *__restrict__ colidx, int64_t nnz, uint8_t *__restrict__ flag) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nnz) flag[colidx[k]] = 1;
}

__global__ void k_count_empty_referenced(const int64_t *__restrict__ rowptr, int64_t n, const uint8_t *__restrict__ flag, int *__restrict__ count) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n && rowptr[r + 1] == rowptr[r] && flag[r]) atomicAdd(count, 1);
}

__global__ void k_fill_long(const int64_t *__restrict__ rowptr, int64_t n_rows, int long_row, int long_chunk,
                            const int32_t *__restrict__ pos,
                            const int64_t *__restrict__ cpos, int32_t *__restrict__ long_rows,
                            int64_t *__restrict__ long_chunk_ptr, int32_t *__restrict__ chunk_long) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    int64_t d = rowptr[r + 1] - rowptr[r];
    if (d <= long_row) return;
    const int32_t p = pos[r];
    const int64_t c0 = cpos[r];
    const int64_t nc = (d + long_chunk - 1) / long_chunk;
    long_rows[p] = (int32_t)r;
    long_chunk_ptr[p] = c0;
    for (int64_t c = 0; c < nc; ++c) chunk_long[c0 + c] = p;
}

// key of a long-row chunk = the first column it touches: chunks are then processed in column-window order, so
// that at any time the long-row waves gather from one window of columns and its hub rows stay cached
__global__ void k_chunk_keys(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
                             const int32_t *__restrict__ long_rows, const int64_t *__restrict__ long_chunk_ptr,
                             const int32_t *__restrict__ chunk_long, int64_t n_chunks, int long_chunk,
                             uint32_t *__restrict__ keys, int32_t *__restrict__ ids) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const int32_t li = chunk_long[c];
    const int64_t beg = rowptr[long_rows[li]] + (c - long_chunk_ptr[li]) * long_chunk;
    keys[c] = (uint32_t)colidx[beg];
    ids[c] = (int32_t)c;
}

__global__ void k_make_tkeys(const int32_t *__restrict__ rowidx, const int32_t *__restrict__ colidx, int64_t nnz,
                             int64_t n_rows, uint64_t *__restrict__ keys, int32_t *__restrict__ payload) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nnz) return;
    keys[k] = (uint64_t)colidx[k] * (uint64_t)n_rows + (uint64_t)rowidx[k];
    payload[k] = (int32_t)k;
}

// relabelling (vertex row_order[i] -> i): newid = inverse of row_order; key of an entry = newid[row] * n + newid[col]
__global__ void k_invert_order(const int32_t *__restrict__ order, int64_t n, int32_t *__restrict__ newid) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) newid[order[i]] = (int32_t)i;
}

__global__ void k_make_rkeys(const int32_t *__restrict__ rowidx, const int32_t *__restrict__ colidx, const int32_t *__restrict__ newid,
                             int64_t nnz, int64_t n, uint64_t *__restrict__ keys, int32_t *__restrict__ payload) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nnz) return;
    keys[k] = (uint64_t)newid[rowidx[k]] * (uint64_t)n + (uint64_t)newid[colidx[k]];
    payload[k] = (int32_t)k;
}

// relabelling order inside a degree bin: key[v] = smallest degree rank among v's neighbours (its most popular neighbour)
__global__ void k_min_neighbour_rank(const int32_t *__restrict__ rowidx, const int32_t *__restrict__ colidx, const int32_t *__restrict__ rank,
                                     int64_t nnz, int32_t *__restrict__ key) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nnz) atomicMin(&key[rowidx[k]], rank[colidx[k]]);
}

__global__ void k_relabel_keys(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ key, int64_t n, int clamp,
                               uint64_t *__restrict__ keys, int32_t *__restrict__ ids) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int64_t d = rowptr[r + 1] - rowptr[r];
    const uint64_t bin = (uint64_t)(clamp - (d < clamp ? d : clamp));          // k_order_keys' bins: heaviest first
    keys[r] = (bin << 32) | (uint64_t)(uint32_t)key[r];
    ids[r] = (int32_t)r;
}

__global__ void k_permute_vals(const float *__restrict__ vals, const int32_t *__restrict__ perm, int64_t n,
                               float *__restrict__ out) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = vals[perm[k]];
}

__global__ void k_split_tkeys(const uint64_t *__restrict__ keys, int64_t nnz, int64_t n_rows,
                              int32_t *__restrict__ t_row /* = column of A */, int32_t *__restrict__ t_col /* = row of A */) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nnz) return;
    t_row[k] = (int32_t)(keys[k] / (uint64_t)n_rows);
    t_col[k] = (int32_t)(keys[k] % (uint64_t)n_rows);
}

// ---- helpers -------------------------------------------------------------------------------
void free_csr(Csr &m) {
    if (m.rowptr) (void)hipFree(m.rowptr);
    if (m.colidx) (void)hipFree(m.colidx);
    if (m.long_rows) (void)hipFree(m.long_rows);
    if (m.long_chunk_ptr) (void)hipFree(m.long_chunk_ptr);
    if (m.chunk_long) (void)hipFree(m.chunk_long);
    if (m.chunk_order) (void)hipFree(m.chunk_order);
    if (m.row_order) (void)hipFree(m.row_order);
    if (m.nonempty_rows) (void)hipFree(m.nonempty_rows);
    if (m.slot_beg) (void)hipFree(m.slot_beg);
    if (m.slot_cnt) (void)hipFree(m.slot_cnt);
    m = Csr();
}

// Everything build_long_plan makes (the structure itself stays): for a rebuild under another row order
void free_plan(Csr &m) {
    if (m.long_rows) (void)hipFree(m.long_rows);
    if (m.long_chunk_ptr) (void)hipFree(m.long_chunk_ptr);
    if (m.chunk_long) (void)hipFree(m.chunk_long);
    if (m.chunk_order) (void)hipFree(m.chunk_order);
    if (m.row_order) (void)hipFree(m.row_order);
    if (m.nonempty_rows) (void)hipFree(m.nonempty_rows);
    if (m.slot_beg) (void)hipFree(m.slot_beg);
    if (m.slot_cnt) (void)hipFree(m.slot_cnt);
    m.long_rows = nullptr; m.long_chunk_ptr = nullptr; m.chunk_long = nullptr; m.chunk_order = nullptr; m.row_order = nullptr;
    m.nonempty_rows = nullptr; m.slot_beg = nullptr; m.slot_cnt = nullptr;
    m.n_long = 0; m.n_chunks = 0; m.n_nonempty = 0;
}

int build_long_plan(Csr &m, hipStream_t s) {
    m.n_long = 0; m.n_chunks = 0;
    const bool small = m.n_rows >= TINY_ROWS && m.n_rows < SMALL_ROWS;       // see gnx_internal.h
    m.long_row = small ? SMALL_LONG_ROW : LONG_ROW;
    m.long_chunk = small ? SMALL_LONG_ROW : LONG_CHUNK;
    const int order_clamp = m.long_row < 65535 ? m.long_row : 65535;
    if (m.n_rows == 0) return GNX_OK;
    if (m.order_window > 0) {   // degree bins inside windows of the caller's numbering
        DevBuf k0, k1, ids, t;
        const unsigned bin_bits = bits_for((uint64_t)order_clamp + 1);
        const uint64_t n_windows = (uint64_t)((m.n_rows + m.order_window - 1) / m.order_window);
        const unsigned key_bits = bin_bits + bits_for(n_windows + 1);
        GNX_CHECK_ARG(key_bits <= 32, "row window: %lld windows of %lld rows do not fit the order key", (long long)n_windows,
                      (long long)m.order_window);
        GNX_HIP(k0.alloc(m.n_rows * 4)); GNX_HIP(k1.alloc(m.n_rows * 4)); GNX_HIP(ids.alloc(m.n_rows * sizeof(int32_t)));
        GNX_HIP(hipMalloc((void **)&m.row_order, m.n_rows * sizeof(int32_t)));
        hipLaunchKernelGGL(k_order_keys_windowed, dim3(blocks_for(m.n_rows)), dim3(256), 0, s, m.rowptr, m.n_rows, order_clamp, m.order_window,
                           (uint32_t)n_windows, bin_bits, k0.as<uint32_t>(), ids.as<int32_t>());
        size_t tb = 0;
        GNX_HIP(rocprim::radix_sort_pairs(nullptr, tb, k0.as<uint32_t>(), k1.as<uint32_t>(), ids.as<int32_t>(), m.row_order,
                                          (size_t)m.n_rows, 0u, key_bits, s));
        GNX_HIP(t.alloc(tb));
        GNX_HIP(rocprim::radix_sort_pairs(t.p, tb, k0.as<uint32_t>(), k1.as<uint32_t>(), ids.as<int32_t>(), m.row_order,
                                          (size_t)m.n_rows, 0u, key_bits, s));
        GNX_HIP(hipStreamSynchronize(s));
    } else {   // degree-binned row order
        DevBuf k0, k1, ids, t;
        GNX_HIP(k0.alloc(m.n_rows * 2)); GNX_HIP(k1.alloc(m.n_rows * 2)); GNX_HIP(ids.alloc(m.n_rows * sizeof(int32_t)));
        GNX_HIP(hipMalloc((void **)&m.row_order, m.n_rows * sizeof(int32_t)));
        hipLaunchKernelGGL(k_order_keys, dim3(blocks_for(m.n_rows)), dim3(256), 0, s, m.rowptr, m.n_rows, order_clamp, k0.as<uint16_t>(),
                           ids.as<int32_t>());
        const unsigned key_bits = bits_for((uint64_t)order_clamp + 1);
        size_t tb = 0;
        GNX_HIP(rocprim::radix_sort_pairs(nullptr, tb, k0.as<uint16_t>(), k1.as<uint16_t>(), ids.as<int32_t>(), m.row_order,
                                          (size_t)m.n_rows, 0u, key_bits, s));
        GNX_HIP(t.alloc(tb));
        GNX_HIP(rocprim::radix_sort_pairs(t.p, tb, k0.as<uint16_t>(), k1.as<uint16_t>(), ids.as<int32_t>(), m.row_order,
                                          (size_t)m.n_rows, 0u, key_bits, s));
        GNX_HIP(hipStreamSynchronize(s));
    }
    if (m.n_rows >= SMALL_ROWS) {   // big structures: the rows' entry ranges in slot order
        GNX_HIP(hipMalloc((void **)&m.slot_beg, m.n_rows * sizeof(int64_t)));
        GNX_HIP(hipMalloc((void **)&m.slot_cnt, m.n_rows * sizeof(int32_t)));
        hipLaunchKernelGGL(k_slot_ptrs, dim3(blocks_for(m.n_rows)), dim3(256), 0, s, m.rowptr, m.row_order, m.n_rows, m.slot_beg, m.slot_cnt);
    }
    {   // rows with entries: they lead the order (heaviest first), the empty ones trail it
        DevBuf count;
        GNX_HIP(count.alloc(sizeof(unsigned long long)));
        GNX_HIP(hipMemsetAsync(count.p, 0, sizeof(unsigned long long), s));
        hipLaunchKernelGGL(k_count_nonempty, dim3(blocks_for(m.n_rows)), dim3(256), 0, s, m.rowptr, m.n_rows, count.as<unsigned long long>());
        unsigned long long host = 0;
        GNX_HIP(hipMemcpyAsync(&host, count.p, sizeof(host), hipMemcpyDeviceToHost, s));
        GNX_HIP(hipStreamSynchronize(s));
        m.n_nonempty = (int64_t)host;
    }
    m.empty_rows_unreferenced = false;
    if (m.n_nonempty < m.n_rows) {
        if (m.n_nonempty > 0) {   // the rows with entries in ascending order: the leading slots of row_order, sorted by id
            DevBuf t;
            GNX_HIP(hipMalloc((void **)&m.nonempty_rows, m.n_nonempty * sizeof(int32_t)));
            size_t tb = 0;
            GNX_HIP(rocprim::radix_sort_keys(nullptr, tb, m.row_order, m.nonempty_rows, (size_t)m.n_nonempty, 0u, bits_for((uint64_t)m.n_rows), s));
            GNX_HIP(t.alloc(tb));
            GNX_HIP(rocprim::radix_sort_keys(t.p, tb, m.row_order, m.nonempty_rows, (size_t)m.n_nonempty, 0u, bits_for((uint64_t)m.n_rows), s));
            GNX_HIP(hipStreamSynchronize(s));
        }
        if (m.n_rows == m.n_cols) {   // does any entry point at a row that has no entries itself?
            DevBuf flag, count;
            GNX_HIP(flag.alloc(m.n_rows)); GNX_HIP(count.alloc(sizeof(int)));
            GNX_HIP(hipMemsetAsync(flag.p, 0, m.n_rows, s));
            GNX_HIP(hipMemsetAsync(count.p, 0, sizeof(int), s));
            if (m.nnz > 0) hipLaunchKernelGGL(k_mark_referenced, dim3(blocks_for(m.nnz)), dim3(256), 0, s, m.colidx, m.nnz, flag.as<uint8_t>());
            hipLaunchKernelGGL(k_count_empty_referenced, dim3(blocks_for(m.n_rows)), dim3(256), 0, s, m.rowptr, m.n_rows, flag.as<uint8_t>(), count.as<int>());
            int host_count = 1;
            GNX_HIP(hipMemcpyAsync(&host_count, count.p, sizeof(int), hipMemcpyDeviceToHost, s));
            GNX_HIP(hipStreamSynchronize(s));
            m.empty_rows_unreferenced = host_count == 0;
        }
    }
    if (m.nnz == 0) return GNX_OK;
    DevBuf flag, cnt, pos, cpos, tmp;
    GNX_HIP(flag.alloc(m.n_rows * sizeof(int32_t)));
    GNX_HIP(cnt.alloc(m.n_rows * sizeof(int64_t)));
    GNX_HIP(pos.alloc(m.n_rows * sizeof(int32_t)));
    GNX_HIP(cpos.alloc(m.n_rows * sizeof(int64_t)));
    hipLaunchKernelGGL(k_flag_long, dim3(blocks_for(m.n_rows)), dim3(256), 0, s, m.rowptr, m.n_rows, m.long_row, m.long_chunk,
                       flag.as<int32_t>(), cnt.as<int64_t>());
    size_t t1 = 0, t2 = 0;
    GNX_HIP(rocprim::exclusive_scan(nullptr, t1, flag.as<int32_t>(), pos.as<int32_t>(), (int32_t)0, (size_t)m.n_rows,
                                    rocprim::plus<int32_t>(), s));
    GNX_HIP(rocprim::exclusive_scan(nullptr, t2, cnt.as<int64_t>(), cpos.as<int64_t>(), (int64_t)0, (size_t)m.n_rows,
                                    rocprim::plus<int64_t>(), s));
    GNX_HIP(tmp.alloc(t1 > t2 ? t1 : t2));
    GNX_HIP(rocprim::exclusive_scan(tmp.p, t1, flag.as<int32_t>(), pos.as<int32_t>(), (int32_t)0, (size_t)m.n_rows,
                                    rocprim::plus<int32_t>(), s));
    GNX_HIP(rocprim::exclusive_scan(tmp.p, t2, cnt.as<int64_t>(), cpos.as<int64_t>(), (int64_t)0, (size_t)m.n_rows,
                                    rocprim::plus<int64_t>(), s));
    int32_t last_flag = 0, last_pos = 0;
    int64_t last_cnt = 0, last_cpos = 0;
    GNX_HIP(hipMemcpyAsync(&last_flag, flag.as<int32_t>() + m.n_rows - 1, 4, hipMemcpyDeviceToHost, s));
    GNX_HIP(hipMemcpyAsync(&last_pos, pos.as<int32_t>() + m.n_rows - 1, 4, hipMemcpyDeviceToHost, s));
    GNX_HIP(hipMemcpyAsync(&last_cnt, cnt.as<int64_t>() + m.n_rows - 1, 8, hipMemcpyDeviceToHost, s));
    GNX_HIP(hipMemcpyAsync(&last_cpos, cpos.as<int64_t>() + m.n_rows - 1, 8, hipMemcpyDeviceToHost, s));
    GNX_HIP(hipStreamSynchronize(s));
    m.n_long = (int64_t)last_flag + last_pos;
    m.n_chunks = last_cnt + last_cpos;
    if (m.n_long == 0) return GNX_OK;
    GNX_HIP(hipMalloc((void **)&m.long_rows, m.n_long * sizeof(int32_t)));
    GNX_HIP(hipMalloc((void **)&m.long_chunk_ptr, (m.n_long + 1) * sizeof(int64_t)));
    GNX_HIP(hipMalloc((void **)&m.chunk_long, m.n_chunks * sizeof(int32_t)));
    hipLaunchKernelGGL(k_fill_long, dim3(blocks_for(m.n_rows)), dim3(256), 0, s, m.rowptr, m.n_rows, m.long_row, m.long_chunk,
                       pos.as<int32_t>(), cpos.as<int64_t>(), m.long_rows, m.long_chunk_ptr, m.chunk_long);
    GNX_HIP(hipMemcpyAsync(m.long_chunk_ptr + m.n_long, &m.n_chunks, 8, hipMemcpyHostToDevice, s));
    {   // column-window order of the chunks
        DevBuf k0, k1, ids, t;
        GNX_HIP(k0.alloc(m.n_chunks * 4)); GNX_HIP(k1.alloc(m.n_chunks * 4)); GNX_HIP(ids.alloc(m.n_chunks * 4));
        GNX_HIP(hipMalloc((void **)&m.chunk_order, m.n_chunks * sizeof(int32_t)));
        hipLaunchKernelGGL(k_chunk_keys, dim3(blocks_for(m.n_chunks)), dim3(256), 0, s, m.rowptr, m.colidx, m.long_rows,
                           m.long_chunk_ptr, m.chunk_long, m.n_chunks, m.long_chunk, k0.as<uint32_t>(), ids.as<int32_t>());
        size_t tb = 0;
        GNX_HIP(rocprim::radix_sort_pairs(nullptr, tb, k0.as<uint32_t>(), k1.as<uint32_t>(), ids.as<int32_t>(), m.chunk_order,
                                          (size_t)m.n_chunks, 0u, 32u, s));
        GNX_HIP(t.alloc(tb));
        GNX_HIP(rocprim::radix_sort_pairs(t.p, tb, k0.as<uint32_t>(), k1.as<uint32_t>(), ids.as<int32_t>(), m.chunk_order,
                                          (size_t)m.n_chunks, 0u, 32u, s));
        GNX_HIP(hipStreamSynchronize(s));
    }
    GNX_HIP(hipStreamSynchronize(s));
    return GNX_OK;
}

// A stream that is being captured into a hipGraph must not see hipMalloc / hipFree / synchronisation: the lazily built parts of
// a handle (long-row slab, transposed structure, relabelled copy) refuse to grow there and say how to prepare them.
bool stream_is_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st != hipStreamCaptureStatusNone;
}

#define GNX_NOT_WHILE_CAPTURING(s, what)                                                                                         \
    do {                                                                                                                         \
        if (stream_is_capturing(s)) {                                                                                            \
            set_error("%s would have to be allocated while the stream is being captured into a hipGraph: call "                  \
                      "gnx_graph_reserve(handle, C, flags) -- or run the same call once eagerly -- before the capture begins", what); \
            return GNX_ERR_UNSUPPORTED;                                                                                          \
        }                                                                                                                        \
    } while (0)

// The slab the long-row chunks of a launch write their partial sums to ([chunks x C] floats).  It only ever grows; growing frees
// and allocates (an implicit device synchronisation), which is why it never happens under capture (gnx_graph_reserve sizes it
// ahead of time for the widest C a client will use).
int ensure_partial(gnx_graph *g, size_t bytes, hipStream_t s) {
    if (bytes <= g->partial_bytes) return GNX_OK;
    GNX_NOT_WHILE_CAPTURING(s, "the long-row slab of this handle");
    if (g->partial) (void)hipFree(g->partial);
    g->partial = nullptr; g->partial_bytes = 0;
    GNX_HIP(hipMalloc((void **)&g->partial, bytes));
    g->partial_bytes = bytes;
    return GNX_OK;
}

static void drop_transpose(gnx_graph *g) {
    free_csr(g->t);
    if (g->t_perm) (void)hipFree(g->t_perm);
    if (g->t_vals) (void)hipFree(g->t_vals);
    if (g->t_raw) (void)hipFree(g->t_raw);
    if (g->t_rowidx) (void)hipFree(g->t_rowidx);
    if (g->t_mask) (void)hipFree(g->t_mask);
    g->t_perm = nullptr; g->t_vals = nullptr; g->t_raw = nullptr; g->t_rowidx = nullptr; g->t_mask = nullptr;
    g->has_t = false;
}

int ensure_transpose(gnx_graph *g, hipStream_t s) {
    if (g->has_t) return GNX_OK;
    GNX_NOT_WHILE_CAPTURING(s, "the transposed structure of this handle");
    struct Undo { gnx_graph *g; ~Undo() { if (g && !g->has_t) drop_transpose(g); } } undo{g};   // no half-built state on failure
    const Csr &a = g->a;
    Csr &t = g->t;
    t.n_rows = a.n_cols; t.n_cols = a.n_rows; t.nnz = a.nnz;
    t.order_window = a.n_rows == a.n_cols ? a.order_window : 0;     // a square graph's columns share the rows' numbering
    GNX_HIP(hipMalloc((void **)&t.rowptr, (t.n_rows + 1) * sizeof(int64_t)));
    GNX_HIP(hipMalloc((void **)&t.colidx, (t.nnz ? t.nnz : 1) * sizeof(int32_t)));
    GNX_HIP(hipMalloc((void **)&g->t_perm, (t.nnz ? t.nnz : 1) * sizeof(int32_t)));
    GNX_HIP(hipMalloc((void **)&g->t_vals, (t.nnz ? t.nnz : 1) * sizeof(float)));
    GNX_HIP(hipMalloc((void **)&g->t_raw, (t.nnz ? t.nnz : 1) * sizeof(float)));
    if (t.nnz == 0) {
        GNX_HIP(hipMemsetAsync(t.rowptr, 0, (t.n_rows + 1) * sizeof(int64_t), s));
        GNX_HIP(hipStreamSynchronize(s));
        int rc0 = build_long_plan(t, s);
        if (rc0 != GNX_OK) return rc0;
        g->has_t = true;
        return GNX_OK;
    }
    DevBuf k0, k1, p0, tmp;
    GNX_HIP(k0.alloc(t.nnz * 8)); GNX_HIP(k1.alloc(t.nnz * 8));
    GNX_HIP(p0.alloc(t.nnz * 4));
    GNX_HIP(hipMalloc((void **)&g->t_rowidx, t.nnz * sizeof(int32_t)));
    hipLaunchKernelGGL(k_make_tkeys, dim3(blocks_for(t.nnz)), dim3(256), 0, s, g->rowidx, a.colidx, a.nnz, a.n_rows,
                       k0.as<uint64_t>(), p0.as<int32_t>());
    const unsigned end_bit = bits_for((uint64_t)a.n_rows * (uint64_t)a.n_cols);
    size_t tb = 0;
    GNX_HIP(rocprim::radix_sort_pairs(nullptr, tb, k0.as<uint64_t>(), k1.as<uint64_t>(), p0.as<int32_t>(), g->t_perm,
                                      (size_t)t.nnz, 0u, end_bit, s));
    GNX_HIP(tmp.alloc(tb));
    GNX_HIP(rocprim::radix_sort_pairs(tmp.p, tb, k0.as<uint64_t>(), k1.as<uint64_t>(), p0.as<int32_t>(), g->t_perm,
                                      (size_t)t.nnz, 0u, end_bit, s));
    hipLaunchKernelGGL(k_split_tkeys, dim3(blocks_for(t.nnz)), dim3(256), 0, s, k1.as<uint64_t>(), t.nnz, a.n_rows,
                       g->t_rowidx, t.colidx);
    hipLaunchKernelGGL(k_lower_bound_rows, dim3(blocks_for(t.n_rows + 1)), dim3(256), 0, s, g->t_rowidx, t.nnz,
                       t.n_rows, t.rowptr);
    hipLaunchKernelGGL(k_permute_vals, dim3(blocks_for(t.nnz)), dim3(256), 0, s, g->raw_vals, g->t_perm, t.nnz, g->t_raw);
    GNX_HIP(hipStreamSynchronize(s));
    int rc = build_long_plan(t, s);
    if (rc != GNX_OK) return rc;
    g->has_t = true;
    return GNX_OK;
}

static void drop_relabel(gnx_graph *g) {
    free_csr(g->r);
    if (g->r_perm) (void)hipFree(g->r_perm);
    if (g->r_vals) (void)hipFree(g->r_vals);
    if (g->r_order) (void)hipFree(g->r_order);
    g->r_perm = nullptr; g->r_vals = nullptr; g->r_order = nullptr;
    g->has_r = false;
}

int ensure_relabel_features(gnx_graph *g, size_t bytes, hipStream_t s) {
    if (bytes <= g->r_feat_bytes) return GNX_OK;
    GNX_NOT_WHILE_CAPTURING(s, "the feature scratch of this handle's relabelled copy");
    if (g->r_feat) (void)hipFree(g->r_feat);
    g->r_feat = nullptr; g->r_feat_bytes = 0;
    GNX_HIP(hipMalloc((void **)&g->r_feat, bytes));
    g->r_feat_bytes = bytes;
    return GNX_OK;
}

// The matrix with its vertices renumbered in the degree-binned order (heaviest first; inside a bin by the degree rank of a vertex's
// most popular neighbour -- profiles/NOTES.md round 4: -4 % time, -16 % of the long-row kernel's fabric bytes at C = 8): same
// entries, rows and columns permuted alike, columns ascending inside a row.  Built once, on the first narrow-width propagation of
// a large square graph.
int ensure_relabel(gnx_graph *g, hipStream_t s) {
    if (g->has_r) return GNX_OK;
    GNX_NOT_WHILE_CAPTURING(s, "the relabelled copy of this handle");
    struct Undo { gnx_graph *g; ~Undo() { if (g && !g->has_r) drop_relabel(g); } } undo{g};
    const Csr &a = g->a;
    Csr &r = g->r;
    GNX_CHECK_ARG(a.n_rows == a.n_cols && a.row_order != nullptr && a.nnz > 0, "relabelling needs a non-empty square graph");
    const int64_t n = a.n_rows, nnz = a.nnz;
    r.n_rows = n; r.n_cols = n; r.nnz = nnz;
    GNX_HIP(hipMalloc((void **)&r.rowptr, (n + 1) * sizeof(int64_t)));
    GNX_HIP(hipMalloc((void **)&r.colidx, nnz * sizeof(int32_t)));
    GNX_HIP(hipMalloc((void **)&g->r_perm, nnz * sizeof(int32_t)));
    GNX_HIP(hipMalloc((void **)&g->r_vals, nnz * sizeof(float)));
    GNX_HIP(hipMalloc((void **)&g->r_order, n * sizeof(int32_t)));
    DevBuf newid, k0, k1, p0, rrow, tmp;
    GNX_HIP(newid.alloc(n * 4)); GNX_HIP(k0.alloc(nnz * 8)); GNX_HIP(k1.alloc(nnz * 8)); GNX_HIP(p0.alloc(nnz * 4)); GNX_HIP(rrow.alloc(nnz * 4));
    {   // the order: degree bins (heaviest first); inside a bin by the degree rank of the most popular neighbour, then by id
        DevBuf key, ids, otmp;
        GNX_HIP(key.alloc(n * 4)); GNX_HIP(ids.alloc(n * 4));
        hipLaunchKernelGGL(k_invert_order, dim3(blocks_for(n)), dim3(256), 0, s, a.row_order, n, newid.as<int32_t>());   // degree rank
        GNX_HIP(hipMemsetAsync(key.p, 0x7f, n * 4, s));                            // 0x7f7f7f7f: "no neighbour" sorts last
        hipLaunchKernelGGL(k_min_neighbour_rank, dim3(blocks_for(nnz)), dim3(256), 0, s, g->rowidx, a.colidx, newid.as<int32_t>(), nnz,
                           key.as<int32_t>());
        const int clamp = a.long_row < 65535 ? a.long_row : 65535;                 // as build_long_plan binned the rows
        hipLaunchKernelGGL(k_relabel_keys, dim3(blocks_for(n)), dim3(256), 0, s, a.rowptr, key.as<int32_t>(), n, clamp,
                           k0.as<uint64_t>(), ids.as<int32_t>());
        const unsigned order_bits = 32u + bits_for((uint64_t)clamp + 1);
        size_t ob = 0;
        GNX_HIP(rocprim::radix_sort_pairs(nullptr, ob, k0.as<uint64_t>(), k1.as<uint64_t>(), ids.as<int32_t>(), g->r_order, (size_t)n, 0u,
                                          order_bits, s));
        GNX_HIP(otmp.alloc(ob));
        GNX_HIP(rocprim::radix_sort_pairs(otmp.p, ob, k0.as<uint64_t>(), k1.as<uint64_t>(), ids.as<int32_t>(), g->r_order, (size_t)n, 0u,
                                          order_bits, s));
        GNX_HIP(hipStreamSynchronize(s));
    }
    hipLaunchKernelGGL(k_invert_order, dim3(blocks_for(n)), dim3(256), 0, s, g->r_order, n, newid.as<int32_t>());
    hipLaunchKernelGGL(k_make_rkeys, dim3(blocks_for(nnz)), dim3(256), 0, s, g->rowidx, a.colidx, newid.as<int32_t>(), nnz, n,
                       k0.as<uint64_t>(), p0.as<int32_t>());
    const unsigned end_bit = bits_for((uint64_t)n * (uint64_t)n);
    size_t tb = 0;
    GNX_HIP(rocprim::radix_sort_pairs(nullptr, tb, k0.as<uint64_t>(), k1.as<uint64_t>(), p0.as<int32_t>(), g->r_perm, (size_t)nnz, 0u,
                                      end_bit, s));
    GNX_HIP(tmp.alloc(tb));
    GNX_HIP(rocprim::radix_sort_pairs(tmp.p, tb, k0.as<uint64_t>(), k1.as<uint64_t>(), p0.as<int32_t>(), g->r_perm, (size_t)nnz, 0u,
                                      end_bit, s));
    hipLaunchKernelGGL(k_split_tkeys, dim3(blocks_for(nnz)), dim3(256), 0, s, k1.as<uint64_t>(), nnz, n, rrow.as<int32_t>(), r.colidx);
    hipLaunchKernelGGL(k_lower_bound_rows, dim3(blocks_for(n + 1)), dim3(256), 0, s, rrow.as<int32_t>(), nnz, n, r.rowptr);
    GNX_HIP(hipStreamSynchronize(s));
    int rc = build_long_plan(r, s);
    if (rc != GNX_OK) return rc;
    g->has_r = true;
    return GNX_OK;
}

static int finish_graph(gnx_graph *g, hipStream_t s) {
    int rc = build_long_plan(g->a, s);
    if (rc != GNX_OK) return rc;
    GNX_HIP(hipGetLastError());
    return GNX_OK;
}

}  // namespace gnx

using namespace gnx;

extern "C" {

const char *gnx_last_error(void) { return g_err; }
int gnx_version(void) { return GNX_VERSION_NUM; }

int gnx_graph_destroy(gnx_graph_t g) {
    if (!g) return GNX_OK;
    free_csr(g->a);
    free_csr(g->t);
    if (g->raw_vals) (void)hipFree(g->raw_vals);
    if (g->rowidx) (void)hipFree(g->rowidx);
    if (g->e_vals) (void)hipFree(g->e_vals);
    if (g->slot_ptr) (void)hipFree(g->slot_ptr);
    if (g->t_perm) (void)hipFree(g->t_perm);
    if (g->t_vals) (void)hipFree(g->t_vals);
    if (g->t_raw) (void)hipFree(g->t_raw);
    if (g->t_rowidx) (void)hipFree(g->t_rowidx);
    if (g->t_mask) (void)hipFree(g->t_mask);
    if (g->partial) (void)hipFree(g->partial);
    if (g->deg) (void)hipFree(g->deg);
    if (g->blk_col_gid) (void)hipFree(g->blk_col_gid);
    free_csr(g->r);
    if (g->r_perm) (void)hipFree(g->r_perm);
    if (g->r_vals) (void)hipFree(g->r_vals);
    if (g->r_order) (void)hipFree(g->r_order);
    if (g->r_feat) (void)hipFree(g->r_feat);
    delete g;
    return GNX_OK;
}

int gnx_graph_reserve(gnx_graph_t g, int64_t C, int flags, void *stream) {
    GNX_CHECK_ARG(g != nullptr, "gnx_graph_reserve: NULL handle");
    GNX_CHECK_ARG(C >= 1 && (flags & ~(GNX_RESERVE_TRANSPOSED | GNX_RESERVE_K_LOOP)) == 0, "gnx_graph_reserve: bad width / flags");
    hipStream_t s = (hipStream_t)stream;
    GNX_CHECK_ARG(!stream_is_capturing(s), "gnx_graph_reserve: the stream is being captured -- reserve before the capture begins");
    int64_t chunks = g->a.n_chunks;
    if (flags & GNX_RESERVE_TRANSPOSED) {
        int rc = ensure_transpose(g, s);
        if (rc != GNX_OK) return rc;
        chunks = std::max(chunks, g->t.n_chunks);
        if (!g->t_mask && !g->has_dups && g->t.nnz > 0)      // the keep-bit scratch of a training step's column sums
            GNX_HIP(hipMalloc((void **)&g->t_mask, (size_t)g->t.nnz * sizeof(uint16_t)));
    }
    if ((flags & GNX_RESERVE_K_LOOP) && g->a.order_window == 0 && C <= RELABEL_MAX_C && g->a.n_rows == g->a.n_cols &&
        g->a.n_rows >= (1 << 20) && g->a.nnz >= g->a.n_rows) {
        int rc = ensure_relabel(g, s);
        if (rc != GNX_OK) return rc;
        rc = ensure_relabel_features(g, (size_t)g->a.n_rows * (size_t)C * sizeof(float), s);
        if (rc != GNX_OK) return rc;
        chunks = std::max(chunks, g->r.n_chunks);
    }
    if (chunks > 0) return ensure_partial(g, (size_t)chunks * (size_t)C * sizeof(float), s);
    return GNX_OK;
}

int gnx_graph_set_row_window(gnx_graph_t g, int64_t window_rows, void *stream) {
    GNX_CHECK_ARG(g != nullptr, "gnx_graph_set_row_window: NULL handle");
    GNX_CHECK_ARG(window_rows >= 0, "gnx_graph_set_row_window: negative window");
    hipStream_t s = (hipStream_t)stream;
    GNX_CHECK_ARG(!stream_is_capturing(s), "gnx_graph_set_row_window: the stream is being captured -- set the window before the capture begins");
    if (g->a.order_window == window_rows) return GNX_OK;
    // launches in flight on this handle -- on ANY stream -- still read the old order's arrays, which are freed below
    GNX_HIP(hipDeviceSynchronize());
    const int64_t before = g->a.order_window;
    free_plan(g->a);
    g->a.order_window = window_rows;
    int rc = build_long_plan(g->a, s);
    if (rc != GNX_OK) {                                              // (a window count that does not fit the key: back to what it was)
        free_plan(g->a);
        g->a.order_window = before;
        const int rc2 = build_long_plan(g->a, s);
        return rc2 != GNX_OK ? rc2 : rc;
    }
    if (g->has_t) {
        free_plan(g->t);
        g->t.order_window = g->a.n_rows == g->a.n_cols ? window_rows : 0;
        rc = build_long_plan(g->t, s);
        if (rc != GNX_OK) {       // a half-built transposed plan must never be launched: the structure goes, the next user rebuilds it
            drop_transpose(g);
            return rc;
        }
    }
    if (g->has_r) drop_relabel(g);                                   // the degree-relabelled copy belongs to the default order
    GNX_HIP(hipGetLastError());
    return GNX_OK;
}

int gnx_graph_create_coo(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t *d_indices, const float *d_values,
                         void *stream, gnx_graph_t *out) {
    GNX_CHECK_ARG(out != nullptr, "gnx_graph_create_coo: out is NULL");
    *out = nullptr;
    GNX_CHECK_ARG(n_rows >= 0 && n_cols >= 0 && nnz >= 0, "gnx_graph_create_coo: negative size");
    GNX_CHECK_ARG(n_rows < INT32_MAX && n_cols < INT32_MAX, "gnx_graph_create_coo: more than 2^31-1 rows/cols per handle");
    GNX_CHECK_ARG(nnz < INT32_MAX, "gnx_graph_create_coo: more than 2^31-1 entries per handle");
    GNX_CHECK_ARG(nnz == 0 || (d_indices && d_values), "gnx_graph_create_coo: NULL indices/values");
    hipStream_t s = (hipStream_t)stream;
    gnx_graph *g = new gnx_graph();
    struct Guard { gnx_graph *g; ~Guard() { if (g) gnx_graph_destroy(g); } } guard{g};
    Csr &a = g->a;
    a.n_rows = n_rows; a.n_cols = n_cols;
    g->nnz_entries = nnz;
    GNX_HIP(hipMalloc((void **)&a.rowptr, (n_rows + 1) * sizeof(int64_t)));
    if (nnz == 0) {
        a.nnz = 0;
        GNX_HIP(hipMemsetAsync(a.rowptr, 0, (n_rows + 1) * sizeof(int64_t), s));
        GNX_HIP(hipMalloc((void **)&a.colidx, 16));
        GNX_HIP(hipMalloc((void **)&g->raw_vals, 16));
        GNX_HIP(hipMalloc((void **)&g->rowidx, 16));
        GNX_HIP(hipStreamSynchronize(s));
        int rc0 = finish_graph(g, s);
        if (rc0 != GNX_OK) return rc0;
        guard.g = nullptr; *out = g;
        return GNX_OK;
    }
    DevBuf k0, k1, v1, head, scan, tmp, bad;
    GNX_HIP(k0.alloc(nnz * 8)); GNX_HIP(k1.alloc(nnz * 8));
    GNX_HIP(v1.alloc(nnz * 4));
    GNX_HIP(bad.alloc(4));
    GNX_HIP(hipMemsetAsync(bad.p, 0, 4, s));
    hipLaunchKernelGGL(k_make_keys, dim3(blocks_for(nnz)), dim3(256), 0, s, d_indices, nnz, n_rows, n_cols,
                       k0.as<uint64_t>(), bad.as<int>());
    const unsigned end_bit = bits_for((uint64_t)n_rows * (uint64_t)n_cols);
    size_t tb = 0;
    GNX_HIP(rocprim::radix_sort_pairs(nullptr, tb, k0.as<uint64_t>(), k1.as<uint64_t>(), d_values, v1.as<float>(),
                                      (size_t)nnz, 0u, end_bit, s));
    GNX_HIP(tmp.alloc(tb));
    GNX_HIP(rocprim::radix_sort_pairs(tmp.p, tb, k0.as<uint64_t>(), k1.as<uint64_t>(), d_values, v1.as<float>(),
                                      (size_t)nnz, 0u, end_bit, s));
    // k0 is free again: reuse as head flags + scan (2 x int32 per entry fits in 8 bytes/entry)
    int32_t *d_head = k0.as<int32_t>();
    int32_t *d_scan = d_head + nnz;
    hipLaunchKernelGGL(k_heads, dim3(blocks_for(nnz)), dim3(256), 0, s, k1.as<uint64_t>(), nnz, d_head);
    size_t sb = 0;
    GNX_HIP(rocprim::inclusive_scan(nullptr, sb, d_head, d_scan, (size_t)nnz, rocprim::plus<int32_t>(), s));
    if (sb > tb) { (void)hipFree(tmp.release()); GNX_HIP(tmp.alloc(sb)); }
    GNX_HIP(rocprim::inclusive_scan(tmp.p, sb, d_head, d_scan, (size_t)nnz, rocprim::plus<int32_t>(), s));
    int h_bad = 0; int32_t h_nslots = 0;
    GNX_HIP(hipMemcpyAsync(&h_bad, bad.p, 4, hipMemcpyDeviceToHost, s));
    GNX_HIP(hipMemcpyAsync(&h_nslots, d_scan + nnz - 1, 4, hipMemcpyDeviceToHost, s));
    GNX_HIP(hipStreamSynchronize(s));
    GNX_CHECK_ARG(h_bad == 0, "gnx_graph_create_coo: an index lies outside the %lld x %lld shape", (long long)n_rows,
                  (long long)n_cols);
    a.nnz = h_nslots;
    g->has_dups = (a.nnz != nnz);
    GNX_HIP(hipMalloc((void **)&a.colidx, a.nnz * sizeof(int32_t)));
    GNX_HIP(hipMalloc((void **)&g->rowidx, a.nnz * sizeof(int32_t)));
    if (g->has_dups) GNX_HIP(hipMalloc((void **)&g->slot_ptr, (a.nnz + 1) * sizeof(int64_t)));
    hipLaunchKernelGGL(k_fill_slots, dim3(blocks_for(nnz)), dim3(256), 0, s, k1.as<uint64_t>(), d_scan, nnz, n_cols,
                       a.colidx, g->rowidx, g->slot_ptr);
    if (g->has_dups) {
        GNX_HIP(hipMemcpyAsync(g->slot_ptr + a.nnz, &nnz, 8, hipMemcpyHostToDevice, s));
        g->e_vals = (float *)v1.release();
        GNX_HIP(hipMalloc((void **)&g->raw_vals, a.nnz * sizeof(float)));
        hipLaunchKernelGGL(k_sum_slots, dim3(blocks_for(a.nnz)), dim3(256), 0, s, g->e_vals, g->slot_ptr, a.nnz,
                           g->raw_vals);
    } else {
        g->raw_vals = (float *)v1.release();
    }
    hipLaunchKernelGGL(k_lower_bound_rows, dim3(blocks_for(n_rows + 1)), dim3(256), 0, s, g->rowidx, a.nnz, n_rows,
                       a.rowptr);
    GNX_HIP(hipStreamSynchronize(s));
    int rc = finish_graph(g, s);
    if (rc != GNX_OK) return rc;
    guard.g = nullptr; *out = g;
    return GNX_OK;
}

int gnx_graph_create_csr(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t *d_rowptr, const int32_t *d_colidx,
                         const float *d_values, void *stream, gnx_graph_t *out) {
    GNX_CHECK_ARG(out != nullptr, "gnx_graph_create_csr: out is NULL");
    *out = nullptr;
    GNX_CHECK_ARG(n_rows >= 0 && n_cols >= 0 && nnz >= 0, "gnx_graph_create_csr: negative size");
    GNX_CHECK_ARG(n_rows < INT32_MAX && n_cols < INT32_MAX && nnz < INT32_MAX, "gnx_graph_create_csr: size over 2^31-1");
    GNX_CHECK_ARG(d_rowptr && (nnz == 0 || (d_colidx && d_values)), "gnx_graph_create_csr: NULL array");
    hipStream_t s = (hipStream_t)stream;
    gnx_graph *g = new gnx_graph();
    struct Guard { gnx_graph *g; ~Guard() { if (g) gnx_graph_destroy(g); } } guard{g};
    Csr &a = g->a;
    a.n_rows = n_rows; a.n_cols = n_cols; a.nnz = nnz; g->nnz_entries = nnz;
    const size_t nz = nnz ? nnz : 4;
    GNX_HIP(hipMalloc((void **)&a.rowptr, (n_rows + 1) * sizeof(int64_t)));
    GNX_HIP(hipMalloc((void **)&a.colidx, nz * sizeof(int32_t)));
    GNX_HIP(hipMalloc((void **)&g->raw_vals, nz * sizeof(float)));
    GNX_HIP(hipMalloc((void **)&g->rowidx, nz * sizeof(int32_t)));
    GNX_HIP(hipMemcpyAsync(a.rowptr, d_rowptr, (n_rows + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
    if (nnz) {
        GNX_HIP(hipMemcpyAsync(a.colidx, d_colidx, nnz * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        GNX_HIP(hipMemcpyAsync(g->raw_vals, d_values, nnz * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    DevBuf bad;
    GNX_HIP(bad.alloc(4));
    GNX_HIP(hipMemsetAsync(bad.p, 0, 4, s));
    if (n_rows) {
        hipLaunchKernelGGL(k_check_csr, dim3(blocks_for(n_rows)), dim3(256), 0, s, a.rowptr, a.colidx, n_rows, n_cols,
                           nnz, bad.as<int>());
        hipLaunchKernelGGL(k_rows_from_ptr, dim3(blocks_for(n_rows)), dim3(256), 0, s, a.rowptr, n_rows, g->rowidx);
    }
    int h_bad = 0;
    GNX_HIP(hipMemcpyAsync(&h_bad, bad.p, 4, hipMemcpyDeviceToHost, s));
    GNX_HIP(hipStreamSynchronize(s));
    GNX_CHECK_ARG(h_bad == 0, "gnx_graph_create_csr: rowptr/colidx are not a valid sorted CSR for this shape");
    int rc = finish_graph(g, s);
    if (rc != GNX_OK) return rc;
    guard.g = nullptr; *out = g;
    return GNX_OK;
}

int gnx_graph_info(gnx_graph_t g, int64_t *n_rows, int64_t *n_cols, int64_t *nnz_entries, int64_t *nnz_coalesced) {
    GNX_CHECK_ARG(g != nullptr, "gnx_graph_info: NULL handle");
    if (n_rows) *n_rows = g->a.n_rows;
    if (n_cols) *n_cols = g->a.n_cols;
    if (nnz_entries) *nnz_entries = g->nnz_entries;
    if (nnz_coalesced) *nnz_coalesced = g->a.nnz;
    return GNX_OK;
}

int gnx_graph_csr(gnx_graph_t g, const int64_t **d_rowptr, const int32_t **d_colidx, const float **d_raw_values) {
    GNX_CHECK_ARG(g != nullptr, "gnx_graph_csr: NULL handle");
    if (d_rowptr) *d_rowptr = g->a.rowptr;
    if (d_colidx) *d_colidx = g->a.colidx;
    if (d_raw_values) *d_raw_values = g->raw_vals;
    return GNX_OK;
}

int gnx_graph_export(gnx_graph_t g, int64_t *d_rowptr_out, int32_t *d_colidx_out, float *d_raw_values_out,
                     int32_t *d_rowidx_out, void *stream) {
    GNX_CHECK_ARG(g != nullptr, "gnx_graph_export: NULL handle");
    hipStream_t s = (hipStream_t)stream;
    const Csr &a = g->a;
    if (d_rowptr_out) GNX_HIP(hipMemcpyAsync(d_rowptr_out, a.rowptr, (a.n_rows + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
    if (a.nnz > 0) {
        if (d_colidx_out) GNX_HIP(hipMemcpyAsync(d_colidx_out, a.colidx, a.nnz * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        if (d_raw_values_out) GNX_HIP(hipMemcpyAsync(d_raw_values_out, g->raw_vals, a.nnz * sizeof(float), hipMemcpyDeviceToDevice, s));
        if (d_rowidx_out) GNX_HIP(hipMemcpyAsync(d_rowidx_out, g->rowidx, a.nnz * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    }
    return GNX_OK;
}

const char *gnx_graph_last_kernel(gnx_graph_t g) { return g ? g->last_kernel : ""; }

}  // extern "C"

// The hot path on gfx950: CSR SpMM with the residual mix fused into its epilogue.
//
//   out[i,:] = act( beta * ( sum_j A[i,j] X[j,:] + diag[i] X[i,:] ) + alpha * H0[i,:] )
//
// replaces tf.sparse.sparse_dense_matmul + the three element-wise ops behind it in the
// reference (gnntf/core/gnn/architectures/filter.py:19-22; gcn.py:88).
//
// The kernel is HBM-bound (2 flop per 4 gathered bytes), so the design is about memory
// parallelism and cache behaviour, not MFMA:
//   * wide features (more than 32 lanes of VEC columns, e.g. C = 256): one 64-lane wave per row, each
//     lane owning VEC contiguous columns, so every neighbour row is ONE coalesced wave-instruction
//     (C = 256: 64 x float4 = the whole 1 KiB row).  The row's (col, val) pairs are fetched 64 at a
//     time with one coalesced load and broadcast from registers with v_readlane, and U = 8 neighbour
//     rows are kept in flight per wave before the first FMA;
//   * narrow features: G = 8..32 lanes per row, 64/G rows per wave, rows taken in a degree-binned
//     order (Csr::row_order) so that the rows sharing a wave have similar lengths; for G <= 8 the next
//     (col, val) batch is prefetched behind the gathers;
//   * power-law rows: a row with more than p.long_row entries is cut into p.long_chunk-entry chunks summed
//     by separate waves into a partial slab (wide: lanes across columns; narrow: sub-groups of lanes
//     across the chunk's entries + a fixed xor tree), then added in chunk order by a second kernel
//     (fixed order: results are bitwise reproducible, no float atomics).  Chunks are processed in
//     column-window order (Csr::chunk_order) so the hub rows they share stay in L2 / Infinity Cache.
#include "gnx_spmm_device.h"

namespace {

// ---- wide path: one wave per row -----------------------------------------------------------
// tune bits (GNX_TUNE, experiments): 1 = degree-binned row order, 2 = non-temporal H0/out, 4 = non-temporal col/val
template <int VEC, int U, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_spmm_wave(const SpmmArgs p) {
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t slot = p.slot0 + xcd_block(p) * WPB + wib;
    if (slot >= p.n_rows) return;
    const int64_t row = p.row_list ? (int64_t)__builtin_amdgcn_readfirstlane(p.row_list[slot])
                                   : ((p.tune & 1) ? (int64_t)__builtin_amdgcn_readfirstlane(p.row_order[slot]) : slot);
    const int64_t beg = p.rowptr[row], end = p.rowptr[row + 1];
    if (end - beg > p.long_row) return;  // k_spmm_long_* take it
    if (p.skip_empty && beg == end) return;   // GNX_ACT_SKIP_EMPTY: the row already holds alpha * H0 from an earlier iteration
    for (int c0 = 0; c0 < p.C; c0 += 64 * VEC) {
        const int c = c0 + lane * VEC;
        const bool active = c < p.C;
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
        wave_accumulate<VEC, U>(p.colidx, p.vals, p.X, p.ldx, beg, end, active ? c : 0, lane, acc, (p.tune & 4) != 0);
        epilogue_store<VEC>(p, row, c, active, acc, (p.tune & 2) != 0);
    }
}

// ---- narrow path: G lanes per row, 256/G rows per block ---------------------------------------
// PIPE: the (col, val) pairs of batch b+1 are fetched while the gathers of batch b are in flight.
template <int VEC, int G, int U, bool PIPE>
__device__ __forceinline__ void group_rows(const SpmmArgs &p, int64_t block) {
    constexpr int RPB = 256 / G;
    const int sub = threadIdx.x % G;
    const int64_t slot = p.slot0 + block * RPB + threadIdx.x / G;
    if (slot >= p.n_rows) return;
    const int64_t row = p.row_order ? (int64_t)p.row_order[slot] : slot;   // degree-binned: the rows of one wave have similar lengths
    int64_t beg, end;
    if (p.slot_beg) { beg = p.slot_beg[slot]; end = beg + p.slot_cnt[slot]; }      // (slot order: coalesced, independent of the row_order load)
    else { beg = p.rowptr[row]; end = p.rowptr[row + 1]; }
    if (end - beg > p.long_row) return;
    if (p.skip_empty && beg == end) return;   // GNX_ACT_SKIP_EMPTY
    for (int c0 = 0; c0 < p.C; c0 += G * VEC) {
        const int c = c0 + sub * VEC;
        const bool active = c < p.C;
        const float *__restrict__ Xc = p.X + (active ? c : 0);
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
        if (PIPE) {
            int jn[U];
            float wn[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ok = beg + u < end;
                jn[u] = ok ? p.colidx[beg + u] : -1;
                wn[u] = ok ? p.vals[beg + u] : 0.f;
            }
            for (int64_t e = beg; e < end; e += U) {
                float x[U][VEC];
                float w[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {          // gathers of this batch
                    w[u] = wn[u];
                    if (jn[u] >= 0) vload<VEC>(x[u], Xc + (int64_t)jn[u] * p.ldx);
                    else {
#pragma unroll
                        for (int v = 0; v < VEC; ++v) x[u][v] = 0.f;
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {          // indices of the next batch, behind the gathers
                    const bool ok = e + U + u < end;
                    jn[u] = ok ? p.colidx[e + U + u] : -1;
                    wn[u] = ok ? p.vals[e + U + u] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[v] = fmaf(w[u], x[u][v], acc[v]);
            }
        } else {
            for (int64_t e = beg; e < end; e += U) {   // U entries in flight per lane, ragged tail predicated
                float x[U][VEC];
                float w[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (e + u < end) {
                        const int j = p.colidx[e + u];
                        w[u] = p.vals[e + u];
                        vload<VEC>(x[u], Xc + (int64_t)j * p.ldx);
                    } else {
                        w[u] = 0.f;
#pragma unroll
                        for (int v = 0; v < VEC; ++v) x[u][v] = 0.f;
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[v] = fmaf(w[u], x[u][v], acc[v]);
            }
        }
        epilogue_store<VEC>(p, row, c, active, acc);
    }
}

// The same rows with a COOPERATIVE index fetch, for the narrow groups (G <= 8 lanes per row, C <= 32).  In group_rows every lane of a
// row's group loads the same (col, val) pair, so a step of four entries costs four index loads + four value loads + four gathers
// per lane: twelve vector-memory instructions, each served line by line by the CU's L1 pipe (sixteen different lines per wave
// instruction).  At narrow widths that pipe is what a launch waits for next to the fabric (SQ counters at C = 8: 60 % of the
// wave cycles are issue stalls, the TCP is busy for the whole launch; with every gather made to hit, a launch still takes 57 %
// of its time -- profiles/notes/r03_narrow_*).  Here lane `sub` of the group loads the pair of entry base + sub -- ONE index
// load and one value load per four entries -- and the group reads them out of each other's registers (ds_bpermute, off the memory
// pipe); the next batch's pairs are fetched behind the gathers.  Entries are added in ascending order as before: same bits.
// Measured (RMAT 10M / 100M, K = 10): C = 8 17.2 -> 16.4 ms, C = 16 19.5 -> 19.0, C = 32 22.2 -> 21.9; the all-gathers-hit floor
// 9.8 -> 7.6 ms at C = 8.  The wider groups LOSE 2-4 % with it (their gathers dominate the pipe, the shuffles only add latency).
template <int VEC, int G, int B>
__device__ __forceinline__ void group_rows_coop(const SpmmArgs &p, int64_t block) {
    constexpr int RPB = 256 / G;
    const int sub = threadIdx.x % G;
    const int64_t slot = p.slot0 + block * RPB + threadIdx.x / G;
    if (slot >= p.n_rows) return;
    const int64_t row = p.row_order ? (int64_t)p.row_order[slot] : slot;
    int64_t beg, end;
    if (p.slot_beg) { beg = p.slot_beg[slot]; end = beg + p.slot_cnt[slot]; }
    else { beg = p.rowptr[row]; end = p.rowptr[row + 1]; }
    if (end - beg > p.long_row) return;
    if (p.skip_empty && beg == end) return;   // GNX_ACT_SKIP_EMPTY
    for (int c0 = 0; c0 < p.C; c0 += G * VEC) {
        const int c = c0 + sub * VEC;
        const bool active = c < p.C;
        const float *__restrict__ Xc = p.X + (active ? c : 0);
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
        int myj = -1;
        float myw = 0.f;
        if (sub < B && beg + sub < end) { myj = p.colidx[beg + sub]; myw = p.vals[beg + sub]; }
        for (int64_t e = beg; e < end; e += B) {
            float x[B][VEC];
            float w[B];
#pragma unroll
            for (int u = 0; u < B; ++u) {                                   // gathers of this batch
                const int j = __shfl(myj, u, G);
                w[u] = __shfl(myw, u, G);
                if (j >= 0) vload<VEC>(x[u], Xc + (int64_t)j * p.ldx);
                else {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) x[u][v] = 0.f;
                }
            }
            myj = -1; myw = 0.f;                                            // pairs of the next batch, behind the gathers
            if (sub < B && e + B + sub < end) { myj = p.colidx[e + B + sub]; myw = p.vals[e + B + sub]; }
#pragma unroll
            for (int u = 0; u < B; ++u)
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[v] = fmaf(w[u], x[u][v], acc[v]);
        }
        epilogue_store<VEC>(p, row, c, active, acc);
    }
}

template <int VEC, int G, int U, bool PIPE>
__global__ __launch_bounds__(256) void k_spmm_group(const SpmmArgs p) {
    if (G <= 8) group_rows_coop<VEC, G, 4>(p, xcd_block(p));
    else group_rows<VEC, G, U, PIPE>(p, xcd_block(p));
}

// ---- long rows ---------------------------------------------------------------------------------
template <int VEC, int U>
__global__ __launch_bounds__(256) void k_spmm_long_partial(const SpmmArgs p) {
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t cslot = (int64_t)blockIdx.x * 4 + wib;
    if (cslot >= p.n_chunks) return;
    const int64_t chunk = p.chunk_order ? (int64_t)p.chunk_order[cslot] : cslot;   // column-window order
    const int32_t li = p.chunk_long[chunk];
    const int64_t row = p.long_rows[li];
    const int64_t beg = p.rowptr[row] + (chunk - p.long_chunk_ptr[li]) * p.long_chunk;
    const int64_t rend = p.rowptr[row + 1];
    const int64_t end = beg + p.long_chunk < rend ? beg + p.long_chunk : rend;
    for (int c0 = 0; c0 < p.C; c0 += 64 * VEC) {
        const int c = c0 + lane * VEC;
        const bool active = c < p.C;
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
        wave_accumulate<VEC, U>(p.colidx, p.vals, p.X, p.ldx, beg, end, active ? c : 0, lane, acc);
        if (active) vstore<VEC>(p.partial + chunk * (int64_t)p.C + c, acc);
    }
}

// Narrow features: a chunk's entries are dealt round-robin to the wave's 64/G sub-groups of G lanes
// (each sub-group gathers whole C-wide rows), then the sub-group sums are added with a fixed xor tree.
template <int VEC, int G, int U>
__device__ __forceinline__ void long_chunks_group(const SpmmArgs &p, int64_t block) {
    constexpr int NS = 64 / G;
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t cslot = block * 4 + wib;
    if (cslot >= p.n_chunks) return;
    const int64_t chunk = p.chunk_order ? (int64_t)p.chunk_order[cslot] : cslot;   // column-window order
    const int32_t li = p.chunk_long[chunk];
    const int64_t row = p.long_rows[li];
    const int64_t beg = p.rowptr[row] + (chunk - p.long_chunk_ptr[li]) * p.long_chunk;
    const int64_t rend = p.rowptr[row + 1];
    const int64_t end = beg + p.long_chunk < rend ? beg + p.long_chunk : rend;
    const int sub = lane / G;
    const int c = (lane % G) * VEC;
    const bool active = c < p.C;
    const float *__restrict__ Xc = p.X + (active ? c : 0);
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    for (int64_t e = beg + sub; e < end; e += (int64_t)NS * U) {
        float x[U][VEC];
        float w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t eu = e + (int64_t)u * NS;
            if (eu < end) {
                const int j = p.colidx[eu];
                w[u] = p.vals[eu];
                vload<VEC>(x[u], Xc + (int64_t)j * p.ldx);
            } else {
                w[u] = 0.f;
#pragma unroll
                for (int v = 0; v < VEC; ++v) x[u][v] = 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = fmaf(w[u], x[u][v], acc[v]);
    }
#pragma unroll
    for (int off = G; off < 64; off <<= 1)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += __shfl_xor(acc[v], off);
    if (sub == 0 && active) vstore<VEC>(p.partial + chunk * (int64_t)p.C + c, acc);
}

template <int VEC, int G, int U>
__global__ __launch_bounds__(256) void k_spmm_long_partial_group(const SpmmArgs p) {
    long_chunks_group<VEC, G, U>(p, blockIdx.x);
}

// Short rows and the chunks of the long rows in ONE launch, for graphs with few chunks (a citation-graph-sized matrix has a few
// hundred): a chunk is one wave walking 512 entries, so a launch of a few hundred waves is bound by the latency of that walk
// (0.11 ms at C = 128) while most of the card idles; here the chunk blocks are dealt first and the short rows fill the rest of
// the card under them.  Same per-row arithmetic as the two separate launches; k_spmm_long_reduce follows as before.
template <int VEC, int G, bool PIPE>
__global__ __launch_bounds__(256) void k_spmm_group_and_chunks(const SpmmArgs p, int chunk_blocks) {
    if ((int)blockIdx.x < chunk_blocks) long_chunks_group<VEC, G, 4>(p, blockIdx.x);
    else if (G <= 8) group_rows_coop<VEC, G, 4>(p, (int64_t)blockIdx.x - chunk_blocks);
    else group_rows<VEC, G, 4, PIPE>(p, (int64_t)blockIdx.x - chunk_blocks);
}

// ---- small helpers --------------------------------------------------------------------------------
__global__ void k_gather_vals(const float *__restrict__ vals, const int32_t *__restrict__ perm, int64_t n,
                              float *__restrict__ out) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = vals[perm[k]];
}

// out[r, :] = X[idx[r], :] for int32 row ids, any width (the relabelled K loop permutes H0 once per call)
__global__ __launch_bounds__(256) void k_gather_rows32(const float *__restrict__ X, int64_t ldx, const int32_t *__restrict__ idx, int64_t n_idx,
                                                        int C, float *__restrict__ out, int64_t ldo) {
    // grid-stride: a launch may not hold more than 2^32 work-items, and n_idx * C can (80M rows x 128 columns)
    const int64_t total = n_idx * C, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t r = e / C;
        const int c = (int)(e % C);
        out[r * ldo + c] = X[(int64_t)idx[r] * ldx + c];
    }
}

template <int VEC>
const char *launch_rows(const SpmmArgs &p0, hipStream_t s) {
    SpmmArgs p = p0;
    const int lanes = (p.C + VEC - 1) / VEC;  // lanes needed to cover one row
    // GNX_ACT_SKIP_EMPTY on the sub-wave kernels: they walk the rows through row_order, whose trailing slots are exactly the rows
    // without entries -- those slots are not launched at all (on the R-MAT workloads 60 % of the rows: no wave, no row-pointer read)
    bool trim = lanes <= 32 && p.skip_empty && p.row_order != nullptr && p.n_nonempty < p.n_rows;
#ifdef GNX_TUNING
    if (p.tune & (1 << 21)) trim = false;                       // (A/B switch of the tuning build)
#endif
    if (trim) p.n_rows = p.n_nonempty;
    // the one-wave-per-row kernels keep the rows in ascending order (their H0 / out rows stream): they walk the ascending list of
    // the rows that have entries instead
    bool list = lanes > 32 && p.skip_empty && p.nonempty_rows != nullptr && p.n_nonempty < p.n_rows && !(p.tune & 1);
#ifdef GNX_TUNING
    if (p.tune & (1 << 21)) list = false;
#endif
    if (list) { p.row_list = p.nonempty_rows; p.n_rows = p.n_nonempty; }
    if (p.n_rows == 0) return "spmm_none";
    if (lanes > 32) {
        // measured: U = 8 rows in flight is the plateau (U=4 +0.7 %, U=16 +17 %, forcing 8 waves/SIMD +14 %,
        // degree-ordered rows +9 %, non-temporal H0/out/index loads +-0 %)
        // measured (tools/tune_spmm.py): 8 waves per block are 1.6 % faster than 4 when the row is one tile wide
        // (C = 256), 0.6 % slower at two tiles (C = 512); 2 and 16 waves per block lose 3-11 %
        if (p.C <= 64 * VEC) GNX_ROW_PIECES((k_spmm_wave<VEC, 8, 8>), 8, 512);
        else                 GNX_ROW_PIECES((k_spmm_wave<VEC, 8, 4>), 4, 256);
        return "spmm_wave";
    }
    // measured (tools/tune_spmm.py, RMAT 10M/100M): prefetching the next (col, val) batch behind the
    // gathers pays for G <= 8 (C <= 32: -8..-12 %) and not for the wider groups
    const int gsel = (p.tune >> 8) & 3;          // experiments: 1 = force plain, 2 = force pipelined
#define GNX_GROUP(G, RPB_, PIPE_DEFAULT)                                                              \
    do {                                                                                              \
        const bool pipe = gsel == 2 || (gsel == 0 && PIPE_DEFAULT);                                   \
        if (pipe) GNX_ROW_PIECES((k_spmm_group<VEC, G, 4, true>), RPB_, 256);                         \
        else      GNX_ROW_PIECES((k_spmm_group<VEC, G, 4, false>), RPB_, 256);                        \
    } while (0)
    // (round 2, one-process A/B at C = 128 / 64: 8 entries in flight per lane 8.86 / 4.57 ms, pipelined 8.98 / 4.37, 2 entries 8.24 / 4.37
    //  against 8.24 / 4.36 for the shipped 4 -- the sub-wave kernels sit on the bandwidth plateau, not on latency)
    if (lanes > 16) { GNX_GROUP(32, 8, false); return "spmm_group32"; }
    if (lanes > 8)  { GNX_GROUP(16, 16, false); return "spmm_group16"; }
    // Rows of up to 4 lanes (C <= 16) run on 8-lane groups as well: the four lanes beyond the row's width only take part in the
    // cooperative index fetch (round 6, config-4 graph: C = 8 1.543 -> 1.509 ms per iteration, C = 16 1.777 -> 1.741; same bits;
    // 8 entries per batch: 1.515 / 1.755; 16-lane groups: 1.77 / 1.96).  (Round 2 tried the other direction, 2 lanes per row for
    // C <= 8: 2.40 vs 2.28 ms -- every gather is one 128-byte line whatever the width, lane use is not the limit.)
    GNX_GROUP(8, 32, true);
    return "spmm_group8";
#undef GNX_GROUP
}

// few chunks (see k_spmm_group_and_chunks): one launch for the short rows and the chunks, then the reduce

template <int VEC>
const char *launch_rows_and_chunks(const SpmmArgs &p, hipStream_t s) {
    const int lanes = (p.C + VEC - 1) / VEC;
    // (tune bit 65536: tuning builds' A/B of the merged launch on big graphs)
    if (lanes > 32 || p.n_long == 0 || (p.n_rows >= SMALL_ROWS && !(p.tune & 65536)) || ((p.tune >> 8) & 3) != 0 || (p.tune & 4096)) return nullptr;
    const unsigned cb = blocks_for(p.n_chunks, 4);
    const char *name;
#define GNX_BOTH(G, RPB_, PIPE_)                                                                                          \
    hipLaunchKernelGGL((k_spmm_group_and_chunks<VEC, G, PIPE_>), dim3(cb + blocks_for(p.n_rows, RPB_)), dim3(256), 0, s, p, (int)cb)
    if (lanes > 16)     { GNX_BOTH(32, 8, false); name = "spmm_group32+chunks"; }
    else if (lanes > 8) { GNX_BOTH(16, 16, false); name = "spmm_group16+chunks"; }
    else if (lanes > 4) { GNX_BOTH(8, 32, true); name = "spmm_group8+chunks"; }
    // (rows of up to 4 lanes keep 4-lane groups HERE: the group width is also how a chunk's entries are dealt to sub-groups, i.e. the
    //  long rows' summation order, which the training kernels of the same width reproduce bit for bit; these launches are latency-bound)
    else                { GNX_BOTH(4, 64, true); name = "spmm_group4+chunks"; }
#undef GNX_BOTH
    GNX_LAUNCH((k_spmm_long_reduce<VEC>), blocks_for(p.n_long, 4), p);
    return name;
}

template <int VEC>
void launch_long(const SpmmArgs &p, hipStream_t s) {
    const int lanes = (p.C + VEC - 1) / VEC;
    if (lanes > 32)      GNX_LAUNCH((k_spmm_long_partial<VEC, 8>), blocks_for(p.n_chunks, 4), p);
    else if (lanes > 16) GNX_LAUNCH((k_spmm_long_partial_group<VEC, 32, 4>), blocks_for(p.n_chunks, 4), p);
    else if (lanes > 8)  GNX_LAUNCH((k_spmm_long_partial_group<VEC, 16, 4>), blocks_for(p.n_chunks, 4), p);
    else if (lanes > 4)  GNX_LAUNCH((k_spmm_long_partial_group<VEC, 8, 4>), blocks_for(p.n_chunks, 4), p);
    else                 GNX_LAUNCH((k_spmm_long_partial_group<VEC, 4, 4>), blocks_for(p.n_chunks, 4), p);
    GNX_LAUNCH((k_spmm_long_reduce<VEC>), blocks_for(p.n_long, 4), p);
}

}  // namespace

namespace gnx {

#ifdef GNX_TUNING
int tune_override = -1;   // set through gnx_debug_set_tune (tuning builds only: make TUNING=1)
#endif

int launch_spmm(gnx_graph *g, const Csr &m, SpmmArgs &p, hipStream_t s) {
    p.rowptr = m.rowptr; p.colidx = m.colidx; p.n_rows = m.n_rows; p.n_nonempty = m.n_nonempty; p.nonempty_rows = m.nonempty_rows; p.row_list = nullptr;
    p.slot_beg = m.slot_beg; p.slot_cnt = m.slot_cnt;
    p.long_rows = m.long_rows; p.long_chunk_ptr = m.long_chunk_ptr; p.chunk_long = m.chunk_long;
    p.row_order = m.row_order;
    p.xcd_rows = m.order_window;
    p.chunk_order = m.chunk_order;
#ifdef GNX_TUNING   // kernel-variant switches exist only in tuning builds (tools/tune_spmm.py); the product library has none
    {
        static const int tune = [] { const char *e = getenv("GNX_TUNE"); return e ? atoi(e) : 0; }();
        p.tune = tune_override >= 0 ? tune_override : tune;
    }
#else
    p.tune = 0;
#endif
    p.n_long = m.n_long; p.n_chunks = m.n_chunks; p.long_row = m.long_row; p.long_chunk = m.long_chunk;
    p.partial = nullptr;
    if (p.tune & 16384) p.ldx = 0;             // (tuning builds: every gather reads row 0 -- what a launch costs without its gather misses; wrong results)
    p.skip_empty = (p.act & GNX_ACT_SKIP_EMPTY) != 0 && p.diag == nullptr;
    p.act &= ~GNX_ACT_SKIP_EMPTY;
    if (m.n_rows == 0) return GNX_OK;
    if (m.n_long > 0) {
        int rc = ensure_partial(g, (size_t)m.n_chunks * (size_t)p.C * sizeof(float), s);
        if (rc != GNX_OK) return rc;
        p.partial = g->partial;
    }
    const int vec = pick_vec(p);   // (tried for C = 128: one wave per row with float2 lanes instead of 32-lane groups of float4 -- 10.9 vs 8.2 ms)
    const char *name;
    if (p.fuse.D != nullptr) {                 // training iteration: the kernels of gnx_spmm_train.hip
        name = launch_spmm_dropped(p, vec, m.n_long > 0, s);
        g->last_kernel = name;
        GNX_HIP(hipGetLastError());
        return GNX_OK;
    }
    if (vec == 4)      { if (!(name = launch_rows_and_chunks<4>(p, s))) { name = launch_rows<4>(p, s); if (m.n_long) launch_long<4>(p, s); } }
    else if (vec == 2) { if (!(name = launch_rows_and_chunks<2>(p, s))) { name = launch_rows<2>(p, s); if (m.n_long) launch_long<2>(p, s); } }
    else               { if (!(name = launch_rows_and_chunks<1>(p, s))) { name = launch_rows<1>(p, s); if (m.n_long) launch_long<1>(p, s); } }
    g->last_kernel = name;
    GNX_HIP(hipGetLastError());
    return GNX_OK;
}

// the long rows of a launch whose short rows another translation unit's kernel took (gnx_gcnii.hip): partial sums + reduce
void launch_long_rows(const SpmmArgs &p, hipStream_t s) {
    const int vec = pick_vec(p);
    if (vec == 4) launch_long<4>(p, s);
    else if (vec == 2) launch_long<2>(p, s);
    else launch_long<1>(p, s);
}

}  // namespace gnx

extern "C" {

#ifdef GNX_TUNING
// tuning builds only (make TUNING=1), not part of include/gnx.h: lets tools/tune_spmm.py flip kernel variants inside one process
int gnx_debug_set_tune(int t) { tune_override = t; return 0; }
#endif

int gnx_spmm(gnx_graph_t g, const float *d_vals, const float *d_diag, const float *d_X, int64_t ldx, int64_t C,
             const float *d_H0, int64_t ldh0, float beta, float alpha, int act, float *d_out, int64_t ldo, void *stream) {
    int rc = check_common("gnx_spmm", g, d_X, ldx, C, d_H0, ldh0, d_out, ldo);
    if (rc != GNX_OK) return rc;
    GNX_CHECK_ARG((act & ~GNX_ACT_SKIP_EMPTY) == GNX_ACT_NONE || (act & ~GNX_ACT_SKIP_EMPTY) == GNX_ACT_RELU, "gnx_spmm: invalid activation %d", act);
    GNX_CHECK_ARG(d_diag == nullptr || g->a.n_rows == g->a.n_cols, "gnx_spmm: diag needs a square graph");
    SpmmArgs p{};
    p.vals = d_vals ? d_vals : g->raw_vals;
    p.diag = d_diag; p.X = d_X; p.ldx = ldx; p.H0 = d_H0; p.ldh0 = ldh0; p.beta = beta; p.alpha = alpha; p.act = act;
    p.out = d_out; p.ldo = ldo; p.C = (int)C;
    return launch_spmm(g, g->a, p, (hipStream_t)stream);
}

int gnx_spmm_t(gnx_graph_t g, const float *d_vals, const float *d_diag, const float *d_X, int64_t ldx, int64_t C,
               const float *d_H0, int64_t ldh0, float beta, float alpha, int act, float *d_out, int64_t ldo, void *stream) {
    int rc = check_common("gnx_spmm_t", g, d_X, ldx, C, d_H0, ldh0, d_out, ldo);
    if (rc != GNX_OK) return rc;
    GNX_CHECK_ARG(act == GNX_ACT_NONE || act == GNX_ACT_RELU, "gnx_spmm_t: invalid activation %d", act);
    GNX_CHECK_ARG(d_diag == nullptr || g->a.n_rows == g->a.n_cols, "gnx_spmm_t: diag needs a square graph");
    hipStream_t s = (hipStream_t)stream;
    rc = ensure_transpose(g, s);
    if (rc != GNX_OK) return rc;
    const float *src = d_vals ? d_vals : g->raw_vals;
    if (g->a.nnz > 0)
        hipLaunchKernelGGL(k_gather_vals, dim3(blocks_for(g->a.nnz, 256)), dim3(256), 0, s, src, g->t_perm, g->a.nnz, g->t_vals);
    SpmmArgs p{};
    p.vals = g->t_vals;
    p.diag = d_diag; p.X = d_X; p.ldx = ldx; p.H0 = d_H0; p.ldh0 = ldh0; p.beta = beta; p.alpha = alpha; p.act = act;
    p.out = d_out; p.ldo = ldo; p.C = (int)C;
    return launch_spmm(g, g->t, p, s);
}

int gnx_spmm_scatter(gnx_graph_t g, const float *d_vals, const float *d_diag, const float *d_X, int64_t ldx, int64_t C,
                     const float *d_H0, int64_t ldh0, float beta, float alpha, int act, const int32_t *d_out_rows,
                     float *d_out, int64_t ldo, void *stream) {
    int rc = check_common("gnx_spmm_scatter", g, d_X, ldx, C, d_H0, ldh0, d_out, ldo);
    if (rc != GNX_OK) return rc;
    GNX_CHECK_ARG(act == GNX_ACT_NONE || act == GNX_ACT_RELU, "gnx_spmm_scatter: invalid activation %d", act);
    GNX_CHECK_ARG(d_diag == nullptr || g->a.n_rows == g->a.n_cols, "gnx_spmm_scatter: diag needs a square graph");
    SpmmArgs p{};
    p.vals = d_vals ? d_vals : g->raw_vals;
    p.diag = d_diag; p.X = d_X; p.ldx = ldx; p.H0 = d_H0; p.ldh0 = ldh0; p.beta = beta; p.alpha = alpha; p.act = act;
    p.out = d_out; p.ldo = ldo; p.C = (int)C; p.out_rows = d_out_rows;
    return launch_spmm(g, g->a, p, (hipStream_t)stream);
}

int gnx_spmm_rows(gnx_graph_t g, const float *d_vals, const float *d_X, int64_t ldx, int64_t C, const float *d_H0, int64_t ldh0,
                  float beta, float alpha, int act, const int32_t *d_rows, float *d_out, int64_t ldo, void *stream) {
    int rc = check_common("gnx_spmm_rows", g, d_X, ldx, C, d_H0, ldh0, d_out, ldo);
    if (rc != GNX_OK) return rc;
    GNX_CHECK_ARG((act & ~GNX_ACT_SKIP_EMPTY) == GNX_ACT_NONE || (act & ~GNX_ACT_SKIP_EMPTY) == GNX_ACT_RELU, "gnx_spmm_rows: invalid activation %d", act);
    GNX_CHECK_ARG(d_rows != nullptr || g->a.n_rows == 0, "gnx_spmm_rows: NULL row map");
    SpmmArgs p{};
    p.vals = d_vals ? d_vals : g->raw_vals;
    p.X = d_X; p.ldx = ldx; p.H0 = d_H0; p.ldh0 = ldh0; p.beta = beta; p.alpha = alpha; p.act = act;
    p.out = d_out; p.ldo = ldo; p.C = (int)C; p.out_rows = d_rows; p.map_h0 = true;
    return launch_spmm(g, g->a, p, (hipStream_t)stream);
}

int gnx_graph_permute_values_t(gnx_graph_t g, const float *d_vals, float *d_vals_t_out, void *stream) {
    GNX_CHECK_ARG(g != nullptr, "gnx_graph_permute_values_t: NULL handle");
    hipStream_t s = (hipStream_t)stream;
    int rc = ensure_transpose(g, s);
    if (rc != GNX_OK) return rc;
    if (g->a.nnz == 0) return GNX_OK;
    GNX_CHECK_ARG(d_vals_t_out != nullptr, "gnx_graph_permute_values_t: NULL output");
    hipLaunchKernelGGL(k_gather_vals, dim3(blocks_for(g->a.nnz, 256)), dim3(256), 0, s, d_vals ? d_vals : g->raw_vals, g->t_perm,
                       g->a.nnz, d_vals_t_out);
    GNX_HIP(hipGetLastError());
    return GNX_OK;
}

int gnx_spmm_tv(gnx_graph_t g, const float *d_vals_t, const float *d_diag, const float *d_X, int64_t ldx, int64_t C,
                const float *d_H0, int64_t ldh0, float beta, float alpha, int act, float *d_out, int64_t ldo, void *stream) {
    int rc = check_common("gnx_spmm_tv", g, d_X, ldx, C, d_H0, ldh0, d_out, ldo);
    if (rc != GNX_OK) return rc;
    GNX_CHECK_ARG(act == GNX_ACT_NONE || act == GNX_ACT_RELU, "gnx_spmm_tv: invalid activation %d", act);
    GNX_CHECK_ARG(d_diag == nullptr || g->a.n_rows == g->a.n_cols, "gnx_spmm_tv: diag needs a square graph");
    GNX_CHECK_ARG(g->a.nnz == 0 || d_vals_t != nullptr, "gnx_spmm_tv: NULL values");
    hipStream_t s = (hipStream_t)stream;
    rc = ensure_transpose(g, s);
    if (rc != GNX_OK) return rc;
    SpmmArgs p{};
    p.vals = d_vals_t;
    p.diag = d_diag; p.X = d_X; p.ldx = ldx; p.H0 = d_H0; p.ldh0 = ldh0; p.beta = beta; p.alpha = alpha; p.act = act;
    p.out = d_out; p.ldo = ldo; p.C = (int)C;
    return launch_spmm(g, g->t, p, s);
}

int gnx_ppr_step(gnx_graph_t g, const float *d_vals, const float *d_diag, const float *d_H, const float *d_H0, float a,
                 int64_t C, int act, float *d_out, void *stream) {
    GNX_CHECK_ARG(d_H0 != nullptr, "gnx_ppr_step: NULL H0");
    return gnx_spmm(g, d_vals, d_diag, d_H, C, C, d_H0, C, (float)(1.0 - (double)a), a, act, d_out, C, stream);
}

int gnx_appnp_propagate(gnx_graph_t g, const float *d_vals, const float *d_diag, const float *d_H0, float a, int K,
                        int64_t C, float *d_out, float *d_work, void *stream) {
    return gnx_appnp_propagate_act(g, d_vals, d_diag, d_H0, a, K, C, GNX_ACT_NONE, d_out, d_work, stream);
}

int gnx_appnp_propagate_act(gnx_graph_t g, const float *d_vals, const float *d_diag, const float *d_H0, float a, int K,
                            int64_t C, int act, float *d_out, float *d_work, void *stream) {
    GNX_CHECK_ARG(g != nullptr, "gnx_appnp_propagate: NULL handle");
    GNX_CHECK_ARG(act == GNX_ACT_NONE || act == GNX_ACT_RELU, "gnx_appnp_propagate: invalid activation %d", act);
    GNX_CHECK_ARG(K >= 0, "gnx_appnp_propagate: negative iteration count");
    GNX_CHECK_ARG(g->a.n_rows == g->a.n_cols, "gnx_appnp_propagate: needs a square graph");
    GNX_CHECK_ARG(d_H0 && d_out && (K < 2 || d_work), "gnx_appnp_propagate: NULL buffer");
    GNX_CHECK_ARG(d_out != d_H0 && d_work != d_H0 && d_out != d_work, "gnx_appnp_propagate: H0, out and work must be distinct");
    if (K == 0) {
        GNX_HIP(hipMemcpyAsync(d_out, d_H0, (size_t)g->a.n_rows * C * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        return GNX_OK;
    }
    const int64_t n = g->a.n_rows;
    hipStream_t s = (hipStream_t)stream;
    // Narrow features on a large graph: every gather moves a whole 128-byte line for a 16..64-byte row, so what counts is how
    // often a line is found in cache.  The K iterations then run on the degree-relabelled copy of the matrix (hub rows adjacent:
    // four to eight of the rows that receive most gathers share a line): H0 is permuted once on the way in, the LAST iteration
    // scatters its rows straight back into the caller's order.  -16..-21 % per iteration at C = 16 / 8 (RMAT 10M / 100M); the
    // sums run over a row's columns in the relabelled order, so results agree with the plain path to float32 rounding.
    if (C <= RELABEL_MAX_C && n >= (1 << 20) && g->a.nnz >= n && d_diag == nullptr && g->a.order_window == 0) {
        int rc = ensure_relabel(g, s);
        if (rc != GNX_OK) return rc;
        rc = ensure_relabel_features(g, (size_t)n * C * sizeof(float), s);
        if (rc != GNX_OK) return rc;
        hipLaunchKernelGGL(k_gather_vals, dim3(blocks_for(g->a.nnz, 256)), dim3(256), 0, s, d_vals ? d_vals : g->raw_vals, g->r_perm, g->a.nnz,
                           g->r_vals);
        GNX_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_gather_rows32, dim3((unsigned)std::min<int64_t>(blocks_for(n * C, 256), 1 << 22)), dim3(256), 0, s, d_H0, C,
                           g->r_order, n, (int)C, g->r_feat, C);
        GNX_HIP(hipGetLastError());
        const float *src = g->r_feat;
        for (int k = 0; k < K; ++k) {
            const bool last = k == K - 1;
            float *dst = last ? d_out : (((K - 2 - k) % 2 == 0) ? d_work : d_out);
            SpmmArgs p{};
            p.vals = g->r_vals; p.X = src; p.ldx = C; p.H0 = g->r_feat; p.ldh0 = C; p.beta = (float)(1.0 - (double)a); p.alpha = a;
            // rows without entries: written by the last iteration (it scatters every row); in between only if somebody gathers them
            // (with a relu such a row is relu(a * H0) after every iteration: just as constant)
            p.act = (!last && (k >= 2 || g->r.empty_rows_unreferenced)) ? (act | GNX_ACT_SKIP_EMPTY) : act;
            p.out = dst; p.ldo = C; p.C = (int)C;
            p.out_rows = last ? g->r_order : nullptr;               // relabelled row i is the caller's row r_order[i]
            rc = launch_spmm(g, g->r, p, s);
            if (rc != GNX_OK) return rc;
            src = dst;
        }
        return GNX_OK;
    }
    // A row without entries is act(a * H0) after every iteration and nobody's sum depends on when it was written: such rows are
    // computed the first time each of the two buffers is a destination (k = 0, 1) and left alone afterwards (GNX_ACT_SKIP_EMPTY)
    // -- on the R-MAT workloads 60 % of the rows, 7-10 % of an iteration's bytes.  Same arithmetic, same bits.
    const float *src = d_H0;
    for (int k = 0; k < K; ++k) {
        float *dst = ((K - 1 - k) % 2 == 0) ? d_out : d_work;
        // ... and when no entry points at such a row (g->a.empty_rows_unreferenced: every symmetric pattern) nobody ever gathers it: the
        // work buffer never needs it, the result buffer gets it the first time it is a destination
        const bool settled = k >= 2 || (g->a.empty_rows_unreferenced && dst == d_work);
        const int act_k = (settled && d_diag == nullptr) ? (act | GNX_ACT_SKIP_EMPTY) : act;
        int rc = gnx_spmm(g, d_vals, d_diag, src, C, C, d_H0, C, (float)(1.0 - (double)a), a, act_k, dst, C, stream);
        if (rc != GNX_OK) return rc;
        src = dst;
    }
    return GNX_OK;
}

}  // extern "C"

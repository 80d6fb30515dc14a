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
//   * narrow features: G = 4..32 lanes per row, 64/G rows per wave, rows taken in a degree-binned
//     order (Csr::row_order) so that the rows sharing a wave have similar lengths; for G <= 8 the next
//     (col, val) batch is prefetched behind the gathers;
//   * power-law rows: a row with more than p.long_row entries is cut into p.long_chunk-entry chunks summed
//     by separate waves into a partial slab (wide: lanes across columns; narrow: sub-groups of lanes
//     across the chunk's entries + a fixed xor tree), then added in chunk order by a second kernel
//     (fixed order: results are bitwise reproducible, no float atomics).  Chunks are processed in
//     column-window order (Csr::chunk_order) so the hub rows they share stay in L2 / Infinity Cache.
#include <stdlib.h>
#include <algorithm>

#include "gnx_internal.h"

using namespace gnx;

namespace {

template <int VEC> struct VecT;
template <> struct VecT<1> { using type = float; };
template <> struct VecT<2> { using type = float2; };
template <> struct VecT<4> { using type = float4; };

template <int VEC>
__device__ __forceinline__ void vload(float (&x)[VEC], const float *__restrict__ p) {
    using T = typename VecT<VEC>::type;
    const T v = *reinterpret_cast<const T *>(p);
    __builtin_memcpy(x, &v, sizeof(T));
}
template <int VEC>
__device__ __forceinline__ void vstore(float *__restrict__ p, const float (&x)[VEC]) {
    using T = typename VecT<VEC>::type;
    T v;
    __builtin_memcpy(&v, x, sizeof(T));
    *reinterpret_cast<T *>(p) = v;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int VEC> struct NatT;
template <> struct NatT<1> { using type = float; };
template <> struct NatT<2> { using type = f32x2; };
template <> struct NatT<4> { using type = f32x4; };

// streaming (touched once per launch) data: non-temporal so it does not evict gathered rows
template <int VEC>
__device__ __forceinline__ void vload_nt(float (&x)[VEC], const float *__restrict__ p) {
    using T = typename NatT<VEC>::type;
    const T v = __builtin_nontemporal_load(reinterpret_cast<const T *>(p));
    __builtin_memcpy(x, &v, sizeof(T));
}
template <int VEC>
__device__ __forceinline__ void vstore_nt(float *__restrict__ p, const float (&x)[VEC]) {
    using T = typename NatT<VEC>::type;
    T v;
    __builtin_memcpy(&v, x, sizeof(T));
    __builtin_nontemporal_store(v, reinterpret_cast<T *>(p));
}

__device__ __forceinline__ int readlane_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// Sum of w_e * X[col_e, c .. c+VEC) over entries [beg, end) of one row; the whole wave works
// on the same entries (beg/end wave-uniform), lane `lane` owns columns c .. c+VEC.
template <int VEC, int U, bool FUSE = false>
__device__ __forceinline__ void wave_accumulate(const int32_t *__restrict__ colidx, const float *__restrict__ vals,
                                                const float *__restrict__ X, int64_t ldx, int64_t beg, int64_t end,
                                                int c, int lane, float (&acc)[VEC], bool nt_index = false,
                                                const DropFuse *fuse = nullptr, int64_t row = 0) {
    for (int64_t base = beg; base < end; base += 64) {
        const int n = (int)((end - base) < 64 ? (end - base) : 64);
        int mycol = 0;
        float myval = 0.f;
        if (lane < n) {
            if (nt_index) {
                mycol = __builtin_nontemporal_load(colidx + base + lane);
                myval = __builtin_nontemporal_load(vals + base + lane);
            } else {
                mycol = colidx[base + lane];
                myval = vals[base + lane];
            }
            if (FUSE) myval = dropped_weight(*fuse, myval, row, mycol);      // one entry per lane: 64 weights per wave instruction
        }
        int i = 0;
        if (FUSE) {
            // a dropped entry has weight exactly 0: its row is not gathered at all (fmaf(0, x, acc) == acc for finite x), so a
            // training iteration moves only the kept rows -- half of them at p = 0.5; kept entries stay in ascending order
            uint64_t keep = __ballot(myval != 0.f);
            while (keep) {
                float x[U][VEC];
                int idx[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    idx[u] = keep ? (int)__builtin_ctzll(keep) : -1;
                    if (keep) keep &= keep - 1;
                    if (idx[u] >= 0) {
                        const int j = readlane_i(mycol, idx[u]);
                        vload<VEC>(x[u], X + (int64_t)j * ldx + c);
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (idx[u] >= 0) {
                        const float w = readlane_f(myval, idx[u]);
#pragma unroll
                        for (int v = 0; v < VEC; ++v) acc[v] = fmaf(w, x[u][v], acc[v]);
                    }
                }
            }
            continue;
        }
        for (; i + U <= n; i += U) {
            float x[U][VEC];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = readlane_i(mycol, i + u);
                vload<VEC>(x[u], X + (int64_t)j * ldx + c);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float w = readlane_f(myval, i + u);
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[v] = fmaf(w, x[u][v], acc[v]);
            }
        }
        if (i < n) {  // 1 .. U-1 entries left: issue all loads, then all FMAs (wave-uniform branches)
            float x[U][VEC];
#pragma unroll
            for (int u = 0; u < U - 1; ++u) {
                if (i + u < n) {
                    const int j = readlane_i(mycol, i + u);
                    vload<VEC>(x[u], X + (int64_t)j * ldx + c);
                }
            }
#pragma unroll
            for (int u = 0; u < U - 1; ++u) {
                if (i + u < n) {
                    const float w = readlane_f(myval, i + u);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[v] = fmaf(w, x[u][v], acc[v]);
                }
            }
        }
    }
}

// filter.py:20-22: out = act(acc*beta + h0*alpha), with the add_eye diagonal folded in first.
template <int VEC>
__device__ __forceinline__ void epilogue_store(const SpmmArgs &p, int64_t row, int c, bool active, float (&acc)[VEC],
                                               bool nt = false) {
    if (!active) return;
    if (p.diag) {
        const float d = p.diag[row];
        float xr[VEC];
        vload<VEC>(xr, p.X + row * p.ldx + c);
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = fmaf(d, xr[v], acc[v]);
    }
    if (p.out2) {                                  // second result of the same sums (see SpmmArgs::out2)
        const float f2 = p.out2_scale ? p.out2_scale[row] : 1.f;
        float o2[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) o2[v] = (acc[v] * p.beta2) * f2;
        vstore<VEC>(p.out2 + row * p.ldo2 + c, o2);
    }
    float o[VEC];
    const int64_t orow = p.out_rows ? (int64_t)p.out_rows[row] : row;
    if (p.H0) {
        const int64_t hrow = p.map_h0 ? orow : row;   // gnx_spmm_rows: H0 is indexed like the output
        float h0[VEC];
        if (nt) vload_nt<VEC>(h0, p.H0 + hrow * p.ldh0 + c);
        else vload<VEC>(h0, p.H0 + hrow * p.ldh0 + c);
#pragma unroll
        for (int v = 0; v < VEC; ++v) o[v] = fmaf(acc[v], p.beta, h0[v] * p.alpha);   // spelled out: every kernel variant rounds alike
    } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) o[v] = acc[v] * p.beta;
    }
    if (p.act == GNX_ACT_RELU) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) o[v] = fmaxf(o[v], 0.f);
    }
    if (p.out_scale) {
        const float os = p.out_scale[row];
#pragma unroll
        for (int v = 0; v < VEC; ++v) o[v] *= os;
    }
    if (nt) vstore_nt<VEC>(p.out + orow * p.ldo + c, o);
    else vstore<VEC>(p.out + orow * p.ldo + c, o);
}

// Which block of row slots this workgroup takes.  Default: its own index.  With a locality order (SpmmArgs::xcd_rows > 0) the index
// is remapped so that the workgroups the dispatcher places on one XCD (observed: round-robin, blockIdx % 8 -- a speed assumption,
// never a correctness one: the map is a bijection of the padded grid whatever the placement) take whole CHUNKS of xcd_chunk
// consecutive blocks, chunk j * 8 + x going to group x: an XCD then works on one contiguous stretch of the numbering at a time and
// its L2 holds THAT neighbourhood of H, instead of every L2 holding a slice of everything in flight (cdna_hip_programming.md T1).
// A chunk is one WINDOW's worth of slots: inside a window the rows are sorted by length, so any finer chunk hands the same XCDs the
// heavy part of every window (measured: chunks of a quarter window 2 x slower on orders with heavy heads), and contiguous eighths
// of the whole order hold unequal work (profiles/NOTES.md round 5).  The launcher pads the grid to a multiple of 8 chunks; padded
// blocks map past the last slot and leave.
__device__ __forceinline__ int64_t xcd_block(const SpmmArgs &p) {
    const uint32_t b = blockIdx.x;
    if (p.xcd_rows <= 0) return (int64_t)b;
    const uint32_t x = b & 7u, i = b >> 3, ch = p.xcd_chunk;
    return ((int64_t)(i / ch) * 8 + x) * ch + i % ch;
}

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

// ---- GCNII layer: SpMM + mix + C x C transform on the matrix cores + activation, one launch ------------------------
//   out[i,:] = act( (beta * sum_j A[i,j] X[j,:] + alpha * H0[i,:]) . M ),   M = (1-b) I + b W   (gcn.py:22-27)
// A 512-thread block: every wave gathers a tile of 16 rows (4 NT lanes of float4 per row, U entries in flight per lane,
// rows in degree-binned order), leaves the mixed rows in its LDS tile -- in inference they never go to HBM; in training
// (`mixed` given) each lane also stores its piece of the mixed row, which the backward needs for dM = T^T g, so that the row is
// written once and NOT read back for the transform -- multiplies the tile by M (shared by the block in LDS, row stride = 4 mod 32
// banks) with v_mfma_f32_16x16x4_f32 (exact f32), and stores whole rows.  C = 16 NT for NT in {1, 2, 4}.  (NT = 8, C = 128, fits --
// 135 KB of the CU's 160 KB of LDS -- but leaves one block of eight waves per CU: measured 19.2 ms against 11.8 ms for SpMM+mix
// followed by the dense kernel, so wide layers keep the two launches.)  Rows longer than p.long_row are left to the long-row
// kernels + the dense kernel.
template <int NT, int U, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_spmm_gcnii(const SpmmArgs p, const float *__restrict__ M, int64_t ldm, float *__restrict__ mixed) {
    constexpr int C = 16 * NT, G = 4 * NT, RPP = 64 / G, PASSES = 16 / RPP, STRIDE = C + 4;
    __shared__ float Ms[C * STRIDE];
    __shared__ float Ts[WPB][16 * STRIDE];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (int idx = threadIdx.x; idx < C * C; idx += 64 * WPB) Ms[(idx / C) * STRIDE + idx % C] = M[(int64_t)(idx / C) * ldm + idx % C];
    __syncthreads();
    const int64_t tile = (int64_t)blockIdx.x * WPB + wave;
    if (tile * 16 >= p.n_rows) return;
    float *__restrict__ T = Ts[wave];
    const int sub = lane % G, c = sub * 4;
    int64_t rows[PASSES];
    bool live[PASSES];
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const int rr = ps * RPP + lane / G;
        const int64_t slot = tile * 16 + rr;
        int64_t row = -1;
        int64_t beg = 0, end = 0;
        if (slot < p.n_rows) {
            row = p.row_order ? (int64_t)p.row_order[slot] : slot;
            beg = p.rowptr[row]; end = p.rowptr[row + 1];
        }
        live[ps] = row >= 0 && end - beg <= p.long_row;
        rows[ps] = row;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (live[ps]) {
            const float *__restrict__ Xc = p.X + c;
            for (int64_t e = beg; e < end; e += U) {
                float x[U][4];
                float w[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (e + u < end) {
                        const int j = p.colidx[e + u];
                        w[u] = p.vals[e + u];
                        vload<4>(x[u], Xc + (int64_t)j * p.ldx);
                    } else {
                        w[u] = 0.f;
#pragma unroll
                        for (int v = 0; v < 4; ++v) x[u][v] = 0.f;
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int v = 0; v < 4; ++v) acc[v] = fmaf(w[u], x[u][v], acc[v]);
            }
            float h0[4];
            vload<4>(h0, p.H0 + row * p.ldh0 + c);
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[v] = fmaf(acc[v], p.beta, h0[v] * p.alpha);      // filter.py:20-21 / gcn.py:25
            if (mixed) vstore<4>(mixed + row * (int64_t)C + c, acc);
        }
        vstore<4>(T + rr * STRIDE + c, acc);
    }
    __builtin_amdgcn_wave_barrier();
    // tile . M : A[m = lane & 15][k = 4 kk + (lane >> 4)] from the tile, B[k][n = lane & 15] from Ms
    const int cc = lane & 15, g = lane >> 4;
    f32x4 d[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) d[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int kk = 0; kk < C / 4; ++kk) {
        const float a = T[cc * STRIDE + 4 * kk + g];
        const float *__restrict__ mrow = Ms + (4 * kk + g) * STRIDE + cc;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) d[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, mrow[16 * nt], d[nt], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
    // D: lane (cc, g), register r -> row 4g + r, column 16 nt + cc; back through the tile so that rows leave as whole float4 rows
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = d[nt][r];
            if (p.act == GNX_ACT_RELU) v = fmaxf(v, 0.f);
            T[(4 * g + r) * STRIDE + 16 * nt + cc] = v;
        }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        if (!live[ps]) continue;
        const int rr = ps * RPP + lane / G;
        float o[4];
        vload<4>(o, T + rr * STRIDE + c);
        vstore<4>(p.out + rows[ps] * p.ldo + c, o);
    }
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

template <int VEC>
__global__ __launch_bounds__(256) void k_spmm_long_reduce(const SpmmArgs p) {
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t li = (int64_t)blockIdx.x * 4 + wib;
    if (li >= p.n_long) return;
    const int64_t row = p.long_rows[li];
    const int64_t cb = p.long_chunk_ptr[li], ce = p.long_chunk_ptr[li + 1];
    for (int c0 = 0; c0 < p.C; c0 += 64 * VEC) {
        const int c = c0 + lane * VEC;
        const bool active = c < p.C;
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
        if (active) {
            for (int64_t k = cb; k < ce; ++k) {  // chunk order
                float x[VEC];
                vload<VEC>(x, p.partial + k * (int64_t)p.C + c);
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[v] += x[v];
            }
        }
        epilogue_store<VEC>(p, row, c, active, acc);
    }
}

// ---- training iterations: the dropped + re-normalised values are produced inside the SpMM (gnx_spmm_dropped) ------------
// Same row / lane mapping as the kernels above; what differs is where an entry's weight comes from: p.vals holds the RAW
// values and every weight is (D[row] * drop(raw)) * D[col] (layered.py:47-50 + gnn.py:41-42), computed ONCE per entry by one
// lane and handed to the lanes that need it (readlane / shuffles), so the hash costs one evaluation per stored entry.
template <int VEC, int U, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_spmm_wave_drop(const SpmmArgs p) {
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t slot = p.slot0 + xcd_block(p) * WPB + wib;
    if (slot >= p.n_rows) return;
    const int64_t row = p.row_list ? (int64_t)__builtin_amdgcn_readfirstlane(p.row_list[slot]) : slot;
    const int64_t beg = p.rowptr[row], end = p.rowptr[row + 1];
    if (end - beg > p.long_row) return;
    if (p.skip_empty && beg == end) return;   // GNX_ACT_SKIP_EMPTY (chained training loops: nobody gathers this row, a later launch writes it)
    for (int c0 = 0; c0 < p.C; c0 += 64 * VEC) {
        const int c = c0 + lane * VEC;
        const bool active = c < p.C;
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
        wave_accumulate<VEC, U, true>(p.colidx, p.vals, p.X, p.ldx, beg, end, active ? c : 0, lane, acc, false, &p.fuse, row);
        epilogue_store<VEC>(p, row, c, active, acc);
    }
}

// PIPE: the (col, raw value) pair a lane owns in the NEXT round is loaded before this round's kept entries are gathered, so a row
// of more than G entries pays the index latency once instead of once per round.
template <int VEC, int G, int U, bool PIPE>
__global__ __launch_bounds__(256) void k_spmm_group_drop(const SpmmArgs p) {
    constexpr int RPB = 256 / G;
    const int sub = threadIdx.x % G;
    const int64_t slot = p.slot0 + xcd_block(p) * RPB + threadIdx.x / G;
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
        int ncol = 0;
        float nraw = 0.f;
        if (PIPE && beg + sub < end) { ncol = p.colidx[beg + sub]; nraw = p.vals[beg + sub]; }
        for (int64_t base = beg; base < end; base += G) {          // G entries per round: lane `sub` owns entry base + sub
            const int n = (int)((end - base) < G ? (end - base) : G);
            int mycol = 0;
            float myw = 0.f;
            if (PIPE) {
                const int ccol = ncol;
                const float craw = nraw;
                if (base + G + sub < end) { ncol = p.colidx[base + G + sub]; nraw = p.vals[base + G + sub]; }
                if (sub < n) { mycol = ccol; myw = dropped_weight(p.fuse, craw, row, ccol); }
            } else if (sub < n) {
                mycol = p.colidx[base + sub];
                myw = dropped_weight(p.fuse, p.vals[base + sub], row, mycol);
            }
            // dropped entries (weight exactly 0) are not gathered: the group walks only the kept entries of its round, in order
            const uint64_t all = __ballot(myw != 0.f);
            uint32_t keep = (uint32_t)(all >> ((threadIdx.x & 63) / G * G)) & (G == 32 ? 0xFFFFFFFFu : ((1u << G) - 1u));
            while (keep) {
                float x[U][VEC];
                float w[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (keep) {
                        const int idx = __builtin_ctz(keep);
                        keep &= keep - 1;
                        const int j = __shfl(mycol, idx, G);
                        w[u] = __shfl(myw, idx, G);
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

template <int VEC, int U>
__global__ __launch_bounds__(256) void k_spmm_long_partial_drop(const SpmmArgs p) {
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t cslot = (int64_t)blockIdx.x * 4 + wib;
    if (cslot >= p.n_chunks) return;
    const int64_t chunk = p.chunk_order ? (int64_t)p.chunk_order[cslot] : cslot;
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
        wave_accumulate<VEC, U, true>(p.colidx, p.vals, p.X, p.ldx, beg, end, active ? c : 0, lane, acc, false, &p.fuse, row);
        if (active) vstore<VEC>(p.partial + chunk * (int64_t)p.C + c, acc);
    }
}

// narrow long rows: the wave computes 64 weights per round (one per lane); sub-group s then takes entries s, s + NS, ... of the
// round, which is the entry -> sub-group dealing of k_spmm_long_partial_group (so the partial sums are bitwise the same)
template <int VEC, int G, int U>
__global__ __launch_bounds__(256) void k_spmm_long_partial_group_drop(const SpmmArgs p) {
    constexpr int NS = 64 / G;
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t cslot = (int64_t)blockIdx.x * 4 + wib;
    if (cslot >= p.n_chunks) return;
    const int64_t chunk = p.chunk_order ? (int64_t)p.chunk_order[cslot] : cslot;
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
    // the reference kernel walks e = beg + sub + k * NS (k = 0, 1, ...) in batches of U: entry index within the chunk = sub + k NS.
    // A round of 64 entries covers k = 0 .. 64/NS - 1 = G - 1 for every sub-group.
    for (int64_t base = beg; base < end; base += 64) {
        const int n = (int)((end - base) < 64 ? (end - base) : 64);
        int mycol = 0;
        float myw = 0.f;
        if (lane < n) {
            mycol = p.colidx[base + lane];
            myw = dropped_weight(p.fuse, p.vals[base + lane], row, mycol);
        }
#pragma unroll 1
        for (int k = 0; k < G; k += U) {
            float x[U][VEC];
            float w[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int src = sub + (k + u) * NS;                  // entry of the round this sub-group takes in slot k + u
                const int j = __shfl(mycol, src);
                w[u] = __shfl(myw, src);
                if (k + u < G && src < n && w[u] != 0.f) vload<VEC>(x[u], Xc + (int64_t)j * p.ldx);   // dropped: not gathered
                else {
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
#pragma unroll
    for (int off = G; off < 64; off <<= 1)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += __shfl_xor(acc[v], off);
    if (sub == 0 && active) vstore<VEC>(p.partial + chunk * (int64_t)p.C + c, acc);
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
__global__ __launch_bounds__(256) void k_gather_rows(const float *__restrict__ X, int64_t ldx, const int64_t *__restrict__ idx,
                                                     int64_t n_idx, int C, float *__restrict__ out, int64_t ldo) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_idx) return;
    const int64_t src = idx[r];
    for (int c = lane * VEC; c < C; c += 64 * VEC) {
        float x[VEC];
        vload<VEC>(x, X + src * ldx + c);
        vstore<VEC>(out + r * ldo + c, x);
    }
}

// The HBM yardsticks bench.py measures beside the SpMM (gnx_stream_copy / gnx_stream_read): every block walks tiles of
// blockDim.x * U float4, U independent 16-byte non-temporal loads in flight per lane (each wave instruction moves 1 KiB of
// consecutive bytes), non-temporal stores.  Launch shapes from tools/stream_bench.hip's sweep on an MI355X
// (profiles/r03_stream_sweep.txt): a copy is fastest with many waves (1024 threads x 4 blocks per CU, 4 loads in flight:
// 5.8 TB/s read + write), a read with FEW (one 256-thread block per CU, 8 loads in flight: 7.2 TB/s) -- more waves only
// spread the DRAM pages thinner.
template <int U, bool COPY>
__global__ void k_stream(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, int64_t n4, float *__restrict__ sink) {
    const int64_t tile = (int64_t)blockDim.x * U;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    int64_t base = (int64_t)blockIdx.x * tile;
    for (; base + tile <= n4; base += (int64_t)gridDim.x * tile) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(src + base + (int64_t)u * blockDim.x + threadIdx.x);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (COPY) __builtin_nontemporal_store(v[u], dst + base + (int64_t)u * blockDim.x + threadIdx.x);
            else acc += v[u];
        }
    }
    for (int64_t i = base + threadIdx.x; i < n4; i += blockDim.x) {      // the one ragged tile: exactly one block has base < n4 here
        if (COPY) dst[i] = src[i];
        else acc += src[i];
    }
    if (!COPY) {
        float v = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((threadIdx.x & 63) == 0) atomicAdd(sink + blockIdx.x % 64, v);
    }
}

// out = sum_j coef[j] * src[j], elementwise over up to 16 equally long arrays, summed in index order (fixed rounding): the
// backward of the K-iteration loop ends with dH0 = g_0 + a (g_1 + ... + g_K) -- one pass over the K + 1 gradients it kept
// instead of a read-modify-write of dH0 after every iteration
constexpr int LINCOMB_MAX = 16;
struct LinComb {
    const float *src[LINCOMB_MAX];
    float coef[LINCOMB_MAX];
    int k;
};
__global__ __launch_bounds__(256) void k_lincomb(LinComb a, int64_t n4, int64_t n, float *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 v[LINCOMB_MAX];
#pragma unroll
        for (int j = 0; j < LINCOMB_MAX; ++j)
            if (j < a.k) v[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(a.src[j]) + i);
        f32x4 acc = v[0] * a.coef[0];
#pragma unroll
        for (int j = 1; j < LINCOMB_MAX; ++j)
            if (j < a.k) {
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = fmaf(v[j][c], a.coef[j], acc[c]);
            }
        __builtin_nontemporal_store(acc, reinterpret_cast<f32x4 *>(out) + i);
    }
    if (blockIdx.x == 0 && (int64_t)threadIdx.x < n - 4 * n4) {           // the last n % 4 elements
        const int64_t i = 4 * n4 + threadIdx.x;
        float acc = a.src[0][i] * a.coef[0];
        for (int j = 1; j < a.k; ++j) acc = fmaf(a.src[j][i], a.coef[j], acc);
        out[i] = acc;
    }
}

inline unsigned blocks_for(int64_t n, int per_block) { return (unsigned)((n + per_block - 1) / per_block); }

inline bool aligned(const void *p, size_t a) { return p == nullptr || ((uintptr_t)p % a) == 0; }

// widest vector width every row start allows
int pick_vec(const SpmmArgs &p) {
    for (int vec = 4; vec > 1; vec >>= 1) {
        const size_t a = vec * sizeof(float);
        if (p.C % vec == 0 && p.ldx % vec == 0 && p.ldo % vec == 0 && (p.H0 == nullptr || p.ldh0 % vec == 0) &&
            aligned(p.X, a) && aligned(p.out, a) && aligned(p.H0, a))
            return vec;
    }
    return 1;
}

#define GNX_LAUNCH(kern, grid, ...) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, s, __VA_ARGS__)

// One launch holds at most 2^32 work-items (the dispatch packet's grid size is 32 bits).  A wave per row reaches that at 67M rows,
// 32 lanes per row at 134M -- sizes a 288 GB card holds -- so the row kernels are dealt in pieces of at most 2^31 work-items
// (SpmmArgs::slot0 = first row slot of the piece; one piece for everything smaller).
#define GNX_ROW_PIECES(kern, rows_per_block, threads)                                                                       \
    do {                                                                                                                    \
        const int64_t per_launch_ = (((int64_t)1 << 31) / (threads)) * (rows_per_block);                                     \
        for (int64_t r0_ = 0; r0_ < p.n_rows; r0_ += per_launch_) {                                                          \
            SpmmArgs q_ = p;                                                                                                \
            q_.slot0 = r0_;                                                                                                 \
            const int64_t rows_ = p.n_rows - r0_ < per_launch_ ? p.n_rows - r0_ : per_launch_;                               \
            q_.n_rows = r0_ + rows_;    /* a piece ends where the next begins (padded blocks of the XCD map must not run on) */       \
            unsigned grid_ = blocks_for(rows_, rows_per_block);                                                             \
            if (q_.xcd_rows > 0 && rows_ < 64 * q_.xcd_rows) q_.xcd_rows = 0;   /* a few windows only: they would not fill 8 XCDs evenly */   \
            if (q_.xcd_rows > 0) {     /* xcd_block: whole chunks, the grid padded to 8 of them */                           \
                q_.xcd_chunk = (uint32_t)((q_.xcd_rows + (rows_per_block) - 1) / (rows_per_block));                          \
                const unsigned span_ = 8u * q_.xcd_chunk;                                                                   \
                grid_ = (grid_ + span_ - 1) / span_ * span_;                                                                \
            }                                                                                                               \
            hipLaunchKernelGGL(kern, dim3(grid_), dim3(threads), 0, s, q_);                                                  \
        }                                                                                                                   \
    } while (0)

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
    if (lanes > 4)  { GNX_GROUP(8, 32, true); return "spmm_group8"; }
    // (tried: 2 lanes per row for C <= 8 -- 2.40 vs 2.28 ms: every gather is one 128-byte line whatever the width, lane use is not the limit)
    GNX_GROUP(4, 64, true);
    return "spmm_group4";
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

[[maybe_unused]] constexpr bool DROP_U8 = false, DROP_PIPE = false;     // product defaults of the training row kernels (see launch_rows_drop)
constexpr int DROP_LONG_U = 4;

template <int VEC>
const char *launch_rows_drop(const SpmmArgs &p0, hipStream_t s) {
    SpmmArgs p = p0;
    const int lanes = (p.C + VEC - 1) / VEC;
    if (p.skip_empty && p.n_nonempty < p.n_rows) {            // as launch_rows: the slots of the rows without entries are not launched
        if (lanes <= 32 && p.row_order != nullptr) p.n_rows = p.n_nonempty;
        else if (lanes > 32 && p.nonempty_rows != nullptr) { p.row_list = p.nonempty_rows; p.n_rows = p.n_nonempty; }
    }
    if (p.n_rows == 0) return "spmm_none_drop";
    if (lanes > 32) {
        if (p.C <= 64 * VEC) GNX_ROW_PIECES((k_spmm_wave_drop<VEC, 8, 8>), 8, 512);
        else                 GNX_ROW_PIECES((k_spmm_wave_drop<VEC, 8, 4>), 4, 256);
        return "spmm_wave_drop";
    }
    // U gathers in flight per lane and the index prefetch: round-4 A/B on the config-4 graph (tuning build bits 1 << 17 = U 8,
    // 1 << 19 = PIPE; profiles/NOTES.md)
#ifdef GNX_TUNING
    const bool u8 = (p.tune & (1 << 17)) != 0, pipe = (p.tune & (1 << 19)) != 0;
#else
    const bool u8 = DROP_U8, pipe = DROP_PIPE;
#endif
#define GNX_DROP_ROWS(G, RPB)                                                                                         \
    do {                                                                                                              \
        if (u8 && pipe)  GNX_ROW_PIECES((k_spmm_group_drop<VEC, G, 8, true>), RPB, 256);                              \
        else if (u8)     GNX_ROW_PIECES((k_spmm_group_drop<VEC, G, 8, false>), RPB, 256);                             \
        else if (pipe)   GNX_ROW_PIECES((k_spmm_group_drop<VEC, G, 4, true>), RPB, 256);                              \
        else             GNX_ROW_PIECES((k_spmm_group_drop<VEC, G, 4, false>), RPB, 256);                             \
    } while (0)
    if (lanes > 16) { GNX_DROP_ROWS(32, 8); return "spmm_group32_drop"; }
    if (lanes > 8)  { GNX_DROP_ROWS(16, 16); return "spmm_group16_drop"; }
    if (lanes > 4)  { GNX_DROP_ROWS(8, 32); return "spmm_group8_drop"; }
    GNX_DROP_ROWS(4, 64);
#undef GNX_DROP_ROWS
    return "spmm_group4_drop";
}

template <int VEC>
void launch_long_drop(const SpmmArgs &p, hipStream_t s) {
    const int lanes = (p.C + VEC - 1) / VEC;
    if (lanes > 32)      GNX_LAUNCH((k_spmm_long_partial_drop<VEC, 8>), blocks_for(p.n_chunks, 4), p);
#ifdef GNX_TUNING
    else if (p.tune & (1 << 18)) {
        if (lanes > 16)      GNX_LAUNCH((k_spmm_long_partial_group_drop<VEC, 32, 8>), blocks_for(p.n_chunks, 4), p);
        else if (lanes > 8)  GNX_LAUNCH((k_spmm_long_partial_group_drop<VEC, 16, 8>), blocks_for(p.n_chunks, 4), p);
        else if (lanes > 4)  GNX_LAUNCH((k_spmm_long_partial_group_drop<VEC, 8, 8>), blocks_for(p.n_chunks, 4), p);
        else                 GNX_LAUNCH((k_spmm_long_partial_group_drop<VEC, 4, 4>), blocks_for(p.n_chunks, 4), p);
    }
#endif
    else if (lanes > 16) GNX_LAUNCH((k_spmm_long_partial_group_drop<VEC, 32, DROP_LONG_U>), blocks_for(p.n_chunks, 4), p);
    else if (lanes > 8)  GNX_LAUNCH((k_spmm_long_partial_group_drop<VEC, 16, DROP_LONG_U>), blocks_for(p.n_chunks, 4), p);
    else if (lanes > 4)  GNX_LAUNCH((k_spmm_long_partial_group_drop<VEC, 8, DROP_LONG_U>), blocks_for(p.n_chunks, 4), p);
    else                 GNX_LAUNCH((k_spmm_long_partial_group_drop<VEC, 4, 4>), blocks_for(p.n_chunks, 4), p);
    GNX_LAUNCH((k_spmm_long_reduce<VEC>), blocks_for(p.n_long, 4), p);
}

int check_common(const char *fn, gnx_graph *g, const float *X, int64_t ldx, int64_t C, const float *H0, int64_t ldh0,
                 float *out, int64_t ldo) {
    GNX_CHECK_ARG(g != nullptr, "%s: NULL handle", fn);
    GNX_CHECK_ARG(C >= 1 && C <= (1 << 20), "%s: feature width %lld not in [1, 2^20]", fn, (long long)C);
    GNX_CHECK_ARG(X != nullptr && out != nullptr, "%s: NULL X/out", fn);
    GNX_CHECK_ARG(ldx >= C && ldo >= C && (H0 == nullptr || ldh0 >= C || ldh0 == 0), "%s: leading dimension smaller than C", fn);
    GNX_CHECK_ARG((const void *)X != (const void *)out, "%s: out must not alias X", fn);
    return GNX_OK;
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
    if (p.fuse.D != nullptr) {
        if (vec == 4)      { name = launch_rows_drop<4>(p, s); if (m.n_long) launch_long_drop<4>(p, s); }
        else if (vec == 2) { name = launch_rows_drop<2>(p, s); if (m.n_long) launch_long_drop<2>(p, s); }
        else               { name = launch_rows_drop<1>(p, s); if (m.n_long) launch_long_drop<1>(p, s); }
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

int gnx_spmm_dropped(gnx_graph_t g, const float *d_D, float dropout_p, uint64_t seed, uint64_t stream_id, int transposed,
                     const float *d_X, int64_t ldx, int64_t C, const float *d_H0, int64_t ldh0, float beta, float alpha, int act,
                     float *d_out, int64_t ldo, void *stream) {
    int rc = check_common("gnx_spmm_dropped", g, d_X, ldx, C, d_H0, ldh0, d_out, ldo);
    if (rc != GNX_OK) return rc;
    GNX_CHECK_ARG(act == GNX_ACT_NONE || act == GNX_ACT_RELU, "gnx_spmm_dropped: invalid activation %d", act);
    GNX_CHECK_ARG(d_D != nullptr, "gnx_spmm_dropped: NULL degree scales");
    GNX_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "dropout rate %g outside [0, 1)", (double)dropout_p);
    GNX_CHECK_ARG(g->a.n_rows == g->a.n_cols || g->blk_col_gid != nullptr, "gnx_spmm_dropped: needs a square graph or a vertex block (gnx_graph_set_block)");
    if (g->has_dups) {   // per-entry dropout of duplicated COO entries needs the entry lists: use gnx_graph_normalize + gnx_spmm
        set_error("gnx_spmm_dropped: the graph holds duplicate COO entries");
        return GNX_ERR_UNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    if (transposed) {
        rc = ensure_transpose(g, s);
        if (rc != GNX_OK) return rc;
    }
    SpmmArgs p{};
    p.vals = transposed ? g->t_raw : g->raw_vals;
    p.X = d_X; p.ldx = ldx; p.H0 = d_H0; p.ldh0 = ldh0; p.beta = beta; p.alpha = alpha; p.act = act;
    p.out = d_out; p.ldo = ldo; p.C = (int)C;
    p.fuse.D = d_D; p.fuse.seed = seed; p.fuse.stream = stream_id; p.fuse.offset = g->stream_offset;
    p.fuse.thr = (uint32_t)((double)dropout_p * 16777216.0);
    p.fuse.scale = 1.0f / (1.0f - dropout_p);
    p.fuse.transposed = transposed ? 1 : 0;
    p.fuse.row0_key = g->blk_row0_global; p.fuse.row0_D = g->blk_row0_buf; p.fuse.gid = g->blk_col_gid;
    return launch_spmm(g, transposed ? g->t : g->a, p, s);
}

int gnx_spmm_dropped_chained(gnx_graph_t g, const float *d_D, float dropout_p, uint64_t seed, uint64_t stream_id, int x_prescaled,
                             const float *d_D_next, const float *d_X, int64_t ldx, int64_t C, const float *d_H0, int64_t ldh0, float beta,
                             float alpha, int act, float *d_out, int64_t ldo, void *stream) {
    int rc = check_common("gnx_spmm_dropped_chained", g, d_X, ldx, C, d_H0, ldh0, d_out, ldo);
    if (rc != GNX_OK) return rc;
    GNX_CHECK_ARG((act & ~GNX_ACT_SKIP_EMPTY) == GNX_ACT_NONE || (act & ~GNX_ACT_SKIP_EMPTY) == GNX_ACT_RELU,
                  "gnx_spmm_dropped_chained: invalid activation %d", act);
    if (!g->a.empty_rows_unreferenced) act &= ~GNX_ACT_SKIP_EMPTY;       // honoured only when nobody gathers the rows it would leave untouched
    GNX_CHECK_ARG(d_D != nullptr, "gnx_spmm_dropped_chained: NULL degree scales");
    GNX_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "dropout rate %g outside [0, 1)", (double)dropout_p);
    GNX_CHECK_ARG(g->a.n_rows == g->a.n_cols || g->blk_col_gid != nullptr, "gnx_spmm_dropped_chained: needs a square graph or a vertex block");
    if (g->has_dups) {
        set_error("gnx_spmm_dropped_chained: the graph holds duplicate COO entries");
        return GNX_ERR_UNSUPPORTED;
    }
    SpmmArgs p{};
    p.vals = g->raw_vals;
    p.X = d_X; p.ldx = ldx; p.H0 = d_H0; p.ldh0 = ldh0; p.beta = beta; p.alpha = alpha; p.act = act;
    p.out = d_out; p.ldo = ldo; p.C = (int)C;
    p.out_scale = d_D_next ? d_D_next + g->blk_row0_buf : nullptr;
    p.fuse.D = d_D; p.fuse.seed = seed; p.fuse.stream = stream_id; p.fuse.offset = g->stream_offset;
    p.fuse.thr = (uint32_t)((double)dropout_p * 16777216.0);
    p.fuse.scale = 1.0f / (1.0f - dropout_p);
    p.fuse.transposed = 0;
    p.fuse.col_prescaled = x_prescaled ? 1 : 0;
    p.fuse.row0_key = g->blk_row0_global; p.fuse.row0_D = g->blk_row0_buf; p.fuse.gid = g->blk_col_gid;
    return launch_spmm(g, g->a, p, (hipStream_t)stream);
}

int gnx_spmm_dropped_back(gnx_graph_t g, const float *d_D, float dropout_p, uint64_t seed, uint64_t stream_id, int x_prescaled,
                          const float *d_D_next, const float *d_X, int64_t ldx, int64_t C, const float *d_S_in, int64_t lds_in,
                          float s_alpha, float s_beta, float *d_S_out, int64_t lds_out, float y_beta, float *d_Y_out, int64_t ldy,
                          int act, void *stream) {
    int rc = check_common("gnx_spmm_dropped_back", g, d_X, ldx, C, d_S_in, lds_in, d_S_out, lds_out);
    if (rc != GNX_OK) return rc;
    GNX_CHECK_ARG(act == GNX_ACT_NONE || act == GNX_ACT_SKIP_EMPTY, "gnx_spmm_dropped_back: act must be GNX_ACT_NONE or GNX_ACT_SKIP_EMPTY");
    GNX_CHECK_ARG(act == GNX_ACT_NONE || (const void *)d_S_in == (const void *)d_S_out,
                  "gnx_spmm_dropped_back: GNX_ACT_SKIP_EMPTY needs the sum updated in place");
    if (rc != GNX_OK) return rc;
    GNX_CHECK_ARG(d_D != nullptr && d_S_in != nullptr, "gnx_spmm_dropped_back: NULL degree scales / running sum");
    GNX_CHECK_ARG(d_Y_out == nullptr || (ldy >= C && (const void *)d_Y_out != (const void *)d_X && (const void *)d_Y_out != (const void *)d_S_out
                                         && (const void *)d_Y_out != (const void *)d_S_in),
                  "gnx_spmm_dropped_back: the pre-scaled output needs a buffer of its own");
    GNX_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "dropout rate %g outside [0, 1)", (double)dropout_p);
    GNX_CHECK_ARG(g->a.n_rows == g->a.n_cols && g->blk_col_gid == nullptr, "gnx_spmm_dropped_back: needs a square stand-alone graph");
    if (g->has_dups) {
        set_error("gnx_spmm_dropped_back: the graph holds duplicate COO entries");
        return GNX_ERR_UNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    rc = ensure_transpose(g, s);
    if (rc != GNX_OK) return rc;
    if (!g->t.empty_rows_unreferenced) act = GNX_ACT_NONE;               // honoured only when nobody gathers the rows it would leave untouched
    SpmmArgs p{};
    p.vals = g->t_raw;
    p.X = d_X; p.ldx = ldx; p.H0 = d_S_in; p.ldh0 = lds_in; p.beta = s_beta; p.alpha = s_alpha; p.act = act;
    p.out = d_S_out; p.ldo = lds_out; p.C = (int)C;
    p.out2 = d_Y_out; p.ldo2 = ldy; p.beta2 = y_beta; p.out2_scale = d_Y_out ? d_D_next : nullptr;
    p.fuse.D = d_D; p.fuse.seed = seed; p.fuse.stream = stream_id; p.fuse.offset = g->stream_offset;
    p.fuse.thr = (uint32_t)((double)dropout_p * 16777216.0);
    p.fuse.scale = 1.0f / (1.0f - dropout_p);
    p.fuse.transposed = 1;
    p.fuse.col_prescaled = x_prescaled ? 1 : 0;
    return launch_spmm(g, g->t, p, s);
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

int gnx_gcnii_step(gnx_graph_t g, const float *d_vals, const float *d_H, const float *d_H0, float a, int64_t C, const float *d_M,
                   int64_t ldm, int act, float *d_out, float *d_mixed, void *stream) {
    int rc = check_common("gnx_gcnii_step", g, d_H, C, C, d_H0, C, d_out, C);
    if (rc != GNX_OK) return rc;
    GNX_CHECK_ARG(act == GNX_ACT_NONE || act == GNX_ACT_RELU, "gnx_gcnii_step: invalid activation %d", act);
    GNX_CHECK_ARG(g->a.n_rows == g->a.n_cols, "gnx_gcnii_step: needs a square graph");
    GNX_CHECK_ARG(d_H0 != nullptr && d_M != nullptr && ldm >= C, "gnx_gcnii_step: NULL H0 / M or ldm < C");
    hipStream_t s = (hipStream_t)stream;
    const Csr &m = g->a;
    const float beta = (float)(1.0 - (double)a);
    GNX_CHECK_ARG(d_mixed == nullptr || (d_mixed != d_out && d_mixed != d_H && d_mixed != d_H0), "gnx_gcnii_step: d_mixed must be a buffer of its own");
    // C = 128 fits the kernel (135 KB of LDS: one block of eight waves per CU) and was measured: 19.2 ms against 11.8 ms for the two
    // launches on the config-4 graph -- eight waves per CU cannot keep the gathers fed -- so it takes the two-launch form
    const bool fusable = (C == 16 || C == 32 || C == 64) && aligned(d_H, 16) && aligned(d_H0, 16) && aligned(d_out, 16) && aligned(d_mixed, 16);
    if (!fusable) {   // other widths: the fused SpMM+mix into d_mixed, then the transform on the matrix cores
        GNX_CHECK_ARG(d_mixed != nullptr, "gnx_gcnii_step: width %lld needs d_mixed [n, C] (the mixed rows go through memory)", (long long)C);
        rc = gnx_spmm(g, d_vals, nullptr, d_H, C, C, d_H0, C, beta, a, GNX_ACT_NONE, d_mixed, C, stream);
        if (rc != GNX_OK) return rc;
        g->last_kernel = "spmm+dense_mfma";
        return dense_rows(d_mixed, C, m.n_rows, C, d_M, ldm, C, nullptr, act, nullptr, nullptr, d_out, C, s);
    }
    if (m.n_rows == 0) return GNX_OK;
    SpmmArgs p{};
    p.vals = d_vals ? d_vals : g->raw_vals;
    p.X = d_H; p.ldx = C; p.H0 = d_H0; p.ldh0 = C; p.beta = beta; p.alpha = a; p.act = act; p.out = d_out; p.ldo = C; p.C = (int)C;
    p.rowptr = m.rowptr; p.colidx = m.colidx; p.n_rows = m.n_rows; p.n_nonempty = m.n_nonempty; p.row_order = m.row_order;
    p.long_rows = m.long_rows; p.long_chunk_ptr = m.long_chunk_ptr; p.chunk_long = m.chunk_long; p.chunk_order = m.chunk_order;
    p.n_long = m.n_long; p.n_chunks = m.n_chunks; p.long_row = m.long_row; p.long_chunk = m.long_chunk;
    const unsigned grid = blocks_for(blocks_for(m.n_rows, 16), 8);
    if (C == 64)      hipLaunchKernelGGL((k_spmm_gcnii<4, 4, 8>), dim3(grid), dim3(512), 0, s, p, d_M, ldm, d_mixed);
    else if (C == 32) hipLaunchKernelGGL((k_spmm_gcnii<2, 4, 8>), dim3(grid), dim3(512), 0, s, p, d_M, ldm, d_mixed);
    else              hipLaunchKernelGGL((k_spmm_gcnii<1, 4, 8>), dim3(grid), dim3(512), 0, s, p, d_M, ldm, d_mixed);
    g->last_kernel = "spmm_gcnii_mfma";
    if (m.n_long > 0) {   // hub rows: chunked partial sums -> mixed rows (into d_mixed when kept, else in place) -> transform of those rows alone
        rc = ensure_partial(g, (size_t)m.n_chunks * (size_t)C * sizeof(float), s);
        if (rc != GNX_OK) return rc;
        p.partial = g->partial;
        p.act = GNX_ACT_NONE;
        float *rows_at = d_mixed ? d_mixed : d_out;
        p.out = rows_at;
        launch_long<4>(p, s);
        rc = dense_rows(rows_at, C, m.n_long, C, d_M, ldm, C, nullptr, act, m.long_rows, m.long_rows, d_out, C, s);
        if (rc != GNX_OK) return rc;
    }
    GNX_HIP(hipGetLastError());
    return GNX_OK;
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

int gnx_stream_copy(const float *d_src, float *d_dst, int64_t n_floats, void *stream) {
    GNX_CHECK_ARG(n_floats >= 0 && n_floats % 4 == 0, "gnx_stream_copy: the length must be a multiple of 4 floats");
    if (n_floats == 0) return GNX_OK;
    GNX_CHECK_ARG(d_src && d_dst && aligned(d_src, 16) && aligned(d_dst, 16), "gnx_stream_copy: NULL or unaligned pointer");
    hipLaunchKernelGGL((k_stream<4, true>), dim3(256 * 4), dim3(1024), 0, (hipStream_t)stream, (const f32x4 *)d_src, (f32x4 *)d_dst,
                       n_floats / 4, (float *)nullptr);
    GNX_HIP(hipGetLastError());
    return GNX_OK;
}

int gnx_linear_combination(int k, const float *const *d_src, const float *coef, int64_t n, float *d_out, void *stream) {
    GNX_CHECK_ARG(k >= 1 && k <= LINCOMB_MAX, "gnx_linear_combination: 1 to %d terms, got %d", LINCOMB_MAX, k);
    GNX_CHECK_ARG(n >= 0 && d_src != nullptr && coef != nullptr, "gnx_linear_combination: bad arguments");
    if (n == 0) return GNX_OK;
    GNX_CHECK_ARG(d_out != nullptr && aligned(d_out, 16), "gnx_linear_combination: NULL or unaligned output");
    LinComb a{};
    a.k = k;
    for (int j = 0; j < k; ++j) {
        GNX_CHECK_ARG(d_src[j] != nullptr && aligned(d_src[j], 16), "gnx_linear_combination: term %d is NULL or unaligned", j);
        a.src[j] = d_src[j]; a.coef[j] = coef[j];
    }
    const int64_t n4 = n / 4;
    const unsigned nb = (unsigned)std::min<int64_t>(std::max<int64_t>((n4 + 255) / 256, 1), 256 * 16);
    hipLaunchKernelGGL(k_lincomb, dim3(nb), dim3(256), 0, (hipStream_t)stream, a, n4, n, d_out);
    GNX_HIP(hipGetLastError());
    return GNX_OK;
}

int gnx_stream_read(const float *d_src, int64_t n_floats, float *d_sink64, void *stream) {
    GNX_CHECK_ARG(n_floats >= 0 && n_floats % 4 == 0, "gnx_stream_read: the length must be a multiple of 4 floats");
    if (n_floats == 0) return GNX_OK;
    GNX_CHECK_ARG(d_src && d_sink64 && aligned(d_src, 16), "gnx_stream_read: NULL or unaligned pointer");
    hipLaunchKernelGGL((k_stream<8, false>), dim3(256), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)d_src, (f32x4 *)nullptr,
                       n_floats / 4, d_sink64);
    GNX_HIP(hipGetLastError());
    return GNX_OK;
}

__global__ void k_probe_xcd(int32_t *__restrict__ out) {
    if (threadIdx.x == 0) {
        uint32_t id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        out[blockIdx.x] = (int32_t)(id & 0xf);
    }
}

int gnx_probe_block_xcd(int64_t n_blocks, int32_t *d_xcd_out, void *stream) {
    GNX_CHECK_ARG(n_blocks >= 0 && n_blocks < ((int64_t)1 << 24), "gnx_probe_block_xcd: bad block count");
    if (n_blocks == 0) return GNX_OK;
    GNX_CHECK_ARG(d_xcd_out != nullptr, "gnx_probe_block_xcd: NULL output");
    hipLaunchKernelGGL(k_probe_xcd, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream, d_xcd_out);
    GNX_HIP(hipGetLastError());
    return GNX_OK;
}

int gnx_gather_rows(const float *d_X, int64_t ldx, const int64_t *d_idx, int64_t n_idx, int64_t C, float *d_out, int64_t ldo,
                    void *stream) {
    GNX_CHECK_ARG(n_idx >= 0 && C >= 1 && ldx >= C && ldo >= C, "gnx_gather_rows: bad sizes");
    GNX_CHECK_ARG(n_idx < ((int64_t)1 << 26), "gnx_gather_rows: at most 2^26 - 1 rows per call (one wave per row, 2^32 work-items per launch)");
    if (n_idx == 0) return GNX_OK;
    GNX_CHECK_ARG(d_X && d_idx && d_out, "gnx_gather_rows: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    const bool v4 = C % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && aligned(d_X, 16) && aligned(d_out, 16);
    if (v4) hipLaunchKernelGGL(k_gather_rows<4>, dim3(blocks_for(n_idx, 4)), dim3(256), 0, s, d_X, ldx, d_idx, n_idx, (int)C, d_out, ldo);
    else    hipLaunchKernelGGL(k_gather_rows<1>, dim3(blocks_for(n_idx, 4)), dim3(256), 0, s, d_X, ldx, d_idx, n_idx, (int)C, d_out, ldo);
    GNX_HIP(hipGetLastError());
    return GNX_OK;
}

}  // extern "C"

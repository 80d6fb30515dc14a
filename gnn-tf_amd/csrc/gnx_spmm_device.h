// Device-side pieces every SpMM translation unit shares (gnx_spmm.hip: the eval kernels and their launchers; gnx_spmm_train.hip: the
// training kernels; gnx_gcnii.hip: the GCNII layer): vector loads / stores, the wave-wide accumulate, the fused epilogue of
// filter.py:20-22, the XCD block map, the long rows' second pass, and the launch plumbing.  Internal linkage throughout: every
// translation unit gets its own copies (all of it is templates / forceinline device code / small host helpers).
#pragma once
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

// ---- long rows, second pass: the chunks' partial sums added in chunk order + the epilogue (shared by the eval and training paths) ----
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

inline unsigned blocks_for(int64_t n, int per_block) { return (unsigned)((n + per_block - 1) / per_block); }

inline bool aligned(const void *p, size_t a) { return p == nullptr || ((uintptr_t)p % a) == 0; }

// widest vector width every row start allows
[[maybe_unused]] int pick_vec(const SpmmArgs &p) {
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

[[maybe_unused]] int check_common(const char *fn, gnx_graph *g, const float *X, int64_t ldx, int64_t C, const float *H0, int64_t ldh0,
                 float *out, int64_t ldo) {
    GNX_CHECK_ARG(g != nullptr, "%s: NULL handle", fn);
    GNX_CHECK_ARG(C >= 1 && C <= (1 << 20), "%s: feature width %lld not in [1, 2^20]", fn, (long long)C);
    GNX_CHECK_ARG(X != nullptr && out != nullptr, "%s: NULL X/out", fn);
    GNX_CHECK_ARG(ldx >= C && ldo >= C && (H0 == nullptr || ldh0 >= C || ldh0 == 0), "%s: leading dimension smaller than C", fn);
    GNX_CHECK_ARG((const void *)X != (const void *)out, "%s: out must not alias X", fn);
    return GNX_OK;
}

}  // namespace

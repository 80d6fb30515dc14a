// The dense ends of the path on gfx950 matrix cores: the pre-MLP transform and the task head.
//
//   gnx_dense      out = act(X . W + b)                      reference gnntf/core/nn/layers.py:135-136 (Dense),
//                                                            gnntf/core/gnn/architectures/gcn.py:89 (the transform of GCNLayer)
//   gnx_node_ce    mean_i CE(log_softmax(logits[nodes_i]))   reference gnntf/core/gnn/graph_predictor.py:19-25
//   gnx_node_argmax  argmax(logits[nodes_i])                 reference gnntf/core/gnn/graph_predictor.py:16-17, 27-31
//
// gnx_dense is a tall-and-skinny GEMM (N rows in the millions, F and O in the tens to hundreds): float32 in, float32
// accumulate on v_mfma_f32_16x16x4_f32 (bit-for-bit a k-ordered fmaf chain, no reduced precision).  A 256-thread block owns
// 64 rows; each of its 4 waves keeps a 16-row x O accumulator strip in registers.  X is read ONCE, straight from HBM into
// the A operand (16 bytes per lane; the k index inside a 16-wide step is permuted so that a lane's four consecutive floats
// feed four MFMAs), W streams through LDS in K chunks shared by the block (row stride = 4 mod 32 banks: the four k-groups
// of a wave read disjoint banks).  Arithmetic intensity is O/2 flop per byte of X: HBM-bound up to O = 32, MFMA-bound beyond.
// Tall inputs (n >= 16K rows) take one of two persistent kernels instead: k_dense_wreg (W in registers, shapes up to 256 x 64) or
// k_dense_ring (W in LDS); both stream X through per-wave LDS-DMA rings and add the k terms in the same order as this kernel.
// gnx_dense_wgrad: k_wgrad_acc (accumulators stationary, tall inputs) or k_wgrad_mfma.
// Tried and dropped (round 2): a persistent W-resident variant (W once in LDS, the whole K extent of a 16-row tile in
// registers, next tile prefetched, no barrier in the loop) -- 5.1 ms vs 4.2 ms at 10M x 256 -> 64: with two waves per SIMD the
// 64-byte-per-row A loads no longer hide (a 256 -> 7 product ran at 3.2 TB/s); the chunked kernel's eight blocks per CU do.
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

#include "gnx_internal.h"
#include <mutex>

using namespace gnx;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct DenseArgs {
    const float *X; int64_t ldx; int64_t n; int F;
    const float *W; int64_t ldw; int O;
    const float *bias;            // [O] or null
    int act;
    float *out; int64_t ldo;
    const int32_t *out_rows;      // optional: result row r goes to out[out_rows[r]]
    const int32_t *in_rows;       // optional: input row r is X[in_rows[r]]
    bool w_aligned;               // W rows start 16-byte aligned (ldw % 4 == 0, aligned base): float4 loads of W
    const float *zeros = nullptr; // 16 bytes of device zeros (k_dense_wreg with padded widths stages them for the columns past F)
};

template <int NT> struct DenseCfg {
    static constexpr int OP = NT * 16;                 // padded output width
    static constexpr int KC = NT <= 4 ? 64 : 32;       // W rows per LDS chunk
    static constexpr int STRIDE = OP + 4;              // OP is a multiple of 16; +4 makes row stride = 4 (mod 8): see header
};

constexpr int DENSE_WAVES = 8;       // waves (16-row strips) per block: 128 rows share one W chunk in LDS

template <int NT, bool ALIGNED>
__global__ __launch_bounds__(64 * DENSE_WAVES) void k_dense_mfma(const DenseArgs p) {
    using Cfg = DenseCfg<NT>;
    __shared__ float Ws[Cfg::KC * Cfg::STRIDE];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * (16 * DENSE_WAVES) + wave * 16;
    int64_t arow = row0 + c < p.n ? row0 + c : p.n - 1;                  // rows past the end read a valid row and are not stored
    if (p.in_rows) arow = p.in_rows[arow];
    const float *__restrict__ xrow = p.X + arow * p.ldx;
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // A values of one K chunk: 16 bytes per lane and 16-wide step, straight from HBM into the MFMA operand registers
    auto load_a = [&](int k0, float (&a)[Cfg::KC / 16][4]) {
#pragma unroll
        for (int T = 0; T < Cfg::KC / 16; ++T) {
            const int kb = k0 + 16 * T + 4 * g;
            if (ALIGNED && kb + 3 < p.F) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(xrow + kb);
                a[T][0] = v[0]; a[T][1] = v[1]; a[T][2] = v[2]; a[T][3] = v[3];
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) a[T][t] = kb + t < p.F ? xrow[kb + t] : 0.f;
            }
        }
    };
    float a[Cfg::KC / 16][4], a_next[Cfg::KC / 16][4];
    load_a(0, a);
    for (int k0 = 0; k0 < p.F; k0 += Cfg::KC) {
        __syncthreads();                                                   // the previous chunk has been consumed
        for (int idx = threadIdx.x; idx < Cfg::KC * (Cfg::OP / 4); idx += 64 * DENSE_WAVES) {      // 16 bytes of W per thread and step
            const int r = idx / (Cfg::OP / 4), cc = (idx % (Cfg::OP / 4)) * 4;
            const int k = k0 + r;
            f32x4 w = f32x4{0.f, 0.f, 0.f, 0.f};
            if (k < p.F) {
                const float *__restrict__ wr = p.W + (int64_t)k * p.ldw + cc;
                if (p.w_aligned && cc + 3 < p.O) w = *reinterpret_cast<const f32x4 *>(wr);
                else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) if (cc + t < p.O) w[t] = wr[t];
                }
            }
            *reinterpret_cast<f32x4 *>(Ws + r * Cfg::STRIDE + cc) = w;
        }
        if (k0 + Cfg::KC < p.F) load_a(k0 + Cfg::KC, a_next);              // the next chunk's rows are in flight under this chunk's MFMAs
        __syncthreads();
#pragma unroll
        for (int T = 0; T < Cfg::KC / 16; ++T) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float *__restrict__ wrow = Ws + (16 * T + 4 * g + t) * Cfg::STRIDE + c;   // B[k-slot g][col c] = W[k0 + 16T + 4g + t][16 nt + c]
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[T][t], wrow[16 * nt], acc[nt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int T = 0; T < Cfg::KC / 16; ++T)
#pragma unroll
            for (int t = 0; t < 4; ++t) a[T][t] = a_next[T][t];
    }
    // D layout: lane (c, g), register r -> row 4g + r, column 16 nt + c
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = 16 * nt + c;
        if (col >= p.O) continue;
        const float b = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = row0 + 4 * g + r;
            if (row >= p.n) continue;
            float v = acc[nt][r] + b;
            if (p.act == GNX_ACT_RELU) v = fmaxf(v, 0.f);
            const int64_t orow = p.out_rows ? (int64_t)p.out_rows[row] : row;
            p.out[orow * p.ldo + col] = v;
        }
    }
}

// ---- the same product with X through LDS in full lines (LDS-DMA ring), W resident in LDS --------------------------------------
// k_dense_mfma loads the A operand "fragment-shaped": one wave instruction fetches 64 bytes of each of 16 rows, so every 128-byte
// line of X is touched twice and the CU's L1 pipe does twice the line work of a row-contiguous read; its 32 A registers per lane
// also cap the occupancy.  Here every wave owns 16-row tiles and streams their K chunks (64 floats = two whole lines per row)
// into a private RING of LDS stages by LDS-DMA (global_load_lds_dwordx4: no VGPR destination, four 1-KiB wave instructions per
// stage, each writing four rows), keeps RING - 1 stages in flight behind a counted s_waitcnt vmcnt, and reads the A fragments
// back with ds_read_b128.  The LDS image is lane-linear (an LDS-DMA cannot pad or scatter), so the 16-byte pieces of a row are
// XOR-swizzled by the row index on the SOURCE address; the fragment reads of a wave are then conflict-free.  W ([F, O], at most
// 64 KB) is laid out once per block as [k][column group][c][4] so that a lane's four accumulator columns are ONE ds_read_b128,
// again conflict-free (row stride = a multiple of the 256-byte bank row).  The block is persistent (one per CU, WAVES waves, no
// barrier inside the loop: a wave only ever reads what it staged itself), walks tiles of 16 rows, and adds the k terms in the same
// order as k_dense_mfma: same bits.
constexpr int RING_BK = 64;            // floats of one row per stage

template <int NT, int WAVES, int RING>
__global__ __launch_bounds__(64 * WAVES) void k_dense_ring(const DenseArgs p, int64_t n_tiles) {
    static_assert(NT % 4 == 0, "the W image groups the accumulator columns in fours");
    constexpr int NQ = NT / 4;                                    // column groups of 4 x 16 outputs
    constexpr int STAGE = 16 * RING_BK;                           // floats per stage: 16 rows x 64
    extern __shared__ float lds[];                                // ONE array: [W image: F * NT * 16 floats][WAVES][RING][STAGE]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int w_floats = p.F * NT * 16;
    float *__restrict__ Wl = lds;
    float *__restrict__ ring = lds + w_floats + wave * (RING * STAGE);
    // W image: Wl[(k * NQ + q) * 64 + cc * 4 + j] = W[k][16 (4 q + j) + cc]
    for (int idx = threadIdx.x; idx < p.F * NT * 16; idx += 64 * WAVES) {
        const int k = idx / (NT * 16), col = idx % (NT * 16);      // col = 16 nt + cc: coalesced reads of W's row
        const int nt = col >> 4, cc = col & 15;
        Wl[(k * NQ + (nt >> 2)) * 64 + cc * 4 + (nt & 3)] = col < p.O ? p.W[(int64_t)k * p.ldw + col] : 0.f;
    }
    float bias[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bias[nt] = (p.bias && 16 * nt + c < p.O) ? p.bias[16 * nt + c] : 0.f;
    __syncthreads();                                              // the only barrier: W is in place

    const int kchunks = p.F / RING_BK;
    const int64_t tile_stride = (int64_t)gridDim.x * WAVES;
    const int64_t first = (int64_t)blockIdx.x * WAVES + wave;
    const int64_t my_tiles = first < n_tiles ? (n_tiles - first + tile_stride - 1) / tile_stride : 0;
    const int64_t steps = my_tiles * kchunks;                     // (tile, K chunk) pairs, walked in order

    // Staging runs RING - 1 steps ahead of the arithmetic: (pf_tile, pf_kc) is the next step to stage, counted up without divisions.
    // One stage = four LDS-DMA instructions, lane l -> row 4 i + l / 16, 16-byte slot l % 16 of that row, filled with piece slot ^ row.
    int64_t pf_tile = first, pf_step = 0;
    int pf_kc = 0;
    auto issue_next = [&]() {
        float *dst = ring + (pf_step % RING) * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 4 * i + (lane >> 4);
            int64_t row = pf_tile * 16 + r;
            row = row < p.n ? row : p.n - 1;                      // rows past the end read a valid row and are not stored
            const float *src = p.X + row * p.ldx + pf_kc * RING_BK + 4 * ((lane & 15) ^ r);
            __builtin_amdgcn_global_load_lds(src, dst + i * 256, 16, 0, 0);
        }
        ++pf_step;
        if (++pf_kc == kchunks) { pf_kc = 0; pf_tile += tile_stride; }
    };
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < RING - 1 && s0 < steps; ++s0) issue_next();
    int64_t step = 0;
    for (int64_t tile = first; tile < n_tiles; tile += tile_stride) {
        for (int kc = 0; kc < kchunks; ++kc, ++step) {
            // the reads of stage step - 1 (the slot that stage step + RING - 1 overwrites) were consumed by its MFMAs: drain them, then restage
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (pf_step < steps) issue_next();
            // all but the youngest 4 (RING - 1) vector-memory operations are done => stage `step` has landed (stores issued in between only
            // make the wait stricter); near the end fewer stages are in flight behind it
            const int64_t behind = steps - 1 - step;
            if (behind >= RING - 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * (RING - 1)) : "memory");
            else if (RING > 2 && behind == RING - 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * (RING > 2 ? RING - 2 : 0)) : "memory");
            else if (RING > 3 && behind == RING - 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * (RING > 3 ? RING - 3 : 0)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const float *__restrict__ A = ring + (step % RING) * STAGE;
            if constexpr (NQ <= 2) {
                // the stage's A fragments up front (X[row c][16 T + 4 g .. + 3]), the B fragments one k step ahead of the MFMAs that use
                // them: the wave hides its own LDS latency instead of waiting for every fragment it has just asked for
                f32x4 a4[RING_BK / 16];
#pragma unroll
                for (int T = 0; T < RING_BK / 16; ++T) a4[T] = *reinterpret_cast<const f32x4 *>(A + c * RING_BK + 4 * ((4 * T + g) ^ c));
                const float *__restrict__ Wk = Wl + ((kc * RING_BK + 4 * g) * NQ) * 64 + c * 4;      // fragment of k = kc 64 + 4 g (+ 16 T + t)
                f32x4 b[2][NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) b[0][q] = *reinterpret_cast<const f32x4 *>(Wk + q * 64);
#pragma unroll
                for (int st = 0; st < RING_BK / 4; ++st) {                    // st = 4 T + t
                    const int T = st >> 2, t = st & 3;
                    if (st + 1 < RING_BK / 4) {
                        const int T1 = (st + 1) >> 2, t1 = (st + 1) & 3;
#pragma unroll
                        for (int q = 0; q < NQ; ++q) b[(st + 1) & 1][q] = *reinterpret_cast<const f32x4 *>(Wk + ((16 * T1 + t1) * NQ + q) * 64);
                    }
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[4 * q + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[T][t], b[st & 1][q][j], acc[4 * q + j], 0, 0, 0);
                }
                // pin that order for the scheduler (it would otherwise sink every read next to its first use and wait for all of them):
                // the A fragments and the first B fragment, then { next B fragment, this step's MFMAs } sixteen times
                __builtin_amdgcn_sched_group_barrier(0x100, RING_BK / 16 + NQ, 0);
#pragma unroll
                for (int st = 0; st < RING_BK / 4; ++st) {
                    if (st + 1 < RING_BK / 4) __builtin_amdgcn_sched_group_barrier(0x100, NQ, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4 * NQ, 0);
                }
            } else {
                // wide outputs (O = 192 / 256): sixteen accumulator tiles leave no room for fragments in flight (measured: +5 % with them)
#pragma unroll
                for (int T = 0; T < RING_BK / 16; ++T) {
                    const f32x4 a4 = *reinterpret_cast<const f32x4 *>(A + c * RING_BK + 4 * ((4 * T + g) ^ c));
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int k = kc * RING_BK + 16 * T + 4 * g + t;
#pragma unroll
                        for (int q = 0; q < NQ; ++q) {
                            const f32x4 b4 = *reinterpret_cast<const f32x4 *>(Wl + (k * NQ + q) * 64 + c * 4);
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                acc[4 * q + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[t], b4[j], acc[4 * q + j], 0, 0, 0);
                        }
                    }
                }
            }
        }
        // the tile is complete.  D layout: lane (c, g), register r -> row 4 g + r, column 16 nt + c
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = 16 * nt + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = tile * 16 + 4 * g + r;
                float v = acc[nt][r] + bias[nt];
                if (p.act == GNX_ACT_RELU) v = fmaxf(v, 0.f);
                if (col < p.O && row < p.n) p.out[row * p.ldo + col] = v;
            }
            acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
}

// the ring kernel takes: whole float4 rows of X, F a multiple of 64, W image of at most 64 KB, accumulator columns in groups of four
bool ring_eligible(const DenseArgs &p, bool x_aligned, int nt) {
    return x_aligned && p.in_rows == nullptr && p.out_rows == nullptr && p.F % RING_BK == 0 && nt % 4 == 0 && (int64_t)p.F * nt * 16 * 4 <= (64 << 10) &&
           p.n >= 16 * 1024;
}

template <int NT, int WAVES, int RING>
int launch_ring_as(const DenseArgs &p, hipStream_t s) {
    const size_t lds_bytes = ((size_t)p.F * NT * 16 + (size_t)WAVES * RING * 16 * RING_BK) * sizeof(float);
    static PerDeviceOnce configured;
    const int attr_dev = PerDeviceOnce::device();
    if (configured.need(attr_dev)) {
        GNX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dense_ring<NT, WAVES, RING>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10));
        configured.set(attr_dev);
    }
    const int64_t n_tiles = (p.n + 15) / 16;
    int cus = 256;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const unsigned grid = (unsigned)std::min<int64_t>((n_tiles + WAVES - 1) / WAVES, cus);
    hipLaunchKernelGGL((k_dense_ring<NT, WAVES, RING>), dim3(grid), dim3(64 * WAVES), lds_bytes, s, p, n_tiles);
    return GNX_OK;
}

// Waves per block: as many 2-stage rings (8 KB each) as fit beside the W image in the CU's 160 KB of LDS, up to 16 -- measured at
// 10M x 256 -> 64: 4 waves x 4 stages 4.14 ms, 8 x 2 3.54, 8 x 3 3.61, 12 x 2 3.41 (k_dense_mfma: 3.59); the matrix pipe wants
// three waves per SIMD more than it wants a deeper ring.
template <int NT>
int launch_ring(const DenseArgs &p, hipStream_t s) {
    const size_t w_bytes = (size_t)p.F * NT * 16 * sizeof(float);
    const size_t rings = ((160u << 10) - w_bytes) / (2 * 16 * RING_BK * sizeof(float));
    if (rings >= 16) return launch_ring_as<NT, 16, 2>(p, s);
    if (rings >= 12) return launch_ring_as<NT, 12, 2>(p, s);
    return launch_ring_as<NT, 8, 2>(p, s);
}

// ---- W in REGISTERS, X through a deep LDS-DMA ring ---------------------------------------------------------------------------------
// When the whole W fits the register file -- (F / 4) x NT fragments per lane, 256 for the pre-MLP's 256 -> 64 -- a wave keeps it there
// for the life of the (persistent) block: no LDS read per MFMA at all, and the CU's LDS is ALL staging ring.  One wave per SIMD (512
// registers each), so the matrix pipe is fed by a single straight-line instruction stream, and everything that is not an MFMA has to
// sit in the issue slots between two of them:
//   * W is the MFMA's A operand and the X fragment its B operand (out^T = W^T X^T): a lane then holds FOUR CONSECUTIVE output columns
//     of one row, so a tile is stored with four 16-byte stores instead of sixteen 4-byte ones;
//   * two accumulator sets: the stores of tile t are in the same basic block as the first MFMAs of the wave's next tile (no branch:
//     only the wave's LAST tile, the one that can be ragged, takes the guarded path);
//   * staging addresses are 32-bit row x 32-bit pitch (one v_mad_u64_u32 per load); the stage's K offset and LDS slot are compile-time;
//   * staging never stops (past the wave's last tile it re-stages the last row tile, which nobody reads), so "all but the youngest
//     4 (RING - 2) vector-memory operations" is the wait for every stage (stores in between only make it stricter).
// Same k order as the other two kernels: same bits.  Measured (10M x 256 -> 64, profiles/NOTES.md): the ring depth does not matter from
// 4 stages up and the staging waits are never taken -- what the launch costs beyond the MFMAs is issue slots and the clock.
// PAD: F < 64 KS and / or O < 16 NT (both multiples of 4): the columns of X past F are staged from a block of zeros (times the zero rows
// W gets there: exact, whatever X holds), the columns past O are neither loaded from W nor stored.
template <int NT, int KS, int RING, bool RELU, bool PAD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_dense_wreg(const DenseArgs p, int64_t n_tiles64) {
    constexpr int STAGE = 16 * RING_BK;
    extern __shared__ float lds[];                                // [4 waves][RING][STAGE]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;
    float *__restrict__ ring = lds + wave * (RING * STAGE);
    // W fragments: wreg[kc][T][t][nt] = W[64 kc + 16 T + 4 g + t][16 nt + c]
    float wreg[KS][4][4][NT];
#pragma unroll
    for (int kc = 0; kc < KS; ++kc)
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                {
                    const int k = 64 * kc + 16 * T + 4 * g + t, col = 16 * nt + c;
                    wreg[kc][T][t][nt] = (!PAD || (k < p.F && col < p.O)) ? p.W[(int64_t)k * p.ldw + col] : 0.f;
                    asm volatile("" : "+a"(wreg[kc][T][t][nt]));      // W lives in the accumulation half of the register file; the MFMAs read it there
                }
    f32x4 bias[NT];                                               // columns 16 nt + 4 g .. + 3
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[nt][r] = (p.bias && (!PAD || 16 * nt + 4 * g + r < p.O)) ? p.bias[16 * nt + 4 * g + r] : 0.f;

    const uint32_t n_tiles = (uint32_t)n_tiles64, n_rows = (uint32_t)p.n;
    const uint32_t f_pieces = (uint32_t)p.F / 4;
    const uint32_t tile_stride = gridDim.x * 4, first = blockIdx.x * 4 + wave;
    const uint32_t x_pitch = (uint32_t)p.ldx * 4u, o_pitch = (uint32_t)p.ldo * 4u;      // bytes
    const char *__restrict__ Xb = reinterpret_cast<const char *>(p.X);
    char *__restrict__ Ob = reinterpret_cast<char *>(p.out);
    uint32_t piece[4];                                            // byte offset of this lane's 16-byte piece inside the stage's 256 bytes of row 4 i + g
#pragma unroll
    for (int i = 0; i < 4; ++i) piece[i] = 16u * (uint32_t)(c ^ (4 * i + g));
    uint32_t pf_tile = first;
    auto issue = [&](int pf_kc, int slot) {                       // both compile-time at every call site
        const uint32_t t16 = (pf_tile < n_tiles ? pf_tile : n_tiles - 1) * 16u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t row = t16 + 4 * i + g;
            row = row < n_rows ? row : n_rows - 1;                // rows past the end read a valid row and are not stored
            const char *src = Xb + (uint64_t)row * x_pitch + (uint32_t)(pf_kc * RING_BK * 4) + piece[i];
            if constexpr (PAD) {
                if ((uint32_t)(pf_kc * (RING_BK / 4)) + piece[i] / 16u >= f_pieces) src = reinterpret_cast<const char *>(p.zeros);
            }
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float *>(src), ring + slot * STAGE + i * 256, 16, 0, 0);
        }
        if (pf_kc == KS - 1) pf_tile += tile_stride;
    };
    auto fragments = [&](int slot, f32x4 (&a)[4]) {               // X[row c][16 T + 4 g .. + 3] of the stage in `slot`
        const float *__restrict__ A = ring + slot * STAGE;
#pragma unroll
        for (int T = 0; T < 4; ++T) a[T] = *reinterpret_cast<const f32x4 *>(A + c * RING_BK + 4 * ((4 * T + g) ^ c));
    };
    // D layout with W as the A operand: lane (c, g), register r -> row c, column 16 nt + 4 g + r
    auto finish = [&](const f32x4 &acc, int nt) {
        f32x4 v = acc + bias[nt];
        if constexpr (RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        return v;
    };
    auto store_tile = [&](f32x4 (&acc)[NT], uint32_t tile) {      // a full tile: no guards
        char *o = Ob + (uint64_t)(tile * 16u + c) * o_pitch + 16u * g;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            if (!PAD || 16 * nt + 4 * g < p.O) *reinterpret_cast<f32x4 *>(o + 64 * nt) = finish(acc[nt], nt);
    };
    auto store_last = [&](f32x4 (&acc)[NT], uint32_t tile) {      // the wave's last tile: may be the ragged one
        const uint32_t row = tile * 16u + c;
        if (row < n_rows) {
            char *o = Ob + (uint64_t)row * o_pitch + 16u * g;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                if (!PAD || 16 * nt + 4 * g < p.O) *reinterpret_cast<f32x4 *>(o + 64 * nt) = finish(acc[nt], nt);
        }
    };
    f32x4 afrag[2][4];                                            // the X fragments of the stage being multiplied and of the next one
    // one tile: KS stages; `slot0` is the ring slot of its first stage (compile-time).  With EPI, the stores of the wave's previous tile
    // sit among the first stage's MFMAs.
    auto tile_body = [&](auto epi, int slot0, f32x4 (&acc)[NT], f32x4 (&prev)[NT], uint32_t prev_tile) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < KS; ++kc) {
            const int slot = (slot0 + kc) % RING;
            f32x4 (&a_cur)[4] = afrag[kc & 1];
            f32x4 (&a_nxt)[4] = afrag[(kc + 1) & 1];
            // a_cur was read out of `slot`; the slot before it is free (its fragments were consumed by the previous stage's MFMAs)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            issue((kc + RING - 1) % KS, (slot + RING - 1) % RING);
            if constexpr (decltype(epi)::value) {
                if (kc == 0) store_tile(prev, prev_tile);
            }
#pragma unroll
            for (int T = 0; T < 3; ++T)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[kc][T][t][nt], a_cur[T][t], acc[nt], 0, 0, 0);
            // the next stage: RING - 2 younger stages may still be in flight behind it
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * (RING - 2)) : "memory");
            fragments((slot + 1) % RING, a_nxt);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[kc][3][t][nt], a_cur[3][t], acc[nt], 0, 0, 0);
        }
    };
    static_assert(KS % 2 == 0, "the fragment registers alternate per stage: a tile must start on the same one every time");
    static_assert((2 * KS) % RING == 0, "two tiles must take a whole number of ring turns (their slots are compile-time)");
    if (first < n_tiles) {
#pragma unroll
        for (int s0 = 0; s0 < RING - 1; ++s0) issue(s0 % KS, s0);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * (RING - 2)) : "memory");          // stage 0 has landed
        fragments(0, afrag[0]);
        f32x4 accA[NT], accB[NT];
        uint32_t tile = first;
        constexpr int SLOT_B = KS % RING;                         // first slot of every second tile
        tile_body(std::false_type{}, 0, accA, accB, 0u);
        for (;;) {
            if (tile + tile_stride >= n_tiles) { store_last(accA, tile); break; }
            tile_body(std::true_type{}, SLOT_B, accB, accA, tile);
            tile += tile_stride;
            if (tile + tile_stride >= n_tiles) { store_last(accB, tile); break; }
            tile_body(std::true_type{}, 0, accA, accB, tile);
            tile += tile_stride;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // nothing of this wave may still be writing LDS when the block retires
}

// registers for W: 16 KS x NT fragments per lane (F <= 64 KS, O <= 16 NT), at most 256 of the wave's 512; widths multiples of 4
bool wreg_eligible(const DenseArgs &p, bool x_aligned) {
    const bool out_aligned = reinterpret_cast<uintptr_t>(p.out) % 16 == 0 && p.ldo % 4 == 0;
    return x_aligned && out_aligned && p.in_rows == nullptr && p.out_rows == nullptr && p.F % 4 == 0 && p.O % 4 == 0 && p.F > 64 && p.F <= 256 &&
           p.n >= 16 * 1024 && p.n < (1ll << 31) && p.ldx < (1ll << 30) && p.ldo < (1ll << 30);
}

// a block of zeros on the CURRENT device (one per device the process drives; never freed)
const float *device_zeros() {
    static std::mutex lock;
    static float *z[64] = {};
    const int dev = PerDeviceOnce::device();
    if (dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> hold(lock);
    if (!z[dev]) {
        float *q = nullptr;
        if (hipMalloc((void **)&q, 256) != hipSuccess) return nullptr;
        if (hipMemset(q, 0, 256) != hipSuccess) { (void)hipFree(q); return nullptr; }
        z[dev] = q;
    }
    return z[dev];
}

template <int NT, int KS, int RING, bool RELU, bool PAD>
int launch_wreg_as(const DenseArgs &p, hipStream_t s) {
    const size_t lds_bytes = (size_t)4 * RING * 16 * RING_BK * sizeof(float);
    static PerDeviceOnce configured;
    const int attr_dev = PerDeviceOnce::device();
    if (configured.need(attr_dev)) {
        GNX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dense_wreg<NT, KS, RING, RELU, PAD>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10));
        configured.set(attr_dev);
    }
    const int64_t n_tiles = (p.n + 15) / 16;
    int cus = 256;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const unsigned grid = (unsigned)std::min<int64_t>((n_tiles + 3) / 4, cus);
    hipLaunchKernelGGL((k_dense_wreg<NT, KS, RING, RELU, PAD>), dim3(grid), dim3(256), lds_bytes, s, p, n_tiles);
    return GNX_OK;
}

template <int NT, int KS, int RING>
int launch_wreg(const DenseArgs &p0, hipStream_t s) {
    DenseArgs p = p0;
    const bool pad = p.F != 64 * KS || p.O != 16 * NT;
    if (pad) {
        p.zeros = device_zeros();
        if (!p.zeros) return GNX_ERR_ALLOC;
        return p.act == GNX_ACT_RELU ? launch_wreg_as<NT, KS, RING, true, true>(p, s) : launch_wreg_as<NT, KS, RING, false, true>(p, s);
    }
    return p.act == GNX_ACT_RELU ? launch_wreg_as<NT, KS, RING, true, false>(p, s) : launch_wreg_as<NT, KS, RING, false, false>(p, s);
}

template <int NT>
void launch_dense(const DenseArgs &p, bool aligned, hipStream_t s) {
    const unsigned grid = (unsigned)((p.n + 16 * DENSE_WAVES - 1) / (16 * DENSE_WAVES));
    if (aligned) hipLaunchKernelGGL((k_dense_mfma<NT, true>), dim3(grid), dim3(64 * DENSE_WAVES), 0, s, p);
    else         hipLaunchKernelGGL((k_dense_mfma<NT, false>), dim3(grid), dim3(64 * DENSE_WAVES), 0, s, p);
}

// ---- weight gradient of the dense layer: dW[F, O] = X^T . G, a reduction over the N rows ------------------------------------
// M = F, N = O, K = rows.  grid.x = row slabs, grid.y = panels of 256 features, grid.z = panels of 16 NT outputs.  A block
// stages 32-row tiles of X (its 256 features) and G (its outputs) in LDS with coalesced 16-byte loads; wave w owns features
// [64 w, 64 w + 64) of the panel as 4 x NT accumulator tiles: A[m = feature][k = row] and B[k = row][n = output] fragments are
// read from LDS (row stride = 16 mod 32 banks, so the two k-groups of a half-wave hit disjoint banks).  Every slab writes its
// partial dW; a second kernel adds the slabs in order (fixed order: reproducible, no float atomics).
template <int NT>
__global__ __launch_bounds__(256) void k_wgrad_mfma(const float *__restrict__ X, int64_t ldx, const float *__restrict__ G, int64_t ldg,
                                                     int64_t n, int F, int O, int64_t rows_per_slab, bool aligned,
                                                     float *__restrict__ partial) {
    constexpr int R = 32, XS = 256 + 16, GS = 16 * NT + 16;
    __shared__ float Xs[R * XS];
    __shared__ float Gs[R * GS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int f0 = blockIdx.y * 256, o0 = blockIdx.z * 16 * NT;
    const int64_t r_beg = (int64_t)blockIdx.x * rows_per_slab;
    const int64_t r_end = r_beg + rows_per_slab < n ? r_beg + rows_per_slab : n;
    f32x4 acc[4][NT];
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[ft][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int64_t r0 = r_beg; r0 < r_end; r0 += R) {
        __syncthreads();
        for (int idx = threadIdx.x; idx < R * 64; idx += 256) {                 // X tile: 32 rows x 256 features
            const int rr = idx / 64, cc = (idx % 64) * 4;
            const int64_t row = r0 + rr;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < r_end) {
                const float *__restrict__ src = X + row * ldx + f0 + cc;
                if (aligned && f0 + cc + 3 < F) v = *reinterpret_cast<const f32x4 *>(src);
                else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) if (f0 + cc + t < F) v[t] = src[t];
                }
            }
            *reinterpret_cast<f32x4 *>(Xs + rr * XS + cc) = v;
        }
        for (int idx = threadIdx.x; idx < R * 4 * NT; idx += 256) {             // G tile: 32 rows x 16 NT outputs
            const int rr = idx / (4 * NT), cc = (idx % (4 * NT)) * 4;
            const int64_t row = r0 + rr;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < r_end) {
                const float *__restrict__ src = G + row * ldg + o0 + cc;
                if (aligned && o0 + cc + 3 < O) v = *reinterpret_cast<const f32x4 *>(src);
                else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) if (o0 + cc + t < O) v[t] = src[t];
                }
            }
            *reinterpret_cast<f32x4 *>(Gs + rr * GS + cc) = v;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < R / 4; ++kk) {
            const float *__restrict__ xrow = Xs + (4 * kk + g) * XS + 64 * wave + c;
            const float *__restrict__ grow = Gs + (4 * kk + g) * GS + c;
            float b[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[nt] = grow[16 * nt];
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) {
                const float a = xrow[16 * ft];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[ft][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[nt], acc[ft][nt], 0, 0, 0);
            }
        }
    }
    float *__restrict__ out = partial + (int64_t)blockIdx.x * F * O;
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int o = o0 + 16 * nt + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = f0 + 64 * wave + 16 * ft + 4 * g + r;
                if (f < F && o < O) out[(int64_t)f * O + o] = acc[ft][nt][r];
            }
        }
}

// ---- the same gradient with the ACCUMULATORS stationary: every wave keeps a whole F x O partial in its registers --------------------
// For F x O <= 16384 (256 x 64, 128 x 128, 64 x 64, ...: the layers of the path) one wave's 512 registers hold the complete result,
// (F / 16) x (O / 16) accumulator tiles (wider layers: one panel of at most 128 outputs x 16384 / 128 features per wave, every slab of
// rows walked once per panel; widths that are not 32 / 64 / 128 / 256 are padded inside the LDS image only).  A wave then needs
// nobody: it owns a slab of rows, streams X[rows, :] and G[rows, :] through a
// private LDS ring by LDS-DMA (whole lines, no VGPR staging, RING - 1 stages in flight behind a counted s_waitcnt vmcnt), and per 4 rows
// reads F / 16 + O / 16 single-word fragments for (F / 16) (O / 16) MFMAs -- no barrier anywhere, nothing recomputed, and shapes narrower
// than k_wgrad_mfma's 256-feature panel waste nothing.  The LDS image is lane-linear (an LDS-DMA cannot scatter), so the 16-byte pieces
// of ODD rows are swapped in groups of four (piece ^ 4) on the source address: a half-wave's fragment read -- 2 rows x 16 consecutive
// words -- then covers all 32 banks once.  G is the MFMA's A operand: a lane ends up with four consecutive outputs of one feature row
// (16-byte stores of the partial).  The partials of the waves are added in wave order by k_sum_slabs (fixed order: reproducible).
template <int MT, int NT> struct WgradAcc {
    static constexpr int F = 16 * MT, O = 16 * NT, R = 8;                 // rows per stage
    static constexpr int XI = R * F / 256, GI = R * O / 256, NI = XI + GI; // LDS-DMA instructions per stage
    static constexpr int STAGE = R * (F + O);                             // floats
    static constexpr int RING_FIT = (36 << 10) / (STAGE * 4);
    static constexpr int RING = RING_FIT > 8 ? 8 : (RING_FIT < 2 ? 2 : RING_FIT);
    static_assert(NI * (RING - 1) <= 63, "vmcnt is a 6-bit counter");
    static_assert(MT * NT <= 64, "the accumulators must fit the register file");
};

template <int MT, int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_wgrad_acc(const float *__restrict__ X, uint32_t x_pitch, int f_all, const float *__restrict__ G, uint32_t g_pitch, int o_all, uint32_t n,
                 uint32_t rows_per_wave, uint32_t f_panels, uint32_t panels, float *__restrict__ partial) {
    using Cfg = WgradAcc<MT, NT>;
    // F x O: the PANEL this wave accumulates (padded to whole 16-column tiles of a power-of-two count).  Columns past the real widths
    // stage a piece that exists (piece 0 of the row) and feed only accumulator cells that are never stored.
    // Block -> (row slabs, panel): the panels of one group of slabs are 8 blocks apart, i.e. dispatched together AND on the same XCD
    // (blocks go round the 8 XCDs), so the rows of G that every panel reads again come out of that XCD's L2.
    constexpr int F = Cfg::F, O = Cfg::O, R = Cfg::R, RING = Cfg::RING, STAGE = Cfg::STAGE;
    const uint32_t bgroup = blockIdx.x / (8 * panels), brem = blockIdx.x % (8 * panels);
    const uint32_t slab_block = bgroup * 8 + brem % 8;
    const int f0 = (int)((brem / 8) % f_panels) * F, o0 = (int)((brem / 8) / f_panels) * O;       // panels = feature panels x output panels
    const uint32_t f_pieces = (uint32_t)((f_all - f0 < F ? f_all - f0 : F) / 4), o_pieces = (uint32_t)((o_all - o0 < O ? o_all - o0 : O) / 4);
    X += f0;
    G += o0;
    extern __shared__ float lds[];                                // [4 waves][RING][STAGE: R rows of X | R rows of G]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;
    const uint32_t wid = slab_block * 4 + wave;
    const uint32_t r_beg = wid * rows_per_wave;
    if (r_beg >= n) return;                                       // (whole waves; there is no barrier in this kernel)
    const uint32_t r_end = r_beg + rows_per_wave < n ? r_beg + rows_per_wave : n;
    const uint32_t n_stages = (r_end - r_beg + R - 1) / R;
    float *__restrict__ ring = lds + wave * (RING * STAGE);
    const char *__restrict__ Xb = reinterpret_cast<const char *>(X);
    const char *__restrict__ Gb = reinterpret_cast<const char *>(G);

    // staging: instruction i of the X part fills slots 64 i .. 64 i + 63 of the stage's X image (slot = 16-byte piece, F / 4 per row)
    uint32_t pf = 0, pf_slot = 0;
    auto issue_next = [&]() {
        const uint32_t row0 = r_beg + (pf < n_stages ? pf : n_stages - 1) * R;      // past the slab: the last stage again, which nobody reads
        float *dst = ring + pf_slot * STAGE;
#pragma unroll
        for (int i = 0; i < Cfg::XI; ++i) {
            const uint32_t q = 64 * i + lane, r = q / (F / 4);
            uint32_t piece = (q % (F / 4)) ^ (4 * (r & 1));
            piece = piece < f_pieces ? piece : 0;
            uint32_t row = row0 + r;
            row = row < n ? row : n - 1;                          // rows past the end read a valid row; their fragments are zeroed
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float *>(Xb + (uint64_t)row * x_pitch + 16 * piece), dst + i * 256, 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < Cfg::GI; ++i) {
            const uint32_t q = 64 * i + lane, r = q / (O / 4);
            uint32_t piece = (q % (O / 4)) ^ (4 * (r & 1));
            piece = piece < o_pieces ? piece : 0;
            uint32_t row = row0 + r;
            row = row < n ? row : n - 1;
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float *>(Gb + (uint64_t)row * g_pitch + 16 * piece), dst + R * F + i * 256, 16, 0, 0);
        }
        ++pf;
        pf_slot = pf_slot + 1 == RING ? 0 : pf_slot + 1;
    };
    // fragments of k step s (rows 4 s + g): word 16 t + c of the row sits in piece group t ^ (g & 1)
    const int flip = g & 1;
    const int x_lane = g * F + c, g_lane = g * O + c;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int ft = 0; ft < MT; ++ft)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[ft][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto multiply = [&](auto masked, uint32_t slot, uint32_t valid_rows) {
        const float *__restrict__ Xs = ring + slot * STAGE + x_lane;
        const float *__restrict__ Gs = ring + slot * STAGE + R * F + g_lane;
#pragma unroll
        for (int s = 0; s < R / 4; ++s) {
            float a[MT], b[NT];
#pragma unroll
            for (int ft = 0; ft < MT; ++ft) a[ft] = Xs[4 * s * F + 16 * (ft ^ flip)];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[nt] = Gs[4 * s * O + 16 * (nt ^ flip)];
            if constexpr (decltype(masked)::value) {
                if ((uint32_t)(4 * s + g) >= valid_rows) {        // 0 x 0: a row past the slab adds nothing, whatever the row that was read holds
#pragma unroll
                    for (int ft = 0; ft < MT; ++ft) a[ft] = 0.f;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) b[nt] = 0.f;
                }
            }
#pragma unroll
            for (int ft = 0; ft < MT; ++ft)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[ft][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[nt], a[ft], acc[ft][nt], 0, 0, 0);
        }
    };
#pragma unroll
    for (int s0 = 0; s0 < RING - 1; ++s0) issue_next();
    uint32_t slot = 0;
    for (uint32_t st = 0; st + 1 < n_stages; ++st) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the fragments of the stage whose slot is restaged next have been read
        issue_next();
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(Cfg::NI * (RING - 1)) : "memory");    // all but the youngest RING - 1 stages: stage st has landed
        multiply(std::false_type{}, slot, R);
        slot = slot + 1 == RING ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // (also: nothing of this wave may still be writing LDS when it retires)
    multiply(std::true_type{}, slot, r_end - (r_beg + (n_stages - 1) * R));
    // D layout with G as the A operand: lane (c, g), register r -> feature 16 ft + c, output 16 nt + 4 g + r
    float *__restrict__ out = partial + (uint64_t)wid * ((uint32_t)f_all * (uint32_t)o_all) + (uint32_t)f0 * (uint32_t)o_all + o0;
#pragma unroll
    for (int ft = 0; ft < MT; ++ft)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            if (f0 + 16 * ft + c < f_all && o0 + 16 * nt + 4 * g < o_all)
                *reinterpret_cast<f32x4 *>(out + (16 * ft + c) * o_all + 16 * nt + 4 * g) = acc[ft][nt];
}

template <int MT, int NT>
int launch_wgrad_acc(const float *X, int64_t ldx, int64_t F, const float *G, int64_t ldg, int64_t O, int64_t n, float *work, int64_t max_slabs,
                     int64_t *n_slabs, hipStream_t s) {
    using Cfg = WgradAcc<MT, NT>;
    const size_t lds_bytes = (size_t)4 * Cfg::RING * Cfg::STAGE * sizeof(float);
    static PerDeviceOnce configured;
    const int attr_dev = PerDeviceOnce::device();
    if (configured.need(attr_dev)) {
        GNX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_wgrad_acc<MT, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10));
        configured.set(attr_dev);
    }
    int cus = 256;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int64_t f_panels = (F + Cfg::F - 1) / Cfg::F, panels = f_panels * ((O + Cfg::O - 1) / Cfg::O);
    int64_t waves = std::min<int64_t>(std::max<int64_t>((int64_t)cus * 4 / panels, 64), max_slabs);        // row slabs; every slab is walked once per panel
    int64_t rows_per_wave = (n + waves - 1) / waves;
    rows_per_wave = std::max<int64_t>((rows_per_wave + Cfg::R - 1) / Cfg::R * Cfg::R, 8 * Cfg::R);
    waves = (n + rows_per_wave - 1) / rows_per_wave;
    const int64_t slab_blocks = (waves + 3) / 4, grid = (slab_blocks + 7) / 8 * 8 * panels;
    hipLaunchKernelGGL((k_wgrad_acc<MT, NT>), dim3((unsigned)grid), dim3(256), lds_bytes, s, X, (uint32_t)(ldx * 4), (int)F, G, (uint32_t)(ldg * 4), (int)O,
                       (uint32_t)n, (uint32_t)rows_per_wave, (uint32_t)f_panels, (uint32_t)panels, work);
    *n_slabs = waves;
    return GNX_OK;
}

// Whole aligned rows, F and O multiples of 4: the result is cut into panels of FP features x OP outputs, FP x OP <= 16384, each of which
// one wave holds (OP = 32 / 64 / 128 >= O where that exists; FP = 16384 / OP, at most 256, narrower for narrow inputs); widths that
// are not a power of two are padded inside the LDS image.  Returns < 0 when the shape is not taken.
int wgrad_acc_dispatch(const float *X, int64_t ldx, const float *G, int64_t ldg, int64_t n, int64_t F, int64_t O, bool aligned, float *work,
                       int64_t max_slabs, int64_t *n_slabs, hipStream_t s) {
    if (!aligned || n < 16 * 1024 || n >= (1ll << 31) || ldx >= (1ll << 30) || ldg >= (1ll << 30) || F % 4 || O % 4 || F * O >= (1ll << 31) || max_slabs < 64)
        return -1;
    const int op = O <= 32 ? 32 : O <= 64 ? 64 : 128;
    int fp = std::min(16384 / op, 256);
    while (fp > 32 && fp / 2 >= F) fp /= 2;                                   // a narrow input does not need the widest panel
    const int64_t panels = ((F + fp - 1) / fp) * ((O + op - 1) / op);
    if (panels > 64) return -1;                                                // (very wide layers: every slab would be walked too often)
#define GNX_WGRAD_ACC(MT_, NT_) if (fp == 16 * MT_ && op == 16 * NT_) return launch_wgrad_acc<MT_, NT_>(X, ldx, F, G, ldg, O, n, work, max_slabs, n_slabs, s)
    GNX_WGRAD_ACC(2, 2); GNX_WGRAD_ACC(2, 4); GNX_WGRAD_ACC(2, 8);
    GNX_WGRAD_ACC(4, 2); GNX_WGRAD_ACC(4, 4); GNX_WGRAD_ACC(4, 8);
    GNX_WGRAD_ACC(8, 2); GNX_WGRAD_ACC(8, 4); GNX_WGRAD_ACC(8, 8);
    GNX_WGRAD_ACC(16, 2); GNX_WGRAD_ACC(16, 4);
#undef GNX_WGRAD_ACC
    return -1;
}

// slabs [group * per, min(group * per + per, n_slabs)) added in order into out[group]: the first level of a two-level sum whose second
// level is k_sum_slabs over the groups (fixed association: reproducible); blockIdx.y = group
__global__ void k_sum_slab_groups(const float *__restrict__ partial, int64_t n_slabs, int64_t per, int64_t elems, float *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= elems) return;
    const int64_t s0 = (int64_t)blockIdx.y * per, s1 = s0 + per < n_slabs ? s0 + per : n_slabs;
    float acc = 0.f;
    for (int64_t s = s0; s < s1; ++s) acc += partial[s * elems + e];
    out[(int64_t)blockIdx.y * elems + e] = acc;
}

__global__ void k_sum_slabs(const float *__restrict__ partial, int64_t n_slabs, int64_t elems, float *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= elems) return;
    float acc = 0.f;
    for (int64_t s = 0; s < n_slabs; ++s) acc += partial[s * elems + e];      // slab order
    out[e] = acc;
}

// ---- task head -----------------------------------------------------------------------------------------------------------
// One 16-lane group per listed node: gather the row, max, sum of exponentials, loss_i = logsumexp - x[label]
// (graph_predictor.py:24-25: CE-from-logits applied to log_softmax(x); softmax(log_softmax(x)) = softmax(x), so this IS the
// plain cross entropy).  Fixed reduction trees: bitwise reproducible.
__device__ __forceinline__ float group16_max(float v) {
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__global__ __launch_bounds__(256) void k_node_ce_fwd(const float *__restrict__ logits, int64_t ldl, int C, int64_t n_rows,
                                                      const int64_t *__restrict__ nodes, const int64_t *__restrict__ labels, int64_t m,
                                                      float *__restrict__ loss) {
    const int sub = threadIdx.x & 15;
    const int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (i >= m) return;
    const int64_t node = nodes[i], label = labels[i];
    if (node < 0 || node >= n_rows || label < 0 || label >= C) {      // never read out of bounds: the loss of such an item is NaN
        if (sub == 0) loss[i] = NAN;
        return;
    }
    const float *__restrict__ x = logits + node * ldl;
    float mx = -INFINITY;
    for (int c = sub; c < C; c += 16) mx = fmaxf(mx, x[c]);
    mx = group16_max(mx);
    float se = 0.f;
    for (int c = sub; c < C; c += 16) se += expf(x[c] - mx);
    se = group16_sum(se);
    if (sub == 0) loss[i] = (logf(se) + mx) - x[label];
}

// d logits[node_i, :] += scale * (softmax(x) - onehot(label)); atomics because a node may be listed twice
__global__ __launch_bounds__(256) void k_node_ce_bwd(const float *__restrict__ logits, int64_t ldl, int C, int64_t n_rows,
                                                      const int64_t *__restrict__ nodes, const int64_t *__restrict__ labels, int64_t m,
                                                      const float *__restrict__ gout, float inv_m, float *__restrict__ grad, int64_t ldg) {
    const int sub = threadIdx.x & 15;
    const int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (i >= m) return;
    const int64_t node = nodes[i], label = labels[i];
    if (node < 0 || node >= n_rows || label < 0 || label >= C) return;
    const float *__restrict__ x = logits + node * ldl;
    float mx = -INFINITY;
    for (int c = sub; c < C; c += 16) mx = fmaxf(mx, x[c]);
    mx = group16_max(mx);
    float se = 0.f;
    for (int c = sub; c < C; c += 16) se += expf(x[c] - mx);
    se = group16_sum(se);
    const float scale = gout[0] * inv_m, inv = 1.0f / se;
    for (int c = sub; c < C; c += 16) {
        const float pr = expf(x[c] - mx) * inv;
        atomicAdd(grad + node * ldg + c, scale * (pr - (c == label ? 1.0f : 0.0f)));
    }
}

// mean of m values in a fixed order: MEAN_BLOCKS blocks each reduce a contiguous slice (strided partial sums + a fixed LDS
// tree) into partial[block]; one block then adds the partials in the same way and divides
constexpr int MEAN_BLOCKS = 256;
__global__ __launch_bounds__(256) void k_mean_partial(const float *__restrict__ v, int64_t m, float *__restrict__ partial) {
    __shared__ float red[256];
    const int64_t per = (m + MEAN_BLOCKS - 1) / MEAN_BLOCKS;
    const int64_t b = (int64_t)blockIdx.x * per, e = b + per < m ? b + per : m;
    float acc = 0.f;
    for (int64_t i = b + threadIdx.x; i < e; i += 256) acc += v[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void k_mean(const float *__restrict__ v, int64_t n_partial, int64_t m, float *__restrict__ out) {
    __shared__ float red[256];
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < n_partial; i += 256) acc += v[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0] / (float)m;
}

// first index of the row maximum (tf.argmax / np.argmax tie rule)
__global__ __launch_bounds__(256) void k_node_argmax(const float *__restrict__ logits, int64_t ldl, int C, int64_t n_rows,
                                                      const int64_t *__restrict__ nodes, int64_t m, int64_t *__restrict__ out) {
    const int sub = threadIdx.x & 15;
    const int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (i >= m) return;
    const int64_t node = nodes ? nodes[i] : i;
    if (node < 0 || node >= n_rows) {                                   // out of range: -1
        if (sub == 0) out[i] = -1;
        return;
    }
    const float *__restrict__ x = logits + node * ldl;
    float best = -INFINITY;
    int arg = C;                                           // rows of NaNs: no element compares greater; report 0 like np.argmax of all-equal
    for (int c = sub; c < C; c += 16) {
        const float v = x[c];
        if (v > best || (v == best && c < arg)) { best = v; arg = c; }
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off);
        const int oa = __shfl_xor(arg, off);
        if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
    }
    if (sub == 0) out[i] = arg < C ? arg : 0;
}

// ---- link head: logit_i = sum_c F[u_i, c] * F[v_i, c] * (r[c] or 1)   (graph_predictor.py:122-126) ---------------------
__global__ __launch_bounds__(256) void k_edge_scores(const float *__restrict__ F, int64_t ldf, int C, int64_t n_rows,
                                                      const int64_t *__restrict__ edges, int64_t m, const float *__restrict__ r,
                                                      float *__restrict__ out) {
    const int sub = threadIdx.x & 15;
    const int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (i >= m) return;
    const int64_t u = edges[2 * i], v = edges[2 * i + 1];
    if (u < 0 || u >= n_rows || v < 0 || v >= n_rows) {                 // out of range: NaN
        if (sub == 0) out[i] = NAN;
        return;
    }
    const float *__restrict__ fu = F + u * ldf, *__restrict__ fv = F + v * ldf;
    float acc = 0.f;
    for (int c = sub; c < C; c += 16) acc = fmaf(fu[c] * fv[c], r ? r[c] : 1.0f, acc);
    acc = group16_sum(acc);
    if (sub == 0) out[i] = acc;
}

// dF[u_i, :] += g_i * F[v_i, :] * r,  dF[v_i, :] += g_i * F[u_i, :] * r   (atomics: endpoints repeat across edges)
__global__ __launch_bounds__(256) void k_edge_scores_bwd(const float *__restrict__ F, int64_t ldf, int C, int64_t n_rows,
                                                          const int64_t *__restrict__ edges, int64_t m, const float *__restrict__ r,
                                                          const float *__restrict__ g, float *__restrict__ dF, int64_t ldg) {
    const int sub = threadIdx.x & 15;
    const int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (i >= m) return;
    const int64_t u = edges[2 * i], v = edges[2 * i + 1];
    if (u < 0 || u >= n_rows || v < 0 || v >= n_rows) return;
    const float gi = g[i];
    for (int c = sub; c < C; c += 16) {
        const float w = gi * (r ? r[c] : 1.0f);
        atomicAdd(dF + u * ldg + c, w * F[v * ldf + c]);
        atomicAdd(dF + v * ldg + c, w * F[u * ldf + c]);
    }
}

inline bool aligned16(const void *p) { return ((uintptr_t)p % 16) == 0; }

}  // namespace

namespace gnx {

// used by gnx_spmm.hip (GCNII's long rows go through the dense kernel with a row scatter)
// (tuning builds: GNX_DENSE_RING=0 keeps the register-staged kernel, for A/B runs)
static bool ring_enabled() {
#ifdef GNX_TUNING
    static const bool on = [] { const char *e = getenv("GNX_DENSE_RING"); return !(e && e[0] == '0'); }();
    return on;
#else
    return true;
#endif
}

// (tuning builds: GNX_WGRAD_ACC=0 keeps the panel kernel)
static bool wgrad_acc_enabled() {
#ifdef GNX_TUNING
    static const bool on = [] { const char *e = getenv("GNX_WGRAD_ACC"); return !(e && e[0] == '0'); }();
    return on;
#else
    return true;
#endif
}

// (tuning builds: GNX_DENSE_WREG=0 skips the W-in-registers kernel)
static bool wreg_enabled() {
#ifdef GNX_TUNING
    static const bool on = [] { const char *e = getenv("GNX_DENSE_WREG"); return !(e && e[0] == '0'); }();
    return on;
#else
    return true;
#endif
}

int dense_rows(const float *X, int64_t ldx, int64_t n, int64_t F, const float *W, int64_t ldw, int64_t O, const float *bias, int act,
               const int32_t *in_rows, const int32_t *out_rows, float *out, int64_t ldo, hipStream_t s) {
    if (n == 0) return GNX_OK;
    DenseArgs p{X, ldx, n, (int)F, W, ldw, (int)O, bias, act, out, ldo, out_rows, in_rows, false};
    const bool al = ldx % 4 == 0 && aligned16(X);
    for (int64_t o0 = 0; o0 < O; o0 += 256) {                         // column panels of at most 256 outputs
        DenseArgs q = p;
        q.W = W + o0; q.bias = bias ? bias + o0 : nullptr; q.out = out + o0;
        q.w_aligned = ldw % 4 == 0 && aligned16(q.W);
        q.O = (int)(O - o0 < 256 ? O - o0 : 256);
        const int nt = (q.O + 15) / 16;
        const int nt4 = nt;                                                // the ring kernel takes accumulator columns in whole groups of four
        if (wreg_enabled() && wreg_eligible(q, al)) {                     // shapes whose W fits the registers (padded to 128 / 256 x 32 / 64 / 128)
            int rc = 1;                                                       // (> 0: no shape of this kernel)
            if (q.F > 128 && q.O <= 32)       rc = launch_wreg<2, 4, 8>(q, s);
            else if (q.F > 128 && q.O <= 64)  rc = launch_wreg<4, 4, 8>(q, s);
            else if (q.F <= 128 && q.O <= 64) rc = launch_wreg<4, 2, 4>(q, s);
            else if (q.F <= 128 && q.O <= 128) rc = launch_wreg<8, 2, 4>(q, s);
            if (rc < 0) return rc;
            if (rc == GNX_OK) continue;
        }
        if (ring_enabled() && nt % 4 == 0 && ring_eligible(q, al, nt4) && nt4 <= 16) {
            int rc = nt4 == 4 ? launch_ring<4>(q, s) : nt4 == 8 ? launch_ring<8>(q, s) : nt4 == 12 ? launch_ring<12>(q, s) : launch_ring<16>(q, s);
            if (rc != GNX_OK) return rc;
            continue;
        }
        if (nt <= 1) launch_dense<1>(q, al, s);
        else if (nt <= 2) launch_dense<2>(q, al, s);
        else if (nt <= 3) launch_dense<3>(q, al, s);
        else if (nt <= 4) launch_dense<4>(q, al, s);
        else if (nt <= 6) launch_dense<6>(q, al, s);
        else if (nt <= 8) launch_dense<8>(q, al, s);
        else if (nt <= 12) launch_dense<12>(q, al, s);
        else launch_dense<16>(q, al, s);
    }
    GNX_HIP(hipGetLastError());
    return GNX_OK;
}

}  // namespace gnx

extern "C" {

int gnx_dense(const float *d_X, int64_t ldx, int64_t n, int64_t F, const float *d_W, int64_t ldw, int64_t O, const float *d_bias,
              int act, float *d_out, int64_t ldo, void *stream) {
    GNX_CHECK_ARG(n >= 0 && F >= 1 && O >= 1 && F <= (1 << 24) && O <= (1 << 20), "gnx_dense: bad sizes (n=%lld, F=%lld, O=%lld)",
                  (long long)n, (long long)F, (long long)O);
    GNX_CHECK_ARG(ldx >= F && ldw >= O && ldo >= O, "gnx_dense: leading dimension smaller than the row");
    GNX_CHECK_ARG(act == GNX_ACT_NONE || act == GNX_ACT_RELU, "gnx_dense: invalid activation %d", act);
    if (n == 0) return GNX_OK;
    GNX_CHECK_ARG(d_X && d_W && d_out, "gnx_dense: NULL pointer");
    GNX_CHECK_ARG((const void *)d_X != (const void *)d_out, "gnx_dense: out must not alias X");
    return dense_rows(d_X, ldx, n, F, d_W, ldw, O, d_bias, act, nullptr, nullptr, d_out, ldo, (hipStream_t)stream);
}

int gnx_dense_wgrad(const float *d_X, int64_t ldx, const float *d_G, int64_t ldg, int64_t n, int64_t F, int64_t O, float *d_dW,
                    float *d_work, int64_t work_floats, void *stream) {
    GNX_CHECK_ARG(n >= 0 && F >= 1 && O >= 1 && F <= (1 << 20) && O <= (1 << 20), "gnx_dense_wgrad: bad sizes");
    GNX_CHECK_ARG(ldx >= F && ldg >= O, "gnx_dense_wgrad: leading dimension smaller than the row");
    GNX_CHECK_ARG(d_dW != nullptr, "gnx_dense_wgrad: NULL output");
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        GNX_HIP(hipMemsetAsync(d_dW, 0, (size_t)F * O * sizeof(float), s));
        return GNX_OK;
    }
    GNX_CHECK_ARG(d_X && d_G, "gnx_dense_wgrad: NULL input");
    // row slabs: as many as the scratch holds (each slab leaves an F x O partial), at least 256 rows each, at most 2048 slabs
    const int64_t fo = F * O;
    int64_t max_slabs = work_floats / fo;
    GNX_CHECK_ARG(d_work != nullptr && max_slabs >= 1, "gnx_dense_wgrad: the scratch must hold at least F * O floats");
    if (max_slabs > 2048) max_slabs = 2048;
    int64_t rows_per_slab = (n + max_slabs - 1) / max_slabs;
    if (rows_per_slab < 256) rows_per_slab = 256;
    rows_per_slab = (rows_per_slab + 31) / 32 * 32;
    const int64_t n_slabs = (n + rows_per_slab - 1) / rows_per_slab;
    const bool al = ldx % 4 == 0 && ldg % 4 == 0 && aligned16(d_X) && aligned16(d_G);
    if (wgrad_acc_enabled() && aligned16(d_work)) {                          // the shapes whose whole result fits one wave's registers
        int64_t waves = 0;
        const int rc = wgrad_acc_dispatch(d_X, ldx, d_G, ldg, n, F, O, al, d_work, work_floats / fo > 2048 ? 2048 : work_floats / fo, &waves, s);
        if (rc > 0) return rc;
        if (rc == GNX_OK) {
            // the waves' partials, added in wave order: in groups of 32 first when the scratch has room for the group sums (a sum over a
            // thousand slabs of a few thousand elements is otherwise a launch of a few blocks walking a long chain each)
            const int64_t per = 32, groups = (waves + per - 1) / per;
            if (waves > 64 && work_floats >= (waves + groups) * fo) {
                float *tmp = d_work + waves * fo;
                hipLaunchKernelGGL(k_sum_slab_groups, dim3((unsigned)((fo + 255) / 256), (unsigned)groups), dim3(256), 0, s, d_work, waves, per, fo, tmp);
                hipLaunchKernelGGL(k_sum_slabs, dim3((unsigned)((fo + 255) / 256)), dim3(256), 0, s, tmp, groups, fo, d_dW);
            } else {
                hipLaunchKernelGGL(k_sum_slabs, dim3((unsigned)((fo + 255) / 256)), dim3(256), 0, s, d_work, waves, fo, d_dW);
            }
            GNX_HIP(hipGetLastError());
            return GNX_OK;
        }
    }
    const int nt_all = (int)((O + 15) / 16);
    const int NTsel = nt_all >= 4 ? 4 : (nt_all >= 2 ? 2 : 1);
    dim3 grid((unsigned)n_slabs, (unsigned)((F + 255) / 256), (unsigned)((nt_all + NTsel - 1) / NTsel));
    if (NTsel == 4)      hipLaunchKernelGGL(k_wgrad_mfma<4>, grid, dim3(256), 0, s, d_X, ldx, d_G, ldg, n, (int)F, (int)O, rows_per_slab, al, d_work);
    else if (NTsel == 2) hipLaunchKernelGGL(k_wgrad_mfma<2>, grid, dim3(256), 0, s, d_X, ldx, d_G, ldg, n, (int)F, (int)O, rows_per_slab, al, d_work);
    else                 hipLaunchKernelGGL(k_wgrad_mfma<1>, grid, dim3(256), 0, s, d_X, ldx, d_G, ldg, n, (int)F, (int)O, rows_per_slab, al, d_work);
    hipLaunchKernelGGL(k_sum_slabs, dim3((unsigned)((fo + 255) / 256)), dim3(256), 0, s, d_work, n_slabs, fo, d_dW);
    GNX_HIP(hipGetLastError());
    return GNX_OK;
}

int gnx_node_ce(const float *d_logits, int64_t ldl, int64_t n_rows, int64_t C, const int64_t *d_nodes, const int64_t *d_labels, int64_t m,
                float *d_loss_per_node, float *d_mean_loss, void *stream) {
    GNX_CHECK_ARG(m >= 1 && C >= 1 && n_rows >= 1 && ldl >= C, "gnx_node_ce: bad sizes");
    GNX_CHECK_ARG(d_logits && d_nodes && d_labels && d_loss_per_node && d_mean_loss, "gnx_node_ce: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_node_ce_fwd, dim3((unsigned)((m + 15) / 16)), dim3(256), 0, s, d_logits, ldl, (int)C, n_rows, d_nodes, d_labels, m,
                       d_loss_per_node);
    if (m > 4096) {      // two-level mean; the partial sums live in the scratch tail of d_loss_per_node
        float *partial = d_loss_per_node + m;
        hipLaunchKernelGGL(k_mean_partial, dim3(MEAN_BLOCKS), dim3(256), 0, s, d_loss_per_node, m, partial);
        hipLaunchKernelGGL(k_mean, dim3(1), dim3(256), 0, s, partial, (int64_t)MEAN_BLOCKS, m, d_mean_loss);
    } else {
        hipLaunchKernelGGL(k_mean, dim3(1), dim3(256), 0, s, d_loss_per_node, m, m, d_mean_loss);
    }
    GNX_HIP(hipGetLastError());
    return GNX_OK;
}

int gnx_node_ce_backward(const float *d_logits, int64_t ldl, int64_t n_rows, int64_t C, const int64_t *d_nodes, const int64_t *d_labels,
                         int64_t m, const float *d_grad_loss, float *d_grad_logits, int64_t ldg, void *stream) {
    GNX_CHECK_ARG(m >= 1 && C >= 1 && n_rows >= 1 && ldl >= C && ldg >= C, "gnx_node_ce_backward: bad sizes");
    GNX_CHECK_ARG(d_logits && d_nodes && d_labels && d_grad_loss && d_grad_logits, "gnx_node_ce_backward: NULL pointer");
    hipLaunchKernelGGL(k_node_ce_bwd, dim3((unsigned)((m + 15) / 16)), dim3(256), 0, (hipStream_t)stream, d_logits, ldl, (int)C, n_rows,
                       d_nodes, d_labels, m, d_grad_loss, 1.0f / (float)m, d_grad_logits, ldg);
    GNX_HIP(hipGetLastError());
    return GNX_OK;
}

int gnx_edge_scores(const float *d_F, int64_t ldf, int64_t n_rows, int64_t C, const int64_t *d_edges, int64_t m, const float *d_r,
                    float *d_out, void *stream) {
    GNX_CHECK_ARG(m >= 0 && C >= 1 && n_rows >= 0 && ldf >= C, "gnx_edge_scores: bad sizes");
    if (m == 0) return GNX_OK;
    GNX_CHECK_ARG(d_F && d_edges && d_out, "gnx_edge_scores: NULL pointer");
    hipLaunchKernelGGL(k_edge_scores, dim3((unsigned)((m + 15) / 16)), dim3(256), 0, (hipStream_t)stream, d_F, ldf, (int)C, n_rows, d_edges, m,
                       d_r, d_out);
    GNX_HIP(hipGetLastError());
    return GNX_OK;
}

int gnx_edge_scores_backward(const float *d_F, int64_t ldf, int64_t n_rows, int64_t C, const int64_t *d_edges, int64_t m, const float *d_r,
                             const float *d_grad_out, float *d_grad_F, int64_t ldg, void *stream) {
    GNX_CHECK_ARG(m >= 0 && C >= 1 && ldf >= C && ldg >= C, "gnx_edge_scores_backward: bad sizes");
    if (m == 0) return GNX_OK;
    GNX_CHECK_ARG(d_F && d_edges && d_grad_out && d_grad_F, "gnx_edge_scores_backward: NULL pointer");
    hipLaunchKernelGGL(k_edge_scores_bwd, dim3((unsigned)((m + 15) / 16)), dim3(256), 0, (hipStream_t)stream, d_F, ldf, (int)C, n_rows, d_edges,
                       m, d_r, d_grad_out, d_grad_F, ldg);
    GNX_HIP(hipGetLastError());
    return GNX_OK;
}

int gnx_node_argmax(const float *d_logits, int64_t ldl, int64_t n_rows, int64_t C, const int64_t *d_nodes, int64_t m, int64_t *d_out,
                    void *stream) {
    GNX_CHECK_ARG(m >= 0 && C >= 1 && n_rows >= 0 && ldl >= C, "gnx_node_argmax: bad sizes");
    if (m == 0) return GNX_OK;
    GNX_CHECK_ARG(d_logits && d_out, "gnx_node_argmax: NULL pointer");
    hipLaunchKernelGGL(k_node_argmax, dim3((unsigned)((m + 15) / 16)), dim3(256), 0, (hipStream_t)stream, d_logits, ldl, (int)C, n_rows, d_nodes, m,
                       d_out);
    GNX_HIP(hipGetLastError());
    return GNX_OK;
}

}  // extern "C"

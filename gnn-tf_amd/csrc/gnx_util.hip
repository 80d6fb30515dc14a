// Small entries beside the hot path: row gather (halo packing's fallback), the stream yardsticks bench.py measures next to the SpMM,
// the one-pass linear combination that ends the K loop's backward, and the XCD placement probe.
#include "gnx_spmm_device.h"

namespace {

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

}  // namespace

extern "C" {

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

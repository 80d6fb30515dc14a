// Standalone sweep for the HBM yardsticks of bench.py (gnx_stream_copy / gnx_stream_read): which launch shape reaches the
// ~6.3 TB/s float4-copy rate of /opt/skills/guides/MI355X_MICROARCH.md on this box?
//   hipcc -O3 --offload-arch=gfx950 tools/stream_bench.hip -o /tmp/stream_bench && /tmp/stream_bench
// Prints one line per (threads per block, blocks per CU, loads in flight per lane, temporal / non-temporal).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(e) do { hipError_t s_ = (e); if (s_ != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(s_), __LINE__); exit(1); } } while (0)

template <int U, bool NT, bool COPY>
__global__ void k_stream(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, long n4, float *__restrict__ sink) {
    const long tile = (long)blockDim.x * U;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long base = (long)blockIdx.x * tile; base + tile <= n4; base += (long)gridDim.x * tile) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const f32x4 *p = src + base + (long)u * blockDim.x + threadIdx.x;
            v[u] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (COPY) {
                f32x4 *q = dst + base + (long)u * blockDim.x + threadIdx.x;
                if (NT) __builtin_nontemporal_store(v[u], q); else *q = v[u];
            } else acc += v[u];
        }
    }
    if (!COPY) {
        float s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        if (s == 12345.678f) sink[0] = s;          // keeps the loads alive without a reduction
    }
}

template <int U, bool NT, bool COPY>
float run(const f32x4 *src, f32x4 *dst, long n4, float *sink, int threads, int blocks, int reps) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_stream<U, NT, COPY>), dim3(blocks), dim3(threads), 0, 0, src, dst, n4, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_stream<U, NT, COPY>), dim3(blocks), dim3(threads), 0, 0, src, dst, n4, sink);
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    const long bytes = 4L << 30, n4 = bytes / 16;
    f32x4 *src, *dst;
    float *sink;
    CHECK(hipMalloc((void **)&src, bytes)); CHECK(hipMalloc((void **)&dst, bytes)); CHECK(hipMalloc((void **)&sink, 256));
    CHECK(hipMemset(src, 1, bytes)); CHECK(hipMemset(dst, 0, bytes));
    const int cus = 256;
    const int threads_opts[] = {256, 512, 1024};
    const int bpc_opts[] = {1, 2, 4, 8, 16};
    printf("%8s %8s %4s %4s %12s %12s\n", "threads", "blk/CU", "U", "nt", "copy GB/s", "read GB/s");
    for (int threads : threads_opts)
        for (int bpc : bpc_opts) {
            if (threads * bpc > 2048 * 2) continue;
            const int blocks = cus * bpc;
#define ROW(U, NT)                                                                                              \
    do {                                                                                                        \
        const float c = run<U, NT, true>(src, dst, n4, sink, threads, blocks, 5);                               \
        const float r = run<U, NT, false>(src, dst, n4, sink, threads, blocks, 5);                              \
        printf("%8d %8d %4d %4d %12.0f %12.0f\n", threads, bpc, U, (int)NT, 2.0 * bytes / c / 1e6, 1.0 * bytes / r / 1e6); \
        fflush(stdout);                                                                                         \
    } while (0)
            ROW(4, false); ROW(4, true); ROW(8, false); ROW(8, true); ROW(16, true);
#undef ROW
        }
    // hipMemcpy device-to-device for comparison
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice));
    CHECK(hipEventRecord(a, 0));
    for (int r = 0; r < 5; ++r) CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0));
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    printf("hipMemcpyAsync D2D: %.0f GB/s (read + write)\n", 2.0 * bytes / (ms / 5) / 1e6);
    return 0;
}

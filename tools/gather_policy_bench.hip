// Does any load flavour make the L2 fetch LESS than a 128-byte line for a 32-byte gathered row?  Random 32-byte rows of a 2 GiB
// table (far beyond L2 + Infinity Cache), each row fetched once per launch, with plain / nt / sc1 / sc0 sc1 loads.
//   hipcc -O3 --offload-arch=gfx950 tools/gather_policy_bench.hip -o /tmp/gpb && /tmp/gpb
//   rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -- /tmp/gpb
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(e) do { hipError_t s_ = (e); if (s_ != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(s_), __LINE__); exit(1); } } while (0)

template <int MODE>
__device__ __forceinline__ f32x4 ld(const f32x4 *p) {
    f32x4 v;
    if (MODE == 0) v = *p;
    else if (MODE == 1) v = __builtin_nontemporal_load(p);
    else if (MODE == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (MODE == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// thread t gathers row idx[t] (two 16-byte halves) and folds it
template <int MODE>
__global__ void k_gather(const f32x4 *__restrict__ table, const int *__restrict__ idx, long n, float *__restrict__ out) {
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const f32x4 *row = table + (long)idx[t] * 2;
    f32x4 a = ld<MODE>(row), b = ld<MODE>(row + 1);
    out[t] = (a[0] + a[1] + a[2] + a[3]) + (b[0] + b[1] + b[2] + b[3]);
}

int main() {
    const long rows = 64L << 20, n = 32L << 20;             // 2 GiB table, 32M gathers
    f32x4 *table; int *idx; float *out;
    CHECK(hipMalloc((void **)&table, rows * 32)); CHECK(hipMalloc((void **)&idx, n * 4)); CHECK(hipMalloc((void **)&out, n * 4));
    CHECK(hipMemset(table, 0, rows * 32));
    int *h = (int *)malloc(n * 4);
    unsigned long long s = 88172645463325252ull;
    for (long i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (int)(s % (unsigned long long)rows); }
    CHECK(hipMemcpy(idx, h, n * 4, hipMemcpyHostToDevice));
    const char *names[5] = {"plain", "nt", "sc1", "sc0_sc1", "sc0_sc1_nt"};
    for (int mode = 0; mode < 5; ++mode) {
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        for (int rep = 0; rep < 3; ++rep) {
            if (rep == 1) CHECK(hipEventRecord(a, 0));
            const dim3 grid((unsigned)((n + 255) / 256));
            if (mode == 0) hipLaunchKernelGGL(k_gather<0>, grid, dim3(256), 0, 0, table, idx, n, out);
            if (mode == 1) hipLaunchKernelGGL(k_gather<1>, grid, dim3(256), 0, 0, table, idx, n, out);
            if (mode == 2) hipLaunchKernelGGL(k_gather<2>, grid, dim3(256), 0, 0, table, idx, n, out);
            if (mode == 3) hipLaunchKernelGGL(k_gather<3>, grid, dim3(256), 0, 0, table, idx, n, out);
            if (mode == 4) hipLaunchKernelGGL(k_gather<4>, grid, dim3(256), 0, 0, table, idx, n, out);
        }
        CHECK(hipEventRecord(b, 0)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        printf("%-12s %.3f ms per launch  %.1f G gathers/s\n", names[mode], ms / 2, n / (ms / 2) / 1e6);
    }
    return 0;
}

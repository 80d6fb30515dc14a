/* Plain-C client of include/gnx.h: the boundary is usable without Python or torch.
 * Path graph P3 (0-1-2), symmetric normalisation, one PPR step from H = H0 = e0.
 * Expected (tests/test_oracle_kat.py::test_kat2_path_p3): out = [0.1, 0.9/sqrt(2), 0].
 * Build: gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/c_abi_smoke.c -Lgnn-tf_amd/lib -lgnx -L/opt/rocm/lib -lamdhip64 -lm */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>

#include "gnx.h"

#define CHECK_HIP(e) do { hipError_t _s = (e); if (_s != hipSuccess) { printf("hip error %d at line %d\n", (int)_s, __LINE__); return 2; } } while (0)
#define CHECK_GNX(e) do { int _s = (e); if (_s != GNX_OK) { printf("gnx error %d: %s (line %d)\n", _s, gnx_last_error(), __LINE__); return 3; } } while (0)

int main(void) {
    const int64_t idx[8] = {0, 1, 1, 2, 1, 0, 2, 1};            /* graph2adj of P3: edges + reversed edges */
    const float val[4] = {1.f, 1.f, 1.f, 1.f};
    const float h0[3] = {1.f, 0.f, 0.f};
    int64_t *d_idx; float *d_val, *d_h0, *d_out, *d_vals_n;
    CHECK_HIP(hipMalloc((void **)&d_idx, sizeof idx));
    CHECK_HIP(hipMalloc((void **)&d_val, sizeof val));
    CHECK_HIP(hipMalloc((void **)&d_h0, sizeof h0));
    CHECK_HIP(hipMalloc((void **)&d_out, sizeof h0));
    CHECK_HIP(hipMalloc((void **)&d_vals_n, 4 * sizeof(float)));
    CHECK_HIP(hipMemcpy(d_idx, idx, sizeof idx, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_val, val, sizeof val, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_h0, h0, sizeof h0, hipMemcpyHostToDevice));

    gnx_graph_t g = NULL;
    CHECK_GNX(gnx_graph_create_coo(3, 3, 4, d_idx, d_val, NULL, &g));
    int64_t n_rows, n_cols, nnz_e, nnz_c;
    CHECK_GNX(gnx_graph_info(g, &n_rows, &n_cols, &nnz_e, &nnz_c));
    if (n_rows != 3 || nnz_e != 4 || nnz_c != 4) { printf("unexpected sizes\n"); return 4; }
    CHECK_GNX(gnx_graph_normalize(g, GNX_NORM_SYMMETRIC, GNX_EYE_NONE, 0.f, 0, 0, d_vals_n, NULL, NULL));
    float *d_h;                                                   /* out must not alias X: separate H buffer */
    CHECK_HIP(hipMalloc((void **)&d_h, sizeof h0));
    CHECK_HIP(hipMemcpy(d_h, h0, sizeof h0, hipMemcpyHostToDevice));
    CHECK_GNX(gnx_ppr_step(g, d_vals_n, NULL, d_h, d_h0, 0.1f, 1, GNX_ACT_NONE, d_out, NULL));
    float out[3];
    CHECK_HIP(hipMemcpy(out, d_out, sizeof out, hipMemcpyDeviceToHost));
    const float want[3] = {0.1f, 0.9f / sqrtf(2.f), 0.f};
    for (int i = 0; i < 3; ++i)
        if (fabsf(out[i] - want[i]) > 1e-6f) { printf("row %d: got %g want %g\n", i, out[i], want[i]); return 5; }
    if (gnx_spmm(g, NULL, NULL, d_h, 1, 1, NULL, 0, 1.f, 0.f, 0, d_h, 1, NULL) != GNX_ERR_INVALID) { printf("aliasing not refused\n"); return 6; }
    CHECK_GNX(gnx_graph_destroy(g));
    printf("C ABI OK (version %d, kernel path verified: %g %g %g)\n", gnx_version(), out[0], out[1], out[2]);
    return 0;
}

/* Plain-C client of include/gnx.h: the boundary is usable without Python or torch.
 * Path graph P3 (0-1-2), symmetric normalisation, one PPR step from H = H0 = e0.
 * Expected (tests/test_oracle_kat.py::test_kat2_path_p3): out = [0.1, 0.9/sqrt(2), 0].
 * Then a two-block vertex partition of P4 through the halo-plan entry points (pull and push), exchange by the caller.
 * Build: gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/c_abi_smoke.c -Lgnn-tf_amd/lib -lgnx -L/opt/rocm/lib -lamdhip64 -lm */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>

#include "gnx.h"

#define CHECK_HIP(e) do { hipError_t _s = (e); if (_s != hipSuccess) { printf("hip error %d at line %d\n", (int)_s, __LINE__); return 2; } } while (0)
#define CHECK_GNX(e) do { int _s = (e); if (_s != GNX_OK) { printf("gnx error %d: %s (line %d)\n", _s, gnx_last_error(), __LINE__); return 3; } } while (0)

/* Two vertex blocks of the path graph P4 (0-1-2-3) on ONE GPU through the halo-plan entry points, the exchange done by
 * the caller (hipMemcpy between the two blocks' buffers with the offsets of gnx_halo_plan_layout):
 *   block 0 = rows {0, 1}: row 1's remote entry (1, 2) is PUSHED -- block 1 sends the partial sum w12 * H[2], block 0 adds it
 *             through a weight-1 entry on the push slot;
 *   block 1 = rows {2, 3}: row 2's remote entry (2, 1) is PULLED -- block 0 sends the row H[1].
 * One fused step per block must equal the step on the whole graph. */
static int upload(const void *src, size_t bytes, void **dst) {
    if (hipMalloc(dst, bytes) != hipSuccess) return 1;
    return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) != hipSuccess;
}

static int two_block_plan(void) {
    enum { C = 2 };
    const float w01 = 1.f / sqrtf(2.f), w12 = 0.5f, w23 = 1.f / sqrtf(2.f);     /* D^-1/2 A D^-1/2 of P4, degrees 1 2 2 1 */
    const float H[4 * C] = {1.f, 10.f, 2.f, 20.f, 3.f, 30.f, 4.f, 40.f};
    /* the whole graph, for the expected answer */
    const int64_t idx_all[12] = {0, 1, 1, 0, 1, 2, 2, 1, 2, 3, 3, 2};
    const float val_all[6] = {w01, w01, w12, w12, w23, w23};
    void *d_i, *d_v, *d_H, *d_want;
    if (upload(idx_all, sizeof idx_all, &d_i) || upload(val_all, sizeof val_all, &d_v) || upload(H, sizeof H, &d_H)) return 20;
    CHECK_HIP(hipMalloc(&d_want, sizeof H));
    gnx_graph_t whole = NULL;
    CHECK_GNX(gnx_graph_create_coo(4, 4, 6, (const int64_t *)d_i, (const float *)d_v, NULL, &whole));
    CHECK_GNX(gnx_ppr_step(whole, NULL, NULL, (const float *)d_H, (const float *)d_H, 0.1f, C, GNX_ACT_NONE, (float *)d_want, NULL));
    float want[4 * C];
    CHECK_HIP(hipMemcpy(want, d_want, sizeof want, hipMemcpyDeviceToHost));

    /* block 0: X0 = [v0, v1 | slot pushed by block 1];  block 1: X1 = [v1 pulled from block 0 | v2, v3] */
    const int64_t m0_idx[6] = {0, 1, 1, 0, 1, 2};   const float m0_val[3] = {w01, w01, 1.f};
    const int64_t m1_idx[6] = {0, 0, 0, 2, 1, 1};   const float m1_val[3] = {w12, w23, w23};
    const int32_t pull0_src[1] = {1};                                                   /* block 0 -> 1: its local row 1 (= H[1]), a copy */
    const int64_t push1_idx[2] = {0, 0};            const float push1_val[1] = {w12};   /* block 1 -> 0: w12 * H[2], summed by the sender */
    void *d_m0i, *d_m0v, *d_m1i, *d_m1v, *d_p0s, *d_p1i, *d_p1v;
    if (upload(m0_idx, sizeof m0_idx, &d_m0i) || upload(m0_val, sizeof m0_val, &d_m0v) || upload(m1_idx, sizeof m1_idx, &d_m1i) ||
        upload(m1_val, sizeof m1_val, &d_m1v) || upload(pull0_src, sizeof pull0_src, &d_p0s) || upload(push1_idx, sizeof push1_idx, &d_p1i) ||
        upload(push1_val, sizeof push1_val, &d_p1v)) return 21;
    gnx_graph_t main_g[2] = {NULL, NULL}, push_g[2] = {NULL, NULL};
    CHECK_GNX(gnx_graph_create_coo(2, 3, 3, (const int64_t *)d_m0i, (const float *)d_m0v, NULL, &main_g[0]));
    CHECK_GNX(gnx_graph_create_coo(2, 3, 3, (const int64_t *)d_m1i, (const float *)d_m1v, NULL, &main_g[1]));
    CHECK_GNX(gnx_graph_create_coo(1, 2, 1, (const int64_t *)d_p1i, (const float *)d_p1v, NULL, &push_g[1]));
    gnx_halo_plan_t plan[2] = {NULL, NULL};
    /* rows per peer q, for block r: received as pulled rows / as pushed sums, sent as pulled rows / as pushed sums */
    const int64_t recv_pull[2][2] = {{0, 0}, {1, 0}}, recv_push[2][2] = {{0, 1}, {0, 0}};
    const int64_t send_pull[2][2] = {{0, 1}, {0, 0}}, send_push[2][2] = {{0, 0}, {1, 0}};
    const int32_t *pull_src[2] = {(const int32_t *)d_p0s, NULL};
    float *d_X[2], *d_send[2], *d_out[2];
    int64_t n_buf[2], local0[2], n_send[2], n_send_pull[2], recv0[2][2], spull0[2][2], spush0[2][2];
    for (int r = 0; r < 2; ++r) {
        CHECK_GNX(gnx_halo_plan_create(2, r, 2, recv_pull[r], recv_push[r], send_pull[r], send_push[r], pull_src[r], push_g[r], &plan[r]));
        CHECK_GNX(gnx_halo_plan_layout(plan[r], &n_buf[r], &local0[r], &n_send[r], &n_send_pull[r], recv0[r], spull0[r], spush0[r]));
        if (n_buf[r] != 3 || n_send[r] != 1 || n_send_pull[r] != (r == 0 ? 1 : 0) || local0[r] != (r == 0 ? 0 : 1) || recv0[r][1 - r] != (r == 0 ? 2 : 0)) {
            printf("layout of block %d\n", r); return 22;
        }
        CHECK_HIP(hipMalloc((void **)&d_X[r], 3 * C * sizeof(float)));
        CHECK_HIP(hipMalloc((void **)&d_send[r], C * sizeof(float)));
        CHECK_HIP(hipMalloc((void **)&d_out[r], 2 * C * sizeof(float)));
        CHECK_HIP(hipMemset(d_X[r], 0, 3 * C * sizeof(float)));
        CHECK_HIP(hipMemcpy(d_X[r] + local0[r] * C, (const float *)d_H + 2 * r * C, 2 * C * sizeof(float), hipMemcpyDeviceToDevice));
        CHECK_GNX(gnx_halo_pack(plan[r], GNX_HALO_PULL, d_X[r], C, C, d_send[r], C, NULL));      /* the two halves, one after the other */
        CHECK_GNX(gnx_halo_pack(plan[r], GNX_HALO_PUSH, d_X[r], C, C, d_send[r], C, NULL));
    }
    /* the caller's exchange: block 0's pulled row lands in block 1's region(0), block 1's pushed sum in block 0's region(1) */
    CHECK_HIP(hipMemcpy(d_X[1] + recv0[1][0] * C, d_send[0] + spull0[0][1] * C, C * sizeof(float), hipMemcpyDeviceToDevice));
    CHECK_HIP(hipMemcpy(d_X[0] + recv0[0][1] * C, d_send[1] + spush0[1][0] * C, C * sizeof(float), hipMemcpyDeviceToDevice));
    for (int r = 0; r < 2; ++r) {
        float got[2 * C];
        CHECK_GNX(gnx_spmm(main_g[r], NULL, NULL, d_X[r], C, C, (const float *)d_H + 2 * r * C, C, 0.9f, 0.1f, GNX_ACT_NONE, d_out[r], C, NULL));
        CHECK_HIP(hipMemcpy(got, d_out[r], sizeof got, hipMemcpyDeviceToHost));
        for (int i = 0; i < 2 * C; ++i)
            if (fabsf(got[i] - want[2 * r * C + i]) > 1e-5f * fabsf(want[2 * r * C + i])) { printf("block %d element %d: got %g want %g\n", r, i, got[i], want[2 * r * C + i]); return 23; }
    }
    if (gnx_halo_exchange(plan[0], GNX_HALO_ALL, NULL, d_send[0], d_X[0], C, NULL) != GNX_ERR_INVALID) { printf("NULL communicator not refused\n"); return 24; }
    if (gnx_halo_pack(plan[0], 7, d_X[0], C, C, d_send[0], C, NULL) != GNX_ERR_INVALID) { printf("invalid part not refused\n"); return 25; }
    /* no RCCL in this process and none bound: the exchange must say so, not load a library of its own */
    if (gnx_halo_exchange(plan[0], GNX_HALO_ALL, (void *)plan, d_send[0], d_X[0], C, NULL) != GNX_ERR_UNSUPPORTED) { printf("exchange without RCCL: %s\n", gnx_last_error()); return 26; }
    for (int r = 0; r < 2; ++r) {
        CHECK_GNX(gnx_halo_plan_destroy(plan[r]));
        CHECK_GNX(gnx_graph_destroy(main_g[r]));
        if (push_g[r]) CHECK_GNX(gnx_graph_destroy(push_g[r]));
    }
    CHECK_GNX(gnx_graph_destroy(whole));
    return 0;
}

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
    { int rc2 = two_block_plan(); if (rc2 != 0) return rc2; }
    printf("C ABI OK (version %d, kernel path verified: %g %g %g)\n", gnx_version(), out[0], out[1], out[2]);
    return 0;
}

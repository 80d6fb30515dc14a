/* gnx_halo_pack + gnx_halo_exchange over a REAL RCCL communicator, from a plain C program (no Python, no torch).
 *
 *   c_abi_rccl                      one rank on one GPU: the block lists ITSELF as its only peer (loop-back region after the
 *                                   local rows), so a one-rank communicator carries real ncclSend / ncclRecv pairs -- what a
 *                                   one-GPU box can execute.  Checks both halves of the message (pulled rows, pushed partial
 *                                   sums) in one group and as two groups, the bound entry points and the in-process lookup.
 *   c_abi_rccl 2 RANK IDFILE        two processes, GPU = RANK: the two vertex blocks of the path graph P4 of c_abi_smoke.c
 *                                   (block 0 is sent a pushed partial sum, block 1 pulls a row) exchanged over xGMI / PCIe;
 *                                   each rank's fused step must equal the step on the whole graph.  Rank 0 writes the
 *                                   ncclUniqueId to IDFILE, rank 1 waits for it.  For a box with two GPUs.
 * Build: gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/c_abi_rccl.c -Lgnn-tf_amd/lib -lgnx
 *        -L/opt/rocm/lib -lamdhip64 -lrccl -lm */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "gnx.h"

#define CHECK_HIP(e) do { hipError_t _s = (e); if (_s != hipSuccess) { printf("hip error %d at line %d\n", (int)_s, __LINE__); return 2; } } while (0)
#define CHECK_GNX(e) do { int _s = (e); if (_s != GNX_OK) { printf("gnx error %d: %s (line %d)\n", _s, gnx_last_error(), __LINE__); return 3; } } while (0)
#define CHECK_NCCL(e) do { ncclResult_t _s = (e); if (_s != ncclSuccess) { printf("rccl error %d at line %d\n", (int)_s, __LINE__); return 4; } } while (0)

static int upload(const void *src, size_t bytes, void **dst) {
    if (hipMalloc(dst, bytes) != hipSuccess) return 1;
    return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) != hipSuccess;
}

/* One rank, loop-back: 4 local rows, region(self) = [2 pulled rows | 1 pushed partial sum]. */
static int loop_back(void) {
    enum { C = 3, NL = 4 };
    CHECK_HIP(hipSetDevice(0));
    ncclUniqueId id;
    ncclComm_t comm;
    CHECK_NCCL(ncclGetUniqueId(&id));
    CHECK_NCCL(ncclCommInitRank(&comm, 1, id, 0));
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    float H[NL * C];
    for (int i = 0; i < NL * C; ++i) H[i] = 1.f + 0.25f * i;
    const int32_t pull_src[2] = {3, 1};                                       /* pulled rows: local rows 3 and 1, in this order */
    const int64_t push_idx[6] = {0, 0, 0, 2, 0, 3};                           /* pushed sum: 0.5 H[0] + 2 H[2] - H[3] */
    const float push_val[3] = {0.5f, 2.f, -1.f};
    void *d_src, *d_pi, *d_pv;
    if (upload(pull_src, sizeof pull_src, &d_src) || upload(push_idx, sizeof push_idx, &d_pi) || upload(push_val, sizeof push_val, &d_pv)) return 20;
    gnx_graph_t push_g = NULL;
    CHECK_GNX(gnx_graph_create_coo(1, NL, 3, (const int64_t *)d_pi, (const float *)d_pv, NULL, &push_g));
    const int64_t two[1] = {2}, one[1] = {1};
    gnx_halo_plan_t plan = NULL;
    CHECK_GNX(gnx_halo_plan_create(1, 0, NL, two, one, two, one, (const int32_t *)d_src, push_g, &plan));
    int64_t n_buf, local0, n_send, n_send_pull, recv0[1], spull0[1], spush0[1];
    CHECK_GNX(gnx_halo_plan_layout(plan, &n_buf, &local0, &n_send, &n_send_pull, recv0, spull0, spush0));
    if (n_buf != NL + 3 || local0 != 0 || n_send != 3 || n_send_pull != 2 || recv0[0] != NL || spull0[0] != 0 || spush0[0] != 2) { printf("layout\n"); return 21; }
    float *d_X, *d_send;
    CHECK_HIP(hipMalloc((void **)&d_X, (size_t)n_buf * C * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&d_send, (size_t)n_send * C * sizeof(float)));
    float want[3 * C];
    for (int c = 0; c < C; ++c) {
        want[0 * C + c] = H[3 * C + c];
        want[1 * C + c] = H[1 * C + c];
        want[2 * C + c] = 0.5f * H[0 * C + c] + 2.f * H[2 * C + c] - H[3 * C + c];
    }
    /* variant 0: entry points found in the process (this program links librccl), both halves in one group;
     * variant 1: entry points bound by the caller, the two halves as two groups (pulled rows first) */
    for (int variant = 0; variant < 2; ++variant) {
        CHECK_HIP(hipMemsetAsync(d_X, 0, (size_t)n_buf * C * sizeof(float), stream));
        CHECK_HIP(hipMemcpyAsync(d_X, H, sizeof H, hipMemcpyHostToDevice, stream));
        if (variant == 0) {
            CHECK_GNX(gnx_halo_pack(plan, GNX_HALO_ALL, d_X, C, C, d_send, C, stream));
            CHECK_GNX(gnx_halo_exchange(plan, GNX_HALO_ALL, comm, d_send, d_X, C, stream));
        } else {
            CHECK_GNX(gnx_halo_bind_rccl((void *)ncclGroupStart, (void *)ncclGroupEnd, (void *)ncclSend, (void *)ncclRecv));
            CHECK_GNX(gnx_halo_pack(plan, GNX_HALO_PULL, d_X, C, C, d_send, C, stream));
            CHECK_GNX(gnx_halo_exchange(plan, GNX_HALO_PULL, comm, d_send, d_X, C, stream));
            CHECK_GNX(gnx_halo_pack(plan, GNX_HALO_PUSH, d_X, C, C, d_send, C, stream));
            CHECK_GNX(gnx_halo_exchange(plan, GNX_HALO_PUSH, comm, d_send, d_X, C, stream));
        }
        CHECK_HIP(hipStreamSynchronize(stream));
        float got[3 * C];
        CHECK_HIP(hipMemcpy(got, d_X + recv0[0] * C, sizeof got, hipMemcpyDeviceToHost));
        for (int i = 0; i < 3 * C; ++i)
            if (got[i] != want[i]) { printf("variant %d, region element %d: got %g want %g\n", variant, i, got[i], want[i]); return 22; }
    }
    CHECK_GNX(gnx_halo_bind_rccl(NULL, NULL, NULL, NULL));
    CHECK_GNX(gnx_halo_plan_destroy(plan));
    CHECK_GNX(gnx_graph_destroy(push_g));
    CHECK_NCCL(ncclCommDestroy(comm));
    printf("RCCL loop-back OK (pulled rows and pushed sums through ncclSend / ncclRecv, one group and two groups)\n");
    return 0;
}

/* Two ranks, one GPU each: the two blocks of P4 (see c_abi_smoke.c for the plan). */
static int two_ranks(int rank, const char *idfile) {
    enum { C = 2 };
    CHECK_HIP(hipSetDevice(rank));
    ncclUniqueId id;
    if (rank == 0) {
        CHECK_NCCL(ncclGetUniqueId(&id));
        char tmp[4096];
        snprintf(tmp, sizeof tmp, "%s.tmp", idfile);
        FILE *f = fopen(tmp, "wb");
        if (!f || fwrite(&id, sizeof id, 1, f) != 1) return 30;
        fclose(f);
        if (rename(tmp, idfile) != 0) return 31;
    } else {
        FILE *f = NULL;
        for (int tries = 0; tries < 600 && !(f = fopen(idfile, "rb")); ++tries) usleep(100000);
        if (!f || fread(&id, sizeof id, 1, f) != 1) return 32;
        fclose(f);
    }
    ncclComm_t comm;
    CHECK_NCCL(ncclCommInitRank(&comm, 2, id, rank));
    const float w01 = 1.f / sqrtf(2.f), w12 = 0.5f, w23 = 1.f / sqrtf(2.f);
    const float H[4 * C] = {1.f, 10.f, 2.f, 20.f, 3.f, 30.f, 4.f, 40.f};
    /* expected rows of one fused step on the whole graph: out = 0.9 A_hat H + 0.1 H */
    float want[4 * C];
    for (int c = 0; c < C; ++c) {
        want[0 * C + c] = 0.9f * (w01 * H[1 * C + c]) + 0.1f * H[0 * C + c];
        want[1 * C + c] = 0.9f * (w01 * H[0 * C + c] + w12 * H[2 * C + c]) + 0.1f * H[1 * C + c];
        want[2 * C + c] = 0.9f * (w12 * H[1 * C + c] + w23 * H[3 * C + c]) + 0.1f * H[2 * C + c];
        want[3 * C + c] = 0.9f * (w23 * H[2 * C + c]) + 0.1f * H[3 * C + c];
    }
    const int64_t m_idx[2][6] = {{0, 1, 1, 0, 1, 2}, {0, 0, 0, 2, 1, 1}};
    const float m_val[2][3] = {{w01, w01, 1.f}, {w12, w23, w23}};
    const int32_t pull0_src[1] = {1};
    const int64_t push1_idx[2] = {0, 0};
    const float push1_val[1] = {w12};
    const int64_t recv_pull[2][2] = {{0, 0}, {1, 0}}, recv_push[2][2] = {{0, 1}, {0, 0}};
    const int64_t send_pull[2][2] = {{0, 1}, {0, 0}}, send_push[2][2] = {{0, 0}, {1, 0}};
    void *d_mi, *d_mv, *d_ps = NULL, *d_pi, *d_pv, *d_H0;
    if (upload(m_idx[rank], sizeof m_idx[rank], &d_mi) || upload(m_val[rank], sizeof m_val[rank], &d_mv) || upload(H + 2 * rank * C, 2 * C * sizeof(float), &d_H0)) return 33;
    gnx_graph_t main_g = NULL, push_g = NULL;
    CHECK_GNX(gnx_graph_create_coo(2, 3, 3, (const int64_t *)d_mi, (const float *)d_mv, NULL, &main_g));
    if (rank == 0) { if (upload(pull0_src, sizeof pull0_src, &d_ps)) return 34; }
    else {
        if (upload(push1_idx, sizeof push1_idx, &d_pi) || upload(push1_val, sizeof push1_val, &d_pv)) return 35;
        CHECK_GNX(gnx_graph_create_coo(1, 2, 1, (const int64_t *)d_pi, (const float *)d_pv, NULL, &push_g));
    }
    gnx_halo_plan_t plan = NULL;
    CHECK_GNX(gnx_halo_plan_create(2, rank, 2, recv_pull[rank], recv_push[rank], send_pull[rank], send_push[rank], (const int32_t *)d_ps, push_g, &plan));
    int64_t n_buf, local0, n_send;
    CHECK_GNX(gnx_halo_plan_layout(plan, &n_buf, &local0, &n_send, NULL, NULL, NULL, NULL));
    float *d_X, *d_send, *d_out;
    CHECK_HIP(hipMalloc((void **)&d_X, (size_t)n_buf * C * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&d_send, (size_t)(n_send > 0 ? n_send : 1) * C * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&d_out, 2 * C * sizeof(float)));
    CHECK_HIP(hipMemset(d_X, 0, (size_t)n_buf * C * sizeof(float)));
    CHECK_HIP(hipMemcpy(d_X + local0 * C, H + 2 * rank * C, 2 * C * sizeof(float), hipMemcpyHostToDevice));
    CHECK_GNX(gnx_halo_pack(plan, GNX_HALO_ALL, d_X, C, C, d_send, C, NULL));
    CHECK_GNX(gnx_halo_exchange(plan, GNX_HALO_ALL, comm, d_send, d_X, C, NULL));
    CHECK_GNX(gnx_spmm(main_g, NULL, NULL, d_X, C, C, (const float *)d_H0, C, 0.9f, 0.1f, GNX_ACT_NONE, d_out, C, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    float got[2 * C];
    CHECK_HIP(hipMemcpy(got, d_out, sizeof got, hipMemcpyDeviceToHost));
    for (int i = 0; i < 2 * C; ++i)
        if (fabsf(got[i] - want[2 * rank * C + i]) > 1e-5f * fabsf(want[2 * rank * C + i])) { printf("rank %d element %d: got %g want %g\n", rank, i, got[i], want[2 * rank * C + i]); return 36; }
    CHECK_GNX(gnx_halo_plan_destroy(plan));
    CHECK_NCCL(ncclCommDestroy(comm));
    printf("RCCL two-rank exchange OK on rank %d\n", rank);
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 4 && atoi(argv[1]) == 2) return two_ranks(atoi(argv[2]), argv[3]);
    return loop_back();
}

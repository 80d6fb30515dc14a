// The GCNII layer (gcn.py:7-27,54-74) in one launch on gfx950: SpMM + residual mix + the C x C transform on the matrix cores.
// Shares the gathers' device code and the long-row path with gnx_spmm.hip (gnx_spmm_device.h, gnx::launch_long_rows).
#include "gnx_spmm_device.h"

namespace {

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

}  // namespace

extern "C" {

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
        launch_long_rows(p, s);
        rc = dense_rows(rows_at, C, m.n_long, C, d_M, ldm, C, nullptr, act, m.long_rows, m.long_rows, d_out, C, s);
        if (rc != GNX_OK) return rc;
    }
    GNX_HIP(hipGetLastError());
    return GNX_OK;
}

}  // extern "C"

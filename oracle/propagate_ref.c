/* CPU port of gnntf's propagation hot path, in plain C + OpenMP.
 *
 * TEST INFRASTRUCTURE ONLY: built by __graft_entry__.build() into oracle/_build/, loaded
 * only by tests/ and by bench.py's `cpu_baseline` leg.  Never linked into libgnx.so and
 * never called by the product path.
 *
 * PARITY UNPINNED (see oracle/gnntf_oracle.py header): this is a restatement, validated
 * against the numpy oracle in tests/test_oracle_c.py, not against reference outputs.
 *
 * It follows the reference op for op (paths relative to /root/reference):
 *   gnntf/core/gnn/gnn.py:36-50           get_adjacency -- re-normalises on EVERY call
 *   gnntf/core/gnn/architectures/filter.py:17-22   PPRIteration.__forward__
 * i.e. each of the K iterations does: column sums (gnn.py:41) -> D = divide_no_nan(1,
 * sqrt(d)) (gnn.py:41) -> v_ij = D[i] v_ij D[j] (gnn.py:42) -> P = A_hat.H (filter.py:19)
 * -> H' = P*(1-a) + H0*a (filter.py:20-21).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* gnn.py:41-42 on a CSR (rows sorted, duplicates allowed): writes normalised values. */
void oracle_normalize_symmetric(int64_t n, const int64_t *rowptr, const int32_t *colidx,
                                const float *vals, float *vals_out, float *colsum /* [n] scratch */) {
    memset(colsum, 0, (size_t)n * sizeof(float));
    for (int64_t i = 0; i < n; ++i)                      /* tf.sparse.reduce_sum(axis=0) */
        for (int64_t e = rowptr[i]; e < rowptr[i + 1]; ++e) colsum[colidx[e]] += vals[e];
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < n; ++j) {                    /* divide_no_nan(1, sqrt(d)) */
        float s = sqrtf(colsum[j]);
        colsum[j] = (s != 0.0f) ? 1.0f / s : 0.0f;
    }
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t i = 0; i < n; ++i)
        for (int64_t e = rowptr[i]; e < rowptr[i + 1]; ++e)
            vals_out[e] = colsum[i] * vals[e] * colsum[colidx[e]];
}

/* filter.py:19-21: out = (A_hat . H)*(1-a) + H0*a */
void oracle_ppr_step(int64_t n, const int64_t *rowptr, const int32_t *colidx, const float *vals,
                     const float *H, const float *H0, float a, int64_t C, float *out) {
    const float beta = (float)(1.0 - (double)a);
#pragma omp parallel
    {
        float *acc = (float *)malloc((size_t)C * sizeof(float));
#pragma omp for schedule(dynamic, 256)
        for (int64_t i = 0; i < n; ++i) {
            for (int64_t c = 0; c < C; ++c) acc[c] = 0.0f;
            for (int64_t e = rowptr[i]; e < rowptr[i + 1]; ++e) {
                const float w = vals[e];
                const float *x = H + (int64_t)colidx[e] * C;
                for (int64_t c = 0; c < C; ++c) acc[c] += w * x[c];
            }
            const float *h0 = H0 + i * C;
            float *o = out + i * C;
            for (int64_t c = 0; c < C; ++c) o[c] = acc[c] * beta + h0[c] * a;
        }
        free(acc);
    }
}

/* Bounded sample for bench.py's cpu_baseline: ONE iteration (normalise + step) restricted to
 * the first n_sample rows (their gathers still span all of H).  The normalisation pass runs
 * over the whole graph, as the reference's does.  Returns 0 / -1. */
int oracle_sample_iteration(int64_t n, int64_t n_sample, const int64_t *rowptr, const int32_t *colidx,
                            const float *raw_vals, const float *H, const float *H0, float a, int64_t C,
                            float *out /* [n_sample, C] */) {
    const int64_t nnz = rowptr[n];
    float *nv = (float *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(float));
    float *cs = (float *)malloc((size_t)(n > 0 ? n : 1) * sizeof(float));
    if (!nv || !cs) { free(nv); free(cs); return -1; }
    oracle_normalize_symmetric(n, rowptr, colidx, raw_vals, nv, cs);
    oracle_ppr_step(n_sample, rowptr, colidx, nv, H, H0, a, C, out);
    free(nv); free(cs);
    return 0;
}

/* The same normalisation with every pass spread over the OpenMP threads (column sums through atomic adds, so the
 * float summation order -- and the last bits -- vary from run to run).  Only bench.py's cpu_baseline uses it:
 * it lets the CPU figure renormalise per iteration like the reference without a single-threaded pass inside. */
void oracle_normalize_symmetric_par(int64_t n, const int64_t *rowptr, const int32_t *colidx,
                                    const float *vals, float *vals_out, float *colsum /* [n] scratch */) {
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < n; ++j) colsum[j] = 0.0f;
#pragma omp parallel for schedule(dynamic, 4096)
    for (int64_t i = 0; i < n; ++i)
        for (int64_t e = rowptr[i]; e < rowptr[i + 1]; ++e) {
#pragma omp atomic
            colsum[colidx[e]] += vals[e];
        }
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < n; ++j) {
        float s = sqrtf(colsum[j]);
        colsum[j] = (s != 0.0f) ? 1.0f / s : 0.0f;
    }
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t i = 0; i < n; ++i)
        for (int64_t e = rowptr[i]; e < rowptr[i + 1]; ++e)
            vals_out[e] = colsum[i] * vals[e] * colsum[colidx[e]];
}

/* oracle_sample_iteration with the parallel normalisation; renorm == 0 skips the normalisation and treats
 * raw_vals as A_hat's values (the SpMM + mix alone, which is what the GPU figure times). */
int oracle_sample_iteration_par(int64_t n, int64_t n_sample, const int64_t *rowptr, const int32_t *colidx,
                                const float *raw_vals, const float *H, const float *H0, float a, int64_t C,
                                float *out /* [n_sample, C] */, int renorm) {
    if (!renorm) {
        oracle_ppr_step(n_sample, rowptr, colidx, raw_vals, H, H0, a, C, out);
        return 0;
    }
    const int64_t nnz = rowptr[n];
    float *nv = (float *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(float));
    float *cs = (float *)malloc((size_t)(n > 0 ? n : 1) * sizeof(float));
    if (!nv || !cs) { free(nv); free(cs); return -1; }
    oracle_normalize_symmetric_par(n, rowptr, colidx, raw_vals, nv, cs);
    oracle_ppr_step(n_sample, rowptr, colidx, nv, H, H0, a, C, out);
    free(nv); free(cs);
    return 0;
}

/* K iterations, renormalising each time like the reference.  `out` and `work` are [n, C];
 * the result lands in `out`.  Returns 0, or -1 on allocation failure. */
int oracle_appnp_propagate(int64_t n, const int64_t *rowptr, const int32_t *colidx,
                           const float *raw_vals, const float *H0, float a, int K, int64_t C,
                           float *out, float *work) {
    const int64_t nnz = rowptr[n];
    float *nv = (float *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(float));
    float *cs = (float *)malloc((size_t)(n > 0 ? n : 1) * sizeof(float));
    if (!nv || !cs) { free(nv); free(cs); return -1; }
    const float *src = H0;
    /* ping-pong so that the K-th result is written into `out` */
    float *bufs[2] = { (K % 2) ? out : work, (K % 2) ? work : out };
    for (int k = 0; k < K; ++k) {
        oracle_normalize_symmetric(n, rowptr, colidx, raw_vals, nv, cs);
        float *dst = bufs[k % 2];
        oracle_ppr_step(n, rowptr, colidx, nv, src, H0, a, C, dst);
        src = dst;
    }
    if (K == 0) memcpy(out, H0, (size_t)n * C * sizeof(float));
    free(nv); free(cs);
    return 0;
}

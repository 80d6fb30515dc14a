/* gnx.h -- C ABI of libgnx.so: gnntf's sparse propagation hot path on MI355X (gfx950).
 *
 * The reference (gnntf 0.0.20, paths relative to /root/reference) has NO FFI of its own:
 * the path is Python calling TensorFlow eager ops.  Each entry point below therefore
 * replaces a TensorFlow call site (or a short run of them) in the reference's Python, and
 * is what a ctypes binding inside gnntf would bind (INTEGRATION.md shows that binding).
 *
 * Conventions
 *   - every pointer named d_* is a DEVICE pointer (HBM of the current HIP device); the
 *     library borrows it for the duration of the call and never frees it;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); all work is
 *     enqueued on it and the call returns without waiting, except the graph constructors,
 *     which synchronise that stream (they read sizes back to size their allocations);
 *   - return value: 0 = ok, < 0 = error code below; gnx_last_error() then holds a
 *     thread-local message.  Nothing throws across this boundary;
 *   - a gnx_graph_t owns its device-side index arrays (hipMalloc) until gnx_graph_destroy.
 *     ONE STREAM PER HANDLE AT A TIME: a handle keeps per-handle scratch (the long-row partial slab, which is
 *     re-allocated when a wider C arrives, the transposed-value and degree scratch), so two streams or threads
 *     launching on the same handle concurrently race on it.  Different handles are independent;
 *   - features are row-major float32, leading dimension (`ld*`, in elements) given per call.
 */
#ifndef GNX_H
#define GNX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gnx_graph *gnx_graph_t;
typedef struct gnx_halo_plan *gnx_halo_plan_t;

enum { GNX_OK = 0, GNX_ERR_INVALID = -1, GNX_ERR_HIP = -2, GNX_ERR_ALLOC = -3, GNX_ERR_UNSUPPORTED = -4 };

/* `normalized` argument of GNN.get_adjacency (gnntf/core/gnn/gnn.py:36,40-47) */
enum { GNX_NORM_NONE = 0, GNX_NORM_SYMMETRIC = 1, GNX_NORM_BIPARTITE = 2 };
/* `add_eye` argument of GNN.get_adjacency (gnn.py:36,38-39,48-49) */
enum { GNX_EYE_NONE = 0, GNX_EYE_BEFORE = 1, GNX_EYE_AFTER = 2 };
/* epilogue activation: identity (filter.py:8 default) or relu (gcn.py:78 default) */
enum { GNX_ACT_NONE = 0, GNX_ACT_RELU = 1 };
/* flag OR'ed into `act` of gnx_spmm / gnx_spmm_rows: rows without stored entries are NOT written (their output would be
 * alpha * H0[row], which is what an earlier iteration of a propagation loop already left there) -- and cost nothing: the kernels
 * that walk the rows in degree-binned order do not launch their slots at all, the one-wave-per-row kernels walk the ascending
 * list of the rows that have entries.  Ignored when a diagonal weight is given. */
enum { GNX_ACT_SKIP_EMPTY = 256 };

/* Thread-local message of the last failing call on this thread ("" if none). */
const char *gnx_last_error(void);
/* ABI version: major*10000 + minor*100 + patch.  GNX_ABI_VERSION is what THIS header describes; a client compares it with
 * gnx_version() of the library it loaded and refuses a mismatch in major or minor: entry points changed argument lists under
 * the same names between 0.2 and 0.3 (gnx_halo_plan_create / _layout / _pack / _exchange gained `part` and split pull / push
 * counts; gnx_gcnii_step's d_work became d_mixed), so a 0.2 client linked against a 0.3+ library passes shifted arguments.
 * 0.4 adds gnx_graph_reserve and changes no existing signature; 0.5 adds gnx_graph_set_row_window, 0.6 gnx_appnp_propagate_act,
 * likewise. */
#define GNX_ABI_VERSION 600
int gnx_version(void);

/* ---- graph construction ------------------------------------------------------------
 * Replaces tf.sparse.SparseTensor(indices, values, shape) built by graph2adj
 * (gnntf/core/gnn/graph_manipulation.py:24-31): d_indices is int64 [nnz, 2] row-major
 * (row, col) pairs -- UNSORTED, duplicates allowed -- d_values float32 [nnz].
 * The handle keeps every entry (sorted by (row, col), input order kept among duplicates,
 * so per-entry edge dropout follows layered.py:47-50) plus the coalesced CSR in which
 * duplicates are summed (what tf.sparse.sparse_dense_matmul / reduce_sum compute).
 * Fails with GNX_ERR_INVALID on an index outside the shape. */
int gnx_graph_create_coo(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t *d_indices,
                         const float *d_values, void *stream, gnx_graph_t *out);

/* Same from a ready CSR (rows ascending, columns ascending and unique inside a row):
 * d_rowptr int64 [n_rows+1], d_colidx int32 [nnz], d_values float32 [nnz].  Used for the
 * column-remapped shard of a vertex-partitioned graph. */
int gnx_graph_create_csr(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t *d_rowptr,
                         const int32_t *d_colidx, const float *d_values, void *stream, gnx_graph_t *out);

int gnx_graph_destroy(gnx_graph_t g);

/* Sizes: stored COO entries and coalesced (unique (row, col)) entries. */
int gnx_graph_info(gnx_graph_t g, int64_t *n_rows, int64_t *n_cols, int64_t *nnz_entries,
                   int64_t *nnz_coalesced);

/* Borrowed device pointers to the coalesced CSR (valid until destroy):
 * rowptr int64 [n_rows+1], colidx int32 [nnz_coalesced], raw summed values float32. */
int gnx_graph_csr(gnx_graph_t g, const int64_t **d_rowptr, const int32_t **d_colidx,
                  const float **d_raw_values);
/* Copies of the coalesced CSR into caller-owned device buffers (any may be NULL = skip):
 * rowptr int64 [n_rows+1], colidx int32 [nnz_coalesced], raw values float32 [nnz_coalesced],
 * rowidx int32 [nnz_coalesced] (the row of every coalesced entry).  Stream-ordered. */
int gnx_graph_export(gnx_graph_t g, int64_t *d_rowptr_out, int32_t *d_colidx_out, float *d_raw_values_out,
                     int32_t *d_rowidx_out, void *stream);

/* ---- normalisation: GNN.get_adjacency (gnn.py:36-50) --------------------------------
 * One call = sparse_dropout (layered.py:47-50; only when dropout_p > 0 -- the caller passes
 * 0 in eval mode) -> add_eye "before" -> symmetric / bipartite scaling by COLUMN sums with
 * divide_no_nan -> add_eye "after".  Writes the values of A_hat in coalesced-CSR order to
 * d_vals_out [nnz_coalesced].  The identity added by add_eye is returned as a per-row
 * diagonal weight in d_diag_out [n_rows] (pass it to gnx_spmm); d_diag_out may be NULL when
 * add_eye == GNX_EYE_NONE.  Dropout draws come from the counter RNG keyed by (seed,
 * stream_id, row, col, duplicate rank), so they do not depend on how the graph is sharded.
 * Requires a square graph for symmetric/bipartite. */
int gnx_graph_normalize(gnx_graph_t g, int normalized, int add_eye, float dropout_p, uint64_t seed,
                        uint64_t stream_id, float *d_vals_out, float *d_diag_out, void *stream);

/* gnx_graph_reserve: builds, NOW, what the compute entries otherwise build on first use, sized for feature rows of up to C floats:
 * the slab the long rows' partial sums go through (every entry), with GNX_RESERVE_TRANSPOSED the transposed structure (gnx_spmm_t,
 * gnx_spmm_dropped(transposed), the column sums of a training step), with GNX_RESERVE_K_LOOP the relabelled copy
 * gnx_appnp_propagate runs narrow widths on.  Those lazy builds allocate and synchronise, which a stream that is being captured
 * into a hipGraph must not see: a compute entry that would have to grow something under capture fails with
 * GNX_ERR_UNSUPPORTED and a message naming this call.  Reserve (or run the launch once eagerly) BEFORE capturing; replays then
 * touch no allocator.  ("an optional caller-provided workspace" -- the workspace stays owned by the handle, the
 * caller decides when it is sized.)  Nothing in the reference corresponds: TensorFlow eager allocates per op. */
enum { GNX_RESERVE_TRANSPOSED = 1, GNX_RESERVE_K_LOOP = 2 };
int gnx_graph_reserve(gnx_graph_t g, int64_t C, int flags, void *stream);

/* gnx_graph_set_row_window: the caller declares that ITS numbering of the vertices carries locality -- neighbours in the graph
 * are neighbours in the numbering (a community / breadth-first order of the dataset; gnntf's GNN(reorder="locality") computes one).
 * The propagation launches then take the rows in WINDOWS of `window_rows` consecutive ids -- inside a window still in the
 * degree-binned order that gives the rows sharing a wave equal lengths -- so that the rows in flight together gather from one
 * neighbourhood of H (at narrow widths a gather moves a whole 128-byte line for a 16..160-byte row: what counts is how often the
 * line is still in a cache), and gnx_appnp_propagate does NOT run narrow widths on its degree-relabelled copy, which would scatter
 * that neighbourhood.  0 restores the default (global degree bins).  Results are the same sums in the same per-row order: bitwise
 * those of the default order except where the default would have used the relabelled copy.  Rebuilds the handle's launch plan:
 * synchronises the DEVICE (work on any stream may still read the old plan's arrays, which are freed), not capturable, and
 * INVALIDATES every hipGraph captured on this handle before the call (a replay would read the freed arrays): capture again.
 * Nothing in the reference corresponds (TensorFlow's kernel walks the COO as stored). */
int gnx_graph_set_row_window(gnx_graph_t g, int64_t window_rows, void *stream);

/* gnx_graph_set_dropout_counter: from now on every dropout stream id used with this handle is `stream_id + *d_counter`
 * (d_counter: one uint64 in device memory, read by the kernels when they run; NULL switches it off).  This is what lets a
 * whole training step be captured ONCE in a hipGraph and replayed every epoch with fresh masks: the ids baked into the
 * captured launches stay fixed, the step's last node advances the counter (layered.py:47-50 draws new masks per call). */
int gnx_graph_set_dropout_counter(gnx_graph_t g, const uint64_t *d_counter);

/* gnx_graph_set_block: declares the handle to be ONE VERTEX BLOCK of a larger square graph (multi-GPU training with edge
 * dropout): its row r is global vertex row0_global + r, its column c is global vertex d_col_gid[c]
 * (int32 [n_cols], copied), and local row r is column row0_buf + r of the block's own column space (the [regions | local rows |
 * regions] buffer of gnx_halo_plan_layout).  From then on the dropout draws of gnx_graph_colsum(_streams) /
 * gnx_graph_scale_values / gnx_spmm_dropped are keyed by the GLOBAL (row, col) -- the masks equal the one-GPU masks whatever
 * the partition -- and gnx_spmm_dropped accepts the rectangular block: d_D is then [n_cols] (scales of every buffer column),
 * row r's own scale being d_D[row0_buf + r].  gnx_graph_colsum gives the block's PARTIAL column sums; the caller adds the
 * blocks up (halo columns go back to their owners) before gnx_degree_scale.  d_col_gid == NULL undoes the declaration. */
int gnx_graph_set_block(gnx_graph_t g, int64_t row0_global, int64_t row0_buf, const int32_t *d_col_gid, void *stream);

/* gnx_graph_normalize_t: the same normalisation, but the values are written in the order of the TRANSPOSED
 * structure (the order gnx_spmm_tv consumes).  The backward pass of a training step only needs A_hat^T, so it
 * regenerates the iteration's dropped adjacency straight into this order instead of permuting a CSR-order
 * array (what tf.GradientTape keeps alive, trainable.py:70-78, is recomputed here from the counter RNG). */
int gnx_graph_normalize_t(gnx_graph_t g, int normalized, int add_eye, float dropout_p, uint64_t seed,
                          uint64_t stream_id, float *d_vals_t_out, float *d_diag_out, void *stream);

/* The three steps separately, for vertex-partitioned graphs where column sums need a
 * cross-rank all-reduce in between:
 *   colsum: d_colsum_out[j] = sum_i v_ij over this handle's rows (tf.sparse.reduce_sum
 *           axis=0, gnn.py:41,44), v = dropped raw values; fixed summation order.
 *   degree_scale: in place d[j] <- divide_no_nan(1, sqrt(d[j] + eye)) (symmetric) or
 *           divide_no_nan(1, d[j] + eye) (bipartite); eye = 1 for add_eye "before".
 *   scale_values: d_vals_out[k] = row_scale[row_k] * v_k * col_scale[col_k]; either scale
 *           may be NULL (= 1).  (gnn.py:42,45) */
int gnx_graph_colsum(gnx_graph_t g, float dropout_p, uint64_t seed, uint64_t stream_id,
                     float *d_colsum_out, void *stream);
/* gnx_graph_colsum for n_streams consecutive dropout streams (first_stream, first_stream + 1, ...) in one pass over the
 * structure: the K iterations of a training step drop their edges independently but share everything else.
 * d_colsum_out is [n_streams, n_cols]; every row is bit for bit what gnx_graph_colsum gives for that stream. */
int gnx_graph_colsum_streams(gnx_graph_t g, float dropout_p, uint64_t seed, uint64_t first_stream, int n_streams,
                             float *d_colsum_out, void *stream);
int gnx_degree_scale(float *d_deg, int64_t n, int normalized, int add_eye_before, void *stream);
int gnx_graph_scale_values(gnx_graph_t g, float dropout_p, uint64_t seed, uint64_t stream_id,
                           const float *d_row_scale, const float *d_col_scale, float *d_vals_out,
                           void *stream);

/* ---- the hot path ---------------------------------------------------------------------
 * gnx_spmm: out[i,:] = act( beta * ( sum_j A[i,j] X[j,:] + diag[i] X[i,:] ) + alpha * H0[i,:] )
 *   - replaces tf.sparse.sparse_dense_matmul (filter.py:19, gcn.py:88) and, with
 *     beta = 1-a, alpha = a, the residual mix fused behind it (filter.py:20-21) and the
 *     optional activation (filter.py:22);
 *   - d_vals: values in coalesced-CSR order (from gnx_graph_normalize), NULL = raw values;
 *   - d_diag: per-row diagonal weight or NULL; d_H0 may be NULL (then alpha is ignored); ldh0 == 0 broadcasts
 *     ONE row of C values to every output row (a bias: gcn.py:89);
 *   - X is [n_cols, C], out and H0 are [n_rows, C]; out must not alias X.
 * gnx_spmm_t: the same with the transposed matrix (out is [n_cols, C], X and H0 index by
 *   the other side) -- the backward of gnx_spmm (what tf.GradientTape derives,
 *   trainable.py:70-78).  d_vals are still given in coalesced-CSR order of A. */
int gnx_spmm(gnx_graph_t g, const float *d_vals, const float *d_diag, const float *d_X, int64_t ldx,
             int64_t C, const float *d_H0, int64_t ldh0, float beta, float alpha, int act,
             float *d_out, int64_t ldo, void *stream);
int gnx_spmm_t(gnx_graph_t g, const float *d_vals, const float *d_diag, const float *d_X, int64_t ldx,
               int64_t C, const float *d_H0, int64_t ldh0, float beta, float alpha, int act,
               float *d_out, int64_t ldo, void *stream);

/* gnx_spmm_tv: gnx_spmm_t with the values ALREADY in transposed order (from gnx_graph_normalize_t or
 * gnx_graph_permute_values_t); skips the per-call permutation. */
int gnx_spmm_tv(gnx_graph_t g, const float *d_vals_t, const float *d_diag, const float *d_X, int64_t ldx,
                int64_t C, const float *d_H0, int64_t ldh0, float beta, float alpha, int act,
                float *d_out, int64_t ldo, void *stream);
/* d_vals_t_out[p] = d_vals[perm[p]]: CSR-order values -> transposed order (for a constant adjacency). */
int gnx_graph_permute_values_t(gnx_graph_t g, const float *d_vals, float *d_vals_t_out, void *stream);

/* gnx_spmm_scatter: gnx_spmm whose result row i is written to out[d_out_rows[i], :] (int32 [n_rows], a
 * permutation).  Lets the LAST iteration of a propagation over a relabelled graph put the rows straight back
 * into the caller's order instead of paying a separate un-permute pass. */
int gnx_spmm_scatter(gnx_graph_t g, const float *d_vals, const float *d_diag, const float *d_X, int64_t ldx,
                     int64_t C, const float *d_H0, int64_t ldh0, float beta, float alpha, int act,
                     const int32_t *d_out_rows, float *d_out, int64_t ldo, void *stream);

/* gnx_spmm_dropped: gnx_spmm (transposed == 0) or gnx_spmm_tv (transposed != 0) over the dropped + symmetrically
 * re-normalised adjacency of ONE training iteration WITHOUT materialising its values: every entry's weight
 * (D[row] * dropout(raw)) * D[col] is produced inside the kernel from the counter RNG -- the same arithmetic, bit for bit, as
 * gnx_graph_normalize(g, GNX_NORM_SYMMETRIC, GNX_EYE_NONE, dropout_p, seed, stream_id, ...) followed by gnx_spmm.
 * d_D [n]: the degree scales of that iteration = gnx_graph_colsum(g, dropout_p, seed, stream_id, d_D) then
 * gnx_degree_scale(d_D, n, GNX_NORM_SYMMETRIC, 0).  Saves the nnz-sized value array and the pass that writes it
 * (layered.py:47-50 + gnn.py:41-42 happen in the SpMM's value fetch).  PRECONDITION: finite d_X and d_D -- a dropped entry is
 * skipped (its row of d_X is not read), while the two-call form multiplies it by an explicit zero, so a non-finite row behind a
 * dropped entry, or a NaN scale, turns the two-call result into NaN and not this one.  Square graphs or vertex blocks (gnx_graph_set_block);
 * GNX_ERR_UNSUPPORTED when the COO held
 * duplicate entries (their per-entry dropout needs the entry lists: use gnx_graph_normalize). */
int gnx_spmm_dropped(gnx_graph_t g, const float *d_D, float dropout_p, uint64_t seed, uint64_t stream_id, int transposed,
                     const float *d_X, int64_t ldx, int64_t C, const float *d_H0, int64_t ldh0, float beta, float alpha,
                     int act, float *d_out, int64_t ldo, void *stream);

/* gnx_spmm_dropped_chained: the forward gnx_spmm_dropped for a LOOP of training iterations.  The weight of an entry needs its
 * column's degree scale, a random 4-byte gather per entry (0.7 of the 3.8 ms of a launch at C = 64); in a loop the previous
 * iteration can deliver it with the row instead: with d_D_next != NULL every finished row is multiplied by d_D_next[row]
 * (= the NEXT iteration's scale of that vertex as a column) on its way out, and with x_prescaled != 0 the kernel takes the rows
 * of d_X as carrying their column scale already.  Iteration 0 runs with x_prescaled = 0, the last one with d_D_next = NULL; the
 * H0 mix term is the plain one throughout.  Same masks and the same value as K gnx_spmm_dropped calls up to float32 rounding
 * (the scale is applied to the gathered row instead of the weight). */
int gnx_spmm_dropped_chained(gnx_graph_t g, const float *d_D, float dropout_p, uint64_t seed, uint64_t stream_id, int x_prescaled,
                             const float *d_D_next, const float *d_X, int64_t ldx, int64_t C, const float *d_H0, int64_t ldh0,
                             float beta, float alpha, int act, float *d_out, int64_t ldo, void *stream);

/* One BACKWARD training iteration inside a loop -- what tf.GradientTape derives for K PPRIteration layers (trainable.py:70-78 over
 * filter.py:19-21): g_k = (1-a) A_k^T g_{k+1}, dH0 = g_0 + a (g_1 + ... + g_K).  Walks the TRANSPOSED structure with the weights
 * of dropout stream `stream_id` made inside the kernel, acc[r] = sum_c A_k[c][r] X[c], and writes TWO results:
 *     S_out[r] = s_beta * acc[r] + s_alpha * S_in[r]          the running sum dH0 is built in (S_in may be S_out: in place)
 *     Y_out[r] = y_beta * acc[r] * D_next[r]                  g_k carrying the column scale of the NEXT step (stream_id - 1), or
 *                                                             plain when d_D_next is NULL; skipped when d_Y_out is NULL
 * With x_prescaled the rows of X carry their own scale D[c] already (the previous call's Y_out), so no per-entry gather of a
 * column scale is left -- in the un-chained form (gnx_spmm_dropped, transposed) that gather fetches a 128-byte line per kept
 * entry (profiles/NOTES.md round 4: +4.5 GB per launch at config 4, C = 64).  The loop: X = upstream gradient, S_in = X,
 * s_alpha = a for the first call; afterwards X = the previous Y_out, S_in = S_out, s_alpha = 1; s_beta = a (1-a), y_beta = 1-a;
 * the last call (stream 0) adds g_0 whole: s_beta = 1-a, no Y_out.  Square stand-alone graphs without duplicate entries.
 * act: GNX_ACT_NONE, or GNX_ACT_SKIP_EMPTY (with S_in == S_out) for every call but the first of a loop: rows without entries
 * contribute g_k = 0, so their sum stays what the first call made it and their Y row is never gathered -- they are left untouched
 * (honoured only on graphs where no entry references a row without entries; gnx_spmm_dropped_chained takes the same flag for
 * every iteration but the last of a forward loop).
 * Finite operands only (a dropped entry's row is not gathered), as for gnx_spmm_dropped. */
int gnx_spmm_dropped_back(gnx_graph_t g, const float *d_D, float dropout_p, uint64_t seed, uint64_t stream_id, int x_prescaled,
                          const float *d_D_next, const float *d_X, int64_t ldx, int64_t C, const float *d_S_in, int64_t lds_in,
                          float s_alpha, float s_beta, float *d_S_out, int64_t lds_out, float y_beta, float *d_Y_out, int64_t ldy,
                          int act, void *stream);

/* gnx_spmm_rows: the fused step for a handle that holds only a SUBSET of the output rows (the interior or the
 * boundary rows of a vertex block, compacted): result row r lands in out[d_rows[r], :] and mixes in
 * H0[d_rows[r], :] (d_rows int32 [n_rows of the handle]; out and H0 are the full-height matrices).  Same
 * arithmetic per row as gnx_spmm (filter.py:19-21); no diagonal term. */
int gnx_spmm_rows(gnx_graph_t g, const float *d_vals, const float *d_X, int64_t ldx, int64_t C, const float *d_H0,
                  int64_t ldh0, float beta, float alpha, int act, const int32_t *d_rows, float *d_out, int64_t ldo,
                  void *stream);

/* One PPRIteration.__forward__ (filter.py:17-22) with a fixed adjacency:
 * out = act( (A_hat . H)*(1-a) + H0*a ).  Thin wrapper over gnx_spmm. */
int gnx_ppr_step(gnx_graph_t g, const float *d_vals, const float *d_diag, const float *d_H,
                 const float *d_H0, float a, int64_t C, int act, float *d_out, void *stream);

/* The K-iteration loop of APPNP in eval mode (filter.py:34-35 with a constant A_hat):
 * H <- H0; repeat K times H <- (A_hat . H)*(1-a) + H0*a.  All matrices [n, C] contiguous
 * (square graph).  d_work is a caller-provided scratch [n, C]; the result lands in d_out.
 * d_out, d_work and d_H0 must be distinct buffers.  Two things the loop does that K separate gnx_ppr_step calls do not:
 *   - rows without stored entries (a * H0 after every iteration) are computed when each buffer is first a destination and left
 *     alone afterwards (GNX_ACT_SKIP_EMPTY) -- same bits, fewer bytes; when no entry references such a row (every symmetric
 *     pattern) they are written into d_out only and d_work never receives them (its rows of that kind stay as the caller left them);
 *   - for C <= 16 on graphs of at least 2^20 vertices (no diagonal) the iterations run on a degree-relabelled copy of the
 *     matrix (heaviest vertices first; inside a degree bin the vertices follow their most popular neighbour's rank) that the
 *     handle builds on first use (+ about 12 bytes per entry and an [n, C] scratch, owned by the handle): H0 is
 *     permuted on the way in, the last iteration scatters back into the caller's row order; a row's columns are then summed in
 *     the relabelled order, so the result equals the step-by-step one up to float32 rounding. */
int gnx_appnp_propagate(gnx_graph_t g, const float *d_vals, const float *d_diag, const float *d_H0,
                        float a, int K, int64_t C, float *d_out, float *d_work, void *stream);

/* The same loop with the reference's per-iteration activation (filter.py:22,28,35: APPNP hands its ``activation`` to every
 * PPRIteration): H <- act((A_hat . H)*(1-a) + H0*a), act = GNX_ACT_NONE (= gnx_appnp_propagate) or GNX_ACT_RELU, applied in every
 * iteration's epilogue -- bit for bit what K gnx_ppr_step calls with that act return (below the relabelling threshold).  A row
 * without entries is act(a * H0) after every iteration, so the settled-row rule above holds unchanged. */
int gnx_appnp_propagate_act(gnx_graph_t g, const float *d_vals, const float *d_diag, const float *d_H0,
                            float a, int K, int64_t C, int act, float *d_out, float *d_work, void *stream);

/* One GCNIILayer.__forward__ (gnntf/core/gnn/architectures/gcn.py:22-27) with a fixed adjacency:
 *   out = act( ((A_hat . H)*(1-a) + H0*a) . M ),   M = (1-b) I + b W  given by the caller as a [C, C] matrix (ldm).
 * For C in {16, 32, 64} (16-byte aligned buffers) this is ONE launch for all rows of at most 512 entries: the mixed rows
 * T = (A_hat . H)*(1-a) + H0*a stay in LDS and meet M on the matrix cores (v_mfma_f32_16x16x4_f32, exact float32) before the only
 * store; hub rows go through the long-row kernels and the dense kernel.
 * d_mixed [n, C]: NULL in inference (T never reaches HBM).  In training pass a buffer: the same launch also writes T there -- what
 * tf.GradientTape keeps for dM = T^T g (trainable.py:70-78) -- so T is written once and not read back for the transform.
 * Other widths NEED d_mixed: they run gnx_spmm into it and then gnx_dense.  All matrices contiguous [n, C]; square graph. */
int gnx_gcnii_step(gnx_graph_t g, const float *d_vals, const float *d_H, const float *d_H0, float a, int64_t C,
                   const float *d_M, int64_t ldm, int act, float *d_out, float *d_mixed, void *stream);

/* ---- the dense ends of the path (matrix cores) -----------------------------------------------------------------------
 * gnx_dense: out = act(X . W + bias) -- Dense.__forward__ (gnntf/core/nn/layers.py:135-136) and the transform of
 * GCNLayer (gcn.py:89).  X [n, F] (ldx), W [F, O] (ldw), bias [O] or NULL, out [n, O] (ldo); float32 in and out, float32
 * accumulate on v_mfma_f32_16x16x4_f32.  out must not alias X. */
int gnx_dense(const float *d_X, int64_t ldx, int64_t n, int64_t F, const float *d_W, int64_t ldw, int64_t O,
              const float *d_bias, int act, float *d_out, int64_t ldo, void *stream);

/* gnx_dense_wgrad: dW = X^T . G, the weight gradient of gnx_dense (what tf.GradientTape derives for layers.py:136,
 * trainable.py:70-78): X [n, F], G [n, O] (the gradient of the pre-activation), dW [F, O] contiguous.  float32 MFMA over row
 * slabs; d_work [work_floats] is scratch for the per-slab partial sums (>= F * O floats; more scratch = more slabs in flight,
 * 2048 * F * O is the most it uses); the slabs are added in a fixed order (reproducible). */
int gnx_dense_wgrad(const float *d_X, int64_t ldx, const float *d_G, int64_t ldg, int64_t n, int64_t F, int64_t O, float *d_dW,
                    float *d_work, int64_t work_floats, void *stream);

/* The task head, NodeClassification (gnntf/core/gnn/graph_predictor.py:16-31), fused over the listed nodes.  All three are
 * stream-ordered and allocation-free (no synchronisation: they sit in the per-epoch loop of small graphs); node ids / labels
 * out of range are never dereferenced -- such an item's loss is NaN (and so is the mean), its argmax -1, its gradient zero.
 * gnx_node_ce:      d_loss_per_node[i] = logsumexp(logits[nodes[i], :]) - logits[nodes[i], labels[i]] and their mean in
 *                   d_mean_loss[0] (= SparseCategoricalCrossentropy(from_logits) of log_softmax, graph_predictor.py:24-25).
 *                   d_loss_per_node must hold m + 256 floats (the tail is scratch of the fixed-order two-level mean).
 * gnx_node_ce_backward: d_grad_logits[nodes[i], :] += d_grad_loss[0] / m * (softmax(logits[nodes[i], :]) - onehot(labels[i]));
 *                   the caller zero-fills d_grad_logits [n_rows, C] first.
 * gnx_node_argmax:  d_out[i] = first index of the maximum of logits[nodes[i], :] (d_nodes NULL: row i); graph_predictor.py:17. */
int gnx_node_ce(const float *d_logits, int64_t ldl, int64_t n_rows, int64_t C, const int64_t *d_nodes,
                const int64_t *d_labels, int64_t m, float *d_loss_per_node, float *d_mean_loss, void *stream);
int gnx_node_ce_backward(const float *d_logits, int64_t ldl, int64_t n_rows, int64_t C, const int64_t *d_nodes,
                         const int64_t *d_labels, int64_t m, const float *d_grad_loss, float *d_grad_logits, int64_t ldg,
                         void *stream);
int gnx_node_argmax(const float *d_logits, int64_t ldl, int64_t n_rows, int64_t C, const int64_t *d_nodes, int64_t m,
                    int64_t *d_out, void *stream);

/* ---- vertex-partitioned propagation (multi-GPU; the reference has no counterpart -- SURVEY.md section 8(e)) ----------------
 * One process per GPU owns a contiguous block of rows.  Its feature buffer is
 *     X = [ region(0) .. region(self-1) | n_local local rows | region(self+1) .. region(n_ranks-1) ],
 * region(q) = [ rows of H PULLED from rank q | partial sums sum_j A_hat[i, j] H[j] PUSHED by q for rows i of this block ]
 * (the caller's plan decides which entries are pulled and which pushed; gnntf/sharded.py chooses a vertex cover of the cross
 * entries).  The main CSR of the block indexes X directly (gnx_graph_create_coo / _csr over n_buf columns; pushed partial sums
 * appear as weight-1 entries).  The SEND buffer is [ pulled rows for peer 0, 1, .. | pushed rows for peer 0, 1, .. ]:
 * the pulled half is a gather of local rows (d_send_pull_src: the LOCAL row of every pulled row, int32, in send order), the
 * pushed half the product of the PUSH graph [sum(send_push_rows) x n_local] with the local rows.  The two halves can be packed
 * and exchanged separately (GNX_HALO_PULL, GNX_HALO_PUSH): the pulled rows cost a short copy and can be on the links while the
 * partial sums -- most of the pack time -- are still being summed; GNX_HALO_ALL does both in one call / one RCCL group.
 *   gnx_halo_plan_create   per-peer row counts of both halves in both directions + the pulled-row list + the push graph (both
 *                          borrowed; NULL when the half is empty).  A rank may list itself as a peer (loop-back region after the
 *                          local rows): a one-rank communicator then carries a real send / receive pair (tests on one GPU)
 *   gnx_halo_plan_layout   n_buf, first local row, rows to send (total, pulled half), first row of every peer's region in X,
 *                          first row of every peer's slice of the pulled / of the pushed half of the send buffer (any NULL = skip)
 *   gnx_halo_pack          d_send (row length lds) <- the chosen half (or both) from the local rows of d_X; stream-ordered
 *   gnx_halo_exchange      one RCCL group: ncclRecv into the chosen half of every region of d_X + ncclSend of the matching slices
 *                          of d_send; d_X and d_send contiguous with row length C; `nccl_comm` is the caller's ncclComm_t whose
 *                          ranks are the plan's ranks.  GNX_ERR_UNSUPPORTED when no RCCL entry points are known: move the rows
 *                          yourself (any transport) using the offsets of gnx_halo_plan_layout
 *   gnx_halo_bind_rccl     hands over ncclGroupStart / ncclGroupEnd / ncclSend / ncclRecv of the library instance that created
 *                          the communicator (all NULL = unbind).  Without it the exchange looks the symbols up in the RCCL that
 *                          is ALREADY loaded in the process; it never loads a copy of its own
 * One iteration = gnx_halo_pack -> exchange -> gnx_spmm (or gnx_spmm_rows) over X into the local rows of the other buffer. */
enum { GNX_HALO_ALL = 0, GNX_HALO_PULL = 1, GNX_HALO_PUSH = 2 };
int gnx_halo_plan_create(int n_ranks, int self, int64_t n_local, const int64_t *recv_pull_rows, const int64_t *recv_push_rows,
                         const int64_t *send_pull_rows, const int64_t *send_push_rows, const int32_t *d_send_pull_src,
                         gnx_graph_t push_graph, gnx_halo_plan_t *out);
int gnx_halo_plan_destroy(gnx_halo_plan_t plan);
int gnx_halo_plan_layout(gnx_halo_plan_t plan, int64_t *n_buf, int64_t *local_row0, int64_t *n_send, int64_t *n_send_pull,
                         int64_t *recv_row0, int64_t *send_pull_row0, int64_t *send_push_row0);
int gnx_halo_pack(gnx_halo_plan_t plan, int part, const float *d_X, int64_t ldx, int64_t C, float *d_send, int64_t lds, void *stream);
int gnx_halo_exchange(gnx_halo_plan_t plan, int part, void *nccl_comm, const float *d_send, float *d_X, int64_t C, void *stream);
int gnx_halo_bind_rccl(void *nccl_group_start, void *nccl_group_end, void *nccl_send, void *nccl_recv);
/* out[r,:] = X[idx[r],:] for int32 row ids (the pulled half of gnx_halo_pack as a call of its own); any width, grid-stride. */
int gnx_gather_rows32(const float *d_X, int64_t ldx, const int32_t *d_idx, int64_t n_idx, int64_t C, float *d_out, int64_t ldo,
                      void *stream);

/* The link head, LinkPrediction.predict / loss (gnntf/core/gnn/graph_predictor.py:122-126, 136-144): the logit of every listed
 * edge, d_out[i] = sum_c F[u_i, c] * F[v_i, c] * (d_r[c] or 1) -- gather of both endpoint rows + product + (DistMult) weights
 * + reduction in one launch; d_edges int64 [m, 2].  Stream-ordered, allocation-free; an edge with an endpoint out of range
 * gets NaN (nothing is dereferenced) and no gradient.
 * gnx_edge_scores_backward: d_grad_F[u_i, :] += g_i * F[v_i, :] * r and the mirror for v_i (caller zero-fills d_grad_F). */
int gnx_edge_scores(const float *d_F, int64_t ldf, int64_t n_rows, int64_t C, const int64_t *d_edges, int64_t m,
                    const float *d_r, float *d_out, void *stream);
int gnx_edge_scores_backward(const float *d_F, int64_t ldf, int64_t n_rows, int64_t C, const int64_t *d_edges, int64_t m,
                             const float *d_r, const float *d_grad_out, float *d_grad_F, int64_t ldg, void *stream);

/* Halo packing for the vertex-partitioned path: out[r,:] = X[idx[r],:], idx int64 [n_idx]. */
int gnx_gather_rows(const float *d_X, int64_t ldx, const int64_t *d_idx, int64_t n_idx, int64_t C,
                    float *d_out, int64_t ldo, void *stream);

/* gnx_linear_combination: d_out[i] = sum_j coef[j] * d_src[j][i] over k <= 16 device arrays of n floats (d_src and coef are HOST
 * arrays of k entries; terms are added in index order; d_out may be one of the sources).  Ends the backward of the K-iteration
 * loop -- dH0 = g_0 + a (g_1 + ... + g_K), the adjoint of filter.py:20-21's "+ a * H0" in every iteration -- in one pass. */
int gnx_linear_combination(int k, const float *const *d_src, const float *coef, int64_t n, float *d_out, void *stream);

/* Measurement aid: a float4 streaming copy d_dst[0..n) = d_src[0..n) (n a multiple of 4, 16-byte aligned pointers).
 * bench.py times it next to the propagation as the measured-peak HBM rate (read + write bytes per second). */
int gnx_stream_copy(const float *d_src, float *d_dst, int64_t n_floats, void *stream);
/* The read-only yardstick: streams d_src[0..n) once and folds it into d_sink64 [64 floats] (accumulated, not cleared). */
int gnx_stream_read(const float *d_src, int64_t n_floats, float *d_sink64, void *stream);
/* Measurement aid: launches n_blocks workgroups and writes, per workgroup, the id of the XCD it ran on (HW_REG_XCC_ID, 0..7) to
 * d_xcd_out [n_blocks].  The row-window launch order (gnx_graph_set_row_window) ASSUMES, for speed only, that the dispatcher deals
 * workgroups round-robin over the eight XCDs (blocks b and b + 8 share one); tests record here whether the box at hand does. */
int gnx_probe_block_xcd(int64_t n_blocks, int32_t *d_xcd_out, void *stream);

/* Name of the SpMM kernel the last gnx_spmm/_t call on this handle dispatched (static
 * string; for profiles and tests): "spmm_wave", "spmm_group8" ... "spmm_group32" (lanes per row; "spmm_group4+chunks": the merged small-graph launch), "..._drop" (weights made
 * in the kernel), "...+chunks" (structures below 2^20 rows: the chunks of the long rows share the launch of the short rows),
 * "spmm_gcnii_mfma", "spmm+dense_mfma". */
const char *gnx_graph_last_kernel(gnx_graph_t g);

#ifdef __cplusplus
}
#endif
#endif /* GNX_H */

// Training-mode propagation on gfx950: the dropped + re-normalised adjacency values of one iteration (layered.py:47-50, gnn.py:37-42)
// are produced INSIDE the SpMM kernels from the counter RNG, so a training iteration reads every stored entry's column and raw
// value, gathers only the kept entries' rows, and writes no value array.  Entries: gnx_spmm_dropped, gnx_spmm_dropped_chained (forward
// loop), gnx_spmm_dropped_back (backward loop, over the transposed structure).  Launch plumbing and the fused epilogue: gnx_spmm_device.h.
#include "gnx_spmm_device.h"

namespace {

// ---- training iterations: the dropped + re-normalised values are produced inside the SpMM (gnx_spmm_dropped) ------------
// Same row / lane mapping as the kernels above; what differs is where an entry's weight comes from: p.vals holds the RAW
// values and every weight is (D[row] * drop(raw)) * D[col] (layered.py:47-50 + gnn.py:41-42), computed ONCE per entry by one
// lane and handed to the lanes that need it (readlane / shuffles), so the hash costs one evaluation per stored entry.
template <int VEC, int U, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_spmm_wave_drop(const SpmmArgs p) {
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t slot = p.slot0 + xcd_block(p) * WPB + wib;
    if (slot >= p.n_rows) return;
    const int64_t row = p.row_list ? (int64_t)__builtin_amdgcn_readfirstlane(p.row_list[slot]) : slot;
    const int64_t beg = p.rowptr[row], end = p.rowptr[row + 1];
    if (end - beg > p.long_row) return;
    if (p.skip_empty && beg == end) return;   // GNX_ACT_SKIP_EMPTY (chained training loops: nobody gathers this row, a later launch writes it)
    for (int c0 = 0; c0 < p.C; c0 += 64 * VEC) {
        const int c = c0 + lane * VEC;
        const bool active = c < p.C;
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
        wave_accumulate<VEC, U, true>(p.colidx, p.vals, p.X, p.ldx, beg, end, active ? c : 0, lane, acc, false, &p.fuse, row);
        epilogue_store<VEC>(p, row, c, active, acc);
    }
}

// PIPE: the (col, raw value) pair a lane owns in the NEXT round is loaded before this round's kept entries are gathered, so a row
// of more than G entries pays the index latency once instead of once per round.
template <int VEC, int G, int U, bool PIPE>
__global__ __launch_bounds__(256) void k_spmm_group_drop(const SpmmArgs p) {
    constexpr int RPB = 256 / G;
    const int sub = threadIdx.x % G;
    const int64_t slot = p.slot0 + xcd_block(p) * RPB + threadIdx.x / G;
    if (slot >= p.n_rows) return;
    const int64_t row = p.row_order ? (int64_t)p.row_order[slot] : slot;
    int64_t beg, end;
    if (p.slot_beg) { beg = p.slot_beg[slot]; end = beg + p.slot_cnt[slot]; }
    else { beg = p.rowptr[row]; end = p.rowptr[row + 1]; }
    if (end - beg > p.long_row) return;
    if (p.skip_empty && beg == end) return;   // GNX_ACT_SKIP_EMPTY
    for (int c0 = 0; c0 < p.C; c0 += G * VEC) {
        const int c = c0 + sub * VEC;
        const bool active = c < p.C;
        const float *__restrict__ Xc = p.X + (active ? c : 0);
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
        int ncol = 0;
        float nraw = 0.f;
        if (PIPE && beg + sub < end) { ncol = p.colidx[beg + sub]; nraw = p.vals[beg + sub]; }
        for (int64_t base = beg; base < end; base += G) {          // G entries per round: lane `sub` owns entry base + sub
            const int n = (int)((end - base) < G ? (end - base) : G);
            int mycol = 0;
            float myw = 0.f;
            if (PIPE) {
                const int ccol = ncol;
                const float craw = nraw;
                if (base + G + sub < end) { ncol = p.colidx[base + G + sub]; nraw = p.vals[base + G + sub]; }
                if (sub < n) { mycol = ccol; myw = dropped_weight(p.fuse, craw, row, ccol); }
            } else if (sub < n) {
                mycol = p.colidx[base + sub];
                myw = dropped_weight(p.fuse, p.vals[base + sub], row, mycol);
            }
            // dropped entries (weight exactly 0) are not gathered: the group walks only the kept entries of its round, in order
            const uint64_t all = __ballot(myw != 0.f);
            uint32_t keep = (uint32_t)(all >> ((threadIdx.x & 63) / G * G)) & (G == 32 ? 0xFFFFFFFFu : ((1u << G) - 1u));
            while (keep) {
                float x[U][VEC];
                float w[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (keep) {
                        const int idx = __builtin_ctz(keep);
                        keep &= keep - 1;
                        const int j = __shfl(mycol, idx, G);
                        w[u] = __shfl(myw, idx, G);
                        vload<VEC>(x[u], Xc + (int64_t)j * p.ldx);
                    } else {
                        w[u] = 0.f;
#pragma unroll
                        for (int v = 0; v < VEC; ++v) x[u][v] = 0.f;
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[v] = fmaf(w[u], x[u][v], acc[v]);
            }
        }
        epilogue_store<VEC>(p, row, c, active, acc);
    }
}

template <int VEC, int U>
__global__ __launch_bounds__(256) void k_spmm_long_partial_drop(const SpmmArgs p) {
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t cslot = (int64_t)blockIdx.x * 4 + wib;
    if (cslot >= p.n_chunks) return;
    const int64_t chunk = p.chunk_order ? (int64_t)p.chunk_order[cslot] : cslot;
    const int32_t li = p.chunk_long[chunk];
    const int64_t row = p.long_rows[li];
    const int64_t beg = p.rowptr[row] + (chunk - p.long_chunk_ptr[li]) * p.long_chunk;
    const int64_t rend = p.rowptr[row + 1];
    const int64_t end = beg + p.long_chunk < rend ? beg + p.long_chunk : rend;
    for (int c0 = 0; c0 < p.C; c0 += 64 * VEC) {
        const int c = c0 + lane * VEC;
        const bool active = c < p.C;
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
        wave_accumulate<VEC, U, true>(p.colidx, p.vals, p.X, p.ldx, beg, end, active ? c : 0, lane, acc, false, &p.fuse, row);
        if (active) vstore<VEC>(p.partial + chunk * (int64_t)p.C + c, acc);
    }
}

// narrow long rows: the wave computes 64 weights per round (one per lane); sub-group s then takes entries s, s + NS, ... of the
// round, which is the entry -> sub-group dealing of k_spmm_long_partial_group (so the partial sums are bitwise the same)
template <int VEC, int G, int U>
__global__ __launch_bounds__(256) void k_spmm_long_partial_group_drop(const SpmmArgs p) {
    constexpr int NS = 64 / G;
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t cslot = (int64_t)blockIdx.x * 4 + wib;
    if (cslot >= p.n_chunks) return;
    const int64_t chunk = p.chunk_order ? (int64_t)p.chunk_order[cslot] : cslot;
    const int32_t li = p.chunk_long[chunk];
    const int64_t row = p.long_rows[li];
    const int64_t beg = p.rowptr[row] + (chunk - p.long_chunk_ptr[li]) * p.long_chunk;
    const int64_t rend = p.rowptr[row + 1];
    const int64_t end = beg + p.long_chunk < rend ? beg + p.long_chunk : rend;
    const int sub = lane / G;
    const int c = (lane % G) * VEC;
    const bool active = c < p.C;
    const float *__restrict__ Xc = p.X + (active ? c : 0);
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    // the reference kernel walks e = beg + sub + k * NS (k = 0, 1, ...) in batches of U: entry index within the chunk = sub + k NS.
    // A round of 64 entries covers k = 0 .. 64/NS - 1 = G - 1 for every sub-group.
    for (int64_t base = beg; base < end; base += 64) {
        const int n = (int)((end - base) < 64 ? (end - base) : 64);
        int mycol = 0;
        float myw = 0.f;
        if (lane < n) {
            mycol = p.colidx[base + lane];
            myw = dropped_weight(p.fuse, p.vals[base + lane], row, mycol);
        }
#pragma unroll 1
        for (int k = 0; k < G; k += U) {
            float x[U][VEC];
            float w[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int src = sub + (k + u) * NS;                  // entry of the round this sub-group takes in slot k + u
                const int j = __shfl(mycol, src);
                w[u] = __shfl(myw, src);
                if (k + u < G && src < n && w[u] != 0.f) vload<VEC>(x[u], Xc + (int64_t)j * p.ldx);   // dropped: not gathered
                else {
                    w[u] = 0.f;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) x[u][v] = 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[v] = fmaf(w[u], x[u][v], acc[v]);
        }
    }
#pragma unroll
    for (int off = G; off < 64; off <<= 1)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += __shfl_xor(acc[v], off);
    if (sub == 0 && active) vstore<VEC>(p.partial + chunk * (int64_t)p.C + c, acc);
}

[[maybe_unused]] constexpr bool DROP_U8 = false, DROP_PIPE = false;     // product defaults of the training row kernels (see launch_rows_drop)
constexpr int DROP_LONG_U = 4;

template <int VEC>
const char *launch_rows_drop(const SpmmArgs &p0, hipStream_t s) {
    SpmmArgs p = p0;
    const int lanes = (p.C + VEC - 1) / VEC;
    if (p.skip_empty && p.n_nonempty < p.n_rows) {            // as launch_rows: the slots of the rows without entries are not launched
        if (lanes <= 32 && p.row_order != nullptr) p.n_rows = p.n_nonempty;
        else if (lanes > 32 && p.nonempty_rows != nullptr) { p.row_list = p.nonempty_rows; p.n_rows = p.n_nonempty; }
    }
    if (p.n_rows == 0) return "spmm_none_drop";
    if (lanes > 32) {
        if (p.C <= 64 * VEC) GNX_ROW_PIECES((k_spmm_wave_drop<VEC, 8, 8>), 8, 512);
        else                 GNX_ROW_PIECES((k_spmm_wave_drop<VEC, 8, 4>), 4, 256);
        return "spmm_wave_drop";
    }
    // U gathers in flight per lane and the index prefetch: round-4 A/B on the config-4 graph (tuning build bits 1 << 17 = U 8,
    // 1 << 19 = PIPE; profiles/NOTES.md)
#ifdef GNX_TUNING
    const bool u8 = (p.tune & (1 << 17)) != 0, pipe = (p.tune & (1 << 19)) != 0;
#else
    const bool u8 = DROP_U8, pipe = DROP_PIPE;
#endif
#define GNX_DROP_ROWS(G, RPB)                                                                                         \
    do {                                                                                                              \
        if (u8 && pipe)  GNX_ROW_PIECES((k_spmm_group_drop<VEC, G, 8, true>), RPB, 256);                              \
        else if (u8)     GNX_ROW_PIECES((k_spmm_group_drop<VEC, G, 8, false>), RPB, 256);                             \
        else if (pipe)   GNX_ROW_PIECES((k_spmm_group_drop<VEC, G, 4, true>), RPB, 256);                              \
        else             GNX_ROW_PIECES((k_spmm_group_drop<VEC, G, 4, false>), RPB, 256);                             \
    } while (0)
    // A group of G lanes takes G entries per round (lane `sub` draws the weight of entry base + sub), so G is also how many index
    // loads and draws are in flight per row.  Round 6 (profiles/NOTES.md, config-4 graph, middle iteration): rows of up to 4 lanes
    // (C <= 16) on 8-lane groups instead of 4-lane ones -- half the lanes then only fetch and draw, their gather repeats a
    // neighbour's line -- C = 8: 1.94 -> 1.52 ms forward, 2.03 -> 1.61 backward; C = 16: 2.00 -> 1.56 / 2.08 -> 1.63; same bits.
    // 16 lanes: 1.83 / 1.85 ms, 32 lanes: 2.7 ms (fewer rows per wave than the gathers need in flight).
    if (lanes > 16) { GNX_DROP_ROWS(32, 8); return "spmm_group32_drop"; }
    if (lanes > 8)  { GNX_DROP_ROWS(16, 16); return "spmm_group16_drop"; }
    GNX_DROP_ROWS(8, 32);
#undef GNX_DROP_ROWS
    return "spmm_group8_drop";
}

template <int VEC>
void launch_long_drop(const SpmmArgs &p, hipStream_t s) {
    const int lanes = (p.C + VEC - 1) / VEC;
    if (lanes > 32)      GNX_LAUNCH((k_spmm_long_partial_drop<VEC, 8>), blocks_for(p.n_chunks, 4), p);
#ifdef GNX_TUNING
    else if (p.tune & (1 << 18)) {
        if (lanes > 16)      GNX_LAUNCH((k_spmm_long_partial_group_drop<VEC, 32, 8>), blocks_for(p.n_chunks, 4), p);
        else if (lanes > 8)  GNX_LAUNCH((k_spmm_long_partial_group_drop<VEC, 16, 8>), blocks_for(p.n_chunks, 4), p);
        else if (lanes > 4)  GNX_LAUNCH((k_spmm_long_partial_group_drop<VEC, 8, 8>), blocks_for(p.n_chunks, 4), p);
        else                 GNX_LAUNCH((k_spmm_long_partial_group_drop<VEC, 4, 4>), blocks_for(p.n_chunks, 4), p);
    }
#endif
    else if (lanes > 16) GNX_LAUNCH((k_spmm_long_partial_group_drop<VEC, 32, DROP_LONG_U>), blocks_for(p.n_chunks, 4), p);
    else if (lanes > 8)  GNX_LAUNCH((k_spmm_long_partial_group_drop<VEC, 16, DROP_LONG_U>), blocks_for(p.n_chunks, 4), p);
    else if (lanes > 4)  GNX_LAUNCH((k_spmm_long_partial_group_drop<VEC, 8, DROP_LONG_U>), blocks_for(p.n_chunks, 4), p);
    else                 GNX_LAUNCH((k_spmm_long_partial_group_drop<VEC, 4, 4>), blocks_for(p.n_chunks, 4), p);
    GNX_LAUNCH((k_spmm_long_reduce<VEC>), blocks_for(p.n_long, 4), p);
}

}  // namespace

namespace gnx {

const char *launch_spmm_dropped(const SpmmArgs &p, int vec, bool has_long, hipStream_t s) {
    const char *name;
    if (vec == 4)      { name = launch_rows_drop<4>(p, s); if (has_long) launch_long_drop<4>(p, s); }
    else if (vec == 2) { name = launch_rows_drop<2>(p, s); if (has_long) launch_long_drop<2>(p, s); }
    else               { name = launch_rows_drop<1>(p, s); if (has_long) launch_long_drop<1>(p, s); }
    return name;
}

}  // namespace gnx

extern "C" {

int gnx_spmm_dropped(gnx_graph_t g, const float *d_D, float dropout_p, uint64_t seed, uint64_t stream_id, int transposed,
                     const float *d_X, int64_t ldx, int64_t C, const float *d_H0, int64_t ldh0, float beta, float alpha, int act,
                     float *d_out, int64_t ldo, void *stream) {
    int rc = check_common("gnx_spmm_dropped", g, d_X, ldx, C, d_H0, ldh0, d_out, ldo);
    if (rc != GNX_OK) return rc;
    GNX_CHECK_ARG(act == GNX_ACT_NONE || act == GNX_ACT_RELU, "gnx_spmm_dropped: invalid activation %d", act);
    GNX_CHECK_ARG(d_D != nullptr, "gnx_spmm_dropped: NULL degree scales");
    GNX_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "dropout rate %g outside [0, 1)", (double)dropout_p);
    GNX_CHECK_ARG(g->a.n_rows == g->a.n_cols || g->blk_col_gid != nullptr, "gnx_spmm_dropped: needs a square graph or a vertex block (gnx_graph_set_block)");
    if (g->has_dups) {   // per-entry dropout of duplicated COO entries needs the entry lists: use gnx_graph_normalize + gnx_spmm
        set_error("gnx_spmm_dropped: the graph holds duplicate COO entries");
        return GNX_ERR_UNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    if (transposed) {
        rc = ensure_transpose(g, s);
        if (rc != GNX_OK) return rc;
    }
    SpmmArgs p{};
    p.vals = transposed ? g->t_raw : g->raw_vals;
    p.X = d_X; p.ldx = ldx; p.H0 = d_H0; p.ldh0 = ldh0; p.beta = beta; p.alpha = alpha; p.act = act;
    p.out = d_out; p.ldo = ldo; p.C = (int)C;
    p.fuse.D = d_D; p.fuse.seed = seed; p.fuse.stream = stream_id; p.fuse.offset = g->stream_offset;
    p.fuse.thr = (uint32_t)((double)dropout_p * 16777216.0);
    p.fuse.scale = 1.0f / (1.0f - dropout_p);
    p.fuse.transposed = transposed ? 1 : 0;
    p.fuse.row0_key = g->blk_row0_global; p.fuse.row0_D = g->blk_row0_buf; p.fuse.gid = g->blk_col_gid;
    return launch_spmm(g, transposed ? g->t : g->a, p, s);
}

int gnx_spmm_dropped_chained(gnx_graph_t g, const float *d_D, float dropout_p, uint64_t seed, uint64_t stream_id, int x_prescaled,
                             const float *d_D_next, const float *d_X, int64_t ldx, int64_t C, const float *d_H0, int64_t ldh0, float beta,
                             float alpha, int act, float *d_out, int64_t ldo, void *stream) {
    int rc = check_common("gnx_spmm_dropped_chained", g, d_X, ldx, C, d_H0, ldh0, d_out, ldo);
    if (rc != GNX_OK) return rc;
    GNX_CHECK_ARG((act & ~GNX_ACT_SKIP_EMPTY) == GNX_ACT_NONE || (act & ~GNX_ACT_SKIP_EMPTY) == GNX_ACT_RELU,
                  "gnx_spmm_dropped_chained: invalid activation %d", act);
    if (!g->a.empty_rows_unreferenced) act &= ~GNX_ACT_SKIP_EMPTY;       // honoured only when nobody gathers the rows it would leave untouched
    GNX_CHECK_ARG(d_D != nullptr, "gnx_spmm_dropped_chained: NULL degree scales");
    GNX_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "dropout rate %g outside [0, 1)", (double)dropout_p);
    GNX_CHECK_ARG(g->a.n_rows == g->a.n_cols || g->blk_col_gid != nullptr, "gnx_spmm_dropped_chained: needs a square graph or a vertex block");
    if (g->has_dups) {
        set_error("gnx_spmm_dropped_chained: the graph holds duplicate COO entries");
        return GNX_ERR_UNSUPPORTED;
    }
    SpmmArgs p{};
    p.vals = g->raw_vals;
    p.X = d_X; p.ldx = ldx; p.H0 = d_H0; p.ldh0 = ldh0; p.beta = beta; p.alpha = alpha; p.act = act;
    p.out = d_out; p.ldo = ldo; p.C = (int)C;
    p.out_scale = d_D_next ? d_D_next + g->blk_row0_buf : nullptr;
    p.fuse.D = d_D; p.fuse.seed = seed; p.fuse.stream = stream_id; p.fuse.offset = g->stream_offset;
    p.fuse.thr = (uint32_t)((double)dropout_p * 16777216.0);
    p.fuse.scale = 1.0f / (1.0f - dropout_p);
    p.fuse.transposed = 0;
    p.fuse.col_prescaled = x_prescaled ? 1 : 0;
    p.fuse.row0_key = g->blk_row0_global; p.fuse.row0_D = g->blk_row0_buf; p.fuse.gid = g->blk_col_gid;
    return launch_spmm(g, g->a, p, (hipStream_t)stream);
}

int gnx_spmm_dropped_back(gnx_graph_t g, const float *d_D, float dropout_p, uint64_t seed, uint64_t stream_id, int x_prescaled,
                          const float *d_D_next, const float *d_X, int64_t ldx, int64_t C, const float *d_S_in, int64_t lds_in,
                          float s_alpha, float s_beta, float *d_S_out, int64_t lds_out, float y_beta, float *d_Y_out, int64_t ldy,
                          int act, void *stream) {
    int rc = check_common("gnx_spmm_dropped_back", g, d_X, ldx, C, d_S_in, lds_in, d_S_out, lds_out);
    if (rc != GNX_OK) return rc;
    GNX_CHECK_ARG(act == GNX_ACT_NONE || act == GNX_ACT_SKIP_EMPTY, "gnx_spmm_dropped_back: act must be GNX_ACT_NONE or GNX_ACT_SKIP_EMPTY");
    GNX_CHECK_ARG(act == GNX_ACT_NONE || (const void *)d_S_in == (const void *)d_S_out,
                  "gnx_spmm_dropped_back: GNX_ACT_SKIP_EMPTY needs the sum updated in place");
    if (rc != GNX_OK) return rc;
    GNX_CHECK_ARG(d_D != nullptr && d_S_in != nullptr, "gnx_spmm_dropped_back: NULL degree scales / running sum");
    GNX_CHECK_ARG(d_Y_out == nullptr || (ldy >= C && (const void *)d_Y_out != (const void *)d_X && (const void *)d_Y_out != (const void *)d_S_out
                                         && (const void *)d_Y_out != (const void *)d_S_in),
                  "gnx_spmm_dropped_back: the pre-scaled output needs a buffer of its own");
    GNX_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "dropout rate %g outside [0, 1)", (double)dropout_p);
    GNX_CHECK_ARG(g->a.n_rows == g->a.n_cols && g->blk_col_gid == nullptr, "gnx_spmm_dropped_back: needs a square stand-alone graph");
    if (g->has_dups) {
        set_error("gnx_spmm_dropped_back: the graph holds duplicate COO entries");
        return GNX_ERR_UNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    rc = ensure_transpose(g, s);
    if (rc != GNX_OK) return rc;
    if (!g->t.empty_rows_unreferenced) act = GNX_ACT_NONE;               // honoured only when nobody gathers the rows it would leave untouched
    SpmmArgs p{};
    p.vals = g->t_raw;
    p.X = d_X; p.ldx = ldx; p.H0 = d_S_in; p.ldh0 = lds_in; p.beta = s_beta; p.alpha = s_alpha; p.act = act;
    p.out = d_S_out; p.ldo = lds_out; p.C = (int)C;
    p.out2 = d_Y_out; p.ldo2 = ldy; p.beta2 = y_beta; p.out2_scale = d_Y_out ? d_D_next : nullptr;
    p.fuse.D = d_D; p.fuse.seed = seed; p.fuse.stream = stream_id; p.fuse.offset = g->stream_offset;
    p.fuse.thr = (uint32_t)((double)dropout_p * 16777216.0);
    p.fuse.scale = 1.0f / (1.0f - dropout_p);
    p.fuse.transposed = 1;
    p.fuse.col_prescaled = x_prescaled ? 1 : 0;
    return launch_spmm(g, g->t, p, s);
}

}  // extern "C"

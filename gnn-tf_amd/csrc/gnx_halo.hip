// The exchange plan of one vertex block through the C ABI: what a client needs to run the
// vertex-partitioned propagation -- the layout of the [regions | local | regions] feature buffer and of the send buffer,
// packing of every outgoing row, and the pairwise exchange either by the caller (offsets from gnx_halo_plan_layout) or over
// a caller-supplied RCCL communicator (ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd: point-to-point, one xGMI link
// per peer pair).
//
// What a peer receives is a region [rows PULLED from this rank | partial sums PUSHED by this rank].  The two halves cost
// very different amounts to produce: a pulled row is a copy of a local row (a gather, bandwidth bound, short), a pushed
// partial sum is an SpMM over the entries this rank sums for its peer (most of the pack time).  The send buffer is therefore
// [all pulled rows, peer by peer | all pushed rows, peer by peer], each half packed by its own launch and exchanged by its own
// RCCL group (GNX_HALO_PULL / GNX_HALO_PUSH), so that the pulled rows are on the links while the partial sums are still being
// summed; GNX_HALO_ALL does both halves in one call (one group: every peer gets its two slices back to back).
//
// libgnx.so has no link-time dependency on RCCL.  The four entry points it needs come from the caller
// (gnx_halo_bind_rccl: the addresses out of the SAME library instance that created the communicator), or, when nothing was
// bound, from whatever RCCL is already loaded in the process (never a second copy).
// The reference has no distributed code: nothing to mirror, the contract is "same logits as one GPU".
#include <dlfcn.h>

#include <atomic>
#include <vector>

#include "gnx_internal.h"

struct gnx_halo_plan {
    int n_ranks = 0, self = 0;
    int64_t n_local = 0, n_buf = 0, n_send = 0, n_send_pull = 0, local_row0 = 0;
    std::vector<int64_t> recv_pull, recv_push, send_pull, send_push;      // rows per peer
    std::vector<int64_t> recv_row0, send_pull_row0, send_push_row0;       // first row of region(q) / of q's slices of the send buffer
    const int32_t *d_pull_src = nullptr;  // borrowed: local row of every pulled row, in send order
    gnx_graph *push_graph = nullptr;      // borrowed: [sum(send_push) x n_local]
};

namespace {

typedef int (*nccl_group_fn)(void);
typedef int (*nccl_p2p_fn)(void *, size_t, int, int, void *, hipStream_t);
struct Rccl { nccl_group_fn start = nullptr, end = nullptr; nccl_p2p_fn send = nullptr, recv = nullptr; };

Rccl g_bound;                              // set by gnx_halo_bind_rccl
std::atomic<bool> g_have_bound{false};

bool complete(const Rccl &r) { return r.start && r.end && r.send && r.recv; }

// The RCCL already loaded in this process, or nothing: the communicator the caller hands over belongs to ONE library instance,
// and its send / recv must come out of that same instance, so a copy is never loaded here (RTLD_NOLOAD only finds what is
// mapped; it also sees a library some extension loaded RTLD_LOCAL, which RTLD_DEFAULT does not).  A miss is not remembered:
// the library may be loaded later.
bool find_loaded_rccl(Rccl &r) {
    void *handles[3] = {RTLD_DEFAULT, nullptr, nullptr};
    handles[1] = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    handles[2] = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    for (void *h : handles) {
        if (h == nullptr) continue;
        Rccl c;
        c.start = (nccl_group_fn)dlsym(h, "ncclGroupStart");
        c.end = (nccl_group_fn)dlsym(h, "ncclGroupEnd");
        c.send = (nccl_p2p_fn)dlsym(h, "ncclSend");
        c.recv = (nccl_p2p_fn)dlsym(h, "ncclRecv");
        if (complete(c)) { r = c; return true; }
    }
    return false;
}

// out[r, :] = X[src[r], :] -- the pulled rows of every peer in one launch.  One 16-byte unit per thread, grid-stride.
template <int VEC>
__global__ __launch_bounds__(256) void k_pack_rows(const float *__restrict__ X, int64_t ldx, const int32_t *__restrict__ src, int64_t n_rows,
                                                   int C, float *__restrict__ out, int64_t ldo) {
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    const int upr = C / VEC;                                       // units per row
    const int64_t total = n_rows * upr, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t r = e / upr;
        const int c = (int)(e - r * upr) * VEC;
        const vec_t v = *reinterpret_cast<const vec_t *>(X + (int64_t)src[r] * ldx + c);
        __builtin_nontemporal_store(v, reinterpret_cast<vec_t *>(out + r * ldo + c));     // read next by the copy engine / the link, not by a kernel
    }
}

inline bool aligned16(const void *p) { return ((uintptr_t)p % 16) == 0; }

int pack_rows(const float *X, int64_t ldx, const int32_t *src, int64_t n_rows, int64_t C, float *out, int64_t ldo, hipStream_t s) {
    if (n_rows == 0) return GNX_OK;
    const bool v4 = C % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && aligned16(X) && aligned16(out);
    const int64_t units = n_rows * (v4 ? C / 4 : C);
    const unsigned grid = (unsigned)std::min<int64_t>((units + 255) / 256, 256 * 64);
    if (v4) hipLaunchKernelGGL(k_pack_rows<4>, dim3(grid), dim3(256), 0, s, X, ldx, src, n_rows, (int)C, out, ldo);
    else    hipLaunchKernelGGL(k_pack_rows<1>, dim3(grid), dim3(256), 0, s, X, ldx, src, n_rows, (int)C, out, ldo);
    GNX_HIP(hipGetLastError());
    return GNX_OK;
}

}  // namespace

using namespace gnx;

extern "C" {

int gnx_halo_bind_rccl(void *group_start, void *group_end, void *send, void *recv) {
    if (!group_start && !group_end && !send && !recv) {          // unbind
        g_have_bound.store(false);
        g_bound = Rccl{};
        return GNX_OK;
    }
    GNX_CHECK_ARG(group_start && group_end && send && recv, "gnx_halo_bind_rccl: all four entry points, or all NULL to unbind");
    g_bound.start = (nccl_group_fn)group_start; g_bound.end = (nccl_group_fn)group_end;
    g_bound.send = (nccl_p2p_fn)send; g_bound.recv = (nccl_p2p_fn)recv;
    g_have_bound.store(true);
    return GNX_OK;
}

int gnx_halo_plan_create(int n_ranks, int self, int64_t n_local, const int64_t *recv_pull_rows, const int64_t *recv_push_rows,
                         const int64_t *send_pull_rows, const int64_t *send_push_rows, const int32_t *d_send_pull_src,
                         gnx_graph_t push_graph, gnx_halo_plan_t *out) {
    GNX_CHECK_ARG(out != nullptr, "gnx_halo_plan_create: out is NULL");
    *out = nullptr;
    GNX_CHECK_ARG(n_ranks >= 1 && self >= 0 && self < n_ranks && n_local >= 0, "gnx_halo_plan_create: bad rank / size");
    GNX_CHECK_ARG(recv_pull_rows && recv_push_rows && send_pull_rows && send_push_rows, "gnx_halo_plan_create: NULL row counts");
    gnx_halo_plan *p = new gnx_halo_plan();
    p->n_ranks = n_ranks; p->self = self; p->n_local = n_local; p->push_graph = push_graph; p->d_pull_src = d_send_pull_src;
    p->recv_pull.assign(recv_pull_rows, recv_pull_rows + n_ranks);
    p->recv_push.assign(recv_push_rows, recv_push_rows + n_ranks);
    p->send_pull.assign(send_pull_rows, send_pull_rows + n_ranks);
    p->send_push.assign(send_push_rows, send_push_rows + n_ranks);
    p->recv_row0.resize(n_ranks); p->send_pull_row0.resize(n_ranks); p->send_push_row0.resize(n_ranks);
    bool ok = true;
    int64_t at = 0, pulled = 0, pushed = 0;
    for (int q = 0; q < n_ranks; ++q) {
        ok = ok && recv_pull_rows[q] >= 0 && recv_push_rows[q] >= 0 && send_pull_rows[q] >= 0 && send_push_rows[q] >= 0;
        // a rank may list ITSELF as a peer (a loop-back region, placed after the local rows): a one-rank communicator can then
        // carry a real ncclSend / ncclRecv pair -- how the RCCL path is exercised on a one-GPU box
        if (q == self) { p->local_row0 = at; at += n_local; }
        p->recv_row0[q] = at; at += recv_pull_rows[q] + recv_push_rows[q];
        pulled += send_pull_rows[q];
    }
    for (int64_t q = 0, row = 0; q < n_ranks; ++q) { p->send_pull_row0[q] = row; row += send_pull_rows[q]; }
    for (int q = 0; q < n_ranks; ++q) { p->send_push_row0[q] = pulled + pushed; pushed += send_push_rows[q]; }
    p->n_buf = at; p->n_send_pull = pulled; p->n_send = pulled + pushed;
    ok = ok && (pulled == 0 || d_send_pull_src != nullptr);
    ok = ok && (pushed == 0 || (push_graph != nullptr && push_graph->a.n_rows == pushed && push_graph->a.n_cols == n_local));
    if (!ok) {
        delete p;
        GNX_CHECK_ARG(false, "gnx_halo_plan_create: negative counts, pulled rows without their source list, or the push graph is not "
                             "[sum(send_push_rows) x n_local]");
    }
    *out = p;
    return GNX_OK;
}

int gnx_halo_plan_destroy(gnx_halo_plan_t p) {
    delete p;
    return GNX_OK;
}

int gnx_halo_plan_layout(gnx_halo_plan_t p, int64_t *n_buf, int64_t *local_row0, int64_t *n_send, int64_t *n_send_pull,
                         int64_t *recv_row0, int64_t *send_pull_row0, int64_t *send_push_row0) {
    GNX_CHECK_ARG(p != nullptr, "gnx_halo_plan_layout: NULL plan");
    if (n_buf) *n_buf = p->n_buf;
    if (local_row0) *local_row0 = p->local_row0;
    if (n_send) *n_send = p->n_send;
    if (n_send_pull) *n_send_pull = p->n_send_pull;
    for (int q = 0; q < p->n_ranks; ++q) {
        if (recv_row0) recv_row0[q] = p->recv_row0[q];
        if (send_pull_row0) send_pull_row0[q] = p->send_pull_row0[q];
        if (send_push_row0) send_push_row0[q] = p->send_push_row0[q];
    }
    return GNX_OK;
}

int gnx_halo_pack(gnx_halo_plan_t p, int part, const float *d_X, int64_t ldx, int64_t C, float *d_send, int64_t lds, void *stream) {
    GNX_CHECK_ARG(p != nullptr, "gnx_halo_pack: NULL plan");
    GNX_CHECK_ARG(part == GNX_HALO_ALL || part == GNX_HALO_PULL || part == GNX_HALO_PUSH, "gnx_halo_pack: invalid part %d", part);
    if (p->n_send == 0) return GNX_OK;
    GNX_CHECK_ARG(d_X && d_send && C >= 1 && ldx >= C && lds >= C, "gnx_halo_pack: NULL buffer or bad sizes");
    const float *local = d_X + p->local_row0 * ldx;
    if (part != GNX_HALO_PUSH && p->n_send_pull > 0) {
        int rc = pack_rows(local, ldx, p->d_pull_src, p->n_send_pull, C, d_send, lds, (hipStream_t)stream);
        if (rc != GNX_OK) return rc;
    }
    if (part != GNX_HALO_PULL && p->n_send > p->n_send_pull)
        return gnx_spmm(p->push_graph, nullptr, nullptr, local, ldx, C, nullptr, 0, 1.0f, 0.0f, GNX_ACT_NONE,
                        d_send + p->n_send_pull * lds, lds, stream);
    return GNX_OK;
}

int gnx_halo_exchange(gnx_halo_plan_t p, int part, void *nccl_comm, const float *d_send, float *d_X, int64_t C, void *stream) {
    GNX_CHECK_ARG(p != nullptr && nccl_comm != nullptr, "gnx_halo_exchange: NULL plan / communicator");
    GNX_CHECK_ARG(part == GNX_HALO_ALL || part == GNX_HALO_PULL || part == GNX_HALO_PUSH, "gnx_halo_exchange: invalid part %d", part);
    GNX_CHECK_ARG(C >= 1 && (p->n_send == 0 || d_send) && d_X, "gnx_halo_exchange: NULL buffer");
    Rccl rccl;
    if (g_have_bound.load()) rccl = g_bound;
    else if (!find_loaded_rccl(rccl)) {
        set_error("gnx_halo_exchange: no RCCL entry points: hand them over with gnx_halo_bind_rccl (from the library that created the "
                  "communicator), load librccl before the call, or move the rows yourself with the offsets of gnx_halo_plan_layout");
        return GNX_ERR_UNSUPPORTED;
    }
    const int nccl_float = 7;             // ncclFloat32 of nccl.h's ncclDataType_t (ncclInt8 = 0 ... ncclFloat16 = 6, ncclFloat32 = 7)
    hipStream_t s = (hipStream_t)stream;
    const bool pull = part != GNX_HALO_PUSH, push = part != GNX_HALO_PULL;
    int rc = rccl.start();
    // rows are contiguous [rows, C] on both sides; receives are posted before sends, peer by peer
    for (int q = 0; q < p->n_ranks && rc == 0; ++q) {
        float *region = d_X + p->recv_row0[q] * C;
        if (pull && p->recv_pull[q] > 0) rc = rccl.recv(region, (size_t)(p->recv_pull[q] * C), nccl_float, q, nccl_comm, s);
        if (rc == 0 && push && p->recv_push[q] > 0)
            rc = rccl.recv(region + p->recv_pull[q] * C, (size_t)(p->recv_push[q] * C), nccl_float, q, nccl_comm, s);
        if (rc == 0 && pull && p->send_pull[q] > 0)
            rc = rccl.send((void *)(d_send + p->send_pull_row0[q] * C), (size_t)(p->send_pull[q] * C), nccl_float, q, nccl_comm, s);
        if (rc == 0 && push && p->send_push[q] > 0)
            rc = rccl.send((void *)(d_send + p->send_push_row0[q] * C), (size_t)(p->send_push[q] * C), nccl_float, q, nccl_comm, s);
    }
    const int rc_end = rccl.end();
    if (rc != 0 || rc_end != 0) {
        set_error("gnx_halo_exchange: RCCL returned %d / %d", rc, rc_end);
        return GNX_ERR_HIP;
    }
    return GNX_OK;
}

int gnx_gather_rows32(const float *d_X, int64_t ldx, const int32_t *d_idx, int64_t n_idx, int64_t C, float *d_out, int64_t ldo,
                      void *stream) {
    GNX_CHECK_ARG(n_idx >= 0 && C >= 1 && ldx >= C && ldo >= C, "gnx_gather_rows32: bad sizes");
    if (n_idx == 0) return GNX_OK;
    GNX_CHECK_ARG(d_X && d_idx && d_out, "gnx_gather_rows32: NULL pointer");
    return pack_rows(d_X, ldx, d_idx, n_idx, C, d_out, ldo, (hipStream_t)stream);
}

}  // extern "C"

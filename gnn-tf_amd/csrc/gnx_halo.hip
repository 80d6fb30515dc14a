// The exchange plan of one vertex block through the C ABI: what a client without
// torch.distributed needs to run the vertex-partitioned propagation -- the layout of the [regions | local | regions]
// feature buffer, packing of every outgoing message with one SpMM launch (a pulled row is a 1-entry row of the send
// graph, a pushed partial sum a many-entry row), and the pairwise exchange either by the caller (offsets from
// gnx_halo_plan_layout) or over a caller-supplied RCCL communicator (ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd:
// point-to-point, one xGMI link per peer pair).  The RCCL symbols are looked up in the running process (the client links
// or loads librccl itself); libgnx.so has no link-time dependency on it.
// The reference has no distributed code: nothing to mirror, the contract is "same logits as one GPU".
#include <dlfcn.h>

#include <vector>

#include "gnx_internal.h"

struct gnx_halo_plan {
    int n_ranks = 0, self = 0;
    int64_t n_local = 0, n_buf = 0, n_send = 0, local_row0 = 0;
    std::vector<int64_t> recv_rows, send_rows, recv_row0, send_row0;
    gnx_graph *send_graph = nullptr;      // borrowed: [n_send x n_local]
};

namespace {

typedef int (*nccl_group_fn)(void);
typedef int (*nccl_p2p_fn)(void *, size_t, int, int, void *, hipStream_t);
struct Rccl { nccl_group_fn start = nullptr, end = nullptr; nccl_p2p_fn send = nullptr, recv = nullptr; };

bool find_rccl(Rccl &r) {
    void *h = RTLD_DEFAULT;
    r.start = (nccl_group_fn)dlsym(h, "ncclGroupStart");
    if (!r.start) {                       // not loaded yet: try the system library
        h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return false;
        r.start = (nccl_group_fn)dlsym(h, "ncclGroupStart");
    }
    r.end = (nccl_group_fn)dlsym(h, "ncclGroupEnd");
    r.send = (nccl_p2p_fn)dlsym(h, "ncclSend");
    r.recv = (nccl_p2p_fn)dlsym(h, "ncclRecv");
    return r.start && r.end && r.send && r.recv;
}

}  // namespace

using namespace gnx;

extern "C" {

int gnx_halo_plan_create(int n_ranks, int self, int64_t n_local, const int64_t *recv_rows, const int64_t *send_rows,
                         gnx_graph_t send_graph, gnx_halo_plan_t *out) {
    GNX_CHECK_ARG(out != nullptr, "gnx_halo_plan_create: out is NULL");
    *out = nullptr;
    GNX_CHECK_ARG(n_ranks >= 1 && self >= 0 && self < n_ranks && n_local >= 0, "gnx_halo_plan_create: bad rank / size");
    GNX_CHECK_ARG(recv_rows && send_rows, "gnx_halo_plan_create: NULL row counts");
    gnx_halo_plan *p = new gnx_halo_plan();
    p->n_ranks = n_ranks; p->self = self; p->n_local = n_local; p->send_graph = send_graph;
    p->recv_rows.assign(recv_rows, recv_rows + n_ranks);
    p->send_rows.assign(send_rows, send_rows + n_ranks);
    p->recv_row0.resize(n_ranks); p->send_row0.resize(n_ranks);
    int64_t at = 0, sent = 0;
    bool ok = recv_rows[self] == 0 && send_rows[self] == 0;
    for (int q = 0; q < n_ranks; ++q) {
        ok = ok && recv_rows[q] >= 0 && send_rows[q] >= 0;
        if (q == self) { p->local_row0 = at; at += n_local; }
        p->recv_row0[q] = at; at += recv_rows[q];
        p->send_row0[q] = sent; sent += send_rows[q];
    }
    p->n_buf = at; p->n_send = sent;
    if (!ok || (sent > 0 && (send_graph == nullptr || send_graph->a.n_rows != sent || send_graph->a.n_cols != n_local))) {
        delete p;
        GNX_CHECK_ARG(false, "gnx_halo_plan_create: negative / self counts, or the send graph is not [sum(send_rows) x n_local]");
    }
    *out = p;
    return GNX_OK;
}

int gnx_halo_plan_destroy(gnx_halo_plan_t p) {
    delete p;
    return GNX_OK;
}

int gnx_halo_plan_layout(gnx_halo_plan_t p, int64_t *n_buf, int64_t *local_row0, int64_t *n_send, int64_t *recv_row0,
                         int64_t *send_row0) {
    GNX_CHECK_ARG(p != nullptr, "gnx_halo_plan_layout: NULL plan");
    if (n_buf) *n_buf = p->n_buf;
    if (local_row0) *local_row0 = p->local_row0;
    if (n_send) *n_send = p->n_send;
    for (int q = 0; q < p->n_ranks; ++q) {
        if (recv_row0) recv_row0[q] = p->recv_row0[q];
        if (send_row0) send_row0[q] = p->send_row0[q];
    }
    return GNX_OK;
}

int gnx_halo_pack(gnx_halo_plan_t p, const float *d_X, int64_t ldx, int64_t C, float *d_send, int64_t lds, void *stream) {
    GNX_CHECK_ARG(p != nullptr, "gnx_halo_pack: NULL plan");
    if (p->n_send == 0) return GNX_OK;
    GNX_CHECK_ARG(d_X && d_send, "gnx_halo_pack: NULL buffer");
    return gnx_spmm(p->send_graph, nullptr, nullptr, d_X + p->local_row0 * ldx, ldx, C, nullptr, 0, 1.0f, 0.0f, GNX_ACT_NONE, d_send, lds,
                    stream);
}

int gnx_halo_exchange(gnx_halo_plan_t p, void *nccl_comm, const float *d_send, float *d_X, int64_t C, void *stream) {
    GNX_CHECK_ARG(p != nullptr && nccl_comm != nullptr, "gnx_halo_exchange: NULL plan / communicator");
    GNX_CHECK_ARG(C >= 1 && (p->n_send == 0 || d_send) && d_X, "gnx_halo_exchange: NULL buffer");
    static Rccl rccl;
    static const bool have = find_rccl(rccl);
    if (!have) {
        set_error("gnx_halo_exchange: ncclSend / ncclRecv are not available in this process (load librccl first, or move the "
                  "rows yourself with the offsets of gnx_halo_plan_layout)");
        return GNX_ERR_UNSUPPORTED;
    }
    const int nccl_float = 7;             // ncclFloat32
    hipStream_t s = (hipStream_t)stream;
    int rc = rccl.start();
    for (int q = 0; q < p->n_ranks && rc == 0; ++q) {      // rows are contiguous [rows, C] on both sides: one message per peer
        if (q == p->self) continue;
        if (p->recv_rows[q] > 0) rc = rccl.recv(d_X + p->recv_row0[q] * C, (size_t)(p->recv_rows[q] * C), nccl_float, q, nccl_comm, s);
        if (rc == 0 && p->send_rows[q] > 0)
            rc = rccl.send((void *)(d_send + p->send_row0[q] * C), (size_t)(p->send_rows[q] * C), nccl_float, q, nccl_comm, s);
    }
    const int rc_end = rccl.end();
    if (rc != 0 || rc_end != 0) {
        set_error("gnx_halo_exchange: RCCL returned %d / %d", rc, rc_end);
        return GNX_ERR_HIP;
    }
    return GNX_OK;
}

}  // extern "C"

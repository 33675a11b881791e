// halo_rccl.hip -- FillBoundary on RCCL behind the C ABI (include/castro_hydro_amd.h: castro_amd_comm_*, castro_amd_halo_plan_*,
// castro_amd_fill_boundary, castro_amd_allreduce_min).
//
// Replaces the same-level part of AmrLevel::FillPatch as Castro::expand_state uses it (Source/driver/Castro.cpp:4201-4209)
// plus the physical-boundary fill (Source/problems/Castro_bc_fill_nd.cpp:11-125) for a C++ / AMReX host that has no
// torch.distributed: pack every region of the FAB with one launch -> ONE grouped ncclSend / ncclRecv exchange over xGMI ->
// unpack (one launch for the periodic wraps onto this rank, one for the remote regions) -> k_bc_fill, all on the caller's
// stream.  No host synchronisation, no allocation after the plan has been created.
//
// RCCL is bound at run time (dlopen / dlsym of librccl), so the kernel library has no link-time dependency on it: a
// single-GPU host never loads RCCL, and inside a PyTorch process the copy PyTorch has already loaded is the one used.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>
#include "../../include/castro_hydro_amd.h"
#include "ctu_kernels.h"

using namespace cad;

// ---- the slice of rccl.h this file needs (ABI of RCCL 2.x / NCCL 2.x: stable C interface) -------------------------
namespace {
typedef struct ncclComm* ncclComm_t;
struct ncclUniqueId { char internal[128]; };
enum { ncclSuccess = 0 };
enum { ncclFloat64 = 8 };          // ncclDataType_t: ncclDouble
enum { ncclMin = 3 };              // ncclRedOp_t

struct Rccl {
    void* handle = nullptr;
    int (*GetVersion)(int*) = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*CommCount)(const ncclComm_t, int*) = nullptr;
    int (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    char version[48] = "RCCL (not loaded)";
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl()
{
    // an already-loaded copy first (PyTorch ships its own librccl), then the ROCm installation's
    const char* names[] = { std::getenv("CASTRO_AMD_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    void* h = nullptr;
    for (const char* n : names) { if (n && *n && (h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break; }
    if (!h) for (const char* n : names) { if (n && *n && (h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break; }
    if (!h) { std::fprintf(stderr, "castro_hydro_amd: cannot load librccl (%s)\n", dlerror()); return; }
    Rccl r;
    r.handle = h;
#define SYM(field, name) *(void**)(&r.field) = dlsym(h, name); if (!r.field) { std::fprintf(stderr, "castro_hydro_amd: librccl lacks %s\n", name); return; }
    SYM(GetVersion, "ncclGetVersion") SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy") SYM(CommCount, "ncclCommCount") SYM(CommUserRank, "ncclCommUserRank")
    SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd") SYM(Send, "ncclSend") SYM(Recv, "ncclRecv")
    SYM(AllReduce, "ncclAllReduce") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    int v = 0;
    if (r.GetVersion(&v) == ncclSuccess)
        std::snprintf(r.version, sizeof r.version, "RCCL %d.%d.%d", v / 10000, (v / 100) % 100, v % 100);
    g_rccl = r;
}

const Rccl* rccl()
{
    std::call_once(g_rccl_once, load_rccl);
    return g_rccl.handle ? &g_rccl : nullptr;
}

int nccl_check(const Rccl* R, int rc, const char* what)
{
    if (rc == ncclSuccess) return CASTRO_AMD_OK;
    std::fprintf(stderr, "castro_hydro_amd: %s failed: %s\n", what, R->GetErrorString ? R->GetErrorString(rc) : "?");
    return CASTRO_AMD_ERR_HIP;
}
} // namespace

struct castro_amd_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, size = 1, device = 0;
    bool owned = true;          // created here (destroyed here) or adopted from the host application
};

struct castro_amd_halo_plan {
    castro_amd_comm* comm = nullptr;
    int ncomp = 0;
    int nreg = 0;
    // region tables in the order given by the caller
    std::vector<int> slo, shi, rlo, rhi;               // 3 ints per region
    std::vector<long long> off, count;                 // offset / length in doubles of region r in both buffers
    // derived
    std::vector<int> local, remote;                    // region indices: periodic wraps onto this rank / exchanged
    std::vector<int> u_lo_local, u_hi_local, u_lo_remote, u_hi_remote;
    std::vector<long long> u_off_local, u_off_remote;
    std::vector<int> send_order, recv_order;           // remote regions sorted by (peer, tag)
    std::vector<int> peer, stag, rtag;
    double* sbuf = nullptr;
    double* rbuf = nullptr;
    long long total = 0;
    int self_send = 0;          // test mode: periodic wraps go through ncclSend / ncclRecv to this rank as well
    hipEvent_t ev_packed = nullptr;   // castro_amd_fill_boundary_ex: recorded behind the pack launch
    bool packed_recorded = false;
};

extern "C" {

const char* castro_amd_comm_version(void)
{
    const Rccl* R = rccl();
    return R ? R->version : "RCCL (not available)";
}

int castro_amd_comm_unique_id(void* id)
{
    const Rccl* R = rccl();
    if (!R || !id) return R ? CASTRO_AMD_ERR_ARG : CASTRO_AMD_ERR_UNSUPPORTED;
    ncclUniqueId u;
    int rc = nccl_check(R, R->GetUniqueId(&u), "ncclGetUniqueId");
    if (rc == CASTRO_AMD_OK) std::memcpy(id, u.internal, sizeof u.internal);
    return rc;
}

int castro_amd_comm_create(castro_amd_comm** out, int nranks, int rank, const void* id, int device)
{
    const Rccl* R = rccl();
    if (!R) return CASTRO_AMD_ERR_UNSUPPORTED;
    if (!out || !id || nranks < 1 || rank < 0 || rank >= nranks) return CASTRO_AMD_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return CASTRO_AMD_ERR_HIP;
    castro_amd_comm* c = new (std::nothrow) castro_amd_comm();
    if (!c) return CASTRO_AMD_ERR_NOMEM;
    ncclUniqueId u;
    std::memcpy(u.internal, id, sizeof u.internal);
    if (nccl_check(R, R->CommInitRank(&c->comm, nranks, u, rank), "ncclCommInitRank") != CASTRO_AMD_OK) { delete c; return CASTRO_AMD_ERR_HIP; }
    c->rank = rank; c->size = nranks; c->device = device; c->owned = true;
    *out = c;
    return CASTRO_AMD_OK;
}

int castro_amd_comm_adopt(castro_amd_comm** out, void* nccl_comm, int device)
{
    const Rccl* R = rccl();
    if (!R) return CASTRO_AMD_ERR_UNSUPPORTED;
    if (!out || !nccl_comm) return CASTRO_AMD_ERR_ARG;
    castro_amd_comm* c = new (std::nothrow) castro_amd_comm();
    if (!c) return CASTRO_AMD_ERR_NOMEM;
    c->comm = (ncclComm_t)nccl_comm;
    c->owned = false;
    c->device = device;
    if (nccl_check(R, R->CommCount(c->comm, &c->size), "ncclCommCount") != CASTRO_AMD_OK ||
        nccl_check(R, R->CommUserRank(c->comm, &c->rank), "ncclCommUserRank") != CASTRO_AMD_OK) { delete c; return CASTRO_AMD_ERR_HIP; }
    *out = c;
    return CASTRO_AMD_OK;
}

int castro_amd_comm_rank(const castro_amd_comm* c) { return c ? c->rank : CASTRO_AMD_ERR_ARG; }
int castro_amd_comm_size(const castro_amd_comm* c) { return c ? c->size : CASTRO_AMD_ERR_ARG; }

int castro_amd_comm_destroy(castro_amd_comm* c)
{
    if (!c) return CASTRO_AMD_OK;
    int rc = CASTRO_AMD_OK;
    const Rccl* R = rccl();
    if (c->owned && c->comm && R) rc = nccl_check(R, R->CommDestroy(c->comm), "ncclCommDestroy");
    delete c;
    return rc;
}

int castro_amd_allreduce_min(castro_amd_comm* c, double* d_buf, int n, void* stream)
{
    const Rccl* R = rccl();
    if (!R) return CASTRO_AMD_ERR_UNSUPPORTED;
    if (!c || !d_buf || n < 1) return CASTRO_AMD_ERR_ARG;
    return nccl_check(R, R->AllReduce(d_buf, d_buf, (size_t)n, ncclFloat64, ncclMin, c->comm, (hipStream_t)stream), "ncclAllReduce");
}

int castro_amd_halo_plan_create(castro_amd_halo_plan** out, castro_amd_comm* comm, int nregions,
                                const castro_amd_halo_region* regions, int ncomp)
{
    if (!out || !comm || nregions < 0 || nregions > CASTRO_AMD_MAX_REGIONS || (nregions > 0 && !regions) || ncomp < 1) return CASTRO_AMD_ERR_ARG;
    castro_amd_halo_plan* p = new (std::nothrow) castro_amd_halo_plan();
    if (!p) return CASTRO_AMD_ERR_NOMEM;
    p->comm = comm; p->ncomp = ncomp; p->nreg = nregions;
    if (const char* e = std::getenv("CASTRO_AMD_HALO_SELF_SEND")) p->self_send = std::atoi(e);
    long long off = 0;
    for (int r = 0; r < nregions; ++r) {
        const castro_amd_halo_region& g = regions[r];
        if (g.peer < 0 || g.peer >= comm->size) { delete p; return CASTRO_AMD_ERR_ARG; }
        long long ns = 1, nr = 1;
        for (int d = 0; d < 3; ++d) {
            if (g.sbox_hi[d] < g.sbox_lo[d] || g.rbox_hi[d] < g.rbox_lo[d]) { delete p; return CASTRO_AMD_ERR_ARG; }
            ns *= g.sbox_hi[d] - g.sbox_lo[d] + 1; nr *= g.rbox_hi[d] - g.rbox_lo[d] + 1;
            p->slo.push_back(g.sbox_lo[d]); p->shi.push_back(g.sbox_hi[d]);
            p->rlo.push_back(g.rbox_lo[d]); p->rhi.push_back(g.rbox_hi[d]);
        }
        if (ns != nr) { delete p; return CASTRO_AMD_ERR_ARG; }     // what goes out towards `off` comes back from `-off`
        p->off.push_back(off); p->count.push_back(ns * ncomp);
        off += ns * ncomp;
        p->peer.push_back(g.peer); p->stag.push_back(g.send_tag); p->rtag.push_back(g.recv_tag);
        const bool is_local = g.peer == comm->rank && !p->self_send;
        (is_local ? p->local : p->remote).push_back(r);
    }
    p->total = off;
    // a periodic wrap onto this rank: what this rank sends with tag t is what it receives as the region whose recv_tag is t
    for (int r : p->local) {
        int src = -1;
        for (int q : p->local) if (p->stag[q] == p->rtag[r]) { src = q; break; }
        if (src < 0) { delete p; return CASTRO_AMD_ERR_ARG; }
        for (int d = 0; d < 3; ++d) { p->u_lo_local.push_back(p->rlo[3 * r + d]); p->u_hi_local.push_back(p->rhi[3 * r + d]); }
        p->u_off_local.push_back(p->off[src]);
    }
    for (int r : p->remote) {
        for (int d = 0; d < 3; ++d) { p->u_lo_remote.push_back(p->rlo[3 * r + d]); p->u_hi_remote.push_back(p->rhi[3 * r + d]); }
        p->u_off_remote.push_back(p->off[r]);
    }
    // Both sides of a pair of ranks must issue their sends and receives in the same order: sends sorted by (peer, send_tag),
    // receives by (peer, recv_tag) -- the receiver's recv_tag of a region IS the sender's send_tag (castro_amd_halo_region).
    p->send_order = p->remote; p->recv_order = p->remote;
    std::sort(p->send_order.begin(), p->send_order.end(), [&](int a, int b) {
        return p->peer[a] != p->peer[b] ? p->peer[a] < p->peer[b] : p->stag[a] < p->stag[b]; });
    std::sort(p->recv_order.begin(), p->recv_order.end(), [&](int a, int b) {
        return p->peer[a] != p->peer[b] ? p->peer[a] < p->peer[b] : p->rtag[a] < p->rtag[b]; });
    if (hipSetDevice(comm->device) != hipSuccess) { delete p; return CASTRO_AMD_ERR_HIP; }
    const size_t bytes = (size_t)std::max<long long>(off, 1) * sizeof(double);
    if (hipMalloc(&p->sbuf, bytes) != hipSuccess || hipMalloc(&p->rbuf, bytes) != hipSuccess) {
        if (p->sbuf) hipFree(p->sbuf);
        delete p;
        return CASTRO_AMD_ERR_NOMEM;
    }
    if (hipEventCreateWithFlags(&p->ev_packed, hipEventDisableTiming) != hipSuccess) {
        hipFree(p->sbuf); hipFree(p->rbuf);
        delete p;
        return CASTRO_AMD_ERR_HIP;
    }
    *out = p;
    return CASTRO_AMD_OK;
}

int castro_amd_halo_plan_destroy(castro_amd_halo_plan* p)
{
    if (!p) return CASTRO_AMD_OK;
    if (p->sbuf) hipFree(p->sbuf);
    if (p->rbuf) hipFree(p->rbuf);
    if (p->ev_packed) hipEventDestroy(p->ev_packed);
    delete p;
    return CASTRO_AMD_OK;
}

long long castro_amd_halo_plan_bytes_sent(const castro_amd_halo_plan* p)
{
    if (!p) return 0;
    long long n = 0;
    for (int r : p->remote) n += p->count[r] * (long long)sizeof(double);
    return n;
}

static int fill_boundary_impl(castro_amd_ctx* ctx, castro_amd_halo_plan* p, const castro_amd_fab* state,
                              const castro_amd_geom* geom, void* stream, bool mark_packed);

int castro_amd_fill_boundary(castro_amd_ctx* ctx, castro_amd_halo_plan* p, const castro_amd_fab* state,
                             const castro_amd_geom* geom, void* stream)
{
    return fill_boundary_impl(ctx, p, state, geom, stream, false);
}

int castro_amd_fill_boundary_ex(castro_amd_ctx* ctx, castro_amd_halo_plan* p, const castro_amd_fab* state,
                                const castro_amd_geom* geom, int flags, void* stream)
{
    if (flags != 0) return CASTRO_AMD_ERR_ARG;
    return fill_boundary_impl(ctx, p, state, geom, stream, true);
}

int castro_amd_halo_plan_wait_packed(castro_amd_halo_plan* p, void* other_stream)
{
    if (!p || !p->ev_packed) return CASTRO_AMD_ERR_ARG;
    if (!p->packed_recorded) return CASTRO_AMD_ERR_ARG;          // no castro_amd_fill_boundary_ex has been issued with this plan
    if (hipSetDevice(p->comm->device) != hipSuccess) return CASTRO_AMD_ERR_HIP;
    return hipStreamWaitEvent((hipStream_t)other_stream, p->ev_packed, 0) == hipSuccess ? CASTRO_AMD_OK : CASTRO_AMD_ERR_HIP;
}

static int fill_boundary_impl(castro_amd_ctx* ctx, castro_amd_halo_plan* p, const castro_amd_fab* state,
                              const castro_amd_geom* geom, void* stream, bool mark_packed)
{
    if (!ctx || !p || !state || !state->p || state->ncomp != p->ncomp) return CASTRO_AMD_ERR_ARG;
    // the plan's buffers and the communicator live on the communicator's device; a plan is single-stream: two calls with one
    // plan on two streams would race on its send / receive buffers (one plan per stream, like one context per stream)
    if (hipSetDevice(p->comm->device) != hipSuccess) return CASTRO_AMD_ERR_HIP;
    hipStream_t s = (hipStream_t)stream;
    DFab f;
    {
        f.p = state->p;
        const long nx = state->hi[0] - state->lo[0] + 1, ny = state->hi[1] - state->lo[1] + 1, nz = state->hi[2] - state->lo[2] + 1;
        for (int d = 0; d < 3; ++d) f.lo[d] = state->lo[d];
        f.sy = nx; f.sz = nx * ny; f.sn = nx * ny * nz;
    }
    for (int r = 0; r < p->nreg; ++r)
        for (int d = 0; d < 3; ++d)
            if (p->slo[3 * r + d] < state->lo[d] || p->shi[3 * r + d] > state->hi[d] ||
                p->rlo[3 * r + d] < state->lo[d] || p->rhi[3 * r + d] > state->hi[d]) return CASTRO_AMD_ERR_ARG;
    int rc = CASTRO_AMD_OK;
    if (p->nreg > 0) {
        rc = launch_pack_regions(f, p->nreg, p->slo.data(), p->shi.data(), p->off.data(), p->ncomp, p->sbuf, 0, s, nullptr);
        if (rc != CASTRO_AMD_OK) return rc;
    }
    if (mark_packed) {
        if (hipEventRecord(p->ev_packed, s) != hipSuccess) return CASTRO_AMD_ERR_HIP;
        p->packed_recorded = true;
    }
    if (!p->remote.empty()) {
        const Rccl* R = rccl();
        if (!R) return CASTRO_AMD_ERR_UNSUPPORTED;
        int e = R->GroupStart();
        for (int r : p->recv_order)
            if (e == ncclSuccess) e = R->Recv(p->rbuf + p->off[r], (size_t)p->count[r], ncclFloat64, p->peer[r], p->comm->comm, s);
        for (int r : p->send_order)
            if (e == ncclSuccess) e = R->Send(p->sbuf + p->off[r], (size_t)p->count[r], ncclFloat64, p->peer[r], p->comm->comm, s);
        const int e2 = R->GroupEnd();        // always closed, even after an error inside the group
        if ((rc = nccl_check(R, e != ncclSuccess ? e : e2, "ncclSend/ncclRecv group")) != CASTRO_AMD_OK) return rc;
    }
    if (!p->local.empty()) {
        rc = launch_pack_regions(f, (int)p->local.size(), p->u_lo_local.data(), p->u_hi_local.data(), p->u_off_local.data(), p->ncomp,
                                 p->sbuf, 1, s, nullptr);
        if (rc != CASTRO_AMD_OK) return rc;
    }
    if (!p->remote.empty()) {
        // With CASTRO_AMD_HALO_SELF_SEND (test mode) the periodic wraps onto this rank travelled through RCCL too: the k-th
        // receive of the (self, self) pair, ordered by recv_tag, matched the k-th send, ordered by send_tag -- the same set of
        // tags -- so region r holds, at rbuf + off[r], what was sent with tag recv_tag(r), like any remote region.
        rc = launch_pack_regions(f, (int)p->remote.size(), p->u_lo_remote.data(), p->u_hi_remote.data(), p->u_off_remote.data(), p->ncomp,
                                 p->rbuf, 1, s, nullptr);
        if (rc != CASTRO_AMD_OK) return rc;
    }
    if (geom) {
        DevGeom G;
        for (int d = 0; d < 3; ++d) {
            G.dx[d] = geom->dx[d];
            G.domlo[d] = geom->domlo[d]; G.domhi[d] = geom->domhi[d];
            G.wall_lo[d] = (geom->lo_bc[d] >= 3) ? 1 : 0; G.wall_hi[d] = (geom->hi_bc[d] >= 3) ? 1 : 0;
            G.sym_lo[d] = (geom->lo_bc[d] == 3) ? 1 : 0; G.sym_hi[d] = (geom->hi_bc[d] == 3) ? 1 : 0;
        }
        rc = launch_bc_fill(f, state->lo, state->hi, state->ncomp, G, geom->lo_bc, geom->hi_bc, s, nullptr);
    }
    return rc;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Many boxes per rank (round 6): one grouped exchange for SEVERAL FABs of a level -- an AMR level, or two boxes per GPU.  Sends and
// receives are separate lists (with boxes of unequal size the zones a box sends to a neighbour and those it receives from it
// differ in shape); a message is identified by its tag, which both ends derive from (source box, destination box, periodic
// shift), so the k-th send of rank A to rank B in tag order IS the k-th receive of B from A in tag order.
// ---------------------------------------------------------------------------------------------------------------------------
struct castro_amd_halo_group {
    castro_amd_comm* comm = nullptr;
    int ncomp = 0, nfab = 0, self_send = 0;
    struct Msg { int fab, peer, tag; int lo[3], hi[3]; long long off, count; bool local; };
    std::vector<Msg> sends, recvs;              // off: doubles into sbuf (sends) / into rbuf -- or, local, into sbuf -- (recvs)
    std::vector<int> send_order, recv_order;    // the remote ones by (peer, tag)
    // per FAB, in chunks of <= CASTRO_AMD_MAX_REGIONS regions: pack, unpack from sbuf (local), unpack from rbuf (remote)
    struct Chunk { int fab; std::vector<int> lo, hi; std::vector<long long> off; };
    std::vector<Chunk> pack, unpack_local, unpack_remote;
    double* sbuf = nullptr;
    double* rbuf = nullptr;
    long long stotal = 0, rtotal = 0;
    hipEvent_t ev_packed = nullptr;             // castro_amd_fill_boundary_group_ex: recorded behind the last pack launch
    bool packed_recorded = false;
    std::vector<DFab> fabs;                     // descriptors of the FABs of the call in flight (sized once: no allocation per call)
};

static void group_chunks(const std::vector<castro_amd_halo_group::Msg>& msgs, int nfab, bool want_local, bool any,
                         std::vector<castro_amd_halo_group::Chunk>& out)
{
    for (int f = 0; f < nfab; ++f) {
        castro_amd_halo_group::Chunk c;
        c.fab = f;
        auto flush = [&]() { if (!c.off.empty()) { out.push_back(c); c.lo.clear(); c.hi.clear(); c.off.clear(); } };
        for (const auto& m : msgs) {
            if (m.fab != f || (!any && m.local != want_local)) continue;
            for (int d = 0; d < 3; ++d) { c.lo.push_back(m.lo[d]); c.hi.push_back(m.hi[d]); }
            c.off.push_back(m.off);
            if ((int)c.off.size() == CASTRO_AMD_MAX_REGIONS) flush();
        }
        flush();
    }
}

int castro_amd_halo_group_create(castro_amd_halo_group** out, castro_amd_comm* comm, int nfabs, int nsends, const castro_amd_halo_msg* sends,
                                 int nrecvs, const castro_amd_halo_msg* recvs, int ncomp)
{
    if (!out || !comm || nfabs < 1 || nsends < 0 || nrecvs < 0 || (nsends > 0 && !sends) || (nrecvs > 0 && !recvs) || ncomp < 1) return CASTRO_AMD_ERR_ARG;
    castro_amd_halo_group* g = new (std::nothrow) castro_amd_halo_group();
    if (!g) return CASTRO_AMD_ERR_NOMEM;
    g->comm = comm; g->ncomp = ncomp; g->nfab = nfabs;
    if (const char* e = std::getenv("CASTRO_AMD_HALO_SELF_SEND")) g->self_send = std::atoi(e);
    auto take = [&](const castro_amd_halo_msg& m, castro_amd_halo_group::Msg& o) {
        if (m.fab < 0 || m.fab >= nfabs || m.peer < 0 || m.peer >= comm->size) return false;
        o.fab = m.fab; o.peer = m.peer; o.tag = m.tag; o.count = ncomp;
        for (int d = 0; d < 3; ++d) {
            if (m.hi[d] < m.lo[d]) return false;
            o.lo[d] = m.lo[d]; o.hi[d] = m.hi[d];
            o.count *= m.hi[d] - m.lo[d] + 1;
        }
        o.local = m.peer == comm->rank && !g->self_send;
        o.off = 0;
        return true;
    };
    long long off = 0;
    for (int i = 0; i < nsends; ++i) {
        castro_amd_halo_group::Msg m;
        if (!take(sends[i], m)) { delete g; return CASTRO_AMD_ERR_ARG; }
        m.off = off; off += m.count;
        g->sends.push_back(m);
    }
    g->stotal = off;
    off = 0;
    for (int i = 0; i < nrecvs; ++i) {
        castro_amd_halo_group::Msg m;
        if (!take(recvs[i], m)) { delete g; return CASTRO_AMD_ERR_ARG; }
        if (m.local) {
            // a copy between two boxes of this rank (or a periodic wrap onto the same box): read from where the matching send was packed
            int src = -1;
            for (int q = 0; q < (int)g->sends.size(); ++q) if (g->sends[q].local && g->sends[q].tag == m.tag) { src = q; break; }
            if (src < 0 || g->sends[src].count != m.count) { delete g; return CASTRO_AMD_ERR_ARG; }
            m.off = g->sends[src].off;
        } else {
            m.off = off; off += m.count;
        }
        g->recvs.push_back(m);
    }
    g->rtotal = off;
    for (int i = 0; i < (int)g->sends.size(); ++i) if (!g->sends[i].local) g->send_order.push_back(i);
    for (int i = 0; i < (int)g->recvs.size(); ++i) if (!g->recvs[i].local) g->recv_order.push_back(i);
    std::sort(g->send_order.begin(), g->send_order.end(), [&](int a, int b) {
        return g->sends[a].peer != g->sends[b].peer ? g->sends[a].peer < g->sends[b].peer : g->sends[a].tag < g->sends[b].tag; });
    std::sort(g->recv_order.begin(), g->recv_order.end(), [&](int a, int b) {
        return g->recvs[a].peer != g->recvs[b].peer ? g->recvs[a].peer < g->recvs[b].peer : g->recvs[a].tag < g->recvs[b].tag; });
    g->fabs.resize((size_t)nfabs);
    // two local sends with one tag: a local receive could not tell them apart
    for (size_t i = 0; i < g->sends.size(); ++i)
        for (size_t j = i + 1; j < g->sends.size(); ++j)
            if (g->sends[i].local && g->sends[j].local && g->sends[i].tag == g->sends[j].tag) { delete g; return CASTRO_AMD_ERR_ARG; }
    // two messages of one pair of ranks with one tag would be matched arbitrarily: refuse
    for (size_t i = 1; i < g->send_order.size(); ++i) {
        const auto &a = g->sends[g->send_order[i - 1]], &b = g->sends[g->send_order[i]];
        if (a.peer == b.peer && a.tag == b.tag) { delete g; return CASTRO_AMD_ERR_ARG; }
    }
    for (size_t i = 1; i < g->recv_order.size(); ++i) {
        const auto &a = g->recvs[g->recv_order[i - 1]], &b = g->recvs[g->recv_order[i]];
        if (a.peer == b.peer && a.tag == b.tag) { delete g; return CASTRO_AMD_ERR_ARG; }
    }
    group_chunks(g->sends, nfabs, false, true, g->pack);
    group_chunks(g->recvs, nfabs, true, false, g->unpack_local);
    group_chunks(g->recvs, nfabs, false, false, g->unpack_remote);
    if (hipSetDevice(comm->device) != hipSuccess) { delete g; return CASTRO_AMD_ERR_HIP; }
    if (hipMalloc(&g->sbuf, (size_t)std::max<long long>(g->stotal, 1) * sizeof(double)) != hipSuccess ||
        hipMalloc(&g->rbuf, (size_t)std::max<long long>(g->rtotal, 1) * sizeof(double)) != hipSuccess) {
        if (g->sbuf) hipFree(g->sbuf);
        delete g;
        return CASTRO_AMD_ERR_NOMEM;
    }
    if (hipEventCreateWithFlags(&g->ev_packed, hipEventDisableTiming) != hipSuccess) {
        hipFree(g->sbuf); hipFree(g->rbuf);
        delete g;
        return CASTRO_AMD_ERR_HIP;
    }
    *out = g;
    return CASTRO_AMD_OK;
}

int castro_amd_halo_group_destroy(castro_amd_halo_group* g)
{
    if (!g) return CASTRO_AMD_OK;
    if (g->sbuf) hipFree(g->sbuf);
    if (g->rbuf) hipFree(g->rbuf);
    if (g->ev_packed) hipEventDestroy(g->ev_packed);
    delete g;
    return CASTRO_AMD_OK;
}

long long castro_amd_halo_group_bytes_sent(const castro_amd_halo_group* g)
{
    if (!g) return 0;
    long long n = 0;
    for (int i : g->send_order) n += g->sends[i].count * (long long)sizeof(double);
    return n;
}

static int fill_boundary_group_impl(castro_amd_ctx* ctx, castro_amd_halo_group* g, const castro_amd_fab* states,
                                    const castro_amd_geom* geom, void* stream, bool mark_packed);

int castro_amd_fill_boundary_group(castro_amd_ctx* ctx, castro_amd_halo_group* g, const castro_amd_fab* states,
                                   const castro_amd_geom* geom, void* stream)
{
    return fill_boundary_group_impl(ctx, g, states, geom, stream, false);
}

int castro_amd_fill_boundary_group_ex(castro_amd_ctx* ctx, castro_amd_halo_group* g, const castro_amd_fab* states,
                                      const castro_amd_geom* geom, int flags, void* stream)
{
    if (flags != 0) return CASTRO_AMD_ERR_ARG;
    return fill_boundary_group_impl(ctx, g, states, geom, stream, true);
}

int castro_amd_halo_group_wait_packed(castro_amd_halo_group* g, void* other_stream)
{
    if (!g || !g->ev_packed || !g->packed_recorded) return CASTRO_AMD_ERR_ARG;
    if (hipSetDevice(g->comm->device) != hipSuccess) return CASTRO_AMD_ERR_HIP;
    return hipStreamWaitEvent((hipStream_t)other_stream, g->ev_packed, 0) == hipSuccess ? CASTRO_AMD_OK : CASTRO_AMD_ERR_HIP;
}

static int fill_boundary_group_impl(castro_amd_ctx* ctx, castro_amd_halo_group* g, const castro_amd_fab* states,
                                    const castro_amd_geom* geom, void* stream, bool mark_packed)
{
    if (!ctx || !g || !states) return CASTRO_AMD_ERR_ARG;
    if (hipSetDevice(g->comm->device) != hipSuccess) return CASTRO_AMD_ERR_HIP;
    hipStream_t s = (hipStream_t)stream;
    std::vector<DFab>& f = g->fabs;
    for (int i = 0; i < g->nfab; ++i) {
        const castro_amd_fab& st = states[i];
        if (!st.p || st.ncomp != g->ncomp) return CASTRO_AMD_ERR_ARG;
        const long nx = st.hi[0] - st.lo[0] + 1, ny = st.hi[1] - st.lo[1] + 1, nz = st.hi[2] - st.lo[2] + 1;
        f[(size_t)i].p = st.p;
        for (int d = 0; d < 3; ++d) f[(size_t)i].lo[d] = st.lo[d];
        f[(size_t)i].sy = nx; f[(size_t)i].sz = nx * ny; f[(size_t)i].sn = nx * ny * nz;
    }
    auto inside = [&](const castro_amd_halo_group::Msg& m) {
        for (int d = 0; d < 3; ++d) if (m.lo[d] < states[m.fab].lo[d] || m.hi[d] > states[m.fab].hi[d]) return false;
        return true;
    };
    for (const auto& m : g->sends) if (!inside(m)) return CASTRO_AMD_ERR_ARG;
    for (const auto& m : g->recvs) if (!inside(m)) return CASTRO_AMD_ERR_ARG;
    int rc = CASTRO_AMD_OK;
    for (const auto& c : g->pack) {
        rc = launch_pack_regions(f[(size_t)c.fab], (int)c.off.size(), c.lo.data(), c.hi.data(), c.off.data(), g->ncomp, g->sbuf, 0, s, nullptr);
        if (rc != CASTRO_AMD_OK) return rc;
    }
    if (mark_packed) {
        if (hipEventRecord(g->ev_packed, s) != hipSuccess) return CASTRO_AMD_ERR_HIP;
        g->packed_recorded = true;
    }
    if (!g->send_order.empty() || !g->recv_order.empty()) {
        const Rccl* R = rccl();
        if (!R) return CASTRO_AMD_ERR_UNSUPPORTED;
        int e = R->GroupStart();
        for (int i : g->recv_order)
            if (e == ncclSuccess) e = R->Recv(g->rbuf + g->recvs[i].off, (size_t)g->recvs[i].count, ncclFloat64, g->recvs[i].peer, g->comm->comm, s);
        for (int i : g->send_order)
            if (e == ncclSuccess) e = R->Send(g->sbuf + g->sends[i].off, (size_t)g->sends[i].count, ncclFloat64, g->sends[i].peer, g->comm->comm, s);
        const int e2 = R->GroupEnd();
        if ((rc = nccl_check(R, e != ncclSuccess ? e : e2, "ncclSend/ncclRecv group")) != CASTRO_AMD_OK) return rc;
    }
    for (const auto& c : g->unpack_local) {
        rc = launch_pack_regions(f[(size_t)c.fab], (int)c.off.size(), c.lo.data(), c.hi.data(), c.off.data(), g->ncomp, g->sbuf, 1, s, nullptr);
        if (rc != CASTRO_AMD_OK) return rc;
    }
    for (const auto& c : g->unpack_remote) {
        rc = launch_pack_regions(f[(size_t)c.fab], (int)c.off.size(), c.lo.data(), c.hi.data(), c.off.data(), g->ncomp, g->rbuf, 1, s, nullptr);
        if (rc != CASTRO_AMD_OK) return rc;
    }
    if (geom) {
        DevGeom G;
        for (int d = 0; d < 3; ++d) {
            G.dx[d] = geom->dx[d];
            G.domlo[d] = geom->domlo[d]; G.domhi[d] = geom->domhi[d];
            G.wall_lo[d] = (geom->lo_bc[d] >= 3) ? 1 : 0; G.wall_hi[d] = (geom->hi_bc[d] >= 3) ? 1 : 0;
            G.sym_lo[d] = (geom->lo_bc[d] == 3) ? 1 : 0; G.sym_hi[d] = (geom->hi_bc[d] == 3) ? 1 : 0;
        }
        for (int i = 0; i < g->nfab && rc == CASTRO_AMD_OK; ++i)
            rc = launch_bc_fill(f[(size_t)i], states[i].lo, states[i].hi, states[i].ncomp, G, geom->lo_bc, geom->hi_bc, s, nullptr);
    }
    return rc;
}

} // extern "C"

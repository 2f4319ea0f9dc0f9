// d2g_mgpu.hip -- multi-GPU half of K2 behind the C ABI: an RCCL communicator wrapper and the row-sharded
// all-pairs engine (reference seam src/cmp_core.cpp:746-751 -> emit_rectangular).
//
// The path shards (independent units after one exchange, no reduction): rank r HOLDS rows
// [row_lo[r], row_lo[r+1]) of the N x S signature matrix -- the state file-sharded sketching leaves in HBM --
// and COMPUTES a pair-balanced row range of the upper triangle (d2g_ut_partition).  One step is
//   A   pack      my rows -> W blocks, block q = my rows x the register columns rank q prepares       (local kernel)
//   X1  exchange  block q -> rank q  ("all-to-all-v": RCCL send/recv pairs in one group)             (collective)
//   B   prepare   the bit-sliced operand of my column slice (N x S_q), exported in the exchange form (local kernels)
//   X2  exchange  my groups -> everyone ("all-gather-v")                                             (collective)
//   C   compare   the pair kernel over my row range of the gathered operand                          (local kernel)
// Column slices are whole 32-register groups (the operand's independent unit), rows and groups are split as
// evenly as integers allow: no divisibility requirement on N or S.  xGMI is a full point-to-point mesh, so
// direct send/recv pairs use all links at once; nothing here is a ring.
//
// RCCL is resolved at first use with dlopen (the copy already in the process, e.g. PyTorch's, else
// librccl.so.1): libd2g.so itself has no link-time dependency on the 0.5 GB library and the single-GPU CLI
// never pays for loading it.  Contexts that share ONE device (tests on a 1-GPU box) cannot form an RCCL clique;
// d2g_comm_create_all gives them a loopback transport (device-to-device copies ordered by events) that runs
// exactly the same send/recv lists.
#include "d2g_internal.h"
#include "d2g_k2.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <vector>

namespace {

// ------------------------------------------------------------------------------------------- RCCL, late-bound
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

Rccl *rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        // D2G_RCCL_LIB: an explicit library path (deployments with a private RCCL; tests point it at a missing file to
        // exercise the "no RCCL" error path)
        const char *forced = std::getenv("D2G_RCCL_LIB");
        if (forced && forced[0]) r.lib = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
        else {
            for (const char *n : names) if ((r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;   // a copy already loaded
            if (!r.lib) for (const char *n : names) if ((r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        }
        if (!r.lib) {                                 // dlerror() clears the message it returns: read it ONCE
            const char *e = dlerror();
            r.err = std::string("cannot load RCCL: ") + (e ? e : "not found");
            return;
        }
        auto sym = [&](const char *n) { void *p = dlsym(r.lib, n); if (!p && r.err.empty()) r.err = std::string("RCCL lacks ") + n; return p; };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.Broadcast = (decltype(r.Broadcast))sym("ncclBroadcast");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    return &r;
}

}  // namespace
struct d2g_comm;
namespace {

// ------------------------------------------------------------------------------------------- loopback transport
struct LocalOp { int from, to; const void *src; void *dst; size_t bytes; hipStream_t s; bool is_send; };
struct LocalGroup {
    std::vector<::d2g_comm *> members;
    std::vector<LocalOp> pending;
    int depth = 0;
};

}  // namespace

struct d2g_comm {
    d2g_ctx *ctx = nullptr;
    int rank = 0, world = 1;
    ncclComm_t nccl = nullptr;                 // RCCL transport (distinct devices)
    bool ranked = false;                       // created by d2g_comm_create: its peers are other processes / threads
    std::shared_ptr<LocalGroup> lg;            // loopback transport (one process, any devices)
};

namespace {

#define D2G_NCCL(ctx, call)                                                                     \
    do {                                                                                        \
        ncclResult_t r__ = (call);                                                              \
        if (r__ != ncclSuccess) {                                                               \
            (ctx)->last_error = std::string(#call) + ": " + rccl()->GetErrorString(r__);        \
            return D2G_ERR_HIP;                                                                 \
        }                                                                                       \
    } while (0)

// group_begin / send / recv / group_end: the one interface the engine speaks
int comm_group_begin(d2g_comm *c) {
    if (c->lg) { ++c->lg->depth; return D2G_OK; }
    if (c->world == 1) return D2G_OK;
    D2G_NCCL(c->ctx, rccl()->GroupStart());
    return D2G_OK;
}
int comm_send(d2g_comm *c, int peer, const void *src, size_t bytes, hipStream_t s) {
    if (!bytes) return D2G_OK;
    if (c->lg) { c->lg->pending.push_back({c->rank, peer, src, nullptr, bytes, s, true}); return D2G_OK; }
    D2G_CHECK(c->ctx, c->world > 1 && peer != c->rank, "comm_send: self transfers are local copies");
    D2G_NCCL(c->ctx, rccl()->Send(src, bytes, ncclUint8, peer, c->nccl, s));
    return D2G_OK;
}
int comm_recv(d2g_comm *c, int peer, void *dst, size_t bytes, hipStream_t s) {
    if (!bytes) return D2G_OK;
    if (c->lg) { c->lg->pending.push_back({peer, c->rank, nullptr, dst, bytes, s, false}); return D2G_OK; }
    D2G_CHECK(c->ctx, c->world > 1 && peer != c->rank, "comm_recv: self transfers are local copies");
    D2G_NCCL(c->ctx, rccl()->Recv(dst, bytes, ncclUint8, peer, c->nccl, s));
    return D2G_OK;
}
// loopback: match every send with the recv of the same (from, to) pair, in order, and run it as a copy on
// the RECEIVER's stream, fenced against both ranks' streams
int local_flush(d2g_comm *c) {
    LocalGroup &g = *c->lg;
    std::vector<LocalOp> ops;
    ops.swap(g.pending);
    std::vector<char> used(ops.size(), 0);
    for (size_t i = 0; i < ops.size(); ++i) {
        if (!ops[i].is_send) continue;
        size_t j = 0;
        for (; j < ops.size(); ++j)
            if (!used[j] && !ops[j].is_send && ops[j].from == ops[i].from && ops[j].to == ops[i].to) break;
        D2G_CHECK(c->ctx, j < ops.size() && ops[j].bytes == ops[i].bytes, "loopback transport: unmatched send/recv");
        used[j] = 1;
        d2g_ctx *sc = g.members[ops[i].from]->ctx, *rc = g.members[ops[i].to]->ctx;
        hipEvent_t ready = nullptr, done = nullptr;
        D2G_HIP(c->ctx, hipSetDevice(sc->device));
        D2G_HIP(c->ctx, hipEventCreateWithFlags(&ready, hipEventDisableTiming));
        D2G_HIP(c->ctx, hipEventRecord(ready, ops[i].s));                   // sender's data is complete
        D2G_HIP(c->ctx, hipSetDevice(rc->device));
        D2G_HIP(c->ctx, hipStreamWaitEvent(ops[j].s, ready, 0));
        D2G_HIP(c->ctx, hipMemcpyAsync(ops[j].dst, ops[i].src, ops[i].bytes, hipMemcpyDeviceToDevice, ops[j].s));
        D2G_HIP(c->ctx, hipEventCreateWithFlags(&done, hipEventDisableTiming));
        D2G_HIP(c->ctx, hipEventRecord(done, ops[j].s));
        D2G_HIP(c->ctx, hipSetDevice(sc->device));
        D2G_HIP(c->ctx, hipStreamWaitEvent(ops[i].s, done, 0));             // the sender may reuse its buffer afterwards
        (void)hipEventDestroy(ready);
        (void)hipEventDestroy(done);
    }
    for (size_t j = 0; j < ops.size(); ++j)
        D2G_CHECK(c->ctx, ops[j].is_send || used[j], "loopback transport: recv without a matching send");
    return D2G_OK;
}
int comm_group_end(d2g_comm *c) {
    if (c->lg) {
        if (--c->lg->depth > 0) return D2G_OK;
        return local_flush(c);
    }
    if (c->world == 1) return D2G_OK;
    D2G_NCCL(c->ctx, rccl()->GroupEnd());
    return D2G_OK;
}

// ------------------------------------------------------------------------------------------- pack kernel
// rows [n][S] -> consecutive blocks, block b = [n][cols_b] with cols_b = [colstart[b], colstart[b+1]) (blocks are in
// ascending column order, so block b starts at n * colstart[b]): the send layout of the row-slice -> column-slice
// exchange, one launch.  colblk[c] = block of column c.
__global__ __launch_bounds__(256) void mg_pack_kernel(const uint64_t *__restrict__ rows, size_t n, size_t S,
                                                      const uint32_t *__restrict__ colstart, const uint16_t *__restrict__ colblk,
                                                      uint64_t *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * S) return;
    const size_t r = i / S;
    const uint32_t c = (uint32_t)(i - r * S);
    const uint32_t q = colblk[c], c0 = colstart[q], wq = colstart[q + 1] - c0;
    out[(size_t)n * c0 + r * wq + (c - c0)] = rows[i];
}

template <class T> std::vector<T> even_split(T total, int parts) {     // [parts+1] bounds, sizes differ by at most one
    std::vector<T> b(parts + 1);
    for (int p = 0; p <= parts; ++p) b[p] = (T)((unsigned __int128)total * p / parts);
    return b;
}

constexpr int MG_MAX_CHUNKS = 4;

}  // namespace

// ------------------------------------------------------------------------------------------- the engine
// Inside ONE step the exchanges overlap the sharded prepare: a rank's column slice is cut into C chunks of whole
// 32-register groups, and chunk c+1 is on the wire (engine stream `xs`) while chunk c is ranked and bit-sliced
// (compute stream); its planes leave while chunk c+1 is prepared.  The gathered operand keeps its groups in
// CHUNK-MAJOR order (all ranks' chunk 0, then chunk 1, ...): the pair kernel does not care about the order of
// groups, and the plane stream of chunk c can be derived as soon as chunk c has arrived.
struct d2g_allpairs {
    d2g_ctx *ctx = nullptr;
    d2g_comm *comm = nullptr;
    size_t N = 0, S = 0;
    int W = 1, rank = 0, C = 1;
    std::vector<size_t> row_lo;          // [W+1] rows each rank holds
    std::vector<size_t> blk_g;           // [W*C+1] natural group bounds of block q*C + c (rank q, chunk c)
    std::vector<uint32_t> colstart;      // [W*C+1] first register column of each block (last = S)
    std::vector<size_t> gpos;            // [W*C] first group of each block in the gathered (chunk-major) order
    std::vector<size_t> chunk_g0;        // [C+1] gathered groups of chunk c = [chunk_g0[c], chunk_g0[c+1])
    size_t r0 = 0, r1 = 0;               // rows of the triangle this rank computes
    size_t gw = 0, ng = 0;               // words per exchanged group, number of groups
    uint32_t *d_colstart = nullptr;
    uint16_t *d_colblk = nullptr;
    uint64_t *d_send = nullptr, *d_recv = nullptr;
    d2g_cmp_set *local[MG_MAX_CHUNKS] = {nullptr, nullptr, nullptr, nullptr};   // my column slice, chunk by chunk (exporter sets)
    // two operand buffers: the pipelined form prepares buffer i+1 while the pair kernel reads buffer i
    uint32_t *d_planes[2] = {nullptr, nullptr}, *d_meta[2] = {nullptr, nullptr};   // meta: [ng] groups + [W*C] status words
    d2g_cmp_set *full[2] = {nullptr, nullptr};
    hipStream_t xs = nullptr;            // exchange stream
    hipStream_t ps = nullptr;            // compute stream of the pipelined form's prepare
    hipEvent_t ev_pack = nullptr, ev_x1[MG_MAX_CHUNKS] = {}, ev_prep[MG_MAX_CHUNKS] = {}, ev_x2[MG_MAX_CHUNKS] = {};
    hipEvent_t x_done[2] = {nullptr, nullptr}, p_done[2] = {nullptr, nullptr}, in_ready = nullptr, plain_done = nullptr;
    bool p_valid[2] = {false, false}, plain_valid = false;
    unsigned long long nsteps = 0;
    int last = 0;                        // buffer of the most recent prepare
    // output of the step being enqueued (d2g_allpairs_step_*): its fill goes out right behind the pack, under the exchanges
    void *pre_out = nullptr; const float *pre_lut = nullptr;
    // per-phase timing of ONE step (d2g_allpairs_set_phase_timing): timing-enabled event pairs around every phase, on the
    // stream the phase is enqueued on; nothing synchronises until d2g_allpairs_phase_times
    struct PhaseEv { hipEvent_t a, b; int kind, chunk; };
    bool phase_timing = false;
    hipEvent_t ev_step0 = nullptr;
    std::vector<PhaseEv> pev;
    int blk(int q, int c) const { return q * C + c; }
    size_t n_me() const { return row_lo[rank + 1] - row_lo[rank]; }
    size_t w_blk(int b) const { return colstart[b + 1] - colstart[b]; }
    size_t g_blk(int b) const { return blk_g[b + 1] - blk_g[b]; }
    size_t s_me() const { return colstart[blk(rank, C - 1) + 1] - colstart[blk(rank, 0)]; }
    uint64_t *recv_chunk(int c) const { return d_recv + N * (colstart[blk(rank, c)] - colstart[blk(rank, 0)]); }
};

namespace {

int eng_alloc_buffer(d2g_allpairs *e, int b) {
    d2g_ctx *ctx = e->ctx;
    if (e->full[b]) return D2G_OK;
    const size_t nstat = (size_t)e->W * e->C;
    D2G_HIP(ctx, hipMalloc((void **)&e->d_planes[b], std::max<size_t>(e->ng * e->gw, 1) * 4));
    D2G_HIP(ctx, hipMalloc((void **)&e->d_meta[b], (e->ng + nstat + 4) * 4));
    D2G_HIP(ctx, hipMemset(e->d_meta[b], 0, (e->ng + nstat + 4) * 4));      // blocks without groups never write their status word
    if (int rc = d2g_cmp_set_from_planes_dev(ctx, e->N, e->S, e->d_planes[b], e->d_meta[b], &e->full[b])) return rc;
    e->full[b]->managed = true;                                             // the engine derives the plane stream, chunk by chunk
    e->full[b]->status_words = e->d_meta[b] + e->ng;
    e->full[b]->n_status = (int)nstat;
    return d2g_bitslice_managed_sparse_alloc(ctx, e->full[b]);      // N >= 8192: buffers of the sparse-tile path (ids, order, tile bitmaps)
}

// phase timing: begin/end bracket a phase's enqueue on stream s (no-ops unless enabled)
void pt_clear(d2g_allpairs *e) {
    for (auto &p : e->pev) { if (p.a) (void)hipEventDestroy(p.a); if (p.b) (void)hipEventDestroy(p.b); }
    e->pev.clear();
    if (e->ev_step0) { (void)hipEventDestroy(e->ev_step0); e->ev_step0 = nullptr; }
}
int pt_step_begin(d2g_allpairs *e, hipStream_t s) {
    if (!e->phase_timing) return D2G_OK;
    D2G_HIP(e->ctx, hipSetDevice(e->ctx->device));
    pt_clear(e);
    D2G_HIP(e->ctx, hipEventCreate(&e->ev_step0));
    D2G_HIP(e->ctx, hipEventRecord(e->ev_step0, s));
    return D2G_OK;
}
int pt_begin(d2g_allpairs *e, int kind, int chunk, hipStream_t s) {
    if (!e->phase_timing) return D2G_OK;
    D2G_HIP(e->ctx, hipSetDevice(e->ctx->device));
    d2g_allpairs::PhaseEv p{nullptr, nullptr, kind, chunk};
    D2G_HIP(e->ctx, hipEventCreate(&p.a));
    D2G_HIP(e->ctx, hipEventCreate(&p.b));
    e->pev.push_back(p);
    D2G_HIP(e->ctx, hipEventRecord(p.a, s));
    return D2G_OK;
}
int pt_end(d2g_allpairs *e, int kind, int chunk, hipStream_t s) {
    if (!e->phase_timing) return D2G_OK;
    D2G_HIP(e->ctx, hipSetDevice(e->ctx->device));
    for (size_t i = e->pev.size(); i-- > 0;)
        if (e->pev[i].kind == kind && e->pev[i].chunk == chunk) { D2G_HIP(e->ctx, hipEventRecord(e->pev[i].b, s)); return D2G_OK; }
    return D2G_ERR_INTERNAL;
}

// phases of one step on buffer b
int phase_pack(d2g_allpairs *e, const uint64_t *rows_dev, hipStream_t s) {
    const size_t n = e->n_me();
    D2G_HIP(e->ctx, hipSetDevice(e->ctx->device));
    if (n) {
        D2G_CHECK(e->ctx, rows_dev != nullptr, "allpairs: null row block");
        hipLaunchKernelGGL(mg_pack_kernel, dim3((unsigned)div_up<size_t>(n * e->S, 256)), dim3(256), 0, s, rows_dev, n, e->S,
                           e->d_colstart, e->d_colblk, e->d_send);
        D2G_HIP(e->ctx, hipGetLastError());
    }
    return D2G_OK;
}
int phase_x1(d2g_allpairs *e, int c, hipStream_t s) {                  // inside a comm group: chunk c of everybody's column slice
    const size_t n = e->n_me(), wm = e->w_blk(e->blk(e->rank, c));
    D2G_HIP(e->ctx, hipSetDevice(e->ctx->device));
    for (int q = 0; q < e->W; ++q) {
        const int bq = e->blk(q, c);
        const size_t wq = e->w_blk(bq);
        const uint64_t *src = e->d_send + n * e->colstart[bq];          // my rows x the columns of block (q, c)
        uint64_t *dst = e->recv_chunk(c) + e->row_lo[q] * wm;           // rows of rank q x my chunk-c columns
        const size_t nq = e->row_lo[q + 1] - e->row_lo[q];
        if (q == e->rank) {
            if (n * wq) D2G_HIP(e->ctx, hipMemcpyAsync(dst, src, n * wq * 8, hipMemcpyDeviceToDevice, s));
            continue;
        }
        if (int rc = comm_send(e->comm, q, src, n * wq * 8, s)) return rc;
        if (int rc = comm_recv(e->comm, q, dst, nq * wm * 8, s)) return rc;
    }
    return D2G_OK;
}
int phase_prepare(d2g_allpairs *e, int c, int b, hipStream_t s) {
    const int bm = e->blk(e->rank, c);
    if (!e->w_blk(bm)) return D2G_OK;                                  // more ranks (x chunks) than register groups: nothing here
    D2G_HIP(e->ctx, hipSetDevice(e->ctx->device));
    if (!e->local[c])
        if (int rc = d2g_bitslice_exporter_create(e->ctx, e->N, e->w_blk(bm), &e->local[c])) return rc;
    // my groups go straight to their place in the gathered operand, with their meta and this block's status word
    d2g_bitslice_set_export_target(e->local[c], e->d_planes[b] + e->gpos[bm] * e->gw, e->d_meta[b] + e->gpos[bm], e->d_meta[b] + e->ng + bm);
    return d2g_bitslice_prepare_slice(e->ctx, e->local[c], e->recv_chunk(c), s);
}
int phase_x2(d2g_allpairs *e, int c, int b, hipStream_t s) {           // inside a comm group: chunk c of everybody's groups
    D2G_HIP(e->ctx, hipSetDevice(e->ctx->device));
    const int bm = e->blk(e->rank, c);
    const size_t gm = e->g_blk(bm);
    for (int q = 0; q < e->W; ++q) {
        if (q == e->rank) continue;
        const int bq = e->blk(q, c);
        const size_t gq = e->g_blk(bq);
        if (int rc = comm_send(e->comm, q, e->d_planes[b] + e->gpos[bm] * e->gw, gm * e->gw * 4, s)) return rc;
        if (int rc = comm_send(e->comm, q, e->d_meta[b] + e->gpos[bm], gm * 4, s)) return rc;
        if (int rc = comm_send(e->comm, q, e->d_meta[b] + e->ng + bm, gm ? 4 : 0, s)) return rc;
        if (int rc = comm_recv(e->comm, q, e->d_planes[b] + e->gpos[bq] * e->gw, gq * e->gw * 4, s)) return rc;
        if (int rc = comm_recv(e->comm, q, e->d_meta[b] + e->gpos[bq], gq * 4, s)) return rc;
        if (int rc = comm_recv(e->comm, q, e->d_meta[b] + e->ng + bq, gq ? 4 : 0, s)) return rc;
    }
    return D2G_OK;
}

#define MG_TRY(call) do { if (int rc__ = (call)) return rc__; } while (0)

// exchange + sharded prepare for n ranks driven by this thread (n = 1: one process per GPU).  cs[i] = the stream the
// compute phases of engine i are enqueued on (the caller's stream, or the engine's own in the pipelined form); the
// exchanges run on the engines' xs streams.  Afterwards cs[i] holds the complete plane stream of buffer bufs[i].
int prepare_many(d2g_allpairs **es, int n, const uint64_t *const *rows, const int *bufs, hipStream_t const *cs) {
    const int C = es[0]->C;
    for (int i = 0; i < n; ++i) MG_TRY(eng_alloc_buffer(es[i], bufs[i]));
    for (int i = 0; i < n; ++i) {
        d2g_allpairs *e = es[i];
        MG_TRY(pt_step_begin(e, cs[i]));
        MG_TRY(pt_begin(e, D2G_PHASE_PACK, 0, cs[i]));
        MG_TRY(phase_pack(e, rows[i], cs[i]));
        MG_TRY(pt_end(e, D2G_PHASE_PACK, 0, cs[i]));
        D2G_HIP(e->ctx, hipEventRecord(e->ev_pack, cs[i]));
        D2G_HIP(e->ctx, hipStreamWaitEvent(e->xs, e->ev_pack, 0));
        // the slab's fill (HBM-bound, depends on nothing) while the row -> column exchange is on the links
        if (e->pre_out && e->full[bufs[i]] && e->full[bufs[i]]->sparse_ok) {
            MG_TRY(pt_begin(e, D2G_PHASE_FILL, 0, cs[i]));
            MG_TRY(d2g_bitslice_prefill(e->ctx, e->full[bufs[i]], e->r0, e->r1, e->pre_lut ? nullptr : (uint32_t *)e->pre_out, e->pre_lut, e->pre_lut ? (float *)e->pre_out : nullptr, cs[i]));
            MG_TRY(pt_end(e, D2G_PHASE_FILL, 0, cs[i]));
        }
        e->pre_out = nullptr; e->pre_lut = nullptr;
    }
    for (int c = 0; c < C; ++c) {                                       // all row->column exchanges, back to back on xs
        for (int i = 0; i < n; ++i) MG_TRY(pt_begin(es[i], D2G_PHASE_X1, c, es[i]->xs));
        MG_TRY(comm_group_begin(es[0]->comm));
        for (int i = 0; i < n; ++i) if (int rc = phase_x1(es[i], c, es[i]->xs)) { (void)comm_group_end(es[0]->comm); return rc; }
        MG_TRY(comm_group_end(es[0]->comm));
        for (int i = 0; i < n; ++i) { D2G_HIP(es[i]->ctx, hipSetDevice(es[i]->ctx->device)); D2G_HIP(es[i]->ctx, hipEventRecord(es[i]->ev_x1[c], es[i]->xs)); MG_TRY(pt_end(es[i], D2G_PHASE_X1, c, es[i]->xs)); }
    }
    for (int c = 0; c < C; ++c)                                         // chunk c is prepared while chunk c+1 is still arriving
        for (int i = 0; i < n; ++i) {
            d2g_allpairs *e = es[i];
            D2G_HIP(e->ctx, hipSetDevice(e->ctx->device));
            D2G_HIP(e->ctx, hipStreamWaitEvent(cs[i], e->ev_x1[c], 0));
            MG_TRY(pt_begin(e, D2G_PHASE_PREPARE, c, cs[i]));
            MG_TRY(phase_prepare(e, c, bufs[i], cs[i]));
            MG_TRY(pt_end(e, D2G_PHASE_PREPARE, c, cs[i]));
            D2G_HIP(e->ctx, hipEventRecord(e->ev_prep[c], cs[i]));
        }
    for (int c = 0; c < C; ++c) {                                       // its planes leave while chunk c+1 is prepared
        for (int i = 0; i < n; ++i) { D2G_HIP(es[i]->ctx, hipSetDevice(es[i]->ctx->device)); D2G_HIP(es[i]->ctx, hipStreamWaitEvent(es[i]->xs, es[i]->ev_prep[c], 0)); MG_TRY(pt_begin(es[i], D2G_PHASE_X2, c, es[i]->xs)); }
        MG_TRY(comm_group_begin(es[0]->comm));
        for (int i = 0; i < n; ++i) if (int rc = phase_x2(es[i], c, bufs[i], es[i]->xs)) { (void)comm_group_end(es[0]->comm); return rc; }
        MG_TRY(comm_group_end(es[0]->comm));
        for (int i = 0; i < n; ++i) { D2G_HIP(es[i]->ctx, hipSetDevice(es[i]->ctx->device)); D2G_HIP(es[i]->ctx, hipEventRecord(es[i]->ev_x2[c], es[i]->xs)); MG_TRY(pt_end(es[i], D2G_PHASE_X2, c, es[i]->xs)); }
    }
    for (int c = 0; c < C; ++c)                                         // plane stream of chunk c as soon as it is complete
        for (int i = 0; i < n; ++i) {
            d2g_allpairs *e = es[i];
            D2G_HIP(e->ctx, hipSetDevice(e->ctx->device));
            D2G_HIP(e->ctx, hipStreamWaitEvent(cs[i], e->ev_x2[c], 0));
            MG_TRY(pt_begin(e, D2G_PHASE_DERIVE, c, cs[i]));
            MG_TRY(d2g_bitslice_derive_groups(e->ctx, e->full[bufs[i]], (int)e->chunk_g0[c], (int)e->chunk_g0[c + 1], cs[i]));
            MG_TRY(pt_end(e, D2G_PHASE_DERIVE, c, cs[i]));
        }
    for (int i = 0; i < n; ++i) {                                       // every group is here: the operand ordered for the sparse-tile pair phase
        d2g_allpairs *e = es[i];
        if (!e->full[bufs[i]]->sparse_ok) continue;
        D2G_HIP(e->ctx, hipSetDevice(e->ctx->device));
        MG_TRY(pt_begin(e, D2G_PHASE_ORDER, 0, cs[i]));
        MG_TRY(d2g_bitslice_managed_ready(e->ctx, e->full[bufs[i]], cs[i]));
        MG_TRY(pt_end(e, D2G_PHASE_ORDER, 0, cs[i]));
    }
    for (int i = 0; i < n; ++i) es[i]->last = bufs[i];
    return D2G_OK;
}

// One process per GPU: every rank derives the exchange lists from its OWN (N, S, world, chunks) -- chunks can be overridden by an
// environment variable -- and a disagreement would post transfers of different sizes and counts (a hang or silent corruption).
// Each rank sends its four numbers to every other rank once (fixed-size messages: this exchange itself cannot mismatch) and
// compares.  It is also the first traffic over the communicator: a transport that cannot move bytes fails here, at create.
int verify_shape_across_ranks(d2g_allpairs *e) {
    d2g_comm *c = e->comm;
    if (!c->ranked || !c->nccl || e->W < 2) return D2G_OK;
    d2g_ctx *ctx = e->ctx;
    const int W = e->W;
    uint64_t *d_v = nullptr;
    std::vector<uint64_t> all((size_t)W * 4, 0);
    // [3]: chunks in the low word, a hash of the K2 switches (D2G_BS_* / D2G_SP_*: which kernels a rank runs) in the high one
    const uint64_t mine[4] = {(uint64_t)e->N, (uint64_t)e->S, (uint64_t)e->W, (uint64_t)e->C | (d2g_k2_tuning_hash(e->ctx) << 32)};
    D2G_HIP(ctx, hipMalloc((void **)&d_v, (size_t)W * 32));
    int rc = D2G_OK;
    auto fail = [&](int r) { (void)hipFree(d_v); return r; };
    if (hipMemcpy(d_v + (size_t)e->rank * 4, mine, 32, hipMemcpyHostToDevice) != hipSuccess) return fail(D2G_ERR_HIP);
    if ((rc = comm_group_begin(c))) return fail(rc);
    for (int q = 0; q < W && rc == D2G_OK; ++q) {
        if (q == e->rank) continue;
        rc = comm_send(c, q, d_v + (size_t)e->rank * 4, 32, e->xs);
        if (rc == D2G_OK) rc = comm_recv(c, q, d_v + (size_t)q * 4, 32, e->xs);
    }
    const int rc2 = comm_group_end(c);
    if (rc == D2G_OK) rc = rc2;
    if (rc != D2G_OK) return fail(rc);
    if (hipStreamSynchronize(e->xs) != hipSuccess || hipMemcpy(all.data(), d_v, (size_t)W * 32, hipMemcpyDeviceToHost) != hipSuccess) return fail(D2G_ERR_HIP);
    (void)hipFree(d_v);
    for (int q = 0; q < W; ++q)
        if (std::memcmp(&all[(size_t)q * 4], mine, 32) != 0) {
            char buf[256];
            std::snprintf(buf, sizeof buf, "allpairs: rank %d has (N=%llu, S=%llu, world=%llu, chunks=%llu, switches=%08llx), rank %d has (N=%llu, S=%llu, world=%llu, chunks=%llu, switches=%08llx)"
                          " -- the same shape, D2G_MGPU_CHUNKS and D2G_BS_* / D2G_SP_* switches are required on every rank", e->rank, (unsigned long long)mine[0], (unsigned long long)mine[1],
                          (unsigned long long)mine[2], (unsigned long long)(mine[3] & 0xFFFFFFFFull), (unsigned long long)(mine[3] >> 32), q, (unsigned long long)all[(size_t)q * 4],
                          (unsigned long long)all[(size_t)q * 4 + 1], (unsigned long long)all[(size_t)q * 4 + 2], (unsigned long long)(all[(size_t)q * 4 + 3] & 0xFFFFFFFFull),
                          (unsigned long long)(all[(size_t)q * 4 + 3] >> 32));
            ctx->last_error = buf;
            return D2G_ERR_INVALID;
        }
    return D2G_OK;
}

int check_group(d2g_allpairs **es, int n) {
    if (!es || n < 1 || !es[0]) return D2G_ERR_INVALID;
    for (int i = 0; i < n; ++i) {
        if (!es[i]) return D2G_ERR_INVALID;
        D2G_CHECK(es[i]->ctx, es[i]->N == es[0]->N && es[i]->S == es[0]->S && es[i]->W == es[0]->W && es[i]->C == es[0]->C, "allpairs: engines of different shapes");
        if (d2g_k2_tuning_hash(es[i]->ctx) != d2g_k2_tuning_hash(es[0]->ctx)) {
            const char *msg = "allpairs: the ranks' contexts resolved different D2G_BS_* / D2G_SP_* switches (d2g_ctx_tuning): every rank must run the same kernels";
            es[i]->ctx->last_error = msg; es[0]->ctx->last_error = msg;       // whichever context the caller asks
            return D2G_ERR_INVALID;
        }
    }
    return D2G_OK;
}

}  // namespace

extern "C" {

// ---------------------------------------------------------------------------------------------- communicator
int d2g_comm_unique_id(void *id_out) {
    if (!id_out) return D2G_ERR_INVALID;
    Rccl *r = rccl();
    if (!r->err.empty()) return D2G_ERR_UNSUPPORTED;
    ncclUniqueId id;
    if (r->GetUniqueId(&id) != ncclSuccess) return D2G_ERR_HIP;
    static_assert(sizeof(ncclUniqueId) == D2G_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    std::memcpy(id_out, &id, sizeof id);
    return D2G_OK;
}

int d2g_comm_create(d2g_ctx *ctx, const void *id, int rank, int world, d2g_comm **out) {
    if (!ctx || !out) return D2G_ERR_INVALID;
    *out = nullptr;
    D2G_CHECK(ctx, world >= 1 && rank >= 0 && rank < world, "comm: bad rank/world");
    d2g_comm *c = new (std::nothrow) d2g_comm();
    if (!c) return D2G_ERR_NOMEM;
    c->ctx = ctx; c->rank = rank; c->world = world;
    if (world > 1 || id) {                       // a single rank with an id still forms a real (one-member) RCCL communicator
        Rccl *r = rccl();
        if (!r->err.empty()) { ctx->last_error = r->err; delete c; return D2G_ERR_UNSUPPORTED; }
        if (!id) { ctx->last_error = "comm: null unique id"; delete c; return D2G_ERR_INVALID; }
        ncclUniqueId uid;
        std::memcpy(&uid, id, sizeof uid);
        if (hipSetDevice(ctx->device) != hipSuccess) { delete c; return D2G_ERR_HIP; }
        const ncclResult_t rc = r->CommInitRank(&c->nccl, world, uid, rank);
        if (rc != ncclSuccess) { ctx->last_error = std::string("ncclCommInitRank: ") + r->GetErrorString(rc); delete c; return D2G_ERR_HIP; }
        c->ranked = true;
    }
    *out = c;
    return D2G_OK;
}

int d2g_comm_create_all(d2g_ctx **ctxs, int nctx, d2g_comm **comms_out) {
    if (!ctxs || !comms_out || nctx < 1) return D2G_ERR_INVALID;
    for (int i = 0; i < nctx; ++i) { if (!ctxs[i]) return D2G_ERR_INVALID; comms_out[i] = nullptr; }
    bool distinct = true;
    for (int i = 0; i < nctx; ++i)
        for (int j = 0; j < i; ++j) distinct &= ctxs[i]->device != ctxs[j]->device;
    std::vector<d2g_comm *> cs(nctx);
    for (int i = 0; i < nctx; ++i) {
        cs[i] = new (std::nothrow) d2g_comm();
        if (!cs[i]) { for (int j = 0; j < i; ++j) delete cs[j]; return D2G_ERR_NOMEM; }
        cs[i]->ctx = ctxs[i]; cs[i]->rank = i; cs[i]->world = nctx;
    }
    const char *force = std::getenv("D2G_COMM_LOOPBACK");
    if (nctx > 1 && distinct && !(force && force[0] == '1')) {          // one RCCL clique over the devices of this process
        Rccl *r = rccl();
        std::vector<ncclComm_t> nc(nctx);
        std::vector<int> devs(nctx);
        for (int i = 0; i < nctx; ++i) devs[i] = ctxs[i]->device;
        const ncclResult_t rc = r->err.empty() ? r->CommInitAll(nc.data(), nctx, devs.data()) : ncclSystemError;
        if (rc == ncclSuccess) {
            for (int i = 0; i < nctx; ++i) { cs[i]->nccl = nc[i]; comms_out[i] = cs[i]; }
            return D2G_OK;
        }
        ctxs[0]->last_error = r->err.empty() ? std::string("ncclCommInitAll: ") + r->GetErrorString(rc) : r->err;
        for (auto *c : cs) delete c;
        return D2G_ERR_HIP;
    }
    if (nctx > 1) {                                                      // shared device(s): loopback transport
        auto lg = std::make_shared<LocalGroup>();
        lg->members.assign(cs.begin(), cs.end());
        for (auto *c : cs) c->lg = lg;
    }
    for (int i = 0; i < nctx; ++i) comms_out[i] = cs[i];
    return D2G_OK;
}

void d2g_comm_destroy(d2g_comm *c) {
    if (!c) return;
    if (c->nccl) { (void)hipSetDevice(c->ctx->device); (void)rccl()->CommDestroy(c->nccl); }
    delete c;
}
int d2g_comm_rank(const d2g_comm *c) { return c ? c->rank : -1; }
int d2g_comm_world(const d2g_comm *c) { return c ? c->world : -1; }
int d2g_comm_is_rccl(const d2g_comm *c) { return c && c->nccl != nullptr; }

// the whole matrix from the host to every GPU of this process -- one H2D to ctxs[0], then ONE
// RCCL broadcast over xGMI (loopback: plain copies).  sig_dev_out[i] is allocated on ctxs[i] (d2g_free).
// On failure nothing stays allocated and sig_dev_out[] is all NULL.
static int bcast_sigs_impl(d2g_ctx **ctxs, d2g_comm **comms, int nctx, const uint64_t *host_sig, size_t bytes, uint64_t **sig_dev_out) {
    d2g_ctx *c0 = ctxs[0];
    for (int i = 0; i < nctx; ++i) {
        D2G_CHECK(c0, ctxs[i] && comms[i] && comms[i]->ctx == ctxs[i] && comms[i]->world == nctx && comms[i]->rank == i, "bcast_sigs: comm/ctx mismatch");
        D2G_HIP(ctxs[i], hipSetDevice(ctxs[i]->device));
        D2G_HIP(ctxs[i], hipMalloc((void **)&sig_dev_out[i], bytes));
    }
    D2G_HIP(c0, hipSetDevice(c0->device));
    D2G_HIP(c0, hipMemcpyAsync(sig_dev_out[0], host_sig, bytes, hipMemcpyHostToDevice, nullptr));
    if (nctx > 1) {
        if (comms[0]->nccl) {
            D2G_NCCL(c0, rccl()->GroupStart());
            for (int i = 0; i < nctx; ++i) {
                (void)hipSetDevice(ctxs[i]->device);
                const ncclResult_t rc = rccl()->Broadcast(sig_dev_out[0], sig_dev_out[i], bytes, ncclUint8, 0, comms[i]->nccl, nullptr);
                if (rc != ncclSuccess) { (void)rccl()->GroupEnd(); c0->last_error = std::string("ncclBroadcast: ") + rccl()->GetErrorString(rc); return D2G_ERR_HIP; }
            }
            D2G_NCCL(c0, rccl()->GroupEnd());
        } else {
            // the group is always closed again: an error in between must not leave stale operations for the next group
            int rc = comm_group_begin(comms[0]);
            for (int i = 1; i < nctx && rc == D2G_OK; ++i) {
                rc = comm_send(comms[0], i, sig_dev_out[0], bytes, nullptr);
                if (rc == D2G_OK) rc = comm_recv(comms[i], 0, sig_dev_out[i], bytes, nullptr);
            }
            if (rc != D2G_OK) { if (comms[0]->lg) { comms[0]->lg->pending.clear(); comms[0]->lg->depth = 0; } return rc; }
            if ((rc = comm_group_end(comms[0]))) return rc;
        }
    }
    for (int i = 0; i < nctx; ++i) {
        D2G_HIP(ctxs[i], hipSetDevice(ctxs[i]->device));
        D2G_HIP(ctxs[i], hipStreamSynchronize(nullptr));
    }
    return D2G_OK;
}
int d2g_bcast_sigs(d2g_ctx **ctxs, d2g_comm **comms, int nctx, const uint64_t *host_sig, size_t N, size_t S, uint64_t **sig_dev_out) {
    if (!ctxs || !comms || nctx < 1 || !sig_dev_out || !ctxs[0]) return D2G_ERR_INVALID;
    D2G_CHECK(ctxs[0], host_sig != nullptr && N >= 1 && S >= 1, "bcast_sigs: empty matrix");
    for (int i = 0; i < nctx; ++i) sig_dev_out[i] = nullptr;
    const int rc = bcast_sigs_impl(ctxs, comms, nctx, host_sig, N * S * 8, sig_dev_out);
    if (rc != D2G_OK)
        for (int i = 0; i < nctx; ++i)
            if (sig_dev_out[i] && ctxs[i]) { (void)hipSetDevice(ctxs[i]->device); (void)hipDeviceSynchronize(); (void)hipFree(sig_dev_out[i]); sig_dev_out[i] = nullptr; }
    return rc;
}

// ---------------------------------------------------------------------------------------------- engine
int d2g_allpairs_create(d2g_ctx *ctx, d2g_comm *comm, size_t N, size_t S, d2g_allpairs **out) {
    if (!ctx || !comm || !out) return D2G_ERR_INVALID;
    *out = nullptr;
    D2G_CHECK(ctx, comm->ctx == ctx, "allpairs: the communicator belongs to another context");
    D2G_CHECK(ctx, N >= 1 && S >= 1 && N < (1ull << 30) && S < (1ull << 31), "allpairs: bad shape");
    D2G_HIP(ctx, hipSetDevice(ctx->device));
    d2g_allpairs *e = new (std::nothrow) d2g_allpairs();
    if (!e) return D2G_ERR_NOMEM;
    e->ctx = ctx; e->comm = comm; e->N = N; e->S = S; e->W = comm->world; e->rank = comm->rank;
    e->row_lo = even_split<size_t>(N, e->W);
    if (int rc = d2g_operand_layout(N, S, &e->gw, &e->ng)) { delete e; return rc; }
    // chunks per rank: at least two 32-register groups each, at most MG_MAX_CHUNKS (a function of the shape only: every rank agrees)
    e->C = e->W == 1 ? 1 : (int)std::min<size_t>(MG_MAX_CHUNKS, std::max<size_t>(1, e->ng / e->W / 2));
    if (const char *env = e->ctx->tune.get("D2G_MGPU_CHUNKS")) { const int v = std::atoi(env); if (v >= 1 && v <= MG_MAX_CHUNKS) e->C = v; }
    const int nb = e->W * e->C;
    const std::vector<size_t> grp_lo = even_split<size_t>(e->ng, e->W);
    e->blk_g.assign(nb + 1, e->ng);
    for (int q = 0; q < e->W; ++q) {
        const std::vector<size_t> cb = even_split<size_t>(grp_lo[q + 1] - grp_lo[q], e->C);
        for (int c = 0; c < e->C; ++c) e->blk_g[e->blk(q, c)] = grp_lo[q] + cb[c];
    }
    e->colstart.resize(nb + 1);
    for (int b = 0; b <= nb; ++b) e->colstart[b] = (uint32_t)std::min<size_t>(S, e->blk_g[b] * 32);
    e->gpos.resize(nb);
    e->chunk_g0.assign(e->C + 1, 0);
    {
        size_t g = 0;
        for (int c = 0; c < e->C; ++c) {
            e->chunk_g0[c] = g;
            for (int q = 0; q < e->W; ++q) { e->gpos[e->blk(q, c)] = g; g += e->g_blk(e->blk(q, c)); }
        }
        e->chunk_g0[e->C] = g;                                       // == ng
    }
    std::vector<uint16_t> colblk(S);
    for (int b = 0; b < nb; ++b) for (uint32_t c = e->colstart[b]; c < e->colstart[b + 1]; ++c) colblk[c] = (uint16_t)b;
    std::vector<size_t> ob(e->W + 1);
    d2g_ut_partition(N, e->W, ob.data());
    e->r0 = ob[e->rank]; e->r1 = ob[e->rank + 1];
    hipError_t he;
    if ((he = hipMalloc((void **)&e->d_colstart, (nb + 1) * 4)) != hipSuccess ||
        (he = hipMemcpy(e->d_colstart, e->colstart.data(), (nb + 1) * 4, hipMemcpyHostToDevice)) != hipSuccess ||
        (he = hipMalloc((void **)&e->d_colblk, S * 2)) != hipSuccess ||
        (he = hipMemcpy(e->d_colblk, colblk.data(), S * 2, hipMemcpyHostToDevice)) != hipSuccess ||
        (he = hipMalloc((void **)&e->d_send, std::max<size_t>(e->n_me() * S, 1) * 8)) != hipSuccess ||
        (he = hipMalloc((void **)&e->d_recv, std::max<size_t>(N * e->s_me(), 1) * 8)) != hipSuccess ||
        (he = hipStreamCreateWithFlags(&e->xs, hipStreamNonBlocking)) != hipSuccess ||
        (he = hipEventCreateWithFlags(&e->ev_pack, hipEventDisableTiming)) != hipSuccess ||
        (he = hipEventCreateWithFlags(&e->plain_done, hipEventDisableTiming)) != hipSuccess) {
        ctx->last_error = std::string("allpairs alloc: ") + hipGetErrorString(he);
        d2g_allpairs_destroy(e);
        return he == hipErrorOutOfMemory ? D2G_ERR_NOMEM : D2G_ERR_HIP;
    }
    for (int c = 0; c < e->C; ++c)
        if ((he = hipEventCreateWithFlags(&e->ev_x1[c], hipEventDisableTiming)) != hipSuccess ||
            (he = hipEventCreateWithFlags(&e->ev_prep[c], hipEventDisableTiming)) != hipSuccess ||
            (he = hipEventCreateWithFlags(&e->ev_x2[c], hipEventDisableTiming)) != hipSuccess) {
            ctx->last_error = std::string("allpairs alloc: ") + hipGetErrorString(he);
            d2g_allpairs_destroy(e);
            return D2G_ERR_HIP;
        }
    // the exchange of (N, S, world, chunks, switches) comes BEFORE the large allocations: a rank that then runs out of memory returns
    // without leaving its peers blocked in this (collective) call
    if (int rc = verify_shape_across_ranks(e)) { d2g_allpairs_destroy(e); return rc; }
    if (int rc = eng_alloc_buffer(e, 0)) { d2g_allpairs_destroy(e); return rc; }
    *out = e;
    return D2G_OK;
}

void d2g_allpairs_destroy(d2g_allpairs *e) {
    if (!e) return;
    (void)hipSetDevice(e->ctx->device);
    (void)hipDeviceSynchronize();
    for (int b = 0; b < 2; ++b) {
        d2g_cmp_set_destroy(e->full[b]);
        (void)hipFree(e->d_planes[b]); (void)hipFree(e->d_meta[b]);
        if (e->x_done[b]) (void)hipEventDestroy(e->x_done[b]);
        if (e->p_done[b]) (void)hipEventDestroy(e->p_done[b]);
    }
    for (int c = 0; c < MG_MAX_CHUNKS; ++c) {
        d2g_cmp_set_destroy(e->local[c]);
        if (e->ev_x1[c]) (void)hipEventDestroy(e->ev_x1[c]);
        if (e->ev_prep[c]) (void)hipEventDestroy(e->ev_prep[c]);
        if (e->ev_x2[c]) (void)hipEventDestroy(e->ev_x2[c]);
    }
    pt_clear(e);
    if (e->ev_pack) (void)hipEventDestroy(e->ev_pack);
    if (e->in_ready) (void)hipEventDestroy(e->in_ready);
    if (e->plain_done) (void)hipEventDestroy(e->plain_done);
    (void)hipFree(e->d_colstart); (void)hipFree(e->d_colblk); (void)hipFree(e->d_send); (void)hipFree(e->d_recv);
    if (e->xs) (void)hipStreamDestroy(e->xs);
    if (e->ps) (void)hipStreamDestroy(e->ps);
    delete e;
}

int d2g_allpairs_rows_held(const d2g_allpairs *e, size_t *lo, size_t *hi) {
    if (!e) return D2G_ERR_INVALID;
    if (lo) *lo = e->row_lo[e->rank];
    if (hi) *hi = e->row_lo[e->rank + 1];
    return D2G_OK;
}
int d2g_allpairs_rows_computed(const d2g_allpairs *e, size_t *r0, size_t *r1) {
    if (!e) return D2G_ERR_INVALID;
    if (r0) *r0 = e->r0;
    if (r1) *r1 = e->r1;
    return D2G_OK;
}
const d2g_cmp_set *d2g_allpairs_operand(const d2g_allpairs *e) { return e ? e->full[e->last] : nullptr; }
int d2g_allpairs_chunks(const d2g_allpairs *e) { return e ? e->C : D2G_ERR_INVALID; }
// every rank's (and chunk's) prepare reports through the gathered status words: the same answer on every rank
int d2g_allpairs_status(d2g_allpairs *e, void *stream) {
    if (!e) return D2G_ERR_INVALID;
    D2G_HIP(e->ctx, hipSetDevice(e->ctx->device));
    return d2g_cmp_set_status(e->ctx, e->full[e->last], stream);
}

// per-phase times of one step: see d2g.h
int d2g_allpairs_set_phase_timing(d2g_allpairs *e, int on) {
    if (!e) return D2G_ERR_INVALID;
    e->phase_timing = on != 0;
    if (!on) { (void)hipSetDevice(e->ctx->device); (void)hipDeviceSynchronize(); pt_clear(e); }
    return D2G_OK;
}
int d2g_allpairs_phase_times(d2g_allpairs *e, int cap, int *n_out, int *kind, int *chunk, float *start_ms, float *dur_ms) {
    if (!e || !n_out) return D2G_ERR_INVALID;
    *n_out = 0;
    if (!e->ev_step0) return D2G_OK;
    D2G_HIP(e->ctx, hipSetDevice(e->ctx->device));
    D2G_HIP(e->ctx, hipDeviceSynchronize());
    int n = 0;
    for (const auto &p : e->pev) {
        if (n >= cap) break;
        float t0 = 0.f, dt = 0.f;
        D2G_HIP(e->ctx, hipEventElapsedTime(&t0, e->ev_step0, p.a));
        D2G_HIP(e->ctx, hipEventElapsedTime(&dt, p.a, p.b));
        if (kind) kind[n] = p.kind;
        if (chunk) chunk[n] = p.chunk;
        if (start_ms) start_ms[n] = t0;
        if (dur_ms) dur_ms[n] = dt;
        ++n;
    }
    *n_out = n;
    return D2G_OK;
}

int d2g_allpairs_sparse_info(d2g_allpairs *e, uint32_t *info4) {
    if (!e || !info4) return D2G_ERR_INVALID;
    info4[0] = info4[1] = info4[2] = info4[3] = 0;
    D2G_HIP(e->ctx, hipSetDevice(e->ctx->device));
    D2G_HIP(e->ctx, hipDeviceSynchronize());
    if (e->last < 0 || !e->full[e->last]) return D2G_OK;
    return d2g_bitslice_sparse_info(e->ctx, e->full[e->last], nullptr, info4);
}

int d2g_allpairs_prepare_all(d2g_allpairs **engs, int n, const uint64_t *const *rows_dev, void *const *streams) {
    if (int rc = check_group(engs, n)) return rc;
    std::vector<int> bufs(n, 0);
    std::vector<hipStream_t> ss(n);
    for (int i = 0; i < n; ++i) {
        ss[i] = as_stream(streams ? streams[i] : nullptr);
        // the pipelined form shares the send/receive buffers and the exporter sets: whatever it still has in flight on the
        // engine's own stream must be done before this step's pack overwrites them
        d2g_allpairs *e = engs[i];
        if (e->ps) {
            D2G_HIP(e->ctx, hipSetDevice(e->ctx->device));
            for (int b = 0; b < 2; ++b) if (e->p_valid[b]) D2G_HIP(e->ctx, hipStreamWaitEvent(ss[i], e->x_done[b], 0));
        }
    }
    if (int rc = prepare_many(engs, n, rows_dev, bufs.data(), ss.data())) return rc;
    for (int i = 0; i < n; ++i) {
        D2G_HIP(engs[i]->ctx, hipSetDevice(engs[i]->ctx->device));
        D2G_HIP(engs[i]->ctx, hipEventRecord(engs[i]->plain_done, ss[i]));
        engs[i]->plain_valid = true;
    }
    return D2G_OK;
}
int d2g_allpairs_prepare_dev(d2g_allpairs *e, const uint64_t *my_rows_dev, void *stream) {
    return d2g_allpairs_prepare_all(&e, 1, &my_rows_dev, &stream);
}

// one whole step per engine: exchange + sharded prepare + this rank's slab of pairs (rows_computed)
int d2g_allpairs_step_all(d2g_allpairs **engs, int n, const uint64_t *const *rows_dev, const float *const *lut_dev,
                          void *const *out_dev, void *const *streams) {
    if (int rc = check_group(engs, n)) return rc;
    // (the announced slab is engine state only for the duration of this call -- whatever exit the prepare takes, a later
    // d2g_allpairs_prepare_dev on the engine must not pre-fill a stale, possibly freed, pointer)
    struct Clear { d2g_allpairs **e; int n; ~Clear() { for (int i = 0; i < n; ++i) { e[i]->pre_out = nullptr; e[i]->pre_lut = nullptr; } } } clear{engs, n};
    for (int i = 0; i < n; ++i) { engs[i]->pre_out = out_dev ? out_dev[i] : nullptr; engs[i]->pre_lut = (lut_dev && lut_dev[i]) ? lut_dev[i] : nullptr; }
    if (int rc = d2g_allpairs_prepare_all(engs, n, rows_dev, streams)) return rc;
    for (int i = 0; i < n; ++i) {
        d2g_allpairs *e = engs[i];
        void *s = streams ? streams[i] : nullptr;
        MG_TRY(pt_begin(e, D2G_PHASE_PAIR, 0, as_stream(s)));
        int rc = (lut_dev && lut_dev[i]) ? d2g_cmp_lut_ut_dev(e->ctx, e->full[e->last], e->r0, e->r1, lut_dev[i], (float *)out_dev[i], s)
                                         : d2g_cmp_eqcount_ut_dev(e->ctx, e->full[e->last], e->r0, e->r1, (uint32_t *)out_dev[i], s);
        if (rc) return rc;
        MG_TRY(pt_end(e, D2G_PHASE_PAIR, 0, as_stream(s)));
    }
    return D2G_OK;
}
int d2g_allpairs_step_lut_dev(d2g_allpairs *e, const uint64_t *my_rows_dev, const float *lut_dev, float *out_dev, void *stream) {
    void *o = out_dev;
    return d2g_allpairs_step_all(&e, 1, &my_rows_dev, &lut_dev, &o, &stream);
}
int d2g_allpairs_step_eqcount_dev(d2g_allpairs *e, const uint64_t *my_rows_dev, uint32_t *out_dev, void *stream) {
    void *o = out_dev;
    return d2g_allpairs_step_all(&e, 1, &my_rows_dev, nullptr, &o, &stream);
}

// Software-pipelined step for a STREAM of matrices (one rank per calling thread): the exchange + prepare of this call
// run on the engine's own streams over the operand buffer the PREVIOUS call is not using, so they overlap the previous
// call's pair kernel, which is still busy on `stream`.  Results land in out_dev in call order on `stream`.
// input_ready: 0 = my_rows_dev may have been produced by work queued on `stream` just before this call (the
// exchange waits for everything queued there so far -- safe, but it then also waits for the previous pair
// kernel); 1 = the input is already complete (synchronised earlier): no dependency, full overlap.
// Plain steps (prepare/step entry points) and pipelined ones may be mixed freely: each form waits for what the other
// still has in flight on the buffers they share.
int d2g_allpairs_enqueue_lut_dev(d2g_allpairs *e, const uint64_t *my_rows_dev, const float *lut_dev, float *out_dev, void *stream,
                                 int input_ready) {
    if (!e) return D2G_ERR_INVALID;
    d2g_ctx *ctx = e->ctx;
    D2G_CHECK(ctx, lut_dev && out_dev, "allpairs: null lut/output");
    D2G_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t main = as_stream(stream);
    if (!e->ps) {
        D2G_HIP(ctx, hipStreamCreateWithFlags(&e->ps, hipStreamNonBlocking));
        for (int b = 0; b < 2; ++b) {
            D2G_HIP(ctx, hipEventCreateWithFlags(&e->x_done[b], hipEventDisableTiming));
            D2G_HIP(ctx, hipEventCreateWithFlags(&e->p_done[b], hipEventDisableTiming));
        }
        D2G_HIP(ctx, hipEventCreateWithFlags(&e->in_ready, hipEventDisableTiming));
    }
    const int b = (int)(e->nsteps & 1);
    if (!input_ready) {
        D2G_HIP(ctx, hipEventRecord(e->in_ready, main));
        D2G_HIP(ctx, hipStreamWaitEvent(e->ps, e->in_ready, 0));
    }
    if (e->plain_valid) D2G_HIP(ctx, hipStreamWaitEvent(e->ps, e->plain_done, 0));    // a plain step queued earlier still owns the shared buffers
    if (e->p_valid[b]) D2G_HIP(ctx, hipStreamWaitEvent(e->ps, e->p_done[b], 0));      // buffer b is free once pair(step - 2) is done
    // send/recv/exporter sets are shared by consecutive steps: ps runs their users in order, and the operand
    // buffer is the only thing the pair kernel on `main` still reads
    hipStream_t ps = e->ps;
    if (int rc = prepare_many(&e, 1, &my_rows_dev, &b, &ps)) return rc;
    D2G_HIP(ctx, hipEventRecord(e->x_done[b], e->ps));
    D2G_HIP(ctx, hipStreamWaitEvent(main, e->x_done[b], 0));
    if (int rc = d2g_cmp_lut_ut_dev(ctx, e->full[b], e->r0, e->r1, lut_dev, out_dev, stream)) return rc;
    D2G_HIP(ctx, hipEventRecord(e->p_done[b], main));
    e->p_valid[b] = true;
    ++e->nsteps;
    return D2G_OK;
}

}  // extern "C"

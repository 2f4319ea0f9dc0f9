// d2g_k1.hip -- K1: 2-bit packed bases -> One-Permutation SetSketch registers (gfx950).
//
// Replaces, per k-mer, the reference chain
//   bns::Encoder::for_each (absent bonsai; call site src/fastxsketch.cpp:416-417)
//   -> maskfn            src/enums.h:136-140      h1 = Wang(kmer ^ XORMASK)
//   -> DHasher/BHasher   src/oph.h:44-53,59       id = Wang(h1 ^ seed_ ^ 0x533f8c2151b20f97)
//   -> update            src/oph.h:176-211        reg[id mod m] = min(reg[..], id)
//
// Parallel decomposition (the reference runs one thread per file, fastxsketch.cpp:302):
//   * a genome's k-mers are cut into chunks of 64 consecutive k-mer start positions; one lane
//     owns one chunk at a time (k-1 base warm-up from a 64-bit window, then 64 rolled steps);
//   * a 256-lane workgroup owns up to 1024 chunks of ONE genome and keeps that genome's m
//     registers in LDS (ds_min_u64 behind a read-compare filter), then merges them into HBM
//     with global_atomic_umin_x2.  min is associative and commutative, so any split is exact.
//   * packed bases are read straight from HBM/L2: lane l reads the 16..20 bytes of its chunk,
//     consecutive lanes read consecutive 16-byte pieces (1 KiB per wave-instruction).
#include "d2g_k1.h"
#include <algorithm>
#include <cstring>
#include <new>
#include <vector>

namespace {

struct K1Args {
    KmerArgs km;
    uint64_t *regs_out;            // [n][m], pre-filled with ~0
    uint64_t xormask;
    uint64_t ophxor;
    uint32_t m;
};

template <bool POW2, bool USE_LDS>
__global__ __launch_bounds__(K1_THREADS) void k1_oph_kernel(K1Args a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t lreg[];
    const int tid = threadIdx.x;
    const uint32_t g = a.km.blk_genome[blockIdx.x];
    const uint32_t m = a.m;
    uint64_t *gout = a.regs_out + (size_t)g * m;

    if (USE_LDS) {
        for (uint32_t i = tid; i < m; i += K1_THREADS) lreg[i] = ~0ull;
        __syncthreads();
    }
    const uint64_t xormask = a.xormask, ophxor = a.ophxor;
    d2g_for_each_kmer(a.km, [&](uint64_t x) {
        const uint64_t id = wang64(wang64(x ^ xormask) ^ ophxor);
        // Schismatic<uint32_t>::mod(size_t): argument narrowed to 32 bits (oph.h:184)
        const uint32_t idx = POW2 ? ((uint32_t)id & (m - 1)) : ((uint32_t)id % m);
        if (USE_LDS) {
            if (id < lreg[idx]) atomicMin((unsigned long long *)&lreg[idx], (unsigned long long)id);
        } else {
            if (id < gout[idx]) atomicMin((unsigned long long *)&gout[idx], (unsigned long long)id);
        }
    });
    if (USE_LDS) {
        __syncthreads();
        for (uint32_t i = tid; i < m; i += K1_THREADS) {
            // (a read filter -- skip the atomic when a plain load already shows a value <= v -- was measured: atomic
            // write traffic 0.63 -> 0.12 GB per 1000 genomes, but the 8 KB of register reads per workgroup that replace it
            // and the dependent load at the end of every workgroup made the kernel 1.7 % slower; traffic is not its bound)
            const uint64_t v = lreg[i];
            if (v != ~0ull) atomicMin((unsigned long long *)&gout[i], (unsigned long long)v);
        }
    }
}

template <class T>
static int upload(d2g_ctx *ctx, const std::vector<T> &h, T **d) {
    D2G_HIP(ctx, hipMalloc((void **)d, std::max<size_t>(h.size(), 1) * sizeof(T)));
    if (!h.empty()) D2G_HIP(ctx, hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return D2G_OK;
}

}  // namespace

extern "C" {

void d2g_oph_plan_destroy(d2g_oph_plan *p) {
    if (!p) return;
    (void)hipSetDevice(p->ctx->device);
    (void)hipFree(p->d_run_start); (void)hipFree(p->d_run_len); (void)hipFree(p->d_run_chunk_off);
    (void)hipFree(p->d_blk_genome); (void)hipFree(p->d_blk_chunk0); (void)hipFree(p->d_blk_nchunks);
    (void)hipFree(p->d_blk_run_lo); (void)hipFree(p->d_blk_run_hi);
    delete p;
}

uint64_t d2g_oph_plan_nkmers(const d2g_oph_plan *p) { return p ? p->nkmers : 0; }
uint64_t d2g_oph_plan_nbases(const d2g_oph_plan *p) { return p ? p->nbases : 0; }

}  // extern "C"

int d2g_build_plan_host(d2g_ctx *ctx, const uint32_t *run_len, size_t nrun, const uint64_t *genome_run_off, size_t n, int k, PlanHost &p) {
    D2G_CHECK(ctx, genome_run_off && (nrun == 0 || run_len), "oph plan: null table");
    if (k < 1 || k > 32) { ctx->last_error = "k must be in [1,32] (exact 2-bit encoding path)"; return D2G_ERR_UNSUPPORTED; }
    D2G_CHECK(ctx, genome_run_off[n] == nrun, "oph plan: genome_run_off[n] != nrun");
    D2G_CHECK(ctx, nrun < (1ull << 32), "oph plan: too many runs");
    p.chunk_off.assign(nrun + 1, 0);
    for (size_t r = 0; r < nrun; ++r) {
        D2G_CHECK(ctx, run_len[r] >= (uint32_t)k, "run shorter than k");
        const uint64_t nk = (uint64_t)run_len[r] - k + 1;
        p.chunk_off[r + 1] = p.chunk_off[r] + div_up<uint64_t>(nk, K1_CHUNK);
        p.nkmers += nk;
        p.nbases += run_len[r];
    }
    for (size_t g = 0; g < n; ++g) {
        const size_t r0 = genome_run_off[g], r1 = genome_run_off[g + 1];
        D2G_CHECK(ctx, r1 >= r0 && r1 <= nrun, "genome_run_off not monotone");
        const uint64_t cbeg = p.chunk_off[r0], cend = p.chunk_off[r1];
        size_t r = r0;
        for (uint64_t c = cbeg; c < cend; c += K1_BLOCK_CHUNKS) {
            const uint32_t nc = (uint32_t)std::min<uint64_t>(K1_BLOCK_CHUNKS, cend - c);
            while (p.chunk_off[r + 1] <= c) ++r;
            size_t rl = r;
            while (p.chunk_off[rl + 1] < c + nc) ++rl;
            p.bg.push_back((uint32_t)g); p.bc0.push_back(c); p.bn.push_back(nc);
            p.blo.push_back((uint32_t)r); p.bhi.push_back((uint32_t)rl + 1);
        }
    }
    D2G_CHECK(ctx, p.bg.size() < (1ull << 31), "oph plan: too many workgroups; sketch in smaller batches");
    return D2G_OK;
}

namespace {
int launch_k1(d2g_ctx *ctx, K1Args a, size_t nblk, size_t m, hipStream_t s) {
    const bool pow2 = (m & (m - 1)) == 0;
    const size_t lds = m * sizeof(uint64_t);
    const bool use_lds = lds <= 128 * 1024;
    auto kern = pow2 ? (use_lds ? k1_oph_kernel<true, true> : k1_oph_kernel<true, false>)
                     : (use_lds ? k1_oph_kernel<false, true> : k1_oph_kernel<false, false>);
    if (use_lds && lds > 48 * 1024)
        D2G_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    d2g_timer tm(ctx, &ctx->ev_k1, s);
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(K1_THREADS), use_lds ? lds : 0, s, a);
    tm.stop();
    D2G_HIP(ctx, hipGetLastError());
    return D2G_OK;
}

}  // namespace

extern "C" {

int d2g_oph_plan_create(d2g_ctx *ctx, const uint64_t *run_start, const uint32_t *run_len, size_t nrun,
                        const uint64_t *genome_run_off, size_t n, int k, d2g_oph_plan **out) {
    if (!ctx || !out) return D2G_ERR_INVALID;
    *out = nullptr;
    D2G_CHECK(ctx, nrun == 0 || run_start, "d2g_oph_plan_create: null table");
    PlanHost ph;
    if (int rc = d2g_build_plan_host(ctx, run_len, nrun, genome_run_off, n, k, ph)) return rc;
    D2G_HIP(ctx, hipSetDevice(ctx->device));
    d2g_oph_plan *p = new (std::nothrow) d2g_oph_plan();
    if (!p) return D2G_ERR_NOMEM;
    p->ctx = ctx; p->k = k; p->n = n; p->nrun = nrun;
    p->h_run_len.assign(run_len, run_len + nrun);
    p->h_genome_run_off.assign(genome_run_off, genome_run_off + n + 1);
    p->nkmers = ph.nkmers; p->nbases = ph.nbases; p->nblk = ph.bg.size();
    std::vector<uint64_t> rs(run_start, run_start + nrun);
    std::vector<uint32_t> rl(run_len, run_len + nrun);
    int rc;
    if ((rc = upload(ctx, rs, &p->d_run_start)) || (rc = upload(ctx, rl, &p->d_run_len)) ||
        (rc = upload(ctx, ph.chunk_off, &p->d_run_chunk_off)) || (rc = upload(ctx, ph.bg, &p->d_blk_genome)) ||
        (rc = upload(ctx, ph.bc0, &p->d_blk_chunk0)) || (rc = upload(ctx, ph.bn, &p->d_blk_nchunks)) ||
        (rc = upload(ctx, ph.blo, &p->d_blk_run_lo)) || (rc = upload(ctx, ph.bhi, &p->d_blk_run_hi))) {
        d2g_oph_plan_destroy(p);
        return rc;
    }
    *out = p;
    return D2G_OK;
}

int d2g_oph_sketch_dev(d2g_ctx *ctx, const d2g_oph_plan *plan, const uint8_t *packed_dev, int canon,
                       uint64_t xormask, size_t sketchsize, uint64_t *regs_out_dev, void *stream) {
    if (!ctx || !plan) return D2G_ERR_INVALID;
    D2G_CHECK(ctx, plan->ctx == ctx, "plan belongs to another context");
    D2G_CHECK(ctx, sketchsize >= 1 && sketchsize < (1ull << 31), "sketchsize out of range");
    D2G_CHECK(ctx, regs_out_dev != nullptr, "null regs_out");
    D2G_CHECK(ctx, ((uintptr_t)packed_dev & 3) == 0, "packed stream must be 4-byte aligned");
    D2G_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = as_stream(stream);
    const size_t m = d2g_oph_m(sketchsize);
    // registers_ initialise to T(-1): oph.h:147,233
    D2G_HIP(ctx, hipMemsetAsync(regs_out_dev, 0xFF, plan->n * m * sizeof(uint64_t), s));
    if (plan->nblk == 0) return D2G_OK;
    D2G_CHECK(ctx, packed_dev != nullptr, "null packed stream");
    K1Args a;
    a.km = d2g_plan_args(plan, packed_dev, canon);
    a.regs_out = regs_out_dev; a.xormask = xormask; a.ophxor = d2g_oph_xor_const();
    a.m = (uint32_t)m;
    return launch_k1(ctx, a, plan->nblk, m, s);
}

int d2g_oph_sketch(d2g_ctx *ctx, const uint8_t *packed, size_t packed_bytes, const uint64_t *run_start,
                   const uint32_t *run_len, size_t nrun, const uint64_t *genome_run_off, size_t n, int k,
                   int canon, uint64_t xormask, size_t sketchsize, uint64_t *regs_out) {
    if (!ctx) return D2G_ERR_INVALID;
    D2G_CHECK(ctx, regs_out != nullptr || n == 0, "null regs_out");
    d2g_oph_plan *plan = nullptr;
    int rc = d2g_oph_plan_create(ctx, run_start, run_len, nrun, genome_run_off, n, k, &plan);
    if (rc) return rc;
    // the kernel reads up to 20 bytes past a chunk's first word: require the documented pad
    if (nrun) {
        uint64_t maxend = 0;
        for (size_t r = 0; r < nrun; ++r) maxend = std::max<uint64_t>(maxend, run_start[r] + run_len[r]);
        if (packed_bytes < (maxend + 3) / 4 + 64) {
            d2g_oph_plan_destroy(plan);
            ctx->last_error = "packed stream lacks the 64-byte tail pad";
            return D2G_ERR_INVALID;
        }
    }
    const size_t m = d2g_oph_m(sketchsize);
    uint8_t *d_packed = nullptr;
    uint64_t *d_regs = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_packed); (void)hipFree(d_regs); d2g_oph_plan_destroy(plan); };
    hipError_t e;
    if ((e = hipMalloc((void **)&d_packed, std::max<size_t>(packed_bytes, 4))) != hipSuccess ||
        (e = hipMalloc((void **)&d_regs, std::max<size_t>(n * m, 1) * sizeof(uint64_t))) != hipSuccess) {
        ctx->last_error = hipGetErrorString(e); cleanup(); return D2G_ERR_NOMEM;
    }
    if (packed_bytes && (e = hipMemcpy(d_packed, packed, packed_bytes, hipMemcpyHostToDevice)) != hipSuccess) {
        ctx->last_error = hipGetErrorString(e); cleanup(); return D2G_ERR_HIP;
    }
    rc = d2g_oph_sketch_dev(ctx, plan, d_packed, canon, xormask, sketchsize, d_regs, nullptr);
    if (rc == D2G_OK && n) {
        e = hipMemcpy(regs_out, d_regs, n * m * sizeof(uint64_t), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); rc = D2G_ERR_HIP; }
    }
    cleanup();
    return rc;
}

void d2g_sketcher_destroy(d2g_sketcher *sk) {
    if (!sk) return;
    (void)hipSetDevice(sk->ctx->device);
    (void)hipFree(sk->d_packed); (void)hipFree(sk->d_regs); (void)hipFree(sk->d_arena);
    if (sk->h_arena) (void)hipHostFree(sk->h_arena);
    if (sk->h_stage) (void)hipHostFree(sk->h_stage);
    if (sk->k3) d2g_k3_state_destroy(sk->k3);
    if (sk->k0) d2g_k0_state_destroy(sk->k0);
    if (sk->stream) (void)hipStreamDestroy(sk->stream);
    delete sk;
}

int d2g_sketcher_create(d2g_ctx *ctx, d2g_sketcher **out) {
    if (!ctx || !out) return D2G_ERR_INVALID;
    *out = nullptr;
    D2G_HIP(ctx, hipSetDevice(ctx->device));
    d2g_sketcher *sk = new (std::nothrow) d2g_sketcher();
    if (!sk) return D2G_ERR_NOMEM;
    sk->ctx = ctx;
    hipError_t e = hipStreamCreateWithFlags(&sk->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); delete sk; return D2G_ERR_HIP; }
    *out = sk;
    return D2G_OK;
}

int d2g_sketcher_run(d2g_sketcher *sk, const uint8_t *packed, size_t packed_bytes, const uint64_t *run_start,
                     const uint32_t *run_len, size_t nrun, const uint64_t *genome_run_off, size_t n, int k, int canon,
                     uint64_t xormask, size_t sketchsize, uint64_t *regs_out) {
    if (!sk) return D2G_ERR_INVALID;
    d2g_ctx *ctx = sk->ctx;
    D2G_CHECK(ctx, sketchsize >= 1 && sketchsize < (1ull << 31), "sketchsize out of range");
    D2G_CHECK(ctx, regs_out != nullptr || n == 0, "null regs_out");
    K1Args a;
    size_t nblk = 0;
    if (int rc = d2g_sketcher_stage(sk, packed, packed_bytes, run_start, run_len, nrun, genome_run_off, n, k, canon,
                                    &a.km, &nblk, nullptr)) return rc;
    const size_t m = d2g_oph_m(sketchsize);
    if (int rc = d2g_grow(ctx, &sk->d_regs, &sk->cap_regs, std::max<size_t>(n * m, 1))) return rc;
    hipStream_t s = sk->stream;
    D2G_HIP(ctx, hipMemsetAsync(sk->d_regs, 0xFF, std::max<size_t>(n * m, 1) * sizeof(uint64_t), s));   // registers_ = T(-1): oph.h:147,233
    if (nblk) {
        a.regs_out = sk->d_regs; a.xormask = xormask; a.ophxor = d2g_oph_xor_const();
        a.m = (uint32_t)m;
        if (int rc = launch_k1(ctx, a, nblk, m, s)) return rc;
    }
    if (n) D2G_HIP(ctx, hipMemcpyAsync(regs_out, sk->d_regs, n * m * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    D2G_HIP(ctx, hipStreamSynchronize(s));
    return D2G_OK;
}

}  // extern "C"

int d2g_sketcher_stage(d2g_sketcher *sk, const uint8_t *packed, size_t packed_bytes, const uint64_t *run_start,
                       const uint32_t *run_len, size_t nrun, const uint64_t *genome_run_off, size_t n, int k, int canon,
                       KmerArgs *out, size_t *nblk_out, PlanHost *ph_out) {
    d2g_ctx *ctx = sk->ctx;
    uint64_t ingested_bases = 0;
    const bool use_ingested = packed == nullptr && d2g_k0_ingested(sk, &ingested_bases);
    D2G_CHECK(ctx, nrun == 0 || (run_start && (packed || use_ingested)), "null input");
    if (use_ingested) packed_bytes = (size_t)((ingested_bases + 3) / 4 + 64);        // the device stream's extent (zero-padded by the ingest)
    else d2g_k0_invalidate(sk);                                                       // the device buffer is about to be overwritten
    PlanHost ph_local;
    PlanHost &ph = ph_out ? *ph_out : ph_local;
    if (int rc = d2g_build_plan_host(ctx, run_len, nrun, genome_run_off, n, k, ph)) return rc;
    if (nrun) {
        uint64_t maxend = 0;
        for (size_t r = 0; r < nrun; ++r) maxend = std::max<uint64_t>(maxend, run_start[r] + run_len[r]);
        D2G_CHECK(ctx, packed_bytes >= (maxend + 3) / 4 + 64, "packed stream lacks the 64-byte tail pad");
    }
    D2G_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nblk = ph.bg.size();
    if (!use_ingested)
        if (int rc = d2g_grow(ctx, &sk->d_packed, &sk->cap_packed, std::max<size_t>(packed_bytes, 4))) return rc;
    // arena layout (256-byte aligned pieces)
    auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
    size_t off = 0;
    const size_t o_rs = off;  off = al(off + nrun * 8);
    const size_t o_co = off;  off = al(off + (nrun + 1) * 8);
    const size_t o_c0 = off;  off = al(off + nblk * 8);
    const size_t o_rl = off;  off = al(off + nrun * 4);
    const size_t o_bg = off;  off = al(off + nblk * 4);
    const size_t o_bn = off;  off = al(off + nblk * 4);
    const size_t o_lo = off;  off = al(off + nblk * 4);
    const size_t o_hi = off;  off = al(off + nblk * 4);
    if (off > sk->cap_arena) {
        (void)hipFree(sk->d_arena);
        if (sk->h_arena) (void)hipHostFree(sk->h_arena);
        sk->d_arena = sk->h_arena = nullptr; sk->cap_arena = 0;
        const size_t ncap = off + off / 4 + 65536;
        D2G_HIP(ctx, hipMalloc((void **)&sk->d_arena, ncap));
        D2G_HIP(ctx, hipHostMalloc((void **)&sk->h_arena, ncap, hipHostMallocDefault));
        sk->cap_arena = ncap;
    }
    uint8_t *h = sk->h_arena;
    if (nrun) { std::memcpy(h + o_rs, run_start, nrun * 8); std::memcpy(h + o_rl, run_len, nrun * 4); }
    std::memcpy(h + o_co, ph.chunk_off.data(), (nrun + 1) * 8);
    if (nblk) {
        std::memcpy(h + o_c0, ph.bc0.data(), nblk * 8); std::memcpy(h + o_bg, ph.bg.data(), nblk * 4);
        std::memcpy(h + o_bn, ph.bn.data(), nblk * 4);  std::memcpy(h + o_lo, ph.blo.data(), nblk * 4);
        std::memcpy(h + o_hi, ph.bhi.data(), nblk * 4);
    }
    hipStream_t s = sk->stream;
    D2G_HIP(ctx, hipMemcpyAsync(sk->d_arena, h, off, hipMemcpyHostToDevice, s));
    if (packed_bytes && !use_ingested) {
        // pageable source: stage through our own pinned buffer (the runtime would otherwise pin the
        // caller's pages on the fly, which contends with parser threads on the process' mm locks)
        if (packed_bytes > sk->cap_stage) {
            if (sk->h_stage) (void)hipHostFree(sk->h_stage);
            sk->h_stage = nullptr; sk->cap_stage = 0;
            const size_t ncap = packed_bytes + packed_bytes / 4 + 65536;
            D2G_HIP(ctx, hipHostMalloc((void **)&sk->h_stage, ncap, hipHostMallocDefault));
            sk->cap_stage = ncap;
        }
        std::memcpy(sk->h_stage, packed, packed_bytes);
        D2G_HIP(ctx, hipMemcpyAsync(sk->d_packed, sk->h_stage, packed_bytes, hipMemcpyHostToDevice, s));
    }
    out->packed = reinterpret_cast<const uint32_t *>(sk->d_packed);
    out->run_start = reinterpret_cast<const uint64_t *>(sk->d_arena + o_rs);
    out->run_len = reinterpret_cast<const uint32_t *>(sk->d_arena + o_rl);
    out->run_chunk_off = reinterpret_cast<const uint64_t *>(sk->d_arena + o_co);
    out->blk_genome = reinterpret_cast<const uint32_t *>(sk->d_arena + o_bg);
    out->blk_chunk0 = reinterpret_cast<const uint64_t *>(sk->d_arena + o_c0);
    out->blk_nchunks = reinterpret_cast<const uint32_t *>(sk->d_arena + o_bn);
    out->blk_run_lo = reinterpret_cast<const uint32_t *>(sk->d_arena + o_lo);
    out->blk_run_hi = reinterpret_cast<const uint32_t *>(sk->d_arena + o_hi);
    out->k = k; out->canon = canon; out->blk0 = 0;
    *nblk_out = nblk;
    return D2G_OK;
}

void d2g_warm_k1() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k1_oph_kernel<true, true>)); }

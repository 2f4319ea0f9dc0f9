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
#include "d2g_internal.h"
#include <algorithm>
#include <cstring>
#include <new>
#include <vector>

namespace {

constexpr int K1_THREADS = 256;
constexpr int K1_CHUNK = 64;          // k-mers per lane-chunk
constexpr int K1_CPT = 4;             // chunks per lane (16 measured 3% slower: fewer, longer workgroups)
constexpr int K1_BLOCK_CHUNKS = K1_THREADS * K1_CPT;

__device__ __forceinline__ uint64_t wang64(uint64_t k) {
    k = ~k + (k << 21);
    k ^= k >> 24;
    k = k + (k << 3) + (k << 8);
    k ^= k >> 14;
    k = k + (k << 2) + (k << 4);
    k ^= k >> 28;
    k += k << 31;
    return k;
}

struct K1Args {
    const uint32_t *packed;        // 16 bases per dword, base p at bits [2(p%16), +2)
    const uint64_t *run_start;
    const uint32_t *run_len;
    const uint64_t *run_chunk_off; // [nrun+1] exclusive prefix of chunks per run
    const uint32_t *blk_genome;
    const uint64_t *blk_chunk0;
    const uint32_t *blk_nchunks;
    const uint32_t *blk_run_lo;
    const uint32_t *blk_run_hi;
    uint64_t *regs_out;            // [n][m], pre-filled with ~0
    uint64_t xormask;
    uint64_t ophxor;
    uint32_t m;
    int k;
    int canon;
};

// funnel shift: low 32 bits of (hi:lo) >> sh, 0 <= sh < 32
__device__ __forceinline__ uint32_t fsr(uint32_t hi, uint32_t lo, uint32_t sh) {
    return __builtin_amdgcn_alignbit(hi, lo, sh);
}

template <bool POW2, bool USE_LDS>
__global__ __launch_bounds__(K1_THREADS) void k1_oph_kernel(K1Args a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t lreg[];
    const int tid = threadIdx.x;
    const uint32_t b = blockIdx.x;
    const uint32_t g = a.blk_genome[b];
    const uint64_t c0 = a.blk_chunk0[b];
    const uint32_t nc = a.blk_nchunks[b];
    const uint32_t rlo = a.blk_run_lo[b], rhi = a.blk_run_hi[b];
    const uint32_t m = a.m;
    uint64_t *gout = a.regs_out + (size_t)g * m;

    if (USE_LDS) {
        for (uint32_t i = tid; i < m; i += K1_THREADS) lreg[i] = ~0ull;
        __syncthreads();
    }
    const int k = a.k;
    const uint64_t kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
    const int rcshift = 2 * (k - 1);
    const bool canon = a.canon != 0;
    const uint64_t xormask = a.xormask, ophxor = a.ophxor;

    for (int it = 0; it < K1_CPT; ++it) {
        const uint32_t ci_blk = it * K1_THREADS + tid;
        if (ci_blk >= nc) break;
        const uint64_t c = c0 + ci_blk;
        // run containing chunk c (runs of this block only: usually a single candidate)
        uint32_t lo = rlo, hi = rhi;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (a.run_chunk_off[mid] <= c) lo = mid; else hi = mid;
        }
        const uint64_t ci = c - a.run_chunk_off[lo];
        const uint64_t nk = (uint64_t)a.run_len[lo] - k + 1;
        const uint64_t p = a.run_start[lo] + ci * K1_CHUNK;       // first k-mer start (base index)
        const uint64_t left = nk - ci * K1_CHUNK;
        const int n = left < (uint64_t)K1_CHUNK ? (int)left : K1_CHUNK;

        // warm-up window: bases [p, p+k-1) (<= 31 bases) as one 64-bit value
        uint64_t fwd = 0, rc = 0;
        {
            const uint32_t *w = a.packed + (p >> 4);
            const uint32_t sh = (uint32_t)(p & 15) * 2;
            const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
            uint64_t X = ((uint64_t)fsr(w2, w1, sh) << 32) | fsr(w1, w0, sh);
            for (int j = 0; j < k - 1; ++j) {
                const uint64_t cb = X & 3;
                X >>= 2;
                fwd = (fwd << 2) | cb;
                rc = (rc >> 2) | ((3 - cb) << rcshift);
            }
        }
        // main window: bases [q, q+64), q = p + k - 1, aligned into 4 dwords
        uint32_t M0, M1, M2, M3;
        {
            const uint64_t q = p + (uint64_t)(k - 1);
            const uint32_t *w = a.packed + (q >> 4);
            const uint32_t sh = (uint32_t)(q & 15) * 2;
            const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4];
            M0 = fsr(w1, w0, sh); M1 = fsr(w2, w1, sh); M2 = fsr(w3, w2, sh); M3 = fsr(w4, w3, sh);
        }
#pragma unroll 1
        for (int wi = 0; wi < 4; ++wi) {
            const uint32_t W = wi == 0 ? M0 : wi == 1 ? M1 : wi == 2 ? M2 : M3;
            const int ebase = wi * 16;
            if (ebase >= n) break;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint64_t cb = (W >> (2 * e)) & 3u;
                fwd = ((fwd << 2) | cb) & kmask;
                rc = (rc >> 2) | ((3 - cb) << rcshift);
                if (ebase + e < n) {
                    const uint64_t x = canon ? (fwd < rc ? fwd : rc) : fwd;
                    const uint64_t id = wang64(wang64(x ^ xormask) ^ ophxor);
                    // Schismatic<uint32_t>::mod(size_t): argument narrowed to 32 bits (oph.h:184)
                    const uint32_t idx = POW2 ? ((uint32_t)id & (m - 1)) : ((uint32_t)id % m);
                    if (USE_LDS) {
                        if (id < lreg[idx]) atomicMin((unsigned long long *)&lreg[idx], (unsigned long long)id);
                    } else {
                        if (id < gout[idx]) atomicMin((unsigned long long *)&gout[idx], (unsigned long long)id);
                    }
                }
            }
        }
    }
    if (USE_LDS) {
        __syncthreads();
        for (uint32_t i = tid; i < m; i += K1_THREADS) {
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

struct d2g_oph_plan {
    d2g_ctx *ctx = nullptr;
    int k = 0;
    size_t n = 0, nrun = 0, nblk = 0;
    uint64_t nkmers = 0, nbases = 0;
    uint64_t *d_run_start = nullptr;
    uint32_t *d_run_len = nullptr;
    uint64_t *d_run_chunk_off = nullptr;
    uint32_t *d_blk_genome = nullptr;
    uint64_t *d_blk_chunk0 = nullptr;
    uint32_t *d_blk_nchunks = nullptr;
    uint32_t *d_blk_run_lo = nullptr;
    uint32_t *d_blk_run_hi = nullptr;
};

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

namespace {
// host-side launch plan: 64-k-mer chunks per run, <= K1_BLOCK_CHUNKS chunks of one genome per workgroup
struct PlanHost {
    std::vector<uint64_t> chunk_off, bc0;
    std::vector<uint32_t> bg, bn, blo, bhi;
    uint64_t nkmers = 0, nbases = 0;
};
int build_plan_host(d2g_ctx *ctx, const uint32_t *run_len, size_t nrun, const uint64_t *genome_run_off, size_t n, int k, PlanHost &p) {
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

template <class T> int grow(d2g_ctx *ctx, T **p, size_t *cap, size_t need) {
    if (need <= *cap) return D2G_OK;
    (void)hipFree(*p); *p = nullptr; *cap = 0;
    const size_t ncap = need + need / 4 + 4096;
    D2G_HIP(ctx, hipMalloc((void **)p, ncap * sizeof(T)));
    *cap = ncap;
    return D2G_OK;
}
}  // namespace

// Host ingest feeds K1 in groups of inputs; re-allocating device buffers and uploading eight
// small tables per group costs ~20 ms, the kernel ~0.1 ms.  The sketcher keeps grow-only device
// buffers and ships all launch tables in ONE copy from a pinned arena.
struct d2g_sketcher {
    d2g_ctx *ctx = nullptr;
    hipStream_t stream = nullptr;
    uint8_t *d_packed = nullptr; size_t cap_packed = 0;
    uint64_t *d_regs = nullptr;  size_t cap_regs = 0;      // in u64
    uint8_t *d_arena = nullptr, *h_arena = nullptr; size_t cap_arena = 0;
    uint8_t *h_stage = nullptr; size_t cap_stage = 0;      // pinned staging of the packed stream
};

extern "C" {

int d2g_oph_plan_create(d2g_ctx *ctx, const uint64_t *run_start, const uint32_t *run_len, size_t nrun,
                        const uint64_t *genome_run_off, size_t n, int k, d2g_oph_plan **out) {
    if (!ctx || !out) return D2G_ERR_INVALID;
    *out = nullptr;
    D2G_CHECK(ctx, nrun == 0 || run_start, "d2g_oph_plan_create: null table");
    PlanHost ph;
    if (int rc = build_plan_host(ctx, run_len, nrun, genome_run_off, n, k, ph)) return rc;
    D2G_HIP(ctx, hipSetDevice(ctx->device));
    d2g_oph_plan *p = new (std::nothrow) d2g_oph_plan();
    if (!p) return D2G_ERR_NOMEM;
    p->ctx = ctx; p->k = k; p->n = n; p->nrun = nrun;
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
    a.packed = reinterpret_cast<const uint32_t *>(packed_dev);
    a.run_start = plan->d_run_start; a.run_len = plan->d_run_len; a.run_chunk_off = plan->d_run_chunk_off;
    a.blk_genome = plan->d_blk_genome; a.blk_chunk0 = plan->d_blk_chunk0; a.blk_nchunks = plan->d_blk_nchunks;
    a.blk_run_lo = plan->d_blk_run_lo; a.blk_run_hi = plan->d_blk_run_hi;
    a.regs_out = regs_out_dev; a.xormask = xormask; a.ophxor = d2g_oph_xor_const();
    a.m = (uint32_t)m; a.k = plan->k; a.canon = canon;
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
    D2G_CHECK(ctx, nrun == 0 || (run_start && packed), "null input");
    PlanHost ph;
    if (int rc = build_plan_host(ctx, run_len, nrun, genome_run_off, n, k, ph)) return rc;
    if (nrun) {
        uint64_t maxend = 0;
        for (size_t r = 0; r < nrun; ++r) maxend = std::max<uint64_t>(maxend, run_start[r] + run_len[r]);
        D2G_CHECK(ctx, packed_bytes >= (maxend + 3) / 4 + 64, "packed stream lacks the 64-byte tail pad");
    }
    D2G_HIP(ctx, hipSetDevice(ctx->device));
    const size_t m = d2g_oph_m(sketchsize), nblk = ph.bg.size();
    if (int rc = grow(ctx, &sk->d_packed, &sk->cap_packed, std::max<size_t>(packed_bytes, 4))) return rc;
    if (int rc = grow(ctx, &sk->d_regs, &sk->cap_regs, std::max<size_t>(n * m, 1))) return rc;
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
    if (packed_bytes) {
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
    D2G_HIP(ctx, hipMemsetAsync(sk->d_regs, 0xFF, std::max<size_t>(n * m, 1) * sizeof(uint64_t), s));   // registers_ = T(-1): oph.h:147,233
    if (nblk) {
        K1Args a;
        a.packed = reinterpret_cast<const uint32_t *>(sk->d_packed);
        a.run_start = reinterpret_cast<const uint64_t *>(sk->d_arena + o_rs);
        a.run_len = reinterpret_cast<const uint32_t *>(sk->d_arena + o_rl);
        a.run_chunk_off = reinterpret_cast<const uint64_t *>(sk->d_arena + o_co);
        a.blk_genome = reinterpret_cast<const uint32_t *>(sk->d_arena + o_bg);
        a.blk_chunk0 = reinterpret_cast<const uint64_t *>(sk->d_arena + o_c0);
        a.blk_nchunks = reinterpret_cast<const uint32_t *>(sk->d_arena + o_bn);
        a.blk_run_lo = reinterpret_cast<const uint32_t *>(sk->d_arena + o_lo);
        a.blk_run_hi = reinterpret_cast<const uint32_t *>(sk->d_arena + o_hi);
        a.regs_out = sk->d_regs; a.xormask = xormask; a.ophxor = d2g_oph_xor_const();
        a.m = (uint32_t)m; a.k = k; a.canon = canon;
        if (int rc = launch_k1(ctx, a, nblk, m, s)) return rc;
    }
    if (n) D2G_HIP(ctx, hipMemcpyAsync(regs_out, sk->d_regs, n * m * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    D2G_HIP(ctx, hipStreamSynchronize(s));
    return D2G_OK;
}

}  // extern "C"

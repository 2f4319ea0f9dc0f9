// d2g_kmers.h -- device-side k-mer enumeration shared by K1 (OPH) and K3 (k-mer counting):
// the reference's bns::Encoder::for_each over 2-bit packed runs (call site src/fastxsketch.cpp:416-417).
#pragma once
#include "d2g_internal.h"

constexpr int K1_THREADS = 256;
constexpr int K1_CHUNK = 64;          // k-mers per lane-chunk
constexpr int K1_CPT = 4;             // chunks per lane (16 measured 3% slower: fewer, longer workgroups)
constexpr int K1_BLOCK_CHUNKS = K1_THREADS * K1_CPT;

// Thomas Wang's 64-bit mix (sketch::hash::WangHash::hash; call sites src/enums.h:138, src/oph.h:49)
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

// launch plan of one batch of genomes: 64-k-mer chunks per run, <= K1_BLOCK_CHUNKS chunks of ONE
// genome per workgroup (built on the host by build_plan_host, d2g_k1.hip)
struct KmerArgs {
    const uint32_t *packed;        // 16 bases per dword, base p at bits [2(p%16), +2)
    const uint64_t *run_start;
    const uint32_t *run_len;
    const uint64_t *run_chunk_off; // [nrun+1] exclusive prefix of chunks per run
    const uint32_t *blk_genome;
    const uint64_t *blk_chunk0;
    const uint32_t *blk_nchunks;
    const uint32_t *blk_run_lo;
    const uint32_t *blk_run_hi;
    int k;
    int canon;
    uint32_t blk0;                 // first launch-plan block of this launch (a launch may cover a sub-range of the plan)
};

// funnel shift: low 32 bits of (hi:lo) >> sh, 0 <= sh < 32
__device__ __forceinline__ uint32_t fsr(uint32_t hi, uint32_t lo, uint32_t sh) {
    return __builtin_amdgcn_alignbit(hi, lo, sh);
}

// f(x) once per k-mer of the chunks this lane owns in workgroup blockIdx.x; x = canonical
// (min(fwd, revcomp)) or forward 2-bit k-mer.  A lane owns chunk (it * K1_THREADS + tid).
// IT0 <= it < IT1: the passes ("tiles" of K1_THREADS chunks = K1_THREADS * K1_CHUNK k-mers) of the workgroup to walk
template <class F>
__device__ __forceinline__ void d2g_for_each_kmer_its(const KmerArgs &a, int it0, int it1, F &&f) {
    const int tid = threadIdx.x;
    const uint32_t b = blockIdx.x + a.blk0;
    const uint64_t c0 = a.blk_chunk0[b];
    const uint32_t nc = a.blk_nchunks[b];
    const uint32_t rlo = a.blk_run_lo[b], rhi = a.blk_run_hi[b];
    const int k = a.k;
    const uint64_t kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
    const int rcshift = 2 * (k - 1);
    const bool canon = a.canon != 0;

    for (int it = it0; it < it1; ++it) {
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
                if (ebase + e < n) f(canon ? (fwd < rc ? fwd : rc) : fwd);
            }
        }
    }
}

template <class F>
__device__ __forceinline__ void d2g_for_each_kmer(const KmerArgs &a, F &&f) {
    d2g_for_each_kmer_its(a, 0, K1_CPT, static_cast<F &&>(f));
}

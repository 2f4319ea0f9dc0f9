// d2g_k3_bmh.hip -- K3: the --multiset sketch path on gfx950.
//
//   R11  exact k-mer counting      reference src/counter.h:68-77 (Counter::add(uint64_t)), finalize 118-138,
//                                  driver src/fastxsketch.cpp:425-445
//   R12  BagMinHash                sketch::BagMinHash2<double> (ABSENT dnbaker/sketch bmh.h; parity unpinned);
//                                  restated from Ertl, KDD 2018 under the "BMH-D2G" spec of DESIGN.md
//
// The reference counts with one robin-hood hash map per thread and feeds (kmer, count) to the
// sketch one at a time.  Here:
//   1. k3_hist / k3_scan / k3_scatter / k3_refine : every masked k-mer key = Wang(kmer ^ XORMASK) of a genome is
//      multi-split by its top bits into <= 4096 buckets of ~1-2 K keys (LDS-aggregated histogram, one global
//      reservation per (workgroup, write front)), in two levels so that a workgroup keeps <= 256 write fronts open
//      and the L2 completes the lines (see k3_scatter_kernel); k-mers are re-generated from the packed bases in
//      each enumerating pass instead of being stored (generation is ~100 VALU slots, a store+load is 16 B).
//   2. k3_bmh_main / k3_bmh_survivor / k3_bmh_verify : persistent workgroups walk the buckets; each bucket is counted
//      exactly in a 2048-slot LDS open-addressing table (ds_cmpst_b64 claim + ds_add), then every occupied
//      slot IS one (key, count) element: the first point of each of its top-level strips is tested against a bound
//      of the genome's final maximum register (99 % stop there), the survivors are queued -- in HBM, walked by the
//      survivor kernel one per lane (first pass), or in LDS and drained in place (repeat passes) -- and the
//      BagMinHash Poisson-process tree is walked for them depth-first with a private stack.
//      Registers live in HBM/L2 as the bit patterns of non-negative doubles and are lowered with
//      global_atomic_umin_x2 behind a read filter.
//      min is order-free and pruning by ANY bound >= the final maximum only drops points that cannot win,
//      so the result is bit-identical to the time-ordered sequential algorithm (oracle/d2_bmh_oracle.c).
//      The bound is GUESSED from the genome's total weight (max of m exponentials of rate W/m) and
//      VERIFIED afterwards (max(h) <= guess); a genome that fails is walked again under a 16x larger
//      guess (idempotent: registers only go down).
#include "d2g_k1.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#ifndef D2G_K3_EXP
#define D2G_K3_EXP 0          // timing experiments (tools/build_variant.sh): 1 = scatter without stores, 2 = 4-byte stores (no main pass); 3 = main without the BagMinHash walk, 4 = main counting only
#endif

namespace {

constexpr int K3_THREADS = 256;
constexpr int K3_MAXBBITS = 12;
constexpr int K3_MAXB = 1 << K3_MAXBBITS;   // buckets per genome (LDS histogram)
constexpr int K3_TAB = 2048;                // LDS count-table slots (24.6 KB with the counts: 6 workgroups per CU)
constexpr int K3_ROUND_KEYS = 1400;         // keys one table round is sized for (load <= 0.69)
constexpr int K3_TARGET = 1024;             // mean keys per bucket aimed for
constexpr int K3_L1BITS = 8;                // write fronts of the scatter = 2^K3_L1BITS; the other bucket bits are resolved by k3_refine_kernel
constexpr uint64_t K3_SPLIT_MIN = 4 * 1400; // mean bucket size above which a genome's buckets are split once more
constexpr uint64_t BMH_INF = 0x7FF0000000000000ull;
constexpr int BMH_STACK = 72;

struct K3Args {
    KmerArgs km;
    uint64_t xormask;
    const uint32_t *g_bbits;   // [n]   log2(#buckets) of genome g
    const uint32_t *g_boff;    // [n+1] first global bucket of genome g
    const uint64_t *g_koff;    // [n+1] first key of genome g in `keys` (host prefix of the k-mer counts)
    uint32_t *bucket_cnt;      // [TB]
    uint64_t *bucket_off;      // [TB+1] exclusive prefix of bucket_cnt
    uint64_t *cursor;          // [TB]   scatter cursors (copy of bucket_off)
    uint64_t *keys;            // [total k-mers] bucketed keys
    uint32_t TB;
    // two-level split (genomes with more than 2^l1bits buckets): the scatter writes by the top l1bits of the bucket index
    // only, into `coarse`; k3_refine_kernel then spreads every coarse bucket over its 2^(bb - l1bits) buckets in `keys`
    uint32_t l1bits;
    uint64_t *coarse;          // [total k-mers] or nullptr (no genome needs the second level)
    const uint32_t *l2_tb0;    // [nl2] first bucket of coarse bucket i (global bucket index)
    const uint32_t *l2_bits;   // [nl2] (bb << 8) | (bb - l1bits) of its genome
    uint32_t *blk_coarse;      // [nblk << l1bits] k-mers of launch-plan block b in each of its genome's coarse buckets
    uint32_t g0, n_genomes;    // k3_scan_kernel: first genome of this launch, genomes of the whole batch
};

__device__ __forceinline__ uint32_t bucket_of(uint64_t key, uint32_t bb) { return bb ? (uint32_t)(key >> (64 - bb)) : 0u; }

__global__ __launch_bounds__(K1_THREADS) void k3_hist_kernel(K3Args a) {
    __shared__ uint32_t hist[K3_MAXB];
    const int tid = threadIdx.x;
    const uint32_t blk = blockIdx.x + a.km.blk0;
    const uint32_t g = a.km.blk_genome[blk];
    const uint32_t bb = a.g_bbits[g], B = 1u << bb, boff = a.g_boff[g];
    for (uint32_t i = tid; i < B; i += K1_THREADS) hist[i] = 0;
    __syncthreads();
    const uint64_t xormask = a.xormask;
    d2g_for_each_kmer(a.km, [&](uint64_t x) {
        const uint64_t key = wang64(x ^ xormask);              // maskfn: src/enums.h:136-140
        atomicAdd(&hist[bucket_of(key, bb)], 1u);
    });
    __syncthreads();
    for (uint32_t i = tid; i < B; i += K1_THREADS)
        if (hist[i]) atomicAdd(&a.bucket_cnt[boff + i], hist[i]);
    // the workgroup's count per COARSE bucket (the scatter's write fronts): saves the scatter its counting enumeration
    const uint32_t b1 = bb < a.l1bits ? bb : a.l1bits, sb = bb - b1;
    uint32_t *mine = a.blk_coarse + ((size_t)blk << a.l1bits);
    for (uint32_t i = tid; i < (1u << b1); i += K1_THREADS) {
        uint32_t c = 0;
        for (uint32_t j = 0; j < (1u << sb); ++j) c += hist[(i << sb) + j];
        mine[i] = c;
    }
}

// exclusive prefix of bucket_cnt: one workgroup per genome scans its <= 4096 buckets (coalesced, 16 per
// lane); the genome's first key offset comes from the host, which knows every genome's k-mer count
__global__ __launch_bounds__(K3_THREADS) void k3_scan_kernel(K3Args a) {
    __shared__ uint32_t wsum[K3_THREADS / 64];
    const uint32_t tid = threadIdx.x, g = blockIdx.x + a.g0;
    const uint32_t b0 = a.g_boff[g], B = a.g_boff[g + 1] - b0;
    const uint32_t per = (B + K3_THREADS - 1) / K3_THREADS;             // <= 16
    const uint32_t lo = min(B, tid * per), hi = min(B, lo + per);
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += a.bucket_cnt[b0 + i];
    uint32_t incl = s;                                                  // inclusive scan across the wave, then the 4 waves
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if ((tid & 63) >= (uint32_t)o) incl += v; }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (uint32_t w = 0; w < (tid >> 6); ++w) wbase += wsum[w];
    uint64_t run = a.g_koff[g] + wbase + (incl - s);
    for (uint32_t i = lo; i < hi; ++i) {
        a.bucket_off[b0 + i] = run; a.cursor[b0 + i] = run;
        run += a.bucket_cnt[b0 + i];
    }
    // the end of this launch's last bucket = the first key of the next genome (the whole batch: bucket_off[TB]).  A later
    // launch over the next genomes writes the same value there; the counting pass of THIS range may already be reading it.
    if (blockIdx.x == gridDim.x - 1 && tid == K3_THREADS - 1) a.bucket_off[b0 + B] = a.g_koff[g + 1];
}

// Write-combining decides this pass.  A workgroup's 65 536 k-mers leave ~16 keys = one 128-byte line in each of 4096
// buckets, but every 8-byte store is a separate partial write and 4096 open lines per workgroup x 160 workgroups per XCD
// do not live in a 4 MB L2 until they are full: measured 38 GB of HBM writes for 10 GB of keys, 14.4 ms.  With <= 256
// write fronts per workgroup most lines are completed first (6.4 ms at 256 fronts, 5.1 ms at 64; the number of workgroups per
// CU makes no difference, 2 or 8: it is the fronts one workgroup keeps open that count).  So
// genomes with more than 2^l1bits buckets are split in two levels: here by the top l1bits of the bucket index -- the
// coarse bucket's region is the union of its buckets' regions, reserved through the cursor of its first bucket -- and
// then by the remaining bits, per coarse bucket, in k3_refine_kernel (<= 2^(12 - l1bits) fronts per workgroup there).
__global__ __launch_bounds__(K1_THREADS) void k3_scatter_kernel(K3Args a) {
    // ONE LDS array: first the workgroup's count per (coarse) bucket, then (after one 64-bit reservation
    // per bucket) the next write position relative to the genome's first key -- a genome holds < 2^32
    // k-mers -- so that a key's slot is a single LDS atomic.
    __shared__ uint32_t pos[K3_MAXB];
    const int tid = threadIdx.x;
    const uint32_t blk = blockIdx.x + a.km.blk0;
    const uint32_t g = a.km.blk_genome[blk];
    const uint32_t bb = a.g_bbits[g], boff = a.g_boff[g];
    const uint32_t b1 = bb < a.l1bits ? bb : a.l1bits, sb = bb - b1, B = 1u << b1;
    const uint64_t koff = a.g_koff[g];
    const uint64_t xormask = a.xormask;
    const uint32_t *mine = a.blk_coarse + ((size_t)blk << a.l1bits);              // counted by k3_hist_kernel
    for (uint32_t i = tid; i < B; i += K1_THREADS) {
        const uint32_t c = mine[i];
        pos[i] = c ? (uint32_t)(atomicAdd((unsigned long long *)&a.cursor[boff + (i << sb)], (unsigned long long)c) - koff) : 0u;
    }
    __syncthreads();
    uint64_t *keys = (sb ? a.coarse : a.keys) + koff;
    d2g_for_each_kmer(a.km, [&](uint64_t x) {
        const uint64_t key = wang64(x ^ xormask);
        const uint32_t slot = atomicAdd(&pos[bucket_of(key, b1)], 1u);
        if (D2G_K3_EXP == 1) { if (slot == 0xFFFFFFFFu) keys[slot] = key; }                      // timing experiment: no stores
        else if (D2G_K3_EXP == 2) reinterpret_cast<uint32_t *>(keys)[slot] = (uint32_t)key;      // timing experiment: 4-byte stores
        else keys[slot] = key;
    });
}

// second level: one workgroup per coarse bucket; the offsets of its 2^sb buckets are known from the histogram pass, so
// this is ONE read of the region and one LDS atomic per key
__global__ __launch_bounds__(K3_THREADS) void k3_refine_kernel(K3Args a) {
    __shared__ uint32_t pos[K3_MAXB];
    const uint32_t tid = threadIdx.x;
    const uint32_t tb0 = a.l2_tb0[blockIdx.x], bits = a.l2_bits[blockIdx.x];
    const uint32_t bb = bits >> 8, sb = bits & 255u, R = 1u << sb;
    const uint64_t o0 = a.bucket_off[tb0], nk = a.bucket_off[tb0 + R] - o0;
    for (uint32_t i = tid; i < R; i += K3_THREADS) pos[i] = (uint32_t)(a.bucket_off[tb0 + i] - o0);
    __syncthreads();
    const uint64_t *src = a.coarse + o0;
    uint64_t *dst = a.keys + o0;
    constexpr int PF = 8;                    // 4 and 16 measured: no difference (the pass moves 20 GB: bandwidth)
    for (uint64_t b0 = 0; b0 < nk; b0 += (uint64_t)PF * K3_THREADS) {
        uint64_t kk[PF];
#pragma unroll
        for (int j = 0; j < PF; ++j) { const uint64_t i = b0 + (uint64_t)j * K3_THREADS + tid; kk[j] = i < nk ? src[i] : 0; }
#pragma unroll
        for (int j = 0; j < PF; ++j)
            if (b0 + (uint64_t)j * K3_THREADS + tid < nk) dst[atomicAdd(&pos[bucket_of(kk[j], bb) & (R - 1)], 1u)] = kk[j];
    }
}

// ---------------------------------------------------------------------------------------------
// COMPACT path (k <= 21, D2G_K3_COMPACT=1): the multi-split stores 4 bytes per k-mer and writes them coalesced.
//
// Counting only needs a key that identifies the k-mer, so the split works on the 2k-bit k-mer x itself (the
// masked key Wang(x ^ XORMASK) is a bijection of it and is computed once per DISTINCT k-mer in the main pass).
// x = hi:lo with lo = 32 bits, hi = 2k - 32 <= 10 bits.  With m = lo * 0x9E3779B1 and t = the top bb bits of m,
//     bucket = t ^ (hi << (bb - hb))                           (bb >= hb bucket bits per genome)
// hi is recovered from (bucket, lo), so a bucket stores lo only; and within a bucket lo identifies the k-mer.
// Lower bits of m select sub-ranges and table rounds.
//
// The generic scatter issues one 8-byte store per k-mer to 4096 write fronts: 1.25e9 scattered stores take 11.5 ms
// whatever their width (measured: 13.6 ms with 4-byte stores, 2.9 ms with none) and reach HBM as 38 GB for 10 GB
// of keys.  Here a workgroup sorts each tile of 16 384 k-mers by bucket in LDS first (<= 1024 buckets: runs of
// ~16 keys = 64 B) and then flushes the tile with consecutive lanes writing consecutive words.  The per-tile
// bucket counts come from the histogram pass (u16 per tile and bucket), the scan turns them into per-tile write
// offsets: no atomics on global memory, two enumerations of the k-mers in total.
// ---------------------------------------------------------------------------------------------
constexpr int K3C_MAXBBITS = 10;
constexpr int K3C_MAXB = 1 << K3C_MAXBBITS;
constexpr int K3C_TILE = K1_THREADS * K1_CHUNK;       // k-mers of one pass of a workgroup
constexpr int K3C_TARGET = 4096;                      // mean k-mers per bucket aimed for (3-4 table rounds)
constexpr uint32_t K3C_MUL = 0x9E3779B1u;
static_assert(K3C_TILE <= 65535 + 1, "per-tile bucket counts are stored as u16 (a count of 65536 cannot occur: see k3c_hist)");

__device__ __forceinline__ uint32_t k3c_mix(uint32_t lo) { return lo * K3C_MUL; }
__device__ __forceinline__ uint32_t k3c_bucket(uint64_t x, uint32_t bb, uint32_t hb) {
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    const uint32_t t = bb ? k3c_mix(lo) >> (32 - bb) : 0u;
    return t ^ (hi << (bb - hb));
}
// the k-mer of stored word lo in bucket b (local index) of a genome with bb bucket bits
__device__ __forceinline__ uint64_t k3c_kmer(uint32_t lo, uint32_t b, uint32_t bb, uint32_t hb) {
    const uint32_t t = bb ? k3c_mix(lo) >> (32 - bb) : 0u;
    const uint32_t hi = hb ? (b ^ t) >> (bb - hb) : 0u;
    return ((uint64_t)hi << 32) | lo;
}

struct K3cArgs {
    KmerArgs km;
    const uint32_t *g_bbits;   // [n]
    const uint32_t *g_boff;    // [n+1] first global bucket of genome g
    const uint64_t *g_koff;    // [n+1] first k-mer slot of genome g
    const uint32_t *g_blk;     // [n+1] first workgroup (launch-plan block) of genome g
    uint16_t *tile_cnt;        // [ntiles][K3C_MAXB] k-mers of tile (block * K1_CPT + it) per bucket
    uint32_t *tile_off;        // [ntiles][K3C_MAXB] write offset of that tile in the bucket, relative to the genome's first slot
    uint32_t *bucket_cnt;      // [TB]
    uint64_t *bucket_off;      // [TB+1]
    uint32_t *keys32;          // [total k-mers]
    uint32_t hb;               // hi bits = max(0, 2k - 32)
    uint32_t TB;
};

__global__ __launch_bounds__(K1_THREADS) void k3c_hist_kernel(K3cArgs a) {
    __shared__ uint32_t cnt[K3C_MAXB];
    const int tid = threadIdx.x;
    const uint32_t g = a.km.blk_genome[blockIdx.x];
    const uint32_t bb = a.g_bbits[g], B = 1u << bb, hb = a.hb;
    for (int it = 0; it < K1_CPT; ++it) {
        for (uint32_t i = tid; i < B; i += K1_THREADS) cnt[i] = 0;
        __syncthreads();
        d2g_for_each_kmer_its(a.km, it, it + 1, [&](uint64_t x) { atomicAdd(&cnt[k3c_bucket(x, bb, hb)], 1u); });
        __syncthreads();
        // a tile holds at most 16384 k-mers: the counts fit u16
        uint16_t *dst = a.tile_cnt + ((size_t)blockIdx.x * K1_CPT + it) * K3C_MAXB;
        for (uint32_t i = tid; i < B; i += K1_THREADS) dst[i] = (uint16_t)cnt[i];
        __syncthreads();
    }
}

// one workgroup per genome: bucket totals over the genome's tiles -> exclusive prefix over buckets -> bucket offsets,
// then per tile and bucket the offset at which that tile writes
__global__ __launch_bounds__(K3_THREADS) void k3c_scan_kernel(K3cArgs a) {
    __shared__ uint32_t wsum[K3_THREADS / 64];
    constexpr int PER = K3C_MAXB / K3_THREADS;                       // 4 consecutive buckets per thread
    const uint32_t tid = threadIdx.x, g = blockIdx.x;
    const uint32_t bb = a.g_bbits[g], B = 1u << bb, b0 = a.g_boff[g];
    const size_t t0 = (size_t)a.g_blk[g] * K1_CPT, t1 = (size_t)a.g_blk[g + 1] * K1_CPT;
    uint32_t tot[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) tot[j] = 0;
    for (size_t t = t0; t < t1; ++t) {
        const uint16_t *c = a.tile_cnt + t * K3C_MAXB + tid * PER;
#pragma unroll
        for (int j = 0; j < PER; ++j) if (tid * PER + j < B) tot[j] += c[j];
    }
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) s += tot[j];
    uint32_t incl = s;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if ((tid & 63) >= (uint32_t)o) incl += v; }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    uint32_t run = incl - s;
    for (uint32_t w = 0; w < (tid >> 6); ++w) run += wsum[w];
    uint32_t base[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        base[j] = run;
        if (tid * PER + j < B) { a.bucket_off[b0 + tid * PER + j] = a.g_koff[g] + run; a.bucket_cnt[b0 + tid * PER + j] = tot[j]; }
        run += tot[j];
    }
    if (g == gridDim.x - 1 && tid == K3_THREADS - 1) a.bucket_off[a.TB] = a.g_koff[g + 1];
    for (size_t t = t0; t < t1; ++t) {
        const uint16_t *c = a.tile_cnt + t * K3C_MAXB + tid * PER;
        uint32_t *o = a.tile_off + t * K3C_MAXB + tid * PER;
#pragma unroll
        for (int j = 0; j < PER; ++j) if (tid * PER + j < B) { o[j] = base[j]; base[j] += c[j]; }
    }
}

// LDS: the tile's 16384 stored words sorted by bucket, the tile-local first slot of each bucket, and a cursor
struct K3cScatterLds {
    uint32_t stage[K3C_TILE];
    uint32_t lbase[K3C_MAXB + 1];
    uint32_t lcur[K3C_MAXB];
    uint32_t wsum[K1_THREADS / 64];
};

__global__ __launch_bounds__(K1_THREADS) void k3c_scatter_kernel(K3cArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char k3c_lds_raw[];
    K3cScatterLds &L = *reinterpret_cast<K3cScatterLds *>(k3c_lds_raw);
    constexpr int PER = K3C_MAXB / K1_THREADS;
    const uint32_t tid = threadIdx.x;
    const uint32_t g = a.km.blk_genome[blockIdx.x];
    const uint32_t bb = a.g_bbits[g], B = 1u << bb, hb = a.hb;
    uint32_t *out = a.keys32 + a.g_koff[g];
    for (int it = 0; it < K1_CPT; ++it) {
        const size_t tile = (size_t)blockIdx.x * K1_CPT + it;
        // tile-local exclusive prefix of the bucket counts
        const uint16_t *c = a.tile_cnt + tile * K3C_MAXB + tid * PER;
        uint32_t cnt[PER], s = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) { cnt[j] = (tid * PER + j < B) ? c[j] : 0u; s += cnt[j]; }
        uint32_t incl = s;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if ((tid & 63) >= (uint32_t)o) incl += v; }
        if ((tid & 63) == 63) L.wsum[tid >> 6] = incl;
        __syncthreads();
        uint32_t run = incl - s, total = 0;
        for (uint32_t w = 0; w < K1_THREADS / 64; ++w) { if (w < (tid >> 6)) run += L.wsum[w]; total += L.wsum[w]; }
#pragma unroll
        for (int j = 0; j < PER; ++j) { L.lbase[tid * PER + j] = run; L.lcur[tid * PER + j] = run; run += cnt[j]; }
        if (tid == K1_THREADS - 1) L.lbase[K3C_MAXB] = run;
        __syncthreads();
        if (total == 0) continue;                                    // uniform: every thread computed the same total
        d2g_for_each_kmer_its(a.km, it, it + 1, [&](uint64_t x) {
            L.stage[atomicAdd(&L.lcur[k3c_bucket(x, bb, hb)], 1u)] = (uint32_t)x;
        });
        __syncthreads();
        // flush: runs average 16 words, so a quarter wave (16 lanes) copies one bucket's run at a time -- a wave
        // store covers four runs of ~64 B; no per-element search for the bucket
        if (D2G_K3_EXP == 6) { __syncthreads(); continue; }          // timing experiment: no flush
        const uint32_t *toff = a.tile_off + tile * K3C_MAXB;
        const uint32_t q = tid >> 4, l16 = tid & 15;                 // 16 quarter-waves per workgroup
        for (uint32_t b = q; b < B; b += K1_THREADS / 16) {
            const uint32_t r0 = L.lbase[b], r1 = L.lbase[b + 1];
            if (r0 == r1) continue;
            uint32_t *dst = out + toff[b];
            for (uint32_t i = r0 + l16; i < r1; i += 16) dst[i - r0] = L.stage[i];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// exact counting of one bucket round into the LDS table
// ---------------------------------------------------------------------------------------------
// key type of the main-side code: the 64-bit masked key (generic path) or the 32-bit stored word (compact path)
template <bool C32> struct K3Key;
template <> struct K3Key<false> {
    typedef uint64_t T;
    static constexpr uint64_t EMPTY = ~0ull;
    // the key IS a hash (Wang of the masked k-mer): bits 32..42 are as good as any product of it, and are not the ones that
    // chose the bucket (top 12), the sub-range or the round (low bits)
    static __device__ __forceinline__ uint32_t slot(uint64_t key) { return (uint32_t)(key >> 32) & (K3_TAB - 1); }
    // R rounds: key bits [shift, shift + log2 R) (the bits below `shift` chose the sub-range)
    static __device__ __forceinline__ uint32_t round_of(uint64_t key, uint32_t R, uint32_t shift, uint32_t) { return (uint32_t)(key >> shift) & (R - 1); }
    static __device__ __forceinline__ uint32_t sub_of(uint64_t key, uint32_t R, uint32_t) { return (uint32_t)key & (R - 1); }
};
template <> struct K3Key<true> {
    typedef uint32_t T;
    static constexpr uint32_t EMPTY = ~0u;
    static __device__ __forceinline__ uint32_t slot(uint32_t lo) { return (lo * 0x85EBCA6Bu) >> 21; }
    // bits of m = lo * K3C_MUL from the top: bb bucket bits, then `shift` sub-range bits, then log2 R round bits
    static __device__ __forceinline__ uint32_t round_of(uint32_t lo, uint32_t R, uint32_t shift, uint32_t bb) {
        const uint32_t v = (k3c_mix(lo) << bb) << shift;            // bb + shift + log2 R <= 32 (checked on the host)
        return R > 1 ? v >> (__clz(R) + 1) : 0u;                     // 32 - log2 R = clz(R) + 1
    }
    static __device__ __forceinline__ uint32_t sub_of(uint32_t lo, uint32_t R, uint32_t bb) {
        return R > 1 ? (k3c_mix(lo) << bb) >> (__clz(R) + 1) : 0u;
    }
};
static_assert(K3_TAB == 1 << 11, "the slot hashes take the top 11 bits");

template <bool C32>
struct CountTab {
    typename K3Key<C32>::T *key;   // [K3_TAB]
    uint32_t *cnt;                 // [K3_TAB]
    uint32_t *ones;                // count of the key == EMPTY (cannot live in the table)
};

// keys of round r of R (R a power of two) of one key range; returns false on overflow
template <bool C32>
__device__ bool count_round(const CountTab<C32> &t, const typename K3Key<C32>::T *kb, uint64_t n, uint32_t R, uint32_t r, uint32_t shift,
                            uint32_t bb) {
    typedef typename K3Key<C32>::T KT;
    constexpr KT EMPTY = K3Key<C32>::EMPTY;
    const int tid = threadIdx.x;
    for (int s = tid; s < K3_TAB; s += K3_THREADS) { t.key[s] = EMPTY; t.cnt[s] = 0; }
    if (tid == 0) *t.ones = 0;
    __syncthreads();
    bool ok = true;
    // keys are fetched K3_KPF per lane at a time BEFORE the probe chains: with the load inside the
    // probing loop every key exposed a full HBM/L2 round trip (measured 25 us per 1220-key bucket)
    constexpr int K3_KPF = C32 ? 8 : 6;
    for (uint64_t base = 0; base < n && ok; base += (uint64_t)K3_KPF * K3_THREADS) {
        KT kreg[K3_KPF];
#pragma unroll
        for (int j = 0; j < K3_KPF; ++j) {
            const uint64_t i = base + (uint64_t)j * K3_THREADS + tid;
            kreg[j] = i < n ? kb[i] : 0;
        }
#pragma unroll
        for (int j = 0; j < K3_KPF; ++j) {
            const uint64_t i = base + (uint64_t)j * K3_THREADS + tid;
            const KT key = kreg[j];
            const bool mine = i < n && (R == 1 || K3Key<C32>::round_of(key, R, shift, bb) == r);
            if (mine && key == EMPTY) atomicAdd(t.ones, 1u);
            if (!mine || key == EMPTY) continue;
            if (D2G_K3_EXP == 8) { if (key == 12345) atomicAdd(t.ones, 1u); continue; }   // timing experiment: loads, no inserts
            // one exit test per probe (structured-control-flow bookkeeping is SALU work: the first
            // version of this loop issued 30 scalar instructions per probe)
            uint32_t s = K3Key<C32>::slot(key);
            int probes = 0;
            bool placed;
            for (;;) {
                KT cur = t.key[s];
                if (cur == EMPTY) {
                    if constexpr (C32) cur = atomicCAS(&t.key[s], EMPTY, key) == EMPTY ? key : t.key[s];
                    else cur = atomicCAS((unsigned long long *)&t.key[s], (unsigned long long)EMPTY, (unsigned long long)key) == EMPTY ? key : t.key[s];
                }
                placed = cur == key;
                if (placed | (++probes >= K3_TAB)) break;
                s = (s + 1) & (K3_TAB - 1);
            }
            if (placed) atomicAdd(&t.cnt[s], 1u); else ok = false;
        }
    }
    const bool res = !__syncthreads_or(!ok);
    return res;
}

// The main kernel's round: the table arrives CLEAN (the walk that follows each round empties the slots it reads), so a
// round is insert -> barrier -> walk+clear -> barrier.  Probing goes compare-and-swap first: one LDS operation and two
// compares per probe (the read-first loop of count_round spends twice the vector instructions; the kernel is bound by
// instruction issue, not by the LDS atomic rate: profiles/r02_k3_ablation.txt).
// GUARD = false (single-round ranges: n <= round_keys < K3_TAB keys, the table cannot fill): the probe loop carries no
// probe counter -- its bookkeeping was 13 scalar instructions per probe next to 6 vector ones.
template <bool C32, bool GUARD>
__device__ bool insert_round(const CountTab<C32> &t, const typename K3Key<C32>::T *kb, uint64_t n, uint32_t R, uint32_t r, uint32_t shift,
                             uint32_t bb) {
    typedef typename K3Key<C32>::T KT;
    constexpr KT EMPTY = K3Key<C32>::EMPTY;
    const int tid = threadIdx.x;
    bool ok = true;
    constexpr int K3_KPF = 6;
    for (uint64_t base = 0; base < n && ok; base += (uint64_t)K3_KPF * K3_THREADS) {
        KT kreg[K3_KPF];
#pragma unroll
        for (int j = 0; j < K3_KPF; ++j) {
            const uint64_t i = base + (uint64_t)j * K3_THREADS + tid;
            kreg[j] = i < n ? kb[i] : 0;
        }
        uint32_t n_ones = 0;                                              // the all-ones key cannot live in the table (rare: one branch per batch)
#pragma unroll
        for (int j = 0; j < K3_KPF; ++j) {
            const uint64_t i = base + (uint64_t)j * K3_THREADS + tid;
            const KT key = kreg[j];
            const bool mine = i < n && (R == 1 || K3Key<C32>::round_of(key, R, shift, bb) == r);
            n_ones += mine & (key == EMPTY);
            if (!mine || key == EMPTY) continue;
            if (D2G_K3_EXP == 8) { if (key == 12345) atomicAdd(t.ones, 1u); continue; }   // timing experiment: loads, no inserts
            uint32_t s = K3Key<C32>::slot(key);
            if constexpr (GUARD) {
                int probes = 0;
                bool placed;
                for (;;) {
                    KT old;
                    if constexpr (C32) old = atomicCAS(&t.key[s], EMPTY, key);
                    else old = (KT)atomicCAS((unsigned long long *)&t.key[s], (unsigned long long)EMPTY, (unsigned long long)key);
                    placed = (old == EMPTY) | (old == key);
                    if (placed | (++probes >= K3_TAB)) break;
                    s = (s + 1) & (K3_TAB - 1);
                }
                if (placed) atomicAdd(&t.cnt[s], 1u); else ok = false;
            } else {
                for (;;) {
                    KT old;
                    if constexpr (C32) old = atomicCAS(&t.key[s], EMPTY, key);
                    else old = (KT)atomicCAS((unsigned long long *)&t.key[s], (unsigned long long)EMPTY, (unsigned long long)key);
                    if ((old == EMPTY) | (old == key)) break;
                    s = (s + 1) & (K3_TAB - 1);
                }
                atomicAdd(&t.cnt[s], 1u);
            }
        }
        if (n_ones) atomicAdd(t.ones, n_ones);
    }
    if constexpr (GUARD) return !__syncthreads_or(!ok);
    __syncthreads();
    return true;
}

// ---------------------------------------------------------------------------------------------
// BMH-D2G process machinery (spec: DESIGN.md; sequential twin: oracle/d2_bmh_oracle.c)
// ---------------------------------------------------------------------------------------------
struct Proc {
    uint64_t p, q;       // weight-level range [V(p), V(q)), double bit patterns
    double x;            // time of the current point
    uint64_t rng;        // wyhash64_stateless state (in-tree twin: reference src/ssi.h:26-36)
    uint32_t i;          // register of the current point
    uint32_t pad;
};

__device__ __forceinline__ double V(uint64_t l) { return __longlong_as_double((long long)l); }
__device__ __forceinline__ uint64_t dbits(double d) { return (uint64_t)__double_as_longlong(d); }

__device__ __forceinline__ uint64_t wy_next(uint64_t &s) {
    s += 0x60bee2bee120fc15ull;
    const uint64_t a = s ^ 0xe7037ed1a0b428dbull;
    return (a * s) ^ __umul64hi(a, s);
}

// natural log on [2^-53, 1] in plain IEEE double ops (no FMA contraction: -ffp-contract=off), the
// same operation sequence as d2o_dlog
__device__ __forceinline__ double dlog(double u) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    const uint64_t b = dbits(u);
    int e = (int)(b >> 52) - 1023;
    double m = V((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double R = z * (Lg1 + z * (Lg2 + z * (Lg3 + z * (Lg4 + z * (Lg5 + z * (Lg6 + z * Lg7))))));
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

// advance to the process's next point; false = certainly later than `bound` (drop the process).
// Early-out: -log(u) >= 1 - u, so (1-u)/width > bound already proves x > bound without the log
// and the division (the 1e-9 margin dwarfs every rounding involved).
__device__ __forceinline__ bool proc_next(Proc &P, uint32_t m, double bound) {
    const double width = V(P.q) - V(P.p);
    const uint64_t r1 = wy_next(P.rng);
    const double uu = (double)((r1 >> 11) + 1) * 0x1p-53;            // (0, 1]
    if ((1.0 - uu) > bound * width * 1.000000001) { return false; }
        const double E = -dlog(uu);
    P.x = P.x + E / width;
    const uint64_t r2 = wy_next(P.rng);
    P.i = (uint32_t)__umul64hi(r2, (uint64_t)m);
    return P.x <= bound;
}

__device__ __forceinline__ void reg_min(uint64_t *h, uint32_t i, double x) {
    const uint64_t xb = dbits(x);
    if (xb < h[i]) atomicMin((unsigned long long *)&h[i], (unsigned long long)xb);
}
// second pass over FINAL registers: which element set each register (BagMinHash2::ids(), reference
// src/wsketch.cpp:66-67).  Ties (two elements with the same point) go to the smaller position.
struct ArgSink {
    const uint64_t *h;   // final registers of the set
    uint64_t *arg;       // [m] position of the element that owns the register, pre-set to ~0
    uint64_t pos;        // position of the element being walked
};
__device__ __forceinline__ void reg_min(const ArgSink &s, uint32_t i, double x) {
    if (dbits(x) == s.h[i]) atomicMin((unsigned long long *)&s.arg[i], (unsigned long long)s.pos);
}

// locate P's current point: narrow P to the half that holds it, level by level; the other half
// becomes a fresh process starting at P.x.  Pushes what may still matter.
// The expensive half of proc_next (log + division) is NOT done inside the level loop: a sibling
// that survives the cheap early-out is parked on the stack with its uniform in .x, and all parked
// processes are finished after the loop, where the lanes of a wave are converged again (inside
// the loop each lane would hit the expensive branch at a different level and the wave would pay
// for it once per level).  Same arithmetic in a different order: identical results.
template <class H>
__device__ void bmh_locate(Proc P, uint64_t d, double w, uint32_t m, double bound, H h, Proc *stk, int &sp, int *status) {
    const int sp0 = sp;
    bool counted = false, relevant = true;
    auto park = [&](Proc &S) {
        const double width = V(S.q) - V(S.p);
        const uint64_t r1 = wy_next(S.rng);
        const double uu = (double)((r1 >> 11) + 1) * 0x1p-53;        // (0, 1]
        if ((1.0 - uu) > bound * width * 1.000000001) return;   // see proc_next
        S.x = uu;
        if (sp >= BMH_STACK) { atomicExch(status, 3); return; }        // cannot happen (<= one sibling per tree level); never silent
        stk[sp++] = S;
    };
    for (;;) {
        if (!counted && V(P.q) <= w) { reg_min(h, P.i, P.x); counted = true; }
        if (P.q - P.p <= 1) break;
                const uint64_t r = P.p + ((P.q - P.p) >> 1);
        const uint64_t rb = wy_next(P.rng);
        const double ub = (double)(rb >> 11) * 0x1p-53;              // [0, 1)
        const double vp = V(P.p), vq = V(P.q), vr = V(r);
        const bool left = ub * (vq - vp) < (vr - vp);
        Proc S;
        S.x = 0.; S.i = 0; S.pad = 0;
        S.rng = d ^ (r * 0x9E3779B97F4A7C15ull) ^ 0xD6E8FEB86659FD93ull;
        if (left) { S.p = r; S.q = P.q; P.q = r; }
        else      { S.p = P.p; S.q = r; P.p = r; }
        if (V(S.p) < w) park(S);
        if (!(V(P.p) < w)) { relevant = false; break; }
    }
    if (relevant) park(P);                                           // single-level strip: its next point
    int out = sp0;
    for (int j = sp0; j < sp; ++j) {
        Proc S = stk[j];
                const double E = -dlog(S.x);
        S.x = P.x + E / (V(S.q) - V(S.p));
        const uint64_t r2 = wy_next(S.rng);
        S.i = (uint32_t)__umul64hi(r2, (uint64_t)m);
        if (S.x <= bound) { stk[out++] = S; }
    }
    sp = out;
}

// After count_round: squeeze the occupied slots that pass the count threshold to the front of the
// table arrays (the all-ones key, which cannot live in the table, is appended), so that the walk
// below runs over a dense element list with every lane busy.  Returns the number of elements.
template <bool C32>
__device__ uint32_t compact_elements(const CountTab<C32> &t, uint32_t *nelem, double thr) {
    constexpr int PER = K3_TAB / K3_THREADS;
    const int tid = threadIdx.x;
    typename K3Key<C32>::T k[PER];
    uint32_t c[PER];
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int s = tid + j * K3_THREADS;
        const uint32_t cc = t.cnt[s];
        k[j] = t.key[s];
        c[j] = (cc && (double)cc > thr) ? cc : 0u;                   // counter.h:123: pair.second > threshold
        mine += c[j] != 0;
    }
    const uint32_t ones = *t.ones;
    const bool extra = tid == 0 && ones && (double)ones > thr;
    mine += extra;
    if (tid == 0) *nelem = 0;
    __syncthreads();
    uint32_t pos = mine ? atomicAdd(nelem, mine) : 0u;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; ++j)
        if (c[j]) { t.key[pos] = k[j]; t.cnt[pos] = c[j]; ++pos; }
    if (extra) { t.key[pos] = K3Key<C32>::EMPTY; t.cnt[pos] = ones; }
    __syncthreads();
    return *nelem;
}

// BMH-D2G top level: 65 fixed strips of [0, 2^53) -- 16 unit strips, then octaves -- each an
// independent Poisson process from time 0 (see oracle/d2_bmh_oracle.c).  A k-mer seen once costs
// one seed, one generator step and one compare.
constexpr int BMH_NTOP = 65;
__device__ __forceinline__ double top_edge(int t) {
    return t <= 16 ? (double)t : V((uint64_t)(1023 + t - 12) << 52);           // 2^(t-12)
}
// number of strips whose lower edge is below w (0 < w <= 2^53): strips 0..top_count(w)-1 are the relevant ones
__device__ __forceinline__ int top_count(double w) {
    if (w <= 16.0) { const int c = (int)w; return c + ((double)c < w); }              // ceil(w)
    const uint64_t b = dbits(w);
    const int e = (int)(b >> 52) - 1023;                                                // floor(log2 w) >= 4
    const int c = 12 + e + ((b & 0x000FFFFFFFFFFFFFull) != 0);                          // edges 2^(t-12) < w  <=>  t < 12 + log2 w
    return c < BMH_NTOP ? c : BMH_NTOP;
}
__device__ __forceinline__ Proc top_proc(uint64_t d, int t) {
    Proc P;
    P.p = dbits(top_edge(t)); P.q = dbits(top_edge(t + 1)); P.x = 0.; P.i = 0; P.pad = 0;
    P.rng = d ^ ((uint64_t)(t + 1) * 0xA0761D6478BD642Full) ^ 0x8EBC6AF09C88C6E3ull;
    return P;
}

// everything below a process whose current point is at or before `bound`
template <class H>
__device__ __forceinline__ void walk_process(const Proc &P0, uint64_t d, double w, uint32_t m, double bound, H h, Proc *stk,
                                             int *status) {
    int sp = 0;
    bmh_locate(P0, d, w, m, bound, h, stk, sp, status);
    while (sp) {
        const Proc Q = stk[--sp];
        if (Q.x <= bound) bmh_locate(Q, d, w, m, bound, h, stk, sp, status);
    }
}

// walk every process of element (d, w) that can still matter under `bound`
template <class H>
__device__ __forceinline__ void walk_element(uint64_t d, double w, uint32_t m, double bound, H h, Proc *stk, int *status) {
    const int nt = top_count(w);
    for (int t = 0; t < nt; ++t) {
        Proc P = top_proc(d, t);
        if (proc_next(P, m, bound)) walk_process(P, d, w, m, bound, h, stk, status);
    }
}

// The main pass splits the walk in two.  Phase 1 (every element, every lane): the first point of
// each relevant top strip; 99% die in the early-out of proc_next.  A survivor needs the ~60-level
// descent of bmh_locate, which one lane would run alone while 63 wait -- so survivors are queued in
// LDS and phase 2 drains the queue with one survivor per lane once enough have accumulated.
struct QEntry { uint64_t d; double w; uint32_t t, g; };     // strip t of element (d, w) of genome g: the survivor's
                                                             // process is regenerated in phase 2 (24 B instead of 56)
constexpr int K3_QCAP = 256;
constexpr int K3_QDRAIN = 160;          // drain when at least this many are waiting

__device__ uint64_t block_hmax(const uint64_t *h, uint32_t m, uint64_t *red) {
    const int tid = threadIdx.x;
    uint64_t v = 0;
    for (uint32_t i = tid; i < m; i += K3_THREADS) { const uint64_t x = h[i]; v = x > v ? x : v; }
    for (int o = 32; o; o >>= 1) { const uint64_t x = __shfl_xor(v, o); v = x > v ? x : v; }
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    v = red[0];
    for (int k = 1; k < K3_THREADS / 64; ++k) v = red[k] > v ? red[k] : v;
    return v;
}

__device__ double block_sum(double v, double *red) {
    const int tid = threadIdx.x;
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double s = red[0];
    for (int k = 1; k < K3_THREADS / 64; ++k) s += red[k];
    return s;
}

struct BmhArgs {
    const uint64_t *keys;        // generic path: bucketed 64-bit masked keys
    const uint32_t *keys32;      // compact path: bucketed 32-bit stored words (k3c_*)
    const uint32_t *g_bbits;     // compact path: [n] bucket bits of genome g
    uint32_t hb;                 // compact path: hi bits = max(0, 2k - 32)
    uint64_t xormask;            // compact path: the masked key Wang(x ^ xormask) is formed per distinct k-mer here
    const uint64_t *bucket_off;
    const uint32_t *g_boff;
    uint32_t n;              // genomes
    uint32_t TB;
    uint32_t m;
    double thr;
    uint64_t *h;             // [n][m] register bit patterns, +inf initially
    uint64_t *guess;         // [n] pruning bound of the current pass (bit pattern of a double)
    double *tw;              // [n] total weight
    uint64_t *tw_acc;        // [n] sum of the counted weights of the first pass (integers; zeroed by the host)
    uint32_t *redo;          // [n] 1 = the guess proved too small: walk this genome again
    uint32_t *nredo;         // [1]
    int redo_mode;           // 0 = first pass (every genome, weights are summed); 1 = only genomes with redo[g]
    uint32_t round_keys;     // K3_ROUND_KEYS; D2G_K3_ROUND_KEYS lowers it (tests force multi-round buckets on small inputs)
    int *status;
    // big inputs: buckets of a genome with g_split[g] = s > 0 were split once more into 2^s sub-ranges by
    // their low key bits (k3_split_kernel): sub-range j of bucket tb = skeys[sub_off[i] .. sub_off[i+1]),
    // i = g_sub[g] + ((tb - g_boff[g]) << s) + j
    const uint32_t *g_split;  // [n] or nullptr
    const uint64_t *g_sub;    // [n+1]
    uint64_t *sub_off;        // [nsub + 1]
    uint64_t *skeys;          // [total k-mers]
    uint32_t *skeys32;        // compact path
    uint32_t tb0, tb1;        // buckets this launch of the main kernel walks (a sub-batch, or [0, TB))
    uint32_t g0;              // first genome of this launch of the verify kernel
    // first pass, light form: survivors queue in HBM, one region per main workgroup
    QEntry *gq; const uint64_t *gq_off; uint32_t *gq_n;
    // optional R11 output (k3_count_kernel): distinct (key,count) written in place of the bucket
    uint64_t *out_keys; uint32_t *out_counts; uint32_t *bucket_nd;
};

template <bool C32>
struct SharedK3 {
    typename K3Key<C32>::T key[K3_TAB + 1 + C32];      // +1: the all-ones key joins the compacted element list (+1: alignment)
    uint32_t cnt[K3_TAB + 2];
    uint64_t red[8];
    uint32_t ones;
    uint32_t misc;
    uint32_t nelem;
    uint32_t pad;
};

__device__ __forceinline__ uint32_t genome_of_bucket(const uint32_t *g_boff, uint32_t n, uint32_t tb) {
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (g_boff[mid] <= tb) lo = mid; else hi = mid; }
    return lo;
}

// Buckets of big inputs hold far more keys than one table round takes (a 1 Gbp genome: 2.4e5 keys per
// bucket = 256 rounds, each of which would re-read the whole bucket).  This pass splits such a bucket
// ONCE by its low key bits into 2^s contiguous sub-ranges of ~1000 keys, so that every later round reads
// only its own keys.  One workgroup per bucket at a time; the bucket (<= a few MB) stays in L2 between
// the counting and the scattering read.  (With only 4 sub-ranges -- the compact path's case -- the LDS atomics pile on
// four counters; a ballot/popcount partition without atomics was measured and is no faster, 6.9 vs 6.3 ms per 1.25e9
// keys: the pass is bound by the per-bucket latency chain, not by the counters.)
template <bool C32>
__global__ __launch_bounds__(K3_THREADS) void k3_split_kernel(BmhArgs a) {
    typedef typename K3Key<C32>::T KT;
    __shared__ uint32_t pos[K3_MAXB];
    __shared__ uint32_t wsum[K3_THREADS / 64];
    const uint32_t tid = threadIdx.x;
    for (uint32_t tb = blockIdx.x; tb < a.TB; tb += gridDim.x) {
        const uint32_t g = genome_of_bucket(a.g_boff, a.n, tb);
        const uint32_t sb = a.g_split[g];
        if (sb == 0) continue;
        const uint32_t R = 1u << sb;
        const uint64_t o0 = a.bucket_off[tb], nk = a.bucket_off[tb + 1] - o0;
        const KT *kb = (C32 ? (const KT *)a.keys32 : (const KT *)a.keys) + o0;
        const uint32_t bb = C32 ? a.g_bbits[g] : 0u;
        const uint64_t base = a.g_sub[g] + ((uint64_t)(tb - a.g_boff[g]) << sb);
        for (uint32_t i = tid; i < R; i += K3_THREADS) pos[i] = 0;
        __syncthreads();
        constexpr int PF = 8;
        for (uint64_t b0 = 0; b0 < nk; b0 += (uint64_t)PF * K3_THREADS) {
            KT kk[PF];
#pragma unroll
            for (int j = 0; j < PF; ++j) { const uint64_t i = b0 + (uint64_t)j * K3_THREADS + tid; kk[j] = i < nk ? kb[i] : 0; }
#pragma unroll
            for (int j = 0; j < PF; ++j) if (b0 + (uint64_t)j * K3_THREADS + tid < nk) atomicAdd(&pos[K3Key<C32>::sub_of(kk[j], R, bb)], 1u);
        }
        __syncthreads();
        // exclusive prefix of pos[0..R): each thread owns R / 256 consecutive counters (R <= 4096)
        const uint32_t per = (R + K3_THREADS - 1) / K3_THREADS, lo = min(R, tid * per), hi = min(R, lo + per);
        uint32_t sum = 0;
        for (uint32_t i = lo; i < hi; ++i) sum += pos[i];
        uint32_t incl = sum;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if ((tid & 63) >= (uint32_t)o) incl += v; }
        if ((tid & 63) == 63) wsum[tid >> 6] = incl;
        __syncthreads();
        uint32_t run = incl - sum;
        for (uint32_t w = 0; w < (tid >> 6); ++w) run += wsum[w];
        for (uint32_t i = lo; i < hi; ++i) { const uint32_t c = pos[i]; pos[i] = run; a.sub_off[base + i] = o0 + run; run += c; }
        if (tid == 0) a.sub_off[base + R] = o0 + nk;            // = the next bucket's first entry (same value), or the genome's end
        __syncthreads();
        KT *dst = (C32 ? (KT *)a.skeys32 : (KT *)a.skeys) + o0;
        for (uint64_t b0 = 0; b0 < nk; b0 += (uint64_t)PF * K3_THREADS) {
            KT kk[PF];
#pragma unroll
            for (int j = 0; j < PF; ++j) { const uint64_t i = b0 + (uint64_t)j * K3_THREADS + tid; kk[j] = i < nk ? kb[i] : 0; }
#pragma unroll
            for (int j = 0; j < PF; ++j)
                if (b0 + (uint64_t)j * K3_THREADS + tid < nk) dst[atomicAdd(&pos[K3Key<C32>::sub_of(kk[j], R, bb)], 1u)] = kk[j];
        }
        __syncthreads();
    }
}

// registers to +inf, weights to zero
__global__ __launch_bounds__(K3_THREADS) void k3_bmh_init_kernel(uint64_t *h, size_t nh, double *tw, uint32_t *redo, size_t n) {
    const size_t i = (size_t)blockIdx.x * K3_THREADS + threadIdx.x;
    if (i < nh) h[i] = BMH_INF;
    if (i < n) { tw[i] = 0.; redo[i] = 0; }
}

// the pruning bound for total weight W: the final maximum register is the max of m exponentials of
// rate W/m -- mean (m/W)(ln m + 0.58), sd 1.28 m/W; 1.25 x (mean + 6 sd) fails about once in 4000 genomes
__host__ __device__ inline double bmh_guess(double W, double m, double lnm) { return 1.25 * (m / W) * (lnm + 0.58 + 8.0); }

#ifndef D2G_K3_WPE
#define D2G_K3_WPE 5
#endif
#ifndef D2G_K3_WPE_LIGHT
#define D2G_K3_WPE_LIGHT 6
#endif
// LIGHT (the first pass): survivors are not walked here but appended to the workgroup's region of a queue in HBM and
// walked by k3_bmh_survivor_kernel afterwards, one per lane with every lane busy.  Without the descent (its stack, its
// LDS queue, its drains and their barriers) this kernel needs fewer registers and less LDS: 6 workgroups per CU instead of 5.
// The heavy form stays for the repeat passes (guess too small: rare) and as the fallback when a region overflows.
template <bool C32, bool LIGHT>
__global__ __launch_bounds__(K3_THREADS) __attribute__((amdgpu_waves_per_eu(LIGHT ? D2G_K3_WPE_LIGHT : D2G_K3_WPE))) void k3_bmh_main_kernel(BmhArgs a) {
    typedef typename K3Key<C32>::T KT;
    __shared__ SharedK3<C32> sh;
    __shared__ QEntry queue[LIGHT ? 1 : K3_QCAP];
    __shared__ uint32_t qn;
    const int tid = threadIdx.x;
    const uint32_t m = a.m;
    const CountTab<C32> t{sh.key, sh.cnt, &sh.ones};
    Proc stk[LIGHT ? 1 : BMH_STACK];
    if (tid == 0) qn = 0;
    if (LIGHT && tid == 0) a.gq_n[blockIdx.x] = 0;
    __syncthreads();
    // phase 2: one queued survivor per lane
    auto drain = [&]() {
        if constexpr (LIGHT) return;
        const uint32_t n = qn < (uint32_t)K3_QCAP ? qn : (uint32_t)K3_QCAP;
        for (uint32_t i = tid; i < n; i += K3_THREADS) {
            const QEntry q = queue[i];
            const double bound = V(a.guess[q.g]);
            Proc P = top_proc(q.d, (int)q.t);
            if (proc_next(P, m, bound)) walk_process(P, q.d, q.w, m, bound, a.h + (size_t)q.g * m, stk, a.status);
        }
        __syncthreads();
        if (tid == 0) qn = 0;
        __syncthreads();
    };
    // persistent workgroups: the queue has to live across buckets
    // each workgroup owns a contiguous range of buckets: the genome (and with it bound and
    // registers) changes rarely and is tracked incrementally -- a binary search plus four dependent
    // scalar loads per bucket cost 16 us of exposed latency per bucket
    const uint32_t per = (a.tb1 - a.tb0 + gridDim.x - 1) / gridDim.x;    // this launch walks buckets [tb0, tb1)
    const uint32_t tb_lo = a.tb0 + blockIdx.x * per, tb_hi = tb_lo + per < a.tb1 ? tb_lo + per : a.tb1;
    if (tb_lo >= tb_hi) return;
    QEntry *gq = LIGHT ? a.gq + a.gq_off[blockIdx.x] : nullptr;
    const uint32_t gq_cap = LIGHT ? (uint32_t)(a.gq_off[blockIdx.x + 1] - a.gq_off[blockIdx.x]) : 0u;
    // the count table is cleared ONCE; afterwards every walk hands it back clean
    for (int e = tid; e < K3_TAB; e += K3_THREADS) { sh.key[e] = K3Key<C32>::EMPTY; sh.cnt[e] = 0; }
    if (tid == 0) sh.ones = 0;
    // counter.h:123 `pair.second > threshold` on integer counts: c > thr  <=>  c >= floor(thr) + 1
    uint32_t cmin = 1;
    if (a.thr >= 1.0) cmin = a.thr >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)a.thr + 1u;
    uint32_t g = genome_of_bucket(a.g_boff, a.n, tb_lo), g_end = a.g_boff[g + 1];
    bool skip_g = a.redo_mode && !a.redo[g];
    double bound = V(a.guess[g]);
    uint32_t sbits = a.g_split ? a.g_split[g] : 0u;
    uint32_t bb = C32 ? a.g_bbits[g] : 0u, g_b0 = a.g_boff[g];
    uint64_t o_next = a.bucket_off[tb_lo];
    // Strip 0 = [0, 1) is relevant for EVERY element (counts are >= 1) and has width 1, so the early-out of proc_next,
    // (1 - uu) > bound * 1 * 1.000000001 with uu = ((r1 >> 11) + 1) 2^-53, is an INTEGER test on r1 >> 11: both sides are
    // multiples of 2^-53, 1 - uu exactly.  u0 = smallest r1 >> 11 that is NOT dropped, minus a margin of 2 (whatever passes
    // here takes the unchanged proc_next, which decides): a k-mer seen once costs one xor, one generator step, one compare.
    auto strip0_floor = [](double bnd) -> uint64_t {
        const double b53 = bnd * 1.0 * 1.000000001 * 0x1p53;
        if (!(b53 < 9007199254740988.0)) return 0;                       // large (or NaN) bound: everything goes to the exact test
        if (!(b53 >= 0.)) return 9007199254740989ull;                      // (a negative bound drops everything there too)
        return 9007199254740991ull - 2ull - (uint64_t)b53;
    };
    uint64_t u0 = strip0_floor(bound);
    // total weight = sum of the counts that pass the threshold: integers, summed per thread over the workgroup's buckets
    // of one genome and added to the genome's counter once (exact in any order; no block reduction per bucket)
    uint64_t twi = 0;
    auto flush_tw = [&](uint32_t gg) {
        if (a.redo_mode) { twi = 0; return; }
        uint64_t v = twi;
        for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
        if ((tid & 63) == 0 && v) atomicAdd((unsigned long long *)&a.tw_acc[gg], (unsigned long long)v);
        twi = 0;
    };
    __syncthreads();
    for (uint32_t tb = tb_lo; tb < tb_hi; ++tb) {
        const uint64_t o0 = o_next;
        o_next = a.bucket_off[tb + 1];
        const uint64_t nk = o_next - o0;
        if (tb >= g_end) {
            flush_tw(g);
            while (tb >= g_end) { ++g; g_end = a.g_boff[g + 1]; }
            skip_g = a.redo_mode && !a.redo[g];                              // second passes: only genomes whose guess failed
            bound = V(a.guess[g]);
            u0 = strip0_floor(bound);
            sbits = a.g_split ? a.g_split[g] : 0u;
            bb = C32 ? a.g_bbits[g] : 0u; g_b0 = a.g_boff[g];
        }
        if (nk == 0 || skip_g) continue;
        uint64_t *h = a.h + (size_t)g * m;
        // one range of keys: as many table rounds as its size asks for; each round's elements go through phase 1
        auto process = [&](const KT *kb, uint64_t rn, uint32_t shift) -> bool {
            uint32_t R = 1;
            while ((uint64_t)R * a.round_keys < rn) R <<= 1;
            for (uint32_t r = 0; r < R; ++r) {
                // a single round over at most round_keys (< K3_TAB) keys cannot fill the table
                if (R == 1 && rn < (uint64_t)K3_TAB) insert_round<C32, false>(t, kb, rn, 1, 0, shift, bb);
                else if (!insert_round<C32, true>(t, kb, rn, R, r, shift, bb)) return false;
                // Walk the table slots directly, emptying them on the way.  (r01 squeezed the occupied slots into a dense list
                // first -- worth it when the per-element walk was heavy; the compaction, 3.5 ms per call with its three
                // barriers per round, costs more than the 40 % idle lanes here.)
                auto survivor = [&](const Proc &P, uint64_t d, double w, int tt) {
                    if (D2G_K3_EXP == 5) return;                         // timing experiment: survivors dropped
                    const uint32_t slot = atomicAdd(&qn, 1u);
                    if constexpr (LIGHT) {
                        // the region's capacity is twice the expected count; qn keeps counting so that the end of the kernel sees an overflow
                        if (slot < gq_cap) { QEntry *q = gq + slot; q->d = d; q->w = w; q->t = (uint32_t)tt; q->g = g; }
                    } else {
                        if (slot < (uint32_t)K3_QCAP) { queue[slot].d = d; queue[slot].w = w; queue[slot].t = (uint32_t)tt; queue[slot].g = g; }
                        else walk_process(P, d, w, m, bound, h, stk, a.status);   // queue full: do it now
                    }
                };
                auto element = [&](KT key, uint32_t cc) {
                    // the element's id is the masked key (maskfn, src/enums.h:136-140): on the compact path it is formed
                    // here, once per DISTINCT k-mer, from the stored word and the bucket
                    uint64_t d;
                    if constexpr (C32) d = wang64(k3c_kmer(key, tb - g_b0, bb, a.hb) ^ a.xormask);
                    else d = key;
                    twi += cc;
                    uint64_t rng0 = d ^ (0xA0761D6478BD642Full ^ 0x8EBC6AF09C88C6E3ull);      // top_proc(d, 0).rng
                    if ((wy_next(rng0) >> 11) >= u0) {
                        Proc P = top_proc(d, 0);
                        if (proc_next(P, m, bound)) survivor(P, d, (double)cc, 0);
                    }
                    if (cc > 1) {                                        // the strips above [0, 1)
                        const double w = (double)cc;
                        const int nt = top_count(w);
                        for (int tt = 1; tt < nt; ++tt) {
                            Proc P = top_proc(d, tt);
                            if (proc_next(P, m, bound)) survivor(P, d, w, tt);
                        }
                    }
                };
                for (uint32_t e = tid; e < (uint32_t)K3_TAB; e += K3_THREADS) {
                    const uint32_t cc = sh.cnt[e];
                    if (cc) {
                        const KT key = sh.key[e];
                        sh.key[e] = K3Key<C32>::EMPTY; sh.cnt[e] = 0;
                        if (cc >= cmin && D2G_K3_EXP != 3 && D2G_K3_EXP != 4 && D2G_K3_EXP != 8) element(key, cc);
                    }
                }
                if (tid == 0) {
                    const uint32_t ones = sh.ones;
                    if (ones) { sh.ones = 0; if (ones >= cmin) element(K3Key<C32>::EMPTY, ones); }
                }
                __syncthreads();
                if constexpr (!LIGHT) { if (qn >= (uint32_t)K3_QDRAIN) drain(); }
            }
            return true;
        };
        bool fine;
        if (sbits == 0) {
            fine = process((C32 ? (const KT *)a.keys32 : (const KT *)a.keys) + o0, nk, 0);
        } else {                                             // big inputs: the bucket's pre-split sub-ranges, one after the other
            fine = true;
            const uint64_t sub0 = a.g_sub[g] + ((uint64_t)(tb - a.g_boff[g]) << sbits);
            uint64_t lo = a.sub_off[sub0];
            for (uint32_t rg = 0; rg < (1u << sbits) && fine; ++rg) {
                const uint64_t hi = a.sub_off[sub0 + rg + 1];
                if (hi > lo) fine = process((C32 ? (const KT *)a.skeys32 : (const KT *)a.skeys) + lo, hi - lo, sbits);
                lo = hi;
            }
        }
        if (!fine) { if (tid == 0) atomicExch(a.status, 1); return; }
        // The bound is never tightened per workgroup: thousands of same-address atomics per genome serialise in L2
        // (measured 25 ms per 4e5 workgroups) and the guess is already within ~2x of the final maximum.
    }
    flush_tw(g);
    __syncthreads();
    if constexpr (LIGHT) {
        if (tid == 0) {
            a.gq_n[blockIdx.x] = qn < gq_cap ? qn : gq_cap;
            if (qn > gq_cap) atomicExch(a.status, 4);                    // the host repeats the pass with the heavy kernel
        }
    } else drain();
}

// the survivors of the light first pass, region by region, one per lane
__global__ __launch_bounds__(K3_THREADS) void k3_bmh_survivor_kernel(BmhArgs a) {
    Proc stk[BMH_STACK];
    const uint32_t n = a.gq_n[blockIdx.x];
    const QEntry *q0 = a.gq + a.gq_off[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < n; i += K3_THREADS) {
        const QEntry q = q0[i];
        const double bound = V(a.guess[q.g]);
        Proc P = top_proc(q.d, (int)q.t);
        if (proc_next(P, a.m, bound)) walk_process(P, q.d, q.w, a.m, bound, a.h + (size_t)q.g * a.m, stk, a.status);
    }
}

// after a pass: was the bound that pruned points at least the final maximum register?  The first
// pass also sums the genome's total weight, from which a failed guess is recomputed.
__global__ __launch_bounds__(K3_THREADS) void k3_bmh_verify_kernel(BmhArgs a) {
    __shared__ uint64_t red[8];
    const uint32_t g = blockIdx.x + a.g0;
    if (a.redo_mode && !a.redo[g]) return;
    const uint64_t hm = block_hmax(a.h + (size_t)g * a.m, a.m, red);
    if (threadIdx.x == 0) {
        double tw;
        if (!a.redo_mode) a.tw[g] = tw = (double)a.tw_acc[g]; else tw = a.tw[g];   // counts: exact below 2^53
        const double guess = V(a.guess[g]);
        const bool bad = tw > 0. && !(V(hm) <= guess);               // no element at all: registers stay +inf, nothing to redo
        a.redo[g] = bad;
        if (bad) {
            const double better = bmh_guess(tw, (double)a.m, (double)__logf((float)a.m));
            a.guess[g] = dbits(better > 16. * guess ? better : 16. * guess);
            atomicAdd(a.nredo, 1u);
        }
    }
}

// R11 alone: distinct (key, count) of every bucket, compacted to the front of the bucket's region
template <bool C32>
__global__ __launch_bounds__(K3_THREADS) void k3_count_kernel(BmhArgs a) {
    typedef typename K3Key<C32>::T KT;
    __shared__ SharedK3<C32> sh;
    const int tid = threadIdx.x;
    const uint32_t tb = blockIdx.x;
    const uint64_t o0 = a.bucket_off[tb], nk = a.bucket_off[tb + 1] - o0;
    if (tid == 0) sh.misc = 0;
    if (nk == 0) { if (tid == 0) a.bucket_nd[tb] = 0; return; }
    const CountTab<C32> t{sh.key, sh.cnt, &sh.ones};
    const uint32_t g = (a.g_split || C32) ? genome_of_bucket(a.g_boff, a.n, tb) : 0u;
    const uint32_t sbits = a.g_split ? a.g_split[g] : 0u;
    const uint32_t bb = C32 ? a.g_bbits[g] : 0u;
    const uint32_t nranges = 1u << sbits;
    const uint64_t sub0 = sbits ? a.g_sub[g] + ((uint64_t)(tb - a.g_boff[g]) << sbits) : 0;
    for (uint32_t rg = 0; rg < nranges; ++rg) {
        const uint64_t lo = sbits ? a.sub_off[sub0 + rg] : o0;
        const uint64_t rn = sbits ? a.sub_off[sub0 + rg + 1] - lo : nk;
        if (rn == 0) continue;
        const KT *kb = (C32 ? (const KT *)(sbits ? a.skeys32 : a.keys32) : (const KT *)(sbits ? a.skeys : a.keys)) + lo;
        uint32_t R = 1;
        while ((uint64_t)R * a.round_keys < rn) R <<= 1;
        for (uint32_t r = 0; r < R; ++r) {
            if (!count_round<C32>(t, kb, rn, R, r, sbits, bb)) { if (tid == 0) atomicExch(a.status, 1); return; }
            const uint32_t ne = compact_elements<C32>(t, &sh.nelem, a.thr);
            const uint32_t j0 = sh.misc;
            if (a.out_keys)
                for (uint32_t e = tid; e < ne; e += K3_THREADS) {
                    uint64_t key;
                    if constexpr (C32) key = wang64(k3c_kmer(sh.key[e], tb - a.g_boff[g], bb, a.hb) ^ a.xormask);
                    else key = sh.key[e];
                    a.out_keys[o0 + j0 + e] = key; a.out_counts[o0 + j0 + e] = sh.cnt[e];
                }
            __syncthreads();
            if (tid == 0) sh.misc = j0 + ne;
            __syncthreads();
        }
    }
    if (tid == 0) a.bucket_nd[tb] = sh.misc;
}

// ---------------------------------------------------------------------------------------------
// explicit weighted sets (wsketch.cpp:54-73 minwise_det / 17-51 rowwise CSR): elements come
// from (id, weight) arrays instead of the count table
// ---------------------------------------------------------------------------------------------
struct WsArgs {
    const uint64_t *ids;
    const double *w;             // nullptr => 1.0
    const uint32_t *blk_set;     // set of workgroup
    const uint64_t *blk_lo;      // first element
    const uint32_t *blk_cnt;     // element count
    uint32_t m;
    uint64_t *h;                 // [nsets][m]
    const uint64_t *guess;       // [nsets] pruning bound of this pass
    const uint32_t *redo;        // [nsets] (redo_mode) sets to walk again
    int redo_mode;
    int *status;
    uint64_t *arg;               // argmin pass: [nsets][m] owner positions, pre-set to ~0
    const uint64_t *set_lo;      // argmin pass: [nsets] first element of each set
};

__device__ __forceinline__ bool ws_fetch(const WsArgs &a, uint64_t idx, uint64_t &d, double &w, int *status) {
    d = a.ids[idx];
    w = a.w ? a.w[idx] : 1.0;
    if (!(w > 0.)) return false;                                   // BagMinHash2::update ignores w <= 0
    if (!(w <= 0x1p53)) { atomicExch(status, 2); return false; }   // outside the level set (also NaN)
    return true;
}

__global__ __launch_bounds__(K3_THREADS) void k3_bmh_sets_kernel(WsArgs a) {
    const uint32_t set = a.blk_set[blockIdx.x];
    if (a.redo_mode && !a.redo[set]) return;
    const uint64_t lo = a.blk_lo[blockIdx.x], cnt = a.blk_cnt[blockIdx.x];
    const double bound = V(a.guess[set]);
    uint64_t *hg = a.h + (size_t)set * a.m;
    Proc stk[BMH_STACK];
    for (uint64_t e = threadIdx.x; e < cnt; e += K3_THREADS) {
        uint64_t d; double w;
        if (ws_fetch(a, lo + e, d, w, a.status)) walk_element(d, w, a.m, bound, hg, stk, a.status);
    }
}

// after the registers are final: one more walk under the verified bound records, per register, the position
// (within its set) of the element whose point it holds
__global__ __launch_bounds__(K3_THREADS) void k3_bmh_sets_argmin_kernel(WsArgs a) {
    const uint32_t set = a.blk_set[blockIdx.x];
    const uint64_t lo = a.blk_lo[blockIdx.x], cnt = a.blk_cnt[blockIdx.x];
    const double bound = V(a.guess[set]);
    Proc stk[BMH_STACK];
    for (uint64_t e = threadIdx.x; e < cnt; e += K3_THREADS) {
        uint64_t d; double w;
        if (ws_fetch(a, lo + e, d, w, a.status))
            walk_element(d, w, a.m, bound, ArgSink{a.h + (size_t)set * a.m, a.arg + (size_t)set * a.m, lo + e - a.set_lo[set]}, stk, a.status);
    }
}

// per set: max(h) <= guess ?  else raise the guess and flag the set
__global__ __launch_bounds__(K3_THREADS) void k3_sets_verify_kernel(const uint64_t *h, uint32_t m, uint64_t *guess, uint32_t *redo,
                                                                   const double *tw, uint32_t *nredo, int redo_mode) {
    __shared__ uint64_t red[8];
    const uint32_t set = blockIdx.x;
    if (redo_mode && !redo[set]) return;
    const uint64_t hm = block_hmax(h + (size_t)set * m, m, red);
    if (threadIdx.x == 0) {
        const double g = V(guess[set]);
        const bool bad = tw[set] > 0. && !(V(hm) <= g);
        redo[set] = bad;
        if (bad) { guess[set] = dbits(16. * g); atomicAdd(nredo, 1u); }
    }
}

uint32_t ceil_log2(uint64_t x) { uint32_t b = 0; while ((1ull << b) < x) ++b; return b; }

}  // namespace

// grow-only work buffers of the --multiset path (owned by a sketcher or by a one-shot call)
struct d2g_k3_state {
    d2g_ctx *ctx = nullptr;
    uint32_t *d_gtab = nullptr; size_t cap_gtab = 0;        // g_bbits [n] + g_boff [n+1]
    uint64_t *d_koff = nullptr; size_t cap_koff = 0;        // [n+1]
    uint32_t *d_bucket_cnt = nullptr; size_t cap_bcnt = 0;
    uint64_t *d_bucket_off = nullptr; size_t cap_boff = 0;
    uint64_t *d_cursor = nullptr; size_t cap_cursor = 0;
    uint32_t *d_l2 = nullptr; size_t cap_l2 = 0;     // two-level split: coarse bucket table
    uint32_t *d_blk_coarse = nullptr; size_t cap_blk_coarse = 0;
    uint64_t *d_gq = nullptr; size_t cap_gq = 0;     // survivors of the light first pass (QEntry)
    uint64_t *d_gq_off = nullptr; size_t cap_gq_off = 0;   // [grid+1] region offsets, then [grid] u32 counts
    int light_overflows = 0;
    hipStream_t xs = nullptr;                        // second stream of the sub-batch pipeline
    hipEvent_t ev_a[8] = {}, ev_b = nullptr;
    int pipelined_calls = 0;
    uint64_t *d_keys = nullptr; size_t cap_keys = 0;
    uint64_t *d_skeys = nullptr; size_t cap_skeys = 0;      // big inputs: keys regrouped by sub-range
    uint32_t *d_gblk = nullptr; size_t cap_gblk = 0;        // compact path: first launch-plan block of each genome
    uint16_t *d_tile_cnt = nullptr; size_t cap_tile_cnt = 0;
    uint32_t *d_tile_off = nullptr; size_t cap_tile_off = 0;
    uint64_t *d_sub_off = nullptr; size_t cap_sub_off = 0;
    uint32_t *d_gsplit = nullptr; size_t cap_gsplit = 0;
    uint64_t *d_gsub = nullptr; size_t cap_gsub = 0;
    uint64_t *d_h = nullptr; size_t cap_h = 0;
    double *d_tw = nullptr; size_t cap_tw = 0;
    int *d_status = nullptr;               // [0] status, [1] nredo
    uint64_t *d_guess = nullptr; size_t cap_guess = 0;
    double *d_tw_bucket = nullptr; size_t cap_twb = 0;
    int last_nredo = 0;
    uint32_t *d_redo = nullptr; size_t cap_redo = 0;
    uint32_t *d_out_counts = nullptr; size_t cap_oc = 0;
    uint32_t *d_bucket_nd = nullptr; size_t cap_nd = 0;
    uint64_t *d_out_keys = nullptr; size_t cap_ok = 0;
};

void d2g_k3_state_destroy(d2g_k3_state *st) {
    if (!st) return;
    (void)hipFree(st->d_gtab); (void)hipFree(st->d_koff); (void)hipFree(st->d_bucket_cnt); (void)hipFree(st->d_bucket_off); (void)hipFree(st->d_cursor); (void)hipFree(st->d_l2); (void)hipFree(st->d_blk_coarse); (void)hipFree(st->d_gq); (void)hipFree(st->d_gq_off);
    for (auto &e : st->ev_a) if (e) (void)hipEventDestroy(e);
    if (st->ev_b) (void)hipEventDestroy(st->ev_b);
    if (st->xs) (void)hipStreamDestroy(st->xs);
    (void)hipFree(st->d_keys); (void)hipFree(st->d_skeys); (void)hipFree(st->d_sub_off); (void)hipFree(st->d_gsplit); (void)hipFree(st->d_gsub); (void)hipFree(st->d_h); (void)hipFree(st->d_tw);
    (void)hipFree(st->d_status); (void)hipFree(st->d_guess); (void)hipFree(st->d_tw_bucket); (void)hipFree(st->d_redo); (void)hipFree(st->d_out_counts); (void)hipFree(st->d_bucket_nd);
    (void)hipFree(st->d_out_keys); (void)hipFree(st->d_gblk); (void)hipFree(st->d_tile_cnt); (void)hipFree(st->d_tile_off);
    delete st;
}

namespace {

struct K3Host {
    std::vector<uint32_t> gtab;        // bbits[n] then boff[n+1]
    std::vector<uint64_t> gk;          // k-mers per genome
    std::vector<uint64_t> koff;        // [n+1] exclusive prefix of gk
    std::vector<uint32_t> gsplit;      // [n] log2(sub-ranges per bucket) of big genomes, 0 otherwise
    std::vector<uint64_t> gsub;        // [n+1] first sub-range of genome g
    std::vector<uint32_t> gblk;        // [n+1] first launch-plan block of genome g (compact path)
    bool compact = false;              // k <= 21: 4-byte stored words + tile-sorted split (k3c_*)
    uint32_t hb = 0;                   // compact path: hi bits = max(0, 2k - 32)
    bool any_split = false;
    uint64_t total = 0;
    uint32_t TB = 0;
    uint32_t l1bits = K3_L1BITS;        // generic path: bucket bits the scatter resolves itself
    std::vector<uint32_t> l2_tb0, l2_bits;   // coarse buckets that k3_refine_kernel spreads over their buckets
    std::vector<uint32_t> l2_start;          // [n+1] first entry of genome g in the two arrays above
};

int k3_layout(d2g_ctx *ctx, const uint32_t *run_len, const uint64_t *genome_run_off, size_t n, int k, K3Host &kh) {
    kh.gtab.assign(2 * n + 1, 0);
    kh.gk.assign(n, 0);
    kh.gblk.assign(n + 1, 0);
    kh.hb = k > 16 ? (uint32_t)(2 * k - 32) : 0u;
    // the compact path (4-byte stored words, tile-sorted split) halves the chain's HBM traffic but is ~11 % slower end to
    // end (one Wang mix per distinct k-mer moves into the issue-bound main pass): opt-in with D2G_K3_COMPACT=1, k <= 21
    kh.compact = false;
    if (const char *e = ctx->tune.get("D2G_K3_COMPACT")) if (e[0] == '1') kh.compact = kh.hb <= (uint32_t)K3C_MAXBBITS;
    uint64_t tb = 0;
    uint64_t bucket_keys = K3_TARGET, sub_keys = K3_TARGET;
    kh.l1bits = K3_L1BITS; kh.l2_tb0.clear(); kh.l2_bits.clear(); kh.l2_start.assign(n + 1, 0);
    if (const char *e = ctx->tune.get("D2G_K3_L1BITS")) { const int v = std::atoi(e); if (v >= 0 && v <= K3_MAXBBITS) kh.l1bits = (uint32_t)v; }
    if (const char *e = ctx->tune.get("D2G_K3_BUCKET_KEYS")) { const long v = std::atol(e); if (v >= 1) bucket_keys = (uint64_t)v; }
    if (const char *e = ctx->tune.get("D2G_K3_SUB_KEYS")) { const long v = std::atol(e); if (v >= 1) sub_keys = (uint64_t)v; }
    for (size_t g = 0; g < n; ++g) {
        uint64_t nk = 0, chunks = 0;
        for (uint64_t r = genome_run_off[g]; r < genome_run_off[g + 1]; ++r) {
            const uint64_t rk = (uint64_t)run_len[r] - k + 1;
            nk += rk; chunks += div_up<uint64_t>(rk, K1_CHUNK);
        }
        D2G_CHECK(ctx, nk < (1ull << 32), "--multiset: more than 2^32 k-mers in one input");
        kh.gk[g] = nk; kh.total += nk;
        kh.gblk[g + 1] = kh.gblk[g] + (uint32_t)div_up<uint64_t>(chunks, K1_BLOCK_CHUNKS);
        uint32_t bb;
        if (kh.compact) bb = std::max<uint32_t>(kh.hb, std::min<uint32_t>(K3C_MAXBBITS, ceil_log2((nk + K3C_TARGET - 1) / K3C_TARGET)));
        else bb = std::min<uint32_t>(K3_MAXBBITS, ceil_log2((nk + bucket_keys - 1) / bucket_keys));
        kh.gtab[g] = bb;
        kh.gtab[n + g] = (uint32_t)tb;
        kh.l2_start[g] = (uint32_t)kh.l2_tb0.size();
        if (!kh.compact && bb > kh.l1bits)
            for (uint32_t c = 0; c < (1u << kh.l1bits); ++c) {
                kh.l2_tb0.push_back((uint32_t)tb + (c << (bb - kh.l1bits)));
                kh.l2_bits.push_back((bb << 8) | (bb - kh.l1bits));
            }
        tb += 1ull << bb;
        D2G_CHECK(ctx, tb < (1ull << 31), "--multiset: too many buckets in one batch; use smaller batches");
    }
    kh.gtab[2 * n] = (uint32_t)tb;
    kh.TB = (uint32_t)tb;
    kh.l2_start[n] = (uint32_t)kh.l2_tb0.size();
    kh.koff.assign(n + 1, 0);
    for (size_t g = 0; g < n; ++g) kh.koff[g + 1] = kh.koff[g] + kh.gk[g];
    // big inputs: buckets averaging more than K3_SPLIT_MIN keys are split once more (k3_split_kernel)
    // into sub-ranges of ~K3_TARGET keys; D2G_K3_SPLIT_MIN lowers the limit so that tests reach the path
    // (the compact path aims for 4x larger buckets -- 64-byte runs in the tile sort -- and always splits them here into
    // table-sized sub-ranges: a table round that re-reads its whole key range made the main pass 55 % slower, and a
    // bucket staged in LDS for the rounds cost more in occupancy than it saved: 42 ms instead of 20)
    uint64_t split_min = kh.compact ? K3_ROUND_KEYS : K3_SPLIT_MIN;
    if (const char *e = ctx->tune.get("D2G_K3_SPLIT_MIN")) { const long v = std::atol(e); if (v >= 1) split_min = (uint64_t)v; }
    kh.gsplit.assign(n, 0);
    kh.gsub.assign(n + 1, 0);
    for (size_t g = 0; g < n; ++g) {
        const uint64_t B = 1ull << kh.gtab[g], mean = kh.gk[g] / B;
        if (mean > split_min) {
            kh.gsplit[g] = std::min<uint32_t>(K3_MAXBBITS, ceil_log2((mean + sub_keys - 1) / sub_keys));
            kh.any_split = true;
        }
        // + 1: the end of a genome's last sub-range gets its OWN entry.  Sharing it with the next split genome's first
        // entry is only right when no unsplit genome lies between the two (their key ranges are then not adjacent and
        // the two writers race with different values)
        kh.gsub[g + 1] = kh.gsub[g] + (kh.gsplit[g] ? (B << kh.gsplit[g]) + 1 : 0);
    }
    return D2G_OK;
}

// count (R11) + sketch (R12) of one staged batch; results stay on the device in st->d_h / st->d_tw
int k3_run(d2g_ctx *ctx, d2g_k3_state *st, hipStream_t s, const KmerArgs &km, size_t nblk, const K3Host &kh, size_t n,
           uint64_t xormask, size_t m, double thr, bool count_only, bool distinct_only = false) {
    const uint32_t TB = kh.TB;
    if (int rc = d2g_grow(ctx, &st->d_gtab, &st->cap_gtab, 2 * n + 1)) return rc;
    if (int rc = d2g_grow(ctx, &st->d_bucket_cnt, &st->cap_bcnt, (size_t)TB + 1)) return rc;
    if (int rc = d2g_grow(ctx, &st->d_bucket_off, &st->cap_boff, (size_t)TB + 1)) return rc;
    if (int rc = d2g_grow(ctx, &st->d_cursor, &st->cap_cursor, (size_t)TB + 1)) return rc;
    // 8 bytes of masked key per k-mer on the generic path, 4 bytes of stored word on the compact one
    const uint64_t key_words = kh.compact ? (kh.total + 1) / 2 : kh.total;
    if (int rc = d2g_grow(ctx, &st->d_keys, &st->cap_keys, std::max<uint64_t>(key_words, 1))) return rc;
    if (!st->d_status) D2G_HIP(ctx, hipMalloc((void **)&st->d_status, 2 * sizeof(int)));
    if (int rc = d2g_grow(ctx, &st->d_koff, &st->cap_koff, n + 1)) return rc;
    D2G_HIP(ctx, hipMemcpyAsync(st->d_gtab, kh.gtab.data(), (2 * n + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    D2G_HIP(ctx, hipMemcpyAsync(st->d_koff, kh.koff.data(), (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    D2G_HIP(ctx, hipMemsetAsync(st->d_bucket_cnt, 0, ((size_t)TB + 1) * sizeof(uint32_t), s));
    D2G_HIP(ctx, hipMemsetAsync(st->d_status, 0, 2 * sizeof(int), s));
    d2g_timer tm(ctx, &ctx->ev_k3, s);
    // Sub-batch pipeline (generic path, sketching): the batch is cut into genome ranges; the bucketing passes of range j + 1
    // (hist, scan, scatter, refine: bound by memory) run on the caller's stream while the counting pass of range j (bound by
    // latency and instruction issue) runs on a second stream.  Ranges are independent: disjoint genomes, buckets, key regions.
    K3Args ka;
    std::memset(&ka, 0, sizeof(ka));
    std::vector<size_t> sub{0, n};                                       // genome boundaries of the ranges
    bool pipeline = false;
    if (!kh.compact && !kh.any_split && !count_only && !D2G_K3_EXP && n >= 2 && kh.gblk[n] == nblk) {
        size_t want = (n >= 8 && kh.total >= 200000000ull) ? 4 : 1;
        if (const char *e = ctx->tune.get("D2G_K3_SUBBATCH")) { const int v = std::atoi(e); if (v >= 1 && v <= 8) want = (size_t)v; }
        want = std::min(want, n);
        if (want > 1) {
            sub.assign(1, 0);
            for (size_t j = 1; j < want; ++j) {                          // cut by k-mer count
                const uint64_t target = kh.total / want * j;
                size_t g = sub.back() + 1;
                while (g < n - (want - j) && kh.koff[g] < target) ++g;
                sub.push_back(g);
            }
            sub.push_back(n);
            pipeline = true;
        }
    }
    auto stage_a = [&](size_t g_lo, size_t g_hi) {
        K3Args a = ka;
        const unsigned blk_lo = kh.gblk[g_lo], nb = (g_lo == 0 && g_hi == n) ? (unsigned)nblk : kh.gblk[g_hi] - blk_lo;
        a.km.blk0 = blk_lo; a.g0 = (uint32_t)g_lo;
        if (nb) hipLaunchKernelGGL(k3_hist_kernel, dim3(nb), dim3(K1_THREADS), 0, s, a);
        hipLaunchKernelGGL(k3_scan_kernel, dim3((unsigned)(g_hi - g_lo)), dim3(K3_THREADS), 0, s, a);
        if (nb) hipLaunchKernelGGL(k3_scatter_kernel, dim3(nb), dim3(K1_THREADS), 0, s, a);
        const unsigned l2_lo = kh.l2_start[g_lo], nl = kh.l2_start[g_hi] - l2_lo;
        if (nb && nl && D2G_K3_EXP != 1 && D2G_K3_EXP != 2) {
            a.l2_tb0 += l2_lo; a.l2_bits += l2_lo;
            hipLaunchKernelGGL(k3_refine_kernel, dim3(nl), dim3(K3_THREADS), 0, s, a);
        }
    };
    if (kh.compact) {
        // 4-byte stored words, tile-sorted split: histogram per tile -> per-tile write offsets -> coalesced flush
        D2G_CHECK(ctx, kh.gblk[n] == nblk, "internal: K3 block layout disagrees with the launch plan");
        const size_t ntiles = nblk * K1_CPT;
        if (int rc = d2g_grow(ctx, &st->d_gblk, &st->cap_gblk, n + 1)) return rc;
        if (int rc = d2g_grow(ctx, &st->d_tile_cnt, &st->cap_tile_cnt, std::max<size_t>(ntiles, 1) * K3C_MAXB)) return rc;
        if (int rc = d2g_grow(ctx, &st->d_tile_off, &st->cap_tile_off, std::max<size_t>(ntiles, 1) * K3C_MAXB)) return rc;
        D2G_HIP(ctx, hipMemcpyAsync(st->d_gblk, kh.gblk.data(), (n + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        K3cArgs c;
        c.km = km; c.g_bbits = st->d_gtab; c.g_boff = st->d_gtab + n; c.g_koff = st->d_koff; c.g_blk = st->d_gblk;
        c.tile_cnt = st->d_tile_cnt; c.tile_off = st->d_tile_off; c.bucket_cnt = st->d_bucket_cnt; c.bucket_off = st->d_bucket_off;
        c.keys32 = reinterpret_cast<uint32_t *>(st->d_keys); c.hb = kh.hb; c.TB = TB;
        if (nblk) hipLaunchKernelGGL(k3c_hist_kernel, dim3((unsigned)nblk), dim3(K1_THREADS), 0, s, c);
        hipLaunchKernelGGL(k3c_scan_kernel, dim3((unsigned)n), dim3(K3_THREADS), 0, s, c);
        if (nblk) {
            D2G_HIP(ctx, hipFuncSetAttribute((const void *)k3c_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(K3cScatterLds)));
            hipLaunchKernelGGL(k3c_scatter_kernel, dim3((unsigned)nblk), dim3(K1_THREADS), sizeof(K3cScatterLds), s, c);
        }
    } else {
        K3Args a;
        a.km = km; a.xormask = xormask;
        a.g_bbits = st->d_gtab; a.g_boff = st->d_gtab + n; a.g_koff = st->d_koff;
        a.bucket_cnt = st->d_bucket_cnt; a.bucket_off = st->d_bucket_off; a.cursor = st->d_cursor; a.keys = st->d_keys;
        a.TB = TB;
        a.l1bits = kh.l1bits; a.coarse = nullptr; a.l2_tb0 = nullptr; a.l2_bits = nullptr;
        const size_t nl2 = kh.l2_tb0.size();
        if (nl2) {
            // the coarse keys borrow the sub-range buffer: k3_split_kernel (big inputs) runs after the refinement
            if (int rc = d2g_grow(ctx, &st->d_skeys, &st->cap_skeys, std::max<uint64_t>(key_words, 1))) return rc;
            if (int rc = d2g_grow(ctx, &st->d_l2, &st->cap_l2, 2 * nl2)) return rc;
            D2G_HIP(ctx, hipMemcpyAsync(st->d_l2, kh.l2_tb0.data(), nl2 * sizeof(uint32_t), hipMemcpyHostToDevice, s));
            D2G_HIP(ctx, hipMemcpyAsync(st->d_l2 + nl2, kh.l2_bits.data(), nl2 * sizeof(uint32_t), hipMemcpyHostToDevice, s));
            a.coarse = st->d_skeys; a.l2_tb0 = st->d_l2; a.l2_bits = st->d_l2 + nl2;
        }
        if (int rc = d2g_grow(ctx, &st->d_blk_coarse, &st->cap_blk_coarse, std::max<size_t>(nblk, 1) << kh.l1bits)) return rc;
        a.blk_coarse = st->d_blk_coarse;
        a.g0 = 0; a.n_genomes = (uint32_t)n;
        ka = a;
        if (!pipeline) stage_a(0, n);
    }
    BmhArgs b;
    std::memset(&b, 0, sizeof(b));
    b.keys = st->d_keys; b.keys32 = reinterpret_cast<const uint32_t *>(st->d_keys); b.g_bbits = st->d_gtab; b.hb = kh.hb; b.xormask = xormask;
    b.bucket_off = st->d_bucket_off; b.g_boff = st->d_gtab + n;
    b.n = (uint32_t)n; b.TB = TB; b.m = (uint32_t)m; b.thr = thr; b.status = st->d_status;
    if (kh.any_split) {
        const uint64_t nsub = kh.gsub[n];
        if (int rc = d2g_grow(ctx, &st->d_skeys, &st->cap_skeys, std::max<uint64_t>(key_words, 1))) return rc;
        if (int rc = d2g_grow(ctx, &st->d_sub_off, &st->cap_sub_off, nsub + 1)) return rc;
        if (int rc = d2g_grow(ctx, &st->d_gsplit, &st->cap_gsplit, n)) return rc;
        if (int rc = d2g_grow(ctx, &st->d_gsub, &st->cap_gsub, n + 1)) return rc;
        D2G_HIP(ctx, hipMemcpyAsync(st->d_gsplit, kh.gsplit.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        D2G_HIP(ctx, hipMemcpyAsync(st->d_gsub, kh.gsub.data(), (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
        b.g_split = st->d_gsplit; b.g_sub = st->d_gsub; b.sub_off = st->d_sub_off; b.skeys = st->d_skeys;
        b.skeys32 = reinterpret_cast<uint32_t *>(st->d_skeys);
        const unsigned gs = (unsigned)std::min<size_t>(TB, (size_t)ctx->num_cus * 8);
        hipLaunchKernelGGL(kh.compact ? k3_split_kernel<true> : k3_split_kernel<false>, dim3(gs), dim3(K3_THREADS), 0, s, b);
    }
    b.round_keys = K3_ROUND_KEYS;
    if (const char *e = ctx->tune.get("D2G_K3_ROUND_KEYS")) { const int v = std::atoi(e); if (v >= 1 && v <= K3_ROUND_KEYS) b.round_keys = (uint32_t)v; }
    if (count_only) {
        if (!distinct_only) {
            if (int rc = d2g_grow(ctx, &st->d_out_keys, &st->cap_ok, std::max<uint64_t>(kh.total, 1))) return rc;
            if (int rc = d2g_grow(ctx, &st->d_out_counts, &st->cap_oc, std::max<uint64_t>(kh.total, 1))) return rc;
            b.out_keys = st->d_out_keys; b.out_counts = st->d_out_counts;
        }
        if (int rc = d2g_grow(ctx, &st->d_bucket_nd, &st->cap_nd, (size_t)TB + 1)) return rc;
        b.bucket_nd = st->d_bucket_nd;
        if (TB) hipLaunchKernelGGL(kh.compact ? k3_count_kernel<true> : k3_count_kernel<false>, dim3(TB), dim3(K3_THREADS), 0, s, b);
    } else {
        if (int rc = d2g_grow(ctx, &st->d_h, &st->cap_h, std::max<size_t>(n * m, 1))) return rc;
        if (int rc = d2g_grow(ctx, &st->d_tw, &st->cap_tw, std::max<size_t>(n, 1))) return rc;
        if (int rc = d2g_grow(ctx, &st->d_guess, &st->cap_guess, std::max<size_t>(n, 1))) return rc;
        if (int rc = d2g_grow(ctx, &st->d_redo, &st->cap_redo, std::max<size_t>(n, 1))) return rc;
        if (int rc = d2g_grow(ctx, &st->d_tw_bucket, &st->cap_twb, (size_t)TB + 1)) return rc;
        D2G_HIP(ctx, hipMemsetAsync(st->d_tw_bucket, 0, ((size_t)TB + 1) * sizeof(double), s));
        // first guess: with no count threshold the total weight IS the k-mer count; with one it is an
        // upper bound (a too small guess only costs a second pass, which then knows the exact weight)
        double scale = 1.0;
        if (const char *e = ctx->tune.get("D2G_K3_GUESS_SCALE")) { const double v = std::atof(e); if (v > 0.) scale = v; }   // tests force the redo path
        std::vector<uint64_t> guess(n);
        const double lnm = std::log((double)m);
        for (size_t g = 0; g < n; ++g) {
            const double gv = scale * bmh_guess((double)std::max<uint64_t>(kh.gk[g], 1), (double)m, lnm);
            std::memcpy(&guess[g], &gv, 8);
        }
        D2G_HIP(ctx, hipMemcpyAsync(st->d_guess, guess.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice, s));
        b.h = st->d_h; b.tw = st->d_tw; b.tw_acc = reinterpret_cast<uint64_t *>(st->d_tw_bucket); b.guess = st->d_guess; b.redo = st->d_redo;
        b.nredo = reinterpret_cast<uint32_t *>(st->d_status + 1);
        const size_t ninit = std::max<size_t>(n * m, n);
        hipLaunchKernelGGL(k3_bmh_init_kernel, dim3((unsigned)div_up<size_t>(ninit, K3_THREADS)), dim3(K3_THREADS), 0, s,
                           st->d_h, n * m, st->d_tw, st->d_redo, n);
        // Workgroups own contiguous bucket ranges; 6 are resident per CU.  A grid of 8 per CU (r01) left a third of the ranges
        // to a second, three-quarters-empty round: 9.0 ms.  With many more, smaller ranges the hardware's dispatch evens the
        // tail out: 8.3 ms at 6 per CU, 7.9 at 12, 7.6 at 24, 7.3 at 48 and 64 (the survivor kernel follows: 1.9 -> 1.7 ms).
        size_t per_cu = 48;
        if (const char *e = ctx->tune.get("D2G_K3_GRID_PER_CU")) { const int v = std::atoi(e); if (v >= 1 && v <= 256) per_cu = (size_t)v; }
        const unsigned main_grid = (unsigned)std::min<size_t>(TB, (size_t)ctx->num_cus * per_cu);
        st->last_nredo = 0;
        // First pass in the light form: survivors go to per-workgroup regions of a queue in HBM.  A region holds twice the
        // survivors its buckets are expected to produce: genome g yields at most gk strips in all (an element of count c has
        // <= c strips) and about gk * guess of them survive, spread evenly over its buckets.
        bool light = TB > 0 && !D2G_K3_EXP;
        if (const char *e = ctx->tune.get("D2G_K3_LIGHT")) if (e[0] == '0') light = false;
        double gq_scale = 2.0;
        uint64_t gq_slack = per_cu > 24 ? 256 : 1024;
        if (const char *e = ctx->tune.get("D2G_K3_GQ_SCALE")) { gq_scale = std::max(0.0, std::atof(e)); gq_slack = 1; }   // tests force the overflow path
        if (light) {
            // batches of read-sized inputs: the bound of a tiny input is above 1 and every element survives -- a queue of
            // 24 bytes per k-mer would buy nothing; such batches keep the heavy form
            double etot = 0.;
            for (size_t g = 0; g < n; ++g) { double gv; std::memcpy(&gv, &guess[g], 8); etot += (double)kh.gk[g] * std::min(1.0, gv); }
            if (etot > 0.125 * (double)kh.total) light = false;
        }
        b.tb0 = 0; b.tb1 = TB; b.g0 = 0;
        if (pipeline && !light) { stage_a(0, n); pipeline = false; }       // heavy first pass: one range
        // per-workgroup survivor regions of one light launch over buckets [t0, t1): offsets relative to the launch's first region
        auto gq_offsets = [&](uint32_t t0, uint32_t t1, unsigned grid, std::vector<uint64_t> &off) {
            const uint32_t per = (t1 - t0 + grid - 1) / grid;
            off.assign((size_t)grid + 1, 0);
            size_t g = 0;
            for (unsigned w = 0; w < grid; ++w) {
                const uint64_t lo = (uint64_t)t0 + (uint64_t)w * per, hi = std::min<uint64_t>(lo + per, t1);
                double e = 0.;
                while (g < n && kh.gtab[n + g] + (1ull << kh.gtab[g]) <= lo) ++g;
                for (size_t gg = g; lo < hi && gg < n && kh.gtab[n + gg] < hi; ++gg) {
                    const uint64_t b0 = kh.gtab[n + gg], B = 1ull << kh.gtab[gg];
                    const uint64_t ov = std::min<uint64_t>(hi, b0 + B) - std::max<uint64_t>(lo, b0);
                    double gv; std::memcpy(&gv, &guess[gg], 8);
                    const double eg = (double)kh.gk[gg] * std::min(1.0, gv);
                    e += eg * (double)ov / (double)B;
                }
                off[w + 1] = off[w] + (uint64_t)(gq_scale * e) + gq_slack;
            }
        };
        // launches of the light first pass: one over everything, or one per range of the pipeline
        struct LightLaunch { uint32_t t0, t1, g0, g1; unsigned grid; size_t off_at, n_at; uint64_t entry0; };
        std::vector<LightLaunch> ll;
        if (light) {
            std::vector<uint64_t> all_off;
            uint64_t entries = 0;
            size_t n_words = 0;
            const size_t nr = sub.size() - 1;
            for (size_t j = 0; j < nr; ++j) {
                LightLaunch L;
                L.g0 = (uint32_t)sub[j]; L.g1 = (uint32_t)sub[j + 1];
                L.t0 = kh.gtab[n + sub[j]]; L.t1 = sub[j + 1] < n ? kh.gtab[n + sub[j + 1]] : TB;
                const size_t cap_grid = std::max<size_t>(1, (size_t)ctx->num_cus * per_cu / (nr > 1 ? 2 : 1));
                L.grid = (unsigned)std::min<size_t>(std::max<uint32_t>(L.t1 - L.t0, 1), cap_grid);
                std::vector<uint64_t> off;
                gq_offsets(L.t0, L.t1, L.grid, off);
                L.off_at = all_off.size(); L.n_at = n_words; L.entry0 = entries;
                all_off.insert(all_off.end(), off.begin(), off.end());
                entries += off.back(); n_words += L.grid;
                ll.push_back(L);
            }
            if (int rc = d2g_grow(ctx, &st->d_gq, &st->cap_gq, (size_t)entries * (sizeof(QEntry) / 8) + 8)) return rc;
            if (int rc = d2g_grow(ctx, &st->d_gq_off, &st->cap_gq_off, all_off.size() + (n_words + 1) / 2 + 1)) return rc;
            D2G_HIP(ctx, hipMemcpyAsync(st->d_gq_off, all_off.data(), all_off.size() * sizeof(uint64_t), hipMemcpyHostToDevice, s));
            for (auto &L : ll) L.n_at += 2 * all_off.size();                 // u32 index into the same buffer, behind the offsets
        }
        auto light_args = [&](const LightLaunch &L) {
            BmhArgs x = b;
            x.tb0 = L.t0; x.tb1 = L.t1; x.g0 = L.g0; x.redo_mode = 0;
            x.gq = reinterpret_cast<QEntry *>(st->d_gq) + L.entry0;
            x.gq_off = st->d_gq_off + L.off_at;
            x.gq_n = reinterpret_cast<uint32_t *>(st->d_gq_off) + L.n_at;
            return x;
        };
        void (*light_k)(BmhArgs) = k3_bmh_main_kernel<false, true>, (*heavy_k)(BmhArgs) = k3_bmh_main_kernel<false, false>;
        if (kh.compact) { light_k = k3_bmh_main_kernel<true, true>; heavy_k = k3_bmh_main_kernel<true, false>; }
        for (int pass = 0;; ++pass) {
            b.redo_mode = pass > 0;
            const bool lt = light && pass == 0;
            const bool run = TB && D2G_K3_EXP != 1 && D2G_K3_EXP != 2 && D2G_K3_EXP != 6 && D2G_K3_EXP != 7;
            if (lt && pipeline) {
                if (!st->xs) {
                    D2G_HIP(ctx, hipStreamCreateWithFlags(&st->xs, hipStreamNonBlocking));
                    for (auto &e : st->ev_a) D2G_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
                    D2G_HIP(ctx, hipEventCreateWithFlags(&st->ev_b, hipEventDisableTiming));
                }
                for (size_t j = 0; j < ll.size(); ++j) {
                    stage_a(ll[j].g0, ll[j].g1);
                    D2G_HIP(ctx, hipEventRecord(st->ev_a[j], s));
                    D2G_HIP(ctx, hipStreamWaitEvent(st->xs, st->ev_a[j], 0));
                    const BmhArgs x = light_args(ll[j]);
                    if (run && ll[j].t1 > ll[j].t0) {
                        hipLaunchKernelGGL(light_k, dim3(ll[j].grid), dim3(K3_THREADS), 0, st->xs, x);
                        hipLaunchKernelGGL(k3_bmh_survivor_kernel, dim3(ll[j].grid), dim3(K3_THREADS), 0, st->xs, x);
                    }
                    hipLaunchKernelGGL(k3_bmh_verify_kernel, dim3(ll[j].g1 - ll[j].g0), dim3(K3_THREADS), 0, st->xs, x);
                }
                D2G_HIP(ctx, hipEventRecord(st->ev_b, st->xs));
                D2G_HIP(ctx, hipStreamWaitEvent(s, st->ev_b, 0));
                st->pipelined_calls++;
            } else {
                if (run) {
                    if (lt) {
                        const BmhArgs x = light_args(ll[0]);
                        hipLaunchKernelGGL(light_k, dim3(ll[0].grid), dim3(K3_THREADS), 0, s, x);
                        hipLaunchKernelGGL(k3_bmh_survivor_kernel, dim3(ll[0].grid), dim3(K3_THREADS), 0, s, x);
                    } else
                        hipLaunchKernelGGL(heavy_k, dim3(main_grid), dim3(K3_THREADS), 0, s, b);
                }
                hipLaunchKernelGGL(k3_bmh_verify_kernel, dim3((unsigned)n), dim3(K3_THREADS), 0, s, b);
            }
            int st2[2] = {0, 0};                                  // [0] kernel status, [1] genomes whose guess failed
            D2G_HIP(ctx, hipMemcpyAsync(st2, st->d_status, sizeof(st2), hipMemcpyDeviceToHost, s));
            D2G_HIP(ctx, hipStreamSynchronize(s));
            if (lt && st2[0] == 4) {
                // a survivor region overflowed (the estimate above is an expectation): the pass again, in the heavy form
                light = false;
                D2G_HIP(ctx, hipMemsetAsync(st->d_status, 0, 2 * sizeof(int), s));
                D2G_HIP(ctx, hipMemsetAsync(st->d_tw_bucket, 0, ((size_t)TB + 1) * sizeof(double), s));
                hipLaunchKernelGGL(k3_bmh_init_kernel, dim3((unsigned)div_up<size_t>(ninit, K3_THREADS)), dim3(K3_THREADS), 0, s,
                                   st->d_h, n * m, st->d_tw, st->d_redo, n);
                st->light_overflows++;
                --pass;
                continue;
            }
            const int nredo = st2[1];
            if (st2[0] || !nredo || D2G_K3_EXP) break;
            st->last_nredo += nredo;
            D2G_CHECK(ctx, pass < 40, "internal: BagMinHash bound did not converge");
            D2G_HIP(ctx, hipMemsetAsync(st->d_status + 1, 0, sizeof(int), s));
        }
    }
    tm.stop();
    D2G_HIP(ctx, hipGetLastError());
    return D2G_OK;
}

int k3_check_status(d2g_ctx *ctx, d2g_k3_state *st, hipStream_t s) {
    int status = 0;
    D2G_HIP(ctx, hipMemcpyAsync(&status, st->d_status, sizeof(int), hipMemcpyDeviceToHost, s));
    D2G_HIP(ctx, hipStreamSynchronize(s));
    if (status == 1) { ctx->last_error = "internal: k-mer count table overflow"; return D2G_ERR_INTERNAL; }
    if (status == 2) { ctx->last_error = "BagMinHash weight outside (0, 2^53]"; return D2G_ERR_INVALID; }
    if (status == 3) { ctx->last_error = "internal: BagMinHash process stack overflow"; return D2G_ERR_INTERNAL; }
    return D2G_OK;
}

d2g_k3_state *k3_state_of(d2g_sketcher *sk) {
    if (!sk->k3) { sk->k3 = new (std::nothrow) d2g_k3_state(); if (sk->k3) sk->k3->ctx = sk->ctx; }
    return sk->k3;
}

}  // namespace

extern "C" {

int d2g_sketcher_run_bmh(d2g_sketcher *sk, const uint8_t *packed, size_t packed_bytes, const uint64_t *run_start,
                         const uint32_t *run_len, size_t nrun, const uint64_t *genome_run_off, size_t n, int k, int canon,
                         uint64_t xormask, size_t sketchsize, double count_threshold, double *sig_out,
                         double *total_weight_out) {
    if (!sk) return D2G_ERR_INVALID;
    d2g_ctx *ctx = sk->ctx;
    D2G_CHECK(ctx, sketchsize >= 1 && sketchsize < (1ull << 24), "sketchsize out of range");
    D2G_CHECK(ctx, (sig_out && total_weight_out) || n == 0, "null output");
    D2G_CHECK(ctx, count_threshold == count_threshold, "count_threshold is NaN");
    d2g_k3_state *st = k3_state_of(sk);
    if (!st) return D2G_ERR_NOMEM;
    KmerArgs km;
    size_t nblk = 0;
    if (int rc = d2g_sketcher_stage(sk, packed, packed_bytes, run_start, run_len, nrun, genome_run_off, n, k, canon, &km,
                                    &nblk, nullptr)) return rc;
    K3Host kh;
    if (int rc = k3_layout(ctx, run_len, genome_run_off, n, k, kh)) return rc;
    if (n == 0) return D2G_OK;
    hipStream_t s = sk->stream;
    if (int rc = k3_run(ctx, st, s, km, nblk, kh, n, xormask, sketchsize, count_threshold, false)) return rc;
    D2G_HIP(ctx, hipMemcpyAsync(sig_out, st->d_h, n * sketchsize * sizeof(double), hipMemcpyDeviceToHost, s));
    D2G_HIP(ctx, hipMemcpyAsync(total_weight_out, st->d_tw, n * sizeof(double), hipMemcpyDeviceToHost, s));
    return k3_check_status(ctx, st, s);
}

int d2g_bmh_sketch_dev(d2g_ctx *ctx, const d2g_oph_plan *plan, const uint8_t *packed_dev, int canon, uint64_t xormask,
                       size_t sketchsize, double count_threshold, double *sig_out_dev, double *total_weight_out_dev,
                       void *stream) {
    if (!ctx || !plan) return D2G_ERR_INVALID;
    D2G_CHECK(ctx, plan->ctx == ctx, "plan belongs to another context");
    D2G_CHECK(ctx, sketchsize >= 1 && sketchsize < (1ull << 24), "sketchsize out of range");
    D2G_CHECK(ctx, (sig_out_dev && total_weight_out_dev) || plan->n == 0, "null output");
    D2G_CHECK(ctx, count_threshold == count_threshold, "count_threshold is NaN");
    D2G_CHECK(ctx, ((uintptr_t)packed_dev & 3) == 0, "packed stream must be 4-byte aligned");
    D2G_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->k3) { ctx->k3 = new (std::nothrow) d2g_k3_state(); if (!ctx->k3) return D2G_ERR_NOMEM; ctx->k3->ctx = ctx; }
    d2g_k3_state *st = ctx->k3;
    const size_t n = plan->n;
    K3Host kh;
    if (int rc = k3_layout(ctx, plan->h_run_len.data(), plan->h_genome_run_off.data(), n, plan->k, kh)) return rc;
    if (n == 0) return D2G_OK;
    hipStream_t s = as_stream(stream);
    if (int rc = k3_run(ctx, st, s, d2g_plan_args(plan, packed_dev, canon), plan->nblk, kh, n, xormask, sketchsize,
                        count_threshold, false)) return rc;
    D2G_HIP(ctx, hipMemcpyAsync(sig_out_dev, st->d_h, n * sketchsize * sizeof(double), hipMemcpyDeviceToDevice, s));
    D2G_HIP(ctx, hipMemcpyAsync(total_weight_out_dev, st->d_tw, n * sizeof(double), hipMemcpyDeviceToDevice, s));
    return k3_check_status(ctx, st, s);
}

int d2g_bmh_sketch(d2g_ctx *ctx, const uint8_t *packed, size_t packed_bytes, const uint64_t *run_start,
                   const uint32_t *run_len, size_t nrun, const uint64_t *genome_run_off, size_t n, int k, int canon,
                   uint64_t xormask, size_t sketchsize, double count_threshold, double *sig_out, double *total_weight_out) {
    if (!ctx) return D2G_ERR_INVALID;
    d2g_sketcher *sk = nullptr;
    if (int rc = d2g_sketcher_create(ctx, &sk)) return rc;
    const int rc = d2g_sketcher_run_bmh(sk, packed, packed_bytes, run_start, run_len, nrun, genome_run_off, n, k, canon,
                                        xormask, sketchsize, count_threshold, sig_out, total_weight_out);
    d2g_sketcher_destroy(sk);
    return rc;
}

int d2g_sketcher_run_distinct(d2g_sketcher *sk, const uint8_t *packed, size_t packed_bytes, const uint64_t *run_start,
                              const uint32_t *run_len, size_t nrun, const uint64_t *genome_run_off, size_t n, int k, int canon,
                              uint64_t xormask, uint64_t *ndistinct_out) {
    if (!sk) return D2G_ERR_INVALID;
    d2g_ctx *ctx = sk->ctx;
    D2G_CHECK(ctx, ndistinct_out != nullptr || n == 0, "null output");
    d2g_k3_state *st = k3_state_of(sk);
    if (!st) return D2G_ERR_NOMEM;
    KmerArgs km;
    size_t nblk = 0;
    if (int rc = d2g_sketcher_stage(sk, packed, packed_bytes, run_start, run_len, nrun, genome_run_off, n, k, canon, &km,
                                    &nblk, nullptr)) return rc;
    K3Host kh;
    if (int rc = k3_layout(ctx, run_len, genome_run_off, n, k, kh)) return rc;
    if (n == 0) return D2G_OK;
    hipStream_t s = sk->stream;
    if (int rc = k3_run(ctx, st, s, km, nblk, kh, n, xormask, 1, 0.0, true, true)) return rc;
    if (int rc = k3_check_status(ctx, st, s)) return rc;
    std::vector<uint32_t> nd(kh.TB);
    D2G_HIP(ctx, hipMemcpyAsync(nd.data(), st->d_bucket_nd, kh.TB * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    D2G_HIP(ctx, hipStreamSynchronize(s));
    for (size_t g = 0; g < n; ++g) {
        uint64_t t = 0;
        for (uint32_t tb = kh.gtab[n + g]; tb < kh.gtab[n + g + 1]; ++tb) t += nd[tb];
        ndistinct_out[g] = t;
    }
    return D2G_OK;
}

int d2g_kmer_distinct(d2g_ctx *ctx, const uint8_t *packed, size_t packed_bytes, const uint64_t *run_start,
                      const uint32_t *run_len, size_t nrun, const uint64_t *genome_run_off, size_t n, int k, int canon,
                      uint64_t xormask, uint64_t *ndistinct_out) {
    if (!ctx) return D2G_ERR_INVALID;
    d2g_sketcher *sk = nullptr;
    if (int rc = d2g_sketcher_create(ctx, &sk)) return rc;
    const int rc = d2g_sketcher_run_distinct(sk, packed, packed_bytes, run_start, run_len, nrun, genome_run_off, n, k, canon,
                                             xormask, ndistinct_out);
    d2g_sketcher_destroy(sk);
    return rc;
}

int d2g_kmer_count(d2g_ctx *ctx, const uint8_t *packed, size_t packed_bytes, const uint64_t *run_start,
                   const uint32_t *run_len, size_t nrun, const uint64_t *genome_run_off, size_t n, int k, int canon,
                   uint64_t xormask, double count_threshold, uint64_t *keys_out, uint32_t *counts_out, size_t cap,
                   uint64_t *genome_off_out) {
    if (!ctx) return D2G_ERR_INVALID;
    D2G_CHECK(ctx, genome_off_out != nullptr, "null genome_off_out");
    D2G_CHECK(ctx, cap == 0 || (keys_out && counts_out), "null output");
    d2g_sketcher *sk = nullptr;
    if (int rc = d2g_sketcher_create(ctx, &sk)) return rc;
    int rc = D2G_OK;
    do {
        d2g_k3_state *st = k3_state_of(sk);
        if (!st) { rc = D2G_ERR_NOMEM; break; }
        KmerArgs km;
        size_t nblk = 0;
        if ((rc = d2g_sketcher_stage(sk, packed, packed_bytes, run_start, run_len, nrun, genome_run_off, n, k, canon, &km,
                                     &nblk, nullptr))) break;
        K3Host kh;
        if ((rc = k3_layout(ctx, run_len, genome_run_off, n, k, kh))) break;
        for (size_t g = 0; g <= n; ++g) genome_off_out[g] = 0;
        if (n == 0) break;
        hipStream_t s = sk->stream;
        if ((rc = k3_run(ctx, st, s, km, nblk, kh, n, xormask, 1, count_threshold, true))) break;
        if ((rc = k3_check_status(ctx, st, s))) break;
        // compact the per-bucket prefixes on the host (utility entry point, not the sketch path)
        std::vector<uint32_t> nd(kh.TB);
        std::vector<uint64_t> boff((size_t)kh.TB + 1);
        hipError_t e;
        if ((e = hipMemcpy(nd.data(), st->d_bucket_nd, kh.TB * sizeof(uint32_t), hipMemcpyDeviceToHost)) != hipSuccess ||
            (e = hipMemcpy(boff.data(), st->d_bucket_off, ((size_t)kh.TB + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost)) != hipSuccess) {
            ctx->last_error = hipGetErrorString(e); rc = D2G_ERR_HIP; break;
        }
        std::vector<uint64_t> hk(std::max<uint64_t>(kh.total, 1));
        std::vector<uint32_t> hc(std::max<uint64_t>(kh.total, 1));
        if (kh.total &&
            ((e = hipMemcpy(hk.data(), st->d_out_keys, kh.total * sizeof(uint64_t), hipMemcpyDeviceToHost)) != hipSuccess ||
             (e = hipMemcpy(hc.data(), st->d_out_counts, kh.total * sizeof(uint32_t), hipMemcpyDeviceToHost)) != hipSuccess)) {
            ctx->last_error = hipGetErrorString(e); rc = D2G_ERR_HIP; break;
        }
        size_t w = 0;
        for (size_t g = 0; g < n && rc == D2G_OK; ++g) {
            genome_off_out[g] = w;
            for (uint32_t tb = kh.gtab[n + g]; tb < kh.gtab[n + g + 1]; ++tb) {
                if (w + nd[tb] > cap) { ctx->last_error = "d2g_kmer_count: output capacity too small"; rc = D2G_ERR_INVALID; break; }
                std::memcpy(keys_out + w, hk.data() + boff[tb], nd[tb] * sizeof(uint64_t));
                std::memcpy(counts_out + w, hc.data() + boff[tb], nd[tb] * sizeof(uint32_t));
                w += nd[tb];
            }
        }
        genome_off_out[n] = w;
    } while (0);
    d2g_sketcher_destroy(sk);
    return rc;
}

int d2g_bmh_from_weighted(d2g_ctx *ctx, const uint64_t *ids, const double *weights, const uint64_t *set_off, size_t nsets,
                          size_t sketchsize, double *sig_out, double *total_weight_out) {
    return d2g_bmh_from_weighted_ids(ctx, ids, weights, set_off, nsets, sketchsize, sig_out, total_weight_out, nullptr);
}

int d2g_bmh_from_weighted_ids(d2g_ctx *ctx, const uint64_t *ids, const double *weights, const uint64_t *set_off, size_t nsets,
                              size_t sketchsize, double *sig_out, double *total_weight_out, uint64_t *owner_out) {
    if (!ctx) return D2G_ERR_INVALID;
    D2G_CHECK(ctx, set_off != nullptr && (nsets == 0 || (sig_out && total_weight_out)), "null argument");
    D2G_CHECK(ctx, sketchsize >= 1 && sketchsize < (1ull << 24), "sketchsize out of range");
    D2G_CHECK(ctx, nsets < (1ull << 31), "too many sets");
    if (nsets == 0) return D2G_OK;
    const uint64_t total = set_off[nsets];
    D2G_CHECK(ctx, total == 0 || ids != nullptr, "null ids");
    for (size_t i = 0; i < nsets; ++i) D2G_CHECK(ctx, set_off[i] <= set_off[i + 1], "set_off not monotone");
    D2G_HIP(ctx, hipSetDevice(ctx->device));
    const size_t m = sketchsize;
    // total weights on the host (the caller's arrays are host arrays): the result, and the first guess of the bound
    const uint32_t chunk = 2048;
    std::vector<uint32_t> bset, bcnt;
    std::vector<uint64_t> blo, guess(nsets);
    std::vector<double> tw(nsets, 0.);
    const double lnm = std::log((double)m);
    double scale = 1.0;
    if (const char *e = ctx->tune.get("D2G_K3_GUESS_SCALE")) { const double v = std::atof(e); if (v > 0.) scale = v; }
    for (size_t i = 0; i < nsets; ++i) {
        const uint64_t lo = set_off[i], hi = set_off[i + 1];
        double t = 0.;
        for (uint64_t e = lo; e < hi; ++e) {
            const double w = weights ? weights[e] : 1.0;
            if (w > 0.) {
                if (!(w <= 0x1p53)) { ctx->last_error = "BagMinHash weight outside (0, 2^53]"; return D2G_ERR_INVALID; }
                t += w;
            } else if (w != w) { ctx->last_error = "BagMinHash weight is NaN"; return D2G_ERR_INVALID; }
        }
        tw[i] = t;
        const double gv = t > 0. ? scale * bmh_guess(t, (double)m, lnm) : 0.;
        std::memcpy(&guess[i], &gv, 8);
        for (uint64_t e = lo; e < hi; e += chunk) {
            bset.push_back((uint32_t)i); blo.push_back(e); bcnt.push_back((uint32_t)std::min<uint64_t>(chunk, hi - e));
        }
    }
    D2G_CHECK(ctx, bset.size() < (1ull << 31), "too many workgroups");
    const size_t nb = bset.size();
    uint64_t *d_ids = nullptr, *d_blo = nullptr, *d_h = nullptr, *d_guess = nullptr, *d_arg = nullptr, *d_setlo = nullptr;
    double *d_w = nullptr, *d_tw = nullptr;
    uint32_t *d_bset = nullptr, *d_bcnt = nullptr, *d_redo = nullptr;
    int *d_status = nullptr;
    int rc = D2G_OK;
    auto cleanup = [&]() {
        (void)hipFree(d_ids); (void)hipFree(d_blo); (void)hipFree(d_h); (void)hipFree(d_guess); (void)hipFree(d_redo);
        (void)hipFree(d_w); (void)hipFree(d_tw); (void)hipFree(d_bset); (void)hipFree(d_bcnt); (void)hipFree(d_status);
        (void)hipFree(d_arg); (void)hipFree(d_setlo);
    };
#define K3_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { ctx->last_error = hipGetErrorString(e_); cleanup(); return D2G_ERR_HIP; } } while (0)
    K3_TRY(hipMalloc((void **)&d_ids, std::max<uint64_t>(total, 1) * 8));
    K3_TRY(hipMalloc((void **)&d_blo, std::max<size_t>(nb, 1) * 8));
    K3_TRY(hipMalloc((void **)&d_bset, std::max<size_t>(nb, 1) * 4));
    K3_TRY(hipMalloc((void **)&d_bcnt, std::max<size_t>(nb, 1) * 4));
    K3_TRY(hipMalloc((void **)&d_h, nsets * m * 8));
    K3_TRY(hipMalloc((void **)&d_guess, nsets * 8));
    K3_TRY(hipMalloc((void **)&d_redo, nsets * 4));
    K3_TRY(hipMalloc((void **)&d_tw, nsets * 8));
    K3_TRY(hipMalloc((void **)&d_status, 2 * sizeof(int)));
    if (weights) { K3_TRY(hipMalloc((void **)&d_w, std::max<uint64_t>(total, 1) * 8)); K3_TRY(hipMemcpy(d_w, weights, total * 8, hipMemcpyHostToDevice)); }
    if (total) K3_TRY(hipMemcpy(d_ids, ids, total * 8, hipMemcpyHostToDevice));
    K3_TRY(hipMemcpy(d_guess, guess.data(), nsets * 8, hipMemcpyHostToDevice));
    if (nb) {
        K3_TRY(hipMemcpy(d_blo, blo.data(), nb * 8, hipMemcpyHostToDevice));
        K3_TRY(hipMemcpy(d_bset, bset.data(), nb * 4, hipMemcpyHostToDevice));
        K3_TRY(hipMemcpy(d_bcnt, bcnt.data(), nb * 4, hipMemcpyHostToDevice));
    }
    K3_TRY(hipMemset(d_status, 0, 2 * sizeof(int)));
    WsArgs a;
    a.ids = d_ids; a.w = d_w; a.blk_set = d_bset; a.blk_lo = d_blo; a.blk_cnt = d_bcnt;
    a.m = (uint32_t)m; a.h = d_h; a.guess = d_guess; a.redo = d_redo; a.redo_mode = 0; a.status = d_status;
    a.arg = nullptr; a.set_lo = nullptr;
    {
        d2g_timer tm(ctx, &ctx->ev_k3, nullptr);
        const size_t ninit = std::max<size_t>(nsets * m, nsets);
        hipLaunchKernelGGL(k3_bmh_init_kernel, dim3((unsigned)div_up<size_t>(ninit, K3_THREADS)), dim3(K3_THREADS), 0, nullptr,
                           d_h, nsets * m, d_tw, d_redo, nsets);
        K3_TRY(hipMemcpy(d_tw, tw.data(), nsets * 8, hipMemcpyHostToDevice));
        for (int pass = 0;; ++pass) {
            a.redo_mode = pass > 0;
            if (nb) hipLaunchKernelGGL(k3_bmh_sets_kernel, dim3((unsigned)nb), dim3(K3_THREADS), 0, nullptr, a);
            hipLaunchKernelGGL(k3_sets_verify_kernel, dim3((unsigned)nsets), dim3(K3_THREADS), 0, nullptr, d_h, (uint32_t)m, d_guess, d_redo,
                               d_tw, reinterpret_cast<uint32_t *>(d_status + 1), a.redo_mode);
            int st2[2] = {0, 0};
            K3_TRY(hipMemcpy(st2, d_status, sizeof(st2), hipMemcpyDeviceToHost));
            if (st2[0] || !st2[1]) break;
            if (pass >= 40) { ctx->last_error = "internal: BagMinHash bound did not converge"; cleanup(); return D2G_ERR_INTERNAL; }
            K3_TRY(hipMemset(d_status + 1, 0, sizeof(int)));
        }
        if (owner_out) {
            K3_TRY(hipMalloc((void **)&d_arg, nsets * m * 8));
            K3_TRY(hipMalloc((void **)&d_setlo, nsets * 8));
            K3_TRY(hipMemset(d_arg, 0xFF, nsets * m * 8));
            K3_TRY(hipMemcpy(d_setlo, set_off, nsets * 8, hipMemcpyHostToDevice));
            a.arg = d_arg; a.set_lo = d_setlo;
            if (nb) hipLaunchKernelGGL(k3_bmh_sets_argmin_kernel, dim3((unsigned)nb), dim3(K3_THREADS), 0, nullptr, a);
        }
        tm.stop();
    }
    K3_TRY(hipGetLastError());
    int status = 0;
    K3_TRY(hipMemcpy(&status, d_status, sizeof(int), hipMemcpyDeviceToHost));
    K3_TRY(hipMemcpy(sig_out, d_h, nsets * m * 8, hipMemcpyDeviceToHost));
    if (owner_out) K3_TRY(hipMemcpy(owner_out, d_arg, nsets * m * 8, hipMemcpyDeviceToHost));
    std::memcpy(total_weight_out, tw.data(), nsets * sizeof(double));
#undef K3_TRY
    cleanup();
    if (status == 2) { ctx->last_error = "BagMinHash weight outside (0, 2^53]"; rc = D2G_ERR_INVALID; }
    if (status == 3) { ctx->last_error = "internal: BagMinHash process stack overflow"; rc = D2G_ERR_INTERNAL; }
    return rc;
}

}  // extern "C"

void d2g_warm_k3() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k3_scan_kernel)); }

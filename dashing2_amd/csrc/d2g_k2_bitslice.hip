// d2g_k2_bitslice.hip -- K2, BITSLICE algorithm (gfx950): exact equality counts from bit planes.
//
// Same contract as the direct kernel (reference src/cmp_core.cpp:461,506 count_gtlt/count_eq
// inside src/emitrect.cpp:211-323), different arithmetic:
//
//   the count only needs EQUALITY of 64-bit patterns within one register index t, so each
//   column t of the N x S matrix is first mapped to small ids: a value that occurs >= 2 times in its
//   column gets a dense rank 1..D2; a value that occurs once can never equal anything ("unique").
//   The ids are stored bit-sliced, in TWO codings of the unique values: the ROW operand codes them
//   as 0, the COLUMN operand as all-ones (2^nbits - 1 > D2), so a unique value mismatches whatever
//   it meets without a separate test.  Word P[tb][b][j] holds bit b of the ids of sketch j for the
//   32 registers t = 32*tb .. 32*tb+31.  For a pair (i,j) and a 32-register group
//       z = OR_b ( R[tb][b][i] XOR C[tb][b][j] )          one v_bitop3_b32 per plane
//       mismatches += popcount(z)                          one v_bcnt_u32_b32 (accumulating)
//   i.e. (nbits+1)/32 VALU operations per register compare instead of 2 (v_cmp_eq_u64+v_addc):
//   0.25 at 7 planes.  Row operands come through SCALAR loads (one s_load_dwordx16 per plane
//   serves 16 rows), column operands are one coalesced dword per lane.
//
// Prepare = 3 small kernels (timed as "k2prep"): transpose, per-column open-addressing insert into an
// LDS owner table + compaction to dense ranks, 32 x nbits bit transpose into both codings.
#include "d2g_internal.h"
#include "d2g_k2.h"
#include "d2g_k2_shape.h"
#include <algorithm>
#include <cstdlib>
#include <vector>

namespace {

constexpr uint32_t BS_EMPTY = 0xFFFFFFFFu;
constexpr int BS_RANK_THREADS = 1024;

// ------------------------------------------------------------------ 1. per-column dense ids
// One workgroup per register index t.  owner[] (T slots, pre-set to EMPTY) records the first
// sketch index that claimed a slot; equality is decided against that sketch's value, so no key
// storage and no reserved sentinel value is needed.
constexpr uint32_t BS_DUP = 0x80000000u;       // owner-table flag: the value has been seen again
constexpr uint32_t BS_UNIQ = 0x80000000u;      // id flag: value occurs once in its column (never equal)
constexpr int BS_LOG_TLDS_MAX = 15;            // LDS owner table: at most 32768 slots = 128 KiB

__device__ __forceinline__ uint32_t bs_hash(uint64_t v, int logT) {
    // Fibonacci hashing: the top logT bits of the product (partition = top bits, slot = low bits of those)
    return (uint32_t)((v * 0x9E3779B97F4A7C15ull) >> (64 - logT));
}

// Singleton folding: a value that occurs exactly once in its register column can never compare
// equal to anything, so all such values share id 0 and set the "unique" bit instead; only values
// occurring >= 2 times get dense ids 1..D2.  meta[t/32] = max over the group's columns of D2 + 1.
//
// One workgroup per register index t.  The T-slot hash space is walked in P = T / Tl partitions
// (top hash bits); each pass inserts the values of one partition into a Tl-slot LDS table of
// *owner sketch indices* (equality is decided against the owner's value: no key storage, no
// reserved sentinel), compacts the slots whose value was seen again into dense ranks, and
// writes the ids of that partition.  T >= 1.5 N, so a partition holds <= 2/3 Tl values on average.
template <bool MULTI>   // MULTI: more than one partition (N > 21845)
__global__ __launch_bounds__(BS_RANK_THREADS) void bs_rank_kernel(const uint64_t *__restrict__ cols, size_t N, size_t Npad,
                                                                  uint32_t T, int logT, uint32_t *ids_all, uint32_t *max_distinct,
                                                                  uint32_t *status) {
    extern __shared__ __attribute__((aligned(16))) uint32_t own[];        // Tl owner slots
    const size_t t = blockIdx.x;
    const uint64_t *col = cols + t * Npad;
    uint32_t *ids = ids_all + t * Npad;
    const int tid = threadIdx.x;
    const int logTl = logT < BS_LOG_TLDS_MAX ? logT : BS_LOG_TLDS_MAX;
    const uint32_t Tl = 1u << logTl, mask = Tl - 1, nparts = MULTI ? (T >> logTl) : 1u;
    __shared__ uint32_t wave_tot[BS_RANK_THREADS / 64];
    __shared__ uint32_t running;
    if (tid == 0) running = 1;                                           // id 0 is reserved for singletons
    const int lane = tid & 63, wave = tid >> 6;

    // the compaction of one table pass: slots whose value occurs >= 2 times get the next dense ranks,
    // singletons BS_UNIQ
    auto compact = [&]() {
        // one block-wide prefix over per-thread counts (a thread owns slots tid, tid + 1024, ...): three
        // barriers per pass instead of three per 1024 slots; any bijection onto 1..#dups is a valid ranking
        uint32_t cnt = 0;
        for (uint32_t h = tid; h < Tl; h += BS_RANK_THREADS) {
            const uint32_t cur = own[h];
            cnt += (cur != BS_EMPTY) && (cur & BS_DUP);
        }
        uint32_t incl = cnt;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t x = __shfl_up(incl, o); if (lane >= o) incl += x; }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t woff = 0, tot = 0;
        for (int w = 0; w < BS_RANK_THREADS / 64; ++w) {
            const uint32_t x = wave_tot[w];
            if (w < wave) woff += x;
            tot += x;
        }
        uint32_t r = running + woff + (incl - cnt);
        for (uint32_t h = tid; h < Tl; h += BS_RANK_THREADS) {
            const uint32_t cur = own[h];
            if (cur != BS_EMPTY) own[h] = (cur & BS_DUP) ? r++ : BS_UNIQ;
        }
        __syncthreads();
        if (tid == 0) running += tot;
        __syncthreads();
    };
    // at most Tl probes: a partition that receives more than Tl distinct values (a skewed / adversarial
    // column; T >= 1.5 N only bounds the AVERAGE load) must not spin forever.  The overflow is
    // reported through *status; the host then falls back to the DIRECT algorithm or fails loudly.
    auto insert = [&](uint64_t v, uint32_t j, uint32_t h) {
        for (uint32_t probes = 0; probes < Tl; ++probes) {
            const uint32_t cur = atomicCAS(&own[h], BS_EMPTY, j);
            if (cur == BS_EMPTY) return h;                                // first occurrence: we own the slot
            if (col[cur & ~BS_DUP] == v) {                                // same value seen again
                if (!(cur & BS_DUP)) atomicOr(&own[h], BS_DUP);
                return h;
            }
            h = (h + 1) & mask;
        }
        atomicOr(status, 1u);
        return h;                                                         // garbage id, flagged
    };

    constexpr int PF = 12;      // values a thread keeps in registers (fast path: N <= 12288, one partition)
    if (!MULTI && N <= (size_t)PF * BS_RANK_THREADS) {
        // every value is fetched BEFORE the probe chains (a load inside the chain exposed a full
        // HBM/L2 round trip per value: 76 us for the 1024 columns of config 3) and its slot stays in
        // a register until the ids are written
        uint64_t v[PF];
        uint32_t hs[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const size_t j = (size_t)i * BS_RANK_THREADS + tid;
            v[i] = j < N ? col[j] : 0;
        }
        for (uint32_t h = tid; h < Tl; h += BS_RANK_THREADS) own[h] = BS_EMPTY;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const size_t j = (size_t)i * BS_RANK_THREADS + tid;
            hs[i] = j < N ? insert(v[i], (uint32_t)j, bs_hash(v[i], logT) & mask) : 0u;
        }
        __syncthreads();
        compact();
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const size_t j = (size_t)i * BS_RANK_THREADS + tid;
            if (j < N) ids[j] = own[hs[i]];
        }
        if (tid == 0) atomicMax(&max_distinct[t >> 5], running);
        return;
    }

    for (uint32_t part = 0; part < nparts; ++part) {
        for (uint32_t h = tid; h < Tl; h += BS_RANK_THREADS) own[h] = BS_EMPTY;
        __syncthreads();
        for (size_t j = tid; j < N; j += BS_RANK_THREADS) {
            const uint64_t v = col[j];
            const uint32_t hh = bs_hash(v, logT);
            if (MULTI && (hh >> logTl) != part) continue;
            ids[j] = insert(v, (uint32_t)j, hh & mask);
        }
        __syncthreads();
        compact();
        for (size_t j = tid; j < N; j += BS_RANK_THREADS) {
            if (MULTI && (bs_hash(col[j], logT) >> logTl) != part) continue;
            ids[j] = own[ids[j]];
        }
        __syncthreads();
    }
    if (tid == 0) atomicMax(&max_distinct[t >> 5], running);              // per 32-register group
}

// ------------------------------------------------------------------ 2. 32 x nbits bit transpose
// thread (tb, j): reads the ids of sketch j for 32 consecutive registers, writes nbits words.
__device__ __forceinline__ int live_planes(const uint32_t *meta, int tb) {
    const uint32_t md = meta[tb];                      // max over the group's columns of (#shared values D2) + 1
    return md <= 1 ? 1 : 32 - __clz(md);              // smallest nb with 2^nb >= D2 + 2: ranks 1..D2, 0 and 2^nb-1 all distinct
}

__global__ __launch_bounds__(256) void bs_planes_kernel(const uint32_t *__restrict__ ids, size_t S, size_t N, size_t Npad,
                                                        uint32_t *__restrict__ planes, uint32_t *__restrict__ cplanes,
                                                        size_t Nstride, int nbits_cap, const uint32_t *__restrict__ meta) {
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t tb = blockIdx.y;
    if (j >= Nstride) return;
    const int nbits = live_planes(meta, (int)tb);
    uint32_t id[32];
#pragma unroll
    for (int x = 0; x < 32; ++x) {
        const size_t t = tb * 32 + x;
        id[x] = (t < S && j < N) ? ids[t * Npad + j] : 0u;     // padded registers/sketches: id 0 in both codings
    }
    uint32_t u = 0;                                    // the "unique" plane lives in slot nbits_cap of the row operand
#pragma unroll
    for (int x = 0; x < 32; ++x) u |= (id[x] >> 31) << x;
    uint32_t *dst = planes + tb * (size_t)(nbits_cap + 1) * Nstride + j;
    uint32_t *cdst = cplanes + tb * (size_t)nbits_cap * Nstride + j;
    for (int b = 0; b < nbits; ++b) {
        uint32_t w = 0;
#pragma unroll
        for (int x = 0; x < 32; ++x) w |= ((id[x] >> b) & 1u) << x;
        dst[(size_t)b * Nstride] = w;                  // row coding: unique = 0 (BS_UNIQ ids have zero low bits)
        cdst[(size_t)b * Nstride] = w | u;             // column coding: unique = all ones
    }
    dst[(size_t)nbits_cap * Nstride] = u;
}

// column coding of an operand that arrived as (row-coded id planes + unique plane): the gathered operand of
// the multi-GPU path (d2g_cmp_set_from_planes_dev)
__global__ __launch_bounds__(256) void bs_derive_kernel(const uint32_t *__restrict__ planes, uint32_t *__restrict__ cplanes,
                                                        size_t Nstride, int nbits_cap, const uint32_t *__restrict__ meta) {
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t tb = blockIdx.y;
    if (j >= Nstride) return;
    const int nbits = live_planes(meta, (int)tb);
    const uint32_t *src = planes + tb * (size_t)(nbits_cap + 1) * Nstride + j;
    uint32_t *cdst = cplanes + tb * (size_t)nbits_cap * Nstride + j;
    const uint32_t u = src[(size_t)nbits_cap * Nstride];
    for (int b = 0; b < nbits; ++b) cdst[(size_t)b * Nstride] = src[(size_t)b * Nstride] | u;
}

// ------------------------------------------------------------------ 3. the pair kernel
constexpr int BS_THREADS = 256;
constexpr int BS_CB = 256;                // columns per workgroup tile (all variants)

typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
typedef u32x16 __attribute__((aligned(4))) u32x16_u;
// v_bitop3_b32 truth table: src0 = 0xF0, src1 = 0xCC, src2 = 0xAA
constexpr unsigned BITOP3_C_OR_A_XOR_B = 0xAA | (0xF0 ^ 0xCC);   // mismatch accumulation

// IW = 16 rows per wave (one s_load_dwordx16 per plane), JR = 64-column groups per lane,
// WC = waves side by side along the columns (WC * JR * 64 = 256).  Per 32-register group: plane 0
// initialises z = r ^ c (no zeroing), planes 1.. accumulate with v_bitop3 z |= r ^ c, and one accumulating
// v_bcnt finishes the group: nbits + 1 VALU operations per pair and group.  Plane operands are addressed
// as uniform plane pointer (SGPR pair, advanced by SALU) + per-lane 32-bit offset.
//
// The operands of plane p+1 -- or of the next group's plane 0 -- are requested before plane p is
// computed and land in the other of two explicitly alternating register sets (see bs_group).
//
// Epilogue: interior tiles (every pair of the wave's 16 x 64*JR block is wanted and off the diagonal --
// all but the ones on the triangle's edge) take a branch-free path: the row's output base is a scalar,
// the lane adds its column, so an output costs one table gather and one store.
constexpr int BS_IW = 16;

template <int JR>
struct BsOperands {                         // the prefetched operands of one plane
    u32x16_u sa;                            // 16 row words (SGPRs)
    uint32_t vb[JR];                        // this lane's column words
};

template <int JR>
__device__ __forceinline__ BsOperands<JR> bs_fetch(const uint32_t *rp, const uint32_t *cp, uint32_t row, uint32_t colb) {
    BsOperands<JR> o;
    o.sa = *reinterpret_cast<const u32x16_u *>(rp + row);                       // s_load_dwordx16
#pragma unroll
    for (int c = 0; c < JR; ++c)
        o.vb[c] = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(cp) + colb + 256 * c);
    return o;
}

template <int JR, bool FIRST>
__device__ __forceinline__ void bs_plane(const BsOperands<JR> &o, uint32_t (&z)[BS_IW][JR]) {
#pragma unroll
    for (int i = 0; i < BS_IW; ++i)
#pragma unroll
        for (int c = 0; c < JR; ++c)
            z[i][c] = FIRST ? (o.sa[i] ^ o.vb[c]) : __builtin_amdgcn_bitop3_b32(o.sa[i], o.vb[c], z[i][c], BITOP3_C_OR_A_XOR_B);
}

// one 32-register group with `nbits` id planes.  `a` holds plane 0 of this group on entry and plane 0 of the
// next group (rnext/cnext) on exit.  Two operand sets alternate (a, b): the fetch of plane p+1 is issued
// before plane p is computed; a copy (8 s_mov_b64 + JR v_mov) happens at most once per GROUP, when the plane
// count is odd -- a rolled loop with one "next" set copied it once per PLANE.
template <int JR>
__device__ __forceinline__ void bs_group(int nbits, const uint32_t *rbase, const uint32_t *cbase, const uint32_t *rnext,
                                         const uint32_t *cnext, size_t Nstride, uint32_t row, uint32_t colb, BsOperands<JR> &a,
                                         uint32_t (&acc)[BS_IW][JR]) {
    uint32_t z[BS_IW][JR];
    BsOperands<JR> b;
    // Scalar loads return out of order, so the only wait there is for them is lgkmcnt(0): it must come
    // BEFORE the next s_load is issued -- left to the compiler it lands at the first use of the current
    // operands, after the prefetch was issued, and the prefetch is then waited for on the spot.
    auto fetch = [&](int p) {                        // plane p of this group, or plane 0 of the next group
        const bool more = p < nbits;
        const uint32_t *rp = more ? rbase + (size_t)p * Nstride : rnext, *cp = more ? cbase + (size_t)p * Nstride : cnext;
        __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): the operands about to be used have arrived
        __builtin_amdgcn_sched_barrier(0);
        return bs_fetch<JR>(rp, cp, row, colb);
    };
    b = fetch(1);
    bs_plane<JR, true>(a, z);                        // plane 0
    int p = 1;
    for (;;) {
        if (p >= nbits) { a = b; break; }            // b holds the next group's plane 0
        a = fetch(p + 1);
        bs_plane<JR, false>(b, z);                   // plane p (odd)
        if (++p >= nbits) break;                     // a holds the next group's plane 0
        b = fetch(p + 1);
        bs_plane<JR, false>(a, z);                   // plane p (even)
        ++p;
    }
#pragma unroll
    for (int i = 0; i < BS_IW; ++i)
#pragma unroll
        for (int c = 0; c < JR; ++c) acc[i][c] += __builtin_popcount(z[i][c]);     // v_bcnt_u32_b32 acc, z, acc
}

template <int JR, class Store>
__global__ __launch_bounds__(BS_THREADS) __attribute__((amdgpu_waves_per_eu(6))) void k2_bitslice_kernel(
    const uint32_t *__restrict__ planes, const uint32_t *__restrict__ cplanes, size_t Nstride, int nbits_cap,
    const uint32_t *__restrict__ meta, int ntb, uint32_t S, PairShape sh, Store store) {
    constexpr int IW = BS_IW;
    constexpr int WC = BS_CB / (64 * JR);          // waves along columns: 2 (JR=2)
    constexpr int WR = 4 / WC;                     // waves along rows
    constexpr int RB = WR * IW;                    // rows per workgroup tile
    unsigned ct, rt;
    if (!tile_of_block(sh, blockIdx.x, ct, rt)) return;      // XCD-balanced, column-major wanted tiles
    const size_t i0 = sh.i_lo + (size_t)rt * RB;
    const size_t jt0 = (size_t)(sh.ct0 + ct) * BS_CB;

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const size_t iw0 = i0 + (size_t)(wave / WC) * IW;
    const size_t j0 = jt0 + (size_t)(wave % WC) * (64 * JR);
    if (iw0 >= sh.i_hi) return;
    if (sh.ut && j0 + 64 * JR - 1 <= iw0) return;

    uint32_t acc[IW][JR];
#pragma unroll
    for (int i = 0; i < IW; ++i)
#pragma unroll
        for (int c = 0; c < JR; ++c) acc[i][c] = 0;

    const uint32_t colb = ((uint32_t)j0 + (uint32_t)lane) * 4u;  // byte offset inside a plane (per lane): saddr + voffset form
    const uint32_t row = (uint32_t)iw0;                          // element offset of the wave's 16 row words (uniform)
    const size_t rstride = (size_t)(nbits_cap + 1) * Nstride;    // row operand: nbits_cap id planes + the unique plane per group
    const size_t cstride = (size_t)nbits_cap * Nstride;          // column operand: nbits_cap id planes per group

    BsOperands<JR> nx = bs_fetch<JR>(planes, cplanes, row, colb);             // group 0, plane 0
    int nbits_nx = live_planes(meta, 0);
    for (int tb = 0; tb < ntb; ++tb) {
        const int nbits = nbits_nx;                              // uniform, per 32-register group
        const bool last = tb + 1 >= ntb;                         // last group: harmless reload of group 0
        nbits_nx = live_planes(meta, last ? 0 : tb + 1);         // scalar load, one group ahead
        const uint32_t *rbase = planes + (size_t)tb * rstride, *cbase = cplanes + (size_t)tb * cstride;
        const uint32_t *rnext = last ? planes : rbase + rstride, *cnext = last ? cplanes : cbase + cstride;
        bs_group<JR>(nbits, rbase, cbase, rnext, cnext, Nstride, row, colb, nx, acc);
    }

    // interior: all 16 rows and all 64*JR columns of this wave are wanted pairs off the diagonal
    const bool interior = iw0 + IW <= sh.i_hi && j0 >= sh.j_lo && j0 + 64 * JR <= sh.j_hi &&
                          (sh.ut ? j0 > iw0 + IW - 1 : (j0 > iw0 + IW - 1 || j0 + 64 * JR <= iw0));
    if (interior) {
        uint32_t val[IW][JR];
#pragma unroll
        for (int i = 0; i < IW; ++i)
#pragma unroll
            for (int c = 0; c < JR; ++c) val[i][c] = store.value_from_mismatches(S, acc[i][c]);
        const uint32_t jl = (uint32_t)j0 + (uint32_t)lane;       // < 2^30
#pragma unroll
        for (int i = 0; i < IW; ++i) {
            const size_t rb = out_row_base(sh, iw0 + i);         // uniform: out_pos(ii, jj) = rb + jj
#pragma unroll
            for (int c = 0; c < JR; ++c) store.put_row(rb, jl + 64u * c, val[i][c]);
        }
        return;
    }
    // edge tiles.  Padded registers never mismatch; a sketch equals itself even where its values are
    // column-unique (the two codings of "unique" only separate DIFFERENT sketches)
    uint32_t val[IW][JR];
#pragma unroll
    for (int i = 0; i < IW; ++i)
#pragma unroll
        for (int c = 0; c < JR; ++c)
            val[i][c] = store.value((iw0 + i) == (j0 + lane + 64 * c) ? S : S - acc[i][c]);
#pragma unroll
    for (int i = 0; i < IW; ++i) {
        const size_t ii = iw0 + i;
        if (ii >= sh.i_hi) break;
#pragma unroll
        for (int c = 0; c < JR; ++c) {
            const size_t jj = j0 + lane + 64 * c;
            if (jj < sh.j_hi && jj >= sh.j_lo && (!sh.ut || jj > ii)) store.put(out_pos(sh, ii, jj), val[i][c]);
        }
    }
}

// gathered (caller-owned) operands carry the row coding + the unique plane only: derive the column coding.
// Done before EVERY launch on such a set -- the library cannot know when the caller re-gathered into the
// buffer, and the pass is ~2 % of the pair kernel it precedes.
int refresh_borrowed(d2g_ctx *ctx, const d2g_cmp_set *set, hipStream_t s) {
    if (!set->borrowed) return D2G_OK;
    dim3 grid((unsigned)div_up<size_t>(set->Nstride, 256), (unsigned)set->ntb);
    hipLaunchKernelGGL(bs_derive_kernel, grid, dim3(256), 0, s, set->d_planes, set->d_cplanes, set->Nstride, set->nbits_cap, set->d_meta);
    D2G_HIP(ctx, hipGetLastError());
    return D2G_OK;
}

template <class Store>
int launch_bitslice(d2g_ctx *ctx, const d2g_cmp_set *set, PairShape sh, Store store, hipStream_t s) {
    if (int rc = finish_shape(ctx, sh, 32u)) return rc;          // workgroup tile = 32 rows x 256 columns (4 waves of 16 x 128)
    if (sh.nvalid_total == 0) return D2G_OK;
    D2G_CHECK(ctx, (size_t)set->ntb * (set->nbits_cap + 1) * set->Nstride < (1ull << 32), "bit-sliced operand exceeds 2^32 words");
    if (int rc = refresh_borrowed(ctx, set, s)) return rc;
    d2g_timer tm(ctx, &ctx->ev_k2, s);
    hipLaunchKernelGGL((k2_bitslice_kernel<2, Store>), dim3(sh.per_xcd * 8), dim3(BS_THREADS), 0, s, set->d_planes, set->d_cplanes,
                       set->Nstride, set->nbits_cap, set->d_meta, set->ntb, (uint32_t)set->S, sh, store);
    tm.stop();
    D2G_HIP(ctx, hipGetLastError());
    return D2G_OK;
}

}  // namespace

void d2g_bitslice_free(d2g_cmp_set *set) {
    if (!set) return;
    if (!set->borrowed) { (void)hipFree(set->d_planes); (void)hipFree(set->d_meta); }
    (void)hipFree(set->d_cplanes);
    (void)hipFree(set->d_ids);
    set->d_planes = set->d_cplanes = set->d_meta = set->d_ids = nullptr;
}

// geometry of the bit-sliced operand: a function of N (and S) only, identical on every rank.
// A column holds at most floor(N/2) values that occur twice; ranks 1..D2 plus the two codes of "unique"
// (0 and 2^nbits - 1) need 2^nbits >= D2 + 2.
void d2g_bitslice_geometry(d2g_cmp_set *set) {
    set->nbits_cap = 1;
    while ((1ull << set->nbits_cap) < set->N / 2 + 2) ++set->nbits_cap;
    set->ntb = (int)div_up<size_t>(set->S, 32);
    set->Nstride = set->Npad + 64;
}

int d2g_bitslice_alloc_cplanes(d2g_ctx *ctx, d2g_cmp_set *set) {
    hipError_t e = hipMalloc((void **)&set->d_cplanes, (size_t)set->ntb * set->nbits_cap * set->Nstride * sizeof(uint32_t));
    if (e != hipSuccess) {
        ctx->last_error = std::string("bitslice alloc: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? D2G_ERR_NOMEM : D2G_ERR_HIP;
    }
    return D2G_OK;
}

// one-time allocation of the bit-sliced operand and its workspace
int d2g_bitslice_alloc(d2g_ctx *ctx, d2g_cmp_set *set) {
    const size_t N = set->N, S = set->S, Npad = set->Npad;
    if (N >= (1ull << 30)) { ctx->last_error = "bitslice: N too large"; return D2G_ERR_UNSUPPORTED; }
    d2g_bitslice_geometry(set);
    // hash space: power of two >= 1.5 N (load <= 2/3), at least 64 slots; walked in LDS-sized partitions
    set->T = 64; set->logT = 6;
    while ((uint64_t)set->T * 2 < (uint64_t)N * 3) { set->T <<= 1; ++set->logT; }
    hipError_t e;
    if ((e = hipMalloc((void **)&set->d_ids, S * Npad * sizeof(uint32_t))) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_meta, (size_t)(set->ntb + 4) * sizeof(uint32_t))) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_planes, (size_t)set->ntb * (set->nbits_cap + 1) * set->Nstride * sizeof(uint32_t))) != hipSuccess) {
        ctx->last_error = std::string("bitslice alloc: ") + hipGetErrorString(e);
        d2g_bitslice_free(set);
        return e == hipErrorOutOfMemory ? D2G_ERR_NOMEM : D2G_ERR_HIP;
    }
    if (int rc = d2g_bitslice_alloc_cplanes(ctx, set)) { d2g_bitslice_free(set); return rc; }
    return D2G_OK;
}

// ids + planes for the operand currently in set->d_cols.  Fully asynchronous on `s`.
// meta[0..ntb) = per-group shared-value counts; meta[ntb] = status word (bit 0: the rank kernel's LDS table
// overflowed on some column -- see d2g_bitslice_status)
int d2g_bitslice_prepare(d2g_ctx *ctx, d2g_cmp_set *set, hipStream_t s) {
    const size_t N = set->N, S = set->S, Npad = set->Npad;
    D2G_HIP(ctx, hipMemsetAsync(set->d_meta, 0, (size_t)(set->ntb + 4) * sizeof(uint32_t), s));
    {
        const int logTl = set->logT < BS_LOG_TLDS_MAX ? set->logT : BS_LOG_TLDS_MAX;
        const size_t lds = (size_t(1) << logTl) * sizeof(uint32_t);
        auto kern = set->logT > BS_LOG_TLDS_MAX ? bs_rank_kernel<true> : bs_rank_kernel<false>;
        if (lds > 48 * 1024)
            D2G_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)S), dim3(BS_RANK_THREADS), lds, s, set->d_cols, N, Npad, set->T, set->logT,
                           set->d_ids, set->d_meta, set->d_meta + set->ntb);
    }
    dim3 grid((unsigned)div_up<size_t>(set->Nstride, 256), (unsigned)set->ntb);
    hipLaunchKernelGGL(bs_planes_kernel, grid, dim3(256), 0, s, set->d_ids, S, N, Npad, set->d_planes, set->d_cplanes, set->Nstride,
                       set->nbits_cap, set->d_meta);
    D2G_HIP(ctx, hipGetLastError());
    return D2G_OK;
}

// synchronises `s`; D2G_ERR_INTERNAL when the last prepare overflowed its hash partitions
int d2g_bitslice_status(d2g_ctx *ctx, const d2g_cmp_set *set, hipStream_t s) {
    if (set->borrowed) return D2G_OK;                 // a gathered operand has no status word of its own
    uint32_t st = 0;
    D2G_HIP(ctx, hipMemcpyAsync(&st, set->d_meta + set->ntb, sizeof(st), hipMemcpyDeviceToHost, s));
    D2G_HIP(ctx, hipStreamSynchronize(s));
    if (st & 1u) {
        ctx->last_error = "bitslice prepare: a register column put more distinct values into one hash partition than its LDS table holds "
                          "(adversarial / extremely skewed column); use D2G_CMP_DIRECT for this matrix";
        return D2G_ERR_INTERNAL;
    }
    return D2G_OK;
}

int d2g_bitslice_ut(d2g_ctx *ctx, const d2g_cmp_set *set, size_t r0, size_t r1, uint32_t *eq_out, const float *lut,
                    float *fout, hipStream_t s) {
    PairShape sh{};
    sh.N = set->N; sh.i_lo = r0; sh.i_hi = r1; sh.j_lo = r0 + 1 < set->N ? r0 + 1 : set->N; sh.j_hi = set->N; sh.ut = 1;
    if (eq_out) return launch_bitslice(ctx, set, sh, StoreEq{eq_out}, s);
    return launch_bitslice(ctx, set, sh, StoreLut{fout, lut}, s);
}

int d2g_bitslice_rect(d2g_ctx *ctx, const d2g_cmp_set *set, size_t a0, size_t a1, size_t b0, size_t b1, uint32_t *eq_out,
                      hipStream_t s) {
    PairShape sh{};
    sh.N = set->N; sh.i_lo = a0; sh.i_hi = a1; sh.j_lo = b0; sh.j_hi = b1; sh.ut = 0;
    return launch_bitslice(ctx, set, sh, StoreEq{eq_out}, s);
}

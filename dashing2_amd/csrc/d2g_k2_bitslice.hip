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

typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
typedef u32x16 __attribute__((aligned(4))) u32x16_u;
constexpr uint32_t BS_EMPTY = 0xFFFFFFFFu;
constexpr int BS_RANK_THREADS = 1024;

// ------------------------------------------------------------------ 1. per-column dense ids
// One workgroup per register index t.  owner[] (T slots, pre-set to EMPTY) records the first
// sketch index that claimed a slot; equality is decided against that sketch's value, so no key
// storage and no reserved sentinel value is needed.
constexpr uint32_t BS_DUP = 0x80000000u;       // owner-table flag: the value has been seen again
constexpr uint32_t BS_UNIQ = 0x80000000u;      // id flag: value occurs once in its column (never equal)
constexpr uint32_t BS_PENDING = 0x40000000u;   // multi-partition rank kernel: ids[] holds a table slot of the current pass
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
// MULTI: more than one partition (N > 21845).  FAST: one partition and N <= 12288 -- a thread keeps all its values in
// registers; a kernel of its own so that the general path's registers do not cost it the second workgroup per CU.
// SPLIT (MULTI only): `nsplit` workgroups share one column, workgroup h = blockIdx.x / S walks the partitions
// [h nparts/nsplit, (h+1) nparts/nsplit) and ranks its values from 1 on its own; the ids carry h in bits 28-29 and
// bs_planes_kernel adds the offset of split h (the shared values the splits before it found: colcnt[t][h] after
// bs_colplan_kernel).  A column slice of N = 50 000 sketches x 64 registers -- one chunk of one rank of the 8-GPU
// exchange -- is 64 workgroups of one per CU otherwise: a quarter of the chip.
constexpr uint32_t BS_SPLIT_SHIFT = 28, BS_RANK_MASK = (1u << BS_SPLIT_SHIFT) - 1u;
template <bool MULTI, bool FAST>
__global__ __launch_bounds__(BS_RANK_THREADS) void bs_rank_kernel(const uint64_t *__restrict__ cols, size_t N, size_t Npad,
                                                                  uint32_t T, int logT, uint32_t *ids_all, uint32_t *colcnt,
                                                                  uint32_t *status, int tagbits_max, uint32_t S, int nsplit) {
    extern __shared__ __attribute__((aligned(16))) uint32_t own[];        // Tl owner slots
    const uint32_t split = MULTI ? blockIdx.x / S : 0u;
    const size_t t = MULTI ? blockIdx.x - split * S : blockIdx.x;
    const uint32_t htag = (MULTI && nsplit > 1) ? split << BS_SPLIT_SHIFT : 0u;
    const uint64_t *col = cols + t * Npad;
    uint32_t *ids = ids_all + t * Npad;
    const int tid = threadIdx.x;
    const int logTl = logT < BS_LOG_TLDS_MAX ? logT : BS_LOG_TLDS_MAX;
    const uint32_t Tl = 1u << logTl, mask = Tl - 1, nparts = MULTI ? (T >> logTl) : 1u;
    const uint32_t part_lo = MULTI ? split * (nparts / (uint32_t)nsplit) : 0u, part_hi = MULTI ? part_lo + nparts / (uint32_t)nsplit : 1u;
    __shared__ uint32_t wave_tot[BS_RANK_THREADS / 64];
    __shared__ uint32_t running;
    constexpr uint32_t BS_MAXFIX = 64;
    __shared__ uint32_t nfix, fix_j[BS_MAXFIX], fix_h[BS_MAXFIX];
    if (tid == 0) { running = 1; nfix = 0; }                             // id 0 is reserved for singletons
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
            if (cur != BS_EMPTY) own[h] = (cur & BS_DUP) ? (r++ | htag) : BS_UNIQ;
        }
        __syncthreads();
        if (tid == 0) running += tot;
        __syncthreads();
    };
    // Owner slot = [DUP:1][tag][owner sketch index:ib].  The tag (hash bits below the slot bits) lets a probe step
    // over a slot that holds a DIFFERENT value without fetching the owner's value from the column: only a
    // tag match -- practically always a true repeat -- pays the global load that decides equality exactly.
    // The all-ones index is never a sketch index (2^ib > N), so EMPTY cannot be mistaken for an owner.
    const int ib = 32 - __clz((uint32_t)N);
    const uint32_t idxmask = (ib >= 32 ? 0xFFFFFFFFu : (1u << ib) - 1u) & ~BS_DUP;
    // tagbits_max < 31 only in tests (D2G_BS_TAGBITS): a narrow tag makes tag collisions common
    const uint32_t tagfield = ~BS_DUP & ~idxmask & (ib + tagbits_max >= 31 ? 0xFFFFFFFFu : (1u << (ib + tagbits_max)) - 1u);
    auto slot_word = [&](uint64_t prod, uint32_t j) {                     // prod = v * K (bs_hash's product)
        return j | ((uint32_t)((prod << logT) >> 33) & tagfield);         // the 31 bits below the slot bits, cut to the field
    };
    // at most Tl probes: a partition that receives more than Tl distinct values (a skewed / adversarial
    // column; T >= 1.5 N only bounds the AVERAGE load) must not spin forever.  The overflow is
    // reported through *status; the host then falls back to the DIRECT algorithm or fails loudly.
    auto insert = [&](uint64_t v, uint32_t j, uint32_t h) {
        const uint32_t mine = slot_word(v * 0x9E3779B97F4A7C15ull, j);
        for (uint32_t probes = 0; probes < Tl; ++probes) {
            const uint32_t cur = atomicCAS(&own[h], BS_EMPTY, mine);
            if (cur == BS_EMPTY) return h;                                // first occurrence: we own the slot
            if (!((cur ^ mine) & tagfield) && col[cur & idxmask] == v) {  // same value seen again
                if (!(cur & BS_DUP)) atomicOr(&own[h], BS_DUP);
                return h;
            }
            h = (h + 1) & mask;
        }
        atomicOr(status, 1u);
        return h;                                                         // garbage id, flagged
    };

    constexpr int PF = 12;      // values a thread keeps in registers (fast path: N <= 12288, one partition)
    if constexpr (FAST) {
        // every value is fetched BEFORE the probe chains (a load inside the chain exposed a full HBM/L2 round
        // trip per value) and its slot stays in a register until the ids are written.
        // Two phases.  (1) claim: probe with LDS compare-and-swaps only; stop at the first slot that is won or
        // whose tag matches (the CANDIDATE: same value, up to a tag collision).  (2) confirm: the owners' values
        // of the candidates are fetched a few at a time -- independent loads, one exposed round trip per batch
        // instead of one per value -- and compared exactly; a confirmed repeat flags the slot, the rare tag
        // collision resumes the exact serial chain behind it.  Every thread holding the same value stops at the
        // same candidate (claimed slots never change owner or tag), so the outcome is the exact ranking.
        // (Probing a thread's values TOGETHER, round by round, was measured too: it needs > 64 VGPRs, i.e. one
        // workgroup per CU instead of two, and is slower -- profiles/r02_k2_experiments.txt.)
        uint64_t v[PF];
        uint32_t hs[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const size_t j = (size_t)i * BS_RANK_THREADS + tid;
            v[i] = j < N ? col[j] : 0;
        }
        for (uint32_t h = tid; h < Tl; h += BS_RANK_THREADS) own[h] = BS_EMPTY;
        __syncthreads();
        uint32_t candidate = 0;                                           // bit i: value i stopped at a tag match
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const size_t j = (size_t)i * BS_RANK_THREADS + tid;
            hs[i] = 0;
            if (j < N) {
                const uint64_t prod = v[i] * 0x9E3779B97F4A7C15ull;
                const uint32_t mine = slot_word(prod, (uint32_t)j);
                uint32_t h = (uint32_t)(prod >> (64 - logT)) & mask, probes = 0;
                for (; probes < Tl; ++probes) {
                    const uint32_t cur = atomicCAS(&own[h], BS_EMPTY, mine);
                    if (cur == BS_EMPTY) break;
                    if (!((cur ^ mine) & tagfield)) { candidate |= 1u << i; break; }
                    h = (h + 1) & mask;
                }
                if (probes == Tl) atomicOr(status, 1u);
                hs[i] = h;
            }
        }
        constexpr int CB = 2;                                             // owner fetches in flight per thread
        uint32_t redo = 0;
#pragma unroll
        for (int c = 0; c < PF; c += CB) {
            uint32_t o[CB];
            uint64_t w[CB];
#pragma unroll
            for (int g = 0; g < CB; ++g)
                if (candidate >> (c + g) & 1) o[g] = own[hs[c + g]];
#pragma unroll
            for (int g = 0; g < CB; ++g)
                if (candidate >> (c + g) & 1) w[g] = col[o[g] & idxmask];
#pragma unroll
            for (int g = 0; g < CB; ++g)
                if (candidate >> (c + g) & 1) {
                    if (w[g] == v[c + g]) {
                        if (!(o[g] & BS_DUP)) atomicOr(&own[hs[c + g]], BS_DUP);
                    } else redo |= 1u << (c + g);
                }
            __builtin_amdgcn_sched_barrier(0);                            // one batch of fetches at a time (VGPRs: 2 workgroups per CU)
        }
        // tag collisions (about one value in 10^5): the exact serial chain from the value's home slot; it walks past
        // the false candidate.  Their slots go through a small LDS list instead of hs[] (conditional updates of the
        // register array made the compiler keep several copies of it).
        const uint32_t skip = redo;
        while (redo) {
            const int i = __ffs(redo) - 1;
            redo &= redo - 1;
            const uint32_t j = (uint32_t)i * BS_RANK_THREADS + tid;
            const uint64_t vv = col[j];
            const uint32_t h2 = insert(vv, j, bs_hash(vv, logT) & mask);
            const uint32_t k = atomicAdd(&nfix, 1u);
            if (k < BS_MAXFIX) { fix_j[k] = j; fix_h[k] = h2; }
            else atomicOr(status, 1u);                                    // cannot happen by chance; the host falls back to DIRECT
        }
        __syncthreads();
        compact();
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const size_t j = (size_t)i * BS_RANK_THREADS + tid;
            if (j < N && !(skip >> i & 1)) ids[j] = own[hs[i]];
        }
        for (uint32_t k = tid; k < nfix && k < BS_MAXFIX; k += BS_RANK_THREADS) ids[fix_j[k]] = own[fix_h[k]];
        if (tid == 0) colcnt[t * BS_CC_STRIDE] = running - 1;             // #values shared by >= 2 sketches
        return;
    }

    // split workgroups write the same ids[] concurrently: a pending word names its partition (bits 15..29; N < 2^28 there), so
    // that a pass only resolves its own, and nobody initialises ids[] for the others (the buffer is zeroed once, at
    // allocation; afterwards it only ever holds final ids, which have bit 30 clear)
    const bool tagged = MULTI && nsplit > 1;
    for (uint32_t part = part_lo; part < part_hi; ++part) {
        const uint32_t ptag = BS_PENDING | (tagged ? part << 15 : 0u);
        for (uint32_t h = tid; h < Tl; h += BS_RANK_THREADS) own[h] = BS_EMPTY;
        __syncthreads();
        // values are fetched PG at a time before their probe chains, and a batch goes through the same two phases as the
        // fast path: claim with LDS compare-and-swaps only, then confirm the candidates with their owner fetches in flight
        // together (with one 128 KiB-table workgroup per CU nothing else hides a round trip)
        constexpr int PG = 8;
        for (size_t j0 = 0; j0 < N; j0 += (size_t)PG * BS_RANK_THREADS) {
            uint64_t v[PG];
            uint32_t hs[PG];
#pragma unroll
            for (int i = 0; i < PG; ++i) {
                const size_t j = j0 + (size_t)i * BS_RANK_THREADS + tid;
                v[i] = j < N ? col[j] : 0;
            }
            uint32_t mineb = 0, candidate = 0, redo = 0;
#pragma unroll
            for (int i = 0; i < PG; ++i) {
                const size_t j = j0 + (size_t)i * BS_RANK_THREADS + tid;
                const uint64_t prod = v[i] * 0x9E3779B97F4A7C15ull;
                const uint32_t hh = (uint32_t)(prod >> (64 - logT));
                hs[i] = 0;
                if (j < N && (!MULTI || (hh >> logTl) == part)) {
                    mineb |= 1u << i;
                    const uint32_t mine = slot_word(prod, (uint32_t)j);
                    uint32_t h = hh & mask, probes = 0;
                    for (; probes < Tl; ++probes) {
                        const uint32_t cur = atomicCAS(&own[h], BS_EMPTY, mine);
                        if (cur == BS_EMPTY) break;
                        if (!((cur ^ mine) & tagfield)) { candidate |= 1u << i; break; }
                        h = (h + 1) & mask;
                    }
                    if (probes == Tl) atomicOr(status, 1u);
                    hs[i] = h;
                }
            }
            {
                uint32_t o[PG];
                uint64_t w[PG];
#pragma unroll
                for (int i = 0; i < PG; ++i)
                    if (candidate >> i & 1) o[i] = own[hs[i]];
#pragma unroll
                for (int i = 0; i < PG; ++i)
                    if (candidate >> i & 1) w[i] = col[o[i] & idxmask];
#pragma unroll
                for (int i = 0; i < PG; ++i)
                    if (candidate >> i & 1) {
                        if (w[i] == v[i]) { if (!(o[i] & BS_DUP)) atomicOr(&own[hs[i]], BS_DUP); }
                        else redo |= 1u << i;
                    }
            }
            // MULTI: a slot written in this pass carries BS_PENDING (never set in a final id: ranks stay below 2^30, BS_UNIQ is
            // bit 31), so that the pass's second loop finds its values in ids[] alone, without re-reading and re-hashing the
            // column -- the kernel is bound by its column reads at this size.  Pass 0 writes every id (0 = not yet placed).
#pragma unroll
            for (int i = 0; i < PG; ++i) {
                const size_t j = j0 + (size_t)i * BS_RANK_THREADS + tid;
                if ((mineb & ~redo) >> i & 1) ids[j] = hs[i] | (MULTI ? ptag : 0u);
                else if (MULTI && !tagged && part == 0 && j < N && !(redo >> i & 1)) ids[j] = 0;
            }
            while (redo) {                                                // tag collisions: the exact chain from the home slot
                const int i = __ffs(redo) - 1;
                redo &= redo - 1;
                const size_t j = j0 + (size_t)i * BS_RANK_THREADS + tid;
                const uint64_t vv = col[j];
                ids[j] = insert(vv, (uint32_t)j, bs_hash(vv, logT) & mask) | (MULTI ? ptag : 0u);
            }
        }
        __syncthreads();
        compact();
        for (size_t j0 = 0; j0 < N; j0 += (size_t)PG * BS_RANK_THREADS) {
            uint32_t sl[PG];
#pragma unroll
            for (int i = 0; i < PG; ++i) {
                const size_t j = j0 + (size_t)i * BS_RANK_THREADS + tid;
                sl[i] = j < N ? ids[j] : 0;
            }
#pragma unroll
            for (int i = 0; i < PG; ++i) {
                const size_t j = j0 + (size_t)i * BS_RANK_THREADS + tid;
                if (j < N && (!MULTI || (tagged ? (sl[i] & ~0x7FFFu) == ptag : (sl[i] & BS_PENDING) != 0)))
                    ids[j] = own[sl[i] & (tagged ? 0x7FFFu : ~BS_PENDING)];
            }
        }
        __syncthreads();
    }
    if (tid == 0) colcnt[t * BS_CC_STRIDE + split] = running - 1;          // #values shared by >= 2 sketches (this split's)
}

// ------------------------------------------------------------------ 2. 32 x nbits bit transpose
// thread (tb, j): reads the ids of sketch j for 32 consecutive registers, writes nbits words.
__device__ __forceinline__ int live_planes(const uint32_t *meta, int tb) {
    const uint32_t md = meta[tb];                      // max over the group's columns of (#shared values D2) + 1
    return md <= 1 ? 1 : 32 - __clz(md);              // smallest nb with 2^nb >= D2 + 2: ranks 1..D2, 0 and 2^nb-1 all distinct
}

// first slot of group tb in the compact plane stream = sum of the live plane counts of the groups before it
__device__ __forceinline__ size_t stream_slot(const uint32_t *meta, int tb) {
    size_t q = 0;
    for (int t = 0; t < tb; ++t) q += (size_t)live_planes(meta, t);     // uniform scalar loop, ntb = S/32 is small
    return q;
}

// ------------------------------------------------------------------ 1b. column plan
// One workgroup, a kernel of its own.  (Letting the LAST workgroup of the rank kernel do this -- ticket counter -- was
// measured: with an agent-scope fence per workgroup the rank kernel went 52 -> 111 us at config 3, every fence writes the
// XCD's L2 back; fence-free, with returning device-scope atomics for the counts and the ticket, 52 -> 81 us.  The
// separate launch costs ~7 us.)  Input: colcnt[t][h] = shared values split h of the rank kernel found in column t.  Output:
//   colcnt[t][0..nsplit)  exclusive prefix over the splits (the rank offset bs_planes_kernel adds), colcnt[t][4] = D2(t)
//   perm[slot]            the column that sits in register slot `slot` of the operand (~0 = padding): the columns in
//                         DESCENDING order of their live-plane class (stable), so that a 32-register group holds
//                         columns of similar plane counts and meta[tb] -- a maximum -- does not let one busy column tax
//                         31 quiet ones.  Equality counts are sums over columns (reference src/cmp_core.cpp:461,506):
//                         any permutation gives the same counts.
//   meta[tb]              max over the group's columns of D2 + 1 (also to the export target, with the status word)
constexpr int BS_PLAN_THREADS = 1024;
constexpr int BS_PLAN_MAXS = 4096;                // slots sorted in LDS; larger sketches keep the identity order
constexpr uint32_t BS_NOCOL = 0xFFFFFFFFu;
__device__ __forceinline__ int plane_class(uint32_t d2) { return d2 == 0 ? 1 : 32 - __clz(d2 + 1); }   // = live_planes(D2 + 1)

__global__ __launch_bounds__(BS_PLAN_THREADS) void bs_colplan_kernel(uint32_t *__restrict__ colcnt, uint32_t S, int ntb, int nsplit,
                                                                       uint32_t *__restrict__ perm, uint32_t *__restrict__ meta,
                                                                       const uint32_t *__restrict__ status, uint32_t *__restrict__ ex_meta,
                                                                       uint32_t *__restrict__ ex_status, int sort) {
    __shared__ uint32_t d2s[BS_PLAN_MAXS];
    __shared__ uint16_t perm_s[BS_PLAN_MAXS];
    __shared__ uint32_t cell[32 * (BS_PLAN_MAXS / 64)];   // (class, 64-slot chunk) counts, then their exclusive prefix
    __shared__ uint32_t wave_tot[BS_PLAN_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t Spad = (uint32_t)ntb * 32u;
    auto column_total = [&](uint32_t t) {             // D2 of column t; leaves the per-split offsets behind
        uint32_t run = 0;
        for (int h = 0; h < nsplit; ++h) { const uint32_t c = colcnt[(size_t)t * BS_CC_STRIDE + h]; colcnt[(size_t)t * BS_CC_STRIDE + h] = run; run += c; }
        colcnt[(size_t)t * BS_CC_STRIDE + 4] = run;
        return run;
    };
    if (tid == 0 && ex_status) *ex_status = *status;
    if (!sort || Spad > BS_PLAN_MAXS) {                // identity order
        for (uint32_t t = tid; t < S; t += BS_PLAN_THREADS) (void)column_total(t);
        for (uint32_t p = tid; p < Spad; p += BS_PLAN_THREADS) perm[p] = p < S ? p : BS_NOCOL;
        __threadfence();
        __syncthreads();
        for (int g = tid; g < ntb; g += BS_PLAN_THREADS) {
            uint32_t mx = 0;
            for (uint32_t x = 0; x < 32; ++x) { const uint32_t t = (uint32_t)g * 32 + x; if (t < S) mx = max(mx, colcnt[(size_t)t * BS_CC_STRIDE + 4]); }
            meta[g] = mx + 1;
            if (ex_meta) ex_meta[g] = mx + 1;
        }
        return;
    }
    const uint32_t nchunk = (Spad + 63) / 64, ncell = 32 * nchunk;
    for (uint32_t c = tid; c < ncell; c += BS_PLAN_THREADS) cell[c] = 0;
    __syncthreads();
    constexpr int IT = BS_PLAN_MAXS / BS_PLAN_THREADS;
    uint32_t mycell[IT], myrank[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const uint32_t t = (uint32_t)it * BS_PLAN_THREADS + tid;      // a wave covers the 64 consecutive slots of chunk t / 64
        mycell[it] = 0; myrank[it] = 0;
        if (t - lane < Spad) {                                         // wave-uniform
            uint32_t d2 = 0;
            int cls = 0;                                               // class 0 = padding: sorted behind every real column
            if (t < S) { d2 = column_total(t); cls = plane_class(d2); }
            if (t < Spad) d2s[t] = d2;
            unsigned long long todo = __ballot(t < Spad);
            while (todo) {                                             // one round per distinct class in the wave
                const int k = __shfl(cls, __ffsll((long long)todo) - 1);
                const unsigned long long m = __ballot(cls == k && t < Spad);
                const uint32_t ci = (uint32_t)(31 - k) * nchunk + t / 64;   // descending class, ascending slot
                if (cls == k && t < Spad) { mycell[it] = ci; myrank[it] = __popcll(m & ((1ull << lane) - 1)); }
                if (lane == 0) cell[(uint32_t)(31 - k) * nchunk + (t - lane) / 64] = __popcll(m);
                todo &= ~m;
            }
        }
    }
    __syncthreads();
    {   // exclusive prefix over the cells (<= 2048): two per thread
        const uint32_t c0 = 2u * tid, a = c0 < ncell ? cell[c0] : 0u, b = c0 + 1 < ncell ? cell[c0 + 1] : 0u;
        uint32_t incl = a + b;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t x = __shfl_up(incl, o); if (lane >= o) incl += x; }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; ++w) woff += wave_tot[w];
        const uint32_t ex = woff + incl - (a + b);
        if (c0 < ncell) cell[c0] = ex;
        if (c0 + 1 < ncell) cell[c0 + 1] = ex + a;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const uint32_t t = (uint32_t)it * BS_PLAN_THREADS + tid;
        if (t < Spad) perm_s[cell[mycell[it]] + myrank[it]] = (uint16_t)t;
    }
    __syncthreads();
    for (uint32_t p = tid; p < Spad; p += BS_PLAN_THREADS) { const uint32_t t = perm_s[p]; perm[p] = t < S ? t : BS_NOCOL; }
    for (int g = tid; g < ntb; g += BS_PLAN_THREADS) {
        uint32_t mx = 0;
        for (uint32_t x = 0; x < 32; ++x) mx = max(mx, d2s[perm_s[(uint32_t)g * 32 + x]]);
        meta[g] = mx + 1;
        if (ex_meta) ex_meta[g] = mx + 1;
    }
}

// Writes both forms of the operand:
//   planes  [ntb][nbits_cap+1][Nstride]  fixed geometry (a function of N only): row-coded id planes + the unique plane in the
//                                        last slot -- the form ranks exchange (independent per group);
//   stream  [sum_tb nbits_tb][2][Nstride] what the pair kernel walks: only the LIVE planes, in group order, each as the
//                                        row-coded words followed by the column-coded words, so that the kernel's operand
//                                        pointer simply advances by one block per plane (no per-plane address selection).
// Register slot x of group tb holds column perm[32 tb + x] (bs_colplan_kernel).
constexpr int BS_FORM_STREAM = 1, BS_FORM_EXCHANGE = 2;
// start-of-prepare work of the sparse-tile path (section 4) carried by a kernel that runs anyway (bs_planes_kernel, sp_unpack_kernel)
// instead of a launch and a memset of its own: label[j] = j, cnt[j] = 0, order[1] = `inexact`, `zwords` words at `zero` cleared
struct SpInit {
    uint32_t *label = nullptr, *cnt = nullptr, *order = nullptr, *zero = nullptr;
    uint32_t n = 0, inexact = 0, zwords = 0;
};
__device__ __forceinline__ void sp_init_part(const SpInit &si, size_t lin, size_t nthreads) {
    if (!si.label) return;
    if (lin < si.n) { si.label[lin] = (uint32_t)lin; si.cnt[lin] = 0; }
    if (lin == 0) { si.order[1] = si.inexact; si.order[2] = 0; }
    for (size_t x = lin; x < si.zwords; x += nthreads) si.zero[x] = 0;
}
constexpr size_t BS_SLACK = 64;          // words behind position Npad of every plane (Nstride = Npad + BS_SLACK)
template <bool SPLIT>
__global__ __launch_bounds__(256) void bs_planes_kernel(const uint32_t *__restrict__ ids, size_t N, size_t Npad,
                                                        uint32_t *__restrict__ planes, uint32_t *__restrict__ stream,
                                                        size_t Nstride, int nbits_cap, const uint32_t *__restrict__ meta, int forms,
                                                        const uint32_t *__restrict__ perm, const uint32_t *__restrict__ colcnt,
                                                        const uint32_t *__restrict__ sperm, SpInit si) {
    const size_t jpos = (size_t)blockIdx.x * 256 + threadIdx.x;   // position in the operand
    const size_t tb = blockIdx.y;
    sp_init_part(si, tb * ((size_t)gridDim.x * 256) + jpos, (size_t)gridDim.x * 256 * gridDim.y);
    if (jpos >= Nstride) return;
    // sperm: the operand is written in the sparse path's sorted order -- position p holds sketch sperm[p] (stream form only)
    const size_t j = sperm ? (size_t)sperm[jpos] : jpos;          // 0xFFFFFFFF (padding) fails j < N below
    const int nbits = live_planes(meta, (int)tb);
    // the group's 32 columns: two s_load_dwordx16 (constant address space: never written while this kernel runs), all in
    // flight before the first id is requested
    typedef const u32x16_u __attribute__((address_space(4))) *slots_ptr;
    const u32x16_u pa = *(slots_ptr)(uintptr_t)(perm + tb * 32), pb = *(slots_ptr)(uintptr_t)(perm + tb * 32 + 16);
    uint32_t id[32];
#pragma unroll
    for (int x = 0; x < 32; ++x) {
        const uint32_t t = x < 16 ? pa[x] : pb[x - 16];
        const bool real = t != BS_NOCOL && j < N;             // padded registers/sketches: id 0 in both codings
        uint32_t w = real ? ids[(size_t)t * Npad + j] : 0u;
        if (SPLIT && real && !(w >> 31)) w = (w & BS_RANK_MASK) + colcnt[(size_t)t * BS_CC_STRIDE + (w >> BS_SPLIT_SHIFT)];
        id[x] = w;
    }
    uint32_t u = 0;                                    // the "unique" plane
#pragma unroll
    for (int x = 0; x < 32; ++x) u |= (id[x] >> 31) << x;
    // forms: BS_FORM_STREAM = what this GPU's pair kernel walks; BS_FORM_EXCHANGE = what ranks exchange -- written only
    // once somebody has asked for it (d2g_bitslice_export: 29 MB of stores per prepare at config 3 that a single GPU never reads)
    uint32_t *dst = planes + tb * (size_t)(nbits_cap + 1) * Nstride + jpos;
    const bool ex = forms & BS_FORM_EXCHANGE, st = forms & BS_FORM_STREAM;
    uint32_t *sdst = st ? stream + stream_slot(meta, (int)tb) * 2 * Nstride + jpos : nullptr;
    for (int b = 0; b < nbits; ++b) {
        uint32_t w = 0;
#pragma unroll
        for (int x = 0; x < 32; ++x) w |= ((id[x] >> b) & 1u) << x;
        if (ex) dst[(size_t)b * Nstride] = w;          // row coding: unique = 0 (BS_UNIQ ids have zero low bits)
        if (st) {
            sdst[(size_t)(2 * b) * Nstride] = w;
            sdst[(size_t)(2 * b + 1) * Nstride] = w | u;   // column coding: unique = all ones
        }
    }
    // the 64 slack words behind position Npad of a plane are never read as sketches: in the exchanged form the first 32 of the unique
    // plane's carry D2 of the group's 32 register slots (what the sparse-tile path of a rank that only receives planes sizes its bit
    // sets with: sp_unpack_kernel)
    if (ex) {
        uint32_t uw = u;
        if (jpos >= Npad && jpos < Npad + 32) { const uint32_t t = perm[tb * 32 + (jpos - Npad)]; uw = t != BS_NOCOL ? colcnt[(size_t)t * BS_CC_STRIDE + 4] : 0u; }
        dst[(size_t)nbits_cap * Nstride] = uw;
    }
}

// plane stream of an operand that arrived in the exchanged form (the gathered operand of the multi-GPU path,
// d2g_cmp_set_from_planes_dev)
__global__ __launch_bounds__(256) void bs_derive_kernel(const uint32_t *__restrict__ planes, uint32_t *__restrict__ stream,
                                                        size_t Nstride, int nbits_cap, const uint32_t *__restrict__ meta, int tb0) {
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t tb = (size_t)tb0 + blockIdx.y;
    if (j >= Nstride) return;
    const int nbits = live_planes(meta, (int)tb);
    const uint32_t *src = planes + tb * (size_t)(nbits_cap + 1) * Nstride + j;
    uint32_t *sdst = stream + stream_slot(meta, (int)tb) * 2 * Nstride + j;
    const uint32_t u = j + BS_SLACK < Nstride ? src[(size_t)nbits_cap * Nstride] : 0u;     // the slack words of the unique plane carry D2 per slot, not sketches
    for (int b = 0; b < nbits; ++b) {
        const uint32_t w = src[(size_t)b * Nstride];
        sdst[(size_t)(2 * b) * Nstride] = w;
        sdst[(size_t)(2 * b + 1) * Nstride] = w | u;
    }
}

// ------------------------------------------------------------------ 3. the pair kernel
constexpr int BS_THREADS = 256;
constexpr int BS_CB = 256;                // columns per workgroup tile (all variants)

// v_bitop3_b32 truth table: src0 = 0xF0, src1 = 0xCC, src2 = 0xAA
constexpr unsigned BITOP3_C_OR_A_XOR_B = 0xAA | (0xF0 ^ 0xCC);   // mismatch accumulation

__device__ __forceinline__ bool sp_dense_mode(const uint32_t *__restrict__ ctl, uint32_t cand);   // section 4

// IW = 16 rows per wave (one s_load_dwordx16 per plane), JR = 64-column groups per lane,
// WC = waves side by side along the columns (WC * JR * 64 = 256).  Per 32-register group: plane 0
// initialises z = r ^ c (no zeroing), planes 1.. accumulate with v_bitop3 z |= r ^ c, and one accumulating
// v_bcnt finishes the group: nbits + 1 VALU operations per pair and group.
//
// The kernel is bound by instruction ISSUE, all kinds counted: at 4, 5 or 6 waves per SIMD it takes the same
// time, and it got 13 % faster when the per-plane address arithmetic was (experimentally) constant-folded
// away -- scalar instructions are not free riders next to the vector ones.  Hence the plane STREAM: the
// live planes of all groups lie back to back (row words, then column words), the wave keeps ONE uniform
// pointer that advances by one block per plane, row words are an s_load_dwordx16 at the pointer and column
// words a global_load_dword at pointer + per-lane offset (saddr + voffset form: no vector address math).
// The operands of the next plane are requested before the current one is computed and land in the other
// of two explicitly alternating register sets.
//
// Epilogue: interior tiles (every pair of the wave's 16 x 64*JR block is wanted and off the diagonal --
// all but the ones on the triangle's edge) take a branch-free path: the row's output base is a scalar,
// the lane adds its column, so an output costs one table gather and one store.
#ifndef D2G_BS_WPE
#define D2G_BS_WPE 7
#endif
#ifndef D2G_BS_JR
#define D2G_BS_JR 2
#endif
constexpr int BS_IW = 16;
constexpr int BS_JR = D2G_BS_JR;              // 64-column groups per lane: a wave owns 16 x (64*JR) pairs

template <int JR>
struct BsOperands {                         // the prefetched operands of one plane
    u32x16_u sa;                            // 16 row words (SGPRs)
    uint32_t vb[JR];                        // this lane's column words
};

// Scalar loads return out of order, so the only wait there is for them is lgkmcnt(0): it must come BEFORE
// the next s_load is issued -- left to the compiler it lands at the first use of the current operands,
// after the prefetch was issued, and the prefetch is then waited for on the spot.
template <int JR>
__device__ __forceinline__ BsOperands<JR> bs_fetch(const uint32_t *&ptr, uint32_t coff, size_t step) {
    __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0): the operands about to be used have arrived
    __builtin_amdgcn_sched_barrier(0);
    BsOperands<JR> o;
    // constant address space: the operand is never written while this kernel runs, and a uniform address in
    // that space is always a scalar load (s_load_dwordx16), whatever the optimiser can or cannot prove
    typedef const u32x16_u __attribute__((address_space(4))) *row_words_ptr;
    o.sa = *(row_words_ptr)(uintptr_t)ptr;
    // the per-lane offset is laundered so that loop strength reduction cannot turn (uniform pointer + lane
    // offset) into a per-lane 64-bit running address: that costs a vector add per plane and four more VGPRs;
    // as written the loads select the saddr + voffset form
    uint32_t co = coff;
    asm volatile("" : "+v"(co));
#pragma unroll
    for (int c = 0; c < JR; ++c)
        o.vb[c] = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(ptr) + co + 256 * c);
    ptr += step;
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
// next group on exit (the stream has one block of slack after its last plane).  A copy between the two
// operand sets (8 s_mov_b64 + JR v_mov) happens at most once per GROUP, when the plane count is odd.
template <int JR>
__device__ __forceinline__ void bs_group(int nbits, const uint32_t *&ptr, uint32_t coff, size_t step, BsOperands<JR> &a,
                                         uint32_t (&acc)[BS_IW][JR]) {
    uint32_t z[BS_IW][JR];
    BsOperands<JR> b = bs_fetch<JR>(ptr, coff, step);
    bs_plane<JR, true>(a, z);                        // plane 0
    const int rest = nbits - 1;                      // planes 1 .. nbits-1, two per iteration (a counted loop: z stays in place)
    for (int k = rest >> 1; k > 0; --k) {
        a = bs_fetch<JR>(ptr, coff, step);
        bs_plane<JR, false>(b, z);                   // odd plane
        b = bs_fetch<JR>(ptr, coff, step);
        bs_plane<JR, false>(a, z);                   // even plane
    }
    if (rest & 1) {
        a = bs_fetch<JR>(ptr, coff, step);           // the next group's plane 0
        bs_plane<JR, false>(b, z);
    } else {
        a = b;                                       // b holds the next group's plane 0
    }
#pragma unroll
    for (int i = 0; i < BS_IW; ++i)
#pragma unroll
        for (int c = 0; c < JR; ++c) acc[i][c] += __builtin_popcount(z[i][c]);     // v_bcnt_u32_b32 acc, z, acc
}

template <int JR, class Store>
__global__ __launch_bounds__(BS_THREADS) __attribute__((amdgpu_waves_per_eu(D2G_BS_WPE))) void k2_bitslice_kernel(
    const uint32_t *__restrict__ stream, size_t Nstride, const uint32_t *__restrict__ meta, int ntb, uint32_t S, PairShape sh, Store store,
    const uint32_t *__restrict__ gate, uint32_t gate_cand) {
    constexpr int IW = BS_IW;
    constexpr int WC = BS_CB / (64 * JR);          // waves along columns: 2 (JR=2)
    constexpr int WR = 4 / WC;                     // waves along rows
    constexpr int RB = WR * IW;                    // rows per workgroup tile
    if (gate && !sp_dense_mode(gate, gate_cand)) return;   // launched behind the sparse path: only when that decided for the dense walk
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

    // one uniform pointer: the wave's 16 row words of the current plane; its column words sit at a fixed
    // per-lane byte offset from it (Nstride >= Npad + 64 > iw0, so the offset is positive and < 2^32)
    const uint32_t *ptr = stream + iw0;
    const uint32_t coff = (uint32_t)(Nstride - iw0 + j0 + (size_t)lane) * 4u;
    const size_t step = 2 * Nstride;                             // words per plane block (row words + column words)

    BsOperands<JR> nx = bs_fetch<JR>(ptr, coff, step);           // group 0, plane 0
    int nbits_nx = live_planes(meta, 0);
    for (int tb = 0; tb < ntb; ++tb) {
        const int nbits = nbits_nx;                              // uniform, per 32-register group
        nbits_nx = live_planes(meta, tb + 1 < ntb ? tb + 1 : 0); // scalar load, one group ahead
        bs_group<JR>(nbits, ptr, coff, step, nx, acc);
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

// ------------------------------------------------------------------ 4. sparse tiles
// An equality count is zero unless the two sketches share a value in at least one register column.  In a collection of related
// genomes most pairs share nothing (different species), and the ones that do come in families.  So:
//   prepare  FAMILIES = the connected components of "shares a value in some column": label propagation (sp_prop*: a label is an earlier
//            member of the sketch's family) and ONE lock-free union-find pass over every column (sp_flatten / sp_union: exact components,
//            no iteration); the sketches are counting-sorted by root (sp_count / scan / place) and the finished plane stream is permuted
//            into THAT order (sp_permute) -- a family becomes a run of adjacent positions, a "segment";
//   launch   the 32-row x 256-column tiles a segment's rows and columns meet in become a work list (every pair with a shared value lies
//            inside one segment); when the segments would cover too much -- families that are large but sparse inside -- every shared
//            value marks the tiles its holders meet in instead (sp_mark_kernel: bit sets per value in LDS); the output is pre-filled with
//            the value of "0 equal registers"; the pair kernel walks the listed tiles only and stores where the count is not 0.
// Nothing here is approximate: a tile that is not listed holds no pair with a common value -- by the components being exact in the
// first case, by construction of the marks (whatever the order looks like) in the second.  If one family would take more than half of
// the sketches (everything is connected), or the labels form long chains, the caller's order is kept and every tile is walked.  Rows of
// a partial launch [r0, r1) are gathered (in sorted order) into a row operand of their own.
constexpr uint32_t SP_NONE = 0xFFFFFFFFu;
#ifndef D2G_SP_WIDE_U
#define D2G_SP_WIDE_U 8          // sketches per thread and step of the wide mark kernel (measured at N = 50 000: 8 -> 532 us, 16 -> 856 us)
#endif
#ifndef D2G_SP_WIDE_FOLD
#define D2G_SP_WIDE_FOLD 0       // 0: a thread per row block ORs the values' sets into its bitmap row (532 us); 1: a work item per value and row word, atomicOr into the slot (777 us)
#endif
#ifndef D2G_SP_KS
#define D2G_SP_KS 4
#endif
#ifndef SP_EXP_NO_GLOBAL_MARKS
#define SP_EXP_NO_GLOBAL_MARKS 0       // timing experiment (tools/build_variant.sh): the mark kernel without its global phase
#endif

__device__ __forceinline__ uint32_t sp_rank(uint32_t w, const uint32_t *__restrict__ colcnt, size_t t, bool split) {
    if ((w >> 31) || w == 0) return 0;          // unique (or padding): never equal to anything
    return split ? (w & BS_RANK_MASK) + colcnt[t * BS_CC_STRIDE + (w >> BS_SPLIT_SHIFT)] : w;
}

__device__ __forceinline__ uint32_t sp_ld(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Label propagation, one workgroup walking SEVERAL columns one after the other: per column, the smallest LIVE label among the holders of
// each shared value (LDS), then every holder takes it.  The columns a workgroup walks later see what all workgroups wrote before, so one
// launch does the work of several synchronous rounds: a family whose first member shares registers with only some of the others is under
// one root after it, where one synchronous round (min over frozen labels, then a gather per sketch: rounds 1-3 of this file) left two.
// Labels are written with PLAIN stores: a label is only ever replaced by a smaller index of the same family (v <= the holder's own label
// <= its index), so whichever of two racing stores lands last -- or whichever XCD's L2 writes its copy of the line back last -- the array
// still holds, per sketch, a member of its family that is no later than itself: all the sort needs (it hops to the root), and the tiles
// are checked (sp_check_kernel) or marked exactly afterwards whatever the order.  (atomicMin instead: every column moves every holder's
// label through a device-scope atomic on one 40 KB array: 46 us instead of 17 at config 3.)
__global__ __launch_bounds__(1024) void sp_prop_kernel(const uint32_t *__restrict__ ids, size_t N, size_t Npad, uint32_t ncols, const uint32_t *__restrict__ colcnt,
                                                       int split, uint32_t cap, uint32_t *__restrict__ label) {
    extern __shared__ uint32_t sp_g[];
    const uint32_t T = blockDim.x;
    for (size_t t = blockIdx.x; t < ncols; t += gridDim.x) {
        const uint32_t d2 = colcnt[t * BS_CC_STRIDE + 4];
        if (d2 == 0) continue;
        const uint32_t nv = min(d2, cap);                              // values beyond the table take no part (the order is a heuristic; exactness is checked later)
        for (uint32_t r = threadIdx.x; r < nv; r += T) sp_g[r] = SP_NONE;
        __syncthreads();
        for (size_t j0 = 0; j0 < N; j0 += (size_t)T * 8) {
            uint32_t w[8], kk[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const size_t j = j0 + (size_t)x * T + threadIdx.x;
                w[x] = j < N ? ids[t * Npad + j] : 0u;
                kk[x] = j < N ? sp_ld(&label[j]) : SP_NONE;
            }
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const uint32_t r = sp_rank(w[x], colcnt, t, split != 0);
                if (r && r <= nv) atomicMin(&sp_g[r - 1], kk[x]);
            }
        }
        __syncthreads();
        for (size_t j0 = 0; j0 < N; j0 += (size_t)T * 8) {
            uint32_t w[8], kk[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const size_t j = j0 + (size_t)x * T + threadIdx.x;
                w[x] = j < N ? ids[t * Npad + j] : 0u;
                kk[x] = j < N ? sp_ld(&label[j]) : 0u;
            }
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const uint32_t r = sp_rank(w[x], colcnt, t, split != 0);
                if (!r || r > nv) continue;
                const uint32_t v = sp_g[r - 1];
                if (v < kk[x]) label[j0 + (size_t)x * T + threadIdx.x] = v;
            }
        }
        __syncthreads();
    }
}

// the same for N <= 1024 U sketches: a thread keeps its U ids and labels of a column in registers between the two passes (one round of
// loads per column instead of two times ceil(N / 8192))
template <int U>
__global__ __launch_bounds__(1024) void sp_prop_reg_kernel(const uint32_t *__restrict__ ids, size_t N, size_t Npad, uint32_t ncols, const uint32_t *__restrict__ colcnt,
                                                           int split, uint32_t cap, uint32_t *__restrict__ label) {
    extern __shared__ uint32_t sp_g[];
    for (size_t t = blockIdx.x; t < ncols; t += gridDim.x) {
        const uint32_t d2 = colcnt[t * BS_CC_STRIDE + 4];
        if (d2 == 0) continue;
        const uint32_t nv = min(d2, cap);
        uint32_t w[U], kk[U];
#pragma unroll
        for (int x = 0; x < U; ++x) {
            const size_t j = (size_t)x * 1024 + threadIdx.x;
            w[x] = j < N ? ids[t * Npad + j] : 0u;
            kk[x] = j < N ? sp_ld(&label[j]) : SP_NONE;
        }
        for (uint32_t r = threadIdx.x; r < nv; r += 1024) sp_g[r] = SP_NONE;
        __syncthreads();
#pragma unroll
        for (int x = 0; x < U; ++x) {
            w[x] = sp_rank(w[x], colcnt, t, split != 0);
            if (w[x] > nv) w[x] = 0;
            if (w[x]) atomicMin(&sp_g[w[x] - 1], kk[x]);
        }
        __syncthreads();
#pragma unroll
        for (int x = 0; x < U; ++x) {
            if (!w[x]) continue;
            const uint32_t v = sp_g[w[x] - 1];
            if (v < kk[x]) label[(size_t)x * 1024 + threadIdx.x] = v;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void sp_jump_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, size_t N) {
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j < N) out[j] = in[in[j]];
}

// label[] maps a sketch to an earlier (or the same) sketch of its family; the root is where that stops -- the TRUE root, always: the
// segments are exact only then.  Most chains end after one or two hops; the union pass's hooks can leave long ones (a collection that is
// one chain hooks i + 1 under i for every i), so the walk halves the path behind it like sp_flatten_kernel (label[l] < l off the root: it ends)
__device__ __forceinline__ uint32_t sp_root(uint32_t *label, size_t j) {
    // (plain loads: nobody hooks roots while this runs, and a stale label is still an ancestor -- a device-scope load per hop costs 3-4x as much)
    uint32_t l = (uint32_t)j;
    for (;;) {
        const uint32_t p = label[l];
        if (p == l) break;
        const uint32_t g = label[p];
        if (g != p) label[l] = g;
        l = p;
    }
    return l;
}

// block-wide exclusive scan of one value per thread (1024 threads); returns the exclusive prefix, *total = the sum
__device__ __forceinline__ uint32_t sp_block_scan(uint32_t v, uint32_t *wave_tot, uint32_t *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t x = __shfl_up(incl, o); if (lane >= o) incl += x; }
    __syncthreads();
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
    for (int w = 0; w < 16; ++w) { const uint32_t x = wave_tot[w]; if (w < wave) woff += x; tot += x; }
    *total = tot;
    return woff + incl - v;
}

// counting sort of the sketches by the root of their label, three small kernels (one thread per sketch, then one workgroup for the
// prefix, then one thread per sketch again).  The roots of a family collection are few and their counters hot, but with a thread
// per sketch every thread waits for ONE atomic; a single workgroup walking all sketches waited for ten in a row (N = 50 000: 225 us,
// now ~25).  One root holding more than half of the sketches (everything is connected) keeps the caller's order: order[0] = 1.
__global__ __launch_bounds__(256) void sp_count_kernel(uint32_t *label, uint32_t *__restrict__ root, size_t N, uint32_t *__restrict__ cnt, const uint32_t *__restrict__ order) {
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = j < N;
    if (order[2]) { if (live) root[j] = (uint32_t)j; return; }        // deep chains: no walks; the scan keeps the caller's order
    const uint32_t r = live ? sp_root(label, j) : SP_NONE;
    if (live) root[j] = r;
    // when everything hangs together ONE counter takes all N increments (measured: 115 us at N = 10 000): the lanes that share the
    // wave's first root add once; the others go one by one (matching every distinct root of a wave costs more than it saves when a
    // wave holds 64 different ones: 64 rounds of ballot + shuffle, + 40-100 us on the family / unrelated matrices)
    const unsigned long long alive = __ballot(live);
    if (!alive) return;
    const uint32_t lead = __shfl(r, __ffsll((long long)alive) - 1);
    const unsigned long long m = __ballot(live && r == lead);
    if (live && r == lead) { if ((threadIdx.x & 63) == (unsigned)(__ffsll((long long)m) - 1)) atomicAdd(&cnt[lead], (uint32_t)__popcll(m)); }
    else if (live) atomicAdd(&cnt[r], 1u);
}
__global__ __launch_bounds__(1024) void sp_scan_kernel(uint32_t *__restrict__ cnt, size_t N, uint32_t *__restrict__ order, uint32_t *__restrict__ start, uint32_t seg_tile_limit) {
    // exclusive prefix in place, 8192 counters at a time through LDS (coalesced both ways; a thread scans its eight in LDS)
    __shared__ uint32_t wave_tot[16];
    __shared__ __attribute__((aligned(16))) uint32_t tile[8192];
    __shared__ uint32_t s_big, s_run, s_est;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x;
    if (tid == 0) { s_big = 0; s_run = 0; s_est = 0; }
    uint32_t est = 0;                                                  // tiles the segments would cover (both triangles), saturating
    __syncthreads();
    // the next tile's counters are requested before this tile is scanned (one workgroup: nothing else hides the round trip)
    uint32_t pre[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { const size_t x = (size_t)k * 1024 + tid; pre[k] = x < N ? cnt[x] : 0u; }
    for (size_t base = 0; base < N; base += 8192) {
        const uint32_t n = (uint32_t)min((size_t)8192, N - base);
#pragma unroll
        for (int k = 0; k < 8; ++k) tile[k * 1024 + tid] = pre[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) { const size_t x = base + 8192 + (size_t)k * 1024 + tid; pre[k] = x < N ? cnt[x] : 0u; }
        uint32_t v[8], sum = 0, big = 0;
        {   // a thread's eight counters as two 16-byte LDS reads (one word at a time: stride 8 words, an 8-way bank conflict)
            const u32x4 a = reinterpret_cast<const u32x4 *>(tile)[tid * 2], b = reinterpret_cast<const u32x4 *>(tile)[tid * 2 + 1];
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        }
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            sum += v[x]; big = max(big, v[x]);
            // in sixteenths of a tile: a segment of c >= 32 sketches covers at most (rows + 1) x (columns + 1) tiles; smaller ones share their
            // row block with their neighbours (two column tiles for c / 32 of a row block)
            const uint32_t c = min(v[x], 32768u);                    // (a segment that long is past any limit by itself)
            est += c >= 32 ? 16u * ((c + 31) / 32 + 1) * ((c + 255) / 256 + 1) : (c >= 2 ? c : 0u);
        }
        est = min(est, 0x0FFFFFFFu);
        if ((size_t)big * 2 > N) s_big = 1;
        uint32_t total;
        uint32_t run = sp_block_scan(sum, wave_tot, &total) + s_run;
        {
            uint32_t o[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) { o[x] = run; run += v[x]; }
            reinterpret_cast<u32x4 *>(tile)[tid * 2] = u32x4{o[0], o[1], o[2], o[3]};
            reinterpret_cast<u32x4 *>(tile)[tid * 2 + 1] = u32x4{o[4], o[5], o[6], o[7]};
        }
        __syncthreads();
        for (uint32_t x = tid; x < n; x += 1024) { cnt[base + x] = tile[x]; start[base + x] = tile[x]; }   // cnt becomes the placing cursor (-> segment end), start stays
        if (tid == 0) s_run += total;
        __syncthreads();
    }
    est = min(est, 0x3FFFFFu) / 16 + 1;                                 // 1024 threads x 2^18: no overflow
    for (int o = 32; o > 0; o >>= 1) est += __shfl_down(est, o);
    if ((tid & 63) == 0) atomicAdd(&s_est, est);
    __syncthreads();
    if (tid == 0) { const uint32_t keep = s_big | (order[2] ? 1u : 0u); order[0] = keep; if (keep || s_est > seg_tile_limit + 1024) order[1] = 1; }
}
__global__ __launch_bounds__(256) void sp_place_kernel(const uint32_t *__restrict__ root, size_t N, size_t Nstride, uint32_t *__restrict__ cnt,
                                                        uint32_t *__restrict__ sperm, uint32_t *__restrict__ sinv, const uint32_t *__restrict__ order) {
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = j < N;
    if (!live && j < Nstride) sperm[j] = SP_NONE;
    if (order[0]) { if (live) { sperm[j] = (uint32_t)j; sinv[j] = (uint32_t)j; } return; }
    const uint32_t r = live ? root[j] : SP_NONE;
    const int lane = threadIdx.x & 63;
    uint32_t p = 0;
    const unsigned long long alive = __ballot(live);
    if (!alive) return;
    // as in sp_count_kernel: the lanes that share the wave's first root move its cursor once (and keep their order), the others one by one
    const uint32_t lead = __shfl(r, __ffsll((long long)alive) - 1);
    const unsigned long long m = __ballot(live && r == lead);
    const int first = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (lane == first) base = atomicAdd(&cnt[lead], (uint32_t)__popcll(m));
    base = __shfl(base, first);
    if (live && r == lead) p = base + (uint32_t)__popcll(m & ((1ull << lane) - 1));
    else if (live) p = atomicAdd(&cnt[r], 1u);
    if (live) { sperm[p] = (uint32_t)j; sinv[j] = p; }
}

// the sorted stream from the caller's-order stream: position p takes the words of sketch sperm[p].  Only the row-coded words and
// ONE column-coded word per group are gathered (the unique plane is their difference in any plane: r ^ c = u where the register
// is column-unique, 0 elsewhere); both codings are written.
__global__ __launch_bounds__(256) void sp_permute_kernel(const uint32_t *__restrict__ nat, uint32_t *__restrict__ srt, size_t Nstride, const uint32_t *__restrict__ meta,
                                                         const uint32_t *__restrict__ sperm, const uint32_t *__restrict__ order) {
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int tb = blockIdx.y;
    if (p >= Nstride || order[0]) return;                             // the caller's order was kept: every launch walks the caller's-order stream (dense), nobody reads this one
    const int nbits = live_planes(meta, tb);
    const size_t slot = stream_slot(meta, tb);
    const uint32_t j = sperm[p];
    uint32_t *dst = srt + slot * 2 * Nstride + p;
    if (j == SP_NONE) {
        for (int b = 0; b < nbits; ++b) { dst[(size_t)(2 * b) * Nstride] = 0; dst[(size_t)(2 * b + 1) * Nstride] = 0; }
        return;
    }
    const uint32_t *src = nat + slot * 2 * Nstride + j;
    const uint32_t u = src[0] ^ src[Nstride];
    for (int b = 0; b < nbits; ++b) {
        const uint32_t w = src[(size_t)(2 * b) * Nstride];
        dst[(size_t)(2 * b) * Nstride] = w;
        dst[(size_t)(2 * b + 1) * Nstride] = w | u;
    }
}

// launch rows: the sorted positions whose sketch lies in [r0, r1), in sorted order (stable compaction, one workgroup; 8192 positions at
// a time through LDS so that the loads are coalesced and a thread still owns eight consecutive positions: N = 50 000 80 -> ~12 us)
__global__ __launch_bounds__(1024) void sp_rows_kernel(const uint32_t *__restrict__ sperm, size_t N, uint32_t r0, uint32_t r1, uint32_t nrows_pad,
                                                       uint32_t *__restrict__ rowpos, uint32_t *__restrict__ rowk) {
    __shared__ uint32_t wave_tot[16];
    __shared__ __attribute__((aligned(16))) uint32_t tile[8192];
    __shared__ uint32_t s_run;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x;
    if (tid == 0) s_run = 0;
    __syncthreads();
    uint32_t pre[8];                                                   // the next tile is requested before this one is compacted (as in sp_scan_kernel)
#pragma unroll
    for (int k = 0; k < 8; ++k) { const size_t x = (size_t)k * 1024 + tid; pre[k] = x < N ? sperm[x] : SP_NONE; }
    for (size_t base = 0; base < N; base += 8192) {
#pragma unroll
        for (int k = 0; k < 8; ++k) tile[k * 1024 + tid] = pre[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) { const size_t x = base + 8192 + (size_t)k * 1024 + tid; pre[k] = x < N ? sperm[x] : SP_NONE; }
        uint32_t jv[8], cnt = 0;
        {
            const u32x4 a = reinterpret_cast<const u32x4 *>(tile)[tid * 2], b = reinterpret_cast<const u32x4 *>(tile)[tid * 2 + 1];
            jv[0] = a.x; jv[1] = a.y; jv[2] = a.z; jv[3] = a.w; jv[4] = b.x; jv[5] = b.y; jv[6] = b.z; jv[7] = b.w;
        }
#pragma unroll
        for (int x = 0; x < 8; ++x) cnt += (jv[x] >= r0 && jv[x] < r1) ? 1u : 0u;   // SP_NONE (beyond N) is in no range
        uint32_t total;
        uint32_t k = sp_block_scan(cnt, wave_tot, &total) + s_run;
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            if (jv[x] == SP_NONE) continue;
            const bool w = jv[x] >= r0 && jv[x] < r1;
            rowk[jv[x]] = w ? k : SP_NONE;
            if (w) rowpos[k++] = (uint32_t)(base + tid * 8 + x);
        }
        __syncthreads();
        if (tid == 0) s_run += total;
        __syncthreads();
    }
    for (uint32_t x = s_run + tid; x < nrows_pad; x += 1024) rowpos[x] = SP_NONE;
}

__global__ __launch_bounds__(256) void sp_gather_kernel(const uint32_t *__restrict__ stream, size_t Nstride, const uint32_t *__restrict__ meta, int ntb,
                                                        const uint32_t *__restrict__ rowpos, uint32_t nrows_pad, uint32_t *__restrict__ rowstream, size_t rstride) {
    const size_t q = blockIdx.y;
    if (q >= stream_slot(meta, ntb)) return;
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nrows_pad) return;
    const uint32_t p = rowpos[k];
    rowstream[q * rstride + k] = p != SP_NONE ? stream[2 * q * Nstride + p] : 0u;
}

// sorted position p: the first position of (its segment x its row block) sets the tiles of that row block against the segment's column blocks
__device__ __forceinline__ void sp_segtiles(size_t p, const uint32_t *__restrict__ sperm, const uint32_t *__restrict__ root, const uint32_t *__restrict__ start,
                                            const uint32_t *__restrict__ end, size_t N, uint32_t CW, uint32_t *__restrict__ gbm) {
    if (p >= N) return;
    const uint32_t r = root[sperm[p]];
    const uint32_t a = start[r], b = end[r];
    if (b - a < 2 || !(p == a || (p & 31) == 0)) return;              // a sketch alone under its root shares nothing with anybody
    const uint32_t rb = (uint32_t)(p >> 5), cb0 = a >> 8, cb1 = (b - 1) >> 8;
    for (uint32_t cw = cb0 >> 5; cw <= cb1 >> 5; ++cw) {
        const uint32_t lo = cw == (cb0 >> 5) ? (cb0 & 31) : 0u, hi = cw == (cb1 >> 5) ? (cb1 & 31) : 31u;
        const uint32_t m = (hi == 31 ? 0xFFFFFFFFu : ((1u << (hi + 1)) - 1u)) & ~((1u << lo) - 1u);
        atomicOr(&gbm[(size_t)rb * CW + cw], m);
    }
}

// Per register column: every shared value marks the tiles its holders meet in.  LDS: for `gm` values at a time, a bit set of the
// launch-row blocks (32 rows) and one of the column blocks (256 sorted positions) the value occurs in.  A value whose holders
// meet in more than a quarter of all tiles says the matrix is not sparse: it raises the ALL flag (ctl[1] bit 0) and marking stops.
// The marks of a column go to the column's OWN copy of the tile bitmap (`slots`), which sp_or_kernel folds into one afterwards:
// every column marks the same few hundred tiles, and 1024 workgroups testing / setting the same 2.5 KB through device-scope
// operations all queue at one memory channel (measured: 48 of the kernel's 70 us at config 3).
template <int U>
__global__ __launch_bounds__(512) void sp_mark_kernel(const uint32_t *__restrict__ ids, size_t N, size_t Npad, const uint32_t *__restrict__ colcnt, int split,
                                                      const uint32_t *__restrict__ sinv, const uint32_t *__restrict__ rowk, uint32_t gm, uint32_t RW, uint32_t CW,
                                                      uint32_t nrb, uint32_t ncb, uint32_t lbm_words, uint32_t *__restrict__ slots, uint32_t *__restrict__ ctl,
                                                      const uint32_t *__restrict__ order_kept, const uint32_t *__restrict__ sperm, const uint32_t *__restrict__ root,
                                                      const uint32_t *__restrict__ seg_start, const uint32_t *__restrict__ seg_end, uint32_t *__restrict__ segbm) {
    extern __shared__ uint32_t sp_lds_all[];
    __shared__ uint32_t s_stop;
    // lbm_words != 0: the column's bitmap is built in LDS and stored once; otherwise (large N) it is built in the slot with atomics
    uint32_t *lbm = sp_lds_all;
    uint32_t *sp_lds = sp_lds_all + lbm_words;
    const size_t t = blockIdx.x;
    const uint32_t T = blockDim.x;                                    // 256, or 512 with a bigger share of the LDS (large N)
    const uint32_t words = nrb * CW;
    uint32_t *slot = slots + t * (size_t)words;
    if (!order_kept[1]) {                                             // the sort's segments give the tiles (sp_union_kernel): no marking, the grid's threads
        if (segbm)                                                    // share the sorted positions instead
            for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < N; p += (size_t)gridDim.x * blockDim.x) sp_segtiles(p, sperm, root, seg_start, seg_end, N, CW, segbm);
        return;
    }
    const uint32_t d2 = colcnt[t * BS_CC_STRIDE + 4];
    // one family holds most sketches (the prepare kept the caller's order), or this column's shared values have fewer than four holders
    // on average (pairs, not families), or it would take more than 32 passes of bit sets: not a matrix the tile list can help
    if (d2 && (order_kept[0] || (size_t)d2 * 4 > N || d2 > 32 * gm)) { if (threadIdx.x == 0) atomicOr(&ctl[1], 1u); return; }
    if (d2 == 0 || !lbm_words) {
        for (uint32_t x = threadIdx.x; x < words; x += T) slot[x] = 0;
        if (d2 == 0) return;
        __threadfence();
    }
    const uint32_t W = RW + CW;
    const uint32_t dense_limit = max(16u, (uint32_t)(((size_t)nrb * ncb) / 4));
    for (uint32_t x = threadIdx.x; x < lbm_words; x += T) lbm[x] = 0;
    for (uint32_t base = 0; base < d2; base += gm) {
        if (threadIdx.x == 0) s_stop = sp_ld(&ctl[1]) & 1u;            // marking stops for everybody once somebody gave up
        const uint32_t n = min(gm, d2 - base);
        for (uint32_t x = threadIdx.x; x < n * W; x += T) sp_lds[x] = 0;
        __syncthreads();
        if (s_stop) return;
        // U sketches per thread and step; the loads of the next step are issued before this step's bit sets are updated
        uint32_t w[U], k[U], c[U];
        auto load = [&](size_t j0) {
#pragma unroll
            for (int x = 0; x < U; ++x) {
                const size_t j = j0 + (size_t)x * T + threadIdx.x;
                w[x] = j < N ? ids[t * Npad + j] : 0u;
                c[x] = j < N ? sinv[j] : 0u;
                k[x] = j < N ? (rowk ? rowk[j] : c[x]) : SP_NONE;
            }
        };
        load(0);
        for (size_t j0 = 0; j0 < N; j0 += (size_t)T * U) {
            uint32_t cw_[U], ck[U], cc[U];
#pragma unroll
            for (int x = 0; x < U; ++x) { cw_[x] = w[x]; ck[x] = k[x]; cc[x] = c[x]; }
            if (j0 + (size_t)T * U < N) load(j0 + (size_t)T * U);
#pragma unroll
            for (int x = 0; x < U; ++x) {
                const uint32_t r = sp_rank(cw_[x], colcnt, t, split != 0);
                if (!r || r - 1 < base || r - 1 >= base + n) continue;
                uint32_t *bits = sp_lds + (size_t)(r - 1 - base) * W;
#if SP_EXP_NO_GLOBAL_MARKS == 2
                if (ck[x] == 0x12345678u) atomicOr(&bits[0], cc[x]);          // timing experiment: the loop without its LDS atomics
#else
                if (ck[x] != SP_NONE) atomicOr(&bits[ck[x] >> 10], 1u << ((ck[x] >> 5) & 31));
                atomicOr(&bits[RW + (cc[x] >> 13)], 1u << ((cc[x] >> 8) & 31));
#endif
            }
        }
        __syncthreads();
        // density guard, one thread per value
        for (uint32_t q = threadIdx.x; q < n; q += T) {
            const uint32_t *bits = sp_lds + (size_t)q * W;
            uint32_t nr = 0, nc = 0;
            for (uint32_t rw = 0; rw < RW; ++rw) nr += __popc(bits[rw]);
            for (uint32_t cw = 0; cw < CW; ++cw) nc += __popc(bits[RW + cw]);
            if (nr * nc > dense_limit) atomicOr(&ctl[1], 1u);
        }
        if (lbm_words) {
            // fold into the LDS bitmap, one thread per value: row bits x column words, ds_or (typed LDS pointer: through a generic
            // pointer that could also be the global slot these were flat atomics and cost 47 of the kernel's 69 us at config 3)
            // (one work item per value AND row word: 66 values alone would leave three of the four waves idle)
            for (uint32_t it = threadIdx.x; it < (SP_EXP_NO_GLOBAL_MARKS == 1 ? 0u : n * RW); it += T) {
                const uint32_t q = it / RW, rw = it - q * RW;
                const uint32_t *bits = sp_lds + (size_t)q * W;
                uint32_t rbits = bits[rw];
                while (rbits) {
                    const uint32_t rb = rw * 32 + (uint32_t)__ffs(rbits) - 1;
                    rbits &= rbits - 1;
                    for (uint32_t cw = 0; cw < CW; ++cw) { const uint32_t cbits = bits[RW + cw]; if (cbits) atomicOr(&lbm[(size_t)rb * CW + cw], cbits); }
                }
            }
        } else {
#if D2G_SP_WIDE_FOLD == 0
            for (uint32_t rb = threadIdx.x; rb < nrb; rb += T) {
                const uint32_t rw = rb >> 5, rbit = 1u << (rb & 31);
                for (uint32_t cw0 = 0; cw0 < CW; cw0 += 8) {
                    uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    const uint32_t ncw = min(8u, CW - cw0);
                    for (uint32_t q = 0; q < n; ++q) {
                        const uint32_t *bits = sp_lds + (size_t)q * W;
                        if (!(bits[rw] & rbit)) continue;
#pragma unroll
                        for (uint32_t x = 0; x < 8; ++x) if (x < ncw) acc[x] |= bits[RW + cw0 + x];
                    }
#pragma unroll
                    for (uint32_t x = 0; x < 8; ++x) if (x < ncw && acc[x]) slot[(size_t)rb * CW + cw0 + x] |= acc[x];
                }
            }
        }
#else
            // fold into the global slot (large N), one work item per value and row word: atomicOr without a return value into the column's
            // OWN slot (nobody else touches it; zeroed + fenced above).  (One thread per row block reading all values' row words from
            // LDS was measured at N = 50 000: 96 us of the workgroup's 250.)
            for (uint32_t it = threadIdx.x; it < (SP_EXP_NO_GLOBAL_MARKS == 1 ? 0u : n * RW); it += T) {
                const uint32_t q = it / RW, rw = it - q * RW;
                const uint32_t *bits = sp_lds + (size_t)q * W;
                uint32_t rbits = bits[rw];
                while (rbits) {
                    const uint32_t rb = rw * 32 + (uint32_t)__ffs(rbits) - 1;
                    rbits &= rbits - 1;
                    for (uint32_t cw = 0; cw < CW; ++cw) {
                        const uint32_t cbits = bits[RW + cw];
                        if (cbits) (void)__hip_atomic_fetch_or(&slot[(size_t)rb * CW + cw], cbits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        }
#endif
        __syncthreads();
    }
    for (uint32_t x = threadIdx.x; x < lbm_words; x += T) slot[x] = lbm[x];
}

// ---- tiles from the sort's segments.  After the propagation, ONE pass over every column unites whatever roots the holders of one shared
// value still sit under (lock-free union-find on the label array: a root is hooked under a smaller one with a compare-and-swap; a pass
// over every "edge" of the shares-a-value graph leaves exactly its connected components, no iteration).  The sort then puts every
// component's sketches side by side, every pair with a shared value lies inside one segment, and the tiles a segment's rows and columns
// meet in are a superset of the tiles that hold such a pair -- without the marking pass and its two bit sets per value (config 3:
// 29 + 5 us -> 12 + 5; config 4: 531 + 9 -> 63 + 5).  The propagation has done nearly all the uniting with plain stores; this pass mostly
// confirms (few compare-and-swaps).  order[1] != 0 ("inexact": a column has more shared values than the LDS table holds, the caller's
// order was kept, or the segments would cover more than an eighth of all tiles -- components that are large but sparse inside, where
// exact marks list far fewer tiles) sends the launches to the exact marking instead.
__device__ __forceinline__ uint32_t sp_find(uint32_t *label, uint32_t l) {
    for (int h = 0; h < 64; ++h) {                                    // bounded: sp_union retries, and gives up (-> exact marks) in the end
        const uint32_t p = sp_ld(&label[l]);
        if (p == l) break;
        const uint32_t g = sp_ld(&label[p]);
        if (g != p) label[l] = g;                                     // path halving: only a non-root's label moves, to one of its ancestors
        l = p;
    }
    return l;
}
__device__ __forceinline__ bool sp_union(uint32_t *label, uint32_t a, uint32_t b) {
    for (int it = 0; it < 64; ++it) {
        a = sp_find(label, a); b = sp_find(label, b);
        if (a == b) return true;
        if (a < b) { const uint32_t x = a; a = b; b = x; }              // the larger root goes under the smaller one
        if (atomicCAS(&label[a], a, b) == a) return true;               // a was still a root: hooked
    }
    return false;
}
// every label straight at its root before the union pass compares labels; the walk halves the path behind it.  Families leave chains of two
// or three labels.  A chain that is not at its root after SP_MAX_HOPS hops means long strings of sketches that share registers with their
// neighbours only (a time series; the extreme, ONE chain of N sketches, costs N / 2 dependent loads in the last thread: 0.7 ms at N = 12 000):
// one big component in all likelihood, and nothing the tile list could help -- order[2] is raised, the union pass and the sort's root walks are
// skipped, the caller's order is kept and the launches walk every tile (the same outcome as "one family holds most sketches", found early).
// Racing with itself is harmless: a label is only ever replaced by an ancestor.  (plain loads, as in sp_root)
constexpr int SP_MAX_HOPS = 64;
__global__ __launch_bounds__(256) void sp_flatten_kernel(uint32_t *label, size_t N, uint32_t *__restrict__ order) {
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    uint32_t l = (uint32_t)j;
    for (int h = 0;; ++h) {
        const uint32_t p = label[l];
        if (p == l) break;
        if (h == SP_MAX_HOPS) { order[2] = 1; return; }
        const uint32_t g = label[p];
        if (g != p) label[l] = g;
        l = p;
    }
    label[j] = l;
}
__global__ void sp_union_kernel(const uint32_t *__restrict__ ids, size_t N, size_t Npad, const uint32_t *__restrict__ colcnt, int split,
                                uint32_t *label, uint32_t cap, uint32_t *__restrict__ order) {
    extern __shared__ uint32_t sp_chk[];
    const size_t t = blockIdx.x;
    const uint32_t T = blockDim.x;
    const uint32_t d2 = colcnt[t * BS_CC_STRIDE + 4];
    if (d2 == 0 || order[2]) return;                                  // order[2]: deep chains (sp_flatten_kernel): the caller's order will be kept
    if (d2 > cap) { if (threadIdx.x == 0) atomicOr(&order[1], 1u); return; }
    for (uint32_t r = threadIdx.x; r < d2; r += T) sp_chk[r] = SP_NONE;
    __syncthreads();
    bool bad = false;
    for (size_t j0 = 0; j0 < N; j0 += (size_t)T * 8) {
        uint32_t w[8], rt[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            const size_t j = j0 + (size_t)x * T + threadIdx.x;
            w[x] = j < N ? ids[t * Npad + j] : 0u;
            rt[x] = j < N ? sp_ld(&label[j]) : 0u;
        }
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            const uint32_t r = sp_rank(w[x], colcnt, t, split != 0);
            if (!r) continue;
            // the value's first holder leaves its label; a holder with the same label is under the same root without looking (the usual
            // case after the propagation); only a different label costs the walk to the roots
            const uint32_t old = atomicCAS(&sp_chk[r - 1], SP_NONE, rt[x]);
            if (old != SP_NONE && old != rt[x]) bad |= !sp_union(label, old, rt[x]);
        }
    }
    if (bad) atomicOr(&order[1], 1u);
}

// tile bitmap = OR over the columns' copies.  grid (words / 256, SP_OR_SPLIT): a thread folds S / SP_OR_SPLIT copies of one word.
constexpr int SP_OR_SPLIT = 32;
__global__ __launch_bounds__(256) void sp_or_kernel(const uint32_t *__restrict__ slots, uint32_t words, uint32_t S, uint32_t *__restrict__ tilebm,
                                                    const uint32_t *__restrict__ ctl, const uint32_t *__restrict__ order) {
    const uint32_t x = blockIdx.x * 256 + threadIdx.x;
    if (x >= words || (ctl[1] & 1u) || !order[1]) return;
    const uint32_t t0 = (uint32_t)((size_t)S * blockIdx.y / SP_OR_SPLIT), t1 = (uint32_t)((size_t)S * (blockIdx.y + 1) / SP_OR_SPLIT);
    uint32_t acc = 0;
    uint32_t t = t0;
    for (; t + 8 <= t1; t += 8) {
        uint32_t v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = slots[(size_t)(t + i) * words + x];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc |= v[i];
    }
    for (; t < t1; ++t) acc |= slots[(size_t)t * words + x];
    if (acc) atomicOr(&tilebm[x], acc);
}

// tile bitmap of a PARTIAL launch from the global one: a block of 32 launch rows may meet what any of the sorted row blocks its rows
// come from may meet (a superset of the exact marks of those rows: still no tile with a match is missed)
__global__ __launch_bounds__(256) void sp_rowbm_kernel(const uint32_t *__restrict__ gbm, const uint32_t *__restrict__ rowpos, uint32_t nrb, uint32_t CW,
                                                       uint32_t *__restrict__ tilebm) {
    const uint32_t x = blockIdx.x * 256 + threadIdx.x;
    if (x >= nrb * CW) return;
    const uint32_t rb = x / CW, cw = x - rb * CW;
    uint32_t acc = 0;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) { const uint32_t p = rowpos[rb * 32 + i]; if (p != SP_NONE) acc |= gbm[(size_t)(p >> 5) * CW + cw]; }
    tilebm[x] = acc;
}

// the marked tiles as a work list (any order: a workgroup reserves the range of its tiles with one atomic).  full: rows are ALL
// sorted positions and a pair is computed where row position < column position, so tiles entirely below that diagonal are not
// candidates.  ctl[0] = tiles listed, ctl[3] = candidates (what the dense / sparse decision compares it with).
__global__ __launch_bounds__(1024) void sp_list_kernel(const uint32_t *__restrict__ tilebm, uint32_t nrb, uint32_t ncb, uint32_t CW, int full,
                                                       uint32_t *__restrict__ tiles, uint32_t *__restrict__ ctl, uint32_t cand, const uint32_t *__restrict__ gflags,
                                                       uint32_t *__restrict__ ctl_next) {
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t s_base;
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && tid < 8) ctl_next[tid] = 0;                // the next launch's control words (nobody else touches them during this launch)
    if (blockIdx.x == 0 && tid == 0) ctl[3] = cand;                  // for d2g_cmp_set_sparse_info
    if (gflags[1] & 1u) {                                           // the (global) marking gave up: the dense kernel runs instead
        if (blockIdx.x == 0 && tid == 0) atomicOr(&ctl[1], 1u);
        return;
    }
    const size_t ntile = (size_t)nrb * ncb;
    const size_t a = ((size_t)blockIdx.x * 1024 + tid) * 8, b = min(ntile, a + 8);
    uint32_t n = 0, mask = 0;
    for (size_t x = a; x < b; ++x) {
        const uint32_t rb = (uint32_t)(x / ncb), cb = (uint32_t)(x % ncb);
        if (full && (size_t)rb * 32 > (size_t)cb * 256 + 255) continue;
        if ((tilebm[(size_t)rb * CW + (cb >> 5)] >> (cb & 31)) & 1u) { mask |= 1u << (x - a); ++n; }
    }
    uint32_t total;
    uint32_t o = sp_block_scan(n, wave_tot, &total);
    if (tid == 0) s_base = total ? atomicAdd(&ctl[0], total) : 0u;
    __syncthreads();
    o += s_base;
    for (uint32_t x = 0; x < 8; ++x) if ((mask >> x) & 1u) tiles[o++] = (uint32_t)(a + x);
}

// dense or sparse?  DENSE: the plain pair kernel walks every tile (and writes every output itself); otherwise the output is
// pre-filled and the sparse kernel walks the list.  Dense when marking gave up (ALL) or when more than `cand * 0.4` tiles are
// listed -- the sparse kernel pays for its generality with a per-element epilogue.
// evaluated by every consumer of the list (fill, sparse kernel, gated dense kernel) from the same two words: no kernel of its own
__device__ __forceinline__ bool sp_dense_mode(const uint32_t *__restrict__ ctl, uint32_t cand) {
    return (ctl[1] & 1u) != 0 || (size_t)ctl[0] * 5 > (size_t)cand * 2;
}

constexpr int SP_FILL_PER_THREAD = 8, SP_FILL_THREADS = 256;
template <class Store>
__global__ __launch_bounds__(SP_FILL_THREADS) void sp_fill_kernel(uint32_t *__restrict__ out, size_t cnt, Store store, uint32_t S, const uint32_t *__restrict__ ctl, uint32_t cand) {
    if (sp_dense_mode(ctl, cand)) return;                           // dense mode: the pair kernel writes every output
    const uint32_t v = store.value_from_mismatches(S, S);           // the value of "no register equal"
    const size_t n4 = cnt / 4;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    // the output pointer of a slab is only 4-byte aligned in general: head, 16-byte body, tail
    const size_t head = min(cnt, (size_t)((16 - ((uintptr_t)out & 15)) & 15) / 4);
    u32x4 *body = reinterpret_cast<u32x4 *>(out + head);
    const size_t nb = (cnt - head) / 4;
    (void)n4;
    // a workgroup writes ONE contiguous 32 KB piece (8 x 256 16-byte stores), the workgroups in dispatch order: a streaming write
    const size_t base = (size_t)blockIdx.x * (SP_FILL_THREADS * SP_FILL_PER_THREAD) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < SP_FILL_PER_THREAD; ++k) { const size_t i = base + (size_t)k * SP_FILL_THREADS; if (i < nb) body[i] = u32x4{v, v, v, v}; }
    if (blockIdx.x == 0) {
        if (threadIdx.x < head) out[threadIdx.x] = v;
        const size_t tail0 = head + nb * 4;
        if (tail0 + threadIdx.x < cnt) out[tail0 + threadIdx.x] = v;
    }
}

// two-pointer operand fetch: 16 row words at a uniform pointer of the (possibly gathered) row operand, column words at a uniform
// pointer + lane offset of the sorted stream's column coding
template <int JR>
__device__ __forceinline__ BsOperands<JR> sp_fetch(const uint32_t *&rp, size_t rstep, const uint32_t *&cp, uint32_t coff, size_t cstep) {
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_sched_barrier(0);
    BsOperands<JR> o;
    typedef const u32x16_u __attribute__((address_space(4))) *row_words_ptr;
    o.sa = *(row_words_ptr)(uintptr_t)rp;
    uint32_t co = coff;
    asm volatile("" : "+v"(co));
#pragma unroll
    for (int c = 0; c < JR; ++c)
        o.vb[c] = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(cp) + co + 256 * c);
    rp += rstep;
    cp += cstep;
    return o;
}

template <int JR>
__device__ __forceinline__ void sp_group(int nbits, const uint32_t *&rp, size_t rstep, const uint32_t *&cp, uint32_t coff, size_t cstep, BsOperands<JR> &a,
                                         uint32_t (&acc)[BS_IW][JR]) {
    uint32_t z[BS_IW][JR];
    BsOperands<JR> b = sp_fetch<JR>(rp, rstep, cp, coff, cstep);
    bs_plane<JR, true>(a, z);
    const int rest = nbits - 1;
    for (int k = rest >> 1; k > 0; --k) {
        a = sp_fetch<JR>(rp, rstep, cp, coff, cstep);
        bs_plane<JR, false>(b, z);
        b = sp_fetch<JR>(rp, rstep, cp, coff, cstep);
        bs_plane<JR, false>(a, z);
    }
    if (rest & 1) {
        a = sp_fetch<JR>(rp, rstep, cp, coff, cstep);
        bs_plane<JR, false>(b, z);
    } else {
        a = b;
    }
#pragma unroll
    for (int i = 0; i < BS_IW; ++i)
#pragma unroll
        for (int c = 0; c < JR; ++c) acc[i][c] += __builtin_popcount(z[i][c]);
}

struct SpArgs {
    const uint32_t *stream;       // sorted plane stream
    size_t Nstride;
    const uint32_t *rowstream;    // gathered row words of a partial launch, or nullptr: the rows are all sorted positions
    size_t rstride;
    const uint32_t *meta;
    int ntb;
    uint32_t S, N;
    const uint32_t *sperm, *rowpos, *tiles, *ctl;
    uint32_t ncb, cand;
};

// The sparse pair kernel.  A listed tile (32 launch rows x 256 sorted columns) is four 16 x 128 sub-tiles; a workgroup takes ONE
// sub-tile and its four waves each walk a quarter of the 32-register groups, then add their mismatch counts in LDS.  (The dense
// kernel gives every wave a sub-tile and all groups: with a few hundred listed tiles that leaves one or two waves per SIMD, each
// waiting out the latency of every plane's loads -- measured 87 us for 432 tiles at config 3, 411 us for 2122 at config 4.)
template <int JR, class Store>
__global__ __launch_bounds__(64 * D2G_SP_KS) __attribute__((amdgpu_waves_per_eu(D2G_BS_WPE))) void k2_bitslice_sparse_kernel(SpArgs a, PairShape sh, Store store) {
    constexpr int IW = BS_IW;
    constexpr int WC = BS_CB / (64 * JR);
    constexpr int KS = D2G_SP_KS;                                   // waves per sub-tile = splits of the group range
    static_assert(JR == 2, "the LDS reduction packs a lane's two column groups into one word");
    __shared__ uint32_t red[IW][64];                                // per row and lane: mismatches of column group 0 | group 1 << 16 (a sum stays below 2^16: S < 65536 asserted by the host)
    if (sp_dense_mode(a.ctl, a.cand)) return;                       // dense mode
    const uint32_t nsub = a.ctl[0] * 4u;
    const int ks = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const bool full = a.rowstream == nullptr;
    const int g0 = a.ntb * ks / KS, g1 = a.ntb * (ks + 1) / KS;
    const size_t slot0 = stream_slot(a.meta, g0);
    for (uint32_t si = blockIdx.x; si < nsub; si += gridDim.x) {
        const uint32_t tile = a.tiles[si >> 2], sub = si & 3u;
        const uint32_t rb = tile / a.ncb, cb = tile - rb * a.ncb;
        const size_t k0 = (size_t)rb * 32 + (size_t)(sub / WC) * IW;              // first launch row of this sub-tile
        const size_t c0 = (size_t)cb * BS_CB + (size_t)(sub % WC) * (64 * JR);   // first sorted column position
        if (full && k0 > c0 + 64 * JR - 1) continue;                             // entirely below the diagonal of sorted positions (uniform for the workgroup)
        for (int x = threadIdx.x; x < IW * 64; x += 64 * KS) (&red[0][0])[x] = 0;
        __syncthreads();
        uint32_t acc[IW][JR];
#pragma unroll
        for (int i = 0; i < IW; ++i)
#pragma unroll
            for (int c = 0; c < JR; ++c) acc[i][c] = 0;
        if (g1 > g0) {
            const size_t rstep = full ? 2 * a.Nstride : a.rstride;
            const size_t cstep = 2 * a.Nstride;
            const uint32_t *rp = (full ? a.stream : a.rowstream) + k0 + slot0 * rstep;
            const uint32_t *cp = a.stream + a.Nstride + c0 + slot0 * cstep;
            const uint32_t coff = (uint32_t)lane * 4u;
            BsOperands<JR> nx = sp_fetch<JR>(rp, rstep, cp, coff, cstep);
            int nbits_nx = live_planes(a.meta, g0);
            for (int tb = g0; tb < g1; ++tb) {
                const int nbits = nbits_nx;
                nbits_nx = live_planes(a.meta, tb + 1 < a.ntb ? tb + 1 : 0);
                sp_group<JR>(nbits, rp, rstep, cp, coff, cstep, nx, acc);
            }
        }
        // every wave adds its share of the mismatch counts in LDS; afterwards wave ks finishes rows [ks IW/KS, (ks+1) IW/KS) -- the epilogue
        // (caller's indices, condensed position, table value, store) is ~45 instructions per pair and would otherwise be one wave's
        // work while the other three wait
#pragma unroll
        for (int i = 0; i < IW; ++i) { const uint32_t v = acc[i][0] | (acc[i][1] << 16); if (v) atomicAdd(&red[i][lane], v); }
        __syncthreads();
        {
            uint32_t oj[JR];
#pragma unroll
            for (int c = 0; c < JR; ++c) oj[c] = a.sperm[c0 + lane + 64 * c];
            for (int i = ks * IW / KS; i < (ks + 1) * IW / KS; ++i) {
                const size_t k = k0 + i;
                const uint32_t rpos = full ? (uint32_t)k : a.rowpos[k];               // uniform
                if (rpos == SP_NONE || rpos >= a.N) continue;
                const uint32_t oi = a.sperm[rpos];                                    // uniform
#pragma unroll
                for (int c = 0; c < JR; ++c) {
                    const uint32_t mm = (red[i][lane] >> (16 * c)) & 0xFFFFu;
                    if (mm == a.S || oj[c] == SP_NONE) continue;
                    const bool want = full ? rpos < (uint32_t)(c0 + lane + 64 * c) : oj[c] > oi;
                    if (!want) continue;
                    const uint32_t lo = min(oi, oj[c]), hi = max(oi, oj[c]);
                    store.put(out_pos(sh, lo, hi), store.value_from_mismatches(a.S, mm));
                }
            }
        }
        __syncthreads();
    }
}

// gathered (caller-owned) operands carry the row coding + the unique plane only: derive the column coding.
// Done before EVERY launch on such a set -- the library cannot know when the caller re-gathered into the
// buffer, and the pass is ~2 % of the pair kernel it precedes.
int refresh_borrowed(d2g_ctx *ctx, const d2g_cmp_set *set, hipStream_t s) {
    if (!set->borrowed || set->managed) return D2G_OK;       // managed: the engine derived the stream as the groups arrived
    return d2g_bitslice_derive_groups(ctx, set, 0, set->ntb, s);
}

bool sparse_enabled(size_t N) {
    const char *e = std::getenv("D2G_BS_SPARSE");             // "0": every launch walks every tile (A/B measurements, tests)
    if (e && e[0] == '0') return false;
    size_t min_n = 8192;                                        // below ~6000 sketches the extra launches cost more than the tiles they skip (measured: N = 4096 0.19 vs 0.15 ms, N = 8192 0.31 vs 0.37 ms)
    if (const char *m = std::getenv("D2G_BS_SPARSE_MIN_N")) min_n = (size_t)std::atoll(m);
    return N >= 2 && N >= min_n;
}

int sp_alloc(d2g_ctx *ctx, d2g_cmp_set *set) {
    const size_t Npad = set->Npad, Nstride = set->Nstride;
    const size_t nrb = Npad / 32, ncb = Npad / BS_CB;
    set->tilebm_words = nrb * ((ncb + 31) / 32) + 1;
    set->tiles_cap = nrb * ncb;
    hipError_t e;
    if ((e = hipMalloc((void **)&set->d_stream_s, ((size_t)set->ntb * set->nbits_cap + 1) * 2 * Nstride * sizeof(uint32_t))) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_sperm, Nstride * 4)) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_sinv, Npad * 4)) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_label, 2 * Npad * 4)) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_lcnt, (Npad + 1) * 4)) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_rowpos, Nstride * 4)) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_rowk, Npad * 4)) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_spctl, (16 + set->tilebm_words) * 4)) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_gbm, (8 + set->tilebm_words) * 4)) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_slots, set->ncols * set->tilebm_words * 4)) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_tiles, std::max<size_t>(set->tiles_cap, 1) * 4)) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_order, 16)) != hipSuccess) {
        ctx->last_error = std::string("bitslice sparse alloc: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? D2G_ERR_NOMEM : D2G_ERR_HIP;
    }
    set->d_tilebm = set->d_spctl + 16;
    if ((e = hipMemset(set->d_spctl, 0, 16 * 4)) != hipSuccess) { ctx->last_error = std::string("bitslice sparse alloc: ") + hipGetErrorString(e); return D2G_ERR_HIP; }
    set->sp_launch = 0;
    return D2G_OK;
}

void sp_free(d2g_cmp_set *set) {
    for (uint32_t **p : {&set->d_stream_s, &set->d_sperm, &set->d_sinv, &set->d_label, &set->d_lcnt, &set->d_rowpos, &set->d_rowk,
                         &set->d_rowstream, &set->d_slots, &set->d_tiles, &set->d_spctl, &set->d_gbm, &set->d_order}) { (void)hipFree(*p); *p = nullptr; }
    set->d_tilebm = nullptr;
}

int sp_label_rounds() {
    int rounds = 1;
    if (const char *e = std::getenv("D2G_BS_LABEL_ROUNDS")) { const int v = std::atoi(e); if (v >= 0 && v <= 8) rounds = v; }
    return rounds;
}
bool sp_segments_on() { const char *se = std::getenv("D2G_SP_SEGMENTS"); return !(se && se[0] == '0'); }   // "0": always the exact marking (experiments, tests)
// what the kernel in front of sp_prepare_order initialises for it (and for the first launch after it: the global tile bitmap + its control words)
SpInit sp_init_of(const d2g_cmp_set *set) {
    SpInit si;
    si.label = set->d_label; si.cnt = set->d_lcnt; si.order = set->d_order; si.zero = set->d_gbm;
    si.n = (uint32_t)set->N; si.inexact = (sp_segments_on() && sp_label_rounds() > 0) ? 0u : 1u;
    si.zwords = (uint32_t)(8 + set->tilebm_words);
    return si;
}

// labels -> counting sort -> d_sperm / d_sinv.  All on `s`, no host round trip.
int sp_prepare_order(d2g_ctx *ctx, d2g_cmp_set *set, bool split, hipStream_t s) {
    const size_t N = set->N, Npad = set->Npad, S = set->ncols;
    const unsigned nb = (unsigned)div_up<size_t>(N, 256);
    const int rounds = sp_label_rounds();
    uint32_t *la = set->d_label, *lb = set->d_label + Npad;
    const bool segs = sp_segments_on() && rounds > 0;                    // (labels, counters, order[1] and the tile bitmap were initialised by the caller's kernel: sp_init_of)
    int gens = 4;                                                        // columns a workgroup walks one after the other
    if (const char *e = std::getenv("D2G_SP_PROP_GENS")) { const int v = std::atoi(e); if (v >= 1 && v <= 64) gens = v; }
    const uint32_t pcap = (uint32_t)std::min<size_t>(N / 2 + 1, 16384);
    auto propk = N <= 4096 ? sp_prop_reg_kernel<4> : N <= 10240 ? sp_prop_reg_kernel<10> : N <= 16384 ? sp_prop_reg_kernel<16> : sp_prop_kernel;
    D2G_HIP(ctx, hipFuncSetAttribute((const void *)propk, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 4));
    for (int r = 0; r < rounds; ++r) {
        hipLaunchKernelGGL(propk, dim3((unsigned)std::max<size_t>(1, div_up<size_t>(S, (size_t)gens))), dim3(1024), (size_t)pcap * 4, s, set->d_ids, N, Npad, (uint32_t)S,
                           set->d_colcnt, split ? 1 : 0, pcap, la);
        if (r + 1 < rounds) {                          // more rounds: the next one starts from the roots (the sort kernel hops by itself)
            hipLaunchKernelGGL(sp_jump_kernel, dim3(nb), dim3(256), 0, s, la, lb, N);
            hipLaunchKernelGGL(sp_jump_kernel, dim3(nb), dim3(256), 0, s, lb, la, N);
        }
    }
    if (rounds > 0) hipLaunchKernelGGL(sp_flatten_kernel, dim3(nb), dim3(256), 0, s, la, N, set->d_order);   // (also the guard against deep chains)
    if (segs) {
        const uint32_t cap = (uint32_t)std::min<size_t>(N / 2 + 1, 36864);               // shared values of a column: at most N / 2; 144 KB of LDS at most
        const unsigned cthreads = cap > 10240 ? 1024 : 256;                              // a big table leaves one workgroup per CU: a wide one
        D2G_HIP(ctx, hipFuncSetAttribute((const void *)sp_union_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 36864 * 4));
        hipLaunchKernelGGL(sp_union_kernel, dim3((unsigned)S), dim3(cthreads), (size_t)cap * 4, s, set->d_ids, N, Npad, set->d_colcnt, split ? 1 : 0, la, cap, set->d_order);
    }
    const size_t ntile_all = (Npad / 32) * (Npad / BS_CB);
    size_t seg_div = 8;                                                  // segments may cover an eighth of all tiles; beyond, exact marks are worth their pass
    if (const char *e = std::getenv("D2G_SP_SEG_DIV")) { const long v = std::atol(e); if (v >= 1 && v <= 1024) seg_div = (size_t)v; }
    const uint32_t seg_limit = (uint32_t)std::min<size_t>(ntile_all / seg_div, 0x3FFFFFFF);
    hipLaunchKernelGGL(sp_count_kernel, dim3(nb), dim3(256), 0, s, la, lb, N, set->d_lcnt, set->d_order);
    hipLaunchKernelGGL(sp_scan_kernel, dim3(1), dim3(1024), 0, s, set->d_lcnt, N, set->d_order, la, seg_limit);    // la (labels) is dead after the count kernel: it keeps the segment starts
    hipLaunchKernelGGL(sp_place_kernel, dim3((unsigned)div_up<size_t>(set->Nstride, 256)), dim3(256), 0, s, lb, N, set->Nstride, set->d_lcnt, set->d_sperm, set->d_sinv, set->d_order);
    D2G_HIP(ctx, hipGetLastError());
    return D2G_OK;
}

// ids of an operand that arrived as bit planes (the multi-GPU engine's gathered operand: ranks exchange planes, not ids): the inverse
// of bs_planes_kernel's bit transpose.  Every register SLOT of the operand is a column here (slot 32 tb + x = whatever column the
// preparing rank's plan put there; padding slots hold id 0 everywhere).  colcnt[slot][4] = the number of the slot's shared values
// (carried by the slack words of the group's unique plane: bs_planes_kernel).
__global__ __launch_bounds__(256) void sp_unpack_kernel(const uint32_t *__restrict__ planes, size_t Nstride, int nbits_cap, const uint32_t *__restrict__ meta,
                                                        size_t N, size_t Npad, uint32_t *__restrict__ ids, uint32_t *__restrict__ colcnt, SpInit si) {
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;          // < Npad: the grid covers Npad exactly
    const size_t tb = blockIdx.y;
    sp_init_part(si, tb * ((size_t)gridDim.x * 256) + j, (size_t)gridDim.x * 256 * gridDim.y);
    const int nbits = live_planes(meta, (int)tb);
    const uint32_t *src = planes + tb * (size_t)(nbits_cap + 1) * Nstride + j;
    if (j < 32) colcnt[(tb * 32 + j) * BS_CC_STRIDE + 4] = planes[tb * (size_t)(nbits_cap + 1) * Nstride + (size_t)nbits_cap * Nstride + Npad + j];
    uint32_t id[32];
#pragma unroll
    for (int x = 0; x < 32; ++x) id[x] = 0;
    if (j < N) {
        for (int b = 0; b < nbits; ++b) {
            const uint32_t w = src[(size_t)b * Nstride];
#pragma unroll
            for (int x = 0; x < 32; ++x) id[x] |= ((w >> x) & 1u) << b;
        }
        const uint32_t u = src[(size_t)nbits_cap * Nstride];
#pragma unroll
        for (int x = 0; x < 32; ++x) if ((u >> x) & 1u) id[x] = BS_UNIQ;
    }
#pragma unroll
    for (int x = 0; x < 32; ++x) ids[(tb * 32 + x) * Npad + j] = id[x];
}

template <class Store>
int launch_sparse(d2g_ctx *ctx, const d2g_cmp_set *cset, PairShape sh, Store store, uint32_t *out_words, hipStream_t s) {
    d2g_cmp_set *set = const_cast<d2g_cmp_set *>(cset);
    const size_t N = set->N, Npad = set->Npad, r0 = sh.i_lo, r1 = sh.i_hi;
    if (r1 <= r0) return D2G_OK;
    const bool full = r0 == 0 && r1 == N;
    const size_t nrows = r1 - r0, nrows_pad = full ? Npad : div_up<size_t>(nrows, 32) * 32;
    const uint32_t nrb = (uint32_t)(nrows_pad / 32), ncb = (uint32_t)(Npad / BS_CB);
    const bool split = !set->borrowed && set->logT > BS_LOG_TLDS_MAX && set->nsplit > 1;
    const size_t cnt = d2g_ut_count(N, r0, r1);
    if (!cnt) return D2G_OK;
    if (!full && !set->d_rowstream)
        D2G_HIP(ctx, hipMalloc((void **)&set->d_rowstream, ((size_t)set->ntb * set->nbits_cap + 1) * set->Nstride * sizeof(uint32_t)));
    PairShape dsh = sh;                                                                 // the dense walk of the same launch, behind the gate
    if (int rc = finish_shape(ctx, dsh, BS_JR == 2 ? 32u : 64u)) return rc;
    d2g_timer tm(ctx, &ctx->ev_k2, s);
    const uint32_t CW = (ncb + 31) / 32;                                                // words of a bitmap row (column blocks)
    const uint32_t nrbG = (uint32_t)(Npad / 32);                                        // all sorted row blocks
    if (!set->gbm_valid) {
        // ONCE per prepare: the tiles that hold a pair with a shared value, over all sorted positions (d_gbm: 8 control words + bitmap)
        const uint32_t RW = (nrbG + 31) / 32, W = RW + CW;
        // (d_gbm -- control words + bitmap -- was cleared by the prepare: sp_init_of)
        // LDS: the column's copy of the tile bitmap (when it is small) + the bit sets; 36 KB in all: four columns per CU
        uint32_t lbm_words = nrbG * CW;
        int lbm_max = 4096;                                                            // words (16 KB)
        if (const char *e = std::getenv("D2G_SP_LOCALBM")) lbm_max = std::atoi(e);       // experiments: 0 = build it in the slot
        if ((int)lbm_words > lbm_max) lbm_words = 0;
        // (36 KB, not 40: with the kernel's few static bytes on top a 40 KB request fits only three times into the CU's 160 KB.)  Wide bit
        // sets (N above ~20 000: 56 words per value at N = 50 000) take 76 KB and 512 threads, two columns per CU: fewer passes over the sketches
        const bool wide = W > 24;
        const uint32_t budget = wide ? 19456u : 9216u;
        const uint32_t gm = std::max(1u, std::min(2048u, (budget - lbm_words) / W));
        const size_t lds = ((size_t)gm * W + lbm_words) * 4;
        // order[1] == 0 (the prepare united every shared value's holders: the sort's segments are the families): the mark kernel's threads set the
        // segments' tiles and the folding kernel returns at once
        auto mark = wide ? sp_mark_kernel<D2G_SP_WIDE_U> : sp_mark_kernel<8>;
        D2G_HIP(ctx, hipFuncSetAttribute((const void *)mark, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        hipLaunchKernelGGL(mark, dim3((unsigned)set->ncols), dim3(wide ? 512 : 256), lds, s, set->d_ids, N, Npad, set->d_colcnt, split ? 1 : 0, set->d_sinv,
                           (const uint32_t *)nullptr, gm, RW, CW, nrbG, ncb, lbm_words, set->d_slots, set->d_gbm, set->d_order, set->d_sperm, set->d_label + Npad, set->d_label,
                           set->d_lcnt, set->d_gbm + 8);
        hipLaunchKernelGGL(sp_or_kernel, dim3((unsigned)div_up<size_t>((size_t)nrbG * CW, 256), SP_OR_SPLIT), dim3(256), 0, s, set->d_slots, nrbG * CW, (uint32_t)set->ncols,
                           set->d_gbm + 8, set->d_gbm, set->d_order);
        set->gbm_valid = true;
    }
    // per launch: 8 control words (ctl[0] = tiles listed, ctl[1] = flags (bit 0 ALL: marking gave up), [3] = candidates) + a partial launch's bitmap
    // double-buffered: this launch's list kernel clears the other set for the next launch (both start cleared: sp_alloc)
    uint32_t *const ctl = set->d_spctl + 8 * (set->sp_launch & 1u), *const ctl_next = set->d_spctl + 8 * ((set->sp_launch + 1) & 1u);
    ++set->sp_launch;
    if (!full) {
        hipLaunchKernelGGL(sp_rows_kernel, dim3(1), dim3(1024), 0, s, set->d_sperm, N, (uint32_t)r0, (uint32_t)r1, (uint32_t)nrows_pad, set->d_rowpos, set->d_rowk);
        hipLaunchKernelGGL(sp_gather_kernel, dim3((unsigned)div_up<size_t>(nrows_pad, 256), (unsigned)(set->ntb * set->nbits_cap)), dim3(256), 0, s,
                           set->d_stream_s, set->Nstride, set->d_meta, set->ntb, set->d_rowpos, (uint32_t)nrows_pad, set->d_rowstream, set->Nstride);
        hipLaunchKernelGGL(sp_rowbm_kernel, dim3((unsigned)div_up<size_t>((size_t)nrb * CW, 256)), dim3(256), 0, s, set->d_gbm + 8, set->d_rowpos, nrb, CW, set->d_tilebm);
    }
    const size_t ntile = (size_t)nrb * ncb;
    // candidates: every tile of a partial launch; the tiles on or above the diagonal of sorted positions of a full one
    size_t cand = ntile;
    if (full) { cand = 0; for (uint32_t cb = 0; cb < ncb; ++cb) cand += std::min<size_t>(nrb, ((size_t)cb * 256 + 255) / 32 + 1); }
    const uint32_t cand32 = (uint32_t)std::min<size_t>(cand, 0xFFFFFFFFu);
    hipLaunchKernelGGL(sp_list_kernel, dim3((unsigned)div_up<size_t>(ntile, 8192)), dim3(1024), 0, s, full ? set->d_gbm + 8 : set->d_tilebm, nrb, ncb, CW, full ? 1 : 0,
                       set->d_tiles, ctl, cand32, set->d_gbm, ctl_next);
    SpArgs a{set->d_stream_s, set->Nstride, full ? (const uint32_t *)nullptr : set->d_rowstream, set->Nstride, set->d_meta, set->ntb, (uint32_t)set->S, (uint32_t)N,
             set->d_sperm, set->d_rowpos, set->d_tiles, ctl, ncb, cand32};
    // contiguous 32 KB per workgroup, workgroups in dispatch order: a streaming write (6.1 TB/s at N = 50 000: 825 us; the grid-stride loop over 16
    // workgroups per CU it replaces, whose iterations lie 16 MB apart, reached 4.6: 1105 us).  One store per thread is faster still (722-760 us) but when
    // the launch turns out dense all of its 19 M waves start only to return: 254 us instead of 34
    hipLaunchKernelGGL((sp_fill_kernel<Store>), dim3((unsigned)std::min<size_t>(div_up<size_t>(cnt / 4 + 1, SP_FILL_THREADS * SP_FILL_PER_THREAD), (size_t)0x7FFFFFFF)), dim3(SP_FILL_THREADS), 0, s,
                       out_words, cnt, store, (uint32_t)set->S, ctl, cand32);
    const unsigned grid = (unsigned)std::max<size_t>(1, std::min<size_t>(ntile * 4, (size_t)ctx->num_cus * (28 / D2G_SP_KS)));
    hipLaunchKernelGGL((k2_bitslice_sparse_kernel<BS_JR, Store>), dim3(grid), dim3(64 * D2G_SP_KS), 0, s, a, sh, store);
    if (dsh.nvalid_total)
        hipLaunchKernelGGL((k2_bitslice_kernel<BS_JR, Store>), dim3(dsh.per_xcd * 8), dim3(BS_THREADS), 0, s, set->d_stream,
                           set->Nstride, set->d_meta, set->ntb, (uint32_t)set->S, dsh, store, (const uint32_t *)ctl, cand32);
    tm.stop();
    D2G_HIP(ctx, hipGetLastError());
    return D2G_OK;
}

template <class Store>
int launch_bitslice(d2g_ctx *ctx, const d2g_cmp_set *set, PairShape sh, Store store, hipStream_t s) {
    if (int rc = finish_shape(ctx, sh, BS_JR == 2 ? 32u : 64u)) return rc;   // workgroup tile = (16 * JR) rows x 256 columns (4 waves of 16 x 64*JR)
    if (sh.nvalid_total == 0) return D2G_OK;
    if (int rc = refresh_borrowed(ctx, set, s)) return rc;
    if (int rc = d2g_bitslice_ensure_natural(ctx, set, s)) return rc;
    d2g_timer tm(ctx, &ctx->ev_k2, s);
    hipLaunchKernelGGL((k2_bitslice_kernel<BS_JR, Store>), dim3(sh.per_xcd * 8), dim3(BS_THREADS), 0, s, set->d_stream,
                       set->Nstride, set->d_meta, set->ntb, (uint32_t)set->S, sh, store, (const uint32_t *)nullptr, 0u);
    tm.stop();
    D2G_HIP(ctx, hipGetLastError());
    return D2G_OK;
}

}  // namespace

void d2g_bitslice_free(d2g_cmp_set *set) {
    if (!set) return;
    if (!set->borrowed) { (void)hipFree(set->d_planes); (void)hipFree(set->d_meta); }
    (void)hipFree(set->d_stream);
    (void)hipFree(set->d_ids);
    (void)hipFree(set->d_colcnt);
    (void)hipFree(set->d_perm);
    set->d_planes = set->d_stream = set->d_meta = set->d_ids = set->d_colcnt = set->d_perm = nullptr;
    sp_free(set);
}

// geometry of the bit-sliced operand: a function of N (and S) only, identical on every rank.
// A column holds at most floor(N/2) values that occur twice; ranks 1..D2 plus the two codes of "unique"
// (0 and 2^nbits - 1) need 2^nbits >= D2 + 2.
void d2g_bitslice_geometry(d2g_cmp_set *set) {
    set->nbits_cap = 1;
    while ((1ull << set->nbits_cap) < set->N / 2 + 2) ++set->nbits_cap;
    set->ntb = (int)div_up<size_t>(set->S, 32);
    set->Nstride = set->Npad + BS_SLACK;
}

// the plane stream: at most nbits_cap live planes per group, two codings each, + one block of slack (the
// kernel's prefetch runs one plane past the end)
int d2g_bitslice_alloc_stream(d2g_ctx *ctx, d2g_cmp_set *set) {
    hipError_t e = hipMalloc((void **)&set->d_stream, ((size_t)set->ntb * set->nbits_cap + 1) * 2 * set->Nstride * sizeof(uint32_t));
    if (e != hipSuccess) {
        ctx->last_error = std::string("bitslice alloc: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? D2G_ERR_NOMEM : D2G_ERR_HIP;
    }
    return D2G_OK;
}

namespace {
// workspace every preparing set needs (ids, per-column counts, column plan, meta + status); the rank kernel's hash space
int alloc_prepare_workspace(d2g_ctx *ctx, d2g_cmp_set *set) {
    const size_t N = set->N, S = set->S, Npad = set->Npad;
    if (N >= (1ull << 30)) { ctx->last_error = "bitslice: N too large"; return D2G_ERR_UNSUPPORTED; }
    d2g_bitslice_geometry(set);
    // hash space: power of two >= 1.5 N (load <= 2/3), at least 64 slots; walked in LDS-sized partitions
    set->T = 64; set->logT = 6;
    while ((uint64_t)set->T * 2 < (uint64_t)N * 3) { set->T <<= 1; ++set->logT; }
    // narrow slices of large N: several workgroups per column (each walks its share of the hash partitions) until the CUs are covered
    set->nsplit = 1;
    if (set->logT > BS_LOG_TLDS_MAX && N < (1ull << BS_SPLIT_SHIFT)) {
        const uint32_t nparts = set->T >> BS_LOG_TLDS_MAX;
        int want = 1;
        while (want < 4 && (uint32_t)want * 2 <= nparts && S * (size_t)want < (size_t)std::max(ctx->num_cus, 1)) want *= 2;
        if (const char *e = std::getenv("D2G_BS_NSPLIT")) {                 // tests / experiments
            const int v = std::atoi(e);
            if ((v == 1 || v == 2 || v == 4) && (uint32_t)v <= nparts) want = v;
        }
        set->nsplit = want;
    }
    hipError_t e;
    if ((e = hipMalloc((void **)&set->d_ids, S * Npad * sizeof(uint32_t))) != hipSuccess ||
        (e = hipMemset(set->d_ids, 0, S * Npad * sizeof(uint32_t))) != hipSuccess ||     // split rank passes rely on "no stale pending word"
        (e = hipMalloc((void **)&set->d_colcnt, S * BS_CC_STRIDE * sizeof(uint32_t))) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_perm, (size_t)set->ntb * 32 * sizeof(uint32_t))) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_meta, (size_t)(set->ntb + 4) * sizeof(uint32_t))) != hipSuccess ||
        (e = hipMemset(set->d_meta, 0, (size_t)(set->ntb + 4) * sizeof(uint32_t))) != hipSuccess) {
        ctx->last_error = std::string("bitslice alloc: ") + hipGetErrorString(e);
        d2g_bitslice_free(set);
        return e == hipErrorOutOfMemory ? D2G_ERR_NOMEM : D2G_ERR_HIP;
    }
    return D2G_OK;
}
bool sort_columns() {
    const char *e = std::getenv("D2G_BS_SORT");       // "0": keep the caller's column order (A/B measurements, tests)
    return !(e && e[0] == '0');
}
}  // namespace

// one-time allocation of the bit-sliced operand and its workspace
int d2g_bitslice_alloc(d2g_ctx *ctx, d2g_cmp_set *set) {
    if (int rc = alloc_prepare_workspace(ctx, set)) return rc;
    hipError_t e;
    if ((e = hipMalloc((void **)&set->d_planes, (size_t)set->ntb * (set->nbits_cap + 1) * set->Nstride * sizeof(uint32_t))) != hipSuccess) {
        ctx->last_error = std::string("bitslice alloc: ") + hipGetErrorString(e);
        d2g_bitslice_free(set);
        return e == hipErrorOutOfMemory ? D2G_ERR_NOMEM : D2G_ERR_HIP;
    }
    if (int rc = d2g_bitslice_alloc_stream(ctx, set)) { d2g_bitslice_free(set); return rc; }
    set->ncols = set->S;
    set->sparse_ok = sparse_enabled(set->N) && set->S < 65536 && set->S * (set->Npad / 32) * ((set->Npad / BS_CB + 31) / 32) * 4 <= ((size_t)1 << 30);   // the columns' tile bitmaps: <= 1 GiB
    if (set->sparse_ok) if (int rc = sp_alloc(ctx, set)) { d2g_bitslice_free(set); return rc; }
    return D2G_OK;
}

// ids + column plan + planes for the operand currently in set->d_cols.  Fully asynchronous on `s`.
// meta[0..ntb) = per-group shared-value counts; meta[ntb] = status word (bit 0: the rank kernel's LDS table
// overflowed on some column -- see d2g_bitslice_status)
int d2g_bitslice_prepare(d2g_ctx *ctx, d2g_cmp_set *set, hipStream_t s) {
    const size_t N = set->N, S = set->S, Npad = set->Npad;
    // the status word meta[ntb] was zeroed by the transpose kernel that filled d_cols (no memset node in the chain)
    {
        const int logTl = set->logT < BS_LOG_TLDS_MAX ? set->logT : BS_LOG_TLDS_MAX;
        const size_t lds = (size_t(1) << logTl) * sizeof(uint32_t);
        void (*kern)(const uint64_t *, size_t, size_t, uint32_t, int, uint32_t *, uint32_t *, uint32_t *, int, uint32_t, int) = bs_rank_kernel<false, false>;
        const bool multi = set->logT > BS_LOG_TLDS_MAX;
        if (multi) kern = bs_rank_kernel<true, false>;
        else if (N <= (size_t)12 * BS_RANK_THREADS) kern = bs_rank_kernel<false, true>;     // PF * BS_RANK_THREADS
        int tagbits_max = 31;
        if (const char *e = std::getenv("D2G_BS_TAGBITS")) { const int v = std::atoi(e); if (v >= 0 && v < 31) tagbits_max = v; }   // tests
        if (lds > 48 * 1024)
            D2G_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int nsplit = multi ? set->nsplit : 1;
        // split passes tell their pending words from final ids by a tag and trust that ids[] holds nothing else that looks pending.  A
        // prepare that overflowed (status bit 0) may leave pending words behind, and the multi-GPU engine re-uses its exporter sets step
        // after step: ids[] starts from zero in every split prepare -- a few MB for the narrow column slices that are split
        if (nsplit > 1) D2G_HIP(ctx, hipMemsetAsync(set->d_ids, 0, S * Npad * sizeof(uint32_t), s));
        hipLaunchKernelGGL(kern, dim3((unsigned)(S * nsplit)), dim3(BS_RANK_THREADS), lds, s, set->d_cols, N, Npad, set->T, set->logT,
                           set->d_ids, set->d_colcnt, set->d_meta + set->ntb, tagbits_max, (uint32_t)S, nsplit);
        hipLaunchKernelGGL(bs_colplan_kernel, dim3(1), dim3(BS_PLAN_THREADS), 0, s, set->d_colcnt, (uint32_t)S, set->ntb, nsplit, set->d_perm,
                           set->d_meta, set->d_meta + set->ntb, set->ex_meta, set->ex_status, sort_columns() ? 1 : 0);
    }
    dim3 grid((unsigned)div_up<size_t>(set->Nstride, 256), (unsigned)set->ntb);
    const bool split = set->logT > BS_LOG_TLDS_MAX && set->nsplit > 1;
    if (set->sparse_ok && !set->export_only && !set->want_exchange) {
        // sparse path: the stream in label order (section 4); the caller's-order stream is only built if a launch asks for it
        // (gathering the ids through d_sperm inside bs_planes_kernel was measured: 74 us instead of 18 at config 3 -- 1024 columns of
        // uncoalesced 4-byte loads; permuting the finished stream touches 256 rows of words and leaves the caller's-order stream valid)
        // (a second queue for the caller's-order planes beside the labelling, and for the fill beside the marking, was measured: the
        // kernels slow each other down by what the overlap hides -- mark 28 -> 57 us next to the fill -- and the events cost more: dropped)
        // (the planes kernel also initialises the ordering's arrays and clears the tile bitmap of the first launch: no launch / memset of their own)
        hipLaunchKernelGGL(split ? bs_planes_kernel<true> : bs_planes_kernel<false>, grid, dim3(256), 0, s, set->d_ids, N, Npad,
                           set->d_planes, set->d_stream, set->Nstride, set->nbits_cap, set->d_meta, BS_FORM_STREAM, set->d_perm, set->d_colcnt, (const uint32_t *)nullptr, sp_init_of(set));
        if (int rc = sp_prepare_order(ctx, set, split, s)) return rc;
        hipLaunchKernelGGL(sp_permute_kernel, grid, dim3(256), 0, s, set->d_stream, set->d_stream_s, set->Nstride, set->d_meta, set->d_sperm, set->d_order);
        set->srt_valid = true; set->nat_valid = true; set->gbm_valid = false;
        D2G_HIP(ctx, hipGetLastError());
        return D2G_OK;
    }
    const int forms = set->export_only ? BS_FORM_EXCHANGE : (BS_FORM_STREAM | (set->want_exchange ? BS_FORM_EXCHANGE : 0));
    hipLaunchKernelGGL(split ? bs_planes_kernel<true> : bs_planes_kernel<false>, grid, dim3(256), 0, s, set->d_ids, N, Npad,
                       set->export_only ? set->ex_planes : set->d_planes, set->d_stream, set->Nstride, set->nbits_cap, set->d_meta, forms,
                       set->d_perm, set->d_colcnt, (const uint32_t *)nullptr, SpInit{});
    set->srt_valid = false; set->nat_valid = true;
    D2G_HIP(ctx, hipGetLastError());
    return D2G_OK;
}

// the caller's-order stream of a set whose last prepare only wrote the sorted one (rectangular launches, exports)
int d2g_bitslice_ensure_natural(d2g_ctx *ctx, const d2g_cmp_set *cset, hipStream_t s) {
    d2g_cmp_set *set = const_cast<d2g_cmp_set *>(cset);
    if (set->nat_valid || set->borrowed || set->export_only) return D2G_OK;
    dim3 grid((unsigned)div_up<size_t>(set->Nstride, 256), (unsigned)set->ntb);
    const bool split = set->logT > BS_LOG_TLDS_MAX && set->nsplit > 1;
    hipLaunchKernelGGL(split ? bs_planes_kernel<true> : bs_planes_kernel<false>, grid, dim3(256), 0, s, set->d_ids, set->N, set->Npad,
                       set->d_planes, set->d_stream, set->Nstride, set->nbits_cap, set->d_meta, BS_FORM_STREAM, set->d_perm, set->d_colcnt, (const uint32_t *)nullptr, SpInit{});
    D2G_HIP(ctx, hipGetLastError());
    set->nat_valid = true;
    return D2G_OK;
}

// the exchange form of the operand last prepared: written from the ids on first request, by every prepare afterwards
int d2g_bitslice_export(d2g_ctx *ctx, d2g_cmp_set *set, hipStream_t s) {
    if (set->borrowed || set->want_exchange || set->export_only) return D2G_OK;
    set->want_exchange = true;
    dim3 grid((unsigned)div_up<size_t>(set->Nstride, 256), (unsigned)set->ntb);
    const bool split = set->logT > BS_LOG_TLDS_MAX && set->nsplit > 1;
    hipLaunchKernelGGL(split ? bs_planes_kernel<true> : bs_planes_kernel<false>, grid, dim3(256), 0, s, set->d_ids, set->N, set->Npad,
                       set->d_planes, set->d_stream, set->Nstride, set->nbits_cap, set->d_meta, BS_FORM_EXCHANGE, set->d_perm, set->d_colcnt, (const uint32_t *)nullptr, SpInit{});
    D2G_HIP(ctx, hipGetLastError());
    return D2G_OK;
}

// ---- exporter sets: the multi-GPU engine's per-rank column slices (d2g_mgpu.hip)
int d2g_bitslice_exporter_create(d2g_ctx *ctx, size_t N, size_t S_local, d2g_cmp_set **out) {
    *out = nullptr;
    d2g_cmp_set *set = new (std::nothrow) d2g_cmp_set();
    if (!set) return D2G_ERR_NOMEM;
    set->ctx = ctx; set->N = N; set->S = S_local;
    set->Npad = div_up<size_t>(N, BS_CB) * BS_CB;
    set->algo = D2G_CMP_BITSLICE;
    set->export_only = true;
    hipError_t e = hipMalloc((void **)&set->d_cols, set->Npad * S_local * sizeof(uint64_t));
    if (e != hipSuccess) { ctx->last_error = std::string("bitslice exporter alloc: ") + hipGetErrorString(e); delete set; return D2G_ERR_NOMEM; }
    if (int rc = alloc_prepare_workspace(ctx, set)) { (void)hipFree(set->d_cols); delete set; return rc; }
    *out = set;
    return D2G_OK;
}
void d2g_bitslice_set_export_target(d2g_cmp_set *set, uint32_t *planes, uint32_t *meta, uint32_t *status) {
    set->ex_planes = planes; set->ex_meta = meta; set->ex_status = status;
}

int d2g_bitslice_derive_groups(d2g_ctx *ctx, const d2g_cmp_set *set, int g0, int g1, hipStream_t s) {
    if (g1 <= g0) return D2G_OK;
    dim3 grid((unsigned)div_up<size_t>(set->Nstride, 256), (unsigned)(g1 - g0));
    hipLaunchKernelGGL(bs_derive_kernel, grid, dim3(256), 0, s, set->d_planes, set->d_stream, set->Nstride, set->nbits_cap, set->d_meta, g0);
    D2G_HIP(ctx, hipGetLastError());
    return D2G_OK;
}

// synchronises `s`; D2G_ERR_INTERNAL when the last prepare overflowed its hash partitions
int d2g_bitslice_status(d2g_ctx *ctx, const d2g_cmp_set *set, hipStream_t s) {
    uint32_t st = 0;
    if (set->borrowed) {
        // a gathered operand has no status word of its own; the multi-GPU engine's carries one per preparing rank and chunk
        if (!set->status_words || set->n_status <= 0) { D2G_HIP(ctx, hipStreamSynchronize(s)); return D2G_OK; }
        std::vector<uint32_t> w((size_t)set->n_status);
        D2G_HIP(ctx, hipMemcpyAsync(w.data(), set->status_words, w.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        D2G_HIP(ctx, hipStreamSynchronize(s));
        for (uint32_t x : w) st |= x;
    } else {
        D2G_HIP(ctx, hipMemcpyAsync(&st, set->d_meta + set->ntb, sizeof(st), hipMemcpyDeviceToHost, s));
        D2G_HIP(ctx, hipStreamSynchronize(s));
    }
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
    if (set->srt_valid) {
        if (eq_out) return launch_sparse(ctx, set, sh, StoreEq{eq_out}, eq_out, s);
        return launch_sparse(ctx, set, sh, StoreLut{fout, lut}, reinterpret_cast<uint32_t *>(fout), s);
    }
    if (eq_out) return launch_bitslice(ctx, set, sh, StoreEq{eq_out}, s);
    return launch_bitslice(ctx, set, sh, StoreLut{fout, lut}, s);
}

int d2g_bitslice_rect(d2g_ctx *ctx, const d2g_cmp_set *set, size_t a0, size_t a1, size_t b0, size_t b1, uint32_t *eq_out,
                      hipStream_t s) {
    PairShape sh{};
    sh.N = set->N; sh.i_lo = a0; sh.i_hi = a1; sh.j_lo = b0; sh.j_hi = b1; sh.ut = 0;
    return launch_bitslice(ctx, set, sh, StoreEq{eq_out}, s);
}

// ---- sparse tiles on the multi-GPU engine's gathered operand (replicated: every rank orders and marks the whole operand itself --
// the pair phase of a block-structured matrix shrinks by the fraction of tiles listed, the fixed costs stay per rank)
int d2g_bitslice_managed_sparse_alloc(d2g_ctx *ctx, d2g_cmp_set *set) {
    if (!set->borrowed || set->sparse_ok) return D2G_OK;
    set->ncols = (size_t)set->ntb * 32;
    if (!(sparse_enabled(set->N) && set->S < 65536 && set->ncols * (set->Npad / 32) * ((set->Npad / BS_CB + 31) / 32) * 4 <= ((size_t)1 << 30))) return D2G_OK;
    hipError_t e;
    if ((e = hipMalloc((void **)&set->d_ids, set->ncols * set->Npad * sizeof(uint32_t))) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_colcnt, set->ncols * BS_CC_STRIDE * sizeof(uint32_t))) != hipSuccess ||
        (e = hipMemset(set->d_colcnt, 0, set->ncols * BS_CC_STRIDE * sizeof(uint32_t))) != hipSuccess) {
        ctx->last_error = std::string("bitslice sparse alloc (gathered operand): ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? D2G_ERR_NOMEM : D2G_ERR_HIP;
    }
    set->ids_owned = true;
    if (int rc = sp_alloc(ctx, set)) return rc;
    set->sparse_ok = true;
    return D2G_OK;
}

int d2g_bitslice_managed_ready(d2g_ctx *ctx, d2g_cmp_set *set, hipStream_t s) {
    if (!set->borrowed || !set->sparse_ok) return D2G_OK;
    dim3 grid((unsigned)div_up<size_t>(set->Npad, 256), (unsigned)set->ntb);
    hipLaunchKernelGGL(sp_unpack_kernel, grid, dim3(256), 0, s, set->d_planes, set->Nstride, set->nbits_cap, set->d_meta, set->N, set->Npad, set->d_ids, set->d_colcnt, sp_init_of(set));
    if (int rc = sp_prepare_order(ctx, set, false, s)) return rc;
    dim3 pgrid((unsigned)div_up<size_t>(set->Nstride, 256), (unsigned)set->ntb);
    hipLaunchKernelGGL(sp_permute_kernel, pgrid, dim3(256), 0, s, set->d_stream, set->d_stream_s, set->Nstride, set->d_meta, set->d_sperm, set->d_order);
    D2G_HIP(ctx, hipGetLastError());
    set->srt_valid = true; set->gbm_valid = false;
    return D2G_OK;
}

// diagnostics of the sparse path's LAST launch on this set (synchronises `s`): see d2g.h
int d2g_bitslice_sparse_info(d2g_ctx *ctx, const d2g_cmp_set *set, hipStream_t s, uint32_t *out4) {
    out4[0] = out4[1] = out4[2] = out4[3] = 0;
    if (!set->srt_valid || !set->d_spctl) { D2G_HIP(ctx, hipStreamSynchronize(s)); return D2G_OK; }
    uint32_t c[4] = {0, 0, 0, 0}, ord[2] = {0, 1}, g[4] = {0, 0, 0, 0};
    D2G_HIP(ctx, hipMemcpyAsync(c, set->d_spctl + 8 * ((set->sp_launch + 1) & 1u), sizeof c, hipMemcpyDeviceToHost, s));   // the set the last launch used
    D2G_HIP(ctx, hipMemcpyAsync(ord, set->d_order, sizeof ord, hipMemcpyDeviceToHost, s));
    D2G_HIP(ctx, hipStreamSynchronize(s));
    c[2] = ord[0]; g[2] = ord[1] ? 1u : 0u;
    // [2]: bit 0 marking gave up, bit 1 the dense kernel ran, bit 2 the tiles came from the sort's segments (no marking pass)
    out4[0] = 1; out4[1] = c[0]; out4[2] = (c[1] & 1u) | (((c[1] & 1u) || (size_t)c[0] * 5 > (size_t)c[3] * 2) ? 2u : 0u) | ((g[2] & 1u) ? 0u : 4u); out4[3] = c[2];
    return D2G_OK;
}

void d2g_warm_k2_bitslice() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&bs_colplan_kernel)); }

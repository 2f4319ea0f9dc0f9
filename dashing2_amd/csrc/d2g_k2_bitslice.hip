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
#include <chrono>
#include <cstring>
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
// (Measured in round 5 and dropped: a MULTI kernel that reads the column ONCE -- a thread keeps the high word of v * K of its <= 49 values
// in registers, every partition pass claims / confirms / ranks from them and writes final ids, no pending words -- for N <= 50 176.
// 128 VGPRs with ~320 spills, and the confirm step still fetches one owner value per candidate, four candidates in flight per thread:
// 2.58 ms instead of 1.04 at N = 50 000, 1.00 instead of 0.36 at N = 30 000.  What bounds this kernel at that size is not the repeated
// column read but the ~N gathers of 8 bytes per column that decide equality against the owner's value (51 million 64-byte sectors).
// Also measured and dropped: the values of a pass's partition COMPACTED into a queue in the LDS behind the table (block scan of the
// per-thread counts, 2560 entries of (value, sketch)) so that claim / confirm run with every lane busy instead of under an execution
// mask -- correct (all K2 / multi-GPU tests), 1.08 ms instead of 1.04 at N = 50 000, 0.43 instead of 0.36 at N = 30 000: the scan, the
// queue traffic and four more barriers per batch cost what the idle lanes had cost.
// And: the column BINNED by partition in global memory first (a kernel of its own: per-wave counts by ballot, then (value, sketch) runs at
// per-wave cursors; 12 bytes per value), then ranked bin by bin from the dense lists with 16384-slot tables -- correct, but the binning
// kernel alone took 0.54 ms at N = 50 000 (0.23 at 30 000) and the bin-ranking kernel 0.62 (0.36): 1.16 ms against 1.04.  The walking
// kernel stays; what it costs is ~84 dependent round trips per workgroup (three per batch of eight values and pass) with one 128 KiB-table
// workgroup per CU to hide them.)
constexpr uint32_t BS_SPLIT_SHIFT = 28, BS_RANK_MASK = (1u << BS_SPLIT_SHIFT) - 1u;
#ifdef D2G_RANK_TRACE
// variant builds only (tools/build_variant.sh ranktrace -DD2G_RANK_TRACE; tools/rank_trace.py): per-workgroup time stamps of the rank kernel --
// FAST: 0 start, 1 values loaded + table cleared, 2 claims done, 3 confirms done, 4 compaction done, 5 ids stored;
// general: 0 start, then per partition pass p (first four): 1+2p walk (load, claim, confirm, pending ids) done, 2+2p compaction + final ids done
__device__ unsigned long long g_rank_trace[8192 * 16];
#define RK_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 8192 && (k) < 16) g_rank_trace[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RK_STAMP(k) do { } while (0)
#endif
template <bool MULTI, bool FAST>
__global__ __launch_bounds__(BS_RANK_THREADS) void bs_rank_kernel(const uint64_t *__restrict__ cols, size_t N, size_t Npad,
                                                                  uint32_t T, int logT, uint32_t *ids_all, uint32_t *colcnt,
                                                                  uint32_t *status, int tagbits_max, uint32_t S, int nsplit,
                                                                  uint32_t *__restrict__ owner_all, size_t ostride) {
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
    RK_STAMP(0);
    if (tid == 0) { running = 1; nfix = 0; }                             // id 0 is reserved for singletons
    const int lane = tid & 63, wave = tid >> 6;
    // owner_all (single-partition kernels of sets that take the sparse path): owner[t][r - 1] = the sketch that owns the slot of the value
    // ranked r -- ONE holder of every shared value, for free: the sparse path's link passes (sp_olink_kernel) compare every holder with its
    // value's owner instead of grouping the holders in LDS again
    uint32_t *owner = (owner_all && nsplit == 1) ? owner_all + t * ostride : nullptr;      // (one workgroup per column: `running` numbers the column's shared values 1 .. D2)
    const int ib0 = 32 - __clz((uint32_t)N);
    const uint32_t ownmask = (ib0 >= 32 ? 0xFFFFFFFFu : (1u << ib0) - 1u) & ~BS_DUP;

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
            if (cur != BS_EMPTY) {
                if ((cur & BS_DUP) && owner) owner[r - 1] = cur & ownmask;
                own[h] = (cur & BS_DUP) ? (r++ | htag) : BS_UNIQ;
            }
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
        RK_STAMP(1);
        uint32_t candidate = 0;                                           // bit i: value i stopped at a tag match
        // (round 6: per-workgroup time stamps -- tools/rank_trace.py, profiles/r06_rank_trace.txt -- put this claim phase at 10.7 of the workgroup's 21 us at
        // config 3 (6.1 of 15.7 on unrelated sketches): a quarter of the LDS atomic rate, i.e. the dependent compare-and-swap round trips of waves whose
        // every step waits for its slowest lane.  Two chains per thread in flight, interleaved, were measured: 13.2 us)
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
        // (CB = 3 / 4 / 6 under amdgpu_waves_per_eu(8) with the slots packed two per register -- 64 VGPRs, 4-6 spilled -- measured in round 5:
        // 53.1 -> 56.3 us for CB = 4, step 0.290 -> 0.294 ms.  Without a single candidate (a matrix of unrelated sketches) the kernel takes 38 us:
        // the confirm phase costs 15 us for ten million candidates whatever its batch size)
        constexpr int CB = 2;                                             // owner fetches in flight per thread
        uint32_t redo = 0;
#ifdef D2G_RANK_TRACE
        __syncthreads();
        RK_STAMP(2);
#endif
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
        RK_STAMP(3);
        compact();
        RK_STAMP(4);
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const size_t j = (size_t)i * BS_RANK_THREADS + tid;
            if (j < N && !(skip >> i & 1)) ids[j] = own[hs[i]];
        }
#ifdef D2G_RANK_TRACE
        __syncthreads();
        RK_STAMP(5);
#endif
        for (uint32_t k = tid; k < nfix && k < BS_MAXFIX; k += BS_RANK_THREADS) ids[fix_j[k]] = own[fix_h[k]];
        if (tid == 0) { colcnt[t * BS_CC_STRIDE] = running - 1; colcnt[t * BS_CC_STRIDE + 5] = running - 1; }   // #values shared by >= 2 sketches ([5]: a copy the column plan leaves alone -- sp_sample_fin_kernel reads it beside that kernel)
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
        // which of this thread's values the pass placed, a byte per batch (up to eight batches: 65 536 sketches): the pass's second loop then
        // fetches only THEIR pending words instead of every id of the column (round 6: 1033 -> 992 us at 50 000 sketches).
        // (Round 6 also measured HASHED later passes -- the first pass leaves the top 30 bits of every later value's hash product in ids[], the later
        // passes walk those 4-byte words instead of the 8-byte values, only tag matches fetch their value: correct, and SLOWER -- first pass 47 -> 74 us,
        // later ones 43 -> 48: the walk is not bound by the bytes it reads but by its claim / confirm round trips; profiles/r06_rank_trace.txt)
        const bool mine_known = N <= (size_t)8 * PG * BS_RANK_THREADS;
        unsigned long long mine_all = 0;
        int batch = 0;
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
            if (batch < 8) mine_all |= (unsigned long long)mineb << (8 * batch);
            ++batch;
        }
        __syncthreads();
        RK_STAMP(1 + 2 * (part - part_lo));
        compact();
        batch = 0;
        for (size_t j0 = 0; j0 < N; j0 += (size_t)PG * BS_RANK_THREADS, ++batch) {
            uint32_t sl[PG];
            const uint32_t mb = mine_known ? (uint32_t)(mine_all >> (8 * batch)) & 0xFFu : 0xFFu;
#pragma unroll
            for (int i = 0; i < PG; ++i) {
                const size_t j = j0 + (size_t)i * BS_RANK_THREADS + tid;
                sl[i] = (j < N && (mb >> i & 1)) ? ids[j] : 0;
            }
#pragma unroll
            for (int i = 0; i < PG; ++i) {
                const size_t j = j0 + (size_t)i * BS_RANK_THREADS + tid;
                if (j < N && (mb >> i & 1) && (mine_known || !MULTI || (tagged ? (sl[i] & ~0x7FFFu) == ptag : (sl[i] & BS_PENDING) != 0)))
                    ids[j] = own[sl[i] & (tagged ? 0x7FFFu : ~BS_PENDING)];
            }
        }
        __syncthreads();
        RK_STAMP(2 + 2 * (part - part_lo));
    }
    if (tid == 0) { colcnt[t * BS_CC_STRIDE + split] = running - 1; if (nsplit == 1) colcnt[t * BS_CC_STRIDE + 5] = running - 1; }   // #values shared by >= 2 sketches (this split's)
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

// the same sum with ONE round trip: lane t of the calling wave takes group t, the wave adds up (the scalar loop above waits out one
// scalar load per group: up to 31 dependent round trips in front of a kernel's first store).  Every lane of the wave must be here.
__device__ __forceinline__ size_t stream_slot_wave(const uint32_t *meta, int tb) {
    const int lane = threadIdx.x & 63;
    uint32_t q = 0;
    for (int t0 = 0; t0 < tb; t0 += 64) {
        const int t = t0 + lane;
        uint32_t v = t < tb ? (uint32_t)live_planes(meta, t) : 0u;
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        q += v;
    }
    return (size_t)__builtin_amdgcn_readfirstlane(q);
}

// ------------------------------------------------------------------ riders
// The fill of an upper-triangle launch (every output = "no register equal", 4 bytes per pair: 200 MB at config 3, 31 us of pure HBM
// writes) depends on nothing the prepare computes, and six kernels of the prepare chain are one to forty workgroups waiting out
// dependent round trips on an otherwise idle chip (column plan, flatten, count, attach, scan, place: 36 us together at config 3).  When the
// output is announced ahead of the prepare (d2g_cmp_ut_announce_dev) those kernels are launched with extra workgroups BEHIND their own --
// the dispatcher starts workgroups in index order, the kernel's own work is never queued behind a rider -- each writing one 32 KB piece.
// (The same fill on a second stream was measured in rounds 4 and 5: the two cross-stream dependencies cost more than the fill.)
__device__ __forceinline__ void sp_ride_piece(const SpRider &r, size_t piece) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t v = r.vsrc ? r.vsrc[0] : r.vimm;
    const size_t head = min(r.cnt, (size_t)((16 - ((uintptr_t)r.out & 15)) & 15) / 4);   // the output pointer is only 4-byte aligned in general
    u32x4 *body = reinterpret_cast<u32x4 *>(r.out + head);
    const size_t nb = (r.cnt - head) / 4;
    const size_t end = min((piece + 1) * (size_t)2048, nb);
    // non-temporal stores (as in sp_fill_kernel): 200 MB of output nobody reads again must not push the prepare's working set out of the MALL
    for (size_t i = piece * 2048 + threadIdx.x; i < end; i += blockDim.x) __builtin_nontemporal_store(u32x4{v, v, v, v}, &body[i]);
    if (piece == 0) {
        if (threadIdx.x < head) r.out[threadIdx.x] = v;
        const size_t tail0 = head + nb * 4;
        if (tail0 + threadIdx.x < r.cnt) r.out[tail0 + threadIdx.x] = v;
    }
}
__device__ __forceinline__ void sp_ride(const SpRider &r) { sp_ride_piece(r, (size_t)r.piece0 + (blockIdx.x - r.own)); }
#define SP_RIDE_OR_WORK(r) do { if (blockIdx.x >= (r).own) { sp_ride(r); return; } } while (0)
// A LARGE announced output (>= SP_SIDE_FILL_PIECES pieces: 1 GB, ~23 000 sketches) is filled by a kernel of its own on a second stream, beside the
// rank kernel: that one waits on its LDS tables for 1.0 ms at 50 000 sketches with the HBM almost idle, while the 5 GB fill spread over the six
// hosts made each of them last 110-180 us (0.75 ms together).  Two cross-stream dependencies (~10 us) that a 200 MB fill does not earn back
// (measured in rounds 4 and 5) and a 5 GB one does many times over.
__global__ __launch_bounds__(256) void sp_side_fill_kernel(SpRider r, uint32_t pieces) {
    for (size_t p = blockIdx.x; p < pieces; p += gridDim.x) sp_ride_piece(r, p);
}
constexpr uint32_t SP_SIDE_FILL_PIECES = 32768;
// How hard the side fill may pull: unthrottled (one workgroup per 32 KB piece, 5.5 TB/s) it stretched the rank kernel beside it from 1.04 to 1.63 ms at
// 50 000 sketches (step 2.54 ms); with persistent workgroups looping over the pieces -- 64 / 128 / 192 / 256 / 384 of them: 2.4 / 3.7 / 4.4 / 4.5 / 5.1 TB/s --
// the two end together at 128 (rank 1.39, fill 1.36 ms: step 2.27 ms).  So: the fewest workgroups whose rate still finishes the fill within 1.3 x the rank
// kernel's own time (21 ns per sketch at S = 1024, measured at 50 000), everything the chip has when even that is too slow (100 000 sketches on).
inline unsigned sp_side_fill_grid(const d2g_ctx *ctx, uint32_t pieces, size_t words, size_t N, size_t S) {
    const double need = (double)words * 4.0 / (1.3 * 21.0 * (double)N * ((double)S / 1024.0));     // bytes per ns = GB/s
    const unsigned cus = (unsigned)ctx->num_cus;
    const unsigned g = need <= 3700.0 ? cus / 2 : need <= 4400.0 ? cus * 3 / 4 : need <= 5000.0 ? cus * 3 / 2 : pieces;
    return std::max(1u, std::min<unsigned>(pieces, g));
}
constexpr SpRider SP_NO_RIDER{nullptr, 0, nullptr, 0u, 0xFFFFFFFFu, 0u};

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
                                                                       uint32_t *__restrict__ ex_status, int sort, SpRider rider) {
    SP_RIDE_OR_WORK(rider);
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
// start-of-prepare work of the sparse path (section 4) carried by a kernel that runs anyway (bs_planes_kernel, sp_unpack_kernel)
// instead of a launch and memsets of its own: label[j] = j, `owords` words at `ones` set to all-ones (the hints), `zwords` words at
// `zero` cleared (counters, linked flags, tile bitmap + control words, order words, the pair list's cursor)
struct SpInit {
    uint32_t *label = nullptr, *ones = nullptr, *zero = nullptr;
    uint32_t n = 0, owords = 0, zwords = 0;
};
__device__ __forceinline__ void sp_init_part(const SpInit &si, size_t lin, size_t nthreads) {
    if (!si.label) return;
    if (lin < si.n) si.label[lin] = (uint32_t)lin;
    for (size_t x = lin; x < si.owords; x += nthreads) si.ones[x] = 0xFFFFFFFFu;
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
    const size_t slot = (forms & BS_FORM_STREAM) ? stream_slot_wave(meta, (int)tb) : 0;     // (every lane still here)
    if (jpos >= Nstride) return;
    // sperm: the operand written in another order -- position p holds sketch sperm[p] (stream form only; unused: the sparse path permutes the finished stream)
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
    uint32_t *sdst = st ? stream + slot * 2 * Nstride + jpos : nullptr;
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

// (the pair kernel's tile shape, explained below: 16 rows per wave, JR 64-column groups per lane)
#ifndef D2G_BS_WPE
#define D2G_BS_WPE 7
#endif
#ifndef D2G_BS_JR
#define D2G_BS_JR 2
#endif
constexpr int BS_IW = 16;
constexpr int BS_JR = D2G_BS_JR;              // 64-column groups per lane: a wave owns 16 x (64*JR) pairs

__device__ __forceinline__ bool sp_dense_mode(const uint32_t *__restrict__ ctl, uint32_t cand);   // section 4
#include "d2g_k2_patch.h"

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
    const uint32_t *__restrict__ gate, uint32_t gate_cand, SpPatchArgs pa) {
    constexpr int IW = BS_IW;
    constexpr int WC = BS_CB / (64 * JR);          // waves along columns: 2 (JR=2)
    constexpr int WR = 4 / WC;                     // waves along rows
    constexpr int RB = WR * IW;                    // rows per workgroup tile
    if (gate && !sp_dense_mode(gate, gate_cand)) {           // launched behind the sparse path: the tile walk only when that decided for the dense walk;
        sp_patch_lut(pa, sh, store, S, (size_t)blockIdx.x * BS_THREADS + threadIdx.x, (size_t)gridDim.x * BS_THREADS);   // otherwise a short pair list's table epilogue
        return;
    }
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

#include "d2g_k2_sparse.h"

// gathered (caller-owned) operands carry the row coding + the unique plane only: derive the column coding.
// Done before EVERY launch on such a set -- the library cannot know when the caller re-gathered into the
// buffer, and the pass is ~2 % of the pair kernel it precedes.
int refresh_borrowed(d2g_ctx *ctx, const d2g_cmp_set *set, hipStream_t s) {
    if (!set->borrowed || set->managed) return D2G_OK;       // managed: the engine derived the stream as the groups arrived
    return d2g_bitslice_derive_groups(ctx, set, 0, set->ntb, s);
}


template <class Store>
int launch_bitslice(d2g_ctx *ctx, const d2g_cmp_set *set, PairShape sh, Store store, hipStream_t s) {
    if (int rc = finish_shape(ctx, sh, BS_JR == 2 ? 32u : 64u)) return rc;   // workgroup tile = (16 * JR) rows x 256 columns (4 waves of 16 x 64*JR)
    if (sh.nvalid_total == 0) return D2G_OK;
    if (int rc = refresh_borrowed(ctx, set, s)) return rc;
    if (int rc = d2g_bitslice_ensure_natural(ctx, set, s)) return rc;
    d2g_timer tm(ctx, &ctx->ev_k2, s);
    hipLaunchKernelGGL((k2_bitslice_kernel<BS_JR, Store>), dim3(sh.per_xcd * 8), dim3(BS_THREADS), 0, s, set->d_stream,
                       set->Nstride, set->d_meta, set->ntb, (uint32_t)set->S, sh, store, (const uint32_t *)nullptr, 0u, SpPatchArgs{});
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
    (void)hipFree(set->d_owner); set->d_owner = nullptr;
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
        if (const char *e = ctx->tune.get("D2G_BS_NSPLIT")) {                 // tests / experiments
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
bool sort_columns(const d2g_ctx *ctx) {
    const char *e = ctx->tune.get("D2G_BS_SORT");     // "0": keep the caller's column order (A/B measurements, tests)
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
    set->sparse_ok = sparse_enabled(ctx, set->N) && set->S < 65536;          // (the sparse kernel packs two mismatch counts into one LDS word)
    if (set->sparse_ok && sp_alloc(ctx, set) != D2G_OK) { sp_free(set); (void)hipGetLastError(); set->sparse_ok = false; }
    if (set->sparse_ok && set->nsplit == 1) {                            // one holder per shared value, left by the rank kernel (sp_olink_kernel)
        set->owner_stride = set->N / 2 + 8;
        if (hipMalloc((void **)&set->d_owner, set->S * set->owner_stride * sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); set->d_owner = nullptr; }
    }   // no memory for the sparse path's buffers: the dense walk works without them
    return D2G_OK;
}

// ids + column plan + planes for the operand currently in set->d_cols.  Fully asynchronous on `s`.
// meta[0..ntb) = per-group shared-value counts; meta[ntb] = status word (bit 0: the rank kernel's LDS table
// overflowed on some column -- see d2g_bitslice_status)
int d2g_bitslice_prepare(d2g_ctx *ctx, d2g_cmp_set *set, hipStream_t s) {
    const size_t N = set->N, S = set->S, Npad = set->Npad;
    // an announced output (d2g_cmp_ut_announce_dev): this prepare's latency-bound kernels carry its fill -- unless the ordering is going to
    // be skipped (the remembered give-up: the launch will be dense and fill nothing)
    const bool sparse_path = set->sparse_ok && !set->export_only && !set->want_exchange;
    set->ride_next = 0; set->ride_total = 0;
    set->ride_mask = sp_tuning(ctx).ride;
    // (the remembered word is device-written host memory read without synchronisation: read ONCE per prepare -- ADVICE r5 -- and handed down)
    set->skip_cached = sparse_path ? (sp_will_skip(ctx, set) ? 1 : 0) : -1;
    if (set->ride_out && sparse_path && set->ride_mask && set->skip_cached != 1)
        set->ride_total = (uint32_t)std::min<size_t>(sp_fill_pieces(set->ride_cnt), 0x7FFFFFFFu);
    if (sp_fill_pieces(set->ride_cnt) > 0x7FFFFFFFu) set->ride_total = 0;
    bool side_fill = false;
    if (set->ride_total >= SP_SIDE_FILL_PIECES) {
        if (!set->fill_stream) {
            hipStream_t st = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr;
            if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&e0, hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&e1, hipEventDisableTiming) == hipSuccess) { set->fill_stream = st; set->fill_fork = e0; set->fill_join = e1; }
            else { (void)hipGetLastError(); if (st) (void)hipStreamDestroy(st); if (e0) (void)hipEventDestroy(e0); }
        }
        if (set->fill_stream) {
            // behind everything already on `s` (the transpose of this update; whoever read the output last), beside everything this prepare enqueues
            hipStream_t fs = (hipStream_t)set->fill_stream;
            D2G_HIP(ctx, hipEventRecord((hipEvent_t)set->fill_fork, s));
            D2G_HIP(ctx, hipStreamWaitEvent(fs, (hipEvent_t)set->fill_fork, 0));
            const SpRider all{set->ride_out, set->ride_cnt, set->ride_vsrc, set->ride_vimm, 0u, 0u};
            const unsigned fgrid = sp_side_fill_grid(ctx, set->ride_total, set->ride_cnt, N, S);
            hipLaunchKernelGGL(sp_side_fill_kernel, dim3(fgrid), dim3(256), 0, fs, all, set->ride_total);
            D2G_HIP(ctx, hipEventRecord((hipEvent_t)set->fill_join, fs));
            set->ride_next = set->ride_total;                            // nothing left for the riders
            side_fill = true;
        }
    }
    // the status word meta[ntb] was zeroed by the transpose kernel that filled d_cols (no memset node in the chain)
    {
        const int logTl = set->logT < BS_LOG_TLDS_MAX ? set->logT : BS_LOG_TLDS_MAX;
        const size_t lds = (size_t(1) << logTl) * sizeof(uint32_t);
        void (*kern)(const uint64_t *, size_t, size_t, uint32_t, int, uint32_t *, uint32_t *, uint32_t *, int, uint32_t, int, uint32_t *, size_t) = bs_rank_kernel<false, false>;
        const bool multi = set->logT > BS_LOG_TLDS_MAX;
        if (multi) kern = bs_rank_kernel<true, false>;
        else if (N <= (size_t)12 * BS_RANK_THREADS) kern = bs_rank_kernel<false, true>;     // PF * BS_RANK_THREADS
        int tagbits_max = 31;
        if (const char *e = ctx->tune.get("D2G_BS_TAGBITS")) { const int v = std::atoi(e); if (v >= 0 && v < 31) tagbits_max = v; }   // tests
        if (lds > 48 * 1024)
            D2G_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int nsplit = multi ? set->nsplit : 1;
        // split passes tell their pending words from final ids by a tag and trust that ids[] holds nothing else that looks pending.  A
        // prepare that overflowed (status bit 0) may leave pending words behind, and the multi-GPU engine re-uses its exporter sets step
        // after step: ids[] starts from zero in every split prepare -- a few MB for the narrow column slices that are split
        if (nsplit > 1) D2G_HIP(ctx, hipMemsetAsync(set->d_ids, 0, S * Npad * sizeof(uint32_t), s));
        hipLaunchKernelGGL(kern, dim3((unsigned)(S * nsplit)), dim3(BS_RANK_THREADS), lds, s, set->d_cols, N, Npad, set->T, set->logT,
                           set->d_ids, set->d_colcnt, set->d_meta + set->ntb, tagbits_max, (uint32_t)S, nsplit, nsplit == 1 ? set->d_owner : (uint32_t *)nullptr, set->owner_stride);
        // a set's first prepare looks at its matrix before it orders it: the sample needs the ids only, so it stands HERE and the two kernels
        // below run while the host waits for its word (sp_sample_collect)
        if (sparse_path && sp_sample_due(ctx, set)) { if (int rc = sp_sample_enqueue(ctx, set, s)) return rc; }
        unsigned g; const SpRider rd = sp_take_rider(set, 1, SP_RW_PLAN, false, &g, 1);
        hipLaunchKernelGGL(bs_colplan_kernel, dim3(g), dim3(BS_PLAN_THREADS), 0, s, set->d_colcnt, (uint32_t)S, set->ntb, nsplit, set->d_perm,
                           set->d_meta, set->d_meta + set->ntb, set->ex_meta, set->ex_status, sort_columns(ctx) ? 1 : 0, rd);
    }
    dim3 grid((unsigned)div_up<size_t>(set->Nstride, 256), (unsigned)set->ntb);
    const bool split = set->logT > BS_LOG_TLDS_MAX && set->nsplit > 1;
    if (sparse_path) {
        // sparse path (section 4): the caller's-order stream first (the dense walk and rectangular launches read it), then the families,
        // the pair list and the stream in family order
        // (gathering the ids through d_sperm inside bs_planes_kernel was measured: 74 us instead of 18 at config 3 -- 1024 columns of
        // uncoalesced 4-byte loads; permuting the finished stream touches 256 rows of words and leaves the caller's-order stream valid)
        // (a second queue for the caller's-order planes beside the ordering, and for the fill beside the tile list, was measured in round 4:
        // the kernels slow each other down by what the overlap hides and the events cost more: dropped)
        // (the planes kernel also initialises the ordering's arrays and clears the tile bitmap: no launch / memset of their own)
        hipLaunchKernelGGL(split ? bs_planes_kernel<true> : bs_planes_kernel<false>, grid, dim3(256), 0, s, set->d_ids, N, Npad,
                           set->d_planes, set->d_stream, set->Nstride, set->nbits_cap, set->d_meta, BS_FORM_STREAM, set->d_perm, set->d_colcnt, (const uint32_t *)nullptr, sp_init_of(set));
        if (int rc = sp_sample_collect(ctx, set, s)) return rc;         // (the first look, enqueued behind the rank kernel: the host waits for its word here)
        if (int rc = sp_prepare_order(ctx, set, split, s)) return rc;
        if (int rc = sp_permute(ctx, set, s)) return rc;
        if (side_fill) D2G_HIP(ctx, hipStreamWaitEvent(s, (hipEvent_t)set->fill_join, 0));     // the launch writes into the filled output
        set->srt_valid = true; set->nat_valid = true;
        if (set->ride_total) { set->prefilled = set->ride_out; set->prefilled_cnt = set->ride_cnt; set->prefilled_pieces = set->ride_next; set->prefilled_src = set->ride_vsrc; set->prefilled_by_riders = true; }   // (the launch fills what is left)
        else if (set->prefilled_by_riders) set->prefilled = nullptr;        // what an EARLIER prepare's riders wrote is void once another prepare has run (the caller may have used the buffer in between)
        set->ride_out = nullptr; set->ride_total = 0;                       // an announcement serves ONE prepare
        D2G_HIP(ctx, hipGetLastError());
        return D2G_OK;
    }
    set->ride_out = nullptr;
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
    const_cast<d2g_cmp_set *>(set)->prefilled = nullptr;               // (an early fill is void once a launch has written every output itself)
    if (eq_out) return launch_bitslice(ctx, set, sh, StoreEq{eq_out}, s);
    return launch_bitslice(ctx, set, sh, StoreLut{fout, lut}, s);
}

int d2g_bitslice_prefill(d2g_ctx *ctx, d2g_cmp_set *set, size_t r0, size_t r1, uint32_t *eq_out, const float *lut, float *fout, hipStream_t s) {
    if (eq_out) return sp_prefill(ctx, set, r0, r1, StoreEq{eq_out}, eq_out, s);
    return sp_prefill(ctx, set, r0, r1, StoreLut{fout, lut}, reinterpret_cast<uint32_t *>(fout), s);
}

int d2g_bitslice_announce(d2g_ctx *, d2g_cmp_set *set, size_t r0, size_t r1, uint32_t *eq_out, const float *lut, float *fout) {
    const size_t cnt = d2g_ut_count(set->N, r0, r1);
    set->ride_out = nullptr;
    if (!eq_out && !fout) return D2G_OK;                               // the cancelling form: nothing rides on the next prepare
    if (!cnt || !set->sparse_ok) return D2G_OK;
    // the fill value = Store::value_from_mismatches(S, S): the count 0, or lut[0] (read by the riders when they run)
    set->ride_out = eq_out ? eq_out : reinterpret_cast<uint32_t *>(fout);
    set->ride_vsrc = eq_out ? nullptr : reinterpret_cast<const uint32_t *>(lut);
    set->ride_vimm = 0;
    set->ride_cnt = cnt;
    return D2G_OK;
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
    if (!(sparse_enabled(ctx, set->N) && set->S < 65536)) return D2G_OK;
    hipError_t e;
    if ((e = hipMalloc((void **)&set->d_ids, set->ncols * set->Npad * sizeof(uint32_t))) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_colcnt, set->ncols * BS_CC_STRIDE * sizeof(uint32_t))) != hipSuccess ||
        (e = hipMemset(set->d_colcnt, 0, set->ncols * BS_CC_STRIDE * sizeof(uint32_t))) != hipSuccess) {
        // not enough memory for the sparse path's buffers: the dense walk works without them
        (void)hipFree(set->d_ids); (void)hipFree(set->d_colcnt); set->d_ids = set->d_colcnt = nullptr;
        (void)hipGetLastError();
        return D2G_OK;
    }
    set->ids_owned = true;
    if (sp_alloc(ctx, set) != D2G_OK) { sp_free(set); (void)hipGetLastError(); return D2G_OK; }
    set->sparse_ok = true;
    return D2G_OK;
}

int d2g_bitslice_managed_ready(d2g_ctx *ctx, d2g_cmp_set *set, hipStream_t s) {
    if (!set->borrowed || !set->sparse_ok) return D2G_OK;
    dim3 grid((unsigned)div_up<size_t>(set->Npad, 256), (unsigned)set->ntb);
    hipLaunchKernelGGL(sp_unpack_kernel, grid, dim3(256), 0, s, set->d_planes, set->Nstride, set->nbits_cap, set->d_meta, set->N, set->Npad, set->d_ids, set->d_colcnt, sp_init_of(set));
    if (int rc = sp_prepare_order(ctx, set, false, s)) return rc;
    if (int rc = sp_permute(ctx, set, s)) return rc;
    D2G_HIP(ctx, hipGetLastError());
    set->srt_valid = true;
    return D2G_OK;
}

// the set's remembered decisions forgotten: its next prepare decides as a new set's first prepare does (measurements: bench.py's first_step_ms)
void d2g_bitslice_forget(d2g_cmp_set *set) {
    if (set->h_gaveup) { set->h_gaveup[0] = 0; set->h_gaveup[1] = 0; }
    set->sp_prepares = 0; set->pred_valid = false;
}

// diagnostics of the sparse path's LAST launch on this set (synchronises `s`): see d2g.h
int d2g_bitslice_sparse_info(d2g_ctx *ctx, const d2g_cmp_set *set, hipStream_t s, uint32_t *out4) {
    out4[0] = out4[1] = out4[2] = out4[3] = 0;
    if (!set->srt_valid || !set->d_spctl) { D2G_HIP(ctx, hipStreamSynchronize(s)); return D2G_OK; }
    uint32_t c[4] = {0, 0, 0, 0}, ord[4] = {0, 0, 0, 0}, pl = 0;
    if (!set->last_ctl) { D2G_HIP(ctx, hipStreamSynchronize(s)); out4[0] = 1; return D2G_OK; }   // prepared, never launched
    D2G_HIP(ctx, hipMemcpyAsync(c, set->last_ctl, sizeof c, hipMemcpyDeviceToHost, s));   // the control words the last launch used
    D2G_HIP(ctx, hipMemcpyAsync(ord, set->d_order, sizeof ord, hipMemcpyDeviceToHost, s));
    D2G_HIP(ctx, hipMemcpyAsync(&pl, set->d_plctl, sizeof pl, hipMemcpyDeviceToHost, s));
    D2G_HIP(ctx, hipStreamSynchronize(s));
    const bool dense = (c[1] & 1u) || (size_t)c[0] * 5 > (size_t)c[3] * 2;
    // [2]: bit 0 the prepare decided for the dense walk, bit 1 the dense kernel ran, bit 2 tiles + pair list were used, bit 3 the caller's order was kept,
    //      bit 4 the last prepare skipped the ordering (the one before had given up: sp_prepare_order)
    out4[0] = 1; out4[1] = c[0]; out4[2] = (ord[0] ? 1u : 0u) | (dense ? 2u : 4u) | (ord[0] ? 8u : 0u) | (set->sp_skipped ? 16u : 0u);
    out4[3] = ord[0] ? 0u : (uint32_t)std::min<size_t>(pl, set->plist_cap);
    return D2G_OK;
}

// diagnostics: the pair list and the sorted order of the last prepare, copied to the host (synchronises `s`)
int d2g_bitslice_debug_read(d2g_ctx *ctx, const d2g_cmp_set *set, hipStream_t s, uint64_t *pairs_out, size_t cap, size_t *npairs, uint32_t *root_out /* [N] or null */) {
    *npairs = 0;
    if (!set->srt_valid || !set->d_plctl) { D2G_HIP(ctx, hipStreamSynchronize(s)); return D2G_OK; }
    uint32_t n = 0;
    D2G_HIP(ctx, hipMemcpyAsync(&n, set->d_plctl, 4, hipMemcpyDeviceToHost, s));
    D2G_HIP(ctx, hipStreamSynchronize(s));
    const size_t m = std::min<size_t>(std::min<size_t>(n, set->plist_cap), cap);
    if (m && pairs_out) D2G_HIP(ctx, hipMemcpy(pairs_out, set->d_plist, m * 8, hipMemcpyDeviceToHost));
    if (root_out) D2G_HIP(ctx, hipMemcpy(root_out, set->d_label + set->Npad, set->N * 4, hipMemcpyDeviceToHost));
    *npairs = m;
    return D2G_OK;
}

// what the engines of one multi-GPU job must agree on: the RESOLVED values (a rank with D2G_SP_LINK=1 and one that leaves it unset resolve the
// same kernels -- ADVICE r5: the textual form refused them as different)
uint64_t d2g_k2_tuning_hash(const d2g_ctx *ctx) {
    const SpTuning t = sp_tuning(ctx);
    long long tag = -1, nsplit = 0;
    if (const char *e = ctx->tune.get("D2G_BS_TAGBITS")) tag = std::atoi(e);
    if (const char *e = ctx->tune.get("D2G_BS_NSPLIT")) nsplit = std::atoi(e);
    const long long v[] = {t.sparse, (long long)t.min_n, t.link, (long long)(t.tile_frac * 1e6), t.olink, t.emit_big, t.ride, t.remember, (long long)t.list_div, (long long)t.long_list,
                           t.list_form, t.predict, sort_columns(ctx) ? 1 : 0, tag, nsplit};
    uint64_t h = 1469598103934665603ull;
    for (long long x : v) for (int b = 0; b < 8; ++b) { h ^= (uint64_t)(x >> (8 * b)) & 0xFF; h *= 1099511628211ull; }
    return h;
}

std::string d2g_k2_tuning_json(const d2g_ctx *ctx) {
    const SpTuning t = sp_tuning(ctx);
    char b[640];
    std::snprintf(b, sizeof b, "{\"D2G_BS_SPARSE\": %d, \"D2G_BS_SPARSE_MIN_N\": %zu, \"D2G_BS_SORT\": %d, \"D2G_SP_LINK\": %d, \"D2G_SP_OLINK\": %d, \"D2G_SP_TILE_FRAC\": %.3f, "
                  "\"D2G_SP_LIST_DIV\": %zu, \"D2G_SP_LONG_LIST\": %zu, \"D2G_SP_LIST_FORM\": %d, \"D2G_SP_PREDICT\": %d, \"D2G_SP_REMEMBER\": %d, \"D2G_SP_RIDE\": %d, \"D2G_SP_EMIT_BIG\": %d}",
                  t.sparse ? 1 : 0, t.min_n, sort_columns(ctx) ? 1 : 0, t.link, t.olink, t.tile_frac, t.list_div, t.long_list, t.list_form, t.predict, t.remember, t.ride, t.emit_big);
    return b;
}

void d2g_warm_k2_bitslice() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&bs_colplan_kernel)); }

#ifdef D2G_RANK_TRACE
extern "C" int d2g_debug_rank_trace(unsigned long long *out, size_t n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rank_trace), std::min<size_t>(n, 8192 * 16) * 8) == hipSuccess ? 0 : -1;
}
#endif
#ifdef D2G_SP_TRACE
// variant builds only: the time stamps of the last sparse pair kernel (tools/sp_trace.py)
extern "C" int d2g_debug_emit_trace(unsigned long long *out, size_t n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_emit_trace), std::min<size_t>(n, 4096 * 16) * 8) == hipSuccess ? 0 : -1;
}
extern "C" int d2g_debug_sp_trace(unsigned long long *out, size_t n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sp_trace), std::min<size_t>(n, 16384 * 8) * 8) == hipSuccess ? 0 : -1;
}
#endif

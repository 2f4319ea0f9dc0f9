// d2g_k2_sparse.h -- internal: section 4 of d2g_k2_bitslice.hip (included there, inside its anonymous namespace).
//
// ------------------------------------------------------------------ 4. sparse tiles + pair list
// An equality count (reference src/cmp_core.cpp:461,506; only the count matters, :465) is zero unless the two sketches share a value in
// at least one register column.  In a collection of related genomes most pairs share nothing, the pairs that share MANY values come in
// families, and a few pairs share one or two values by chance (a conserved k-mer that is the bucket minimum in two genera).  So an
// upper-triangle launch computes
//     family pairs      in 32 x 256 TILES of the bit-sliced operand, put into an order that makes a family a run of adjacent positions;
//     chance pairs      from a PAIR LIST: one entry per (pair, shared value) -- the equality count of a pair outside the listed tiles is
//                       the number of its entries;
//     everything else   by a streaming fill with the value of "0 equal registers".
// Nothing is approximate, whatever the families look like: the PARTITION of the sketches into segments (runs of adjacent sorted positions)
// is a heuristic, and exactness rests on two facts that hold for ANY partition:
//     (a) every pair inside one segment lies in a listed tile (sp_segtiles: the tiles a segment's row blocks and column blocks meet in),
//         and a listed tile is computed exactly for all of its pairs by the pair kernel;
//     (b) sp_emit_kernel walks EVERY shared value of EVERY column and emits every pair of its holders that lie in different segments;
//         the patch kernel adds an entry to the output only where the pair's tile is not listed.
// A good partition makes both cheap; a bad one makes the list long, and a list that outgrows its buffer (or segments that cover too many
// tiles, or one family that holds most sketches) sends the launches to the plain pair kernel over every tile (order[0]).
//
// prepare  sp_link    ROBUST families: one workgroup per PAIR of adjacent columns; two sketches are united (lock-free union-find on
//                     label[]) only where they agree in BOTH columns -- a family's members do so in many column pairs, a stranger that
//                     shares one chance value with a family does not, so one chance collision no longer welds two families together
//                     (round 4 united over single columns: ten collisions per sketch put every sketch into one component);
//          sp_flatten between the two link passes; sp_attach behind them: a sketch no column pair linked joins the family its hints -- some
//                     holder of a value it shares, recorded by even and by odd column pairs -- point to (a weak member, families of two);
//          sp_count / scan / place   counting sort by root -> sperm / sinv, the segments, keep-the-caller's-order decision;
//          sp_emit    (b) above (the segments' tiles are set by the sort's place kernel);
//          sp_permute the finished plane stream in sorted order.
// launch   sp_list -> sp_fill -> k2_bitslice_sparse_kernel (listed tiles; stores the non-zero counts) -> sp_patch_add (list entries
//          outside listed tiles: atomicAdd of 1 onto the filled word; the first adder of a position is its leader) -> sp_patch_lut (table
//          epilogue: the leader turns the count into the float) -> the plain pair kernel, which runs only in dense mode.
constexpr uint32_t SP_NONE = 0xFFFFFFFFu;
constexpr uint32_t SP_SAMPLE_ROWS = 16, SP_SAMPLE_FAM = 4, SP_SAMPLE_COLS = 32;   // the first look at a matrix (sp_sample_kernel)
constexpr unsigned long long SP_NULL_ENTRY = ~0ull;   // an empty slot of the pair list: two outsiders of one value that share a segment (sp_pairs_kernel); every reader skips it
#ifndef D2G_SP_KS
#define D2G_SP_KS 8
#endif
#ifndef D2G_SP_WPE
#define D2G_SP_WPE 8
#endif

__device__ __forceinline__ uint32_t sp_rank(uint32_t w, const uint32_t *__restrict__ colcnt, size_t t, bool split) {
    if ((w >> 31) || w == 0) return 0;          // unique (or padding): never equal to anything
    return split ? (w & BS_RANK_MASK) + colcnt[t * BS_CC_STRIDE + (w >> BS_SPLIT_SHIFT)] : w;
}

__device__ __forceinline__ uint32_t sp_ld(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---- lock-free union-find on label[]: label[j] <= j always; a root is hooked under a smaller root with a compare-and-swap
__device__ __forceinline__ uint32_t sp_find(uint32_t *label, uint32_t l) {
    for (int h = 0; h < 64; ++h) {                                    // bounded: sp_union retries, and gives up in the end (the partition is a heuristic)
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

// One workgroup per pair of adjacent columns (t1, t2) = (2g, 2g + 1).  LDS, per shared value r1 of column t1 (ranks 1 .. nv):
//   r2[r1]   the rank in column t2 of the first holder of r1 that has a shared value there too (claimed with a compare-and-swap); the
//            holders of r1 whose t2 rank equals r2[r1] MATCH: they agree in both columns;
//   mn[r1]   the smallest label among the matchers.
//   MODE 0 (every column pair: propagate)  every matcher takes mn with a PLAIN store.  A label is only ever replaced by a smaller index
//            of the same family, so whichever racing store lands last the array still holds, per sketch, an earlier member of its
//            family.  This does nearly all the uniting without a single global atomic (letting every matcher run the union-find below
//            from identity labels was measured: 264 us at config 3 -- 1.6 million root walks and compare-and-swaps for the 9 934 hooks
//            the families need).  Also here: any[r1] = some holder of r1 -- the hint a holder that did not match takes away
//            (sp_attach_kernel).
//   MODE 1 (every `stride`-th column pair, behind sp_flatten_kernel: unite)  where a matcher's label still differs from mn, ONE
//            matcher per value runs the lock-free union of the two trees: the safety net for families the racing stores left under two
//            roots (every matcher doing so cost 40 us; a quarter of the column pairs still sees every family dozens of times).
// Values beyond the table (nv = min(D2, cap)) take no part: the partition is a heuristic, sp_emit_kernel keeps the result exact.
constexpr uint32_t SP_UDONE = 0x40000000u, SP_R2MASK = 0x3FFFFFFFu;   // r2[]: bit 30 = one of the value's matchers has run the union (ranks stay below 2^29: N < 2^30)
template <int MODE>
__global__ __launch_bounds__(1024) void sp_link_kernel(const uint32_t *__restrict__ ids, size_t N, size_t Npad, uint32_t ncols, const uint32_t *__restrict__ colcnt,
                                                       int split, uint32_t cap, uint32_t stride, uint32_t *label, uint32_t *__restrict__ hint) {
    extern __shared__ uint32_t sp_l[];
    const size_t pair = (size_t)blockIdx.x * stride;
    const size_t t1 = 2 * pair, t2 = t1 + 1;
    if (t2 >= ncols) return;
    const uint32_t d2 = colcnt[t1 * BS_CC_STRIDE + 4];
    if (d2 == 0) return;
    const uint32_t nv = min(d2, cap), T = blockDim.x;
    uint32_t *r2 = sp_l, *mn = sp_l + nv, *any = sp_l + 2 * (size_t)nv;         // any: MODE 0 only
    for (uint32_t r = threadIdx.x; r < (MODE ? 2u : 3u) * nv; r += T) sp_l[r] = SP_NONE;
    __syncthreads();
    constexpr int U = 4;
    for (size_t j0 = 0; j0 < N; j0 += (size_t)T * U) {
        uint32_t w1[U], w2[U], lb[U];
#pragma unroll
        for (int x = 0; x < U; ++x) {
            const size_t j = j0 + (size_t)x * T + threadIdx.x;
            w1[x] = j < N ? ids[t1 * Npad + j] : 0u;
            w2[x] = j < N ? ids[t2 * Npad + j] : 0u;
            lb[x] = j < N ? sp_ld(&label[j]) : SP_NONE;
        }
#pragma unroll
        for (int x = 0; x < U; ++x) {
            const uint32_t j = (uint32_t)(j0 + (size_t)x * T + threadIdx.x);
            const uint32_t a = sp_rank(w1[x], colcnt, t1, split != 0), b = sp_rank(w2[x], colcnt, t2, split != 0);
            if (!a || a > nv) continue;
            if (MODE == 0) any[a - 1] = j;                            // (a plain LDS store: the lanes of a wave that share the value write once; a minimum cost 13 us)
            if (!b) continue;
            const uint32_t old = atomicCAS(&r2[a - 1], SP_NONE, b);
            if (old == SP_NONE || (old & SP_R2MASK) == b) {
                atomicMin(&mn[a - 1], lb[x]);
            }
        }
    }
    __syncthreads();
    uint32_t *myhint = hint + (size_t)(pair & 1u) * Npad;
    for (size_t j0 = 0; j0 < N; j0 += (size_t)T * U) {
        uint32_t w1[U], w2[U], lb[U];
#pragma unroll
        for (int x = 0; x < U; ++x) {
            const size_t j = j0 + (size_t)x * T + threadIdx.x;
            w1[x] = j < N ? ids[t1 * Npad + j] : 0u;
            w2[x] = j < N ? ids[t2 * Npad + j] : 0u;
            lb[x] = j < N ? sp_ld(&label[j]) : 0u;
        }
#pragma unroll
        for (int x = 0; x < U; ++x) {
            const uint32_t j = (uint32_t)(j0 + (size_t)x * T + threadIdx.x);
            const uint32_t a = sp_rank(w1[x], colcnt, t1, split != 0), b = sp_rank(w2[x], colcnt, t2, split != 0);
            if (!a || a > nv) continue;
            const uint32_t rr = r2[a - 1];
            const bool match = b && rr != SP_NONE && (rr & SP_R2MASK) == b;
            if (MODE == 0) {
                if (match) {
                    const uint32_t m = mn[a - 1];
                    if (m < lb[x]) label[j] = m;
                } else {
                    const uint32_t h = any[a - 1];
                    if (h != j) myhint[j] = h;
                }
            } else if (match) {
                const uint32_t m = mn[a - 1];
                if (m != lb[x] && !(atomicOr(&r2[a - 1], SP_UDONE) & SP_UDONE)) (void)sp_union(label, lb[x], m);
            }
        }
    }
}

// The same two passes WITHOUT tables, for sets whose rank kernel left one holder per shared value (owner[t][r - 1]; every set that owns its
// operand unless its columns are split over workgroups): a sketch matches the OWNER of its value in column t1 when the owner holds the same
// shared value as the sketch in column t2 too -- one thread per (column pair, four sketches), no LDS, no barrier: two coalesced loads and
// three gathers per sketch instead of a workgroup per column pair walking the column twice behind barriers (config 3: 23 + 10 us -> TBD).
// The owner is an arbitrary holder, so about half of a value's chances to link are lost (the owner must share in t2 as well); the members
// that find no link take the owner as their hint and are attached afterwards.  MODE 1: the first matcher of a value whose label differs from the
// owner's (bit 31 of the owner word, set with one atomic) unites the two trees.
constexpr uint32_t SP_OWNER_MASK = 0x7FFFFFFFu;
template <int MODE>
__global__ __launch_bounds__(256) void sp_olink_kernel(const uint32_t *__restrict__ ids, size_t N, size_t Npad, uint32_t ncols, uint32_t *__restrict__ owner, size_t ostride,
                                                       uint32_t stride, uint32_t *label, uint32_t *__restrict__ hint) {
    const size_t pair = (size_t)blockIdx.y * stride;
    const size_t t1 = 2 * pair, t2 = t1 + 1;
    if (t2 >= ncols) return;
    const size_t j0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (j0 >= N) return;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 w1 = *reinterpret_cast<const u32x4 *>(ids + t1 * Npad + j0);       // (Npad is a multiple of 256: aligned; the padding holds id 0)
    const u32x4 w2 = *reinterpret_cast<const u32x4 *>(ids + t2 * Npad + j0);
    const u32x4 lb = *reinterpret_cast<const u32x4 *>(label + j0);                  // plain loads: a stale label is still an earlier member of the family
    uint32_t *myhint = hint + (size_t)(pair & 1u) * Npad;
    uint32_t o[4], r2o[4], lo[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const uint32_t r1 = w1[x];
        o[x] = (r1 && !(r1 >> 31) && j0 + x < N) ? owner[t1 * ostride + r1 - 1] & SP_OWNER_MASK : SP_NONE;
    }
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const bool live = o[x] != SP_NONE && o[x] != (uint32_t)(j0 + x);
        r2o[x] = live ? ids[t2 * Npad + o[x]] : 0u;
        lo[x] = live ? label[o[x]] : 0u;
        if (!live) o[x] = SP_NONE;
    }
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        if (o[x] == SP_NONE) continue;
        const uint32_t j = (uint32_t)(j0 + x), r2 = w2[x];
        const bool match = r2 && !(r2 >> 31) && r2 == r2o[x];
        if (MODE == 0) {
            if (match) {
                if (lo[x] < lb[x]) label[j] = lo[x]; else if (lb[x] < lo[x]) label[o[x]] = lb[x];
            } else myhint[j] = o[x];
        } else if (match && lb[x] != lo[x]) {
            uint32_t *ow = &owner[t1 * ostride + w1[x] - 1];
            if (!(atomicOr(ow, ~SP_OWNER_MASK) & ~SP_OWNER_MASK)) (void)sp_union(label, lb[x], lo[x]);
        }
    }
}

// every label straight at its root (the walk halves the path behind it; plain loads and stores: a label is only ever replaced by an
// ancestor, so racing with itself is harmless).  Between the two link passes: the uniting pass compares LABELS, and two members of a
// family the propagation has put under one root still carry different earlier members until this ran (without it every matcher walked
// to the roots through device-scope loads: 140 us at config 3 instead of 15).
__global__ __launch_bounds__(256) void sp_flatten_kernel(uint32_t *label, size_t N, SpRider rider) {
    SP_RIDE_OR_WORK(rider);
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    uint32_t l = (uint32_t)j;
    for (int h = 0; h < 64; ++h) {
        const uint32_t p = label[l];
        if (p == l) break;
        const uint32_t g = label[p];
        if (g != p) label[l] = g;
        l = p;
    }
    if (label[l] == l) label[j] = l;
}

// Behind the count kernel (the roots are final, root[] and cnt[] say who is alone): a sketch ALONE under its root joins the family its hints
// lead to -- both hints when it has two (they must agree: a sketch whose shared values all lie with DIFFERENT strangers -- the adversarial
// matrix -- stays alone and goes to the pair list), the one it has otherwise (a weak family member) -- if that root holds a real family
// (two or more sketches; singletons do not chain up: families of two and chains stay in the pair list).  Only root[] and the counters
// change; the labels are dead by now.  (Judged BEFORE the uniting pass a weak member's two hints point into two fragments of its own
// family and are refused; judged by a "matched somebody" flag instead of the counter, a sketch whose one link a racing store undid stayed
// alone with 45 shared values: 28 resp. 19 stragglers at config 3, a mixed value in most columns.)
__global__ __launch_bounds__(256) void sp_attach_kernel(uint32_t *__restrict__ root, uint32_t *__restrict__ cnt, const uint32_t *__restrict__ hint, size_t N, size_t Npad,
                                                        const uint32_t *__restrict__ order, SpRider rider) {
    SP_RIDE_OR_WORK(rider);
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= N || order[2]) return;
    if (root[j] != (uint32_t)j || sp_ld(&cnt[j]) != 1u) return;       // under somebody, or others are under it
    const uint32_t a = hint[j], b = hint[Npad + j];
    if (a == SP_NONE && b == SP_NONE) return;
    // (root[] of another singleton may be changing under us: it then reads as itself or as its new family -- either is a valid answer)
    const uint32_t ra = a != SP_NONE ? sp_ld(&root[a]) : SP_NONE, rb = b != SP_NONE ? sp_ld(&root[b]) : SP_NONE;
    if (ra != SP_NONE && rb != SP_NONE && ra != rb) return;
    const uint32_t r = ra != SP_NONE ? ra : rb;
    if (r == (uint32_t)j || sp_ld(&cnt[r]) < 2u) return;              // only real families take stragglers in
    root[j] = r;
    atomicAdd(&cnt[r], 1u);
    cnt[j] = 0;
}

// block-wide exclusive scan of one value per thread (NW waves); returns the exclusive prefix, *total = the sum
template <int NW = 16>
__device__ __forceinline__ uint32_t sp_block_scan(uint32_t v, uint32_t *wave_tot, uint32_t *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t x = __shfl_up(incl, o); if (lane >= o) incl += x; }
    __syncthreads();
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
    for (int w = 0; w < NW; ++w) { const uint32_t x = wave_tot[w]; if (w < wave) woff += x; tot += x; }
    *total = tot;
    return woff + incl - v;
}

// the same for three counters packed into 64 bits (sp_emit_kernel: outsiders | pairs with insiders | pairs among outsiders)
template <int NW>
__device__ __forceinline__ unsigned long long sp_block_scan64(unsigned long long v, unsigned long long *wave_tot, unsigned long long *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long incl = v;
    for (int o = 1; o < 64; o <<= 1) { const unsigned long long x = __shfl_up(incl, o); if (lane >= o) incl += x; }
    __syncthreads();
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    unsigned long long woff = 0, tot = 0;
    for (int w = 0; w < NW; ++w) { const unsigned long long x = wave_tot[w]; if (w < wave) woff += x; tot += x; }
    *total = tot;
    return woff + incl - v;
}

// tiles a segment of c sketches covers at most, in sixteenths of a tile: (rows + 1) x (columns + 1) tiles from 32 sketches on; smaller
// ones share their row block with their neighbours (two column tiles for c / 32 of a row block)
__device__ __forceinline__ uint32_t sp_seg_est(uint32_t c) {
    c = min(c, 32768u);                                                // (a segment that long is past any limit by itself)
    return c >= 32 ? 16u * ((c + 31) / 32 + 1) * ((c + 255) / 256 + 1) : (c >= 2 ? c : 0u);
}

// counting sort of the sketches by the root of their label, three small kernels (one thread per sketch, then one workgroup for the
// prefix, then one thread per sketch again).  The root walk halves the path behind it (plain loads: nobody hooks roots while this runs,
// and a stale label is still an ancestor); a chain that is not at its root after SP_MAX_HOPS hops (long strings of sketches each united
// with its neighbour only; one chain of 12 000 cost 0.7 ms in its last thread) raises order[2]: the caller's order is kept.
constexpr int SP_MAX_HOPS = 64;
__global__ __launch_bounds__(256) void sp_count_kernel(uint32_t *label, uint32_t *__restrict__ root, size_t N, uint32_t *__restrict__ cnt, uint32_t *__restrict__ order, SpRider rider) {
    SP_RIDE_OR_WORK(rider);
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = j < N;
    uint32_t r = SP_NONE;
    if (live) {
        uint32_t l = (uint32_t)j;
        for (int h = 0;; ++h) {
            const uint32_t p = label[l];
            if (p == l) break;
            if (h == SP_MAX_HOPS) { order[2] = 1; l = (uint32_t)j; break; }
            const uint32_t g = label[p];
            if (g != p) label[l] = g;
            l = p;
        }
        r = l;
        root[j] = r;
    }
    // when everything hangs together ONE counter takes all N increments (measured: 115 us at N = 10 000): the lanes that share the
    // wave's first root add once; the others go one by one
    const unsigned long long alive = __ballot(live);
    if (!alive) return;
    const uint32_t lead = __shfl(r, __ffsll((long long)alive) - 1);
    const unsigned long long m = __ballot(live && r == lead);
    if (live && r == lead) { if ((threadIdx.x & 63) == (unsigned)(__ffsll((long long)m) - 1)) atomicAdd(&cnt[lead], (uint32_t)__popcll(m)); }
    else if (live) atomicAdd(&cnt[r], 1u);
}
// order[0] = 1: the launches walk every tile of the caller's-order operand (one root holds more than half of the sketches, deep label
// chains, or the segments would cover more than seg_tile_limit tiles -- the sparse kernel costs about twice the plain one per tile)
__global__ __launch_bounds__(1024) void sp_scan_kernel(uint32_t *__restrict__ cnt, size_t N, uint32_t *__restrict__ order, uint32_t *__restrict__ start, uint32_t *__restrict__ segend, uint32_t seg_tile_limit,
                                                       uint32_t *__restrict__ gaveup, SpRider rider) {
    SP_RIDE_OR_WORK(rider);
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
            est += sp_seg_est(v[x]);
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
        for (uint32_t x = tid; x < n; x += 1024) {                     // cnt becomes the placing cursor, start and segend stay (a segment ends where the next one starts)
            cnt[base + x] = tile[x]; start[base + x] = tile[x];
            segend[base + x] = x + 1 < n ? tile[x + 1] : s_run + total;
        }
        if (tid == 0) s_run += total;
        __syncthreads();
    }
    est = min(est, 0x3FFFFFu) / 16 + 1;                                 // 1024 threads x 2^18: no overflow
    for (int o = 32; o > 0; o >>= 1) est += __shfl_down(est, o);
    if ((tid & 63) == 0) atomicAdd(&s_est, est);
    __syncthreads();
    if (tid == 0) { const uint32_t keep = (s_big || order[2] || s_est > seg_tile_limit + 1024) ? 1u : 0u; order[0] = keep; *gaveup = keep; }   // gaveup: a word in host memory (sp_prepare_order reads it before the NEXT prepare)
}
// a segment [a, b) of two or more sketches sets the tiles its row blocks (32 positions) and column blocks (256 positions) meet in: every
// pair inside the segment lies in one of them (fact (a) of the header)
__device__ __forceinline__ void sp_segtiles(uint32_t a, uint32_t b, uint32_t CW, uint32_t *__restrict__ gbm) {
    if (b - a < 2) return;                                            // a sketch alone in its segment has no pair inside it
    const uint32_t cb0 = a >> 8, cb1 = (b - 1) >> 8;
    for (uint32_t rb = a >> 5; rb <= (b - 1) >> 5; ++rb)
        for (uint32_t cw = cb0 >> 5; cw <= cb1 >> 5; ++cw) {
            const uint32_t lo = cw == (cb0 >> 5) ? (cb0 & 31) : 0u, hi = cw == (cb1 >> 5) ? (cb1 & 31) : 31u;
            const uint32_t m = (hi == 31 ? 0xFFFFFFFFu : ((1u << (hi + 1)) - 1u)) & ~((1u << lo) - 1u);
            atomicOr(&gbm[(size_t)rb * CW + cw], m);
        }
}
__global__ __launch_bounds__(256) void sp_place_kernel(const uint32_t *__restrict__ root, size_t N, size_t Nstride, uint32_t *__restrict__ cnt,
                                                        uint32_t *__restrict__ sperm, uint32_t *__restrict__ sinv, const uint32_t *__restrict__ order,
                                                        const uint32_t *__restrict__ seg_start, const uint32_t *__restrict__ seg_end, uint32_t CW, uint32_t *__restrict__ gbm,
                                                        uint2 *__restrict__ posseg, SpRider rider) {
    SP_RIDE_OR_WORK(rider);
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = j < N;
    if (!live && j < Nstride) sperm[j] = SP_NONE;
    if (order[0]) { if (live) { sperm[j] = (uint32_t)j; sinv[j] = (uint32_t)j; } return; }
    const uint32_t r = live ? root[j] : SP_NONE;
    if (live && r == (uint32_t)j) sp_segtiles(seg_start[j], seg_end[j], CW, gbm);     // the root's thread sets its segment's tiles
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
    if (live) { sperm[p] = (uint32_t)j; sinv[j] = p; posseg[p] = uint2{seg_start[r], seg_end[r]}; }
}

// what sp_emit_kernel hands to sp_pairs_kernel: the holders of every mixed value (the insiders, then the outsiders) in one stream, and one
// record per value: (index of its first holder in the stream, place of its pairs in the list, insiders, outsiders)
struct SpColWork { uint32_t *ents; uint4 *vals; uint32_t ecap, vcap; };
// the entry stream's and the value records' cursors as ONE aligned 64-bit word among plctl[8..10] (entries | records << 32): a step of sp_emit_kernel
// reserves both with one atomic
__device__ __forceinline__ unsigned long long *sp_pl_streams(uint32_t *plctl) { return reinterpret_cast<unsigned long long *>(plctl + 8 + (((uintptr_t)(plctl + 8) >> 2) & 1u)); }

// (b): every shared value of every column; the pairs of its holders that lie in different segments go to the pair list.  One workgroup
// per column; the column's shared values are walked in ranges of at most vcap ranks (1536; 768 from N = 65 536 on, where the two
// holder counts of a value no longer fit one word):
//   pass A  first[q] = segment of the first holder of value q, second[q] = the next segment met (compare-and-swap); every holder counts itself as
//           INSIDE the first segment or OUTSIDE it, and the holders of the second segment are counted too.  A value with an outsider is MIXED; its
//           INSIDERS are the larger of the two segments; its outsiders lie in two segments at least exactly when somebody holds it outside both.
//           A range without a mixed value -- every clean family column -- is done after this pass.  The counts stay in registers (three values per thread).
//   scans   the mixed values get consecutive places for their holders (insiders first, outsiders behind them) and for their list slots.  The first
//           value whose holders no longer fit the entry buffer (32 768: never below 65 536 sketches) ends the STEP: the next one starts there, with
//           pass A again.  The step then reserves its places in the entry stream, among the records and in the list: two atomics, in flight together.
//   pass B  the holders of the step's mixed values are placed: the sketch at the value's cursors, straight into the entry stream.
//   hand-off  the step's values leave for sp_pairs_kernel: a record per mixed value (where its holders start in the entry stream, where its
//           pairs go in the list).  The pairs themselves -- an OUTSIDER with every insider of its value, and with the later outsiders of another
//           segment -- are written by that kernel, flat over all columns (round 5 let every outsider write its own run from inside this kernel's
//           per-column latency chain: 64 eight-byte stores to 64 different lines per wave instruction, 50 ps per entry).
// A list that is full raises order[0] (dense walk): the list is then longer than an eighth of all pairs -- not a sparse matrix.
// A list that will not fit is noticed EARLY: every workgroup adds its column's pair count to plctl[2] and bumps plctl[3]; once 32
// columns are in, (pairs so far / columns so far) x columns > 1.5 x capacity raises order[0] and everybody stops at its next range.
#ifndef D2G_SP_EMIT_T
#define D2G_SP_EMIT_T 512
#endif
#ifdef D2G_SP_TRACE
// variant builds only (tools/emit_trace.py): per-workgroup time stamps of sp_emit_kernel's FIRST step -- 0 start, 1 LDS cleared, 2 pass A done, 3 counts read +
// mixed vote, 4 scans + places, 5 places reserved (the step's two global atomics), 6 pass B done, 7 = 6 (the slots are scanned with the places since the merge), 8 records written
__device__ unsigned long long g_emit_trace[4096 * 16];
#define EM_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096 && lo == 0) g_emit_trace[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define EM_STAMP(k) do { } while (0)
#endif
// 36 KB of LDS, 63 VGPRs: four workgroups per CU.  The entries of a step do not live in LDS (pass B writes them to the stream): a step holds up to
// 32 768 of them -- every column below 65 536 sketches is ONE step, pass A + pass B
// (1024 threads and 2560 values per step -- every column of config 3 ONE step, two workgroups per CU -- was measured: 35 us against 24 clean, 108 against 78 at c = 10)
constexpr uint32_t SP_EMIT_VCAP = 1536, SP_EMIT_ECAP = 32768, SP_EMIT_T = D2G_SP_EMIT_T;
__global__ __launch_bounds__(SP_EMIT_T) __attribute__((amdgpu_waves_per_eu(8, 8))) void sp_emit_kernel(const uint32_t *__restrict__ ids, size_t N, size_t Npad, uint32_t ncols, const uint32_t *__restrict__ colcnt, int split,
                                                            const uint32_t *__restrict__ seg, uint32_t *__restrict__ order,
                                                            SpColWork cw, uint32_t *__restrict__ plctl, uint32_t plcap, uint32_t *__restrict__ gaveup, int big) {
    __shared__ uint32_t first[SP_EMIT_VCAP];
    __shared__ uint32_t cnt[SP_EMIT_VCAP];                // pass A: insiders | outsiders << 16 (big: [q] and [1024 + q]); in a round: the two cursors of a mixed value
    __shared__ uint32_t mrec[SP_EMIT_VCAP];               // the step's mixed values: first entry | q << 16
    __shared__ uint32_t vpre[SP_EMIT_VCAP];               // list slots of the step's mixed values before this one
    __shared__ uint32_t second[SP_EMIT_VCAP];             // pass A: the second segment met
    __shared__ uint32_t cntb[SP_EMIT_VCAP];               // pass A: holders in the second segment
    __shared__ unsigned long long wave_tot64[16];
    __shared__ uint32_t s_base, s_ebase, s_vbase, s_stop, s_cut, s_nent, s_nm;
    if (order[0]) return;
    constexpr uint32_t T = SP_EMIT_T, PER = (SP_EMIT_VCAP + T - 1) / T;
    const uint32_t tid = threadIdx.x;
    const size_t t = blockIdx.x;
    const uint32_t d2 = colcnt[t * BS_CC_STRIDE + 4];
    if (d2 == 0) return;
    // fn(rank, segment, sketch) for every sketch of the column that holds a shared value
    // (keeping a thread's 20 (rank, segment) pairs packed in registers instead of re-reading the column in the second pass was measured at
    // N = 10 000: 89 VGPRs, two workgroups per CU instead of four, 37.9 vs 35.6 us)
    const uint32_t *__restrict__ col = ids + t * Npad;                 // (uniform base + 32-bit lane offset: N < 2^30)
    const uint32_t n32 = (uint32_t)N;
    auto for_each_holder = [&](auto &&fn) {
#pragma unroll 1
        for (uint32_t j0 = 0; j0 < n32; j0 += T * 8) {
            uint32_t w[8], sg[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const uint32_t j = j0 + (uint32_t)x * T + tid;
                w[x] = j < n32 ? col[j] : 0u;
                sg[x] = j < n32 ? seg[j] : 0u;
            }
#pragma unroll
            for (int x = 0; x < 8; ++x) { const uint32_t r = sp_rank(w[x], colcnt, t, split != 0); if (r) fn(r, sg[x], j0 + (uint32_t)x * T + tid); }
        }
    };
    const uint32_t vcap = big ? SP_EMIT_VCAP / 2 : SP_EMIT_VCAP;
    uint32_t colpairs = 0;
    bool voted = false;
    for (uint32_t lo = 0; lo < d2;) {                                // steps of whole values (uniform)
        EM_STAMP(0);
        const uint32_t len = min(vcap, d2 - lo);
        for (uint32_t x = tid; x < SP_EMIT_VCAP; x += T) { first[x] = SP_NONE; second[x] = SP_NONE; cnt[x] = 0; cntb[x] = 0; }
        if (tid == 0) s_stop = sp_ld(&order[0]);
        __syncthreads();
        if (s_stop) return;                                           // somebody found that the list will not fit
        EM_STAMP(1);
        for_each_holder([&](uint32_t r, uint32_t sgv, uint32_t) {
            if (r <= lo || r > lo + len) return;
            const uint32_t q = r - 1 - lo;
            const uint32_t old = atomicCAS(&first[q], SP_NONE, sgv);
            const bool outside = old != SP_NONE && old != sgv;
            if (big) atomicAdd(&cnt[outside ? SP_EMIT_VCAP / 2 + q : q], 1u);
            else atomicAdd(&cnt[q], outside ? 0x10000u : 1u);
            if (outside) { const uint32_t o2 = atomicCAS(&second[q], SP_NONE, sgv); if (o2 == SP_NONE || o2 == sgv) atomicAdd(&cntb[q], 1u); }
        });
        __syncthreads();
        EM_STAMP(2);
        uint32_t n0[PER], nO[PER], oo = 0;                          // oo: bit x -- the outsiders of my value x lie in two segments at least
        bool mixed_mine = false;
#pragma unroll
        for (uint32_t x = 0; x < PER; ++x) {
            const uint32_t q = tid * PER + x;
            const uint32_t c = q < len ? cnt[q] : 0u;
            n0[x] = big ? c : (c & 0xFFFFu);
            nO[x] = q < len ? (big ? cnt[SP_EMIT_VCAP / 2 + q] : (c >> 16)) : 0u;
            // the INSIDERS are the larger of the first two segments met (a stranger that happened to come first must not make the family its value's
            // "outsiders": 75 outsiders are 2 775 candidate pairs among them as soon as a second stranger joins)
            if (q < len && nO[x]) {
                const uint32_t nb = cntb[q];
                // a holder outside the first two segments met: whichever of the two holds the insiders, the outsiders lie in two segments at least --
                // pairs among them get list slots (known HERE, before pass B: every reservation of the step is made in one place)
                if (nO[x] > nb) oo |= 1u << x;
                if (nb > n0[x]) { first[q] = second[q]; nO[x] += n0[x] - nb; n0[x] = nb; }
            }
            mixed_mine |= nO[x] != 0;
        }
        if (!__syncthreads_or(mixed_mine)) { EM_STAMP(3); lo += len; continue; }   // no mixed value in this range: every clean family column ends here
        EM_STAMP(3);
        {
            // my values: entries (a thread's sum clamped to ECAP + 1) | mixed values << 32
            unsigned long long v = 0;
            uint32_t ve = 0;
#pragma unroll
            for (uint32_t x = 0; x < PER; ++x) {
                if (!nO[x]) continue;
                ve = min(ve + min(n0[x], SP_EMIT_ECAP + 1u) + min(nO[x], SP_EMIT_ECAP + 1u), SP_EMIT_ECAP + 1u);
                v += 1ull << 32;
            }
            v |= ve;
            if (tid == 0) s_cut = len;
            unsigned long long tot;
            const unsigned long long run = sp_block_scan64<T / 64>(v, wave_tot64, &tot);
            // the first value whose holders no longer fit ends the step (the prefix grows with the value: one minimum)
            if ((uint32_t)tot > SP_EMIT_ECAP) {                        // (uniform; N < 65 536: never)
                uint32_t e = (uint32_t)run;
#pragma unroll
                for (uint32_t x = 0; x < PER; ++x) {
                    if (!nO[x]) continue;
                    e = min(e + min(n0[x], SP_EMIT_ECAP + 1u) + min(nO[x], SP_EMIT_ECAP + 1u), SP_EMIT_ECAP + 1u);
                    if (e > SP_EMIT_ECAP) { atomicMin(&s_cut, tid * PER + x); break; }
                }
            }
            __syncthreads();
            const uint32_t ncut = s_cut;
            // the list slots of the step's mixed values: the pairs with insiders, + one slot per pair of outsiders where those lie in two segments at
            // least (sp_pairs_kernel fills such a slot with the pair or, where the two share a segment, with the NULL entry)
            unsigned long long pv[PER], stot;
            unsigned long long srun;
            {
                unsigned long long sum = 0;
#pragma unroll
                for (uint32_t x = 0; x < PER; ++x) {
                    pv[x] = 0;
                    if (tid * PER + x < ncut && nO[x]) pv[x] = (unsigned long long)n0[x] * nO[x] + (((oo >> x) & 1u) ? (unsigned long long)nO[x] * (nO[x] - 1u) / 2u : 0ull);
                    sum += pv[x];
                }
                srun = sp_block_scan64<T / 64>(sum, wave_tot64, &stot);
            }
            // places and cursors of the step's mixed values, the slots before each; the step's totals from whoever owns its end
            {
                uint32_t eoff = (uint32_t)run, moff = (uint32_t)(run >> 32);
#pragma unroll
                for (uint32_t x = 0; x < PER; ++x) {
                    const uint32_t q = tid * PER + x;
                    if (q == ncut) { s_nent = eoff; s_nm = moff; }
                    if (q >= ncut) continue;
                    if (!nO[x]) { cnt[q] = SP_NONE; continue; }       // (no cursor word looks like this: positions stay below 32 769)
                    mrec[moff] = eoff | (q << 16) | (((oo >> x) & 1u) << 31);
                    vpre[moff] = (uint32_t)srun;                        // (a step whose slots pass 2^32 is given up below: the list holds fewer)
                    cnt[q] = eoff | ((eoff + n0[x]) << 16);
                    ++moff; eoff += n0[x] + nO[x]; srun += pv[x];
                }
                if (ncut >= len && tid == 0) { s_nent = (uint32_t)tot; s_nm = (uint32_t)(tot >> 32); }
            }
            __syncthreads();
            EM_STAMP(4);
            const uint32_t nent = s_nent, nm = s_nm;
            if (nent == 0 || stot > 0xFFFFFFFFull) {                  // (the step's first mixed value alone exceeds the buffer: ONE value with tens of thousands of
                if (tid == 0) { order[0] = 1; *gaveup = 1; }          // holders spread over segments -- or more slots than any list has: not sparse)
                return;
            }
            const uint32_t total = (uint32_t)stot;
            // the step's places in the entry stream and among the value records (ONE 64-bit atomic) and in the list -- the two in flight together.  Round 6's
            // first form made three reservations one after the other, the third behind pass B: with all 1024 workgroups arriving together the same-address
            // atomics cost a workgroup 14 + 4.5 us of its 54 at ten chance collisions per sketch (time stamps: profiles/r06_k2_experiments.txt, section 12)
            if (tid == 0) { const unsigned long long o = atomicAdd(sp_pl_streams(plctl), (unsigned long long)nent | ((unsigned long long)nm << 32)); s_ebase = (uint32_t)o; s_vbase = (uint32_t)(o >> 32); }
            if (tid == 64) s_base = atomicAdd(&plctl[0], total);
            __syncthreads();
            EM_STAMP(5);
            const uint32_t ebase = s_ebase, vbase = s_vbase, base = s_base;
            if ((size_t)ebase + nent > cw.ecap || (size_t)vbase + nm > cw.vcap || (size_t)base + total > plcap) { if (tid == 0) { order[0] = 1; *gaveup = 1; } return; }
            for_each_holder([&](uint32_t r, uint32_t sgv, uint32_t j) {
                if (r <= lo || r > lo + len) return;
                const uint32_t q = r - 1 - lo;
                if (q >= ncut || cnt[q] == SP_NONE) return;
                const bool inside = first[q] == sgv;
                const uint32_t old = atomicAdd(&cnt[q], inside ? 1u : 0x10000u);
                cw.ents[ebase + (inside ? (old & 0xFFFFu) : (old >> 16))] = j;     // (the step's stream is a few KB written within microseconds: the lines fill up in L2)
            });
            __syncthreads();
            EM_STAMP(6);
            EM_STAMP(7);
            // the step's values leave for sp_pairs_kernel: one record per mixed value -- (first holder in the entry stream, first slot in the list, insiders, outsiders)
            {
                colpairs += total;
                for (uint32_t m = tid; m < nm; m += T) {
                    const uint32_t rec = mrec[m], c = cnt[(rec >> 16) & 0x7FFFu];
                    const uint32_t st = rec & 0xFFFFu, mid = c & 0xFFFFu, end = c >> 16;
                    // (bit 31 of the outsider count: they lie in two segments at least -- pairs among them have slots)
                    cw.vals[vbase + m] = uint4{ebase + st, base + vpre[m], mid - st, (end - mid) | (rec & 0x80000000u)};
                }
            }
            __syncthreads();
            EM_STAMP(8);
            lo += ncut;                                               // the next step starts at the first value that did not fit (pass A again from there)
            // this column's pairs so far, scaled to all of its values and all columns: 1.25 times the capacity says the list will not fit.
            // Eight columns must say so (one odd column must not send a sparse matrix to the dense walk).
            if (tid == 0 && colpairs && !voted && (size_t)colpairs * d2 / lo * ncols > (size_t)plcap + plcap / 4) {
                voted = true;
                if (atomicAdd(&plctl[4], 1u) + 1u >= 8u) { order[0] = 1; *gaveup = 1; }
            }
        }
    }
    if (tid == 0 && colpairs) {
        const uint32_t tot = atomicAdd(&plctl[2], colpairs) + colpairs, done = atomicAdd(&plctl[3], 1u) + 1u;
        if (done >= 32 && (size_t)tot / done * ncols > (size_t)plcap + plcap / 4 * 2) { order[0] = 1; *gaveup = 1; }
    }
}

// The pairs of the mixed values sp_emit_kernel left (SpColWork), FLAT: a workgroup takes 32 value records at a time (a few hundred to a few
// thousand pairs: enough workgroups for the stragglers of a clean collection too), one wave scans their pair counts, and pair p of the chunk
// is found by a search over that prefix -- every thread writes entries, consecutive threads consecutive ones, whatever the values look like
// (one family value with one stranger: 75 pairs; a value two strangers share: one).  Four pairs per thread are in flight together (their
// holders come from the entry stream: two dependent gathers each).  A value's pairs:
//   outsider o x insider i   at  the place sp_emit_kernel reserved + o * insiders + i     (no comparison: they differ by construction)
//   outsiders u < v          only where their segments differ: counted here (the same flat walk over the chunk's candidate pairs), ONE
//                            reservation per chunk, written by a second walk.  Rare -- unless the stranger was a value's FIRST holder and the
//                            family its "outsiders": then 75 outsiders make 2 775 candidates and no pair.
#ifndef D2G_SP_PAIRS_CHUNK
#define D2G_SP_PAIRS_CHUNK 32
#endif
constexpr uint32_t SP_PAIRS_CHUNK = D2G_SP_PAIRS_CHUNK;
struct SpPairs { SpColWork cw; const uint32_t *seg; uint32_t *plctl; unsigned long long *plist; uint32_t plcap; uint32_t *order, *gaveup, *fullctl; };
// workgroup wg of nwg, NT threads (256 as a kernel of its own; 1024 as extra workgroups of the permute launch: short lists)
template <uint32_t NT>
__device__ __forceinline__ void sp_pairs_body(uint32_t wg, uint32_t nwg, const SpPairs &pp) {
    __shared__ uint32_t pre[SP_PAIRS_CHUNK + 1];
    __shared__ uint4 rec[SP_PAIRS_CHUNK];
    const SpColWork &cw = pp.cw;
    const uint32_t *__restrict__ seg = pp.seg;
    const uint32_t *__restrict__ order = pp.order;
    unsigned long long *__restrict__ plist = pp.plist;
    if (order[0]) return;
    const uint32_t nv = min((uint32_t)(*sp_pl_streams(pp.plctl) >> 32), cw.vcap), tid = threadIdx.x;
    for (uint32_t v0 = wg * SP_PAIRS_CHUNK; v0 < nv; v0 += nwg * SP_PAIRS_CHUNK) {     // (uniform)
        if (tid < 64) {                                               // one wave: the chunk's records and the prefix of their slot counts
            uint32_t np = 0;
            if (tid < SP_PAIRS_CHUNK && v0 + tid < nv) {
                const uint4 r = cw.vals[v0 + tid];
                const uint32_t nO = r.w & 0x7FFFFFFFu;
                rec[tid] = r; np = r.z * nO + ((r.w >> 31) ? nO * (nO - 1u) / 2u : 0u);
            }
            uint32_t ip = np;
            for (int o = 1; o < 64; o <<= 1) { const uint32_t x = __shfl_up(ip, o); if ((int)tid >= o) ip += x; }
            if (tid < SP_PAIRS_CHUNK) pre[tid] = ip - np;
            if (tid == SP_PAIRS_CHUNK - 1) pre[SP_PAIRS_CHUNK] = ip;
        }
        __syncthreads();
        const uint32_t total = pre[SP_PAIRS_CHUNK];
        for (uint32_t p0 = tid; p0 < total; p0 += NT * 4) {
            uint32_t ja[4], jb[4], at[4];
            bool oo[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t p = p0 + NT * k;
                at[k] = 0xFFFFFFFFu; oo[k] = false;
                if (p >= total) continue;
                uint32_t a = 0, b = SP_PAIRS_CHUNK;                   // the last value whose slots start at or before p
                while (b - a > 1) { const uint32_t m = (a + b) >> 1; if (pre[m] <= p) a = m; else b = m; }
                const uint4 r = rec[a];
                const uint32_t x = p - pre[a], nI = r.z, nO = r.w & 0x7FFFFFFFu;
                at[k] = r.y + x;
                if (x < nI * nO) {
                    const uint32_t o = x / nI, i = x - o * nI;
                    ja[k] = cw.ents[r.x + nI + o]; jb[k] = cw.ents[r.x + i];
                } else {
                    uint32_t y = x - nI * nO, u = 0;                  // the y-th pair u < v of the value's outsiders
                    while (y >= nO - 1u - u) { y -= nO - 1u - u; ++u; }
                    ja[k] = cw.ents[r.x + nI + u]; jb[k] = cw.ents[r.x + nI + u + 1u + y];
                    oo[k] = true;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (at[k] != 0xFFFFFFFFu)
                    plist[at[k]] = (oo[k] && seg[ja[k]] == seg[jb[k]]) ? SP_NULL_ENTRY : ((unsigned long long)min(ja[k], jb[k]) | ((unsigned long long)max(ja[k], jb[k]) << 32));
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void sp_pairs_kernel(SpPairs pp) { sp_pairs_body<256>(blockIdx.x, gridDim.x, pp); }

// launch rows: the sorted positions whose sketch lies in [r0, r1), in sorted order (stable compaction, one workgroup; 8192 positions at
// a time through LDS so that the loads are coalesced and a thread still owns eight consecutive positions: N = 50 000 80 -> ~12 us)
__global__ __launch_bounds__(1024) void sp_rows_kernel(const uint32_t *__restrict__ sperm, size_t N, uint32_t r0, uint32_t r1, uint32_t nrows_pad,
                                                       uint32_t *__restrict__ rowpos, uint32_t *__restrict__ rowk, const uint32_t *__restrict__ order) {
    // (the dense walk was decided -- possibly before any ordering: a set's FIRST prepare may skip it after its first look, and sperm[] then holds
    // whatever the allocation held: nothing of the sorted operand may be touched.  The list kernel and the pair kernel leave on the same word.)
    if (order[0]) return;
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
            if (jv[x] >= N) continue;                                  // (SP_NONE: padding)
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
                                                        const uint32_t *__restrict__ rowpos, uint32_t nrows_pad, uint32_t *__restrict__ rowstream, size_t rstride,
                                                        const uint32_t *__restrict__ order) {
    if (order[0]) return;                                             // dense walk: no launch rows (sp_rows_kernel)
    const size_t q = blockIdx.y;
    if (q >= stream_slot(meta, ntb)) return;
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nrows_pad) return;
    const uint32_t p = rowpos[k];
    rowstream[q * rstride + k] = p != SP_NONE ? stream[2 * q * Nstride + p] : 0u;
}

// tile bitmap of a PARTIAL launch from the global one: a block of 32 launch rows may meet what any of the sorted row blocks its rows
// come from may meet (a superset of the tiles of those rows' own segments: still no pair inside a segment is missed)
__global__ __launch_bounds__(256) void sp_rowbm_kernel(const uint32_t *__restrict__ gbm, const uint32_t *__restrict__ rowpos, uint32_t nrb, uint32_t CW,
                                                       uint32_t *__restrict__ tilebm, const uint32_t *__restrict__ order) {
    if (order[0]) return;                                             // dense walk: no launch rows (sp_rows_kernel)
    const uint32_t x = blockIdx.x * 256 + threadIdx.x;
    if (x >= nrb * CW) return;
    const uint32_t rb = x / CW, cw = x - rb * CW;
    uint32_t acc = 0;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) { const uint32_t p = rowpos[rb * 32 + i]; if (p != SP_NONE) acc |= gbm[(size_t)(p >> 5) * CW + cw]; }
    tilebm[x] = acc;
}

// the marked tiles as EIGHT work lists, one per XCD: list q holds the tiles whose column block cb has cb % 8 == q, and the workgroups
// that run on XCD q (blockIdx % 8 == q) walk it -- every sub-tile that reads a column block's words runs on one XCD, whose L2 then
// fetches them once (a single list handed the four sub-tiles of a tile to four XCDs).  Any order inside a list: a workgroup reserves
// the ranges of its tiles with one atomic per list.  full: rows are ALL sorted positions and a pair is computed where row position <
// column position, so tiles entirely below that diagonal are not candidates.
// ctl[0] = tiles listed, ctl[1] bit 0 = dense walk, ctl[3] = candidates, ctl[8 + q] = tiles in list q (list q starts at tiles + q * cap).
constexpr int SP_CTL_WORDS = 16;
// workgroup `wg` of `NW` waves lists tiles [wg * 512 NW, (wg + 1) * 512 NW)
// The lists hold SUB-TILES (tile * 4 + sub: 16 launch rows x 128 sorted columns, what one workgroup of the pair kernel walks), and only
// those that can hold a pair of one segment (sp_tile_subs).  Round 5 first listed whole tiles and let the pair kernel's workgroups
// find out that their sub-tile was empty: 40 % of them left at once, the CUs were dealt between two and seven WORKING workgroups, and
// the kernel lasted as long as the fullest CU (per-workgroup time stamps: plane walk 18 us on average, 34-38 us on the fullest CUs;
// kernel 45 us for a mean workgroup life of 29).  Dense lists deal the work evenly.
__device__ __forceinline__ uint32_t sp_tile_subs(uint32_t rb, uint32_t cb, int full, uint32_t N, const uint32_t *__restrict__ rowpos, const uint2 *__restrict__ posseg) {
    constexpr uint32_t WC = SP_WC;
    uint32_t m = 0;
#pragma unroll
    for (uint32_t sb = 0; sb < SP_SUBS; ++sb) {
        const uint32_t k0 = rb * 32u + (sb / WC) * BS_IW, c0 = cb * (uint32_t)BS_CB + (sb % WC) * SP_SUBW;
        if (full && k0 > c0 + SP_SUBW - 1u) continue;                // entirely below the diagonal of sorted positions
        const uint32_t pf = full ? k0 : rowpos[k0];
        if (pf == SP_NONE || pf >= N) continue;                       // no row
        const uint32_t pl = full ? min(k0 + (uint32_t)BS_IW - 1u, N - 1u) : rowpos[k0 + BS_IW - 1];
        if (pl != SP_NONE && sp_sub_empty(posseg, pf, pl, c0)) continue;   // no pair of one segment: the pair list has what it holds (sp_entry_wanted asks the same)
        m |= 1u << sb;
    }
    return m;
}
template <int NW>
__device__ __forceinline__ void sp_list_body(uint32_t wg, const uint32_t *__restrict__ tilebm, uint32_t nrb, uint32_t ncb, uint32_t CW, int full,
                                             uint32_t *__restrict__ tiles, uint32_t cap, uint32_t *__restrict__ ctl, uint32_t cand, const uint32_t *__restrict__ order,
                                             uint32_t *__restrict__ ctl_next, uint32_t N, const uint32_t *__restrict__ rowpos, const uint2 *__restrict__ posseg) {
    __shared__ unsigned long long wave_tot[2][NW];
    __shared__ uint32_t s_base[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wg == 0 && ctl_next && tid < SP_CTL_WORDS) ctl_next[tid] = 0;   // the next launch's control words (nobody else touches them during this launch)
    if (wg == 0 && tid == 0) ctl[3] = cand;                          // for d2g_cmp_set_sparse_info
    if (order[0]) {                                                  // the prepare decided for the dense walk
        if (wg == 0 && tid == 0) atomicOr(&ctl[1], 1u);
        return;
    }
    // ONE candidate tile per thread: bitmap word, then (listed tiles only) the segments of its rows -- two dependent round trips for the
    // whole list (eight tiles per thread, one after the other, took ~25 us at config 3)
    const size_t ntile = (size_t)nrb * ncb;
    const size_t x = (size_t)wg * (64 * NW) + tid;
    uint32_t sm = 0, ntl = 0, q = 0;                                   // sm: the tile's sub-tiles to walk
    unsigned long long cnt[2] = {0, 0};                               // eight 16-bit counters: lists 0-3 | lists 4-7 (a workgroup lists at most 1024 tiles)
    if (x < ntile) {
        const uint32_t rb = (uint32_t)(x / ncb), cb = (uint32_t)(x % ncb);
        q = cb & 7u;
        if (!(full && (size_t)rb * 32 > (size_t)cb * 256 + 255) && ((tilebm[(size_t)rb * CW + (cb >> 5)] >> (cb & 31)) & 1u)) {
            sm = sp_tile_subs(rb, cb, full, N, rowpos, posseg);
            ntl = 1;                                                  // (a listed tile counts as listed even when none of its sub-tiles is walked: sp_dense_mode, sparse_info)
            cnt[(cb >> 2) & 1] = (unsigned long long)__popc(sm) << (16 * (cb & 3));
        }
    }
    for (int o = 32; o > 0; o >>= 1) ntl += __shfl_down(ntl, o);
    if (lane == 0 && ntl) atomicAdd(&ctl[0], ntl);
    // block-wide exclusive prefix of the eight counters, two packed scans
    unsigned long long ex[2], tot[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        unsigned long long incl = cnt[h];
        for (int o = 1; o < 64; o <<= 1) { const unsigned long long x = __shfl_up(incl, o); if (lane >= o) incl += x; }
        if (lane == 63) wave_tot[h][wave] = incl;
        ex[h] = incl - cnt[h];
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        unsigned long long woff = 0, t = 0;
        for (int w = 0; w < NW; ++w) { const unsigned long long x = wave_tot[h][w]; if (w < wave) woff += x; t += x; }
        ex[h] += woff; tot[h] = t;
    }
    if (tid < 8) {
        const uint32_t n = (uint32_t)(tot[tid >> 2] >> (16 * (tid & 3))) & 0xFFFFu;
        s_base[tid] = n ? atomicAdd(&ctl[8 + tid], n) : 0u;
    }
    __syncthreads();
    if (sm) {
        uint32_t o = s_base[q] + ((uint32_t)(ex[q >> 2] >> (16 * (q & 3))) & 0xFFFFu);
#pragma unroll
        for (uint32_t sb = 0; sb < SP_SUBS; ++sb)
            if ((sm >> sb) & 1u) tiles[(size_t)q * cap + o++] = (uint32_t)x * SP_SUBS + sb;
    }
}
// a partial launch's lists (its own tile bitmap); a whole-triangle launch uses the lists the prepare left (sp_permute_kernel)
__global__ __launch_bounds__(1024) void sp_list_kernel(const uint32_t *__restrict__ tilebm, uint32_t nrb, uint32_t ncb, uint32_t CW, int full,
                                                       uint32_t *__restrict__ tiles, uint32_t cap, uint32_t *__restrict__ ctl, uint32_t cand, const uint32_t *__restrict__ order,
                                                       uint32_t *__restrict__ ctl_next, uint32_t N, const uint32_t *__restrict__ rowpos, const uint2 *__restrict__ posseg) {
    sp_list_body<16>(blockIdx.x, tilebm, nrb, ncb, CW, full, tiles, cap, ctl, cand, order, ctl_next, N, rowpos, posseg);
}

// ---- the pair list, BINNED by where its entries land in the output (round 6).  Round 5 applied the list with one global atomicAdd per entry plus a
// second "leader" pass for the table epilogue: 78 ps per entry (44 + 35), 0.31 ms for the 3.9 million entries of ten chance collisions per sketch at
// config 3 -- 1 GB of scattered read-modify-write traffic.  Now the entries are grouped by output region, and ONE workgroup composes a region in LDS
// and writes it with coalesced stores (sp_compose_kernel):
//   bin(i, j) = (i >> 5) * nch + (j >> cshift)      band of 32 output rows (i = the smaller caller index = the output row) x chunk of 2^cshift columns
//   sp_hist_body   (extra workgroups of the permute launch: the list is final once sp_emit_kernel is done)  entries per bin -> binc[]
//   sp_bin_kernel  exclusive prefix of binc[] (every workgroup for itself, in LDS), then its slice of the list moved to plist2 bin by bin
//                  (per 8192 entries: LDS counts, one global reservation per bin and workgroup, LDS cursors)
//   sp_compose_kernel (launch)  one workgroup per bin of the launch's bands: 32 x 1024 counts in LDS (16 bits each), entries added with LDS
//                  atomics, then every 64-word span that holds a count is written -- table value or count, the fill value beside it.
// The compose kernel runs BEFORE the pair kernel, which STORES its tiles' non-zero counts: a cross-segment pair inside a walked sub-tile is
// written twice with the same count (its entries ARE its equal registers, fact (b)), so no entry has to ask whether its tile is listed.
struct SpBins { uint32_t *binc, *bstart; uint32_t nbins, nch, cshift; };
__device__ __forceinline__ uint32_t sp_bin_of(const SpBins &bn, unsigned long long e) {
    return ((uint32_t)e >> 5) * bn.nch + ((uint32_t)(e >> 32) >> bn.cshift);
}
// workgroup hw of nhw (1024 threads): its slice of the list counted into LDS, then one global add per bin it met -- whose RETURN value is the
// place of this workgroup's entries inside the bin (hoff[hw][bin]): sp_bin_kernel's workgroup hw moves the same slice and needs no atomics of its own
__device__ __forceinline__ uint32_t sp_slice(uint32_t n, uint32_t nhw) { return ((n + nhw - 1) / nhw + 1023u) & ~1023u; }
__device__ __forceinline__ void sp_hist_body(uint32_t hw, uint32_t nhw, const unsigned long long *__restrict__ plist, const uint32_t *__restrict__ plctl, uint32_t plcap,
                                             const SpBins &bn, uint32_t *__restrict__ hoff, uint32_t *lds) {
    const uint32_t n = min(plctl[0], plcap);
    const uint32_t per = sp_slice(n, nhw), lo = hw * per, hi = min(n, lo + per);
    if (lo >= hi) return;
    for (uint32_t b = threadIdx.x; b < bn.nbins; b += 1024) lds[b] = 0;
    __syncthreads();
    // (eight entries per thread in flight: a load behind every LDS atomic waited out its round trip -- 26 -> 22 us for the launch at ten chance collisions per sketch)
    for (uint32_t k0 = lo + threadIdx.x; k0 < hi; k0 += 8 * 1024) {
        unsigned long long e8[8];
#pragma unroll
        for (uint32_t x = 0; x < 8; ++x) { const uint32_t k = k0 + x * 1024; e8[x] = k < hi ? plist[k] : SP_NULL_ENTRY; }
#pragma unroll
        for (uint32_t x = 0; x < 8; ++x) if (e8[x] != SP_NULL_ENTRY) atomicAdd(&lds[sp_bin_of(bn, e8[x])], 1u);
    }
    __syncthreads();
    uint32_t *mine = hoff + (size_t)hw * bn.nbins;
    for (uint32_t b = threadIdx.x; b < bn.nbins; b += 1024) { const uint32_t c = lds[b]; if (c) mine[b] = atomicAdd(&bn.binc[b], c); }
}

// the sorted stream from the caller's-order stream: position p takes the words of sketch sperm[p], THROUGH LDS: a workgroup takes ONE plane of the
// caller's-order stream -- (group, plane): its row-coded and its column-coded words, 2 x Nstride -- reads it once, coalesced, into LDS, and every
// sorted position p fetches the word of sketch sperm[p] from there (per-lane gathers from global memory pulled 148 MB of 64-byte sectors for 9 MB
// of words at config 3: 21 us against 13).  Planes larger than the LDS (N > ~16 000) are taken in H parts: part h holds the words of the
// sketches [h part, (h + 1) part) and writes the positions whose sketch lies there.  Non-live planes leave at once (the grid is sized for
// nbits_cap planes per group: the live count is on the device).
// BOTH (N <= ~20 000: both codings of a plane fit the LDS): one workgroup per (group, plane), one pass over sperm serves both codings, the
// live workgroups of config 3 (224) are resident together.  Otherwise one workgroup per (group, plane, coding, part).
// The workgroups behind the permuting ones (a) leave the work lists of a whole-triangle launch (the segments' tiles are final once
// sp_emit_kernel is done: such a launch has no list kernel of its own) and (b) count the pair list's entries per output bin (sp_hist_body).
struct SpFullList { const uint32_t *bm; uint32_t nrb, ncb, CW; uint32_t *tiles; uint32_t cap; uint32_t *ctl; uint32_t cand, nwg; uint32_t N; const uint2 *posseg; };
struct SpHist { const unsigned long long *plist; const uint32_t *plctl; uint32_t plcap, nhw; SpBins bn; uint32_t *hoff; int big; uint32_t *stat_n; };   // big: the workgroups behind the list builders count the list per bin; otherwise they make the pairs (nhw of them)
template <bool BOTH>
__global__ __launch_bounds__(1024) void sp_permute_lds_kernel(const uint32_t *__restrict__ nat, uint32_t *__restrict__ srt, size_t Nstride, const uint32_t *__restrict__ meta,
                                                              const uint32_t *__restrict__ sperm, const uint32_t *__restrict__ order, SpFullList fl, uint32_t nperm,
                                                              uint32_t nbits_cap, uint32_t H, uint32_t part, SpHist hs, SpPairs pp) {
    extern __shared__ __attribute__((aligned(16))) uint32_t sp_row[];
    if (blockIdx.x >= nperm) {
        const uint32_t lw = blockIdx.x - nperm;
        if (lw < fl.nwg) {
            if (lw == 0 && threadIdx.x == 0 && hs.stat_n) *hs.stat_n = order[0] ? 0xFFFFFFFFu : min(hs.plctl[0], hs.plcap);   // what the host learns for the set's NEXT prepare (mapped memory): the list's length
            sp_list_body<16>(lw, fl.bm, fl.nrb, fl.ncb, fl.CW, 1, fl.tiles, fl.cap, fl.ctl, fl.cand, order, nullptr, fl.N, nullptr, fl.posseg);
        } else if (!hs.big) sp_pairs_body<1024>(lw - fl.nwg, hs.nhw, pp);
        else if (!order[0]) sp_hist_body(lw - fl.nwg, hs.nhw, hs.plist, hs.plctl, hs.plcap, hs.bn, hs.hoff, sp_row);
        return;
    }
    if (order[0]) return;                                             // the caller's order was kept: nobody reads the sorted stream
    uint32_t w = blockIdx.x;
    const uint32_t h = w % H; w /= H;
    uint32_t coding = 0;
    if (!BOTH) { coding = w & 1u; w >>= 1; }
    const int b = (int)(w % nbits_cap), tb = (int)(w / nbits_cap);
    if (b >= live_planes(meta, tb)) return;
    // the plane's row-coded words in sp_row[0, part), BOTH: the column-coded ones behind them
    const size_t rowi = (stream_slot_wave(meta, tb) + (size_t)b) * 2 + coding;
    const uint32_t *src = nat + rowi * Nstride;
    uint32_t *dst = srt + rowi * Nstride;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t lo = h * part, hi = (uint32_t)min((size_t)lo + part, Nstride);   // (part and Nstride are multiples of 4, rows start 16-byte aligned)
    for (uint32_t i = lo + threadIdx.x * 4; i < hi; i += 4096) {
        *reinterpret_cast<u32x4 *>(&sp_row[i - lo]) = *reinterpret_cast<const u32x4 *>(&src[i]);
        if (BOTH) *reinterpret_cast<u32x4 *>(&sp_row[part + i - lo]) = *reinterpret_cast<const u32x4 *>(&src[Nstride + i]);
    }
    __syncthreads();
    for (uint32_t p0 = threadIdx.x; p0 < Nstride; p0 += 4096) {
        uint32_t j[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) { const uint32_t p = p0 + x * 1024; j[x] = p < Nstride ? sperm[p] : SP_NONE - 1u; }
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const uint32_t p = p0 + x * 1024;
            if (j[x] == SP_NONE) { if (h == 0) { dst[p] = 0; if (BOTH) dst[Nstride + p] = 0; } }
            else if (j[x] >= lo && j[x] < hi) { dst[p] = sp_row[j[x] - lo]; if (BOTH) dst[Nstride + p] = sp_row[part + j[x] - lo]; }
        }
    }
}

// dense or sparse?  DENSE: the plain pair kernel walks every tile of the caller's-order operand (and writes every output itself);
// otherwise the output is pre-filled, the sparse kernel walks the list and the pair list is applied.  Dense when the prepare said so
// (order[0], latched into ctl[1] by the list kernel) or when more than `cand * 0.4` tiles are listed -- the sparse kernel pays for its
// generality with a per-element epilogue.  Evaluated by every consumer from the same two words: no kernel of its own.
__device__ __forceinline__ bool sp_dense_mode(const uint32_t *__restrict__ ctl, uint32_t cand) {
    return (ctl[1] & 1u) != 0 || (size_t)ctl[0] * 5 > (size_t)cand * 2;
}

constexpr int SP_FILL_PER_THREAD = 8, SP_FILL_THREADS = 256;
template <class Store>
__global__ __launch_bounds__(SP_FILL_THREADS) void sp_fill_kernel(uint32_t *__restrict__ out, size_t cnt, Store store, uint32_t S, const uint32_t *__restrict__ ctl, uint32_t cand, uint32_t piece0) {
    if (ctl && sp_dense_mode(ctl, cand)) return;                    // dense mode: the pair kernel writes every output (ctl == nullptr: an early fill, before the mode is known)
    const uint32_t v = store.value_from_mismatches(S, S);           // the value of "no register equal"
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    // the output pointer of a slab is only 4-byte aligned in general: head, 16-byte body, tail
    const size_t head = min(cnt, (size_t)((16 - ((uintptr_t)out & 15)) & 15) / 4);
    u32x4 *body = reinterpret_cast<u32x4 *>(out + head);
    const size_t nb = (cnt - head) / 4;
    // a workgroup writes ONE contiguous 32 KB piece (8 x 256 16-byte stores), the workgroups in dispatch order: a streaming write
    const size_t base = ((size_t)blockIdx.x + piece0) * (SP_FILL_THREADS * SP_FILL_PER_THREAD) + threadIdx.x;   // (piece0: the pieces before it rode on the prepare's kernels)
    // NON-TEMPORAL stores (round 5): the fill is 200 MB at config 3 that nobody reads again; written through the MALL (256 MB) it pushed the
    // prepare's working set out -- columns 82 MB, ids 41, plane streams 37 -- and the caller's matrix with it: the NEXT kernels paid (the transpose
    // of the next step 46 us instead of 31).  With `nt` stores: step 0.287 -> 0.273 ms, + non-temporal loads of the caller's rows in the transpose 0.267
    // (unrelated 0.173 -> 0.159, + 1 collision 0.356 -> 0.341, dense walk 0.523 -> 0.517; the sparse pair kernel pays 37 -> 42 us: its tiles' partial lines
    // now merge in HBM)
#pragma unroll
    for (int k = 0; k < SP_FILL_PER_THREAD; ++k) { const size_t i = base + (size_t)k * SP_FILL_THREADS; if (i < nb) __builtin_nontemporal_store(u32x4{v, v, v, v}, &body[i]); }
    if (blockIdx.x + piece0 == 0) {
        if (threadIdx.x < head) out[threadIdx.x] = v;
        const size_t tail0 = head + nb * 4;
        if (tail0 + threadIdx.x < cnt) out[tail0 + threadIdx.x] = v;
    }
}

// the pair list bin by bin (see SpBins).  Every workgroup scans binc[] for itself (<= 39 936 bins: 39 per thread), workgroup 0 leaves the prefix
// for the compose kernel; then it moves the slice its counting twin (sp_hist_body, same index) counted: the place of an entry is the bin's start
// + what the twin's atomic returned + an LDS cursor.  One pass, no global atomics.
constexpr uint32_t SP_BIN_MAX = 39 * 1024;    // bins a set may have: their cursors live in the LDS of one workgroup (156 KB); 50 000 sketches: 1563 bands x 25 chunks of 2048 columns
__global__ __launch_bounds__(1024) void sp_bin_kernel(const unsigned long long *__restrict__ plist, unsigned long long *__restrict__ plist2, uint32_t *__restrict__ plctl,
                                                      uint32_t plcap, SpBins bn, const uint32_t *__restrict__ hoff, const uint32_t *__restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lc[];     // [nbins] cursors
    __shared__ uint32_t wave_tot[16];
    if (order[0]) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) plctl[SP_PL_BINNED] = 1; // the launches' compose kernel applies the list; the entry-by-entry form stands back
    const uint32_t n = min(plctl[0], plcap);
    if (n == 0) return;                                               // (binc[] is all zero: no compose workgroup looks at bstart[])
    const uint32_t tid = threadIdx.x, nb = bn.nbins;
    const uint32_t per = sp_slice(n, gridDim.x), lo = blockIdx.x * per, hi = min(n, lo + per);
    if (lo >= hi && blockIdx.x) return;
    {
        constexpr uint32_t PER = SP_BIN_MAX / 1024;
        uint32_t v[PER], sum = 0;
#pragma unroll
        for (uint32_t x = 0; x < PER; ++x) { const uint32_t b = tid * PER + x; v[x] = b < nb ? bn.binc[b] : 0u; sum += v[x]; }
        uint32_t total;
        uint32_t run = sp_block_scan(sum, wave_tot, &total);
        const uint32_t *mine = hoff + (size_t)blockIdx.x * nb;
#pragma unroll
        for (uint32_t x = 0; x < PER; ++x) {
            const uint32_t b = tid * PER + x;
            if (b < nb) { lc[b] = run + (v[x] ? mine[b] : 0u); if (blockIdx.x == 0) bn.bstart[b] = run; }   // (mine[b] is garbage where this workgroup met no entry of b: never used)
            run += v[x];
        }
    }
    __syncthreads();
    for (uint32_t k0 = lo; k0 < hi; k0 += 4096) {
        unsigned long long e[4];
#pragma unroll
        for (uint32_t x = 0; x < 4; ++x) { const uint32_t k = k0 + x * 1024 + tid; if (k < hi) e[x] = plist[k]; }
#pragma unroll
        for (uint32_t x = 0; x < 4; ++x) { const uint32_t k = k0 + x * 1024 + tid; if (k < hi && e[x] != SP_NULL_ENTRY) plist2[atomicAdd(&lc[sp_bin_of(bn, e[x])], 1u)] = e[x]; }
    }
}

// One workgroup (256 threads) per (bin of the launch's bands, piece of 1024 columns of its chunk, group of 8 of the band's 32 rows).  LDS: 8 rows x
// 1024 columns of 16-bit counts (a count stays below 2^16: S < 65536); word [r][w] holds column w (low half) and column w + 512 (high half).
// The workgroup reads its bin's entries (they sit in L2; the four row groups -- and, for the wider chunks of large N, every piece -- read them),
// adds its own with LDS atomics, then
//   few entries    every entry's thread writes its pair's value (a pair with several entries: several threads, the same value);
//   otherwise      row by row, every 64-word span that holds a count is written -- table value or count, the fill value beside it: the four waves
//                  write 4 KB of ONE output row together (32 x 256 regions left 1 KB runs 40 KB apart: 2.7 TB/s at ten chance collisions per sketch).
// Entries of rows outside [r0, r1) -- the edge bands of a partial launch -- are skipped; words that hold no pair (j <= i, j >= N) are not stored.
struct SpComposeArgs { const unsigned long long *plist2; SpBins bn; const uint32_t *ctl; uint32_t cand, N, S, r0, r1, band0, ppb; };
constexpr uint32_t SP_CMP_ROWS = 8, SP_CMP_COLS = 1024, SP_CMP_T = 256;
template <class Store>
__global__ __launch_bounds__(SP_CMP_T) void sp_compose_kernel(SpComposeArgs a, PairShape sh, Store store) {
    __shared__ uint32_t tile[SP_CMP_ROWS * SP_CMP_COLS / 2];          // 16 KB
    __shared__ uint32_t s_cnt;
    constexpr uint32_t HW = SP_CMP_COLS / 2, T = SP_CMP_T, RG = 32 / SP_CMP_ROWS;
    if (sp_dense_mode(a.ctl, a.cand)) return;
    const uint32_t rg = blockIdx.x % RG, piece = (blockIdx.x / RG) % a.ppb, bi = blockIdx.x / (RG * a.ppb);
    const uint32_t band = a.band0 + bi / a.bn.nch, ch = bi % a.bn.nch;
    const uint32_t bin = band * a.bn.nch + ch;
    const uint32_t nb = a.bn.binc[bin];
    if (nb == 0) return;
    const uint32_t i0 = band * 32u + rg * SP_CMP_ROWS, c0 = (ch << a.bn.cshift) + piece * SP_CMP_COLS;
    if (c0 >= a.N || c0 + (SP_CMP_COLS - 1u) <= i0 || i0 >= a.r1 || i0 + SP_CMP_ROWS <= a.r0) return;   // no column, every column at or below the first row, or no row of the launch
    const unsigned long long *ent = a.plist2 + a.bn.bstart[bin];
    const uint32_t tid = threadIdx.x;
    for (uint32_t x = tid; x < SP_CMP_ROWS * HW; x += T) tile[x] = 0;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    uint32_t mine = 0;
    // (eight entries per thread in flight, as in sp_hist_body: 68 -> 59 us at ten chance collisions per sketch)
    for (uint32_t k0 = tid; k0 < nb; k0 += 8 * T) {
        unsigned long long e8[8];
#pragma unroll
        for (uint32_t x = 0; x < 8; ++x) { const uint32_t k = k0 + x * T; e8[x] = k < nb ? ent[k] : SP_NULL_ENTRY; }
#pragma unroll
        for (uint32_t x = 0; x < 8; ++x) {
            const uint32_t i = (uint32_t)e8[x], r = i - i0, c = (uint32_t)(e8[x] >> 32) - c0;
            if (c < SP_CMP_COLS && r < SP_CMP_ROWS && i >= a.r0 && i < a.r1) { atomicAdd(&tile[r * HW + (c % HW)], 1u << (16u * (c / HW))); ++mine; }
        }
    }
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o);
    if ((tid & 63) == 0 && mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    const uint32_t total = s_cnt;
    if (total == 0) return;
    if (total <= T) {                                                 // few: one store per entry
        for (uint32_t k = tid; k < nb; k += T) {
            const unsigned long long e = ent[k];
            const uint32_t i = (uint32_t)e, j = (uint32_t)(e >> 32), r = i - i0, c = j - c0;
            if (c < SP_CMP_COLS && r < SP_CMP_ROWS && i >= a.r0 && i < a.r1) {
                const uint32_t cnt = (tile[r * HW + (c % HW)] >> (16u * (c / HW))) & 0xFFFFu;
                store.put_row(out_row_base(sh, i), j, store.value_from_mismatches(a.S, a.S - min(cnt, a.S)));
            }
        }
        return;
    }
    for (uint32_t r = 0; r < SP_CMP_ROWS; ++r) {
        const uint32_t i = i0 + r;                                    // (uniform)
        if (i < a.r0 || i >= a.r1) continue;
        const size_t rb = out_row_base(sh, i);
        const uint32_t w0 = tile[r * HW + tid], w1 = tile[r * HW + tid + T];
        // columns tid, tid + 256, tid + 512, tid + 768 of the piece: the four values first (table gathers in flight together: one after the other, each in
        // front of its store, they were 32 dependent round trips per thread -- 72 us for the kernel at ten chance collisions per sketch), then the stores
        uint32_t cnt[4], val[4];
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const uint32_t w = (q & 1u) ? w1 : w0;
            cnt[q] = (w >> (16u * (q >> 1))) & 0xFFFFu;
            val[q] = store.value_from_mismatches(a.S, a.S - min(cnt[q], a.S));    // (count 0: the fill value)
        }
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const uint32_t j = c0 + tid + T * q;
            if (__ballot(cnt[q] != 0) == 0) continue;                 // nothing in this 64-word span: the fill stays
#ifdef D2G_CMP_NT
            if (j > i && j < a.N) __builtin_nontemporal_store(val[q], reinterpret_cast<uint32_t *>(store.out) + rb + j);
#else
            if (j > i && j < a.N) store.put_row(rb, j, val[q]);
#endif
        }
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
                                         uint32_t *red_lane) {
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
    // the group's mismatch counts go straight to the sub-tile's LDS words (red[i][lane]: group 0 | group 1 << 16): no accumulators carried through the
    // walk -- 32 VGPRs less, which is what lets EIGHT waves share a sub-tile with four such workgroups resident per CU (see the kernel)
#pragma unroll
    for (int i = 0; i < BS_IW; ++i) {
        const uint32_t v = JR == 2 ? ((uint32_t)__builtin_popcount(z[i][0]) | ((uint32_t)__builtin_popcount(z[i][JR - 1]) << 16)) : (uint32_t)__builtin_popcount(z[i][0]);
        atomicAdd(&red_lane[i * 64], v);
    }
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
    uint32_t ncb, cand, tiles_cap;
    const uint2 *posseg;          // (start, end) of the segment of every sorted position
};

// The sparse pair kernel.  A listed tile (32 launch rows x 256 sorted columns) is four 16 x 128 sub-tiles; a workgroup takes ONE
// sub-tile and its eight waves (D2G_SP_KS) each walk an eighth of the 32-register groups, adding every group's mismatch counts in LDS.  (The dense
// kernel gives every wave a sub-tile and all groups: with a few hundred listed tiles that leaves one or two waves per SIMD, each
// waiting out the latency of every plane's loads -- measured 87 us for 432 tiles at config 3, 411 us for 2122 at config 4.)
#ifdef D2G_SP_TRACE
// variant builds only (tools/build_variant.sh trace -DD2G_SP_TRACE; tools/sp_trace.py): per-workgroup time stamps of the sparse pair
// kernel -- 0 start, 1 sub-tile found + LDS cleared, 2 plane walk done, 3 LDS reduction done, 4 epilogue done, 5 loop left, 6 tail done
__device__ unsigned long long g_sp_trace[16384 * 8];
#define SP_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 16384) g_sp_trace[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SP_STAMP(k) do { } while (0)
#endif
template <int JR, class Store>
__global__ __launch_bounds__(64 * D2G_SP_KS) __attribute__((amdgpu_waves_per_eu(D2G_SP_WPE))) void k2_bitslice_sparse_kernel(SpArgs a, PairShape sh, Store store, SpPatchArgs pa) {
    constexpr int IW = BS_IW;
    constexpr int WC = BS_CB / (64 * JR);
    constexpr int KS = D2G_SP_KS;                                   // waves per sub-tile = splits of the group range
    static_assert(JR == 1 || JR == 2, "the LDS reduction holds one word per row and lane: one count, or two packed");
    static_assert(JR == SP_JR, "the work lists are made of SP_JR-wide sub-tiles (sp_tile_subs)");
    __shared__ uint32_t red[IW][64];                                // per row and lane: mismatches of column group 0 | group 1 << 16 (a sum stays below 2^16: S < 65536 asserted by the host)
    SP_STAMP(0);
    if (sp_dense_mode(a.ctl, a.cand)) return;                       // dense mode
    // the workgroups of XCD q (blockIdx % 8: the hardware deals workgroups to the XCDs round-robin) walk list q
    const uint32_t xq = blockIdx.x & 7u;
    // The eight lists differ in length (119-136 sub-tiles at config 3) and a CU of a long list's XCD is dealt a FIFTH working workgroup, which
    // -- everybody resident from the first cycle, the plane walk bound by VALU issue -- finishes 40 % later than the rest (time stamps: 40 us
    // against 33).  So a list longer than the mean hands its surplus to the XCDs with shorter ones: everybody walks `even` = ceil(total / 8)
    // places; a place behind the end of the own list takes the k-th handed-over sub-tile (k counted over the XCDs' free places in order).
    uint32_t even, own = 0, before = 0;
    {
        uint32_t nq[8], total = 0;
#pragma unroll
        for (int p = 0; p < 8; ++p) { nq[p] = a.ctl[8 + p]; total += nq[p]; }
        even = (total + 7u) >> 3;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            if ((uint32_t)p == xq) own = min(nq[p], even);
            if ((uint32_t)p < xq) before += even > nq[p] ? even - nq[p] : 0u;
        }
    }
    const uint32_t nsub = even;                                       // places this XCD's workgroups walk (the lists hold sub-tiles)
    const uint32_t *mytiles = a.tiles + (size_t)xq * a.tiles_cap;
    const int ks = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    // the lane number where it is needed again BEHIND the plane walk (zeroing, reduction, epilogue, tail): two instructions there instead of a
    // register carried through the walk, which has none to spare (the kernel ran with 16 bytes of scratch per lane for such values; 8 are left:
    // a 64-bit constant of the epilogue's position arithmetic, stored and reloaded once per sub-tile, outside the walk)
    auto lane_again = []() -> uint32_t { uint32_t l; asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l)); return l; };
    const bool full = a.rowstream == nullptr;
    const int g0 = a.ntb * ks / KS, g1 = a.ntb * (ks + 1) / KS;
    // first plane slot of this wave's groups: one wave-wide prefix over the groups' live plane counts (a scalar loop over up to 24
    // dependent loads before the first tile otherwise).  (Touching a sub-tile's operands once before the plane walk, all requests in
    // flight together, was measured: no gain -- the words are L2-resident already; the kernel runs at ~60 % of the dense kernel's issue rate.)
    size_t slot0 = 0;
    if ((blockIdx.x >> 3) >= nsub) {
        // nothing in this workgroup's list at its index (the grid is oversubscribed): straight to the pair list's tail
    } else if (a.ntb <= 64) {
        uint32_t incl = lane < a.ntb ? (uint32_t)live_planes(a.meta, lane) : 0u;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t x = __shfl_up(incl, o); if (lane >= o) incl += x; }
        slot0 = __builtin_amdgcn_readfirstlane(g0 ? __shfl(incl, g0 - 1) : 0u);
    } else {
        slot0 = stream_slot(a.meta, g0);
    }
    for (uint32_t si = blockIdx.x >> 3; si < nsub; si += gridDim.x >> 3) {
        uint32_t ent = 0;
        if (si < own) ent = mytiles[si];
        else {
            uint32_t k = before + (si - own);
            bool found = false;
#pragma unroll
            for (int p = 0; p < 8; ++p) {                             // (the eight lengths are read again here: held across the plane walk they cost eight SGPRs the kernel does not have)
                const uint32_t np = a.ctl[8 + p];
                const uint32_t surplus = np > even ? np - even : 0u;
                if (!found) {
                    if (k < surplus) { ent = a.tiles[(size_t)p * a.tiles_cap + even + k]; found = true; }
                    else k -= surplus;
                }
            }
            if (!found) continue;                                     // (more free places than sub-tiles handed over)
        }
        const uint32_t tile = ent / SP_SUBS, sub = ent % SP_SUBS;
        const uint32_t rb = tile / a.ncb, cb = tile - rb * a.ncb;
        const size_t k0 = (size_t)rb * 32 + (size_t)(sub / WC) * IW;              // first launch row of this sub-tile
        const size_t c0 = (size_t)cb * BS_CB + (size_t)(sub % WC) * (64 * JR);   // first sorted column position
        // (the list holds only sub-tiles with a row, on or above the diagonal of sorted positions, that can hold a pair of one segment: sp_tile_subs)
        for (int x = ks * 64 + (int)lane_again(); x < IW * 64; x += 64 * KS) (&red[0][0])[x] = 0;
        __syncthreads();
        SP_STAMP(1);
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
                sp_group<JR>(nbits, rp, rstep, cp, coff, cstep, nx, &red[0][0] + lane);
            }
            // (every wave has added its groups' mismatch counts in LDS: sp_group); afterwards wave ks finishes rows [ks IW/KS, (ks+1) IW/KS) -- the
            // epilogue (caller's indices, condensed position, table value, store) is ~45 instructions per pair and would otherwise be one wave's
            // work while the others wait
            SP_STAMP(2);
        }
        __syncthreads();
        SP_STAMP(3);
        {
            const uint32_t el = lane_again();                         // what the epilogue derives from the lane is computed here, not carried through the plane walk
            uint32_t oj[JR];
#pragma unroll
            for (int c = 0; c < JR; ++c) oj[c] = a.sperm[c0 + el + 64 * c];
            // the rows' caller indices: all of this wave's scalar loads in flight together (one after the other they were four dependent
            // round trips in front of the stores)
            constexpr int RW = IW / KS;
            uint32_t rps[RW], ois[RW];
#pragma unroll
            for (int r = 0; r < RW; ++r) { const size_t k = k0 + (size_t)(ks * RW + r); rps[r] = full ? (uint32_t)k : a.rowpos[k]; }
#pragma unroll
            for (int r = 0; r < RW; ++r) ois[r] = a.sperm[(rps[r] == SP_NONE || rps[r] >= a.N) ? 0u : rps[r]];
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const int i = ks * RW + r;
                const uint32_t rpos = rps[r];                                         // uniform
                if (rpos == SP_NONE || rpos >= a.N) continue;
                const uint32_t oi = ois[r];                                           // uniform
#pragma unroll
                for (int c = 0; c < JR; ++c) {
                    const uint32_t mm = JR == 2 ? (red[i][el] >> (16 * c)) & 0xFFFFu : red[i][el];
                    if (mm == a.S || oj[c] == SP_NONE) continue;
                    const bool want = full ? rpos < (uint32_t)(c0 + el + 64 * c) : oj[c] > oi;
                    if (!want) continue;
                    const uint32_t lo = min(oi, oj[c]), hi = max(oi, oj[c]);
                    store.put(out_pos(sh, lo, hi), store.value_from_mismatches(a.S, mm));
                }
            }
        }
        __syncthreads();
        SP_STAMP(4);
    }
    SP_STAMP(5);
    // a SHORT pair list, entry by entry, spread over all workgroups of the launch (those without a tile start here at once); a binned list was
    // applied by sp_compose_kernel before this kernel
    if (blockIdx.x < pa.nwg) sp_patch_add(pa, sh, store, a.S, (size_t)blockIdx.x * (64 * KS) + ks * 64 + lane_again(), (size_t)pa.nwg * (64 * KS));
    SP_STAMP(6);
}

// (measured in round 5 and dropped: the same kernel with the row words of a sub-tile staged through LDS and a 4-deep ring of column
// words per wave -- 128 VGPRs, 20 KB of LDS, waves_per_eu 4 -- to get more plane loads in flight: 53.9 vs 53.8 us at config 3, 329 vs
// 297 us at N = 50 000)

// ---- host side
constexpr size_t SP_UNITE_STRIDE = 4;     // every 4th column pair takes part in the uniting pass (all of them: 18 -> 10 us at config 3, round 5)
constexpr size_t SP_GRID_MULT = 4;        // workgroups of the sparse pair kernel, in units of what is resident at once (exactly one resident wave of them: 52 -> 85 us)
struct SpTuning {
    bool sparse = true;                 // D2G_BS_SPARSE: 0 = every launch walks every tile
    size_t min_n = 8192;                // D2G_BS_SPARSE_MIN_N: below ~6000 sketches the extra launches cost more than the tiles they skip
    int link = 1;                       // D2G_SP_LINK: 0 = no families (every sketch its own segment: the pair list alone; tests)
    double tile_frac = 0.35;            // D2G_SP_TILE_FRAC: the segments may cover this fraction of all tiles before the dense walk is cheaper
    int olink = 1;                      // D2G_SP_OLINK: 0 = the table form of the link passes even where the rank kernel left an owner per value (tests: the multi-GPU engine's form)
    int emit_big = 0;                   // D2G_SP_EMIT_BIG: sp_emit_kernel counts with two words per value at every N (it does from N = 65 536 on; tests)
    int ride = 63;                      // D2G_SP_RIDE: which kernels of the prepare carry an announced output's fill (d2g_cmp_ut_announce_dev) -- 1 column plan, 2 flatten, 4 count, 8 attach, 16 scan, 32 place; 0 = none, the launch fills (measurements)
    int remember = 1;                   // D2G_SP_REMEMBER: 0 = every prepare runs the ordering, whatever the last one decided
    size_t long_list = 786432;          // D2G_SP_LONG_LIST: a pair list of this many entries or more is binned and composed (the last prepare's length decides)
    int predict = 1;                    // D2G_SP_PREDICT: 0 = no sample before the ordering of a set's first prepare (the ordering finds out by itself, as in round 5)
    int list_form = 0;                  // D2G_SP_LIST_FORM: 1 = always entry by entry, 2 = always binned (tests, measurements)
    size_t list_div = 8;                // D2G_SP_LIST_DIV: the pair list holds at most pairs / list_div entries (and at most 2^27)
};
SpTuning sp_tuning(const d2g_ctx *ctx) {
    SpTuning v;
    if (const char *e = ctx->tune.get("D2G_BS_SPARSE")) v.sparse = !(e[0] == '0');
    if (const char *e = ctx->tune.get("D2G_BS_SPARSE_MIN_N")) v.min_n = (size_t)std::atoll(e);
    if (const char *e = ctx->tune.get("D2G_SP_LINK")) v.link = std::atoi(e) != 0;
    if (const char *e = ctx->tune.get("D2G_SP_TILE_FRAC")) { const double f = std::atof(e); if (f > 0 && f <= 1) v.tile_frac = f; }
    if (const char *e = ctx->tune.get("D2G_SP_OLINK")) v.olink = std::atoi(e) != 0;
    if (const char *e = ctx->tune.get("D2G_SP_EMIT_BIG")) v.emit_big = std::atoi(e) != 0;
    if (const char *e = ctx->tune.get("D2G_SP_REMEMBER")) v.remember = std::atoi(e) != 0;
    if (const char *e = ctx->tune.get("D2G_SP_RIDE")) v.ride = std::atoi(e) & 63;
    if (const char *e = ctx->tune.get("D2G_SP_LONG_LIST")) { const long long d = std::atoll(e); if (d >= 0) v.long_list = (size_t)d; }
    if (const char *e = ctx->tune.get("D2G_SP_PREDICT")) v.predict = std::atoi(e) != 0;
    if (const char *e = ctx->tune.get("D2G_SP_LIST_FORM")) { const int d = std::atoi(e); if (d >= 0 && d <= 2) v.list_form = d; }
    if (const char *e = ctx->tune.get("D2G_SP_LIST_DIV")) { const long d = std::atol(e); if (d >= 1 && d <= (1 << 20)) v.list_div = (size_t)d; }
    return v;
}

bool sparse_enabled(const d2g_ctx *ctx, size_t N) { const SpTuning t = sp_tuning(ctx); return t.sparse && N >= 2 && N >= t.min_n; }

size_t sp_list_cap(const d2g_ctx *ctx, size_t N) {
    const size_t pairs = N * (N - 1) / 2;
    return std::max<size_t>(std::min<size_t>(pairs / sp_tuning(ctx).list_div, (size_t)1 << 27), 1024);
}

int sp_alloc(d2g_ctx *ctx, d2g_cmp_set *set) {
    const size_t Npad = set->Npad, Nstride = set->Nstride;
    const size_t nrb = Npad / 32, ncb = Npad / BS_CB;
    set->tilebm_words = nrb * ((ncb + 31) / 32) + 1;
    set->tiles_cap = nrb * ((ncb + 7) / 8) * SP_SUBS;                        // per list: the sub-tiles of the tiles of every eighth column block
    set->plist_cap = sp_list_cap(ctx, set->N);
    // holders of mixed values: h holders make h - 1 pairs at least -- and a column has N holders at most; a record per mixed value: a pair at least each, N / 2 values per column at most
    set->cw_ecap = std::min<size_t>(std::min<size_t>(2 * set->plist_cap, set->ncols * set->N), 0xFFFFFFF0u);
    set->cw_vcap = std::min<size_t>(set->plist_cap, set->ncols * (set->N / 2 + 1));
    // output bins of the pair list: bands of 32 rows x chunks of 2^cshift columns (1024, wider while there would be more than SP_BIN_MAX bins)
    set->bin_cshift = 10;
    while (div_up<size_t>(set->N, 32) * div_up<size_t>(set->N, (size_t)1 << set->bin_cshift) > SP_BIN_MAX) ++set->bin_cshift;
    set->bin_nch = (uint32_t)div_up<size_t>(set->N, (size_t)1 << set->bin_cshift);
    set->nbins = (uint32_t)(div_up<size_t>(set->N, 32) * set->bin_nch);
    // one zero-initialised block per prepare: [counters Npad + 1 | 8 global control words + tile bitmap | order 8 | list control 8 | control words of a whole-triangle launch 16 | entries per bin | bin cursors]
    set->spz_words = (Npad + 1) + (8 + set->tilebm_words) + 8 + 12 + SP_CTL_WORDS + (size_t)set->nbins;
    // the workgroups that count (and then move) the list's entries: one per ~16 384 entries of a full list, at most two per CU
    set->bin_nwg = (uint32_t)std::max<size_t>(1, std::min<size_t>((size_t)std::max(ctx->num_cus / 2, 1), div_up<size_t>(set->plist_cap, 16384)));   // (128 of them measured best at config 3: 64 / 128 / 512 / 1024 -> counting + moving 90 / 65 / 70 / 84 us)
    const size_t planes_words = (size_t)set->ntb * set->nbits_cap + 1;
    hipError_t e;
    if ((e = hipMalloc((void **)&set->d_stream_s, planes_words * 2 * Nstride * sizeof(uint32_t))) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_rowstream, planes_words * Nstride * sizeof(uint32_t))) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_sperm, Nstride * 4)) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_sinv, Npad * 4)) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_label, 2 * Npad * 4)) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_hint, 2 * Npad * 4)) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_segend, Npad * 4)) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_posseg, Npad * 8)) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_spz, set->spz_words * 4)) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_rowpos, Nstride * 4)) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_rowk, Npad * 4)) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_spctl, (2 * SP_CTL_WORDS + set->tilebm_words) * 4)) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_tiles, 8 * std::max<size_t>(set->tiles_cap, 1) * 4)) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_tiles_full, 8 * std::max<size_t>(set->tiles_cap, 1) * 4)) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_plist, set->plist_cap * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_plist2, set->plist_cap * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_bstart, ((size_t)set->nbins + 1) * 4)) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_samp, (SP_SAMPLE_ROWS * Npad + 8) * 4)) != hipSuccess ||
        (e = hipMemset(set->d_samp, 0, (SP_SAMPLE_ROWS * Npad + 8) * 4)) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_cw_ents, set->cw_ecap * 4)) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_cw_vals, set->cw_vcap * 16)) != hipSuccess ||
        (e = hipMalloc((void **)&set->d_hoff, (size_t)set->bin_nwg * set->nbins * 4)) != hipSuccess) {
        ctx->last_error = std::string("bitslice sparse alloc: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? D2G_ERR_NOMEM : D2G_ERR_HIP;
    }
    set->d_lcnt = set->d_spz;
    set->d_gbm = set->d_lcnt + (Npad + 1);
    set->d_order = set->d_gbm + 8 + set->tilebm_words;
    set->d_plctl = set->d_order + 8;
    set->d_fullctl = set->d_plctl + 12;
    set->d_binc = set->d_fullctl + SP_CTL_WORDS;
    set->d_tilebm = set->d_spctl + 2 * SP_CTL_WORDS;
    if ((e = hipMemset(set->d_spctl, 0, 2 * SP_CTL_WORDS * 4)) != hipSuccess) { ctx->last_error = std::string("bitslice sparse alloc: ") + hipGetErrorString(e); return D2G_ERR_HIP; }
    // one word of host memory the device can write: the remembered give-up (sp_prepare_order)
    if (hipHostMalloc((void **)&set->h_gaveup, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer((void **)&set->d_gaveup, set->h_gaveup, 0) == hipSuccess) std::memset(set->h_gaveup, 0, 64);   // (all 16 words: the first look's ticket word must not hold what an earlier owner of the page left)
    else { (void)hipGetLastError(); if (set->h_gaveup) (void)hipHostFree(set->h_gaveup); set->h_gaveup = nullptr; set->d_gaveup = set->d_order + 7; }   // (no mapped host memory: a spare device word, never read by the host)
    set->sp_launch = 0;
    return D2G_OK;
}

void sp_free(d2g_cmp_set *set) {
    for (uint32_t **p : {&set->d_stream_s, &set->d_sperm, &set->d_sinv, &set->d_label, &set->d_hint, &set->d_segend, &set->d_posseg, &set->d_spz, &set->d_rowpos, &set->d_rowk,
                         &set->d_rowstream, &set->d_tiles, &set->d_tiles_full, &set->d_spctl}) { (void)hipFree(*p); *p = nullptr; }
    (void)hipFree(set->d_plist); set->d_plist = nullptr;
    (void)hipFree(set->d_plist2); set->d_plist2 = nullptr;
    (void)hipFree(set->d_bstart); set->d_bstart = nullptr;
    (void)hipFree(set->d_hoff); set->d_hoff = nullptr;
    (void)hipFree(set->d_samp); set->d_samp = nullptr;
    if (set->fill_stream) { (void)hipStreamDestroy((hipStream_t)set->fill_stream); set->fill_stream = nullptr; }
    if (set->fill_fork) { (void)hipEventDestroy((hipEvent_t)set->fill_fork); set->fill_fork = nullptr; }
    if (set->fill_join) { (void)hipEventDestroy((hipEvent_t)set->fill_join); set->fill_join = nullptr; }
    if (set->samp_stream) { (void)hipStreamDestroy((hipStream_t)set->samp_stream); set->samp_stream = nullptr; }
    if (set->samp_event) { (void)hipEventDestroy((hipEvent_t)set->samp_event); set->samp_event = nullptr; }
    (void)hipFree(set->d_cw_ents); set->d_cw_ents = nullptr;
    (void)hipFree(set->d_cw_vals); set->d_cw_vals = nullptr;
    set->d_binc = nullptr;
    if (set->h_gaveup) { (void)hipHostFree(set->h_gaveup); set->h_gaveup = nullptr; }
    set->d_gaveup = nullptr;
    set->d_tilebm = set->d_lcnt = set->d_gbm = set->d_order = set->d_plctl = set->d_fullctl = nullptr;
}

// what the kernel in front of sp_prepare_order initialises for it: label[j] = j, the hints, and the zero block (counters,
// tile bitmap + global control words, order words, list cursor)
SpInit sp_init_of(const d2g_cmp_set *set) {
    SpInit si;
    si.label = set->d_label; si.n = (uint32_t)set->N;
    si.ones = set->d_hint; si.owords = (uint32_t)(2 * set->Npad);
    si.zero = set->d_spz; si.zwords = (uint32_t)set->spz_words;
    return si;
}

// candidate tiles of a whole-triangle launch: the tiles on or above the diagonal of sorted positions
size_t sp_full_candidates(size_t Npad) {
    const size_t nrb = Npad / 32, ncb = Npad / BS_CB;
    size_t cand = 0;
    for (size_t cb = 0; cb < ncb; ++cb) cand += std::min<size_t>(nrb, (cb * 256 + 255) / 32 + 1);
    return cand;
}

// the whole ordering skipped: dense walk (see sp_prepare_order)
__global__ void sp_giveup_kernel(uint32_t *__restrict__ order, uint32_t *__restrict__ fullctl, uint32_t cand) {
    if (threadIdx.x == 0) { order[0] = 1; fullctl[1] = 1; fullctl[3] = cand; }
}

// ---- the first look at a matrix (VERDICT r5 #1b): which path pays is decided BEFORE the ordering, from a sample.  Sixteen sketches spread over the
// collection are compared with every sketch, register by register, on the ids the rank kernel left (an equal id in a column = an equal register):
//   a pair that shares fewer than SP_SAMPLE_FAM registers   chance / conserved k-mers: its shared registers are list entries   -> E
//   a pair that shares more                                  family: a tile pair                                                -> F
// scaled by N / 32 (a pair is seen from either end) these are the list length and the family pairs the ordering WOULD find.  A list beyond its
// buffer, or families that cover a third of the triangle, mean the dense walk -- known after two small kernels (a few microseconds) and ONE
// synchronisation, on the set's first prepare (and whenever a remembered give-up is due for its retry) instead of after link / sort / emit
// (0.05-0.3 ms).  Later prepares of the set go by what is remembered.  A heuristic: whatever it says, the results are exact.
struct SpSampleRows { uint32_t r[SP_SAMPLE_ROWS]; };
__global__ __launch_bounds__(256) void sp_sample_kernel(const uint32_t *__restrict__ ids, size_t N, size_t Npad, uint32_t ncols, SpSampleRows rows, uint32_t *__restrict__ cntm) {
    __shared__ __attribute__((aligned(16))) uint32_t sm[SP_SAMPLE_COLS][SP_SAMPLE_ROWS];   // the sampled sketches' ids in this workgroup's columns
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t t0 = blockIdx.y * SP_SAMPLE_COLS, t1 = min(ncols, t0 + SP_SAMPLE_COLS);
    for (uint32_t x = threadIdx.x; x < SP_SAMPLE_COLS * SP_SAMPLE_ROWS; x += 256) {
        const uint32_t t = t0 + x / SP_SAMPLE_ROWS, k = x % SP_SAMPLE_ROWS;
        const uint32_t w = t < t1 ? ids[(size_t)t * Npad + rows.r[k]] : 0u;
        sm[x / SP_SAMPLE_ROWS][k] = (w != 0 && !(w >> 31)) ? w : 0xFFFFFFFFu;      // (a value nobody else holds matches nothing: never equal to a shared id)
    }
    __syncthreads();
    uint32_t acc[SP_SAMPLE_ROWS];
#pragma unroll
    for (uint32_t k = 0; k < SP_SAMPLE_ROWS; ++k) acc[k] = 0;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll 4
    for (uint32_t x = 0; x < SP_SAMPLE_COLS; ++x) {
        const uint32_t wx = (j < N && t0 + x < t1) ? ids[(size_t)(t0 + x) * Npad + j] : 0u;      // (0 or bit 31: equal to no staged id)
#pragma unroll
        for (uint32_t q = 0; q < SP_SAMPLE_ROWS / 4; ++q) {
            const u32x4 s4 = reinterpret_cast<const u32x4 *>(&sm[x][0])[q];
            acc[4 * q] += wx == s4.x; acc[4 * q + 1] += wx == s4.y; acc[4 * q + 2] += wx == s4.z; acc[4 * q + 3] += wx == s4.w;
        }
    }
    // four sampled sketches per word, a byte each, a column group's share capped at SP_SAMPLE_FAM (what counts is "fewer than that in all, or not":
    // 32 groups x 4 stay below 256) -- a noisy matrix has a count for every (sample, sketch, group): a quarter of the atomics
#pragma unroll
    for (uint32_t q = 0; q < SP_SAMPLE_ROWS / 4; ++q) {
        const uint32_t w = min(acc[4 * q], SP_SAMPLE_FAM) | (min(acc[4 * q + 1], SP_SAMPLE_FAM) << 8) | (min(acc[4 * q + 2], SP_SAMPLE_FAM) << 16) | (min(acc[4 * q + 3], SP_SAMPLE_FAM) << 24);
        if (w) atomicAdd(&cntm[(size_t)q * Npad + j], w);
    }
}
// the counts -> (E, F) in mapped host memory; the kernel cleans up behind itself (counters, its sums, the ticket)
__global__ __launch_bounds__(256) void sp_sample_fin_kernel(uint32_t *__restrict__ cntm, size_t N, size_t Npad, SpSampleRows rows, uint32_t *__restrict__ acc3, uint32_t *__restrict__ host_out,
                                                            const uint32_t *__restrict__ colcnt, uint32_t ncols, int nsplit, uint32_t ticket) {
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t e = 0, f = 0;
    // (the same grid also sums what the rank kernel left: the shared values of every column, and the id planes a column of that many needs -- the
    // column plan groups columns of one plane class, so the mean over columns is the mean over groups)
    uint32_t v = 0, pl = 0;
    if (j < ncols) {
        v = colcnt[j * BS_CC_STRIDE + 5];                             // (slot 5: the rank kernel's own total, untouched by the column plan)
        pl = v == 0 ? 1u : 32u - __clz(v + 1u);
    }
    if (j < N) {
#pragma unroll
        for (uint32_t q = 0; q < SP_SAMPLE_ROWS / 4; ++q) {
            const uint32_t w = cntm[(size_t)q * Npad + j];
            if (!w) continue;
            cntm[(size_t)q * Npad + j] = 0;
#pragma unroll
            for (uint32_t y = 0; y < 4; ++y) {
                const uint32_t c = (w >> (8 * y)) & 0xFFu;
                if (c && j != rows.r[4 * q + y]) { if (c >= SP_SAMPLE_FAM) ++f; else e += c; }
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) { e += __shfl_down(e, o); f += __shfl_down(f, o); v += __shfl_down(v, o); pl += __shfl_down(pl, o); }
    if ((threadIdx.x & 63) == 0) { if (e) atomicAdd(&acc3[0], e); if (f) atomicAdd(&acc3[1], f); if (v) atomicAdd(&acc3[3], v); if (pl) atomicAdd(&acc3[4], pl); }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&acc3[2], 1u) + 1u == gridDim.x) {              // the last workgroup: everybody's sums are in (device-scope atomics)
            host_out[0] = atomicExch(&acc3[0], 0u);
            host_out[1] = atomicExch(&acc3[1], 0u);
            host_out[2] = atomicExch(&acc3[3], 0u);
            host_out[3] = atomicExch(&acc3[4], 0u);
            acc3[2] = 0;
            // the host waits for THIS word (sp_sample_collect polls it: the kernels enqueued behind this one keep the device busy meanwhile)
            __threadfence_system();
            __hip_atomic_store(&host_out[4], ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---- riders (sp_ride): the share of an announced output's fill that one hosting kernel of the prepare chain carries.  The weights are the
// hosts' own durations at config 3 (us): each hides about what it lasts; the last host (place) takes what is left.
constexpr unsigned SP_RW_PLAN = 6, SP_RW_FLATTEN = 5, SP_RW_COUNT = 5, SP_RW_ATTACH = 5, SP_RW_SCAN = 8, SP_RW_ALL = 35;
inline size_t sp_fill_pieces(size_t cnt) { return div_up<size_t>(cnt / 4 + 1, (size_t)SP_FILL_THREADS * SP_FILL_PER_THREAD); }
SpRider sp_take_rider(d2g_cmp_set *set, unsigned own, unsigned weight, bool last, unsigned *grid, int bit) {
    *grid = own;
    if (set->ride_next >= set->ride_total || !(set->ride_mask & bit)) return SP_NO_RIDER;
    const uint32_t left = set->ride_total - set->ride_next;
    uint32_t n = last ? left : (uint32_t)std::min<uint64_t>(left, ((uint64_t)set->ride_total * weight + SP_RW_ALL - 1) / SP_RW_ALL);
    n = std::min<uint32_t>(n, 0x7FFFFFFFu - own);
    const SpRider r{set->ride_out, set->ride_cnt, set->ride_vsrc, set->ride_vimm, own, set->ride_next};
    set->ride_next += n;
    *grid = own + n;
    return r;
}
SpColWork sp_colwork_of(const d2g_cmp_set *set) {
    return SpColWork{set->d_cw_ents, reinterpret_cast<uint4 *>(set->d_cw_vals), (uint32_t)std::min<size_t>(set->cw_ecap, 0xFFFFFFFFu), (uint32_t)std::min<size_t>(set->cw_vcap, 0xFFFFFFFFu)};
}
SpPairs sp_pairs_of(const d2g_cmp_set *set, const SpColWork &cw, const uint32_t *seg) {
    return SpPairs{cw, seg, set->d_plctl, set->d_plist, (uint32_t)std::min<size_t>(set->plist_cap, 0xFFFFFFFFu), set->d_order, set->d_gaveup, set->d_fullctl};
}
// long list or short list?  What the set's LAST prepare left (its length, written to mapped host memory by the permute launch) decides which form
// THIS prepare enqueues: the binned one (pairs kernel, counting workgroups, sp_bin_kernel; the launches compose) from `long_list` entries on, the
// entry-by-entry one (nothing enqueued for it) below.  A set without history takes the binned form.  Read without synchronisation, like the
// remembered give-up: a prepare still in flight has not written yet and the one before it decides.  Both forms are exact for any list.
bool sp_expect_long_list(const d2g_ctx *ctx, const d2g_cmp_set *set) {
    const SpTuning t = sp_tuning(ctx);
    if (t.list_form == 1) return false;
    if (t.list_form == 2) return true;
    if (set->pred_valid) return set->pred_entries >= (double)t.long_list;
    if (!set->h_gaveup || set->sp_prepares == 0) return true;
    const uint32_t n = ((volatile uint32_t *)set->h_gaveup)[1];
    return n != 0xFFFFFFFFu && n >= t.long_list;
}
// will the next sp_prepare_order skip the ordering (the remembered give-up)?  Asked BEFORE it, by the prepare that decides whether anything rides
bool sp_retry_due(const d2g_cmp_set *set) { return ((set->sp_prepares + 1) & 15u) == 0; }
bool sp_will_skip(const d2g_ctx *ctx, const d2g_cmp_set *set) {
    if (set->pred_valid) return set->pred_dense;                       // this prepare has looked at its matrix (sp_sample)
    return sp_tuning(ctx).remember && set->h_gaveup && *(volatile uint32_t *)set->h_gaveup && !sp_retry_due(set);
}
// does this prepare look at its matrix first?  The set's first prepare, and the retry of a remembered give-up
bool sp_sample_due(const d2g_ctx *ctx, const d2g_cmp_set *set) {
    const SpTuning t = sp_tuning(ctx);
    if (!t.predict || !set->d_samp || set->borrowed || set->nsplit > 1) return false;    // (a column several rank workgroups share has no single count of its shared values)
    if (set->sp_prepares == 0) return true;
    return t.remember && set->h_gaveup && *(volatile uint32_t *)set->h_gaveup && sp_retry_due(set);
}

// the sample (see sp_sample_kernel): SYNCHRONISES `s`.  Leaves the set's prediction (pred_valid, pred_dense, pred_entries) and the remembered word.
// Two halves: sp_sample_enqueue (two small kernels: 18 + 8 us at config 3) and sp_sample_collect (the synchronisation).  (The kernels on a second
// stream beside the column plan and the planes kernel were measured: no gain -- what the first look costs, ~50 us at config 3, is the
// synchronisation itself and the launches behind it, which no longer run ahead of the device.)
int sp_sample_enqueue(d2g_ctx *ctx, d2g_cmp_set *set, hipStream_t s) {
    const size_t N = set->N, Npad = set->Npad;
    if (!set->h_gaveup) return D2G_OK;                                   // (no mapped host memory: no sample)
    { hipStreamCaptureStatus cs = hipStreamCaptureStatusNone; if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return D2G_OK; } }   // (a captured prepare cannot wait for the host)
    SpSampleRows rows;
    for (uint32_t k = 0; k < SP_SAMPLE_ROWS; ++k) rows.r[k] = (uint32_t)std::min<size_t>(N - 1, (size_t)(2 * k + 1) * N / (2 * SP_SAMPLE_ROWS));
    uint32_t *acc3 = set->d_samp + SP_SAMPLE_ROWS * Npad;                // (the counters and the control words are zero: cleared at allocation, then by the kernel itself)
#ifndef D2G_SP_SAMPLE_INLINE
    // on a stream of their own, behind the rank kernel: the column plan (ONE workgroup) and the planes kernel run beside them.  No join: the host
    // waits for the sample's word before it enqueues anything else, and nothing the two kernels read is written before the set's next prepare.
    if (!set->samp_stream) {
        hipStream_t st = nullptr; hipEvent_t ev = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) { set->samp_stream = st; set->samp_event = ev; }
        else { (void)hipGetLastError(); if (st) (void)hipStreamDestroy(st); }
    }
    if (set->samp_stream) {
        D2G_HIP(ctx, hipEventRecord((hipEvent_t)set->samp_event, s));
        D2G_HIP(ctx, hipStreamWaitEvent((hipStream_t)set->samp_stream, (hipEvent_t)set->samp_event, 0));
        s = (hipStream_t)set->samp_stream;
    }
#endif
    hipLaunchKernelGGL(sp_sample_kernel, dim3((unsigned)div_up<size_t>(N, 256), (unsigned)div_up<size_t>(set->ncols, SP_SAMPLE_COLS)), dim3(256), 0, s, set->d_ids, N, Npad, (uint32_t)set->ncols, rows, set->d_samp);
    hipLaunchKernelGGL(sp_sample_fin_kernel, dim3((unsigned)div_up<size_t>(std::max(N, set->ncols), 256)), dim3(256), 0, s, set->d_samp, N, Npad, rows, acc3, set->d_gaveup + 2,
                       set->d_colcnt, (uint32_t)set->ncols, set->nsplit, ++set->sample_ticket);
    D2G_HIP(ctx, hipGetLastError());
    set->sample_pending = true;
    return D2G_OK;
}
// SYNCHRONISES `s`.  Leaves the set's prediction (pred_valid, pred_dense, pred_entries) and the remembered word.
int sp_sample_collect(d2g_ctx *ctx, d2g_cmp_set *set, hipStream_t s) {
    if (!set->sample_pending) return D2G_OK;
    set->sample_pending = false;
    const size_t N = set->N;
    volatile uint32_t *h = (volatile uint32_t *)set->h_gaveup;
    // The two kernels stand right behind the rank kernel; the column plan and the planes kernel are enqueued behind them and run while the host
    // waits for the sample's word and enqueues what it decides -- the device does not idle over the decision.  Polling the mapped word costs a
    // few microseconds; hipStreamSynchronize would also wait for the kernels behind (and took ~50 us to return and refill the queue at
    // config 3).  The wait is bounded: a device that takes longer than a quarter of a second gets the synchronisation.
    {
        const auto t0 = std::chrono::steady_clock::now();
        uint32_t spins = 0;
        while (__atomic_load_n(&set->h_gaveup[6], __ATOMIC_ACQUIRE) != set->sample_ticket) {
            __builtin_ia32_pause();
            if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(250)) { D2G_HIP(ctx, hipStreamSynchronize(set->samp_stream ? (hipStream_t)set->samp_stream : s)); break; }
        }
    }
    const double scale = (double)N / (2.0 * SP_SAMPLE_ROWS), pairs = (double)N * (double)(N - 1) / 2.0;
    set->pred_valid = true;
    set->pred_entries = (double)h[2] * scale;
    set->pred_family_pairs = (double)h[3] * scale;
    const double values = (double)h[4], planes = set->ncols ? (double)h[5] / (double)set->ncols : 1.0;
    // what each path would take BEYOND the prepare both share, in nanoseconds (constants measured at config 3 on MI355X, round 6: profiles/r06_k2_experiments.txt):
    // the dense walk costs 2 + 0.94 x planes ps per pair (426 us at 7 planes); the sparse path a chain of ordering kernels and the fill (66 us on a matrix
    // that shares nothing: 6.6 ns per sketch), 55 ps per list entry (pairs, counting, moving, composing), 77 ps per shared value (grouping its holders, its
    // record) and 9 x the dense rate per family pair (their tiles in the latency-bound sparse pair kernel, their holders in the link and emit passes:
    // ~100 us for 7.4e5 family pairs).  (the per-plane and per-sketch terms are those of 32 register groups, S = 1024: they go with the group count)
    const double g = (double)set->ntb / 32.0, per_pair = 0.002 + 0.00094 * planes * g;
    const double dense_ns = pairs * per_pair, sparse_ns = (3.0 + 3.6 * g) * (double)N + 0.055 * set->pred_entries + 0.077 * values + 9.0 * per_pair * set->pred_family_pairs;
    set->pred_dense = set->pred_entries > (double)set->plist_cap || sparse_ns > 0.97 * dense_ns;
    set->h_gaveup[0] = set->pred_dense ? 1u : 0u;                      // what the next prepares go by (the device kernels that give up write the same word)
    return D2G_OK;
}

// labels -> counting sort -> d_sperm / d_sinv -> pair list + segment tiles.  All on `s`, no host round trip.
// where the path does not pay (one family, heavy noise, adversarial columns) the ordering that finds it out costs 0.05-0.2 ms
// per prepare.  The decision is REMEMBERED per set: the kernels that raise order[0] also write a word in host-visible memory; a later
// prepare of the same set (CLI batches, re-loaded matrices) reads it -- no synchronisation: a prepare still in flight simply has not
// written it yet -- and skips the ordering; every 16th prepare tries again (the matrix may have changed).  Exactness is not involved: the
// dense walk is always right.
int sp_prepare_order(d2g_ctx *ctx, d2g_cmp_set *set, bool split, hipStream_t s) {
    const size_t N = set->N, Npad = set->Npad, S = set->ncols;
    // (a prepare that has handed out riders on a REMEMBERED give-up goes through: its place kernel carries the rest of the fill; one that has just
    // looked at its matrix and found it dense skips -- the dense launch writes every output itself)
    const bool remembered = set->skip_cached >= 0 ? set->skip_cached == 1 : sp_will_skip(ctx, set);      // (d2g_bitslice_prepare read the word already)
    set->skip_cached = -1;
    const bool skip = set->pred_valid ? set->pred_dense : (remembered && set->ride_total == 0);
    set->sp_big = sp_expect_long_list(ctx, set);
    ++set->sp_prepares;
    set->pred_valid = false;                                           // (a prediction serves the prepare that made it)
    if (skip) {
        hipLaunchKernelGGL(sp_giveup_kernel, dim3(1), dim3(64), 0, s, set->d_order, set->d_fullctl, (uint32_t)std::min<size_t>(sp_full_candidates(Npad), 0xFFFFFFFFu));
        D2G_HIP(ctx, hipGetLastError());
        set->sp_skipped = true;
        return D2G_OK;
    }
    set->sp_skipped = false;
    const unsigned nb = (unsigned)div_up<size_t>(N, 256);
    uint32_t *la = set->d_label, *lb = set->d_label + Npad;
    const SpTuning tu = sp_tuning(ctx);
    if (tu.link && S >= 2) {
        const uint32_t cap = (uint32_t)std::min<size_t>(N / 2 + 1, 12288);              // shared values of a column that take part: 3 words each, 144 KB of LDS at most
        D2G_HIP(ctx, hipFuncSetAttribute((const void *)sp_link_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 12288 * 12));
        D2G_HIP(ctx, hipFuncSetAttribute((const void *)sp_link_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 12288 * 8));
        const unsigned npair = (unsigned)(S / 2);
        const uint32_t ustride = (uint32_t)std::max<size_t>(1, std::min<size_t>(SP_UNITE_STRIDE, npair / 32));   // at least 32 column pairs take part in the uniting pass
        if (set->d_owner && !split && tu.olink) {                        // one holder per shared value at hand: the streaming form
            const unsigned nx = (unsigned)div_up<size_t>(N, 1024);
            hipLaunchKernelGGL(sp_olink_kernel<0>, dim3(nx, npair), dim3(256), 0, s, set->d_ids, N, Npad, (uint32_t)S, set->d_owner, set->owner_stride, 1u, la, set->d_hint);
            { unsigned g; const SpRider rd = sp_take_rider(set, nb, SP_RW_FLATTEN, false, &g, 2); hipLaunchKernelGGL(sp_flatten_kernel, dim3(g), dim3(256), 0, s, la, N, rd); }
            hipLaunchKernelGGL(sp_olink_kernel<1>, dim3(nx, div_up<unsigned>(npair, ustride)), dim3(256), 0, s, set->d_ids, N, Npad, (uint32_t)S, set->d_owner, set->owner_stride, ustride, la, set->d_hint);
        } else {
            hipLaunchKernelGGL(sp_link_kernel<0>, dim3(npair), dim3(1024), (size_t)cap * 12, s, set->d_ids, N, Npad, (uint32_t)S, set->d_colcnt, split ? 1 : 0, cap, 1u,
                               la, set->d_hint);
            { unsigned g; const SpRider rd = sp_take_rider(set, nb, SP_RW_FLATTEN, false, &g, 2); hipLaunchKernelGGL(sp_flatten_kernel, dim3(g), dim3(256), 0, s, la, N, rd); }
            hipLaunchKernelGGL(sp_link_kernel<1>, dim3(div_up<unsigned>(npair, ustride)), dim3(1024), (size_t)cap * 8, s, set->d_ids, N, Npad, (uint32_t)S, set->d_colcnt, split ? 1 : 0, cap, ustride,
                               la, set->d_hint);
        }
    }
    const size_t ntile_all = (Npad / 32) * (Npad / BS_CB);
    const uint32_t seg_limit = (uint32_t)std::min<size_t>((size_t)((double)ntile_all * tu.tile_frac), 0x3FFFFFFF);
    // (one single-workgroup kernel for count + scan + place with the counters in LDS was measured at N = 10 000: 25 us against 19 for the three)
    { unsigned g; const SpRider rd = sp_take_rider(set, nb, SP_RW_COUNT, false, &g, 4); hipLaunchKernelGGL(sp_count_kernel, dim3(g), dim3(256), 0, s, la, lb, N, set->d_lcnt, set->d_order, rd); }
    if (tu.link && S >= 2) {
        unsigned g; const SpRider rd = sp_take_rider(set, nb, SP_RW_ATTACH, false, &g, 8);
        hipLaunchKernelGGL(sp_attach_kernel, dim3(g), dim3(256), 0, s, lb, set->d_lcnt, set->d_hint, N, Npad, set->d_order, rd);
    }
    const uint32_t CW = (uint32_t)((Npad / BS_CB + 31) / 32);
    { unsigned g; const SpRider rd = sp_take_rider(set, 1, SP_RW_SCAN, false, &g, 16);
      hipLaunchKernelGGL(sp_scan_kernel, dim3(g), dim3(1024), 0, s, set->d_lcnt, N, set->d_order, la, set->d_segend, seg_limit, set->d_gaveup, rd); }    // la (labels) is dead after the count kernel: it keeps the segment starts
    { unsigned g; const SpRider rd = sp_take_rider(set, (unsigned)div_up<size_t>(set->Nstride, 256), 0, true, &g, 32);
      hipLaunchKernelGGL(sp_place_kernel, dim3(g), dim3(256), 0, s, lb, N, set->Nstride, set->d_lcnt, set->d_sperm, set->d_sinv, set->d_order,
                         la, set->d_segend, CW, set->d_gbm + 8, reinterpret_cast<uint2 *>(set->d_posseg), rd); }
    // (a certificate pass in front -- one thread per (column, sketch) comparing the sketch's segment with that of its value's owner, so that
    // columns where nothing crosses a segment need no workgroup here -- was measured: 17 us for the pass, and the 17 stragglers a clean
    // collection of 10 000 leaves still put a mixed value into a hundred columns, whose workgroups take as long as before: 0.338 vs 0.329 ms)
    const SpColWork cw = sp_colwork_of(set);
    hipLaunchKernelGGL(sp_emit_kernel, dim3((unsigned)S), dim3(SP_EMIT_T), 0, s, set->d_ids, N, Npad, (uint32_t)S, set->d_colcnt, split ? 1 : 0, lb, set->d_order,
                       cw, set->d_plctl, (uint32_t)std::min<size_t>(set->plist_cap, 0xFFFFFFFFu), set->d_gaveup, (N >= 65536 || tu.emit_big) ? 1 : 0);
    // the pairs of the mixed values: a kernel of its own when the list is expected to be long (it must be complete before the workgroups that
    // count it per bin, which ride on the permute launch); otherwise extra workgroups of the permute launch itself (sp_permute)
    if (set->sp_big) hipLaunchKernelGGL(sp_pairs_kernel, dim3((unsigned)ctx->num_cus * 8), dim3(256), 0, s, sp_pairs_of(set, cw, lb));
    D2G_HIP(ctx, hipGetLastError());
    return D2G_OK;
}

// the sorted stream + (in the same launch) the work lists of whole-triangle launches
SpBins sp_bins_of(const d2g_cmp_set *set) { return SpBins{set->d_binc, set->d_bstart, set->nbins, set->bin_nch, set->bin_cshift}; }

int sp_permute(d2g_ctx *ctx, d2g_cmp_set *set, hipStream_t s) {
    if (set->sp_skipped) { set->full_list_valid = true; return D2G_OK; }   // the remembered give-up: sp_giveup_kernel left the control words of a dense launch
    // (the sorted stream gathered straight from the ids -- one group per XCD so that the gathers hit its L2 -- instead of permuting the caller's-order
    // stream was measured again in round 5: 42 us against planes 19 + permute 21 at config 3, 208 against 113 at N = 50 000; round 4 without the XCD
    // mapping: 74)
    const size_t nrb = set->Npad / 32, ncb = set->Npad / BS_CB, ntile = nrb * ncb;
    const bool both = set->Nstride * 8 <= 156 * 1024;               // both codings of a plane in the LDS of one workgroup
    constexpr size_t LDS_PART = 128 * 1024;                         // otherwise: one coding, in parts of at most this
    const uint32_t H = both ? 1u : (uint32_t)div_up<size_t>(set->Nstride * 4, LDS_PART);
    const uint32_t part = (uint32_t)(div_up<size_t>(div_up<size_t>(set->Nstride, H), 4) * 4);
    SpFullList fl{set->d_gbm + 8, (uint32_t)nrb, (uint32_t)ncb, (uint32_t)((ncb + 31) / 32), set->d_tiles_full, (uint32_t)set->tiles_cap, set->d_fullctl,
                  (uint32_t)std::min<size_t>(sp_full_candidates(set->Npad), 0xFFFFFFFFu), (uint32_t)div_up<size_t>(ntile, 1024), (uint32_t)set->N,
                  reinterpret_cast<const uint2 *>(set->d_posseg)};
    const size_t nperm = (size_t)set->ntb * set->nbits_cap * (both ? 1 : 2) * H;
    const uint32_t plcap = (uint32_t)std::min<size_t>(set->plist_cap, 0xFFFFFFFFu);
    // behind the list builders: the workgroups that count a LONG list per output bin, or those that make the pairs of a short one
    SpHist hs{set->d_plist, set->d_plctl, plcap, set->sp_big ? set->bin_nwg : (uint32_t)ctx->num_cus, sp_bins_of(set), set->d_hoff, set->sp_big ? 1 : 0,
              set->h_gaveup ? set->d_gaveup + 1 : nullptr};
    set->full_list_valid = nperm + fl.nwg + hs.nhw < 0x7FFFFFFFu;
    if (!set->full_list_valid) { ctx->last_error = "bitslice sparse: the permute launch does not fit a grid"; return D2G_ERR_INTERNAL; }
    const size_t lds = std::max<size_t>((size_t)part * (both ? 8 : 4), set->sp_big ? (size_t)set->nbins * 4 : 0);
    auto kern = both ? sp_permute_lds_kernel<true> : sp_permute_lds_kernel<false>;
    D2G_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 157 * 1024));   // (+ ~2 KB of static LDS: the list builders, the pairs of a short list)
    hipLaunchKernelGGL(kern, dim3((unsigned)(nperm + fl.nwg + hs.nhw)), dim3(1024), lds, s, set->d_stream, set->d_stream_s, set->Nstride, set->d_meta,
                       set->d_sperm, set->d_order, fl, (uint32_t)nperm, (uint32_t)set->nbits_cap, H, part, hs, sp_pairs_of(set, sp_colwork_of(set), set->d_label + set->Npad));
    if (set->sp_big) {                                                  // the list, bin by bin (d_plist2): as many workgroups as the counting ones
        D2G_HIP(ctx, hipFuncSetAttribute((const void *)sp_bin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SP_BIN_MAX * 4));
        hipLaunchKernelGGL(sp_bin_kernel, dim3(hs.nhw), dim3(1024), (size_t)set->nbins * 4, s, set->d_plist, set->d_plist2, set->d_plctl, plcap, hs.bn, set->d_hoff, set->d_order);
    }
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

// what a store fills with: the table whose first entry is the value of "no register equal", or nullptr for the count 0
inline const uint32_t *sp_fill_source(const StoreEq &) { return nullptr; }
inline const uint32_t *sp_fill_source(const StoreLut &st) { return reinterpret_cast<const uint32_t *>(st.lut); }

// the fill of rows [r0, r1) of the triangle, enqueued NOW (unconditionally: should the launch turn out dense, the pair kernel overwrites it);
// the next sparse launch on the set that writes to the same output skips its own fill
template <class Store>
int sp_prefill(d2g_ctx *ctx, d2g_cmp_set *set, size_t r0, size_t r1, Store store, uint32_t *out_words, hipStream_t s) {
    const size_t cnt = d2g_ut_count(set->N, r0, r1);
    if (!cnt || !set->sparse_ok) return D2G_OK;
    hipLaunchKernelGGL((sp_fill_kernel<Store>), dim3((unsigned)std::min<size_t>(div_up<size_t>(cnt / 4 + 1, SP_FILL_THREADS * SP_FILL_PER_THREAD), (size_t)0x7FFFFFFF)), dim3(SP_FILL_THREADS), 0, s,
                       out_words, cnt, store, (uint32_t)set->S, (const uint32_t *)nullptr, 0u, 0u);
    D2G_HIP(ctx, hipGetLastError());
    set->prefilled = out_words; set->prefilled_cnt = cnt; set->prefilled_pieces = (size_t)-1;
    set->prefilled_src = sp_fill_source(store); set->prefilled_by_riders = false;
    return D2G_OK;
}

// One sparse launch on a set whose last prepare left a sorted operand.  The launch uses per-set scratch (work list, launch rows, control
// words): launches on ONE set must be issued one after the other on ONE stream (include/d2g.h says so).
template <class Store>
int launch_sparse(d2g_ctx *ctx, const d2g_cmp_set *cset, PairShape sh, Store store, uint32_t *out_words, hipStream_t s) {
    d2g_cmp_set *set = const_cast<d2g_cmp_set *>(cset);
    const size_t N = set->N, Npad = set->Npad, r0 = sh.i_lo, r1 = sh.i_hi;
    if (r1 <= r0) return D2G_OK;
    const bool full = r0 == 0 && r1 == N;
    const size_t nrows = r1 - r0, nrows_pad = full ? Npad : div_up<size_t>(nrows, 32) * 32;
    const uint32_t nrb = (uint32_t)(nrows_pad / 32), ncb = (uint32_t)(Npad / BS_CB);
    const size_t cnt = d2g_ut_count(N, r0, r1);
    if (!cnt) return D2G_OK;
    PairShape dsh = sh;                                                                 // the dense walk of the same launch, behind the gate
    if (int rc = finish_shape(ctx, dsh, BS_JR == 2 ? 32u : 64u)) return rc;
    d2g_timer tm(ctx, &ctx->ev_k2, s);
    const uint32_t CW = (ncb + 31) / 32;                                                // words of a bitmap row (column blocks)
    // per launch: 16 control words (ctl[0] = tiles listed, ctl[1] = flags (bit 0: dense walk), [3] = candidates, [8..15] tiles per XCD list) + a partial launch's bitmap
    // double-buffered: this launch's list kernel clears the other set for the next launch (both start cleared: sp_alloc)
    const bool own_list = !(full && set->full_list_valid);            // a whole-triangle launch walks the lists the prepare left (sp_permute)
    uint32_t *const ctl = own_list ? set->d_spctl + SP_CTL_WORDS * (set->sp_launch & 1u) : set->d_fullctl;
    uint32_t *const ctl_next = set->d_spctl + SP_CTL_WORDS * ((set->sp_launch + 1) & 1u);
    if (own_list) ++set->sp_launch;
    set->last_ctl = ctl;
    if (!full) {
        hipLaunchKernelGGL(sp_rows_kernel, dim3(1), dim3(1024), 0, s, set->d_sperm, N, (uint32_t)r0, (uint32_t)r1, (uint32_t)nrows_pad, set->d_rowpos, set->d_rowk, set->d_order);
        hipLaunchKernelGGL(sp_gather_kernel, dim3((unsigned)div_up<size_t>(nrows_pad, 256), (unsigned)(set->ntb * set->nbits_cap)), dim3(256), 0, s,
                           set->d_stream_s, set->Nstride, set->d_meta, set->ntb, set->d_rowpos, (uint32_t)nrows_pad, set->d_rowstream, set->Nstride, set->d_order);
        hipLaunchKernelGGL(sp_rowbm_kernel, dim3((unsigned)div_up<size_t>((size_t)nrb * CW, 256)), dim3(256), 0, s, set->d_gbm + 8, set->d_rowpos, nrb, CW, set->d_tilebm, set->d_order);
    }
    const size_t ntile = (size_t)nrb * ncb;
    // candidates: every tile of a partial launch; the tiles on or above the diagonal of sorted positions of a full one
    const size_t cand = full ? sp_full_candidates(Npad) : ntile;
    const uint32_t cand32 = (uint32_t)std::min<size_t>(cand, 0xFFFFFFFFu);
    const uint32_t *bm = full ? set->d_gbm + 8 : set->d_tilebm;
    uint32_t *const tiles = own_list ? set->d_tiles : set->d_tiles_full;
    if (own_list)
        hipLaunchKernelGGL(sp_list_kernel, dim3((unsigned)div_up<size_t>(ntile, 1024)), dim3(1024), 0, s, bm, nrb, ncb, CW, full ? 1 : 0,
                           tiles, (uint32_t)set->tiles_cap, ctl, cand32, set->d_order, ctl_next, (uint32_t)N, full ? (const uint32_t *)nullptr : set->d_rowpos,
                           reinterpret_cast<const uint2 *>(set->d_posseg));
    SpArgs a{set->d_stream_s, set->Nstride, full ? (const uint32_t *)nullptr : set->d_rowstream, set->Nstride, set->d_meta, set->ntb, (uint32_t)set->S, (uint32_t)N,
             set->d_sperm, set->d_rowpos, tiles, ctl, ncb, cand32, (uint32_t)set->tiles_cap, reinterpret_cast<const uint2 *>(set->d_posseg)};
    // contiguous 32 KB per workgroup, workgroups in dispatch order: a streaming write (6.1 TB/s at N = 50 000: 825 us; the grid-stride loop over 16
    // workgroups per CU it replaces, whose iterations lie 16 MB apart, reached 4.6: 1105 us).  One store per thread is faster still (722-760 us) but when
    // the launch turns out dense all of its 19 M waves start only to return: 254 us instead of 34
    // (the multi-GPU engine fills a rank's slab at the START of its step, under the exchanges: d2g_bitslice_prefill)
    {
        // (pieces [0, prefilled_pieces) were written ahead of the launch: all of them by an early fill, some or all by the prepare's riders)
        const size_t pieces = std::min<size_t>(sp_fill_pieces(cnt), (size_t)0x7FFFFFFF);
        // (the same output, the same rows AND the same fill value -- a count launch into a buffer that was pre-filled for a table launch fills again)
        const size_t done = (set->prefilled == out_words && set->prefilled_cnt == cnt && set->prefilled_src == sp_fill_source(store)) ? std::min<size_t>(set->prefilled_pieces, pieces) : 0;
        if (done < pieces)
            hipLaunchKernelGGL((sp_fill_kernel<Store>), dim3((unsigned)(pieces - done)), dim3(SP_FILL_THREADS), 0, s,
                               out_words, cnt, store, (uint32_t)set->S, ctl, cand32, (uint32_t)done);
    }
    set->prefilled = nullptr;
    // a multiple of 8 (every XCD's list gets the same number of workgroups), four times what is resident at once: the lists differ in
    // length, and a workgroup that finds nothing at its index leaves at once -- the dispatcher evens the lists out sub-tile by sub-tile
    // (exactly one resident wave of workgroups took as long as the longest list: 52 -> 85 us at config 3)
    const unsigned grid = (unsigned)std::max<size_t>(8, std::min<size_t>(div_up<size_t>(ntile * SP_SUBS, 8) * 8, (size_t)ctx->num_cus * (4 * D2G_SP_WPE / D2G_SP_KS) * SP_GRID_MULT) / 8 * 8);
    // the pair list, composed region by region (before the pair kernel: that one STORES, see SpBins) -- when this set's prepare binned it
    if (set->sp_big) {
        const uint32_t band0 = (uint32_t)(r0 >> 5), nband = (uint32_t)((r1 - 1) >> 5) - band0 + 1u;
        const uint32_t ppb = (1u << set->bin_cshift) / SP_CMP_COLS;    // pieces of 1024 columns per chunk (1 up to ~23 000 sketches)
        SpComposeArgs ca{set->d_plist2, sp_bins_of(set), ctl, cand32, (uint32_t)N, (uint32_t)set->S, (uint32_t)r0, (uint32_t)r1, band0, ppb};
        const size_t nwg = (size_t)nband * set->bin_nch * ppb * (32 / SP_CMP_ROWS);
        if (nwg >= 0x7FFFFFFFu) { ctx->last_error = "bitslice sparse: too many compose workgroups"; return D2G_ERR_UNSUPPORTED; }
        hipLaunchKernelGGL((sp_compose_kernel<Store>), dim3((unsigned)nwg), dim3(SP_CMP_T), 0, s, ca, sh, store);
    }
    // (a short list is applied entry by entry: the pair kernel's tail adds, the gated launch behind it turns the sums into table values)
    SpPatchArgs pa{set->d_plist, set->d_plctl, (uint32_t)std::min<size_t>(set->plist_cap, 0xFFFFFFFFu), ctl, cand32, set->d_sinv, set->d_rowk, set->d_rowpos, reinterpret_cast<const uint2 *>(set->d_posseg), (uint32_t)N, bm, CW, (uint32_t)r0, (uint32_t)r1, full ? 1 : 0, std::min<uint32_t>(grid, (uint32_t)ctx->num_cus * 8u)};
    hipLaunchKernelGGL((k2_bitslice_sparse_kernel<SP_JR, Store>), dim3(grid), dim3(64 * D2G_SP_KS), 0, s, a, sh, store, pa);
    // behind the gate: every tile of the caller's-order operand in dense mode; otherwise the second step of a short pair list (table epilogue)
    if (dsh.nvalid_total)
        hipLaunchKernelGGL((k2_bitslice_kernel<BS_JR, Store>), dim3(dsh.per_xcd * 8), dim3(BS_THREADS), 0, s, set->d_stream,
                           set->Nstride, set->d_meta, set->ntb, (uint32_t)set->S, dsh, store, (const uint32_t *)ctl, cand32, pa);
    tm.stop();
    D2G_HIP(ctx, hipGetLastError());
    return D2G_OK;
}

// d2g_k2_patch.h -- internal: the sub-tile test shared by the work lists and the pair kernel of the sparse path (d2g_k2_sparse.h), and the pair
// list applied ENTRY BY ENTRY (the form for short lists); included by d2g_k2_bitslice.hip inside its anonymous namespace, before k2_bitslice_kernel.
// ---- the pair list applied to the filled output, entry by entry.  Short lists only (a clean family collection leaves a few thousand entries,
// its stragglers'): the host enqueues the binned form (d2g_k2_sparse.h: sp_hist_body / sp_bin_kernel / sp_compose_kernel -- three more kernels'
// worth of fixed cost, 20 ps per entry) when the set's last prepare left a long list, and this form (no launch of its own, 78 ps per entry)
// otherwise; plctl[7] says on the device which one applies, so a wrong guess costs time, never exactness.
// An entry (i < j, caller's indices) belongs to the launch when i is one of its rows; it counts only where the pair's tile is NOT listed (a
// listed tile is computed exactly by the pair kernel, shared values across segments included).  Two steps, both folded into kernels the
// launch runs anyway:
//   add   (tail of k2_bitslice_sparse_kernel: after the fill, beside the tile walk -- the two write disjoint positions)  atomicAdd of 1
//         onto the filled word.  Store = StoreEq: the filled word is 0 and the sum of the entries IS the count.  Store = StoreLut: the
//         filled word is the bit pattern of lut[0]; the adder that finds it untouched becomes the position's LEADER (flag in the entry);
//   lut   (k2_bitslice_kernel launched behind the sparse kernel: it walks every tile in dense mode and does THIS otherwise)  one kernel
//         boundary later -- every adder has finished -- the leader turns (word - pattern) into the table value.  Only the table path
//         sets leader flags and it clears every one it set: no stale flag survives a launch.
struct SpPatchArgs {
    unsigned long long *plist; const uint32_t *plctl; uint32_t plcap;
    const uint32_t *ctl; uint32_t cand;
    const uint32_t *sinv, *rowk, *rowpos; const uint2 *posseg; uint32_t N;
    const uint32_t *bm;                   // bm: the launch's tile bitmap (full launches: over sorted row blocks; partial: over launch-row blocks)
    uint32_t CW, r0, r1; int full;
    uint32_t nwg;                         // workgroups of the sparse pair kernel that share the entries (the others leave without looking at the list)
};
constexpr int SP_PL_BINNED = 7;           // plctl[7] != 0: sp_bin_kernel has binned this prepare's list, sp_compose_kernel applies it
// A sub-tile -- BS_IW launch rows whose first and last sorted positions are pf and pl, SP_SUBW sorted column positions from c0 -- holds a
// pair of ONE segment only if the segment of its last row ends behind c0 and the segment of its first row starts before the sub-tile's
// last column (segments are runs of sorted positions: their starts and ends grow with the position).  A listed tile's other sub-tiles
// are not walked; the pair list's entries that fall into them count like those of a tile that is not listed (BOTH sides ask this function).
// The sparse pair kernel's sub-tile: BS_IW rows x 64 SP_JR columns, D2G_SP_KS waves each walking a share of the register groups.  Measured in round 6
// with per-workgroup time stamps (profiles/r06_k2_experiments.txt, sections 7 and 13): at config 3 the bench's families hold ~150 sketches, the listed
// sub-tiles are mostly FULL, and the plane walk is one memory round trip per plane and wave (four waves: 104 planes x 180 ns = 19 us of a workgroup's 30).
// One column word per lane (SP_JR 1): 1 429 sub-tiles instead of 1 013, walk 17.5 us -- nothing gained.  Eight waves per sub-tile halve the walk, but with
// the mismatch counts in registers (72 VGPRs) only three such workgroups fit a CU and the 1 013 sub-tiles needed two rounds (40.9 us); with the counts
// added to LDS group by group (42 VGPRs, no scratch) four fit, every sub-tile is resident from the start: 35.7 us against 40.6 with four waves.
constexpr int SP_JR = BS_JR;
constexpr uint32_t SP_SUBW = 64u * SP_JR, SP_WC = BS_CB / SP_SUBW, SP_SUBS = (32u / BS_IW) * SP_WC;     // columns of a sub-tile, sub-tiles across a tile, sub-tiles of a tile
__device__ __forceinline__ bool sp_sub_empty(const uint2 *__restrict__ posseg, uint32_t pf, uint32_t pl, uint32_t c0) {
    return posseg[pl].y <= c0 || posseg[pf].x >= c0 + SP_SUBW;
}
constexpr unsigned long long SP_LEADER = 0x80000000ull;               // in the low word of an entry (i < 2^30: d2g_bitslice_alloc refuses larger N)
__device__ __forceinline__ bool sp_entry_wanted(const SpPatchArgs &a, uint32_t i, uint32_t j) {
    if (i < a.r0 || i >= a.r1) return false;
    const uint32_t pi = a.sinv[i], pj = a.sinv[j];
    uint32_t k, cpos;                                                  // launch row, sorted column position
    if (a.full) { k = min(pi, pj); cpos = max(pi, pj); }
    else { k = a.rowk[i]; cpos = pj; }
    if (!((a.bm[(size_t)(k >> 5) * a.CW + (cpos >> 13)] >> ((cpos >> 8) & 31)) & 1u)) return true;
    // a listed tile: computed exactly -- but for the sub-tiles the pair kernel skips
    const uint32_t k0 = k & ~(uint32_t)(BS_IW - 1), c0 = cpos & ~(SP_SUBW - 1u);
    uint32_t pf, pl;
    if (a.full) { pf = k0; pl = min(k0 + (uint32_t)BS_IW - 1u, a.N - 1u); }
    else { pf = a.rowpos[k0]; pl = a.rowpos[k0 + BS_IW - 1]; if (pl == 0xFFFFFFFFu) return false; }   // (the launch's last rows: walked)
    return sp_sub_empty(a.posseg, pf, pl, c0);
}
template <class Store> struct SpStoreTraits;
template <> struct SpStoreTraits<StoreEq> { static constexpr bool kLeader = false; };
template <> struct SpStoreTraits<StoreLut> { static constexpr bool kLeader = true; };
template <class Store>
__device__ __forceinline__ void sp_patch_add(const SpPatchArgs &a, const PairShape &sh, const Store &store, uint32_t S, size_t first, size_t stride) {
    if (a.plctl[SP_PL_BINNED]) return;                                   // the composed form has applied the list already
    const uint32_t fillv = store.value_from_mismatches(S, S);
    const uint32_t n = min(a.plctl[0], a.plcap);
    uint32_t *out = reinterpret_cast<uint32_t *>(store.out);
    for (size_t k = first; k < n; k += stride) {
        const unsigned long long e = a.plist[k];
        if (e == ~0ull) continue;                                          // an empty slot (SP_NULL_ENTRY)
        const uint32_t i = (uint32_t)e & 0x7FFFFFFFu, j = (uint32_t)(e >> 32);
        if (!sp_entry_wanted(a, i, j)) continue;
        const uint32_t old = atomicAdd(&out[out_pos(sh, i, j)], 1u);
        if (SpStoreTraits<Store>::kLeader && old == fillv) a.plist[k] = e | SP_LEADER;
    }
}
__device__ __forceinline__ void sp_patch_lut(const SpPatchArgs &a, const PairShape &sh, const StoreLut &store, uint32_t S, size_t first, size_t stride) {
    if (a.plctl[SP_PL_BINNED]) return;
    const uint32_t fillv = store.value_from_mismatches(S, S);
    const uint32_t n = min(a.plctl[0], a.plcap);
    uint32_t *out = reinterpret_cast<uint32_t *>(store.out);
    for (size_t k = first; k < n; k += stride) {
        const unsigned long long e = a.plist[k];
        if (!(e & SP_LEADER) || e == ~0ull) continue;                      // (an empty slot has every bit set)
        a.plist[k] = e & ~SP_LEADER;
        const uint32_t i = (uint32_t)e & 0x7FFFFFFFu, j = (uint32_t)(e >> 32);
        const size_t pos = out_pos(sh, i, j);
        const uint32_t neq = out[pos] - fillv;
        out[pos] = store.value_from_mismatches(S, S - min(neq, S));
    }
}
__device__ __forceinline__ void sp_patch_lut(const SpPatchArgs &, const PairShape &, const StoreEq &, uint32_t, size_t, size_t) {}   // counts need no second step


// d2g_k2_patch.h -- internal: the pair list's two patch steps (d2g_k2_sparse.h), declared ahead of the kernels that carry them
// (included by d2g_k2_bitslice.hip inside its anonymous namespace, before k2_bitslice_kernel).
// ---- the pair list applied to the filled output.  An entry (i < j, caller's indices) belongs to the launch when i is one of its rows;
// it counts only where the pair's tile is NOT listed (a listed tile was computed exactly, shared values across segments included).
// Two steps, both folded into kernels the launch runs anyway:
//   add   (tail of k2_bitslice_sparse_kernel: after the fill, beside the tile walk -- the two write disjoint positions)  atomicAdd of 1
//         onto the filled word.  Store = StoreEq: the filled word is 0 and the sum of the entries IS the count.  Store = StoreLut: the
//         filled word is the bit pattern of lut[0]; the adder that finds it untouched becomes the position's LEADER (flag in the entry);
//   lut   (k2_bitslice_kernel launched behind the sparse kernel: it walks every tile in dense mode and does THIS otherwise)  one kernel
//         boundary later -- every adder has finished -- the leader turns (word - pattern) into the table value.  Only the table path
//         sets leader flags and it clears every one it set: no stale flag survives a launch.
struct SpPatchArgs {
    unsigned long long *plist; const uint32_t *plctl; uint32_t plcap;
    const uint32_t *ctl; uint32_t cand;
    const uint32_t *sinv, *rowk, *bm;     // bm: the launch's tile bitmap (full launches: over sorted row blocks; partial: over launch-row blocks)
    uint32_t CW, r0, r1; int full;
    uint32_t nwg;                         // workgroups of the sparse pair kernel that share the entries (the others leave without looking at the list)
};
constexpr unsigned long long SP_LEADER = 0x80000000ull;               // in the low word of an entry (i < 2^30: d2g_bitslice_alloc refuses larger N)
__device__ __forceinline__ bool sp_entry_wanted(const SpPatchArgs &a, uint32_t i, uint32_t j) {
    if (i < a.r0 || i >= a.r1) return false;
    const uint32_t pi = a.sinv[i], pj = a.sinv[j];
    uint32_t rb, cpos;
    if (a.full) { rb = min(pi, pj) >> 5; cpos = max(pi, pj); }
    else { rb = a.rowk[i] >> 5; cpos = pj; }
    return !((a.bm[(size_t)rb * a.CW + (cpos >> 13)] >> ((cpos >> 8) & 31)) & 1u);
}
template <class Store> struct SpStoreTraits;
template <> struct SpStoreTraits<StoreEq> { static constexpr bool kLeader = false; };
template <> struct SpStoreTraits<StoreLut> { static constexpr bool kLeader = true; };
template <class Store>
__device__ __forceinline__ void sp_patch_add(const SpPatchArgs &a, const PairShape &sh, const Store &store, uint32_t S, size_t first, size_t stride) {
    const uint32_t fillv = store.value_from_mismatches(S, S);
    const uint32_t n = min(a.plctl[0], a.plcap);
    uint32_t *out = reinterpret_cast<uint32_t *>(store.out);
    for (size_t k = first; k < n; k += stride) {
        const unsigned long long e = a.plist[k];
        const uint32_t i = (uint32_t)e & 0x7FFFFFFFu, j = (uint32_t)(e >> 32);
        if (!sp_entry_wanted(a, i, j)) continue;
        const uint32_t old = atomicAdd(&out[out_pos(sh, i, j)], 1u);
        if (SpStoreTraits<Store>::kLeader && old == fillv) a.plist[k] = e | SP_LEADER;
    }
}
__device__ __forceinline__ void sp_patch_lut(const SpPatchArgs &a, const PairShape &sh, const StoreLut &store, uint32_t S, size_t first, size_t stride) {
    const uint32_t fillv = store.value_from_mismatches(S, S);
    const uint32_t n = min(a.plctl[0], a.plcap);
    uint32_t *out = reinterpret_cast<uint32_t *>(store.out);
    for (size_t k = first; k < n; k += stride) {
        const unsigned long long e = a.plist[k];
        if (!(e & SP_LEADER)) continue;
        a.plist[k] = e & ~SP_LEADER;
        const uint32_t i = (uint32_t)e & 0x7FFFFFFFu, j = (uint32_t)(e >> 32);
        const size_t pos = out_pos(sh, i, j);
        const uint32_t neq = out[pos] - fillv;
        out[pos] = store.value_from_mismatches(S, S - min(neq, S));
    }
}
__device__ __forceinline__ void sp_patch_lut(const SpPatchArgs &, const PairShape &, const StoreEq &, uint32_t, size_t, size_t) {}   // counts need no second step


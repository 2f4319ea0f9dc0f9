// d2g_k2.h -- internal: the prepared comparison operand shared by the K2 translation units.
#pragma once
#include "d2g_internal.h"

struct d2g_cmp_set {
    d2g_ctx *ctx = nullptr;
    size_t N = 0, S = 0;
    size_t Npad = 0;              // N rounded up to the column tile (256)
    int algo = D2G_CMP_DIRECT;    // algorithm actually prepared
    uint64_t *d_rows = nullptr;   // [N][S]     row-major 64-bit patterns
    uint64_t *d_cols = nullptr;   // [S][Npad]  register-major (transposed), zero padded
    // bit-sliced operand (algo == D2G_CMP_BITSLICE); all buffers are allocated once per set
    uint32_t *d_planes = nullptr; // exchanged form [ntb][nbits_cap+1][Nstride]: bit x of word = bit b of id[32*tb+x][j], unique values
                                  // coded 0; last slot = the "unique" plane (fixed geometry: groups are independent)
    uint32_t *d_stream = nullptr; // what the pair kernel walks: [live planes of all groups][2][Nstride] -- row-coded words, then the
                                  // same plane with unique values coded all-ones (column coding); + one block of slack
    size_t Nstride = 0;           // Npad + 64 (row tiles may read past Npad)
    int nbits_cap = 0;            // id-plane slots per 32-register group: smallest c with 2^c >= N/2 + 2
    int ntb = 0;                  // ceil(S/32)
    uint32_t *d_meta = nullptr;   // [tb] = max over the group's columns of (#values occurring >= 2 times) + 1 (device side; the kernels
                                  //       derive the live plane count from it, no host round trip)
    uint32_t *d_ids = nullptr;    // workspace: [S][Npad] dense ids
    uint32_t T = 0; int logT = 0; // hash space of the rank kernel (power of two >= 1.5 N)
    bool borrowed = false;        // planes/meta belong to the caller (d2g_cmp_set_from_planes_dev)
    bool want_exchange = false;   // d_planes is kept up to date by every prepare (set by the first export)
    // column plan (bs_colplan_kernel): the 32-register groups are formed from the columns SORTED by their live-plane class, so
    // that meta[tb] is the maximum over 32 similar columns (equality counts are a sum over columns: any permutation is exact)
    uint32_t *d_colcnt = nullptr; // [S][BS_CC_STRIDE]: per column, slots 0..3 = rank offset of each split of the rank kernel, slot 4 = #shared values
    uint32_t *d_perm = nullptr;   // [ntb*32]: the column that sits in each register slot of the operand, ~0 = padding
    int nsplit = 1;               // workgroups per column in the multi-partition rank kernel (narrow slices of large N: fills the CUs)
    // exporter sets (the multi-GPU engine's per-rank column slices): the prepare writes the exchange form, the group meta and its
    // status word straight into the caller's gathered operand; no plane stream, no private d_planes
    bool export_only = false;
    uint32_t *ex_planes = nullptr, *ex_meta = nullptr, *ex_status = nullptr;
    // engine-managed gathered sets (d2g_allpairs): the engine derives the plane stream itself (per chunk, as the groups arrive) and
    // points the set at the status words every rank's prepare contributed
    bool managed = false;
    const uint32_t *status_words = nullptr; int n_status = 0;
    // ---- sparse tiles + pair list (d2g_k2_sparse.h).  The sketches are put in an order that makes a family -- sketches that agree in
    // many registers -- a run of adjacent positions, the plane stream is built in THAT order, and an upper-triangle launch walks only
    // the 32 x 256 tiles a family's rows and columns meet in; pairs of DIFFERENT families that share a value by chance are counted from
    // a pair list; every other pair has 0 matches (the output is pre-filled with the value of 0).  Exact for any input.
    bool sparse_ok = false;       // eligible and enabled (D2G_BS_SPARSE, D2G_BS_SPARSE_MIN_N)
    unsigned sp_launch = 0;       // sparse launches so far: the control words are double-buffered (a launch zeroes the next one's)
    size_t ncols = 0;             // register columns the sparse path walks: S, or all ntb * 32 register slots of an engine-managed gathered operand
    bool ids_owned = false;       // engine-managed gathered operands: d_ids / d_colcnt were allocated for the sparse path (ids re-derived from the planes)
    bool srt_valid = false;       // d_stream_s / d_sperm describe the operand last prepared
    bool nat_valid = false;       // d_stream (caller's order; rectangular launches, dense launches) is up to date
    uint32_t *d_stream_s = nullptr;   // plane stream in sorted order
    uint32_t *d_sperm = nullptr;      // [Nstride] sketch at sorted position p (0xFFFFFFFF = padding)
    uint32_t *d_sinv = nullptr;       // [Npad]    sorted position of sketch j
    uint32_t *d_label = nullptr;      // [2][Npad] union-find labels (-> segment starts after the sort) | root of every sketch
    uint32_t *d_owner = nullptr;      // [S][owner_stride] one holder of every shared value (rank r -> owner[r - 1]); single-partition owning sets only
    size_t owner_stride = 0;
    uint32_t *d_segend = nullptr;     // [Npad]    end of the segment of root r (the sort's scan)
    uint32_t *d_posseg = nullptr;     // [Npad][2] (start, end) of the segment sorted position p lies in (sp_place_kernel; the sub-tile test of the pair kernel and the pair list)
    uint32_t *d_hint = nullptr;       // [2][Npad] per sketch the smallest holder of a value it shares (even / odd column pairs), 0xFFFFFFFF = none
    uint32_t *d_spz = nullptr;        // ONE block the prepare clears: the arrays below
    size_t spz_words = 0;
    uint32_t *d_lcnt = nullptr;       // [Npad+1]  counting sort: sketches per root, then the placing cursors (= segment ends)
    uint32_t *d_gbm = nullptr;        // 8 control words + the tile bitmap over ALL sorted row blocks (the segments' tiles); partial launches derive theirs from it
    uint32_t *d_order = nullptr;      // [8] [0] 1 = the launches walk every tile of the caller's-order operand (dense), [2] deep label chains
    uint32_t *d_plctl = nullptr;      // [12] [0] entries emitted into the pair list
    uint32_t *d_rowpos = nullptr;     // [Nstride] launch rows: sorted position of launch row k
    uint32_t *d_rowk = nullptr;       // [Npad]    launch row of sketch j (0xFFFFFFFF = not a row of this launch)
    uint32_t *d_rowstream = nullptr;  // [planes][Nstride] row-coded words of the launch rows, gathered (partial launches)
    uint32_t *d_tilebm = nullptr, *d_tiles = nullptr, *d_spctl = nullptr;   // a partial launch's tile bitmap, work list, 2 x 8 control words {tiles listed, flags, -, candidates}
    uint32_t *d_tiles_full = nullptr, *d_fullctl = nullptr;   // work lists + control words of a whole-triangle launch, left by the prepare (sp_permute_kernel)
    bool full_list_valid = false;
    uint32_t *h_gaveup = nullptr, *d_gaveup = nullptr;   // a word of mapped host memory: 1 = the last ordering raised order[0] (the next prepare skips the ordering)
    unsigned sp_prepares = 0; bool sp_skipped = false;
    void *fill_stream = nullptr, *fill_fork = nullptr, *fill_join = nullptr;   // (hipStream_t / hipEvent_t) a LARGE announced output is filled beside the rank kernel (d2g_bitslice_prepare)
    void *samp_stream = nullptr, *samp_event = nullptr;   // (hipStream_t / hipEvent_t) the first look runs beside the column plan and the planes kernel
    bool sample_pending = false; uint32_t sample_ticket = 0;   // the first look's kernels are enqueued; the word (h_gaveup[6]) their last workgroup writes when the sums are in
    int skip_cached = -1;             // this prepare's reading of the remembered give-up (-1: not read yet)
    uint32_t *d_samp = nullptr;       // [16][Npad] + 2: the first look at a matrix (sp_sample): registers shared with sixteen sampled sketches
    bool pred_valid = false, pred_dense = false; double pred_entries = 0, pred_family_pairs = 0;   // what the sample of THIS prepare says (valid until its ordering has been enqueued)
    bool sp_big = true;               // this prepare enqueued the binned form of the pair list (sp_expect_long_list)
    uint32_t *prefilled = nullptr;        // output the engine filled at the start of its step (d2g_bitslice_prefill): the next sparse launch into it skips its fill
    size_t prefilled_cnt = 0;             // ... and how many outputs that fill covered
    size_t prefilled_pieces = 0;          // ... and how many 32 KB pieces of it are written (all of them after an early fill; what rode on the prepare's kernels otherwise)
    const uint32_t *prefilled_src = nullptr;   // ... and WHAT was written: the table whose first entry is the fill value (nullptr: the count 0) -- a launch with another store fills for itself
    bool prefilled_by_riders = false;     // ... by the riders of a prepare (void once another prepare has run) or by d2g_cmp_ut_prefill_dev (valid until the next upper-triangle launch)
    int ride_mask = 0;
    // the output of the NEXT upper-triangle launch, announced ahead of the prepare (d2g_cmp_ut_announce_dev): the prepare's latency-bound
    // kernels (column plan, flatten, count, attach, scan, place: one to forty workgroups each) carry the fill as extra workgroups
    uint32_t *ride_out = nullptr; size_t ride_cnt = 0; const uint32_t *ride_vsrc = nullptr; uint32_t ride_vimm = 0;
    uint32_t ride_next = 0, ride_total = 0;   // pieces of 32 KB handed out so far / in all (0: nothing rides in this prepare)
    const uint32_t *last_ctl = nullptr;   // control words of the last sparse launch (d2g_cmp_set_sparse_info)
    unsigned long long *d_plist = nullptr;   // pair list: (i | j << 32), i < j caller's indices, one entry per (pair in different segments, shared value)
    size_t plist_cap = 0;
    // the list BINNED by output region (band of 32 rows x chunk of 2^bin_cshift columns): what sp_compose_kernel reads
    unsigned long long *d_plist2 = nullptr;
    uint32_t *d_binc = nullptr;                       // [nbins] entries per bin (inside the block the prepare clears)
    uint32_t *d_bstart = nullptr;                     // [nbins + 1] first entry of every bin in d_plist2
    uint32_t *d_cw_ents = nullptr; unsigned long long *d_cw_vals = nullptr; size_t cw_ecap = 0, cw_vcap = 0;   // sp_emit_kernel -> sp_pairs_kernel: the holders of the mixed values, one record per value
    uint32_t *d_hoff = nullptr;                       // [bin_nwg][nbins] where the entries a counting workgroup met go inside their bin (sp_hist_body -> sp_bin_kernel)
    uint32_t nbins = 0, bin_nch = 0, bin_cshift = 10, bin_nwg = 1;
    size_t tilebm_words = 0, tiles_cap = 0;
};
constexpr int BS_CC_STRIDE = 8;

// part of an upper-triangle launch's fill, carried by a kernel of the prepare chain: the workgroups from `own` on write pieces
// piece0 .. of 32 KB each (own = 0xFFFFFFFF: nobody rides)
struct SpRider { uint32_t *out; size_t cnt; const uint32_t *vsrc; uint32_t vimm, own, piece0; };

struct PairShape;
int  finish_shape(d2g_ctx *ctx, PairShape &sh, unsigned rb);   // d2g_k2.hip

// d2g_k2_bitslice part (same shared object)
void d2g_bitslice_geometry(d2g_cmp_set *set);
int  d2g_bitslice_alloc(d2g_ctx *ctx, d2g_cmp_set *set);
int  d2g_bitslice_prepare(d2g_ctx *ctx, d2g_cmp_set *set, hipStream_t s);
int  d2g_bitslice_export(d2g_ctx *ctx, d2g_cmp_set *set, hipStream_t s);
void d2g_bitslice_free(d2g_cmp_set *set);
int  d2g_bitslice_alloc_stream(d2g_ctx *ctx, d2g_cmp_set *set);
int  d2g_bitslice_ensure_natural(d2g_ctx *ctx, const d2g_cmp_set *set, hipStream_t s);   // caller's-order stream of a sparse set, on demand
// engine-managed gathered operands (d2g_mgpu.hip): buffers for the sparse-tile path, and the call that says "every group has arrived and
// its plane stream is derived": ids are re-derived from the planes, the sketches ordered, the sorted stream written
int  d2g_bitslice_managed_sparse_alloc(d2g_ctx *ctx, d2g_cmp_set *set);
int  d2g_bitslice_managed_ready(d2g_ctx *ctx, d2g_cmp_set *set, hipStream_t s);
int  d2g_bitslice_sparse_info(d2g_ctx *ctx, const d2g_cmp_set *set, hipStream_t s, uint32_t *out4);
int  d2g_bitslice_debug_read(d2g_ctx *ctx, const d2g_cmp_set *set, hipStream_t s, uint64_t *pairs_out, size_t cap, size_t *npairs, uint32_t *root_out);
int  d2g_bitslice_status(d2g_ctx *ctx, const d2g_cmp_set *set, hipStream_t s);   // synchronises; D2G_ERR_INTERNAL on overflow
// exporter set over an N x S_local column slice (no operand of its own); d2g_bitslice_prepare_slice transposes + prepares it
// into the target last given to d2g_bitslice_set_export_target
int  d2g_bitslice_exporter_create(d2g_ctx *ctx, size_t N, size_t S_local, d2g_cmp_set **out);
void d2g_bitslice_set_export_target(d2g_cmp_set *set, uint32_t *planes, uint32_t *meta, uint32_t *status);
int  d2g_bitslice_prepare_slice(d2g_ctx *ctx, d2g_cmp_set *set, const uint64_t *rows_dev, hipStream_t s);
// plane stream of groups [g0, g1) of a gathered operand (the groups before g0 must have their meta in place)
int  d2g_bitslice_derive_groups(d2g_ctx *ctx, const d2g_cmp_set *set, int g0, int g1, hipStream_t s);
// exactly one of (eq_out) or (lut,fout) is non-null
int  d2g_bitslice_ut(d2g_ctx *ctx, const d2g_cmp_set *set, size_t r0, size_t r1, uint32_t *eq_out,
                     const float *lut, float *fout, hipStream_t s);
// multi-GPU engine: the fill of a rank's slab enqueued at the start of the step (under the exchanges)
int  d2g_bitslice_prefill(d2g_ctx *ctx, d2g_cmp_set *set, size_t r0, size_t r1, uint32_t *eq_out, const float *lut, float *fout, hipStream_t s);
// the output of the next upper-triangle launch, announced before the prepare that precedes it: that prepare carries the fill
int  d2g_bitslice_announce(d2g_ctx *ctx, d2g_cmp_set *set, size_t r0, size_t r1, uint32_t *eq_out, const float *lut, float *fout);
void d2g_bitslice_forget(d2g_cmp_set *set);
int  d2g_bitslice_rect(d2g_ctx *ctx, const d2g_cmp_set *set, size_t a0, size_t a1, size_t b0, size_t b1,
                       uint32_t *eq_out, hipStream_t s);


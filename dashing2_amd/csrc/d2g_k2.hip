// d2g_k2.hip -- K2: dense all-pairs signature comparison (gfx950).
//
// Replaces HOT LOOP B of the reference: emit_rectangular's row loops (src/emitrect.cpp:211-323)
// calling compare() (src/cmp_core.cpp:349-361) whose inner loop is
// sketch::eq::count_gtlt / count_eq over S registers (src/cmp_core.cpp:461,506).
//
// DIRECT algorithm (this file, part 1)
//   operands: the N x S matrix of 64-bit patterns, kept twice in HBM:
//       rows [N][S]      row-major      -> the "A" operand, read with SCALAR loads (s_load_dwordx8)
//       cols [S][Npad]   register-major -> the "B" operand, lane l reads column j0+l (+64c): 512 B
//                                          contiguous per wave-instruction
//   a wave owns IW rows x (64*JR) columns of the pair matrix; per register index t every lane
//   does IW*JR  v_cmp_eq_u64 (SGPR pair vs VGPR pair) + v_addc_co_u32: all VALU issue is compare
//   work, A costs no VGPR/LDS traffic, B is shared by the 4 waves of the workgroup through L1/L2.
//   A workgroup (4 waves) owns a 32 x 256 tile; tiles are enumerated column-major and
//   swizzled so that one XCD's L2 keeps the B columns (S*256*8 B = 2 MiB at S=1024) its
//   workgroups share.
//
// BITSLICE algorithm (part 2, d2g_k2_bitslice.hip): per-column dense ids -> bit planes.
#include "d2g_internal.h"
#include "d2g_k2.h"
#include "d2g_k2_shape.h"
#include <algorithm>
#include <new>
#include <vector>

namespace {

constexpr int K2_THREADS = 256;
constexpr int K2_IW = 8;                 // rows per wave
constexpr int K2_JR = 4;                 // 64-column groups per lane
constexpr int K2_RB = 4 * K2_IW;         // rows per workgroup tile (32)
constexpr int K2_CB = 64 * K2_JR;        // cols per workgroup tile (256)
constexpr int K2_TC = 2;                 // registers per inner step (IW*TC*2 SGPRs of A values)

}  // namespace
// tile grid of a launch: counts the wanted tiles (host mirror of tile_of_block)
int finish_shape(d2g_ctx *ctx, PairShape &sh, unsigned rb) {
    sh.rb = rb;
    sh.nrt = (unsigned)div_up<size_t>(sh.i_hi - sh.i_lo, rb);
    sh.ct0 = (unsigned)(sh.j_lo / 256);
    sh.nct = (unsigned)(div_up<size_t>(sh.j_hi, 256) - sh.ct0);
    size_t total = 0;
    for (unsigned c = 0; c < sh.nct; ++c) total += tiles_in_column(sh, c);
    D2G_CHECK(ctx, total < (1ull << 31), "pair tile grid too large; shard rows");
    sh.nvalid_total = (unsigned)total;
    sh.per_xcd = (unsigned)div_up<size_t>(total, 8);
    return D2G_OK;
}
namespace {

// ---------------------------------------------------------------- transpose [N][S] -> [S][Npad]
// zero2: two words the bit-sliced prepare that follows wants cleared (its status word and the plan ticket), or null
__global__ __launch_bounds__(256) void k2_transpose_kernel(const uint64_t *__restrict__ rows, uint64_t *__restrict__ cols,
                                                           size_t N, size_t S, size_t Npad, uint32_t *zero2) {
    __shared__ uint64_t tile[32][33];
    if (zero2 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 2) zero2[threadIdx.x] = 0;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
    const size_t n0 = (size_t)blockIdx.y * 32, s0 = (size_t)blockIdx.x * 32;
    for (int r = ty; r < 32; r += 8) {
        const size_t n = n0 + r, s = s0 + tx;
        tile[r][tx] = (n < N && s < S) ? __builtin_nontemporal_load(&rows[n * S + s]) : 0ull;   // (read once: non-temporal, see sp_fill_kernel)
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const size_t s = s0 + r, n = n0 + tx;
        if (s < S && n < Npad) cols[s * Npad + n] = tile[tx][r];
    }
}

// (Measured in round 5 and dropped: 64 x 64 tiles, 8 bytes per lane, 16 independent loads per thread before the barrier, 512-byte runs
// both ways -- 46.3 us on average (27.7 at best) against 42.7 (25.5) for the kernel above in the bench's step.  The average is not a
// bytes-in-flight problem: the step before left 200 MB of freshly filled output behind, and whatever kernel comes next shares the HBM with
// that write-back.)
void launch_transpose(d2g_ctx *, const uint64_t *rows, uint64_t *cols, size_t N, size_t S, size_t Npad, uint32_t *zero2, hipStream_t s) {
    dim3 grid((unsigned)div_up<size_t>(S, 32), (unsigned)div_up<size_t>(Npad, 32));
    hipLaunchKernelGGL(k2_transpose_kernel, grid, dim3(256), 0, s, rows, cols, N, S, Npad, zero2);
}

// ---------------------------------------------------------------- direct compare kernel
template <bool GTLT, class Store>
__global__ __launch_bounds__(K2_THREADS) void k2_direct_kernel(const uint64_t *__restrict__ rows, const uint64_t *__restrict__ cols,
                                                               size_t S, size_t Npad, PairShape sh, Store store) {
    unsigned ct, rt;
    if (!tile_of_block(sh, blockIdx.x, ct, rt)) return;
    const size_t i0 = sh.i_lo + (size_t)rt * K2_RB;
    const size_t j0 = (size_t)(sh.ct0 + ct) * K2_CB;

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const size_t iw0 = i0 + (size_t)wave * K2_IW;
    if (iw0 >= sh.i_hi) return;
    if (sh.ut && j0 + K2_CB - 1 <= iw0) return;

    // rows is allocated with K2_RB zero rows of slack, so a partial last row tile reads in bounds
    const uint64_t *arow0 = rows + iw0 * S;
    const uint64_t *bcol = cols + j0 + lane;

    uint32_t acc[K2_IW][K2_JR];
    uint32_t accg[GTLT ? K2_IW : 1][GTLT ? K2_JR : 1];
#pragma unroll
    for (int i = 0; i < K2_IW; ++i)
#pragma unroll
        for (int c = 0; c < K2_JR; ++c) { acc[i][c] = 0; if (GTLT) accg[i][c] = 0; }

    size_t t = 0;
    for (; t + K2_TC <= S; t += K2_TC) {
        uint64_t bv[K2_TC][K2_JR];
#pragma unroll
        for (int tt = 0; tt < K2_TC; ++tt)
#pragma unroll
            for (int c = 0; c < K2_JR; ++c) bv[tt][c] = bcol[(t + tt) * Npad + 64 * c];
#pragma unroll
        for (int i = 0; i < K2_IW; ++i) {
#pragma unroll
            for (int tt = 0; tt < K2_TC; ++tt) {
                const uint64_t av = arow0[(size_t)i * S + t + tt];
#pragma unroll
                for (int c = 0; c < K2_JR; ++c) {
                    acc[i][c] += (av == bv[tt][c]);
                    if (GTLT) accg[i][c] += (av > bv[tt][c]);
                }
            }
        }
    }
    for (; t < S; ++t) {
        uint64_t bv[K2_JR];
#pragma unroll
        for (int c = 0; c < K2_JR; ++c) bv[c] = bcol[t * Npad + 64 * c];
#pragma unroll
        for (int i = 0; i < K2_IW; ++i) {
            const uint64_t av = arow0[(size_t)i * S + t];
#pragma unroll
            for (int c = 0; c < K2_JR; ++c) {
                acc[i][c] += (av == bv[c]);
                if (GTLT) accg[i][c] += (av > bv[c]);
            }
        }
    }
    if constexpr (GTLT) {
#pragma unroll
        for (int i = 0; i < K2_IW; ++i) {
            const size_t ii = iw0 + i;
            if (ii >= sh.i_hi) break;
#pragma unroll
            for (int c = 0; c < K2_JR; ++c) {
                const size_t jj = j0 + lane + 64 * c;
                if (jj < sh.j_hi && jj >= sh.j_lo && (!sh.ut || jj > ii)) store.put2(out_pos(sh, ii, jj), acc[i][c], accg[i][c]);
            }
        }
    } else {
        uint32_t val[K2_IW][K2_JR];
#pragma unroll
        for (int i = 0; i < K2_IW; ++i)
#pragma unroll
            for (int c = 0; c < K2_JR; ++c) val[i][c] = store.value(acc[i][c]);
#pragma unroll
        for (int i = 0; i < K2_IW; ++i) {
            const size_t ii = iw0 + i;
            if (ii >= sh.i_hi) break;
#pragma unroll
            for (int c = 0; c < K2_JR; ++c) {
                const size_t jj = j0 + lane + 64 * c;
                if (jj < sh.j_hi && jj >= sh.j_lo && (!sh.ut || jj > ii)) store.put(out_pos(sh, ii, jj), val[i][c]);
            }
        }
    }
}

template <bool GTLT, class Store>
int launch_direct(d2g_ctx *ctx, const d2g_cmp_set *set, PairShape sh, Store store, hipStream_t s) {
    if (int rc = finish_shape(ctx, sh, K2_RB)) return rc;
    if (sh.nvalid_total == 0) return D2G_OK;
    d2g_timer tm(ctx, &ctx->ev_k2, s);
    hipLaunchKernelGGL((k2_direct_kernel<GTLT, Store>), dim3(sh.per_xcd * 8), dim3(K2_THREADS), 0, s,
                       set->d_rows, set->d_cols, set->S, set->Npad, sh, store);
    tm.stop();
    D2G_HIP(ctx, hipGetLastError());
    return D2G_OK;
}

}  // namespace

// =====================================================================================
namespace {
__global__ __launch_bounds__(256) void k2_pack_slices_kernel(const uint64_t *__restrict__ rows, size_t n, size_t S, size_t Sl,
                                                             uint64_t *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;          // element of the row-major input
    if (i >= n * S) return;
    const size_t r = i / S, c = i - r * S, q = c / Sl;
    out[(q * n + r) * Sl + (c - q * Sl)] = rows[i];                    // reads and writes both coalesced (Sl >= 32)
}
}  // namespace


extern "C" {

void d2g_cmp_set_destroy(d2g_cmp_set *set) {
    if (!set) return;
    (void)hipSetDevice(set->ctx->device);
    (void)hipFree(set->d_rows);
    (void)hipFree(set->d_cols);
    d2g_bitslice_free(set);
    delete set;
}

int d2g_cmp_set_algo(const d2g_cmp_set *set) { return set ? set->algo : D2G_ERR_INVALID; }

// (re)load the operand: copy + transpose + (bitslice) ids/planes.  Everything is enqueued on `s`.
static int cmp_set_load(d2g_ctx *ctx, d2g_cmp_set *set, const uint64_t *sig_bits_dev, hipStream_t s) {
    const size_t N = set->N, S = set->S;
    // only the DIRECT kernel reads the row-major operand; bit-sliced sets transpose straight from the caller's buffer
    if (set->d_rows) D2G_HIP(ctx, hipMemcpyAsync(set->d_rows, sig_bits_dev, N * S * sizeof(uint64_t), hipMemcpyDeviceToDevice, s));
    d2g_timer tm(ctx, &ctx->ev_k2prep, s);
    launch_transpose(ctx, sig_bits_dev, set->d_cols, N, S, set->Npad, set->algo == D2G_CMP_BITSLICE ? set->d_meta + set->ntb : nullptr, s);
    int rc = D2G_OK;
    if (set->algo == D2G_CMP_BITSLICE) rc = d2g_bitslice_prepare(ctx, set, s);
    tm.stop();
    if (rc) return rc;
    D2G_HIP(ctx, hipGetLastError());
    return D2G_OK;
}

}  // extern "C"

// exporter sets (d2g_mgpu.hip): transpose the N x S_local slice and prepare it into the export target; timed as "k2prep"
int d2g_bitslice_prepare_slice(d2g_ctx *ctx, d2g_cmp_set *set, const uint64_t *rows_dev, hipStream_t s) {
    d2g_timer tm(ctx, &ctx->ev_k2prep, s);
    launch_transpose(ctx, rows_dev, set->d_cols, set->N, set->S, set->Npad, set->d_meta + set->ntb, s);
    const int rc = d2g_bitslice_prepare(ctx, set, s);
    tm.stop();
    if (rc) return rc;
    D2G_HIP(ctx, hipGetLastError());
    return D2G_OK;
}

extern "C" {

int d2g_cmp_set_create_dev(d2g_ctx *ctx, const uint64_t *sig_bits_dev, size_t N, size_t S, int algo,
                           void *stream, d2g_cmp_set **out) {
    if (!ctx || !out) return D2G_ERR_INVALID;
    *out = nullptr;
    D2G_CHECK(ctx, N >= 1 && S >= 1, "cmp_set: empty matrix");
    D2G_CHECK(ctx, S < (1ull << 31), "cmp_set: sketchsize too large");
    D2G_CHECK(ctx, sig_bits_dev != nullptr, "cmp_set: null signatures");
    D2G_CHECK(ctx, algo == D2G_CMP_AUTO || algo == D2G_CMP_DIRECT || algo == D2G_CMP_BITSLICE, "cmp_set: bad algo");
    D2G_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = as_stream(stream);
    d2g_cmp_set *set = new (std::nothrow) d2g_cmp_set();
    if (!set) return D2G_ERR_NOMEM;
    set->ctx = ctx; set->N = N; set->S = S;
    set->Npad = div_up<size_t>(N, K2_CB) * K2_CB;
    hipError_t e;
    if ((e = hipMalloc((void **)&set->d_cols, set->Npad * S * sizeof(uint64_t))) != hipSuccess) {
        ctx->last_error = hipGetErrorString(e);
        d2g_cmp_set_destroy(set);
        return D2G_ERR_NOMEM;
    }
    set->algo = D2G_CMP_DIRECT;
    if (algo != D2G_CMP_DIRECT) {
        int rc = d2g_bitslice_alloc(ctx, set);
        if (rc == D2G_OK) set->algo = D2G_CMP_BITSLICE;
        else if (!(rc == D2G_ERR_UNSUPPORTED && algo == D2G_CMP_AUTO)) { d2g_cmp_set_destroy(set); return rc; }
    }
    if (set->algo == D2G_CMP_DIRECT) {
        if ((e = hipMalloc((void **)&set->d_rows, (N + K2_RB) * S * sizeof(uint64_t))) != hipSuccess) {
            ctx->last_error = hipGetErrorString(e);
            d2g_cmp_set_destroy(set);
            return D2G_ERR_NOMEM;
        }
        // slack rows past N are read (never stored) by a partial last row tile
        (void)hipMemsetAsync(set->d_rows + N * S, 0, (size_t)K2_RB * S * sizeof(uint64_t), s);
    }
    int rc = cmp_set_load(ctx, set, sig_bits_dev, s);
    if (rc) { d2g_cmp_set_destroy(set); return rc; }
    *out = set;
    return D2G_OK;
}

int d2g_cmp_set_update_dev(d2g_ctx *ctx, d2g_cmp_set *set, const uint64_t *sig_bits_dev, void *stream) {
    if (!ctx) return D2G_ERR_INVALID;
    D2G_CHECK(ctx, set && set->ctx == ctx, "cmp_set_update: set belongs to another context");
    D2G_CHECK(ctx, !set->borrowed, "cmp_set_update: this set wraps a caller-owned operand");
    D2G_CHECK(ctx, sig_bits_dev != nullptr, "cmp_set_update: null signatures");
    D2G_HIP(ctx, hipSetDevice(ctx->device));
    return cmp_set_load(ctx, set, sig_bits_dev, as_stream(stream));
}

int d2g_cmp_set_planes(d2g_ctx *ctx, const d2g_cmp_set *set, void *stream, unsigned *max_distinct, int *nbits, float *mean_nbits) {
    if (!ctx) return D2G_ERR_INVALID;
    D2G_CHECK(ctx, set && set->ctx == ctx, "cmp_set_planes: set belongs to another context");
    unsigned md = 0; int nb = 0;
    if (set->algo == D2G_CMP_BITSLICE) {
        std::vector<unsigned> m(set->ntb);
        D2G_HIP(ctx, hipSetDevice(ctx->device));
        D2G_HIP(ctx, hipMemcpyAsync(m.data(), set->d_meta, m.size() * sizeof(unsigned), hipMemcpyDeviceToHost, as_stream(stream)));
        D2G_HIP(ctx, hipStreamSynchronize(as_stream(stream)));
        double sum = 0;
        if (int rc = d2g_bitslice_status(ctx, set, as_stream(stream))) return rc;
        for (unsigned x : m) {                       // per 32-register group: smallest b with 2^b > x (= D2 + 1), at least 1
            md = std::max(md, x);
            int b = 1;
            while ((1ull << b) <= x) ++b;
            nb = std::max(nb, b);
            sum += b;
        }
        if (mean_nbits) *mean_nbits = m.empty() ? 0.f : float(sum / m.size());
    } else if (mean_nbits) *mean_nbits = 0.f;
    if (max_distinct) *max_distinct = md;
    if (nbits) *nbits = nb;
    return D2G_OK;
}

int d2g_cmp_set_create(d2g_ctx *ctx, const uint64_t *sig_bits_host, size_t N, size_t S, int algo, d2g_cmp_set **out) {
    if (!ctx || !out) return D2G_ERR_INVALID;
    D2G_CHECK(ctx, sig_bits_host != nullptr && N >= 1 && S >= 1, "cmp_set: bad host matrix");
    D2G_HIP(ctx, hipSetDevice(ctx->device));
    uint64_t *tmp = nullptr;
    D2G_HIP(ctx, hipMalloc((void **)&tmp, N * S * sizeof(uint64_t)));
    hipError_t e = hipMemcpy(tmp, sig_bits_host, N * S * sizeof(uint64_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(tmp); ctx->last_error = hipGetErrorString(e); return D2G_ERR_HIP; }
    int rc = d2g_cmp_set_create_dev(ctx, tmp, N, S, algo, nullptr, out);
    (void)hipStreamSynchronize(nullptr);
    if (rc == D2G_OK && (*out)->algo == D2G_CMP_BITSLICE && d2g_bitslice_status(ctx, *out, nullptr) != D2G_OK) {
        // the rank kernel's partitioned LDS table overflowed (adversarial column): AUTO falls back to the
        // direct algorithm, an explicit BITSLICE request fails loudly
        d2g_cmp_set_destroy(*out);
        *out = nullptr;
        rc = algo == D2G_CMP_AUTO ? d2g_cmp_set_create_dev(ctx, tmp, N, S, D2G_CMP_DIRECT, nullptr, out) : (int)D2G_ERR_INTERNAL;
        (void)hipStreamSynchronize(nullptr);
    }
    (void)hipFree(tmp);
    return rc;
}

int d2g_cmp_set_sparse_info(d2g_ctx *ctx, const d2g_cmp_set *set, void *stream, uint32_t *info4) {
    if (!ctx || !set || !info4) return D2G_ERR_INVALID;
    D2G_HIP(ctx, hipSetDevice(ctx->device));
    if (set->algo != D2G_CMP_BITSLICE) { info4[0] = info4[1] = info4[2] = info4[3] = 0; return D2G_OK; }
    return d2g_bitslice_sparse_info(ctx, set, as_stream(stream), info4);
}

int d2g_cmp_set_debug_pairs(d2g_ctx *ctx, const d2g_cmp_set *set, void *stream, uint64_t *pairs_out, size_t cap, size_t *npairs, uint32_t *root_out) {
    if (!ctx || !set || !npairs) return D2G_ERR_INVALID;
    D2G_HIP(ctx, hipSetDevice(ctx->device));
    *npairs = 0;
    if (set->algo != D2G_CMP_BITSLICE) return D2G_OK;
    return d2g_bitslice_debug_read(ctx, set, as_stream(stream), pairs_out, cap, npairs, root_out);
}

int d2g_cmp_set_status(d2g_ctx *ctx, const d2g_cmp_set *set, void *stream) {
    if (!ctx) return D2G_ERR_INVALID;
    D2G_CHECK(ctx, set && set->ctx == ctx, "cmp_set_status: set belongs to another context");
    D2G_HIP(ctx, hipSetDevice(ctx->device));
    if (set->algo != D2G_CMP_BITSLICE) { D2G_HIP(ctx, hipStreamSynchronize(as_stream(stream))); return D2G_OK; }
    return d2g_bitslice_status(ctx, set, as_stream(stream));
}

static int check_rows(d2g_ctx *ctx, const d2g_cmp_set *set, size_t r0, size_t r1) {
    D2G_CHECK(ctx, set && set->ctx == ctx, "cmp: set belongs to another context");
    D2G_CHECK(ctx, r0 <= r1 && r1 <= set->N, "cmp: row range out of bounds");
    return D2G_OK;
}

static PairShape ut_shape(const d2g_cmp_set *set, size_t r0, size_t r1) {
    PairShape sh{};
    sh.N = set->N; sh.i_lo = r0; sh.i_hi = r1; sh.j_lo = r0 + 1 < set->N ? r0 + 1 : set->N; sh.j_hi = set->N; sh.ut = 1;
    return sh;
}

int d2g_cmp_eqcount_ut_dev(d2g_ctx *ctx, const d2g_cmp_set *set, size_t r0, size_t r1, uint32_t *out, void *stream) {
    if (!ctx) return D2G_ERR_INVALID;
    if (int rc = check_rows(ctx, set, r0, r1)) return rc;
    if (d2g_ut_count(set->N, r0, r1) == 0) return D2G_OK;
    D2G_CHECK(ctx, out != nullptr, "cmp: null output");
    D2G_HIP(ctx, hipSetDevice(ctx->device));
    if (set->algo == D2G_CMP_BITSLICE) return d2g_bitslice_ut(ctx, set, r0, r1, out, nullptr, nullptr, as_stream(stream));
    return launch_direct<false>(ctx, set, ut_shape(set, r0, r1), StoreEq{out}, as_stream(stream));
}

int d2g_cmp_lut_ut_dev(d2g_ctx *ctx, const d2g_cmp_set *set, size_t r0, size_t r1, const float *lut, float *out, void *stream) {
    if (!ctx) return D2G_ERR_INVALID;
    if (int rc = check_rows(ctx, set, r0, r1)) return rc;
    if (d2g_ut_count(set->N, r0, r1) == 0) return D2G_OK;
    D2G_CHECK(ctx, out != nullptr && lut != nullptr, "cmp: null output/lut");
    D2G_HIP(ctx, hipSetDevice(ctx->device));
    if (set->algo == D2G_CMP_BITSLICE) return d2g_bitslice_ut(ctx, set, r0, r1, nullptr, lut, out, as_stream(stream));
    return launch_direct<false>(ctx, set, ut_shape(set, r0, r1), StoreLut{out, lut}, as_stream(stream));
}

// The fill of an upper-triangle launch, enqueued AHEAD of it (include/d2g.h): exactly one of neq_out / (lut, out) is given.
int d2g_cmp_ut_prefill_dev(d2g_ctx *ctx, d2g_cmp_set *set, size_t r0, size_t r1, uint32_t *neq_out, const float *lut, float *out, void *stream) {
    if (!ctx) return D2G_ERR_INVALID;
    if (int rc = check_rows(ctx, set, r0, r1)) return rc;
    if (d2g_ut_count(set->N, r0, r1) == 0) return D2G_OK;
    D2G_CHECK(ctx, (neq_out != nullptr) != (lut != nullptr && out != nullptr), "cmp prefill: give the count output, or the table and the float output");
    if (set->algo != D2G_CMP_BITSLICE) return D2G_OK;                  // the direct kernel writes every output itself
    D2G_HIP(ctx, hipSetDevice(ctx->device));
    return d2g_bitslice_prefill(ctx, set, r0, r1, neq_out, lut, out, as_stream(stream));
}

// The output of the next upper-triangle launch, announced AHEAD of the prepare (include/d2g.h): the prepare carries the fill.
int d2g_cmp_ut_announce_dev(d2g_ctx *ctx, d2g_cmp_set *set, size_t r0, size_t r1, uint32_t *neq_out, const float *lut, float *out) {
    if (!ctx) return D2G_ERR_INVALID;
    if (int rc = check_rows(ctx, set, r0, r1)) return rc;
    if (d2g_ut_count(set->N, r0, r1) == 0) return D2G_OK;
    if (set->algo != D2G_CMP_BITSLICE || set->borrowed) return D2G_OK;   // the direct kernel writes every output itself; a borrowed operand is never re-prepared
    if (!neq_out && !out) return d2g_bitslice_announce(ctx, set, r0, r1, nullptr, nullptr, nullptr);   // cancel: the set forgets the announced pointer
    D2G_CHECK(ctx, (neq_out != nullptr) != (lut != nullptr && out != nullptr), "cmp announce: give the count output, or the table and the float output");
    return d2g_bitslice_announce(ctx, set, r0, r1, neq_out, lut, out);
}

int d2g_cmp_set_forget(d2g_ctx *ctx, d2g_cmp_set *set) {
    if (!ctx) return D2G_ERR_INVALID;
    D2G_CHECK(ctx, set && set->ctx == ctx, "cmp_set_forget: set belongs to another context");
    if (set->algo == D2G_CMP_BITSLICE) d2g_bitslice_forget(set);
    return D2G_OK;
}

int d2g_cmp_gtlt_ut_dev(d2g_ctx *ctx, const d2g_cmp_set *set, size_t r0, size_t r1, uint32_t *gt, uint32_t *lt, void *stream) {
    if (!ctx) return D2G_ERR_INVALID;
    if (int rc = check_rows(ctx, set, r0, r1)) return rc;
    if (d2g_ut_count(set->N, r0, r1) == 0) return D2G_OK;
    D2G_CHECK(ctx, gt != nullptr && lt != nullptr, "cmp: null output");
    D2G_CHECK(ctx, set->d_rows != nullptr, "cmp: (gt,lt) needs the raw patterns: create the set with D2G_CMP_DIRECT");
    D2G_HIP(ctx, hipSetDevice(ctx->device));
    // order needs the raw patterns: the direct kernel on a DIRECT set
    return launch_direct<true>(ctx, set, ut_shape(set, r0, r1), StoreGtLt{gt, lt, (uint32_t)set->S}, as_stream(stream));
}

int d2g_cmp_eqcount_rect_dev(d2g_ctx *ctx, const d2g_cmp_set *set, size_t a0, size_t a1, size_t b0, size_t b1,
                             uint32_t *out, void *stream) {
    if (!ctx) return D2G_ERR_INVALID;
    D2G_CHECK(ctx, set && set->ctx == ctx, "cmp: set belongs to another context");
    D2G_CHECK(ctx, a0 <= a1 && a1 <= set->N && b0 <= b1 && b1 <= set->N, "cmp: rect out of bounds");
    if (a0 == a1 || b0 == b1) return D2G_OK;
    D2G_CHECK(ctx, out != nullptr, "cmp: null output");
    D2G_HIP(ctx, hipSetDevice(ctx->device));
    PairShape sh{};
    sh.N = set->N; sh.i_lo = a0; sh.i_hi = a1; sh.j_lo = b0; sh.j_hi = b1; sh.ut = 0;
    if (set->algo == D2G_CMP_BITSLICE) return d2g_bitslice_rect(ctx, set, a0, a1, b0, b1, out, as_stream(stream));
    return launch_direct<false>(ctx, set, sh, StoreEq{out}, as_stream(stream));
}

int d2g_cmp_gtlt_rect_dev(d2g_ctx *ctx, const d2g_cmp_set *set, size_t a0, size_t a1, size_t b0, size_t b1,
                          uint32_t *gt, uint32_t *lt, void *stream) {
    if (!ctx) return D2G_ERR_INVALID;
    D2G_CHECK(ctx, set && set->ctx == ctx, "cmp: set belongs to another context");
    D2G_CHECK(ctx, a0 <= a1 && a1 <= set->N && b0 <= b1 && b1 <= set->N, "cmp: rect out of bounds");
    if (a0 == a1 || b0 == b1) return D2G_OK;
    D2G_CHECK(ctx, gt != nullptr && lt != nullptr, "cmp: null output");
    D2G_CHECK(ctx, set->d_rows != nullptr, "cmp: (gt,lt) needs the raw patterns: create the set with D2G_CMP_DIRECT");
    D2G_HIP(ctx, hipSetDevice(ctx->device));
    PairShape sh{};
    sh.N = set->N; sh.i_lo = a0; sh.i_hi = a1; sh.j_lo = b0; sh.j_hi = b1; sh.ut = 0;
    return launch_direct<true>(ctx, set, sh, StoreGtLt{gt, lt, (uint32_t)set->S}, as_stream(stream));
}

// ---------------------------------------------------------------- sharded prepare (multi-GPU)
int d2g_operand_layout(size_t N, size_t S, size_t *group_words, size_t *ngroups) {
    if (!N || !S) return D2G_ERR_INVALID;
    d2g_cmp_set tmp;
    tmp.N = N; tmp.S = S; tmp.Npad = div_up<size_t>(N, K2_CB) * K2_CB;
    d2g_bitslice_geometry(&tmp);
    if (group_words) *group_words = (size_t)(tmp.nbits_cap + 1) * tmp.Nstride;
    if (ngroups) *ngroups = (size_t)tmp.ntb;
    return D2G_OK;
}

int d2g_cmp_set_export_operand_dev(d2g_ctx *ctx, const d2g_cmp_set *set, uint32_t *planes_out_dev, uint32_t *meta_out_dev, void *stream) {
    if (!ctx) return D2G_ERR_INVALID;
    D2G_CHECK(ctx, set && set->ctx == ctx, "export_operand: set belongs to another context");
    D2G_CHECK(ctx, set->algo == D2G_CMP_BITSLICE, "export_operand: not a bit-sliced set");
    D2G_CHECK(ctx, planes_out_dev && meta_out_dev, "export_operand: null output");
    D2G_HIP(ctx, hipSetDevice(ctx->device));
    const size_t words = (size_t)set->ntb * (set->nbits_cap + 1) * set->Nstride;
    if (int rc = d2g_bitslice_export(ctx, const_cast<d2g_cmp_set *>(set), as_stream(stream))) return rc;
    D2G_HIP(ctx, hipMemcpyAsync(planes_out_dev, set->d_planes, words * sizeof(uint32_t), hipMemcpyDeviceToDevice, as_stream(stream)));
    D2G_HIP(ctx, hipMemcpyAsync(meta_out_dev, set->d_meta, (size_t)set->ntb * sizeof(uint32_t), hipMemcpyDeviceToDevice, as_stream(stream)));
    return D2G_OK;
}

int d2g_cmp_set_from_planes_dev(d2g_ctx *ctx, size_t N, size_t S, const uint32_t *planes_dev, const uint32_t *meta_dev, d2g_cmp_set **out) {
    if (!ctx || !out) return D2G_ERR_INVALID;
    *out = nullptr;
    D2G_CHECK(ctx, N >= 1 && S >= 1 && N < (1ull << 30) && S < (1ull << 31), "from_planes: bad shape");
    D2G_CHECK(ctx, planes_dev && meta_dev, "from_planes: null operand");
    d2g_cmp_set *set = new (std::nothrow) d2g_cmp_set();
    if (!set) return D2G_ERR_NOMEM;
    set->ctx = ctx; set->N = N; set->S = S;
    set->Npad = div_up<size_t>(N, K2_CB) * K2_CB;
    d2g_bitslice_geometry(set);
    set->algo = D2G_CMP_BITSLICE;
    set->borrowed = true;
    set->d_planes = const_cast<uint32_t *>(planes_dev);
    set->d_meta = const_cast<uint32_t *>(meta_dev);
    D2G_HIP(ctx, hipSetDevice(ctx->device));
    if (int rc = d2g_bitslice_alloc_stream(ctx, set)) { delete set; return rc; }   // plane stream, derived before every launch
    *out = set;
    return D2G_OK;
}

// rows [n][S] -> W consecutive blocks [n][S/W] (block q = columns [q S/W, (q+1) S/W)): the send
// layout of the row-slice -> column-slice all-to-all.  One launch (W strided 2-D copies cost W launch
// latencies on the critical path of every multi-GPU step).
int d2g_pack_column_slices_dev(d2g_ctx *ctx, const uint64_t *rows_dev, size_t n, size_t S, int W, uint64_t *out_dev, void *stream) {
    if (!ctx) return D2G_ERR_INVALID;
    D2G_CHECK(ctx, W >= 1 && S % (size_t)W == 0, "pack_column_slices: S must be divisible by the number of ranks");
    if (!n) return D2G_OK;
    D2G_CHECK(ctx, rows_dev && out_dev, "pack_column_slices: null buffer");
    D2G_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k2_pack_slices_kernel, dim3((unsigned)div_up<size_t>(n * S, 256)), dim3(256), 0, as_stream(stream),
                       rows_dev, n, S, S / W, out_dev);
    D2G_HIP(ctx, hipGetLastError());
    return D2G_OK;
}

// ---------------------------------------------------------------- host-pointer conveniences
int d2g_cmp_eqcount_ut(d2g_ctx *ctx, const uint64_t *sig_bits, size_t N, size_t S, size_t r0, size_t r1, int algo,
                       uint32_t *neq_out) {
    if (!ctx) return D2G_ERR_INVALID;
    d2g_cmp_set *set = nullptr;
    int rc = d2g_cmp_set_create(ctx, sig_bits, N, S, algo, &set);
    if (rc) return rc;
    const size_t cnt = d2g_ut_count(N, r0, r1);
    uint32_t *d_out = nullptr;
    if (cnt) {
        hipError_t e = hipMalloc((void **)&d_out, cnt * sizeof(uint32_t));
        if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); d2g_cmp_set_destroy(set); return D2G_ERR_NOMEM; }
    }
    rc = d2g_cmp_eqcount_ut_dev(ctx, set, r0, r1, d_out, nullptr);
    if (rc == D2G_OK && cnt) {
        hipError_t e = hipMemcpy(neq_out, d_out, cnt * sizeof(uint32_t), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); rc = D2G_ERR_HIP; }
    }
    (void)hipFree(d_out);
    d2g_cmp_set_destroy(set);
    return rc;
}

int d2g_cmp_dist_ut(d2g_ctx *ctx, const uint64_t *sig_bits, const double *cards, size_t N, size_t S, size_t r0,
                    size_t r1, int measure, int k, int multiset_space, int algo, int nthreads, float *out) {
    if (!ctx) return D2G_ERR_INVALID;
    D2G_CHECK(ctx, measure >= D2G_SIMILARITY && measure <= D2G_UNION_SIZE, "cmp: unknown measure");
    D2G_CHECK(ctx, r0 <= r1 && r1 <= N, "cmp: row range out of bounds");
    const size_t cnt = d2g_ut_count(N, r0, r1);
    if (!cnt) return D2G_OK;
    D2G_CHECK(ctx, out != nullptr && cards != nullptr, "cmp: null output/cards");
    if (nthreads < 1) nthreads = 1;
    d2g_cmp_set *set = nullptr;
    std::vector<float> lut(S + 1);
    const bool have_lut = d2g_epilogue_lut(S, measure, k, multiset_space, lut.data()) == D2G_OK;
    // (gt,lt) are only needed in set space when the value is not a function of neq alone
    const bool need_gtlt = !multiset_space && (S & (S - 1)) != 0;
    int rc = d2g_cmp_set_create(ctx, sig_bits, N, S, need_gtlt ? (int)D2G_CMP_DIRECT : algo, &set);
    if (rc) return rc;
    void *d_a = nullptr, *d_b = nullptr, *d_lut = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_a); (void)hipFree(d_b); (void)hipFree(d_lut); d2g_cmp_set_destroy(set); };
#define D2G_TRY(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { ctx->last_error = hipGetErrorString(e__); cleanup(); return D2G_ERR_HIP; } } while (0)
    D2G_TRY(hipMalloc(&d_a, cnt * 4));
    if (have_lut) {
        // card-independent measure, value = f(neq): fused table epilogue on the device
        D2G_TRY(hipMalloc(&d_lut, (S + 1) * sizeof(float)));
        D2G_TRY(hipMemcpy(d_lut, lut.data(), (S + 1) * sizeof(float), hipMemcpyHostToDevice));
        rc = d2g_cmp_lut_ut_dev(ctx, set, r0, r1, (const float *)d_lut, (float *)d_a, nullptr);
        if (rc == D2G_OK) D2G_TRY(hipMemcpy(out, d_a, cnt * 4, hipMemcpyDeviceToHost));
        cleanup();
        return rc;
    }
    // integer counts from the device, x87 epilogue on the host (bit-exact with cmp_core.cpp:458-517)
    std::vector<uint32_t> ca(cnt), cb;
    if (need_gtlt) {
        D2G_TRY(hipMalloc(&d_b, cnt * 4));
        rc = d2g_cmp_gtlt_ut_dev(ctx, set, r0, r1, (uint32_t *)d_a, (uint32_t *)d_b, nullptr);
        if (rc == D2G_OK) {
            cb.resize(cnt);
            D2G_TRY(hipMemcpy(ca.data(), d_a, cnt * 4, hipMemcpyDeviceToHost));
            D2G_TRY(hipMemcpy(cb.data(), d_b, cnt * 4, hipMemcpyDeviceToHost));
        }
    } else {
        rc = d2g_cmp_eqcount_ut_dev(ctx, set, r0, r1, (uint32_t *)d_a, nullptr);
        if (rc == D2G_OK) D2G_TRY(hipMemcpy(ca.data(), d_a, cnt * 4, hipMemcpyDeviceToHost));
    }
#undef D2G_TRY
    cleanup();
    if (rc) return rc;
    d2g_epilogue_ut(ca.data(), need_gtlt ? cb.data() : nullptr, cards, N, S, r0, r1, measure, k,
                         multiset_space, nthreads, out);
    return D2G_OK;
}

}  // extern "C"

void d2g_warm_k2() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k2_transpose_kernel)); }

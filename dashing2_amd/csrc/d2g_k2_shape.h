// d2g_k2_shape.h -- internal: pair-matrix tiling, addressing and store functors shared by the
// direct and bit-sliced K2 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

struct PairShape {
    size_t N;        // sketches in the set
    size_t i_lo, i_hi;   // rows of the pair matrix this launch covers
    size_t j_lo, j_hi;   // columns
    int ut;          // 1: only j > i, condensed upper-triangular addressing
    unsigned nrt, nct;   // tile grid
    unsigned ct0;        // first column tile (j_lo / K2_CB)
    unsigned nblk;       // nrt * nct
    unsigned per_xcd;    // ceil(nblk / 8)
};

__device__ __forceinline__ size_t out_pos(const PairShape &sh, size_t i, size_t j) {
    if (sh.ut) {
        // rows i_lo..i-1 hold (N-1-r) entries each
        const size_t d = i - sh.i_lo;
        const size_t tri_i = i * (i - 1) / 2 * (i != 0), tri_0 = sh.i_lo ? sh.i_lo * (sh.i_lo - 1) / 2 : 0;
        return d * (sh.N - 1) - (tri_i - tri_0) + (j - i - 1);
    }
    return (i - sh.i_lo) * (sh.j_hi - sh.j_lo) + (j - sh.j_lo);
}

struct StoreEq  { uint32_t *out; __device__ __forceinline__ void operator()(size_t pos, uint32_t eq, uint32_t) const { out[pos] = eq; } };
struct StoreLut { float *out; const float *lut; __device__ __forceinline__ void operator()(size_t pos, uint32_t eq, uint32_t) const { out[pos] = lut[eq]; } };
struct StoreGtLt {
    uint32_t *gt, *lt; uint32_t S;
    __device__ __forceinline__ void operator()(size_t pos, uint32_t eq, uint32_t g) const { gt[pos] = g; lt[pos] = S - g - eq; }
};


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
    unsigned nrt, nct;   // tile grid (row tiles of RB rows, column tiles of 256 columns)
    unsigned ct0;        // first column tile (j_lo / 256)
    unsigned rb;         // rows per workgroup tile
    unsigned nvalid_total;   // tiles that contain at least one wanted pair
    unsigned per_xcd;        // ceil(nvalid_total / 8)
};

// number of row tiles of column tile c (0-based within the launch) that hold a pair with j > i:
// tile (rt, c) is wanted iff i_lo + rt*rb < 256*(ct0+c+1) - 1.  256 % rb == 0.
__host__ __device__ __forceinline__ unsigned tiles_in_column(const PairShape &sh, unsigned c) {
    if (!sh.ut) return sh.nrt;
    const long long x = (long long)(256 / sh.rb) * (long long)(sh.ct0 + c + 1) - (long long)((sh.i_lo + 1) / sh.rb);
    return x <= 0 ? 0u : (x >= (long long)sh.nrt ? sh.nrt : (unsigned)x);
}

// Workgroup b -> tile.  Workgroups are dealt to XCDs round-robin by the hardware (b % 8); every XCD
// gets an equal COUNT of wanted tiles, contiguous in column-major order, so its L2 keeps the
// column operand its workgroups share and the triangle stays load-balanced across XCDs.
__device__ __forceinline__ bool tile_of_block(const PairShape &sh, unsigned b, unsigned &ct, unsigned &rt) {
    unsigned v = (b & 7u) * sh.per_xcd + (b >> 3);
    if (v >= sh.nvalid_total) return false;
    unsigned c = 0;
    for (;; ++c) {                                   // uniform scalar scan over <= nct column tiles
        const unsigned n = tiles_in_column(sh, c);
        if (v < n) break;
        v -= n;
    }
    ct = c; rt = v;
    return true;
}

__device__ __forceinline__ size_t out_pos(const PairShape &sh, size_t i, size_t j) {
    if (sh.ut) {
        // rows i_lo..i-1 hold (N-1-r) entries each
        const size_t d = i - sh.i_lo;
        const size_t tri_i = i * (i - 1) / 2 * (i != 0), tri_0 = sh.i_lo ? sh.i_lo * (sh.i_lo - 1) / 2 : 0;
        return d * (sh.N - 1) - (tri_i - tri_0) + (j - i - 1);
    }
    return (i - sh.i_lo) * (sh.j_hi - sh.j_lo) + (j - sh.j_lo);
}

// wave-uniform part of out_pos: out_pos(sh, i, j) == out_row_base(sh, i) + j (mod 2^64) for every wanted pair
__device__ __forceinline__ size_t out_row_base(const PairShape &sh, size_t i) {
    if (sh.ut) {
        const size_t d = i - sh.i_lo;
        const size_t tri_i = i * (i - 1) / 2 * (i != 0), tri_0 = sh.i_lo ? sh.i_lo * (sh.i_lo - 1) / 2 : 0;
        return d * (sh.N - 1) - (tri_i - tri_0) - i - 1;
    }
    return (i - sh.i_lo) * (sh.j_hi - sh.j_lo) - sh.j_lo;
}

// (Non-temporal stores for the pair kernels' outputs were measured in round 5 beside the non-temporal fill: sparse pair kernel 42 -> 55 us,
// step 0.267 -> 0.279 ms at config 3 -- the tiles' 512-byte runs end in partial lines that are merged where the line is cached.  Plain stores.)
// Store functors: value(eq) is evaluated for all of a lane's outputs first (independent gathers
// in flight together), put(pos, v) afterwards -- the table and the output never alias.
struct StoreEq {
    uint32_t *__restrict__ out;
    __device__ __forceinline__ uint32_t value(uint32_t eq) const { return eq; }
    __device__ __forceinline__ void put(size_t pos, uint32_t v) const { out[pos] = v; }
    __device__ __forceinline__ void put_row(size_t row_base, uint32_t j, uint32_t v) const { (out + row_base)[j] = v; }
    // interior-tile form: value from the mismatch count, store at (uniform row base) + (32-bit lane column)
    __device__ __forceinline__ uint32_t value_from_mismatches(uint32_t S, uint32_t mm) const { return S - mm; }
};
struct StoreLut {
    float *__restrict__ out; const float *__restrict__ lut;
    __device__ __forceinline__ uint32_t value(uint32_t eq) const { return __float_as_uint(lut[eq]); }
    __device__ __forceinline__ void put(size_t pos, uint32_t v) const { out[pos] = __uint_as_float(v); }
    __device__ __forceinline__ void put_row(size_t row_base, uint32_t j, uint32_t v) const { (out + row_base)[j] = __uint_as_float(v); }
    __device__ __forceinline__ uint32_t value_from_mismatches(uint32_t S, uint32_t mm) const {
        return __float_as_uint(lut[S - mm]);
    }
};
struct StoreGtLt {
    uint32_t *__restrict__ gt, *__restrict__ lt; uint32_t S;
    __device__ __forceinline__ void put2(size_t pos, uint32_t eq, uint32_t g) const { gt[pos] = g; lt[pos] = S - g - eq; }
};

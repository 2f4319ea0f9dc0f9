// tools/microbench.hip -- VALU issue-rate microbenchmarks for the instructions K1/K2 are built on
// (gfx950).  Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o /tmp/mb ; run on the GPU box.
// Each kernel runs ITER iterations of UNROLL independent chains per lane; reports wave-instructions/s
// per CU-SIMD and the implied cycles per wave-instruction at the measured clock (assume 2.4 GHz).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITER = 4096;

__global__ void k_fma(float *out, float a, float b) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x + i;
    for (int it = 0; it < ITER; ++it)
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], a, b);
    float s = 0; for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_bitop3(uint32_t *out, const uint32_t *__restrict__ sc) {
    uint32_t z[16], v = threadIdx.x * 2654435761u;
    uint32_t s0 = sc[0], s1 = sc[1], s2 = sc[2], s3 = sc[3];
    for (int i = 0; i < 16; ++i) z[i] = i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) z[i] = __builtin_amdgcn_bitop3_b32((i & 3) == 0 ? s0 : (i & 3) == 1 ? s1 : (i & 3) == 2 ? s2 : s3, v, z[i], 0xBE);
        v = v * 3 + 1; asm volatile("" : "+v"(v));
    }
    uint32_t s = 0; for (int i = 0; i < 16; ++i) s ^= z[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_xor_or(uint32_t *out, const uint32_t *__restrict__ sc) {
    uint32_t z[16], v = threadIdx.x * 2654435761u;
    uint32_t s0 = sc[0];
    for (int i = 0; i < 16; ++i) z[i] = i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { uint32_t t; asm volatile("v_xor_b32 %0, %1, %2" : "=v"(t) : "s"(s0), "v"(v)); asm volatile("v_or_b32 %0, %1, %2" : "=v"(z[i]) : "v"(z[i]), "v"(t)); }
        v = v * 3 + 1;
    }
    uint32_t s = 0; for (int i = 0; i < 16; ++i) s ^= z[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// two planes per step without v_bitop3: t1 = s0 ^ v1, t2 = s1 ^ v2 (VOP2, scalar source), z = or3(z, t1, t2)
__global__ void k_xor2_or3(uint32_t *out, const uint32_t *__restrict__ sc) {
    uint32_t z[16], v = threadIdx.x * 2654435761u, v2 = v ^ 0x5bd1e995u;
    uint32_t s0 = sc[0], s1 = sc[1];
    for (int i = 0; i < 16; ++i) z[i] = i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            uint32_t t1, t2;
            asm volatile("v_xor_b32 %0, %1, %2" : "=v"(t1) : "s"(s0), "v"(v));
            asm volatile("v_xor_b32 %0, %1, %2" : "=v"(t2) : "s"(s1), "v"(v2));
            asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(z[i]) : "v"(t1), "v"(t2));
        }
        v = v * 3 + 1; v2 = v2 * 5 + 1;
    }
    uint32_t s = 0; for (int i = 0; i < 16; ++i) s ^= z[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// the same two planes with v_bitop3 (what the pair kernel does): z = z | (s0 ^ v1); z = z | (s1 ^ v2)
__global__ void k_bitop3_x2(uint32_t *out, const uint32_t *__restrict__ sc) {
    uint32_t z[16], v = threadIdx.x * 2654435761u, v2 = v ^ 0x5bd1e995u;
    uint32_t s0 = sc[0], s1 = sc[1];
    for (int i = 0; i < 16; ++i) z[i] = i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { z[i] = __builtin_amdgcn_bitop3_b32(s0, v, z[i], 0xBE); z[i] = __builtin_amdgcn_bitop3_b32(s1, v2, z[i], 0xBE); }
        v = v * 3 + 1; v2 = v2 * 5 + 1; asm volatile("" : "+v"(v), "+v"(v2));
    }
    uint32_t s = 0; for (int i = 0; i < 16; ++i) s ^= z[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_bcnt(uint32_t *out) {
    uint32_t z[16], v = threadIdx.x * 2654435761u;
    for (int i = 0; i < 16; ++i) z[i] = i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(z[i]) : "v"(v));
        v = v * 3 + 1;
    }
    uint32_t s = 0; for (int i = 0; i < 16; ++i) s ^= z[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_cmp64_addc(uint32_t *out, const uint64_t *__restrict__ sc) {
    uint32_t acc[16]; uint64_t v = threadIdx.x * 0x9E3779B97F4A7C15ull;
    uint64_t s0 = sc[0], s1 = sc[1];
    for (int i = 0; i < 16; ++i) acc[i] = i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += ((i & 1 ? s0 : s1) == v + i);
        v = v * 3 + 1;
    }
    uint32_t s = 0; for (int i = 0; i < 16; ++i) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_cmp64_only(uint32_t *out, const uint64_t *__restrict__ sc) {
    uint64_t v = threadIdx.x * 0x9E3779B97F4A7C15ull;
    uint64_t s0 = sc[0];
    unsigned long long m = 0;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { unsigned long long t; asm volatile("v_cmp_eq_u64 %0, %1, %2" : "=s"(t) : "s"(s0), "v"(v)); m ^= t; }
        v = v * 3 + 1;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)m;
}
__global__ void k_cmp32_addc(uint32_t *out, const uint32_t *__restrict__ sc) {
    uint32_t acc[16]; uint32_t v = threadIdx.x * 2654435761u;
    uint32_t s0 = sc[0], s1 = sc[1];
    for (int i = 0; i < 16; ++i) acc[i] = i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += ((i & 1 ? s0 : s1) == v + i);
        v = v * 3 + 1;
    }
    uint32_t s = 0; for (int i = 0; i < 16; ++i) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// ballot path: v_cmp + s_bcnt1 + s_add (SALU accumulate)
__global__ void k_cmp64_sbcnt(uint32_t *out, const uint64_t *__restrict__ sc) {
    uint64_t v = threadIdx.x * 0x9E3779B97F4A7C15ull;
    uint64_t s0 = sc[0];
    uint32_t acc[8] = {0};
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { unsigned long long t; asm volatile("v_cmp_eq_u64 %0, %1, %2" : "=s"(t) : "s"(s0), "v"(v)); acc[i & 7] += __builtin_popcountll(t); }
        v = v * 3 + 1;
    }
    uint32_t s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__device__ __forceinline__ uint64_t wang64(uint64_t k) {
    k = ~k + (k << 21); k ^= k >> 24; k = k + (k << 3) + (k << 8); k ^= k >> 14; k = k + (k << 2) + (k << 4); k ^= k >> 28; k += k << 31; return k;
}
__global__ void k_wang(uint64_t *out) {
    uint64_t x[4];
    for (int i = 0; i < 4; ++i) x[i] = threadIdx.x * 77 + i;
    for (int it = 0; it < ITER; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = wang64(x[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = x[0] ^ x[1] ^ x[2] ^ x[3];
}


// ---- candidates for a cheaper exact Wang mix (same function, different instruction selection)
__device__ __forceinline__ uint64_t lshl_add(uint64_t a, const int sh, uint64_t b) {   // (a << sh) + b, sh in 0..4
    uint64_t r;
    switch (sh) {
        case 1: asm("v_lshl_add_u64 %0, %1, 1, %2" : "=v"(r) : "v"(a), "v"(b)); break;
        case 2: asm("v_lshl_add_u64 %0, %1, 2, %2" : "=v"(r) : "v"(a), "v"(b)); break;
        case 3: asm("v_lshl_add_u64 %0, %1, 3, %2" : "=v"(r) : "v"(a), "v"(b)); break;
        default: asm("v_lshl_add_u64 %0, %1, 4, %2" : "=v"(r) : "v"(a), "v"(b)); break;
    }
    return r;
}
__device__ __forceinline__ uint64_t wang64_v2(uint64_t k) {
    // k = ~k + (k << 21)  in 32-bit halves
    {
        const uint32_t lo = (uint32_t)k, hi = (uint32_t)(k >> 32);
        const uint32_t slo = lo << 21, shi = __builtin_amdgcn_alignbit(hi, lo, 11);
        k = (((uint64_t)shi << 32) | slo) + ~k;
    }
    k ^= k >> 24;
    {   // k * 265 = k + 8k + 256k
        const uint64_t t = lshl_add(k, 3, k);         // 9k
        const uint64_t a = lshl_add(k, 4, 0);         // 16k
        k = lshl_add(a, 4, t);                        // 256k + 9k
    }
    k ^= k >> 14;
    {   // k * 21 = k + 4k + 16k
        const uint64_t t = lshl_add(k, 2, k);         // 5k
        k = lshl_add(k, 4, t);                        // 16k + 5k
    }
    k ^= k >> 28;
    {   // k += k << 31 in halves
        const uint32_t lo = (uint32_t)k, hi = (uint32_t)(k >> 32);
        const uint32_t slo = lo << 31, shi = __builtin_amdgcn_alignbit(hi, lo, 1);
        k += ((uint64_t)shi << 32) | slo;
    }
    return k;
}
__global__ void k_wang_v2(uint64_t *out) {
    uint64_t x[4];
    for (int i = 0; i < 4; ++i) x[i] = threadIdx.x * 77 + i;
    for (int it = 0; it < ITER; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = wang64_v2(x[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = x[0] ^ x[1] ^ x[2] ^ x[3];
}
__global__ void k_wang_check(uint64_t *out) {      // out[t] = number of mismatches between the two forms
    uint64_t bad = 0, x = threadIdx.x * 0x9E3779B97F4A7C15ull + blockIdx.x;
    for (int it = 0; it < 256; ++it) { bad += wang64(x) != wang64_v2(x); x = wang64(x) + it; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = bad;
}
__global__ void k_lshl_add_u64(uint64_t *out) {
    uint64_t z[16], v = threadIdx.x * 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < 16; ++i) z[i] = i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(z[i]) : "v"(v));
        v = v * 3 + 1;
    }
    uint64_t s = 0; for (int i = 0; i < 16; ++i) s ^= z[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mad_u64_u32(uint64_t *out) {
    uint64_t z[16]; uint32_t v = threadIdx.x * 2654435761u;
    for (int i = 0; i < 16; ++i) z[i] = i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(z[i]) : "v"(v), "v"((uint32_t)z[i]) : "vcc");
        v = v * 3 + 1;
    }
    uint64_t s = 0; for (int i = 0; i < 16; ++i) s ^= z[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_lshrrev_b64(uint64_t *out) {
    uint64_t z[16], v = threadIdx.x * 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < 16; ++i) z[i] = v + i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(z[i]));
    }
    uint64_t s = 0; for (int i = 0; i < 16; ++i) s ^= z[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F> float time_kernel(F launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); for (int r = 0; r < 5; ++r) launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", p.gcnArchName, cus, p.clockRate);
    void *out; CHECK(hipMalloc(&out, 64 << 20));
    uint64_t h[8] = {1, 2, 3, 4, 5, 6, 7, 8}; void *sc; CHECK(hipMalloc(&sc, 64)); CHECK(hipMemcpy(sc, h, 64, hipMemcpyHostToDevice));
    {
        k_wang_check<<<64, 256>>>((uint64_t *)out);
        std::vector<uint64_t> h(64 * 256);
        CHECK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
        uint64_t bad = 0; for (auto x : h) bad += x;
        printf("wang64_v2 vs wang64: %llu mismatches over %zu values\n", (unsigned long long)bad, h.size() * 256);
    }
    for (int wps : {2, 8}) {            // waves per SIMD
        const int blocks = cus * wps, threads = 256;   // 4 waves per block -> wps blocks per CU
        const double winst = (double)blocks * 4 * ITER * 16;   // wave-instructions of the measured op
        auto rep = [&](const char *name, float ms, double ops_per = 1.0) {
            const double per_simd_per_s = winst * ops_per / (cus * 4) / (ms * 1e-3);
            printf("  wps=%d %-22s %8.3f ms  %.3f cycles/wave-inst @2.4GHz  (%.2f Tlane-op/s chip)\n", wps, name, ms,
                   2.4e9 / per_simd_per_s, winst * ops_per * 64 / (ms * 1e-3) / 1e12);
        };
        rep("v_fma_f32", time_kernel([&] { k_fma<<<blocks, threads>>>((float *)out, 1.0001f, 0.5f); }));
        rep("v_bitop3_b32(s,v,v)", time_kernel([&] { k_bitop3<<<blocks, threads>>>((uint32_t *)out, (uint32_t *)sc); }));
        rep("v_xor+v_or", time_kernel([&] { k_xor_or<<<blocks, threads>>>((uint32_t *)out, (uint32_t *)sc); }), 2.0);
        rep("2 planes: 2xor+or3 (per plane)", time_kernel([&] { k_xor2_or3<<<blocks, threads>>>((uint32_t *)out, (uint32_t *)sc); }), 2.0);
        rep("2 planes: 2 bitop3 (per plane)", time_kernel([&] { k_bitop3_x2<<<blocks, threads>>>((uint32_t *)out, (uint32_t *)sc); }), 2.0);
        rep("v_bcnt_u32_b32", time_kernel([&] { k_bcnt<<<blocks, threads>>>((uint32_t *)out); }));
        rep("v_cmp_eq_u64 only", time_kernel([&] { k_cmp64_only<<<blocks, threads>>>((uint32_t *)out, (uint64_t *)sc); }));
        rep("cmp_eq_u64+addc (2)", time_kernel([&] { k_cmp64_addc<<<blocks, threads>>>((uint32_t *)out, (uint64_t *)sc); }), 2.0);
        rep("cmp_eq_u32+addc (2)", time_kernel([&] { k_cmp32_addc<<<blocks, threads>>>((uint32_t *)out, (uint32_t *)sc); }), 2.0);
        rep("cmp_eq_u64+s_bcnt1", time_kernel([&] { k_cmp64_sbcnt<<<blocks, threads>>>((uint32_t *)out, (uint64_t *)sc); }));
        rep("v_lshl_add_u64", time_kernel([&] { k_lshl_add_u64<<<blocks, threads>>>((uint64_t *)out); }));
        rep("v_mad_u64_u32", time_kernel([&] { k_mad_u64_u32<<<blocks, threads>>>((uint64_t *)out); }));
        rep("v_lshrrev_b64", time_kernel([&] { k_lshrrev_b64<<<blocks, threads>>>((uint64_t *)out); }));
        {
            const float ms = time_kernel([&] { k_wang_v2<<<blocks, threads>>>((uint64_t *)out); });
            const double hashes = (double)blocks * 256 * ITER * 4;
            printf("  wps=%d %-22s %8.3f ms  %.3e wang64/s chip  (%.1f lane-cycles per hash @2.4GHz)\n", wps, "wang64_v2 (lshl_add)", ms,
                   hashes / (ms * 1e-3), (double)cus * 4 * 32 * 2.4e9 / (hashes / (ms * 1e-3)));
        }
        {   // wang: 4 chains x ITER hashes per lane
            const float ms = time_kernel([&] { k_wang<<<blocks, threads>>>((uint64_t *)out); });
            const double hashes = (double)blocks * 256 * ITER * 4;
            printf("  wps=%d %-22s %8.3f ms  %.3e wang64/s chip  (%.1f lane-cycles per hash @2.4GHz)\n", wps, "wang64", ms,
                   hashes / (ms * 1e-3), (double)cus * 4 * 32 * 2.4e9 / (hashes / (ms * 1e-3)));
        }
    }
    return 0;
}

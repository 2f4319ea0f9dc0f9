#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ __launch_bounds__(256) void k(uint64_t *out, int iters) {
    __shared__ uint64_t tab[2048];
    __shared__ uint32_t cnt[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) { tab[i] = ~0ull; cnt[i] = 0; }
    __syncthreads();
    uint64_t x = (blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull + 1;
    uint64_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        const uint32_t s = (uint32_t)(x >> 53);
        if (MODE == 0) acc += tab[s];                                                                  // ds_read_b64
        if (MODE == 1) acc += atomicCAS((unsigned long long *)&tab[s], ~0ull, (unsigned long long)x);   // ds_cmpst_rtn_b64
        if (MODE == 2) atomicAdd(&cnt[s], 1u);                                                          // ds_add_u32 (no return)
        if (MODE == 3) acc += atomicAdd(&cnt[s], 1u);                                                   // ds_add_rtn_u32
        if (MODE == 4) acc += atomicCAS(&cnt[s], 0u, (uint32_t)x);                                      // ds_cmpst_rtn_b32
    }
    if (acc == 42) out[0] = acc + cnt[5];
}
int main() {
    uint64_t *d; hipMalloc(&d, 8);
    const int blocks = 256 * 5, iters = 4096;
    const char *names[] = {"ds_read_b64", "ds_cmpst_rtn_b64", "ds_add_u32", "ds_add_rtn_u32", "ds_cmpst_rtn_b32"};
    for (int m = 0; m < 5; ++m) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            if (m == 0) k<0><<<blocks, 256>>>(d, iters); if (m == 1) k<1><<<blocks, 256>>>(d, iters);
            if (m == 2) k<2><<<blocks, 256>>>(d, iters); if (m == 3) k<3><<<blocks, 256>>>(d, iters);
            if (m == 4) k<4><<<blocks, 256>>>(d, iters);
            hipEventRecord(b); hipEventSynchronize(b);
        }
        float ms; hipEventElapsedTime(&ms, a, b);
        const double ops = (double)blocks * 256 * iters;
        printf("%-18s %8.3f ms  %.3e lane-ops/s chip  %.2f lane-ops/cycle/CU @2.4GHz\n", names[m], ms, ops / ms * 1e3, ops / ms * 1e3 / 256 / 2.4e9);
    }
    return 0;
}

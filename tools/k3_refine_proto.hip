// k3_refine_proto.hip -- VERDICT r3 #6: a measurement, not an estimate, of the one NEW kernel the "4-byte keys through the
// tile-sorting scatter + a 2-bit refine" variant of the K3 chain needs (profiles/r03_k3_sweep.txt rated it best on paper).
// The compact key path (D2G_K3_COMPACT=1) already exists and is measured in every bench run: k3c_hist 0.8 + k3c_scan 0.5 +
// k3c_scatter 6.9 + k3_split 6.5 + main/survivors/verify = 27 ms per call of 250 x 5 Mbp.  The variant replaces k3_split (which
// reads every 4096-key bucket twice: count its four sub-ranges, then place) by a refine pass whose offsets come from the histogram
// pass (one read, one LDS atomic per key).  This program times exactly that pass on the production layout -- 250 genomes x 1024
// buckets of ~4883 four-byte keys, 1.25e9 keys, one workgroup per bucket -- next to a split-style two-read pass, so that the chain
// with the variant can be written down from measured parts:  chain(variant) = 27.1 - split + refine.
//   build + run (GPU box):  hipcc -O3 --offload-arch=gfx950 tools/k3_refine_proto.hip -o /tmp/k3_refine_proto && /tmp/k3_refine_proto
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

__device__ __forceinline__ uint32_t sub_of(uint32_t key) { return (key * 0x85EBCA6Bu) >> 30; }     // two more bucket bits from the stored word

__global__ void fill_kernel(uint32_t *keys, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t z = i * 0x9E3779B97F4A7C15ull + 0x1234567;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        keys[i] = (uint32_t)(z >> 16);
    }
}

// the histogram pass's extra output in the variant: keys per (bucket, sub-range)
__global__ __launch_bounds__(256) void count_kernel(const uint32_t *keys, const uint64_t *boff, uint32_t *subcnt) {
    __shared__ uint32_t c[4];
    if (threadIdx.x < 4) c[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t a = boff[blockIdx.x], b = boff[blockIdx.x + 1];
    for (uint64_t i = a + threadIdx.x; i < b; i += 256) atomicAdd(&c[sub_of(keys[i])], 1u);
    __syncthreads();
    if (threadIdx.x < 4) subcnt[blockIdx.x * 4 + threadIdx.x] = c[threadIdx.x];
}

// (B) the refine pass of the variant: offsets known, ONE read of the bucket, one LDS atomic per key, four write fronts
__global__ __launch_bounds__(256) void refine_kernel(const uint32_t *keys, const uint64_t *boff, const uint32_t *subcnt, uint32_t *out) {
    __shared__ uint32_t cur[4];
    if (threadIdx.x == 0) { uint32_t r = 0; for (int s = 0; s < 4; ++s) { cur[s] = r; r += subcnt[blockIdx.x * 4 + s]; } }
    __syncthreads();
    const uint64_t a = boff[blockIdx.x], b = boff[blockIdx.x + 1];
    for (uint64_t i0 = a; i0 < b; i0 += 1024) {                 // four keys per thread in flight
        uint32_t k[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) { const uint64_t i = i0 + x * 256 + threadIdx.x; k[x] = i < b ? keys[i] : 0u; }
#pragma unroll
        for (int x = 0; x < 4; ++x) { const uint64_t i = i0 + x * 256 + threadIdx.x; if (i < b) out[a + atomicAdd(&cur[sub_of(k[x])], 1u)] = k[x]; }
    }
}

// (A) split-style: the sub-range sizes are NOT known -- count (first read), then place (second read, L2-hot)
__global__ __launch_bounds__(256) void split_kernel(const uint32_t *keys, const uint64_t *boff, uint32_t *out) {
    __shared__ uint32_t c[4], cur[4];
    if (threadIdx.x < 4) c[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t a = boff[blockIdx.x], b = boff[blockIdx.x + 1];
    for (uint64_t i = a + threadIdx.x; i < b; i += 256) atomicAdd(&c[sub_of(keys[i])], 1u);
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t r = 0; for (int s = 0; s < 4; ++s) { cur[s] = r; r += c[s]; } }
    __syncthreads();
    for (uint64_t i = a + threadIdx.x; i < b; i += 256) { const uint32_t k = keys[i]; out[a + atomicAdd(&cur[sub_of(k)], 1u)] = k; }
}

int main() {
    const int genomes = 250, B = 1024;
    const uint64_t per_genome = 5000000 - 20;
    const size_t nb = (size_t)genomes * B, n = (size_t)genomes * per_genome;
    std::vector<uint64_t> boff(nb + 1);
    for (size_t g = 0; g < (size_t)genomes; ++g)
        for (int b = 0; b <= B; ++b) if (b < B || g + 1 == (size_t)genomes) boff[g * B + b] = g * per_genome + per_genome * b / B;
    uint32_t *keys, *out, *subcnt; uint64_t *dboff;
    CK(hipMalloc(&keys, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&subcnt, nb * 16)); CK(hipMalloc(&dboff, (nb + 1) * 8));
    CK(hipMemcpy(dboff, boff.data(), (nb + 1) * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, keys, n);
    hipLaunchKernelGGL(count_kernel, dim3((unsigned)nb), dim3(256), 0, 0, keys, dboff, subcnt);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char *name, auto launch) {
        launch(); CK(hipDeviceSynchronize());
        float best = 1e9f, sum = 0;
        for (int r = 0; r < 5; ++r) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; sum += ms; }
        std::printf("%-58s %7.3f ms (best of 5; mean %.3f)  = %.2f TB/s of key traffic (read + write %zu MB each)\n", name, best, sum / 5, 2.0 * n * 4 / (best * 1e-3) / 1e12, n * 4 >> 20);
    };
    std::printf("K3 compact-key layout: %d genomes x %d buckets, %zu four-byte keys (one call of 250 x 5 Mbp, k = 21)\n", genomes, B, n);
    time("(B) 2-bit refine, offsets from the histogram (one read)", [&] { hipLaunchKernelGGL(refine_kernel, dim3((unsigned)nb), dim3(256), 0, 0, keys, dboff, subcnt, out); });
    time("(A) split-style, sub-range sizes counted first (two reads)", [&] { hipLaunchKernelGGL(split_kernel, dim3((unsigned)nb), dim3(256), 0, 0, keys, dboff, out); });
    time("    the extra counting of the histogram pass, if done alone", [&] { hipLaunchKernelGGL(count_kernel, dim3((unsigned)nb), dim3(256), 0, 0, keys, dboff, subcnt); });
    // a check that (A) and (B) produce a permutation of each bucket with the sub-ranges in order
    std::vector<uint32_t> h(boff[1]);
    CK(hipMemcpy(h.data(), out, boff[1] * 4, hipMemcpyDeviceToHost));
    uint32_t prev = 0; bool ok = true;
    for (uint32_t k : h) { const uint32_t s = (k * 0x85EBCA6Bu) >> 30; if (s < prev) ok = false; prev = s; }
    std::printf("bucket 0 is ordered by sub-range after the pass: %s\n", ok ? "yes" : "NO");
    return ok ? 0 : 1;
}

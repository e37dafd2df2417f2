// Micro-benchmark: ceiling for random 8-byte gathers over a 1 GiB table on MI355X (what K1 is bound by).
// hipcc --offload-arch=gfx950 -O3 -o ubench_gather scripts/ubench_gather.hip && ./ubench_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint64_t u64;
template <int ILP, bool NT>
__global__ void gather(const u64 * __restrict__ t, u64 mask, u64 n, u64 * out)
{
    u64 tid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 acc = 0;
    for (u64 i = tid; i < n; i += (u64)gridDim.x * blockDim.x * ILP)
    {
        u64 v[ILP];
#pragma unroll
        for (int k = 0; k < ILP; ++k)
        {
            u64 h = (i + (u64)k * gridDim.x * blockDim.x) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
            const u64 * p = t + (h & mask);
            v[k] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int k = 0; k < ILP; ++k) acc &= v[k] | 1;
        acc += v[0];
    }
    if (acc == 0x1234567) out[0] = acc;
}
template <int ILP, bool NT> void run(const u64 * t, u64 words, u64 * out, int blocks, int threads)
{
    const u64 n = 512ull << 20;   // gathers
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((gather<ILP, NT>), dim3(blocks), dim3(threads), 0, 0, t, words - 1, n, out);
    hipEventRecord(a);
    hipLaunchKernelGGL((gather<ILP, NT>), dim3(blocks), dim3(threads), 0, 0, t, words - 1, n, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("ILP %d nt %d blocks %d x %d: %.2f ms, %.1f G gathers/s, %.2f TB/s of 64-B lines\n", ILP, (int)NT, blocks, threads, ms, n / ms / 1e6, n * 64.0 / ms / 1e9);
}
int main()
{
    const u64 words = 1ull << 27;  // 1 GiB
    u64 * t; u64 * out;
    hipMalloc(&t, words * 8); hipMalloc(&out, 8);
    hipMemset(t, 0x5a, words * 8);
    run<1, false>(t, words, out, 256 * 8, 256);
    run<3, false>(t, words, out, 256 * 8, 256);
    run<3, false>(t, words, out, 256 * 16, 512);
    run<8, false>(t, words, out, 256 * 8, 256);
    run<8, true>(t, words, out, 256 * 8, 256);
    run<16, false>(t, words, out, 256 * 8, 256);
    run<3, true>(t, words, out, 256 * 16, 512);
    return 0;
}

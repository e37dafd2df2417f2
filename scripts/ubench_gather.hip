// Micro-benchmark: the chip's ceiling for random gathers of 8 .. 128 bytes over tables of 64 MiB .. 64 GiB on MI355X — what K1 (the
// IBF sweep) is bound by.  A request of `width` bytes is served by G adjacent lanes, each loading V x 8 bytes of one aligned
// group (the production kernels use V = 1: one 64-bit word per lane), every request at an independent pseudo-random position.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_gather scripts/ubench_gather.hip && scripts/ubench_gather [max_gib [min_mib]] > profiles/<round>_gather_sweep.jsonl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef uint64_t u64;
typedef uint32_t u32;

template <int G, int V, int ILP>
__global__ __launch_bounds__(256) void gather(const u64 * __restrict__ t, u64 groupMask, u64 nReq, u64 * out)
{
    const u64 tid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 req0 = tid / G, nLanesReq = ((u64)gridDim.x * blockDim.x) / G;
    const u32 sub = (u32)(tid % G);
    u64 acc = 0;
    for (u64 i = req0; i < nReq; i += nLanesReq * ILP)
    {
        u64 v[ILP][V];
#pragma unroll
        for (int k = 0; k < ILP; ++k)
        {
            u64 h = (i + (u64)k * nLanesReq) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
            const u64 * p = t + (h & groupMask) * (u64)(G * V) + (u64)sub * V;
#pragma unroll
            for (int e = 0; e < V; ++e) v[k][e] = p[e];
        }
#pragma unroll
        for (int k = 0; k < ILP; ++k)
#pragma unroll
            for (int e = 0; e < V; ++e) acc += v[k][e] | 1;
    }
    if (acc == 0x1234567) out[0] = acc;
}

template <int G, int V, int ILP> void run(const u64 * t, u64 tableBytes, u64 * out)
{
    const u64 nReq = 256ull << 20;
    const u64 groups = tableBytes / (8ull * G * V);            // power of two
    const int blocks = 256 * 8, threads = 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((gather<G, V, ILP>), dim3(blocks), dim3(threads), 0, 0, t, groups - 1, nReq, out);
    hipEventRecord(a);
    hipLaunchKernelGGL((gather<G, V, ILP>), dim3(blocks), dim3(threads), 0, 0, t, groups - 1, nReq, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const int width = 8 * G * V;
    printf("{\"table_mib\": %llu, \"width_bytes\": %d, \"lanes_per_request\": %d, \"bytes_per_lane\": %d, \"ilp\": %d, \"ms\": %.3f, \"G_requests_per_s\": %.2f, "
           "\"TB_per_s_requested\": %.3f, \"TB_per_s_of_64B_lines\": %.3f}\n",
           (unsigned long long)(tableBytes >> 20), width, G, 8 * V, ILP, ms, nReq / ms / 1e6, nReq * (double)width / ms / 1e9,
           nReq * (double)(width < 64 ? 64 : width) / ms / 1e9);
    fflush(stdout);
    hipEventDestroy(a); hipEventDestroy(b);
}

// What is the ceiling made of?  Gathers of 8 B with S streamed (coalesced, 16 B per lane) bytes per gather read next to them from a
// second buffer: if a narrow gather only costs a request slot, the stream rides along for free until HBM is full; if every gather
// moves a whole 128-B line, stream and gathers share the same bytes per second.
template <int SPG>   // streamed 16-B loads per 4 gathers per lane
__global__ __launch_bounds__(256) void gatherPlusStream(const u64 * __restrict__ t, u64 groupMask, u64 nReq, const uint4 * __restrict__ stream, u64 streamMask, u64 * out)
{
    const u64 tid = (u64)blockIdx.x * blockDim.x + threadIdx.x, nT = (u64)gridDim.x * blockDim.x;
    u64 acc = 0; u64 sp = tid;
    for (u64 i = tid; i < nReq; i += nT * 4)
    {
        u64 v[4]; uint4 w[SPG > 0 ? SPG : 1];
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
            u64 h = (i + (u64)k * nT) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
            v[k] = t[h & groupMask];
        }
#pragma unroll
        for (int k = 0; k < SPG; ++k) { w[k] = stream[sp & streamMask]; sp += nT; }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc += v[k] | 1;
#pragma unroll
        for (int k = 0; k < SPG; ++k) acc += w[k].x ^ w[k].w;
    }
    if (acc == 0x1234567) out[0] = acc;
}

template <int SPG> void runMix(const u64 * t, u64 tableBytes, const uint4 * stream, u64 streamBytes, u64 * out, bool gathers)
{
    const u64 nReq = gathers ? 256ull << 20 : 256ull << 20;
    const int blocks = 256 * 8, threads = 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep)
    {
        if (rep == 1) hipEventRecord(a);
        hipLaunchKernelGGL((gatherPlusStream<SPG>), dim3(blocks), dim3(threads), 0, 0, t, gathers ? tableBytes / 8 - 1 : 0ull, nReq, stream, streamBytes / 16 - 1, out);
    }
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double streamed = (double)nReq / 4 * SPG * 16;
    printf("{\"mix\": true, \"table_mib\": %llu, \"random_gathers\": %s, \"streamed_bytes_per_gather\": %d, \"ms\": %.3f, \"G_gathers_per_s\": %.2f, \"stream_TB_per_s\": %.3f}\n",
           (unsigned long long)(tableBytes >> 20), gathers ? "true" : "false (all gathers hit one word)", SPG * 4, ms, nReq / ms / 1e6, streamed / ms / 1e9);
    fflush(stdout);
    hipEventDestroy(a); hipEventDestroy(b);
}

int main(int argc, char ** argv)
{
    const u64 maxGib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 64;
    const u64 minMib = argc > 2 ? strtoull(argv[2], nullptr, 10) : 64;          // smaller tables: L2- / Infinity-Cache-resident gathers
    u64 * t; u64 * out;
    if (hipMalloc(&t, maxGib << 30) != hipSuccess) { fprintf(stderr, "cannot allocate %llu GiB\n", (unsigned long long)maxGib); return 1; }
    hipMalloc(&out, 8);
    hipMemset(t, 0x5a, maxGib << 30);
    hipDeviceSynchronize();
    if (argc > 3)
    {   // mix mode: <max_gib> <min_mib> mix — gathers over the first GiB, the stream over the second half of the allocation
        const u64 tb = 1ull << 30; const uint4 * stream = (const uint4 *)(t + (maxGib << 30) / 16); const u64 sb = (maxGib << 30) / 2;
        runMix<0>(t, tb, stream, sb, out, true);
        runMix<1>(t, tb, stream, sb, out, true);
        runMix<2>(t, tb, stream, sb, out, true);
        runMix<4>(t, tb, stream, sb, out, true);
        runMix<8>(t, tb, stream, sb, out, true);
        runMix<16>(t, tb, stream, sb, out, true);
        runMix<4>(t, tb, stream, sb, out, false);
        runMix<8>(t, tb, stream, sb, out, false);
        runMix<16>(t, tb, stream, sb, out, false);
        run<32, 1, 4>(t, tb, out);       // 256 B: two lines per request
        run<16, 2, 4>(t, tb, out);
        return 0;
    }
    for (u64 mib = minMib; mib <= (maxGib << 10); mib *= 4)
    {
        const u64 bytes = mib << 20;
        run<1, 1, 4>(t, bytes, out);     //   8 B: one lane, one word (W = 1: 64 bins)
        run<1, 2, 4>(t, bytes, out);     //  16 B: one lane, dwordx4
        run<2, 1, 4>(t, bytes, out);     //  16 B: two lanes
        run<4, 1, 4>(t, bytes, out);     //  32 B: four lanes (W = 4: 256 bins)
        run<2, 2, 4>(t, bytes, out);     //  32 B: two lanes x 16 B
        run<8, 1, 4>(t, bytes, out);     //  64 B
        run<4, 2, 4>(t, bytes, out);     //  64 B: four lanes x 16 B
        run<16, 1, 4>(t, bytes, out);    // 128 B: sixteen lanes (W = 16: 1024 bins)
        run<8, 2, 4>(t, bytes, out);     // 128 B: eight lanes x 16 B
        run<16, 1, 8>(t, bytes, out);    // 128 B, more requests in flight
        run<4, 1, 8>(t, bytes, out);
        run<1, 1, 8>(t, bytes, out);
    }
    return 0;
}

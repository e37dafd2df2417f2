// Suffix sorting on the device for dream_yara_indexer: prefix doubling over packed 64-bit keys, every round one
// hipcub radix sort of (key, suffix) pairs over the whole text.  It stands where the reference lets SeqAn build the suffix array of
// the reversed, sentinel-terminated contigs (reference src/d_indexer.cpp:278-333: the TopDown iterator instantiation triggers
// index construction).  Every contig ends with its own sentinel symbol, so no two suffixes compare equal and the suffix array is
// unique: whatever sorter builds it, the 12 index files come out byte-identical (the gold bins pin that).
//   round 0   key = the first q symbols (q = 63 / bits per symbol: 21 for a single-contig bin), rank = first position of the key's group
//   round h   key = (rank[i] << 32) | (rank[i + h] + 1), h = q, 2q, 4q, ... until every rank is unique
// A 4 Mbp genome without long repeats is finished after round 0 or 1; a 32 kbp homopolymer run takes ~12 rounds.
#pragma once
#include <hip/hip_runtime.h>
#include "dy_types.h"

namespace dy {

__global__ void k_sa_first_keys(const u32 * __restrict__ text, u32 n, u32 bits, u32 q, u64 * __restrict__ keys, u32 * __restrict__ idx)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 k = 0;
    for (u32 j = 0; j < q; ++j) k = (k << bits) | (i + j < n ? (u64)text[i + j] + 1 : 0);       // symbols shifted by +1; 0 pads beyond the end
    keys[i] = k; idx[i] = i;
}

// groupStart[i] = i where a new key starts, else 0 (an inclusive max-scan then gives every position the first index of its group)
__global__ void k_sa_group_flags(const u64 * __restrict__ sortedKeys, u32 n, u32 * __restrict__ flags, u32 * __restrict__ ties)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool newGroup = i == 0 || sortedKeys[i] != sortedKeys[i - 1];
    flags[i] = newGroup ? i : 0;
    if (!newGroup) atomicAdd(ties, 1u);
}

__global__ void k_sa_scatter_ranks(const u32 * __restrict__ sa, const u32 * __restrict__ groupStart, u32 n, u32 * __restrict__ rank)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rank[sa[i]] = groupStart[i];
}

__global__ void k_sa_doubled_keys(const u32 * __restrict__ rank, u32 n, u32 h, u64 * __restrict__ keys, u32 * __restrict__ idx)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 second = (u64)i + h < n ? (u64)rank[i + h] + 1 : 0;
    keys[i] = ((u64)rank[i] << 32) | second;
    idx[i] = i;
}

struct MaxU32 { __host__ __device__ __forceinline__ u32 operator()(u32 a, u32 b) const { return a > b ? a : b; } };

} // namespace dy

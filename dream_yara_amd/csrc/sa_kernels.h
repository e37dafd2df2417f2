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

// ---- from the suffix array to the index files' arrays (reference src/d_indexer.cpp:278-333 builds them inside SeqAn: BWT, the two-level
// rank dictionary of lf.drv, the sentinel rows of lf.drp, the sampled suffix array of sa.ind / sa.val; layouts: SURVEY.md Appendix A) ----------
// text: contig c reversed (Ns randomised) as symbols nContigs + base, followed by its sentinel symbol c; starts[c] = first position of contig c.
__device__ __forceinline__ u32 contigOfPos(const u32 * __restrict__ starts, u32 nContigs, u32 pos)
{
    u32 lo = 0, hi = nContigs;                      // last c with starts[c] <= pos
    while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (starts[mid] <= pos) lo = mid; else hi = mid; }
    return lo;
}

// per BWT row: its symbol (0..3, 4 = the row of a contig's first suffix: "sentinel row") and whether its suffix is sampled (offset in the
// contig divisible by 10, never the sentinel's own suffix); occ[s] counts the symbols
__global__ __launch_bounds__(256) void k_fm_rows(const u32 * __restrict__ sa, const u32 * __restrict__ text, u32 n, u32 nContigs, const u32 * __restrict__ starts,
                                                 u8 * __restrict__ sym, u8 * __restrict__ sampled, u32 * __restrict__ occ)
{
    __shared__ u32 local[5];
    if (threadIdx.x < 5) local[threadIdx.x] = 0;
    __syncthreads();
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n)
    {
        const u32 pos = sa[r], c = contigOfPos(starts, nContigs, pos);
        const u32 s = pos == starts[c] ? 4u : text[pos - 1] - nContigs;
        sym[r] = (u8)s;
        sampled[r] = (pos + 1 != starts[c + 1] && (pos - starts[c]) % 10 == 0) ? 1 : 0;
        atomicAdd(&local[s], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 5 && local[threadIdx.x]) atomicAdd(&occ[threadIdx.x], local[threadIdx.x]);
}

struct Count4 { u32 c[4]; };
struct Count4Sum { __host__ __device__ Count4 operator()(const Count4 & a, const Count4 & b) const { Count4 r; for (int i = 0; i < 4; ++i) r.c[i] = a.c[i] + b.c[i]; return r; } };

// symbols per block of 64 rows (sentinel rows count as the substitute symbol drs); entry nBlocks = 0
__global__ void k_fm_block_counts(const u8 * __restrict__ sym, u32 n, u32 drs, u32 nBlocks, Count4 * __restrict__ counts)
{
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nBlocks) return;
    Count4 c; c.c[0] = c.c[1] = c.c[2] = c.c[3] = 0;
    if (b < nBlocks) for (u32 t = 0; t < 64 && (u64)b * 64 + t < n; ++t) { const u32 s = sym[(size_t)b * 64 + t]; c.c[s == 4 ? drs : s]++; }
    counts[b] = c;
}

// lf.drv: per 64 rows [2 x u64 of 2-bit symbols, row j in bits (31 - j % 32) * 2 of word j / 32][4 x u32 symbols before the block]
__global__ void k_fm_drv(const u8 * __restrict__ sym, u32 n, u32 drs, u32 nBlocks, const Count4 * __restrict__ before, u8 * __restrict__ drv)
{
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nBlocks) return;
    u64 w[2] = {0, 0};
    for (u32 t = 0; t < 64 && (u64)b * 64 + t < n; ++t)
    {
        u32 s = sym[(size_t)b * 64 + t]; if (s == 4) s = drs;
        w[t >> 5] |= (u64)s << ((31 - (t & 31)) * 2);
    }
    u64 * out = (u64 *)(drv + (size_t)b * 32);
    out[0] = w[0]; out[1] = w[1];
    const Count4 c = before[b];
    u32 * o32 = (u32 *)(drv + (size_t)b * 32 + 16);
    o32[0] = c.c[0]; o32[1] = c.c[1]; o32[2] = c.c[2]; o32[3] = c.c[3];
}

// bit words over the rows (row j of an entry at bit 31 - j) and their popcounts, for the sampled rows and for the sentinel rows; entry nEntries = 0
__global__ void k_fm_bit_words(const u8 * __restrict__ sym, const u8 * __restrict__ sampled, u32 n, u32 nEntries, u32 * __restrict__ sampBits, u32 * __restrict__ sampCnt,
                               u32 * __restrict__ sentBits, u32 * __restrict__ sentCnt)
{
    const u32 e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e > nEntries) return;
    u32 a = 0, z = 0;
    if (e < nEntries) for (u32 t = 0; t < 32 && (u64)e * 32 + t < n; ++t) { const size_t r = (size_t)e * 32 + t; if (sampled[r]) a |= 1u << (31 - t); if (sym[r] == 4) z |= 1u << (31 - t); }
    sampBits[e] = a; sampCnt[e] = (u32)__popc(a); sentBits[e] = z; sentCnt[e] = (u32)__popc(z);
}

// sa.ind: [u32 bits][u32 samples before]; sa.val: packed (contig id u8 / u16, u32 offset) per sample in row order; lf.drp rows in ascending order
__global__ void k_fm_samples(const u32 * __restrict__ sa, u32 n, u32 nContigs, const u32 * __restrict__ starts, u32 nEntries, const u32 * __restrict__ sampBits,
                             const u32 * __restrict__ sampBefore, const u32 * __restrict__ sentBits, const u32 * __restrict__ sentBefore,
                             u8 * __restrict__ ind, u8 * __restrict__ val, u32 idBytes, u32 * __restrict__ drp)
{
    const u32 e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nEntries) return;
    u32 bits = sampBits[e], at = sampBefore[e];
    u32 * io = (u32 *)(ind + (size_t)e * 8);
    io[0] = bits; io[1] = at;
    const u32 rec = idBytes + 4;
    for (u32 t = 0; t < 32; ++t)
        if (bits & (1u << (31 - t)))
        {
            const u32 pos = sa[(size_t)e * 32 + t], c = contigOfPos(starts, nContigs, pos), off = pos - starts[c];
            u8 * o = val + (size_t)at * rec;
            o[0] = (u8)c; if (idBytes == 2) o[1] = (u8)(c >> 8);
            o[idBytes] = (u8)off; o[idBytes + 1] = (u8)(off >> 8); o[idBytes + 2] = (u8)(off >> 16); o[idBytes + 3] = (u8)(off >> 24);
            ++at;
        }
    u32 zb = sentBits[e], zat = sentBefore[e];
    for (u32 t = 0; t < 32; ++t) if (zb & (1u << (31 - t))) drp[zat++] = e * 32 + t;
}

} // namespace dy

// Bins sharded over ranks: device-side packing for the two exchanges of SURVEY.md §8(e).
//   C1  reads -> owners of their selected bins   (loadFilteredReads, reference src/d_mapper.h:394-443, across ranks)
//   C2  raw matches -> home rank of the read     (copyMatches, reference src/d_mapper.h:127-155, across ranks)
// The library only packs and unpacks; the all-to-all itself is the caller's (RCCL over xGMI in bench.py / dist.py).
#pragma once
#include "dy_types.h"
#include <hip/hip_runtime.h>

namespace dy {

// entry t = d * n + i : is read i wanted by rank d, and how many bases does it carry there
__global__ void k_shard_flags(const u64 * __restrict__ masks, const u64 * __restrict__ offs, u32 n, u32 W, const u64 * __restrict__ owner, u32 nRanks,
                              u32 * __restrict__ cnt, u64 * __restrict__ len)
{
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 total = (u64)nRanks * n;
    if (t > total) return;
    if (t == total) { cnt[t] = 0; len[t] = 0; return; }
    const u32 d = (u32)(t / n), i = (u32)(t % n);
    u64 any = 0;
    for (u32 w = 0; w < W; ++w) any |= masks[(size_t)i * W + w] & owner[(size_t)d * W + w];
    cnt[t] = any ? 1u : 0u;
    len[t] = any ? offs[i + 1] - offs[i] : 0;
}

__global__ void k_shard_totals(const u32 * __restrict__ cntOff, const u64 * __restrict__ lenOff, u32 n, u32 nRanks, u64 * __restrict__ out)
{
    const u32 d = threadIdx.x;
    if (d > nRanks) return;
    out[d] = cntOff[(size_t)d * n];
    out[nRanks + 1 + d] = lenOff[(size_t)d * n];
}

// One lane per entry for the scalars; the 64 entries of a wavefront then copy their bases cooperatively.
__global__ void k_shard_pack(const u64 * __restrict__ masks, const u64 * __restrict__ offs, const u8 * __restrict__ seqs, u32 n, u32 W,
                             const u64 * __restrict__ owner, u32 nRanks, const u32 * __restrict__ cnt, const u32 * __restrict__ cntOff,
                             const u64 * __restrict__ lenOff, u32 * __restrict__ outIds, u32 * __restrict__ outLens, u64 * __restrict__ outMasks,
                             u8 * __restrict__ outSeqs)
{
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 total = (u64)nRanks * n;
    const u32 lane = threadIdx.x & 63;
    bool sel = false; u32 i = 0; u64 src = 0, dst = 0; u32 L = 0;
    if (t < total && cnt[t])
    {
        sel = true;
        const u32 d = (u32)(t / n); i = (u32)(t % n);
        const u32 pos = cntOff[t];
        src = offs[i]; L = (u32)(offs[i + 1] - src); dst = lenOff[t];
        outIds[pos] = i; outLens[pos] = L;
        for (u32 w = 0; w < W; ++w) outMasks[(size_t)pos * W + w] = masks[(size_t)i * W + w] & owner[(size_t)d * W + w];
    }
    u64 todo = __ballot(sel);
    while (todo)
    {
        const int l = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const u64 s = __shfl(src, l, 64), dd = __shfl(dst, l, 64);
        const u32 len = __shfl(L, l, 64);
        for (u32 j = lane; j < len; j += 64) outSeqs[dd + j] = seqs[s + j];
    }
}

__global__ void k_widen_lens(const u32 * __restrict__ lens, u32 n, u64 * __restrict__ out)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t <= n) out[t] = t < n ? lens[t] : 0;
}

// stats[0] = sum, stats[1] = max, stats[2] = min of the lengths (stats pre-set to {0, 0, ~0})
__global__ void k_lens_stats(const u32 * __restrict__ lens, u32 n, unsigned long long * __restrict__ stats)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    u64 s = 0; u32 mx = 0, mn = 0xFFFFFFFFu;
    for (u32 i = t; i < n; i += gridDim.x * blockDim.x) { const u32 l = lens[i]; s += l; mx = max(mx, l); mn = min(mn, l); }
    for (int o = 32; o >= 1; o >>= 1)
    {
        s += ((u64)__shfl_xor((u32)(s >> 32), o, 64) << 32) | __shfl_xor((u32)s, o, 64);
        mx = max(mx, (u32)__shfl_xor(mx, o, 64)); mn = min(mn, (u32)__shfl_xor(mn, o, 64));
    }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&stats[0], s); atomicMax(&stats[1], (unsigned long long)mx); atomicMin(&stats[2], (unsigned long long)mn); }
}

__global__ void k_and_masks(u64 * __restrict__ masks, u32 n, u32 W, const u64 * __restrict__ allowed)
{
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < (u64)n * W) masks[t] &= allowed[t % W];
}

__device__ __forceinline__ u32 segmentOf(const u32 * __restrict__ segStarts, u32 nRanks, u32 read)
{
    u32 s = 0;
    while (s + 1 < nRanks && read >= segStarts[s + 1]) ++s;
    return s;
}

// Paired batches on the owner: the reads arrive source rank by source rank, each segment = [first mates | second mates] of the pairs
// that source selected for this owner; the child mapper wants [all first mates | all second mates] (reference src/d_mapper.h:394-443,
// loadFilteredReads: reads, then mates), pairs in source order = the order of the concatenated job.  orig[j] = arrival index of read j.
__global__ void k_pair_layout(const u32 * __restrict__ segStarts, u32 nRanks, u32 n, u32 * __restrict__ orig)
{
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const u32 P = n / 2, jj = j < P ? j : j - P;
    u32 s = 0, before = 0;
    for (;; ++s)
    {
        const u32 ps = (segStarts[s + 1] - segStarts[s]) / 2;
        if (jj < before + ps || s + 1 == nRanks) break;
        before += ps;
    }
    const u32 ps = (segStarts[s + 1] - segStarts[s]) / 2;
    orig[j] = segStarts[s] + (j < P ? 0 : ps) + (jj - before);
}

// lens / masks of the permuted batch
__global__ void k_permute_scalars(const u32 * __restrict__ orig, u32 n, u32 W, const u32 * __restrict__ lensIn, const u64 * __restrict__ masksIn,
                                  u64 * __restrict__ lensOut /* n + 1, widened */, u64 * __restrict__ masksOut)
{
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > n) return;
    if (j == n) { lensOut[j] = 0; return; }
    const u32 o = orig[j];
    lensOut[j] = lensIn[o];
    for (u32 w = 0; w < W; ++w) masksOut[(size_t)j * W + w] = masksIn[(size_t)o * W + w];
}

// bases of the permuted batch: one wavefront per read
__global__ void k_permute_bases(const u32 * __restrict__ orig, u32 n, const u64 * __restrict__ offsIn, const u8 * __restrict__ seqsIn,
                                const u64 * __restrict__ offsOut, u8 * __restrict__ seqsOut)
{
    const u32 lane = threadIdx.x & 63;
    const u32 gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nW = (gridDim.x * blockDim.x) >> 6;
    for (u32 j = gw; j < n; j += nW)
    {
        const u64 src = offsIn[orig[j]], dst = offsOut[j];
        const u32 len = (u32)(offsOut[j + 1] - dst);
        for (u32 k = lane; k < len; k += 64) seqsOut[dst + k] = seqsIn[src + k];
    }
}

// raw matches per source rank (block-local histogram in LDS, one global atomic per block and rank).
// orig (may be NULL): arrival index of every read of the batch (paired batches are permuted on the owner).
__global__ void k_raw_seg_count(const RawMatch * __restrict__ pool, u32 n, const u32 * __restrict__ segStarts, u32 nRanks, const u32 * __restrict__ orig,
                                u32 * __restrict__ counts)
{
    __shared__ u32 local[64];
    if (threadIdx.x < 64) local[threadIdx.x] = 0;
    __syncthreads();
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) { const u32 r = pool[t].read; atomicAdd(&local[segmentOf(segStarts, nRanks, orig ? orig[r] : r)], 1u); }
    __syncthreads();
    if (threadIdx.x < nRanks && local[threadIdx.x]) atomicAdd(&counts[threadIdx.x], local[threadIdx.x]);
}

// cursors[s] starts at the first output slot of rank s; read ids are rewritten to the ids of the home rank
__global__ void k_raw_seg_scatter(const RawMatch * __restrict__ pool, u32 n, const u32 * __restrict__ segStarts, u32 nRanks, const u32 * __restrict__ orig,
                                  const u32 * __restrict__ ids, u32 * __restrict__ cursors, RawMatch * __restrict__ out)
{
    __shared__ u32 local[64], base[64];
    if (threadIdx.x < 64) local[threadIdx.x] = 0;
    __syncthreads();
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    RawMatch m; u32 s = 0, my = 0, arrival = 0;
    if (t < n) { m = pool[t]; arrival = orig ? orig[m.read] : m.read; s = segmentOf(segStarts, nRanks, arrival); my = atomicAdd(&local[s], 1u); }
    __syncthreads();
    if (threadIdx.x < nRanks) base[threadIdx.x] = local[threadIdx.x] ? atomicAdd(&cursors[threadIdx.x], local[threadIdx.x]) : 0;
    __syncthreads();
    if (t < n) { m.read = ids[arrival]; out[base[s] + my] = m; }
}

// ---- paired-end over sharded bins: what the home rank needs of a child mapper besides its raw matches — the primary the pairing
// chose, `paired` + probability (reference src/d_mapper.h:186-215 copyCigars ships exactly these) and a rescued mate.  40 bytes.
struct PeItemRec { u32 read; u16 bin; u8 paired; u8 rescued; u32 primary; u32 reserved; double prob; RawMatch mate; };
static_assert(sizeof(PeItemRec) == 40, "DY_PE_ITEM_BYTES");

__device__ __forceinline__ bool peItemWorthSending(const PeItem & it) { return it.nU2 || it.rescued || it.paired; }

__global__ void k_pe_items_seg_count(const PeItem * __restrict__ items, const u32 * __restrict__ itemRead, u32 nItems, const u32 * __restrict__ segStarts, u32 nRanks,
                                     const u32 * __restrict__ orig, u32 * __restrict__ counts)
{
    __shared__ u32 local[64];
    if (threadIdx.x < 64) local[threadIdx.x] = 0;
    __syncthreads();
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nItems && peItemWorthSending(items[t])) atomicAdd(&local[segmentOf(segStarts, nRanks, orig[itemRead[t]])], 1u);
    __syncthreads();
    if (threadIdx.x < nRanks && local[threadIdx.x]) atomicAdd(&counts[threadIdx.x], local[threadIdx.x]);
}

__global__ void k_pe_items_seg_scatter(const PeItem * __restrict__ items, const u32 * __restrict__ itemRead, const u16 * __restrict__ itemBin, u32 nItems,
                                       const u32 * __restrict__ segStarts, u32 nRanks, const u32 * __restrict__ orig, const u32 * __restrict__ ids,
                                       u32 * __restrict__ cursors, PeItemRec * __restrict__ out)
{
    __shared__ u32 local[64], base[64];
    if (threadIdx.x < 64) local[threadIdx.x] = 0;
    __syncthreads();
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    bool send = false; u32 s = 0, my = 0, arrival = 0;
    PeItem it;
    if (t < nItems)
    {
        it = items[t];
        send = peItemWorthSending(it);
        if (send) { arrival = orig[itemRead[t]]; s = segmentOf(segStarts, nRanks, arrival); my = atomicAdd(&local[s], 1u); }
    }
    __syncthreads();
    if (threadIdx.x < nRanks) base[threadIdx.x] = local[threadIdx.x] ? atomicAdd(&cursors[threadIdx.x], local[threadIdx.x]) : 0;
    __syncthreads();
    if (send)
    {
        PeItemRec r;
        r.read = ids[arrival]; r.bin = itemBin[t]; r.paired = it.paired ? 1 : 0; r.rescued = it.rescued ? 1 : 0; r.primary = it.primary; r.reserved = 0;
        r.prob = it.prob; r.mate = it.mate;
        if (it.rescued) r.mate.read = r.read; else { r.mate.read = 0; r.mate.begin = 0; r.mate.len = 0; r.mate.errors = 0; r.mate.strand = 0; r.mate.bin = 0; r.mate.contig = 0; }
        out[base[s] + my] = r;
    }
}

// home rank: the records land in the items of the home reads' own work list
__global__ void k_pe_items_unpack(const PeItemRec * __restrict__ recs, u32 n, PeView V, u32 nReads, u32 * __restrict__ bad)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const PeItemRec r = recs[t];
    if (r.read >= nReads || r.bin >= V.nBins) { atomicExch(bad, 1u); return; }
    const u32 item = peFindItem(V, r.read, r.bin);
    if (item >= V.binStart[r.bin + 1] || V.itemRead[item] != r.read) { atomicExch(bad, 1u); return; }    // the read did not select that bin
    PeItem & it = V.items[item];
    it.primary = r.primary; it.paired = r.paired; it.rescued = r.rescued; it.prob = r.prob; it.mate = r.mate;
    if (r.rescued) V.readHasRescue[r.read] = 1;
}

__global__ void k_raw_read_counts(const RawMatch * __restrict__ raw, u32 n, u32 nReads, u32 * __restrict__ readMatchCount, u32 * __restrict__ bad)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const u32 r = raw[t].read;
    if (r < nReads) atomicAdd(&readMatchCount[r], 1u); else *bad = 1;
}

} // namespace dy

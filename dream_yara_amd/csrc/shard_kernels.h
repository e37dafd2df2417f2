// Bins sharded over ranks: device-side packing for the two exchanges of SURVEY.md §8(e).
//   C1  reads -> owners of their selected bins   (loadFilteredReads, reference src/d_mapper.h:394-443, across ranks)
//   C2  raw matches -> home rank of the read     (copyMatches / copyCigars, reference src/d_mapper.h:127-215, across ranks)
// The library only packs and unpacks; the all-to-all itself is the caller's (RCCL over xGMI in sharded.py / multi_gpu.hpp).
// Each exchange is ONE packed buffer per rank, segments destination-major, plus a row of 64-bit counts per destination that the
// caller exchanges first; nothing here synchronises with the host (sizes stay on the device until the caller fetches the rows).
//
//   C1 segment (to owner d):   [ids u32 x n][lens u32 x n][pad to 16][masks u64 x n x W][bases, 4 bits each, every read from a byte of its own][pad to 16]
//   C2 segment (to home s):    [raw matches 16 B x r][pairing records 40 B x i][pad to 16]
#pragma once
#include "dy_types.h"
#include "map_kernels.h"        // groupCopyRead
#include <hip/hip_runtime.h>

namespace dy {

enum { MAX_RANKS = 64 };

// one row per destination; the ABI's dy_seg_header
struct SegHeader { u64 a, b, bytes, aux; };            // C1: a = reads, b = bases, aux = maxLen | minLen << 32;  C2: a = raw matches, b = pairing records
struct CountPair { u64 bases; u64 cnt; u64 packed; };      // packed = bytes of the reads' bases at 4 bits each
struct CountPairSum { __host__ __device__ CountPair operator()(const CountPair & x, const CountPair & y) const { CountPair r; r.bases = x.bases + y.bases; r.cnt = x.cnt + y.cnt; r.packed = x.packed + y.packed; return r; } };
// bytes a read of len bases takes in a C1 segment (Dna5 codes 0..4, two per byte, the first in the low half)
__host__ __device__ __forceinline__ u32 packedBytesOf(u32 len) { return (len + 1) >> 1; }

__host__ __device__ __forceinline__ u64 roundUp16(u64 x) { return (x + 15) & ~(u64)15; }
__host__ __device__ __forceinline__ u64 c1ScalarBytes(u64 n) { return roundUp16(8 * n); }                       // ids + lens
__host__ __device__ __forceinline__ u64 c1SeqOffset(u64 n, u32 W) { return c1ScalarBytes(n) + 8 * n * W; }      // masks follow the scalars (16-aligned), bases follow the masks
__host__ __device__ __forceinline__ u64 c1SegBytes(u64 n, u64 packed, u32 W) { return roundUp16(c1SeqOffset(n, W) + packed); }
__host__ __device__ __forceinline__ u64 c2SegBytes(u64 raws, u64 items) { return roundUp16(16 * raws + 40 * items); }

// bit d of the result: read i has a selected bin among those rank d owns
__device__ __forceinline__ u64 ownersOfRead(const u64 * __restrict__ masks, u32 i, u32 W, const u64 * __restrict__ owner, u32 nRanks)
{
    u64 sel = 0;
    for (u32 w = 0; w < W; ++w)
    {
        const u64 m = masks[(size_t)i * W + w];
        if (!m) continue;
        for (u32 d = 0; d < nRanks; ++d) if (m & owner[(size_t)d * W + w]) sel |= 1ull << d;
    }
    return sel;
}

__device__ __forceinline__ u32 waveSumU32(u32 v)
{
    for (int o = 32; o >= 1; o >>= 1) v += (u32)__shfl_xor((int)v, o, 64);
    return v;
}
// exclusive prefix sum over the 64 lanes
__device__ __forceinline__ u32 waveExclusiveSumU32(u32 v)
{
    const u32 lane = threadIdx.x & 63;
    u32 x = v;
    for (int o = 1; o < 64; o <<= 1) { const u32 y = (u32)__shfl_up((int)x, o, 64); if (lane >= (u32)o) x += y; }
    return x - v;
}

// 8 Dna5 codes (one per byte) <-> 32 bits, code j in bits 4j..4j+3
__device__ __forceinline__ u32 packNibbles8(u64 w)
{
    w &= 0x0F0F0F0F0F0F0F0Full;
    w = (w | (w >> 4)) & 0x00FF00FF00FF00FFull;
    w = (w | (w >> 8)) & 0x0000FFFF0000FFFFull;
    return (u32)(w | (w >> 16));
}
__device__ __forceinline__ u64 unpackNibbles8(u32 p)
{
    u64 w = p;
    w = (w | (w << 16)) & 0x0000FFFF0000FFFFull;
    w = (w | (w << 8)) & 0x00FF00FF00FF00FFull;
    return (w | (w << 4)) & 0x0F0F0F0F0F0F0F0Full;
}
struct __attribute__((packed)) PackedU32sh { u32 v; };
// g = lane of a group of 16.  src[0, len) at one code per byte -> dst[0, packedBytesOf(len)) at two per byte
__device__ __forceinline__ void groupPackRead(const u8 * __restrict__ src, u8 * __restrict__ dst, u32 len, u32 g)
{
    for (u32 p = g * 8; p + 8 <= len; p += 128) ((PackedU32sh *)(dst + (p >> 1)))->v = packNibbles8(((const PackedU64rc *)(src + p))->v);
    const u32 t = len & ~7u, rest = len - t;              // < 8 codes -> up to 4 bytes, one per lane
    if (2 * g < rest)
    {
        const u32 lo = src[t + 2 * g], hi = 2 * g + 1 < rest ? src[t + 2 * g + 1] : 0u;
        dst[(t >> 1) + g] = (u8)((lo & 15u) | (hi << 4));
    }
}
// ... and back, with the reverse complement (rc may be NULL)
__device__ __forceinline__ void groupUnpackRead(const u8 * __restrict__ src, u8 * __restrict__ dst, u8 * __restrict__ rc, u32 len, u32 g)
{
    for (u32 p = g * 8; p + 8 <= len; p += 128)
    {
        const u64 w = unpackNibbles8(((const PackedU32sh *)(src + (p >> 1)))->v);
        ((PackedU64rc *)(dst + p))->v = w;
        if (rc) ((PackedU64rc *)(rc + len - 8 - p))->v = revcomp8(w);
    }
    const u32 t = len & ~7u;
    if (g < (len & 7u))
    {
        const u8 c = (u8)((src[(t + g) >> 1] >> (4 * ((t + g) & 1u))) & 15u);
        dst[t + g] = c;
        if (rc) rc[len - 1 - (t + g)] = c < 4 ? (u8)(3 - c) : c;
    }
}

// ---- C1, home side --------------------------------------------------------------------------------------------------------------
// One wavefront per 64 reads: pairs[d * nWaves + wave] = (reads, bases) of the wavefront that go to rank d; entry nRanks * nWaves = 0.
__global__ __launch_bounds__(256) void k_c1_count(const u64 * __restrict__ masks, const u64 * __restrict__ offs, u32 n, u32 W, const u64 * __restrict__ owner,
                                                  u32 nRanks, u32 nWaves, CountPair * __restrict__ pairs)
{
    const u32 wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave == 0 && lane == 0) { CountPair z; z.bases = 0; z.cnt = 0; z.packed = 0; pairs[(size_t)nRanks * nWaves] = z; }
    if (wave >= nWaves) return;
    const u32 i = wave * 64 + lane;
    const bool valid = i < n;
    const u32 len = valid ? (u32)(offs[i + 1] - offs[i]) : 0;
    const u64 sel = valid ? ownersOfRead(masks, i, W, owner, nRanks) : 0;
    for (u32 d = 0; d < nRanks; ++d)
    {
        const bool mine = (sel >> d) & 1;
        const u64 b = __ballot(mine);
        const u32 bases = b ? waveSumU32(mine ? len : 0) : 0;
        const u32 packed = b ? waveSumU32(mine ? packedBytesOf(len) : 0) : 0;
        if (lane == 0) { CountPair p; p.bases = bases; p.cnt = (u64)__popcll(b); p.packed = packed; pairs[(size_t)d * nWaves + wave] = p; }
    }
}

// scan = exclusive scan of pairs (nRanks * nWaves + 1 entries).  One block: row d of `headers`, byte offset of segment d in the packed buffer.
__global__ void k_c1_headers(const CountPair * __restrict__ scan, u32 nRanks, u32 nWaves, u32 W, u64 lenRange, SegHeader * __restrict__ headers, u64 * __restrict__ segOff)
{
    __shared__ u64 bytes[MAX_RANKS];
    const u32 d = threadIdx.x;
    if (d < nRanks)
    {
        const CountPair lo = scan[(size_t)d * nWaves], hi = scan[(size_t)(d + 1) * nWaves];
        SegHeader h; h.a = hi.cnt - lo.cnt; h.b = hi.bases - lo.bases; h.bytes = c1SegBytes(h.a, hi.packed - lo.packed, W); h.aux = lenRange;
        headers[d] = h; bytes[d] = h.bytes;
    }
    __syncthreads();
    if (d == 0) { u64 acc = 0; for (u32 r = 0; r < nRanks; ++r) { segOff[r] = acc; acc += bytes[r]; } segOff[nRanks] = acc; }
}

// One wavefront per 64 reads; the selected lanes of a destination write their scalars, then the wavefront copies their bases together.
__global__ __launch_bounds__(256) void k_c1_pack(const u64 * __restrict__ masks, const u64 * __restrict__ offs, const u8 * __restrict__ seqs, u32 n, u32 W,
                                                 const u64 * __restrict__ owner, u32 nRanks, u32 nWaves, const CountPair * __restrict__ scan,
                                                 const SegHeader * __restrict__ headers, const u64 * __restrict__ segOff, u8 * __restrict__ out)
{
    const u32 wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= nWaves) return;
    const u32 i = wave * 64 + lane;
    const bool valid = i < n;
    const u64 src = valid ? offs[i] : 0;
    const u32 len = valid ? (u32)(offs[i + 1] - src) : 0;
    const u64 sel = valid ? ownersOfRead(masks, i, W, owner, nRanks) : 0;
    for (u32 d = 0; d < nRanks; ++d)
    {
        const bool mine = (sel >> d) & 1;
        const u64 b = __ballot(mine);
        if (!b) continue;
        const CountPair start = scan[(size_t)d * nWaves], at = scan[(size_t)d * nWaves + wave];
        const u64 nSeg = headers[d].a;
        u8 * seg = out + segOff[d];
        const u32 pos = (u32)(at.cnt - start.cnt) + (u32)__popcll(b & ((1ull << lane) - 1));
        const u64 dst = c1SeqOffset(nSeg, W) + (at.packed - start.packed) + waveExclusiveSumU32(mine ? packedBytesOf(len) : 0);
        if (mine)
        {
            ((u32 *)seg)[pos] = i; ((u32 *)seg)[nSeg + pos] = len;
            u64 * mo = (u64 *)(seg + c1ScalarBytes(nSeg)) + (size_t)pos * W;
            for (u32 w = 0; w < W; ++w) mo[w] = masks[(size_t)i * W + w] & owner[(size_t)d * W + w];
        }
        // the bases: four selected reads per step, one per group of 16 lanes (8 bytes per lane)
        u64 todo = b;
        const u32 q = lane >> 4;
        while (todo)
        {
            int mine4 = -1;
            for (u32 k = 0; k < 4 && todo; ++k) { const int l = __ffsll((long long)todo) - 1; todo &= todo - 1; if (k == q) mine4 = l; }
            const int from = mine4 < 0 ? (int)lane : mine4;
            const u64 s = __shfl(src, from, 64), dd = __shfl(dst, from, 64);
            const u32 L = __shfl(len, from, 64);
            if (mine4 >= 0) groupPackRead(seqs + s, seg + dd, L, lane & 15);
        }
    }
}

// ---- C1, owner side -------------------------------------------------------------------------------------------------------------
// Where the segments of the ranks 0..N-1 lie in a received buffer, and the index of each segment's first record in arrival order.
struct SegTable
{
    u64 off[MAX_RANKS];             // byte offset of the segment of source rank s
    u32 start[MAX_RANKS + 1];       // arrival index of its first record (reads for C1, raw matches for C2)
    u32 start2[MAX_RANKS + 1];      // C2: arrival index of its first pairing record
    u32 nRanks;
};
__device__ __forceinline__ u32 segmentOf(const u32 * __restrict__ segStarts, u32 nRanks, u32 idx)
{
    u32 s = 0;
    while (s + 1 < nRanks && idx >= segStarts[s + 1]) ++s;
    return s;
}

// arrival order: home id, widened length (n + 1 entries, the last one 0) and bin masks of every received read
// (+ oddLens: 1 where the length is odd — a read's bytes in the segment are (bases + odd reads before it) / 2 from the segment's first)
__global__ void k_c1_unpack_scalars(const u8 * __restrict__ recv, SegTable T, u32 W, u32 n, u32 * __restrict__ ids, u64 * __restrict__ lensWide, u32 * __restrict__ oddLens,
                                    u64 * __restrict__ masksOut, u32 * __restrict__ bad)
{
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > n) return;
    if (j == n) { lensWide[j] = 0; oddLens[j] = 0; return; }
    const u32 s = segmentOf(T.start, T.nRanks, j), local = j - T.start[s], nSeg = T.start[s + 1] - T.start[s];
    const u8 * seg = recv + T.off[s];
    u32 len = ((const u32 *)seg)[nSeg + local];
    if (len > 8191) { len = 0; atomicExch(bad, 1u); }          // reference src/mapper.h:453-454; a record the sender could not have packed
    ids[j] = ((const u32 *)seg)[local];
    lensWide[j] = len;
    oddLens[j] = len & 1u;
    const u64 * mi = (const u64 *)(seg + c1ScalarBytes(nSeg)) + (size_t)local * W;
    for (u32 w = 0; w < W; ++w) masksOut[(size_t)j * W + w] = mi[w];
}

// bases of the owner-side batch and their reverse complements (appendReverseComplement, reference src/store_seqs.h:271-290), 16 lanes per
// read.  orig (may be NULL): arrival index of read j (paired batches are re-ordered); arrOffs = exclusive scan of the lengths in
// arrival order, arrOdd = exclusive scan of the odd-length flags in arrival order, offsOut = offsets of the batch.
__global__ __launch_bounds__(256) void k_c1_unpack_bases(const u8 * __restrict__ recv, SegTable T, u32 W, const u32 * __restrict__ orig, u32 n, const u64 * __restrict__ arrOffs,
                                                         const u32 * __restrict__ arrOdd, const u64 * __restrict__ offsOut, u64 totalBases, u8 * __restrict__ seqsOut,
                                                         u8 * __restrict__ rcOut, u32 * __restrict__ bad)
{
    const u32 g = threadIdx.x & 15;
    const u32 group = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, nGroups = (gridDim.x * blockDim.x) >> 4;
    for (u32 j = group; j < n; j += nGroups)
    {
        const u32 a = orig ? orig[j] : j;
        const u32 s = segmentOf(T.start, T.nRanks, a), nSeg = T.start[s + 1] - T.start[s];
        const u64 src = T.off[s] + c1SeqOffset(nSeg, W) + ((arrOffs[a] - arrOffs[T.start[s]]) + (arrOdd[a] - arrOdd[T.start[s]])) / 2;
        const u64 dst = offsOut[j];
        const u32 len = (u32)(offsOut[j + 1] - dst);
        if (dst + len > totalBases) { if (g == 0) atomicExch(bad, 1u); continue; }      // the lengths do not add up to the bases announced
        groupUnpackRead(recv + src, seqsOut + dst, rcOut + dst, len, g);
    }
}

__global__ void k_and_masks(u64 * __restrict__ masks, u32 n, u32 W, const u64 * __restrict__ allowed)
{
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < (u64)n * W) masks[t] &= allowed[t % W];
}

// Paired batches on the owner: the reads arrive source rank by source rank, each segment = [first mates | second mates] of the pairs
// that source selected for this owner; the child mapper wants [all first mates | all second mates] (reference src/d_mapper.h:394-443,
// loadFilteredReads: reads, then mates), pairs in source order = the order of the concatenated job.  orig[j] = arrival index of read j.
__global__ void k_pair_layout(SegTable T, u32 n, u32 * __restrict__ orig)
{
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const u32 P = n / 2, jj = j < P ? j : j - P;
    u32 s = 0, before = 0;
    for (;; ++s)
    {
        const u32 ps = (T.start[s + 1] - T.start[s]) / 2;
        if (jj < before + ps || s + 1 == T.nRanks) break;
        before += ps;
    }
    const u32 ps = (T.start[s + 1] - T.start[s]) / 2;
    orig[j] = T.start[s] + (j < P ? 0 : ps) + (jj - before);
}

// lens / masks of the permuted batch
__global__ void k_permute_scalars(const u32 * __restrict__ orig, u32 n, u32 W, const u64 * __restrict__ lensIn, const u64 * __restrict__ masksIn,
                                  u64 * __restrict__ lensOut /* n + 1, widened */, u64 * __restrict__ masksOut)
{
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > n) return;
    if (j == n) { lensOut[j] = 0; return; }
    const u32 o = orig[j];
    lensOut[j] = lensIn[o];
    for (u32 w = 0; w < W; ++w) masksOut[(size_t)j * W + w] = masksIn[(size_t)o * W + w];
}

// ---- C2, owner side -------------------------------------------------------------------------------------------------------------
// raw matches per source rank (block-local histogram in LDS, one global atomic per block and rank).
// orig (may be NULL): arrival index of every read of the batch (paired batches are permuted on the owner).
__global__ void k_raw_seg_count(const RawMatch * __restrict__ pool, const u32 * __restrict__ nPool, SegTable T, const u32 * __restrict__ orig,
                                u32 * __restrict__ counts)
{
    __shared__ u32 local[MAX_RANKS];
    if (threadIdx.x < MAX_RANKS) local[threadIdx.x] = 0;
    __syncthreads();
    const u32 n = *nPool;
    for (u32 t0 = blockIdx.x * blockDim.x; t0 < n; t0 += gridDim.x * blockDim.x)
    {   // one LDS atomic per wavefront and rank met (with one rank, 256 threads on one LDS word took 28 us per 1 M records)
        const u32 t = t0 + threadIdx.x;
        u32 s = 0xFFFFFFFFu;
        if (t < n) { const u32 r = pool[t].read; s = segmentOf(T.start, T.nRanks, orig ? orig[r] : r); }
        u64 todo = __ballot(s != 0xFFFFFFFFu);
        while (todo)
        {
            const u32 d = (u32)__shfl((int)s, __ffsll((long long)todo) - 1, 64);
            const u64 same = __ballot(s == d);
            if ((threadIdx.x & 63) == (u32)(__ffsll((long long)same) - 1)) atomicAdd(&local[d], (u32)__popcll(same));
            todo &= ~same;
        }
    }
    __syncthreads();
    if (threadIdx.x < T.nRanks && local[threadIdx.x]) atomicAdd(&counts[threadIdx.x], local[threadIdx.x]);
}

// ---- paired-end over sharded bins: what the home rank needs of a child mapper besides its raw matches — the primary the pairing
// chose, `paired` + probability (reference src/d_mapper.h:186-215 copyCigars ships exactly these) and a rescued mate.  40 bytes.
struct PeItemRec { u32 read; u16 bin; u8 paired; u8 rescued; u32 primary; u32 reserved; double prob; RawMatch mate; };
static_assert(sizeof(PeItemRec) == 40, "DY_PE_ITEM_BYTES");

__device__ __forceinline__ bool peItemWorthSending(const PeItem & it) { return it.nU2 || it.rescued || it.paired; }

__global__ void k_pe_items_seg_count(const PeItem * __restrict__ items, const u32 * __restrict__ itemRead, const u32 * __restrict__ nItemsDev, SegTable T,
                                     const u32 * __restrict__ orig, u32 * __restrict__ counts)
{
    __shared__ u32 local[MAX_RANKS];
    if (threadIdx.x < MAX_RANKS) local[threadIdx.x] = 0;
    __syncthreads();
    const u32 nItems = *nItemsDev;
    for (u32 t = blockIdx.x * blockDim.x + threadIdx.x; t < nItems; t += gridDim.x * blockDim.x)
        if (peItemWorthSending(items[t])) atomicAdd(&local[segmentOf(T.start, T.nRanks, orig[itemRead[t]])], 1u);
    __syncthreads();
    if (threadIdx.x < T.nRanks && local[threadIdx.x]) atomicAdd(&counts[threadIdx.x], local[threadIdx.x]);
}

// Block d writes row d of the C2 rows: {raw matches to d, pairing records to d, segment bytes, 0} followed by this rank's items per bin
// (every destination gets the same loads: the receivers add the rows up, which is sortedBins()' input, reference src/d_mapper.h:843-854).
// Also: byte offset of every segment in the packed buffer, and the record cursors (zeroed).
__global__ void k_c2_rows(const u32 * __restrict__ rawCounts, const u32 * __restrict__ itemCounts, u32 nRanks, const u32 * __restrict__ binStart, u32 nBins, u32 rowBytes,
                          u8 * __restrict__ rows, u64 * __restrict__ segOff, u32 * __restrict__ cursors)
{
    const u32 d = blockIdx.x;
    SegHeader * h = (SegHeader *)(rows + (size_t)d * rowBytes);
    u32 * loads = (u32 *)(h + 1);
    if (threadIdx.x == 0)
    {
        u64 acc = 0;
        for (u32 r = 0; r < d; ++r) acc += c2SegBytes(rawCounts[r], itemCounts ? itemCounts[r] : 0);
        SegHeader v; v.a = rawCounts[d]; v.b = itemCounts ? itemCounts[d] : 0; v.bytes = c2SegBytes(v.a, v.b); v.aux = 0;
        *h = v; segOff[d] = acc;
        cursors[d] = 0; cursors[MAX_RANKS + d] = 0;
        if (d + 1 == nRanks) segOff[nRanks] = acc + v.bytes;
    }
    for (u32 b = threadIdx.x; b < nBins; b += blockDim.x) loads[b] = binStart ? binStart[b + 1] - binStart[b] : 0;
}

// read ids are rewritten to the ids of the home rank; the records of source s go to its segment in the packed buffer.
// A block owns a contiguous share of the pool and reserves its room in every segment ONCE (a reservation per 256 records made 4 096
// same-address atomics with a return value per 1 M records: 53 us): first pass counts the share per rank, second pass places the records.
__global__ void k_raw_seg_scatter(const RawMatch * __restrict__ pool, const u32 * __restrict__ nPool, SegTable T, const u32 * __restrict__ orig,
                                  const u32 * __restrict__ ids, u32 * __restrict__ cursors, const u64 * __restrict__ segOff, u8 * __restrict__ out)
{
    __shared__ u32 local[MAX_RANKS], base[MAX_RANKS];
    const u32 n = *nPool;
    const u32 share = ((n + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    const u32 from = min(n, blockIdx.x * share), to = min(n, from + share);
    const u32 lane = threadIdx.x & 63;
    if (threadIdx.x < MAX_RANKS) local[threadIdx.x] = 0;
    __syncthreads();
    for (u32 t0 = from; t0 < to; t0 += blockDim.x)
    {
        const u32 t = t0 + threadIdx.x;
        u32 s = 0xFFFFFFFFu;
        if (t < to) { const u32 r = pool[t].read; s = segmentOf(T.start, T.nRanks, orig ? orig[r] : r); }
        u64 todo = __ballot(s != 0xFFFFFFFFu);
        while (todo)
        {
            const u32 d = (u32)__shfl((int)s, __ffsll((long long)todo) - 1, 64);
            const u64 same = __ballot(s == d);
            if (lane == (u32)(__ffsll((long long)same) - 1)) atomicAdd(&local[d], (u32)__popcll(same));
            todo &= ~same;
        }
    }
    __syncthreads();
    if (threadIdx.x < T.nRanks) base[threadIdx.x] = local[threadIdx.x] ? atomicAdd(&cursors[threadIdx.x], local[threadIdx.x]) : 0;
    __syncthreads();
    for (u32 t0 = from; t0 < to; t0 += blockDim.x)
    {
        const u32 t = t0 + threadIdx.x;
        RawMatch m; u32 s = 0xFFFFFFFFu, arrival = 0, slot = 0;
        if (t < to) { m = pool[t]; arrival = orig ? orig[m.read] : m.read; s = segmentOf(T.start, T.nRanks, arrival); }
        u64 todo = __ballot(s != 0xFFFFFFFFu);
        while (todo)
        {   // the block's running cursor of a rank advances once per wavefront that holds records for it
            const u32 d = (u32)__shfl((int)s, __ffsll((long long)todo) - 1, 64);
            const u64 same = __ballot(s == d);
            const u32 leader = (u32)(__ffsll((long long)same) - 1);
            u32 first = 0;
            if (lane == leader) first = atomicAdd(&base[d], (u32)__popcll(same));
            first = (u32)__shfl((int)first, (int)leader, 64);
            if (s == d) slot = first + (u32)__popcll(same & ((1ull << lane) - 1));
            todo &= ~same;
        }
        if (t < to) { m.read = ids[arrival]; ((RawMatch *)(out + segOff[s]))[slot] = m; }
    }
}

__global__ void k_pe_items_seg_scatter(const PeItem * __restrict__ items, const u32 * __restrict__ itemRead, const u16 * __restrict__ itemBin, const u32 * __restrict__ nItemsDev,
                                       SegTable T, const u32 * __restrict__ orig, const u32 * __restrict__ ids,
                                       u32 * __restrict__ cursors, const u64 * __restrict__ segOff, const u32 * __restrict__ rawCounts, u8 * __restrict__ out)
{
    __shared__ u32 local[MAX_RANKS], base[MAX_RANKS];
    const u32 nItems = *nItemsDev;
    for (u32 t0 = blockIdx.x * blockDim.x; t0 < nItems; t0 += gridDim.x * blockDim.x)
    {
        if (threadIdx.x < MAX_RANKS) local[threadIdx.x] = 0;
        __syncthreads();
        const u32 t = t0 + threadIdx.x;
        bool send = false; u32 s = 0, my = 0, arrival = 0;
        PeItem it;
        if (t < nItems)
        {
            it = items[t];
            send = peItemWorthSending(it);
            if (send) { arrival = orig[itemRead[t]]; s = segmentOf(T.start, T.nRanks, arrival); my = atomicAdd(&local[s], 1u); }
        }
        __syncthreads();
        if (threadIdx.x < T.nRanks) base[threadIdx.x] = local[threadIdx.x] ? atomicAdd(&cursors[threadIdx.x], local[threadIdx.x]) : 0;
        __syncthreads();
        if (send)
        {
            PeItemRec r;
            r.read = ids[arrival]; r.bin = itemBin[t]; r.paired = it.paired ? 1 : 0; r.rescued = it.rescued ? 1 : 0; r.primary = it.primary; r.reserved = 0;
            r.prob = it.prob; r.mate = it.mate;
            if (it.rescued) r.mate.read = r.read; else { r.mate.read = 0; r.mate.begin = 0; r.mate.len = 0; r.mate.errors = 0; r.mate.strand = 0; r.mate.bin = 0; r.mate.contig = 0; }
            ((PeItemRec *)(out + segOff[s] + (size_t)16 * rawCounts[s]))[base[s] + my] = r;
        }
        __syncthreads();
    }
}

// ---- C2, home side --------------------------------------------------------------------------------------------------------------
// The raw matches of all sources back to back (K4 wants one array) and the matches per read; a record naming a read outside the
// batch is dropped and flagged.
__global__ void k_c2_unpack_raw(const u8 * __restrict__ recv, SegTable T, u32 n, u32 nReads, RawMatch * __restrict__ flat, u32 * __restrict__ readMatchCount, u32 * __restrict__ bad)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const u32 s = segmentOf(T.start, T.nRanks, t);
    RawMatch m = ((const RawMatch *)(recv + T.off[s]))[t - T.start[s]];
    if (m.read < nReads) atomicAdd(&readMatchCount[m.read], 1u);
    else { *bad = 1; m.read = 0xFFFFFFFFu; }               // k_group_matches drops it
    flat[t] = m;
}

// the pairing records land in the items of the home reads' own work list
__global__ void k_pe_items_unpack(const u8 * __restrict__ recv, SegTable T, u32 n, PeView V, u32 nReads, u32 * __restrict__ bad)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const u32 s = segmentOf(T.start2, T.nRanks, t);
    const PeItemRec r = ((const PeItemRec *)(recv + T.off[s] + (size_t)16 * (T.start[s + 1] - T.start[s])))[t - T.start2[s]];
    if (r.read >= nReads || r.bin >= V.nBins) { atomicExch(bad, 1u); return; }
    const u32 item = peFindItem(V, r.read, r.bin);
    if (item >= V.binStart[r.bin + 1] || V.itemRead[item] != r.read) { atomicExch(bad, 1u); return; }    // the read did not select that bin
    PeItem & it = V.items[item];
    it.primary = r.primary; it.paired = r.paired; it.rescued = r.rescued; it.prob = r.prob; it.mate = r.mate;
    if (r.rescued) V.readHasRescue[r.read] = 1;
}

// sortedBins() (reference src/d_mapper.h:843-854) from the items per bin of the whole job: rank of bin b = bins with a larger load, ties by bin number
__global__ void k_bin_rank_from_loads(const u32 * __restrict__ loads, u32 nBins, u32 * __restrict__ binOrderRank)
{
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nBins) return;
    const u32 mine = loads[b];
    u32 r = 0;
    for (u32 o = 0; o < nBins; ++o) { const u32 l = loads[o]; if (l > mine || (l == mine && o < b)) ++r; }
    binOrderRank[b] = r;
}

} // namespace dy

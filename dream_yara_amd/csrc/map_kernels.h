// K2 (seed search) and K3 (classify, rank, locate, extend) for gfx950, plus the read->bin work-list kernels.
// One work item = one (read, bin) pair selected by the IBF = one read of one child mapper in the reference.
//   work list        replaces loadFilteredReads                         reference src/d_mapper.h:394-443
//   K2 k_seed_search replaces collectSeeds<0> + findSeeds<0>(bucket 0)  reference src/mapper.h:581-649, src/mapper_collector.h:143-160,
//                                                                       src/index_fm.h:132-155, src/mapper_filter.h:62-68
//   K3 k_extend      replaces classifyReads, rankSeeds, extendHits<0..2> and the re-seeding rounds of _mapReadsImpl
//                                                                       reference src/d_mapper.h:257-294, src/mapper_classifier.h:111-155,
//                                                                       src/mapper_ranker.h:122-177, src/mapper_extender.h:147-308,
//                                                                       src/find_extender.h:87-403
// Work items are laid out bin-major, so neighbouring workgroups walk the same bin's rank blocks (L2 reuse).
#pragma once
#include <hip/hip_runtime.h>
#include "dy_types.h"
#include "fm_device.h"
#include "myers_device.h"
#include "extend_sides.h"

namespace dy {

struct ReadsView
{
    const u8 *  seqs;     // forward reads, Dna5 codes
    const u8 *  seqsRc;   // reverse complements, same offsets (appendReverseComplement, reference src/store_seqs.h:271-290)
    const u64 * offs;     // nReads + 1
    u32         nReads;
};

// ---- byte strings moved by groups of 16 lanes, 8 bytes per lane and step (unaligned 8-byte accesses): a 100-base read is one step.
// A thread per read (or a wavefront per read, a byte per lane) spends 4-10x the memory instructions on the same bytes.
struct __attribute__((packed)) PackedU64rc { u64 v; };
__device__ __forceinline__ u64 revcomp8(u64 w)          // the 8 Dna5 codes of w reversed and complemented (A<->T, C<->G is code ^ 3, N = 4 stays)
{
    const u64 r = __builtin_bswap64(w);
    const u64 isN = ((r >> 2) & 0x0101010101010101ull) * 0xFFull;
    return r ^ (0x0303030303030303ull & ~isN);
}
// g = lane of the group (0..15).  Copies src[0, len) to dst; rc (may be NULL) receives the reverse complement.
__device__ __forceinline__ void groupCopyRead(const u8 * __restrict__ src, u8 * __restrict__ dst, u8 * __restrict__ rc, u32 len, u32 g)
{
    for (u32 p = g * 8; p + 8 <= len; p += 128)
    {
        const u64 w = ((const PackedU64rc *)(src + p))->v;
        if (dst) ((PackedU64rc *)(dst + p))->v = w;
        if (rc) ((PackedU64rc *)(rc + len - 8 - p))->v = revcomp8(w);
    }
    const u32 t = len & ~7u;
    if (g < (len & 7u))
    {
        const u8 c = src[t + g];
        if (dst) dst[t + g] = c;
        if (rc) rc[len - 1 - (t + g)] = c < 4 ? (u8)(3 - c) : c;
    }
}

// reverse complements of a batch (appendReverseComplement, reference src/store_seqs.h:271-290): 16 lanes per read
__global__ __launch_bounds__(256) void k_revcomp_reads(const u8 * __restrict__ fwd, u8 * __restrict__ rc, const u64 * __restrict__ offs, u32 nReads)
{
    const u32 g = threadIdx.x & 15;
    const u32 group = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, nGroups = (gridDim.x * blockDim.x) >> 4;
    for (u32 r = group; r < nReads; r += nGroups)
    {
        const u64 o = offs[r];
        groupCopyRead(fwd + o, nullptr, rc + o, (u32)(offs[r + 1] - o), g);
    }
}

// mismatches of a[0..n) against b[0..n), 32 bytes per step with every load issued before the first count; there is no 8-byte / single-byte
// tail: the chunks of the last step are CLAMPED to end at n (chunk k starts at min(i + 8k, n - 8)) and the bytes a clamped chunk re-reads
// are masked out of its flags — K2's 13-byte verification is one round trip instead of two
// CH = chunks per step: 4 where the ranges are long (K3's 25-byte seed check is one round trip), 2 where registers decide the kernel's
// occupancy (K2: its 7..13-byte verification fits one step anyway)
// PADDED = both arrays are readable for 8 bytes from a / b whatever n is (the indexed text and the read arrays end in 64 spare bytes):
// fewer than 8 bytes are then one masked chunk instead of a round trip per byte.
template <int CH = 4, bool PADDED = false>
__device__ __forceinline__ u32 countMismatches(const u8 * __restrict__ a, const u8 * __restrict__ b, int n)
{
    u32 c = 0; int i = 0;
    if (PADDED && n < 8) return n > 0 ? (u32)__popcll(mismatchFlags(load8(a), load8(b)) & (~0ull >> (8 * (8 - n)))) : 0u;
    if (n >= 8)
    {
        const int last = n - 8;
        for (; i < n; i += 8 * CH)
        {
            u64 av[CH], bv[CH];
#pragma unroll
            for (int k = 0; k < CH; ++k)
            {   // only the chunks that start inside [0, n) are fetched (K2 is bound by its requests)
                const int start = i + 8 * k, o = start < last ? start : last;
                av[k] = 0; bv[k] = 0;
                if (start < n) { av[k] = load8(a + o); bv[k] = load8(b + o); }
            }
#pragma unroll
            for (int k = 0; k < CH; ++k)
            {
                const int start = i + 8 * k, o = start < last ? start : last;
                if (start < n) c += (u32)__popcll(mismatchFlags(av[k], bv[k]) & (~0ull << (8 * (start - o))));
            }
        }
        return c;
    }
    for (; i < n; ++i) { const u32 x = a[i], y = b[i]; c += !(x == y && x < 4); }
    return c;
}

// base `pos` of the read sequence of `strand` (1 = reverse complement)
__device__ __forceinline__ u32 readBase(const u8 * __restrict__ seq, u32 len, u32 strand, u32 pos)
{
    u32 c = strand ? seq[len - 1 - pos] : seq[pos];
    return (strand && c < 4) ? 3 - c : c;
}

__host__ __device__ __forceinline__ u32 readErrorsOf(u32 len, float errorRate)
{
    u32 e = (u32)((float)len * errorRate);          // reference src/misc_options.h:258-263
    return e < MAX_ERRORS ? e : MAX_ERRORS;
}
__host__ __device__ __forceinline__ u32 readStrataOf(u32 len, float strataRate)
{
    u32 s = (u32)((float)len * strataRate);         // reference src/misc_options.h:283-287 (strataCount is never set by the CLI)
    return s < MAX_ERRORS ? s : MAX_ERRORS;
}
// seeds per strand for a given seed-error budget: ceil((e+1)/(E+1)) (reference src/mapper_collector.h:151-155)
__host__ __device__ __forceinline__ u32 seedCountOf(u32 readErrors, u32 seedErrors) { return (readErrors + 1 + seedErrors) / (seedErrors + 1); }

// ---------------------------------------------------------------------------------------------------
// Work list: per wave of 64 reads count the reads selected for every bin, one device-wide exclusive scan of the
// bin-major count matrix, scatter (deterministic order).
// ---------------------------------------------------------------------------------------------------
__global__ void k_items_count(const u64 * __restrict__ masks, u32 nReads, u32 W, u32 nBins, u32 * __restrict__ waveCounts)
{
    const u32 gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const u32 read = gw * 64 + lane;
    const u32 nWaves = (nReads + 63) / 64;
    if (gw >= nWaves) return;
    for (u32 w = 0; w < W; ++w)
    {
        const u64 m = read < nReads ? masks[(size_t)read * W + w] : 0;
        // lane j ends up with the number of reads of this wave selected for bin w*64+j
        u32 cnt = 0;
        for (u32 j = 0; j < 64; ++j)
        {
            const u32 c = (u32)__popcll(__ballot((m >> j) & 1));
            if (lane == j) cnt = c;
        }
        const u32 bin = w * 64 + lane;
        if (bin < nBins) waveCounts[(size_t)bin * nWaves + gw] = cnt;
    }
}

// bin starts from the exclusive scan of the bin-major count matrix
__global__ void k_items_bin_starts(const u32 * __restrict__ waveOffsets, const u32 * __restrict__ total, u32 nWaves, u32 nBins, u32 * __restrict__ binStart)
{
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nBins) return;
    binStart[b] = b < nBins ? waveOffsets[(size_t)b * nWaves] : *total;
}

// The bin processing order of sortedBins() — descending load, ties by bin number (reference src/d_mapper.h:843-854; std::sort leaves
// ties unspecified) — as a rank per bin: the loads go through LDS a tile at a time, every thread counts the bins ahead of its own.
__global__ __launch_bounds__(256) void k_items_bin_rank(const u32 * __restrict__ binStart, u32 nBins, u32 * __restrict__ binOrderRank)
{
    __shared__ u32 tile[256];
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 mine = b < nBins ? binStart[b + 1] - binStart[b] : 0;
    u32 r = 0;
    for (u32 base = 0; base < nBins; base += 256)
    {
        const u32 o = base + threadIdx.x;
        tile[threadIdx.x] = o < nBins ? binStart[o + 1] - binStart[o] : 0;
        __syncthreads();
        const u32 cnt = min(256u, nBins - base);
        for (u32 k = 0; k < cnt; ++k)
        {
            const u32 t = tile[k];
            r += (t > mine || (t == mine && base + k < b)) ? 1u : 0u;
        }
        __syncthreads();
    }
    if (b < nBins) binOrderRank[b] = r;
}

// ---- the same work list without the matrix (the default): the masks are sparse (about one bin per read), so instead of a bin-major count
// matrix of nBins x nReads/64 cells the set bits are emitted as (bin, read) pairs in read order — one thread per mask WORD, coalesced — and
// a stable radix sort by bin gives the bin-major, read-ascending order; the bin starts are lower bounds in the sorted keys.  Per 1 M reads:
// 1 024 bins 1.05 -> 0.33 ms, 256 bins 0.245 -> 0.137 ms, 64 bins 0.124 -> 0.114 ms.
// bits at or above nBins in a read's last word are not bins: both passes ignore them (every producer of masks clears them, but a count that
// disagreed with the emit would feed uninitialised pairs to the sort)
__device__ __forceinline__ u64 binBitsOfWord(u64 m, u32 w, u32 nBins)
{
    const u32 left = nBins - w * 64;                   // w < W = ceil(nBins / 64), so left >= 1
    return left >= 64 ? m : m & ((1ull << left) - 1);
}
__global__ void k_items_word_count(const u64 * __restrict__ masks, u64 nWords, u32 W, u32 nBins, u32 * __restrict__ readCount)
{
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nWords) return;
    const u64 m = binBitsOfWord(masks[t], (u32)(t % W), nBins);
    if (m) atomicAdd(&readCount[t / W], (u32)__popcll(m));
}

// cap: entries the arrays hold (the list's size is not known on the host when the kernels are queued, see k_items_clamp)
__global__ void k_items_emit(const u64 * __restrict__ masks, u64 nWords, u32 W, u32 nBins, const u32 * __restrict__ readOffset,
                             u16 * __restrict__ keys, u32 * __restrict__ vals, u32 cap)
{
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nWords) return;
    u64 m = masks[t];
    if (!m) return;
    const u32 read = (u32)(t / W), w = (u32)(t % W);
    u32 dst = readOffset[read];
    for (u32 k = 0; k < w; ++k) dst += (u32)__popcll(masks[t - w + k]);       // the read's earlier words: the same one or two cache lines
    while (m)
    {
        const u32 j = (u32)__ffsll((long long)m) - 1;
        m &= m - 1;
        const u32 bin = w * 64 + j;
        if (bin >= nBins) break;
        if (dst < cap) { keys[dst] = (u16)bin; vals[dst] = read; }
        ++dst;
    }
}

// The work list's size as every later kernel sees it: what was counted, or the capacity of the arrays if that is smaller — then flags[0]
// is raised and the host, which queued the whole step without waiting for the count, maps the batch again with room enough.
__global__ void k_items_clamp(const u32 * __restrict__ counted, u32 cap, u32 * __restrict__ nItems, u32 * __restrict__ flags)
{
    const u32 n = *counted;
    nItems[0] = n < cap ? n : cap; nItems[1] = n;
    if (n > cap) flags[0] = 1;
}

// binStart[b] = first item of bin b in the sorted keys (nItems for b = nBins)
__global__ void k_items_bin_bounds(const u16 * __restrict__ keys, const u32 * __restrict__ nItemsDev, u32 nBins, u32 * __restrict__ binStart)
{
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nBins) return;
    u32 lo = 0, hi = *nItemsDev;
    while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (keys[mid] < b) lo = mid + 1; else hi = mid; }
    binStart[b] = lo;
}

__global__ void k_items_scatter(const u64 * __restrict__ masks, u32 nReads, u32 W, u32 nBins, const u32 * __restrict__ waveOffsets,
                                u32 * __restrict__ itemRead, u16 * __restrict__ itemBin)
{
    const u32 gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const u32 read = gw * 64 + lane;
    const u32 nWaves = (nReads + 63) / 64;
    if (gw >= nWaves) return;
    for (u32 w = 0; w < W; ++w)
    {
        u64 m = read < nReads ? masks[(size_t)read * W + w] : 0;
        u64 any = m;
        for (int off = 32; off >= 1; off >>= 1)
        {
            u32 lo = __shfl_xor((u32)any, off, 64), hi = __shfl_xor((u32)(any >> 32), off, 64);
            any |= ((u64)hi << 32) | lo;
        }
        while (any)
        {
            u32 j = __ffsll((long long)any) - 1;
            any &= any - 1;
            u32 bin = w * 64 + j;
            if (bin >= nBins) continue;
            bool mine = (m >> j) & 1;
            u64 b = __ballot(mine);
            if (mine)
            {
                u32 rank = __popcll(b & ((1ull << lane) - 1));
                u32 dst = waveOffsets[(size_t)bin * nWaves + gw] + rank;      // global offset: the matrix is scanned bin-major
                itemRead[dst] = read;
                itemBin[dst] = (u16)bin;
            }
        }
    }
}

// K2 compiles to 74 registers = 6 waves per SIMD.  Capping it at 64 registers for 8 waves (the LF walk then spills 28 bytes per lane) was
// slower, 0.547 ms against 0.484 ms per 1 M reads: the kernel waits on ~20 random 64-byte DRAM accesses per read, not on latency.
// ---------------------------------------------------------------------------------------------------
// K2: exact search of the round-0 seeds. One thread per (item, strand, seed slot).
// hits0[item * 2*S0 + strand*S0 + seedId] = (lo, hi); seedId k covers read offset (count-1-k)*len
// (seeds are stored in descending offset order, reference src/mapper_collector.h:166-171).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_seed_search(IndexView ix, ReadsView rd, const u32 * __restrict__ itemRead,
                                                     const u16 * __restrict__ itemBin, const u32 * __restrict__ nItemsDev, u32 S0, float errorRate,
                                                     uint2 * __restrict__ hits0)
{
    // the grid is sized for the arrays' capacity; the blocks the list needs are remapped among themselves
    const u32 nItems = *nItemsDev;
    const u32 slots = 2 * S0;
    const u32 useful = (u32)((((u64)nItems * slots + 255) / 256 + 7) / 8 * 8);
    if (blockIdx.x >= useful) return;
    const u64 t = (u64)xcdRemapBlock(blockIdx.x, useful) * blockDim.x + threadIdx.x;
    const u32 item = (u32)(t / slots), slot = (u32)(t % slots);
    if (item >= nItems) return;
    const u32 strand = slot / S0, seedId = slot % S0;
    const u32 read = itemRead[item];
    const u8 * seq = rd.seqs + rd.offs[read];
    const u32 len = (u32)(rd.offs[read + 1] - rd.offs[read]);
    const u32 cnt = seedCountOf(readErrorsOf(len, errorRate), 0);
    uint2 res = make_uint2(0, 0);
    if (seedId < cnt && len >= cnt)
    {
        const u32 seedLen = len / cnt, off = (cnt - 1 - seedId) * seedLen;
        const u32 bin = itemBin[item];
        // the bin's view (~25 registers) is only built on the paths that walk the FM index: the table paths need the contig limits at most
        u32 lo = 0, hi = ix.bins[bin].n, d0 = 0;
        bool ok = true;
        if ((ix.prefixTable || ix.prefixWide) && seedLen >= ix.prefixQ)
        {   // first q characters through the q-mer table
            // the seed is contiguous in the strand's own array (seqsRc holds the reverse complements): 8 bases per load
            u32 idx = 0;
            {   // q <= 12 bases = two 8-byte words, both requested at once (they may run past the read: the arrays end in 64 spare bytes)
                const u8 * pat0 = (strand ? rd.seqsRc : rd.seqs) + rd.offs[read] + off;
                const u64 w0 = load8(pat0), w1 = load8(pat0 + 8);
                for (u32 d = 0; d < ix.prefixQ; ++d)
                {
                    const u32 c = (u32)((d < 8 ? w0 >> (8 * d) : w1 >> (8 * (d - 8))) & 0xff);
                    ok &= c < 4; idx = idx * 4 + (c & 3);
                }
            }
            d0 = ix.prefixQ;
            if (ok)
            {
                uint2 r; u64 next = 0;
                if (ix.prefixWide)
                {
                    const uint4 e = ix.prefixWide[ix.bins[bin].prefixOff + idx];
                    r = make_uint2(e.x, e.y); next = e.z | ((u64)e.w << 32);
                }
                else r = ix.prefixTable[ix.bins[bin].prefixOff + idx];
                if (r.y >= DIRECT_QMER && ix.prefixWide && seedLen - d0 <= 32)
                {   // the q-mer occurs once and the entry holds its position AND the bases that follow it: the rest of the seed is compared
                    // two bits a base against the entry itself (no second miss in the indexed text)
                    const u32 rem = seedLen - d0, contig = r.y & 0xFFFFu, x = r.x;
                    uint2 hit = make_uint2(0u, 0u);
                    if (x >= rem)                                                // else the seed would start before the contig does
                    {
                        const u8 * pat = (strand ? rd.seqsRc : rd.seqs) + rd.offs[read] + off + d0;
                        u64 w[4] = {0, 0, 0, 0};
#pragma unroll
                        for (u32 k = 0; k < 4; ++k) if (rem > 8 * k) w[k] = load8(pat + 8 * k);
                        u64 p = 0, bad = 0;
#pragma unroll
                        for (u32 k = 0; k < 4; ++k)
                        {
                            if (rem <= 8 * k) continue;
                            const u32 nb = min(8u, rem - 8 * k);
                            bad |= w[k] & 0xFCFCFCFCFCFCFCFCull & (~0ull >> (8 * (8 - nb)));          // an N never matches
                            p |= (u64)pack2bit8(w[k]) << (16 * k);
                        }
                        const u64 cmp = rem ? (p ^ next) & (~0ull >> (64 - 2 * rem)) : 0ull;
                        if (!bad && !cmp) hit = make_uint2(x - rem, DIRECT_HIT | contig);
                    }
                    hits0[(size_t)item * slots + slot] = hit;
                    return;
                }
                if (r.y >= DIRECT_QMER)
                {   // the q-mer occurs once and the table holds its position: compare the rest of the seed with the indexed text
                    const u32 rem = seedLen - d0, contig = r.y & 0xFFFFu, x = r.x;
                    const u64 * limits = ix.limits + ix.bins[bin].limOff;
                    const u64 lim = limits[contig];
                    const u32 clen = (u32)(limits[contig + 1] - lim);
                    uint2 hit = make_uint2(0u, 0u);
                    if (x >= rem)                                                // else the seed would start before the contig does
                    {
                        const u8 * pat = (strand ? rd.seqsRc : rd.seqs) + rd.offs[read] + off + d0;
                        if (!countMismatches<2, true>(ix.ftext + ix.bins[bin].contigOff + lim + (clen - x), pat, (int)rem)) hit = make_uint2(x - rem, DIRECT_HIT | contig);
                    }
                    hits0[(size_t)item * slots + slot] = hit;
                    return;
                }
                lo = r.x; hi = r.y; ok = lo < hi;
            }
            if (ok && ix.isa && hi - lo <= 4)
            {   // narrow range: every occurrence of the seed starts with one of these q-mer occurrences (1 locate + 1 compare each,
                // instead of seedLen - q dependent LF steps); several survivors go back to rows through the inverse SA
                const u32 rem = seedLen - d0;
                const u8 * pat = (strand ? rd.seqsRc : rd.seqs) + rd.offs[read] + off + d0;
                const u8 * ftext = ix.ftext + ix.bins[bin].contigOff;
                const u32 * isa = ix.isa + ix.bins[bin].saFullOff;
                const BinView b = makeBinView(ix, bin);
                u32 found = 0, first = 0xFFFFFFFFu, oneX = 0, oneContig = 0;
                for (u32 row = lo; row < hi; ++row)
                {
                    u32 contig, x;
                    fmLocate(b, row, contig, x);
                    const u64 lim = b.limits[contig];
                    const u32 clen = (u32)(b.limits[contig + 1] - lim);
                    if (x < rem) continue;
                    if (countMismatches<2, true>(ftext + lim + (clen - x), pat, (int)rem)) continue;
                    ++found;
                    oneX = x - rem; oneContig = contig;
                    if (found > 1 || row + 1 < hi) first = min(first, isa[lim + x - rem]);
                }
                uint2 hit = make_uint2(0u, 0u);
                if (found == 1) hit = make_uint2(oneX, DIRECT_HIT | oneContig);
                else if (found > 1) hit = make_uint2(first, first + found);
                hits0[(size_t)item * slots + slot] = hit;
                return;
            }
        }
        if (ok && d0 < seedLen)
        {
            const BinView b = makeBinView(ix, bin);
            for (u32 d = d0; d < seedLen && ok; ++d)
            {
                u32 c = readBase(seq, len, strand, off + d);
                ok = c < 4 && fmStep(b, lo, hi, c);          // N never matches (reference src/index_fm.h:147)
            }
        }
        if (ok) res = make_uint2(lo, hi);
    }
    hits0[(size_t)item * slots + slot] = res;
}

struct ItemState
{
    u32 minErrors;     // 255 = none yet (ReadsContext::minErrors, reference src/bits_context.h:86)
    bool mapped;
    // the first two matches of an item are kept in registers and appended with ONE pool reservation per block at
    // the end of the kernel (a single shared counter saturates at ~10^8 atomics/s); further matches go straight to the pool
    RawMatch buf0, buf1;
    u32 nbuf;
    bool defer;        // k_extend's first pass only: the item needs what that pass leaves out (a third match, a full edit distance search)
};

struct MatchSink
{
    RawMatch * pool;
    unsigned long long * poolCount;   // 64-bit: the count keeps growing past the capacity (it sizes the retry) and must not wrap
    u32        poolCap;
    u32 *      overflow;
    u32 *      readMatchCount;
};

// extend() + _addMatchImpl() for one located seed occurrence (reference src/find_extender.h:262-403, src/mapper_extender.h:293-308).
// `seq` is the read sequence of the seed's strand (forward or reverse complement), readable front to back.
// `emit(matchBegin, matchLength, errors)` is called once if the occurrence extends to a match.
// `edit` (-y full, TSeedsDistance = EditDistance, reference src/find_extender.h:314-322): no _checkSeed, and the left extension starts at
// the seed's END, re-aligning the seed region as well.
//
// FAST (the first pass of k_extend, Hamming seeds only): where an extension side stops at a mismatch, the side is closed WITHOUT the edit
// distance search when everything after that mismatch matches — the search's answer is then known:
//   * the mismatching pair cannot be part of an exact alignment, so no end position has distance 0; the diagonal end has distance 1;
//   * the search keeps the LAST end position with the smallest distance: one position beyond the diagonal has distance 1 only if the
//     needle equals the text shifted by one (the text's first character inserted) — checked with one more LCP; positions further out
//     cost at least 2;
//   * right side (last base held out, reference src/find_extender.h:369-395): the same with the held-out base compared at the end the
//     search reports; if it differs on the diagonal the side has two errors and is not closed here.
// Anything else returns EXT_NEEDS_SEARCH before a match is emitted, and the caller hands the ITEM to the second pass (FAST = false: the
// search itself).  A wavefront runs the search loop as long as any of its lanes does, so leaving it out only pays when whole wavefronts
// skip it: tests/native/myers_fuzz.cpp checks the closed forms against the search on random and low-complexity inputs.
enum : u32 { EXT_DONE = 0, EXT_NEEDS_SEARCH = 1 };
template <int NW, bool FAST, typename TEmit>
__device__ __forceinline__ u32 extendOccurrenceT(const u8 * __restrict__ binContigs, const u64 * __restrict__ binLimits, u32 contig, u32 hayBeginSeed, u32 seedLen,
                                                 const u8 * __restrict__ seq, u32 len, u32 needleBeginSeed, u32 maxErrors, bool edit, TEmit emit)
{
    const u8 * __restrict__ ctg = binContigs + binLimits[contig];
    const u32 clen = (u32)(binLimits[contig + 1] - binLimits[contig]);
    const u32 hayEnd = hayBeginSeed + seedLen, needleEnd = needleBeginSeed + seedLen;
    // _checkSeed: recount against the real contig (Ns were randomised in the index); the early-out of the reference's
    // loop only matters when the total exceeds maxErrors, which fails either way
    u32 errors = edit ? 0u : countMismatches(ctg + hayBeginSeed, seq + needleBeginSeed, (int)seedLen);
    if (errors > maxErrors) return EXT_DONE;
    // anchor of the left extension
    const u32 hayBegin = edit ? hayEnd : hayBeginSeed, needleBegin = edit ? needleEnd : needleBeginSeed;
    u32 matchBegin = hayBegin;
    if (needleBegin > 0)
    {   // _extendLeft on the reversed infixes
        const u32 leftOffset = needleBeginSeed + (maxErrors - errors);
        const u32 leftBegin = hayBeginSeed > leftOffset ? hayBeginSeed - leftOffset : 0;
        const int hayLen = (int)(hayBegin - leftBegin), ndlLen = (int)needleBegin;
        const int lcp = lcpBackward(ctg + hayBegin, seq + needleBegin, hayLen < ndlLen ? hayLen : ndlLen);
        if (lcp == ndlLen)
            matchBegin -= lcp;
        else
        {
            const u32 remaining = maxErrors - errors;
            if (!remaining) return EXT_DONE;
            const u32 hb = hayBegin - 1 - lcp, nb = needleBegin - 1 - lcp;
            int end = 0;
            u32 cost = 1;
            if (FAST) { if (!closeLeftSide(ctg, hb, seq, nb, ndlLen - lcp, hayLen - lcp, end)) return EXT_NEEDS_SEARCH; }
            else cost = searchLeftSide<NW>(ctg, hb, seq, nb, ndlLen - lcp, hayLen - lcp, remaining, end);
            errors += cost;
            matchBegin -= (u32)(end + lcp);
            if (errors > maxErrors) return EXT_DONE;
        }
    }
    u32 matchEnd = hayEnd;
    if (needleEnd < len)
    {   // _extendRight, last base held out
        const u32 rightOffset = len - needleBeginSeed + (maxErrors - errors);
        u32 rightEnd = clen;
        if (rightEnd > hayBeginSeed + rightOffset) rightEnd = hayBeginSeed + rightOffset;
        const int hayLen = (int)(rightEnd - hayEnd), ndlLen = (int)(len - needleEnd);
        const int lcp = lcpForward(ctg + hayEnd, seq + needleEnd, hayLen < ndlLen ? hayLen : ndlLen);
        if (lcp == ndlLen)
            matchEnd += lcp;
        else if (lcp == hayLen)
        {
            errors += (u32)(ndlLen - hayLen);
            matchEnd += lcp;
            if (errors > maxErrors) return EXT_DONE;
        }
        else
        {
            const u32 remaining = maxErrors - errors;
            if (!remaining) return EXT_DONE;
            const u32 hb = hayEnd + lcp, nb = needleEnd + lcp;
            int end = 0;
            u32 cost = 1;
            if (FAST) { if (!closeRightSide(ctg, hb, seq, nb, ndlLen - lcp, hayLen - lcp, end)) return EXT_NEEDS_SEARCH; }
            else cost = searchRightSide<NW>(ctg, hb, seq, nb, ndlLen - lcp, hayLen - lcp, remaining, end);
            errors += cost;
            matchEnd += (u32)(end + lcp + 1);
            if (errors > maxErrors) return EXT_DONE;
        }
    }
    emit(matchBegin, matchEnd - matchBegin, errors);
    return EXT_DONE;
}

// one item per lane: _addMatchImpl into the lane's ItemState
template <int NW, bool FAST>
__device__ u32 extendOccurrence(const u8 * __restrict__ binContigs, const u64 * __restrict__ binLimits, u32 contig, u32 hayBegin, u32 seedLen,
                                const u8 * __restrict__ seq, u32 len, u32 strand,
                                u32 needleBegin, u32 maxErrors, u32 read, u32 bin, ItemState & st, const MatchSink & sink, bool edit)
{
    return extendOccurrenceT<NW, FAST>(binContigs, binLimits, contig, hayBegin, seedLen, seq, len, needleBegin, maxErrors, edit, [&](u32 matchBegin, u32 matchLen, u32 errors) {
        RawMatch m;
        m.read = read; m.begin = matchBegin; m.len = (u16)matchLen; m.errors = (u8)errors; m.strand = (u8)strand;
        m.bin = (u16)bin; m.contig = (u16)contig;
        if (st.nbuf == 0) { st.buf0 = m; st.nbuf = 1; }
        else if (st.nbuf == 1) { st.buf1 = m; st.nbuf = 2; }
        else if (FAST) st.defer = true;                  // the first pass never touches the pool
        else
        {
            const u64 slot = atomicAdd(sink.poolCount, 1ull);
            if (slot < sink.poolCap) { sink.pool[slot] = m; atomicAdd(&sink.readMatchCount[read], 1u); }
            else atomicExch(sink.overflow, 1u);
        }
        if (errors < st.minErrors) st.minErrors = errors;
    });
}

// Hamming backtracking over the FM trie with at most maxErr substitutions (SeqAn find(..., Backtracking<HammingDistance>),
// SURVEY.md C.1): calls leaf(lo, hi) for every leaf (any error count <= maxErr). Depth-first with one frame per error spent.
template <typename TLeaf>
__device__ void fmBacktrack(const BinView & b, const u8 * __restrict__ seq, u32 len, u32 strand, u32 off, u32 seedLen, u32 maxErr, TLeaf leaf)
{
    struct Frame { u32 lo, hi, d, alt; };
    Frame st[3];
    int level = 0;
    st[0].lo = 0; st[0].hi = b.n; st[0].d = 0; st[0].alt = 0;
    while (level >= 0)
    {
        Frame & f = st[level];
        if (f.d == seedLen) { leaf(f.lo, f.hi); --level; continue; }
        const u32 sc = readBase(seq, len, strand, off + f.d);
        if ((u32)level < maxErr && f.alt < 4)
        {
            const u32 c = f.alt++;
            if (c == sc) continue;
            u32 lo = f.lo, hi = f.hi;
            if (fmStep(b, lo, hi, c))
            {
                ++level;
                st[level].lo = lo; st[level].hi = hi; st[level].d = f.d + 1; st[level].alt = 0;
            }
            continue;
        }
        // advance along the seed's own character
        if (sc > 3) { --level; continue; }
        u32 lo = f.lo, hi = f.hi;
        if (!fmStep(b, lo, hi, sc)) { --level; continue; }
        f.lo = lo; f.hi = hi; f.d += 1; f.alt = 0;
    }
}

// Edit-distance backtracking over the FM trie (-y full: SeqAn find(..., Backtracking<EditDistance>), reference src/mapper.h:607-649) — the
// enumeration defined in oracle/dyo_mapper.hpp fmSearchEdit (PARITY UNPINNED, see there): depth-first, children A, C, G, T, every node
// carries the band |i - depth| <= maxErr of the DP column D[i] = ed(seed[0..i), path); leaf(lo, hi) for every node with D[seedLen] <= maxErr.
// One frame per trie level: the stack (seedLen + maxErr + 2 frames) lives in a global slab owned by the wavefront, which walks in step.
struct EditFrame { u32 lo, hi; u8 alt; u8 band[5]; u16 pad; };
static_assert(sizeof(EditFrame) == 16, "frame slab stride");
template <typename TLeaf>
__device__ void fmBacktrackEdit(const BinView & b, const u8 * __restrict__ seq, u32 len, u32 strand, u32 off, u32 seedLen, u32 maxErr, EditFrame * __restrict__ st, TLeaf leaf)
{
    if (maxErr > 2) return;          // a frame's band holds 2 * 2 + 1 cells: seed errors are 1 or 2 (reference src/mapper_classifier.h:141-149); the host refuses larger budgets
    const u32 E = maxErr, BW = 2 * E + 1, INF = E + 1;
    int level = 0;
    st[0].lo = 0; st[0].hi = b.n; st[0].alt = 0;
    for (u32 t = 0; t < 5; ++t) st[0].band[t] = (t < BW && t >= E) ? (u8)(t - E) : (u8)INF;
    while (level >= 0)
    {
        EditFrame f = st[level];
        if (f.alt == 0 && level > 0)
        {
            const int t = (int)seedLen - level + (int)E;
            if (t >= 0 && t < (int)BW && f.band[t] <= E) leaf(f.lo, f.hi);
        }
        if (f.alt == 4 || (u32)level == seedLen + E) { --level; continue; }
        const u32 c = f.alt;
        st[level].alt = (u8)(c + 1);
        u8 nb[5]; u32 best = INF;
        for (u32 t = 0; t < 5; ++t)
        {
            u32 v = INF;
            const int i = level + 1 - (int)E + (int)t;
            if (t < BW && i >= 0 && i <= (int)seedLen)
            {
                if (i == 0) v = (u32)level + 1;
                else
                {
                    const u32 sc = readBase(seq, len, strand, off + (u32)i - 1);
                    const u32 diag = (u32)f.band[t] + ((sc == c && sc < 4) ? 0u : 1u);
                    const u32 up = t + 1 < BW ? (u32)f.band[t + 1] + 1 : INF;
                    const u32 left = t > 0 ? (u32)nb[t - 1] + 1 : INF;
                    v = min(diag, min(up, left));
                }
                if (v > INF) v = INF;
            }
            nb[t] = (u8)v;
            best = min(best, v);
        }
        if (best > E) continue;
        u32 lo = f.lo, hi = f.hi;
        if (!fmStep(b, lo, hi, c)) continue;
        ++level;
        EditFrame g; g.lo = lo; g.hi = hi; g.alt = 0; g.pad = 0;
        for (u32 t = 0; t < 5; ++t) g.band[t] = nb[t];
        st[level] = g;
    }
}

// ---------------------------------------------------------------------------------------------------
// K3: one thread per work item runs the read-local remainder of _mapReadsImpl.
// ---------------------------------------------------------------------------------------------------
// ---- K3 work order ---------------------------------------------------------------------------------------------------------
// A wavefront of k_extend runs as many extension rounds as its slowest lane, and the rounds an item needs follow from how many
// of its exact seeds were found: all of them -> the read matches without error, one LCP pass, no Myers; each seed that
// found nothing costs one more rank (the early exit minErrors <= rank) and a Myers call.  Items are therefore processed in
// the order (bin, number of seeds found): wavefronts become homogeneous.  Only the ORDER changes — itemRead / itemBin / hits0
// stay as they are, the raw matches are sorted downstream anyway.
__device__ __forceinline__ u32 itemOrderKey(const uint2 * __restrict__ h0, u32 S0, u32 bin)
{
    u32 nf = 0, nr = 0;
    for (u32 k = 0; k < S0; ++k) { nf += hitCount(h0[k]) != 0; nr += hitCount(h0[S0 + k]) != 0; }
    const u32 cls = max(nf, nr);
    return bin * 8 + (cls < 7 ? cls : 7);
}
// Slot of every thread's key: counters[key] += (threads of the block with that key), ONE global atomic per key and block.
// The list is bin-major, so a block sees a handful of consecutive keys: they are counted in an LDS window that starts at the
// block's smallest key; keys outside the window (pathological: hundreds of tiny bins inside one block) go to the global counter
// one by one.  Every thread of the block must call this.
enum { ORDER_WINDOW = 256 };
__device__ __forceinline__ u32 blockKeySlot(bool valid, u32 key, u32 * __restrict__ counters)
{
    __shared__ u32 winCount[ORDER_WINDOW], winBase[ORDER_WINDOW];
    __shared__ u32 firstKey;
    if (threadIdx.x == 0) firstKey = 0xFFFFFFFFu;
    for (u32 i = threadIdx.x; i < ORDER_WINDOW; i += blockDim.x) winCount[i] = 0;
    __syncthreads();
    if (valid) atomicMin(&firstKey, key);
    __syncthreads();
    const u32 rel = key - firstKey;
    const bool inWin = valid && rel < ORDER_WINDOW;
    u32 rank = 0;
    if (inWin) rank = atomicAdd(&winCount[rel], 1u);
    __syncthreads();
    for (u32 i = threadIdx.x; i < ORDER_WINDOW; i += blockDim.x) winBase[i] = winCount[i] ? atomicAdd(&counters[firstKey + i], winCount[i]) : 0;
    __syncthreads();
    u32 slot = 0;
    if (inWin) slot = winBase[rel] + rank;
    else if (valid) slot = atomicAdd(&counters[key], 1u);
    return slot;
}
__global__ __launch_bounds__(1024) void k_item_order_count(const uint2 * __restrict__ hits0, const u16 * __restrict__ itemBin, const u32 * __restrict__ nItemsDev, u32 S0,
                                                           u32 * __restrict__ keys, u32 * __restrict__ hist)
{
    const u32 nItems = *nItemsDev;
    if (blockIdx.x * blockDim.x >= nItems) return;            // (uniform per block: no thread of it reaches the barriers below)
    const u32 item = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = item < nItems;
    u32 key = 0;
    if (valid) { key = itemOrderKey(hits0 + (size_t)item * 2 * S0, S0, itemBin[item]); keys[item] = key; }
    (void)blockKeySlot(valid, key, hist);
}
__global__ __launch_bounds__(1024) void k_item_order_scatter(const u32 * __restrict__ keys, const u32 * __restrict__ nItemsDev, u32 * __restrict__ cursor /* = exclusive scan of hist */,
                                                             u32 * __restrict__ perm)
{
    const u32 nItems = *nItemsDev;
    if (blockIdx.x * blockDim.x >= nItems) return;
    const u32 item = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = item < nItems;
    const u32 key = valid ? keys[item] : 0;
    const u32 slot = blockKeySlot(valid, key, cursor);
    if (valid) perm[slot] = item;
}

// classifyReads for one item (reference src/mapper_classifier.h:111-155): 0 = easy, 1 / 2 = re-seed with that many seed errors
__device__ __forceinline__ u32 classifyItem(const uint2 * __restrict__ h0, u32 S0, const MapParams & prm, u64 & total)
{
    total = 0;
    for (u32 s = 0; s < 2 * S0; ++s) total += hitCount(h0[s]);
    if (total > prm.hitsThreshold) return (total < 2ull * 200 * prm.hitsThreshold) ? 1 : 2;
    return 0;
}
__device__ __forceinline__ u32 classifyItem(const uint2 * __restrict__ h0, u32 S0, const MapParams & prm) { u64 total; return classifyItem(h0, S0, prm, total); }
// An item that is not re-seeded but has this many exact seed hits still goes to the wavefront-per-item kernel (same result: with
// seedErrors = 0 it runs the exact round only): one lane would extend them one after the other while its 63 neighbours wait.
#ifndef DY_BUSY_ITEM_HITS
#define DY_BUSY_ITEM_HITS 64
#endif

// Easy items (seedErrors == 0): rankSeeds + extendHits<0>(bucket 0) for the whole wavefront.
// Every lane walks its own (rank, strand, SA row) sequence — seeds by ascending hit count, ties keep ascending seed
// id, forward hits before reverse hits, early exit after a rank — but the lanes meet at ONE extension call per
// round, so the wavefront pays for max(#occurrences per lane) extensions instead of one per (rank, strand) pair.
// FAST: see extendOccurrenceT; returns true when the item has to be redone by the full form (nothing of it has left the lane then).
// listHard: this call sees the item for the first time (re-seeded and busy items are listed for k_extend_hard exactly once).
template <int NW, bool FAST>
__device__ bool extendEasyItem(const IndexView & ix, const ReadsView & rd, u32 item, u32 read, u32 bin, u32 S0, const MapParams & prm,
                               const uint2 * __restrict__ hits0, const MatchSink & sink, u32 * __restrict__ hardCount, u32 * __restrict__ hardList,
                               bool listHard, ItemState & st, bool valid)
{
    bool deferred = false;
    const u8 * seq = nullptr, * seqRc = nullptr;
    u32 len = 0, readErrors = 0, readStrata = 0, cnt0 = 1, seedErrors = 0;
    // only what the extension itself needs stays live per lane (the full BinView costs ~25 VGPRs; a row that still has to be
    // located — a seed with several occurrences — rebuilds it on the spot)
    const u8 * binContigs = nullptr; const u64 * binLimits = nullptr;
    const uint2 * h0 = hits0;
    if (valid)
    {
        seq = rd.seqs + rd.offs[read];
        seqRc = rd.seqsRc + rd.offs[read];
        len = (u32)(rd.offs[read + 1] - rd.offs[read]);
        readErrors = readErrorsOf(len, prm.errorRate); readStrata = readStrataOf(len, prm.strataRate);
        binContigs = ix.contigs + ix.bins[bin].contigOff; binLimits = ix.limits + ix.bins[bin].limOff;
        h0 = hits0 + (size_t)item * 2 * S0;
        cnt0 = seedCountOf(readErrors, 0);
        u64 totalHits;
        seedErrors = classifyItem(h0, S0, prm, totalHits);
        if (seedErrors == 0 && totalHits > DY_BUSY_ITEM_HITS) { seedErrors = 3; if (listHard) atomicAdd(hardCount + 10, 1u); }   // not re-seeded, just busy (counted apart: counters[12]); any non-zero value keeps it out of this kernel
        if (seedErrors && listHard) hardList[atomicAdd(hardCount, 1u)] = item;          // left to k_extend_hard
    }
    bool active = valid && seedErrors == 0 && len >= cnt0;
    const u32 seedLen = active ? len / cnt0 : 0;
    u32 done[2] = {0, 0};   // bitmask of seeds already taken per strand (cnt0 <= 32)
    u32 rank = 0, strand = 0, row = 0, rowEnd = 0, needleBegin = 0, directX = 0, directContig = 0;
    bool direct = false;
    auto pickSeed = [&]() {
        u32 best = 0xFFFFFFFFu, bestId = 0;
        for (u32 k = 0; k < cnt0; ++k)
        {
            if (done[strand] >> k & 1) continue;
            const u32 c = hitCount(h0[strand * S0 + k]);
            if (c < best) { best = c; bestId = k; }
        }
        done[strand] |= 1u << bestId;
        const uint2 r = h0[strand * S0 + bestId];
        needleBegin = (cnt0 - 1 - bestId) * seedLen;
        if (r.y >= DIRECT_HIT) { direct = true; directX = r.x; directContig = r.y & 0xFFFFu; row = 0; rowEnd = 1; }     // located by K2
        else { direct = false; row = r.x; rowEnd = r.y; }
    };
    if (active) pickSeed();
    for (;;)
    {
        bool have = false;
        while (active)
        {
            if (row < rowEnd) { have = true; break; }
            if (strand == 0) { strand = 1; pickSeed(); }
            else
            {
                if (st.minErrors + readStrata <= rank) { st.mapped = true; active = false; break; }   // seedErrors = 0: rank * (0 + 1)
                if (++rank >= cnt0) { active = false; break; }
                strand = 0; pickSeed();
            }
        }
        if (!__ballot(have)) break;
        if (have)
        {   // _extendHitImpl for one SA row (reference src/mapper_extender.h:263-285)
            u32 contig, x;
            if (direct) { contig = directContig; x = directX; } else fmLocate(makeBinView(ix, bin), row, contig, x);
            ++row;
            const u32 clen = (u32)(binLimits[contig + 1] - binLimits[contig]);
            if (clen - x >= seedLen)
            {
                const u32 status = extendOccurrence<NW, FAST>(binContigs, binLimits, contig, clen - x - seedLen, seedLen, strand ? seqRc : seq, len, strand, needleBegin, readErrors,
                                                              read, bin, st, sink, FAST ? false : prm.sensitivity == 2);      // the host starts the first pass for Hamming seeds only
                if (FAST && (status == EXT_NEEDS_SEARCH || st.defer))
                {   // the whole item is redone by the full form: nothing of it has left the lane yet
                    deferred = true;
                    st.nbuf = 0; st.defer = false; active = false;
                }
            }
        }
    }
    return deferred;
}

#ifndef DY_K3_WAVES
#define DY_K3_WAVES 5      // waves per SIMD the easy-path kernel is compiled for at NW <= 2 (with the one-word banded edit distance: 4 -> 0.660 ms, 5 -> 0.646 ms, 6 -> 0.692 ms per 1 M reads)
#endif

// Hard item: re-seeded with seedErrors = 1 or 2 (buckets 1 / 2), Hamming backtracking in the FM index.
// ONE WAVEFRONT PER ITEM: a re-seeded read has hundreds to hundreds of thousands of seed occurrences, so the trie walk (cheap) is
// done by all lanes in step and the SA rows of every leaf are dealt to the 64 lanes, each extending its own occurrence.  A rank is
// a unit (all hits of the rank's two seeds are extended, then minErrors decides the exit), so the order inside it is free; the
// minimum over the lanes is taken after each rank.  Matches go to the pool step by step (st's buffers stay unused).
// the same per WAVEFRONT (no barrier: the other wavefronts of the block may have left)
__device__ __forceinline__ void flushWaveMatches(const ItemState & st, u32 read, const MatchSink & sink, u32 lane)
{
    const u32 cnt = st.nbuf;
    u32 incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const u32 t = __shfl_up(incl, o, 64); if ((int)lane >= o) incl += t; }
    const u32 total = (u32)__shfl(incl, 63, 64);
    if (!total) return;
    u64 base = 0;
    if (lane == 0) base = atomicAdd(sink.poolCount, (unsigned long long)total);
    base = ((u64)__shfl((u32)(base >> 32), 0, 64) << 32) | (u32)__shfl((u32)base, 0, 64);
    const u64 first = base + incl - cnt;
    if (cnt >= 1)
    {
        if (first < sink.poolCap) { sink.pool[first] = st.buf0; atomicAdd(&sink.readMatchCount[read], 1u); } else atomicExch(sink.overflow, 1u);
    }
    if (cnt >= 2)
    {
        if (first + 1 < sink.poolCap) { sink.pool[first + 1] = st.buf1; atomicAdd(&sink.readMatchCount[read], 1u); } else atomicExch(sink.overflow, 1u);
    }
}

template <int NW>
__device__ void extendHardItemWave(const IndexView & ix, const ReadsView & rd, u32 item, u32 read, u32 bin, u32 S0, const MapParams & prm,
                                   const uint2 * __restrict__ hits0, const MatchSink & sink, ItemState & st, u32 lane, EditFrame * __restrict__ editStack)
{
    const bool edit = prm.sensitivity == 2;
    const u8 * seq = rd.seqs + rd.offs[read];
    const u8 * seqRc = rd.seqsRc + rd.offs[read];
    const u32 len = (u32)(rd.offs[read + 1] - rd.offs[read]);
    const u32 readErrors = readErrorsOf(len, prm.errorRate), readStrata = readStrataOf(len, prm.strataRate);
    const BinView b = makeBinView(ix, bin);
    const u32 seedErrors = classifyItem(hits0 + (size_t)item * 2 * S0, S0, prm);
    const u32 cntB = seedCountOf(readErrors, seedErrors);
    if (len < cntB) return;
    const u32 seedLen = len / cntB;
    // rounds: E = 0 (exact search of the long seeds), 1, and 2 (only bucket 2, only above LOW sensitivity)
    for (u32 E = 0; E <= 2 && !st.mapped; ++E)
    {
        if (E > seedErrors) break;
        if (E == 2 && prm.sensitivity == 0) break;
        // per-seed hit counts (findSeeds<E> + rankSeeds), the same in every lane
        u32 counts[2][MAX_SEEDS];
        for (u32 strand = 0; strand < 2; ++strand)
            for (u32 k = 0; k < cntB; ++k)
            {
                u32 c = 0;
                if (edit && E > 0) fmBacktrackEdit(b, seq, len, strand, (cntB - 1 - k) * seedLen, seedLen, E, editStack, [&](u32 lo, u32 hi) { c += hi - lo; });
                else fmBacktrack(b, seq, len, strand, (cntB - 1 - k) * seedLen, seedLen, E, [&](u32 lo, u32 hi) { c += hi - lo; });
                counts[strand][k] = c;
            }
        u32 done[2] = {0, 0};
        for (u32 rank = 0; rank < cntB && !st.mapped; ++rank)
        {
            for (u32 strand = 0; strand < 2; ++strand)
            {
                u32 best = 0xFFFFFFFFu, bestId = 0;
                for (u32 k = 0; k < cntB; ++k)
                {
                    if (done[strand] >> k & 1) continue;
                    if (counts[strand][k] < best) { best = counts[strand][k]; bestId = k; }
                }
                done[strand] |= 1u << bestId;
                const u32 off = (cntB - 1 - bestId) * seedLen;
                const u8 * sq = strand ? seqRc : seq;
                auto onLeaf = [&](u32 lo, u32 hi) {
                    for (u32 base = lo; base < hi; base += 64)
                    {   // lanes beyond the leaf idle for this step; the loop bound is the same in every lane
                        const u32 row = base + lane;
                        u32 got = 0, mBegin = 0, mLen = 0, mErrors = 0, mContig = 0;
                        if (row < hi)
                        {
                            u32 contig, x;
                            fmLocate(b, row, contig, x);
                            const u32 clen = (u32)(b.limits[contig + 1] - b.limits[contig]);
                            if (clen - x >= seedLen)      // suffixLength >= seedLength
                                extendOccurrenceT<NW, false>(b.contigs, b.limits, contig, clen - x - seedLen, seedLen, sq, len, off, readErrors, edit,
                                                      [&](u32 matchBegin, u32 matchLen, u32 errors) { got = 1; mBegin = matchBegin; mLen = matchLen; mErrors = errors; mContig = contig; });
                        }
                        // every lane holds at most one new match of the SAME read: one pool reservation and one per-read count update per
                        // step for the whole wavefront (a repeat read yields 10^3..10^5 matches; same-address atomics are served one by one)
                        const u64 have = __ballot(got != 0);
                        if (have)
                        {
                            const u32 total = (u32)__popcll(have);
                            u64 first = 0;
                            if (lane == 0)
                            {
                                first = atomicAdd(sink.poolCount, (unsigned long long)total);
                                if (first + total <= sink.poolCap) atomicAdd(&sink.readMatchCount[read], total); else atomicExch(sink.overflow, 1u);
                            }
                            first = ((u64)__shfl((u32)(first >> 32), 0, 64) << 32) | (u32)__shfl((u32)first, 0, 64);
                            if (got && first + total <= sink.poolCap)
                            {
                                RawMatch m;
                                m.read = read; m.begin = mBegin; m.len = (u16)mLen; m.errors = (u8)mErrors; m.strand = (u8)strand;
                                m.bin = (u16)bin; m.contig = (u16)mContig;
                                sink.pool[first + (u32)__popcll(have & ((1ull << lane) - 1))] = m;
                            }
                            if (got && mErrors < st.minErrors) st.minErrors = mErrors;
                        }
                    }
                };
                if (edit && E > 0) fmBacktrackEdit(b, seq, len, strand, off, seedLen, E, editStack, onLeaf);
                else fmBacktrack(b, seq, len, strand, off, seedLen, E, onLeaf);
            }
            // ReadsContext::minErrors is one value per read
            u32 mn = st.minErrors;
            for (int o = 32; o >= 1; o >>= 1) mn = min(mn, (u32)__shfl_xor(mn, o, 64));
            st.minErrors = mn;
            if (st.minErrors + readStrata <= rank * (E + 1)) st.mapped = true;
        }
    }
}

// one pool reservation per BLOCK for the matches buffered in the threads' ItemStates (same-address atomics are served one at a time;
// every thread of the block must call this)
__device__ __forceinline__ void flushBlockMatches(const ItemState & st, u32 read, const MatchSink & sink, u32 lane)
{
    __shared__ u32 waveOff[16];
    __shared__ u64 blockBase;
    const u32 wave = threadIdx.x >> 6, nWaves = (blockDim.x + 63) >> 6;
    const u32 cnt = st.nbuf;
    u32 incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const u32 t = __shfl_up(incl, o, 64); if ((int)lane >= o) incl += t; }
    if (lane == 63) waveOff[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        u32 total = 0;
        for (u32 w = 0; w < nWaves; ++w) { const u32 t = waveOff[w]; waveOff[w] = total; total += t; }
        blockBase = total ? atomicAdd(sink.poolCount, (unsigned long long)total) : 0;
    }
    __syncthreads();
    const u64 first = blockBase + waveOff[wave] + incl - cnt;
    if (cnt >= 1)
    {
        if (first < sink.poolCap) { sink.pool[first] = st.buf0; atomicAdd(&sink.readMatchCount[read], 1u); } else atomicExch(sink.overflow, 1u);
    }
    if (cnt >= 2)
    {
        if (first + 1 < sink.poolCap) { sink.pool[first + 1] = st.buf1; atomicAdd(&sink.readMatchCount[read], 1u); } else atomicExch(sink.overflow, 1u);
    }
}

// K3 in three forms (MODE), the same raw matches from each:
//   0  the full algorithm in one pass: every edit distance search where it arises.  A wavefront runs the search loop as long as ANY lane
//      needs it, and nearly every wavefront holds such a lane;
//   1  the FAST pass (extendOccurrenceT) over every item; the items it cannot finish go to slowList, which a second launch (MODE 0 with
//      nItemsDev = the list's counter, perm = the list, listHard = 0) redoes in full;
//   2  the FAST pass, then the block's unfinished items — compacted into its first wavefronts through LDS — redone in full by the same
//      block: the searches run in a quarter (or less) of the wavefronts, beside other blocks' first phases instead of in a sparse launch
//      of their own.
template <int NW, int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NW <= 2 ? DY_K3_WAVES : 1, NW <= 2 ? DY_K3_WAVES : 8))) void k_extend(IndexView ix, ReadsView rd, const u32 * __restrict__ itemRead, const u16 * __restrict__ itemBin,
                                                const u32 * __restrict__ nItemsDev, u32 S0, MapParams prm, const uint2 * __restrict__ hits0, MatchSink sink,
                                                u32 * __restrict__ hardCount, u32 * __restrict__ hardList, const u32 * __restrict__ perm,
                                                u32 * __restrict__ slowCount, u32 * __restrict__ slowList, u32 listHard)
{
    const u32 nItems = *nItemsDev;
    const u32 useful = ((nItems + 255) / 256 + 7) / 8 * 8;
    if (blockIdx.x >= useful) return;                         // uniform per block
    const u32 idx = xcdRemapBlock(blockIdx.x, useful) * blockDim.x + threadIdx.x;
    const u32 lane = threadIdx.x & 63;
    ItemState st; st.minErrors = 255; st.mapped = false; st.nbuf = 0; st.defer = false;
    u32 read = 0;
    u32 bin = 0;
    const bool valid = idx < nItems;
    const u32 item = valid ? perm[idx] : 0;           // work order: see k_item_order_*
    if (valid) { read = itemRead[item]; bin = itemBin[item]; }
    if (MODE == 0)
    {
        extendEasyItem<NW, false>(ix, rd, item, read, bin, S0, prm, hits0, sink, hardCount, hardList, listHard != 0, st, valid);
        flushBlockMatches(st, read, sink, lane);
        return;
    }
    const bool deferred = extendEasyItem<NW, true>(ix, rd, item, read, bin, S0, prm, hits0, sink, hardCount, hardList, true, st, valid);
    if (MODE == 1)
    {
        if (deferred) slowList[atomicAdd(slowCount, 1u)] = item;
        flushBlockMatches(st, read, sink, lane);
        return;
    }
    __shared__ u32 blockSlow[256];
    __shared__ u32 blockSlowCount;
    if (threadIdx.x == 0) blockSlowCount = 0;
    __syncthreads();
    if (deferred) blockSlow[atomicAdd(&blockSlowCount, 1u)] = item;
    flushBlockMatches(st, read, sink, lane);                  // (its barriers also close the list)
    const u32 nSlow = blockSlowCount;
    if (slowCount && threadIdx.x == 0 && nSlow) atomicAdd(slowCount, nSlow);      // statistics only
    if ((threadIdx.x & ~63u) >= nSlow) return;                // wavefronts without a listed item leave (uniform per wavefront): their slots go to other blocks
    const bool valid2 = threadIdx.x < nSlow;
    const u32 item2 = valid2 ? blockSlow[threadIdx.x] : 0;
    u32 read2 = 0, bin2 = 0;
    if (valid2) { read2 = itemRead[item2]; bin2 = itemBin[item2]; }
    ItemState st2; st2.minErrors = 255; st2.mapped = false; st2.nbuf = 0; st2.defer = false;
    extendEasyItem<NW, false>(ix, rd, item2, read2, bin2, S0, prm, hits0, sink, hardCount, hardList, false, st2, valid2);
    flushWaveMatches(st2, read2, sink, lane);
}

// The re-seeded items listed by k_extend (the list length is only known on the device).
template <int NW>
__global__ __launch_bounds__(256) void k_extend_hard(IndexView ix, ReadsView rd, const u32 * __restrict__ itemRead, const u16 * __restrict__ itemBin,
                                                     u32 S0, MapParams prm, const uint2 * __restrict__ hits0, MatchSink sink,
                                                     const u32 * __restrict__ hardCount, const u32 * __restrict__ hardList,
                                                     EditFrame * __restrict__ editSlab, u32 editStride)
{
    const u32 lane = threadIdx.x & 63;
    const u32 n = *hardCount;
    const u32 totalWaves = gridDim.x * (blockDim.x >> 6), waveId = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    EditFrame * editStack = editSlab ? editSlab + (size_t)waveId * editStride : nullptr;
    for (u32 i = waveId; i < n; i += totalWaves)
    {   // one wavefront per re-seeded item (see extendHardItemWave)
        ItemState st; st.minErrors = 255; st.mapped = false; st.nbuf = 0; st.defer = false;
        const u32 item = hardList[i];
        const u32 read = itemRead[item];
        extendHardItemWave<NW>(ix, rd, item, read, itemBin[item], S0, prm, hits0, sink, st, lane, editStack);
    }
}

} // namespace dy

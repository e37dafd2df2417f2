// Paired-end stages on the device.  UNPINNED by any reference fixture; parity is against
// the CPU oracle only.
//   k_pe_or_masks      the four sequences of a pair vote together            reference src/d_mapper.h:350-358
//   k_pe_child         per (read, bin): unique matches by coordinate + SE rank  reference src/mapper.h:772-906
//   k_pe_library       per bin: library mean / deviation from unique proper pairs  reference src/mapper.h:929-985
//   k_pe_pair          per (pair, bin): findPrimaryMatch both ways             reference src/mapper.h:989-1061, src/bits_matches.h:819-1041
//   k_pe_rescue(_jobs) mate rescue: unbanded affine DP in the insert window (K4)  reference src/mapper_verifier.h:141-273, src/find_verifier.h:80-140
//   k_pe_route / k_pe_finalize  alignMatches + copyMatches/copyCigars + main mapper  reference src/d_mapper.h:127-215,610-721
// Batch layout like the reference: reads [0,P) are the first mates, [P,2P) the second mates; both mates of a pair are
// work items of every bin the pair selects, so inside a bin's item segment item j and item j + Pc are mates.
#pragma once
#include <hip/hip_runtime.h>
#include "dy_types.h"
#include "dy_detmath.h"
#include "map_kernels.h"
#include "finalize_kernels.h"

namespace dy {

struct PeItem                 // one per work item, zero-initialised
{
    u32 u2Off, nU2;           // unique matches of the item (coordinate order) in peU2
    u32 nOpt, nSet;           // co-optimal / within-strata counts of the child mapper
    u32 primary;              // index (coordinate order) of the child's primary match
    u32 minErr;
    u32 paired, rescued;
    double prob;
    RawMatch mate;            // rescued match (valid iff rescued)
};

struct PeLibrary { u32 length, dev, pairing, nLens; };

struct PeView
{
    PeItem * items;
    RawMatch * u2;            // capacity = number of raw matches
    const u32 * itemRead; const u16 * itemBin; const u32 * binStart;
    const u64 * masks; u32 W; u32 nBins; u32 nPairs; u32 nItems;
    PeLibrary * lib;          // per bin
    u32 * libLens;            // nItems entries of scratch (per bin region [binStart, binStart + Pc))
    u32 * libIdx;             // same size
    u32 optLibLength, optLibDev, verify;
    float errorRate, indelRate, strataRate;
    u32 * readHasRescue;      // per read
};

__global__ void k_pe_or_masks(u64 * __restrict__ masks, u32 nPairs, u32 W)
{
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (u64)nPairs * W) return;
    const u64 m = masks[t] | masks[t + (u64)nPairs * W];
    masks[t] = m; masks[t + (u64)nPairs * W] = m;
}

__global__ void k_pe_mcounts(const u32 * __restrict__ readMatchCount, const u64 * __restrict__ masks, u32 W, u32 nReads, u32 * __restrict__ out)
{
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > nReads) return;
    u32 c = 0;
    if (r < nReads) { c = 3 * readMatchCount[r]; for (u32 w = 0; w < W; ++w) c += (u32)__popcll(masks[(size_t)r * W + w]); }
    out[r] = c;
}

__device__ __forceinline__ u32 peFindItem(const PeView & V, u32 read, u32 bin)
{
    u32 lo = V.binStart[bin], hi = V.binStart[bin + 1];
    while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (V.itemRead[mid] < read) lo = mid + 1; else hi = mid; }
    return lo;
}

// ---- child passes of one (read, bin) segment [p0,p1) of idxA (sorted by lessEnd) -------------------------------
// pass 1 -> idxB[0..n1), optional stale tail 1 -> M; pass 2: idxB sorted by lessBegin; optional stale tail 2 -> M.
struct ChildSeg { u32 n1, n2, minErr, nOpt, nSet; };
__device__ inline ChildSeg peChildPasses(const RawMatch * E, const u32 * idxA, u32 * idxB, u32 p0, u32 p1, u32 lane, RawMatch * M, u32 & nM,
                                         bool emitTails, u32 strata)
{
    ChildSeg r; r.n1 = 0;
    for (u32 base = p0; base < p1; base += 64)
    {
        const u32 p = base + lane;
        bool first = false;
        if (p < p1)
        {
            const RawMatch a = E[idxA[p]];
            first = p == p0;
            if (!first) { const RawMatch b = E[idxA[p - 1]]; first = !(sameBinContigStrand(a, b) && mEnd(a) == mEnd(b)); }
        }
        const u32 slot = wavePush(first, lane, r.n1);
        if (first) idxB[slot] = idxA[p];
    }
    if (emitTails)
        for (u32 base = p0 + r.n1; base < p1; base += 64)
        {
            const u32 p = base + lane;
            const u32 slot = wavePush(p < p1, lane, nM);
            if (p < p1) M[slot] = E[idxA[p]];
        }
    waveFence();
    waveSort(idxB, r.n1, lane, [&](u32 a, u32 b) { return lessBegin(E[a], a, E[b], b); });
    u32 n2 = 0, myMin = 255;
    for (u32 base = 0; base < r.n1; base += 64)
    {
        const u32 q = base + lane;
        bool first = false;
        if (q < r.n1)
        {
            const RawMatch a = E[idxB[q]];
            first = q == 0;
            if (!first) { const RawMatch b = E[idxB[q - 1]]; first = !(sameBinContigStrand(a, b) && a.begin == b.begin); }
            if (first) myMin = min(myMin, (u32)a.errors);
        }
        n2 += (u32)__popcll(__ballot(first));
    }
    r.n2 = n2; r.minErr = waveMin32(myMin);
    if (emitTails)
        for (u32 base = n2; base < r.n1; base += 64)
        {
            const u32 q = base + lane;
            const u32 slot = wavePush(q < r.n1, lane, nM);
            if (q < r.n1) M[slot] = E[idxB[q]];
        }
    u32 nOpt = 0, nSet = 0;
    for (u32 base = 0; base < r.n1; base += 64)
    {
        const u32 q = base + lane;
        bool first = false; u32 e = 0;
        if (q < r.n1)
        {
            const RawMatch a = E[idxB[q]];
            first = q == 0;
            if (!first) { const RawMatch b = E[idxB[q - 1]]; first = !(sameBinContigStrand(a, b) && a.begin == b.begin); }
            e = a.errors;
        }
        nOpt += (u32)__popcll(__ballot(first && e == r.minErr));
        nSet += (u32)__popcll(__ballot(first && e <= r.minErr + strata));
    }
    r.nOpt = nOpt; r.nSet = nSet;
    return r;
}

// ---- stage 1: per read, per bin -> unique matches in coordinate order + SE rank ------------------------------------
// One thread per read: a read with a single raw match is its own child result; reads with more go to the list of k_pe_child.
__global__ __launch_bounds__(1024) void k_pe_child_simple(FinalizeArgs A, PeView V, u32 * __restrict__ list, u32 * __restrict__ listCount)
{
    const u32 read = blockIdx.x * blockDim.x + threadIdx.x;
    bool more = false;
    if (read < A.nReads)
    {
        const u32 s0 = A.readStart[read], n = A.readStart[read + 1] - s0;
        if (n == 1)
        {
            const RawMatch a = A.gp[s0];
            V.u2[s0] = a;
            PeItem & it = V.items[peFindItem(V, read, a.bin)];
            it.u2Off = s0; it.nU2 = 1; it.nOpt = 1; it.nSet = 1; it.primary = 0; it.minErr = a.errors;
        }
        more = n > 1;
    }
    const bool want[1] = {more};
    u32 * const lists[1] = {list};
    u32 * const counts[1] = {listCount};
    blockAppend<1>(want, read, lists, counts);
}

__global__ __launch_bounds__(256) void k_pe_child(FinalizeArgs A, PeView V, const u32 * __restrict__ list, const u32 * __restrict__ listCount)
{
    const u32 lane = threadIdx.x & 63;
    const u32 gwave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const u32 nWaves = (gridDim.x * blockDim.x) >> 6;
    const u32 nList = *listCount;
    for (u32 li = gwave; li < nList; li += nWaves)
    {
        const u32 read = list[li];
        const u32 s0 = A.readStart[read], n = A.readStart[read + 1] - s0;
        if (n == 0) continue;
        const u32 readLen = (u32)(A.readOffs[read + 1] - A.readOffs[read]);
        const u32 strata = readStrataOf(readLen, A.strataRate);
        const RawMatch * E = A.gp + s0;
        u32 * idxA = A.aux + (size_t)4 * s0;
        u32 * idxB = idxA + n;
        for (u32 i = lane; i < n; i += 64) idxA[i] = i;
        waveFence();
        waveSort(idxA, n, lane, [&](u32 a, u32 b) { return lessEnd(E[a], a, E[b], b); });
        u32 nM = 0, p0 = 0, u2Write = s0;
        while (p0 < n)
        {
            const u32 bin = E[idxA[p0]].bin;
            u32 cntBin = 0;
            for (u32 p = p0 + lane; p < n; p += 64) cntBin += E[idxA[p]].bin == bin;
            const u32 p1 = p0 + waveSum32(cntBin);
            const ChildSeg seg = peChildPasses(E, idxA, idxB, p0, p1, lane, nullptr, nM, false, strata);
            const u32 pickB = libcxxFirstUniform(seg.nOpt);
            u32 firstSeen = 0, optSeen = 0, sePick = 0;
            for (u32 base = 0; base < seg.n1; base += 64)
            {
                const u32 q = base + lane;
                RawMatch a; bool first = false;
                if (q < seg.n1)
                {
                    a = E[idxB[q]];
                    first = q == 0;
                    if (!first) { const RawMatch b = E[idxB[q - 1]]; first = !(sameBinContigStrand(a, b) && a.begin == b.begin); }
                }
                const u64 balFirst = __ballot(first);
                const u32 myFirstRank = firstSeen + (u32)__popcll(balFirst & ((1ull << lane) - 1));
                if (first) V.u2[u2Write + myFirstRank] = a;
                const bool opt = first && a.errors == seg.minErr;
                const u64 balOpt = __ballot(opt);
                const u32 myOptRank = optSeen + (u32)__popcll(balOpt & ((1ull << lane) - 1));
                const u64 balPick = __ballot(opt && myOptRank == pickB);
                if (balPick) sePick = __shfl(myFirstRank, __ffsll((long long)balPick) - 1, 64);
                firstSeen += (u32)__popcll(balFirst); optSeen += (u32)__popcll(balOpt);
            }
            if (lane == 0)
            {
                PeItem & it = V.items[peFindItem(V, read, bin)];
                it.u2Off = u2Write; it.nU2 = seg.n2; it.nOpt = seg.nOpt; it.nSet = seg.nSet; it.primary = sePick; it.minErr = seg.minErr;
            }
            u2Write += seg.n2;
            waveFence();
            p0 = p1;
        }
    }
}

// ---- pairing geometry (reference src/bits_matches.h:587-666) -------------------------------------------------------
__device__ __forceinline__ bool peOrientationProper(const RawMatch & one, const RawMatch & two)
{
    const bool oneBeforeTwo = one.begin < two.begin;
    return (!one.strand && two.strand && oneBeforeTwo) || (!two.strand && one.strand && !oneBeforeTwo);
}
__device__ __forceinline__ u32 peLibraryLength(const RawMatch & a, const RawMatch & b)
{
    if (a.len && b.len && a.contig == b.contig) return mEnd(b) > a.begin ? mEnd(b) - a.begin : mEnd(a) - b.begin;
    return 0xFFFFFFFFu;
}
__device__ __forceinline__ u32 peLibraryDeviation(const RawMatch & a, const RawMatch & b, u32 mean)
{
    if (a.len && b.len && a.contig == b.contig) { const int d = (int)peLibraryLength(a, b) - (int)mean; return (u32)(d < 0 ? -d : d); }
    return 0xFFFFFFFFu;
}
__device__ __forceinline__ double peLibraryProb(const RawMatch & one, const RawMatch & two, u32 mean, u32 stdDev)
{
    const bool proper = peOrientationProper(one, two) && peLibraryDeviation(one, two, mean) < 6 * stdDev;
    if (!proper) return 0.0009;
    const double dev = (double)peLibraryDeviation(one, two, mean);
    const double score = dev / stdDev;
    const double e = detErfc(score / 1.41421356237);
    return e > 0.001 ? e : 0.001;
}

// ---- stage 2: library statistics of one bin (one 1024-thread block per bin) ----------------------------------------------
// out[pos] = value for the threads that want it, in thread order, pos continuing from `count` (block-wide, ordered)
__device__ __forceinline__ u32 blockPushOrdered(bool want, u32 & count, u32 * waveTotals /* LDS, 16 */)
{
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nWaves = blockDim.x >> 6;
    const u64 mask = __ballot(want);
    if (lane == 0) waveTotals[wave] = (u32)__popcll(mask);
    __syncthreads();
    u32 before = 0, total = 0;
    for (u32 w = 0; w < nWaves; ++w) { const u32 t = waveTotals[w]; if (w < wave) before += t; total += t; }
    const u32 slot = count + before + (u32)__popcll(mask & ((1ull << lane) - 1));
    count += total;
    __syncthreads();
    return slot;
}

__global__ __launch_bounds__(1024) void k_pe_library(PeView V)
{
    __shared__ u32 hist[2048];
    __shared__ u32 waveTotals[16];
    __shared__ u32 sel[4];          // radix select state: prefix value, remaining rank
    __shared__ float chunk[4096];
    const u32 bin = blockIdx.x, tid = threadIdx.x, nThreads = blockDim.x;
    if (bin >= V.nBins) return;
    const u32 b0 = V.binStart[bin], Pc = (V.binStart[bin + 1] - b0) / 2;
    PeLibrary L; L.length = 0; L.dev = 0; L.pairing = 1; L.nLens = 0;
    if (!V.optLibLength || !V.optLibDev)
    {
        u32 * lens = V.libLens + b0;
        u32 * kept = V.libIdx + b0;
        // unique, properly oriented pairs in pair order (reference src/mapper.h:931-947)
        u32 cnt = 0;
        for (u32 base = 0; base < Pc; base += nThreads)
        {
            const u32 j = base + tid;
            bool take = false; u32 len = 0;
            if (j < Pc)
            {
                const PeItem a = V.items[b0 + j], b = V.items[b0 + Pc + j];
                if (a.nOpt == 1 && b.nOpt == 1)
                {
                    const RawMatch fm = V.u2[a.u2Off + a.primary], sm = V.u2[b.u2Off + b.primary];
                    if (fm.contig == sm.contig && peOrientationProper(fm, sm)) { take = true; len = peLibraryLength(fm, sm); }
                }
            }
            const u32 slot = blockPushOrdered(take, cnt, waveTotals);
            if (take) lens[slot] = len;
        }
        __threadfence_block(); __syncthreads();
        if (cnt == 0) L.pairing = 0;
        else
        {
            // median = element cnt/2 of the sorted lengths (nthElement, :952): radix select, 11 + 11 + 10 bits
            if (tid == 0) { sel[0] = 0; sel[1] = cnt / 2; }
            const u32 shifts[3] = {21, 10, 0}, widths[3] = {11, 11, 10};
            u32 prefixMask = 0;
            for (u32 pass = 0; pass < 3; ++pass)
            {
                for (u32 i = tid; i < 2048; i += nThreads) hist[i] = 0;
                __syncthreads();
                const u32 prefix = sel[0];
                for (u32 i = tid; i < cnt; i += nThreads)
                {
                    const u32 v = lens[i];
                    if ((v & prefixMask) == prefix) atomicAdd(&hist[(v >> shifts[pass]) & ((1u << widths[pass]) - 1)], 1u);
                }
                __syncthreads();
                if (tid == 0)
                {
                    u32 rank = sel[1], d = 0;
                    for (;; ++d) { if (rank < hist[d]) break; rank -= hist[d]; }
                    sel[0] = prefix | (d << shifts[pass]); sel[1] = rank;
                }
                prefixMask |= ((1u << widths[pass]) - 1) << shifts[pass];
                __syncthreads();
            }
            const u32 median = sel[0];
            // removeIf(> 6.0 * median), order kept (:953)
            u32 nKept = 0;
            for (u32 base = 0; base < cnt; base += nThreads)
            {
                const u32 i = base + tid;
                const u32 v = i < cnt ? lens[i] : 0;
                const bool keep = i < cnt && !((double)v > 6.0 * (double)median);
                const u32 slot = blockPushOrdered(keep, nKept, waveTotals);
                if (keep) kept[slot] = v;
            }
            __threadfence_block(); __syncthreads();
            if (nKept == 0) L.pairing = 0;
            else
            {
                // the sum is an unsigned accumulate (order-free); the squared deviations are summed in float in pair order like
                // the single-threaded reference (:959-967): chunks are staged in LDS by the block, one thread walks them
                if (tid == 0) sel[2] = 0;
                __syncthreads();
                u32 part = 0;
                for (u32 i = tid; i < nKept; i += nThreads) part += kept[i];
                atomicAdd(&sel[2], part);
                __syncthreads();
                float mean = (float)sel[2] / (float)nKept; if (!(mean > 1.0f)) mean = 1.0f;
                float sq = 0.0f;
                for (u32 base = 0; base < nKept; base += 4096)
                {
                    const u32 m = min(4096u, nKept - base);
                    for (u32 i = tid; i < m; i += nThreads) chunk[i] = (float)kept[base + i];
                    __syncthreads();
                    if (tid == 0)
                    {   // only the additions form the chain: the squares of eight elements are formed ahead of it
                        u32 i = 0;
                        for (; i + 8 <= m; i += 8)
                        {
                            float q[8];
#pragma unroll
                            for (int k = 0; k < 8; ++k) { const float d = chunk[i + k] - mean; q[k] = d * d; }
#pragma unroll
                            for (int k = 0; k < 8; ++k) sq += q[k];
                        }
                        for (; i < m; ++i) { const float d = chunk[i] - mean; sq += d * d; }
                    }
                    __syncthreads();
                }
                if (tid == 0)
                {
                    float dev = sqrtf(sq / (float)nKept); if (!(dev > 1.0f)) dev = 1.0f;
                    L.length = (u32)mean; L.dev = (u32)dev; L.nLens = nKept;
                }
            }
        }
    }
    if (tid == 0)
    {
        if (L.pairing)
        {
            if (V.optLibLength) L.length = V.optLibLength;
            if (V.optLibDev) L.dev = V.optLibDev;
        }
        V.lib[bin] = L;
    }
}

// findProperMates (reference src/bits_matches.h:819-867): [first,last) in the mate's coordinate-sorted unique matches
__device__ inline void peFindProperMates(const RawMatch * mates, u32 nMates, const RawMatch & match, u32 mateLength, u32 contigLength, u32 mean, u32 stdDev,
                                         u32 & first, u32 & last)
{
    RawMatch leq = match; leq.strand = match.strand ? 0 : 1;
    RawMatch geq = leq;
    u32 leqErr = 0, geqErr = MAX_ERRORS;
    const int ml = (int)mateLength;
    int dm = (int)mean - (int)(6 * stdDev) - ml; if (dm < 0) dm = 0;
    int dp = (int)mean + (int)(6 * stdDev) - ml; if (dp < 0) dp = 0;
    const u32 deltaMinus = (u32)dm, deltaPlus = (u32)dp;
    auto addPos = [&](RawMatch & m, u32 delta) { m.begin = ((u64)m.begin + m.len + delta < contigLength) ? m.begin + delta : contigLength - m.len; };
    auto subPos = [&](RawMatch & m, u32 delta) { m.begin = m.begin > delta ? m.begin - delta : 0; };
    if (!match.strand) { addPos(leq, deltaMinus); addPos(geq, deltaPlus); }
    else { subPos(leq, deltaPlus); subPos(geq, deltaMinus); }
    // lower_bound by (contig, strand, begin, errors)
    auto lessB = [](const RawMatch & a, u32 ea, const RawMatch & b, u32 eb) {
        if (a.contig != b.contig) return a.contig < b.contig;
        if (a.strand != b.strand) return a.strand < b.strand;
        if (a.begin != b.begin) return a.begin < b.begin;
        return ea < eb; };
    auto lessE = [](const RawMatch & a, u32 ea, const RawMatch & b, u32 eb) {
        if (a.contig != b.contig) return a.contig < b.contig;
        if (a.strand != b.strand) return a.strand < b.strand;
        if (mEnd(a) != mEnd(b)) return mEnd(a) < mEnd(b);
        return ea < eb; };
    u32 lo = 0, hi = nMates;
    while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (lessB(mates[mid], mates[mid].errors, leq, leqErr)) lo = mid + 1; else hi = mid; }
    first = lo;
    lo = 0; hi = nMates;
    while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (!lessE(geq, geqErr, mates[mid], mates[mid].errors)) lo = mid + 1; else hi = mid; }
    last = lo;
    if (first > last) { first = 0; last = 0; }
}

// findPrimaryMatch (reference src/bits_matches.h:972-1041): weight of first-mate match f summed over its proper mates, in the mates' order
__device__ inline double pePairWeight(const RawMatch & f, const RawMatch * sm, u32 nS, double firstOpt, double secondOpt, u32 secondOptCount, u32 secondSubCount,
                                      u32 firstLen, u32 secondLen, const u64 * limits, u32 mean, u32 stdDev)
{
    double w = 0.0;
    u64 optImproper = secondOptCount, subImproper = secondSubCount;
    u32 first, last;
    peFindProperMates(sm, nS, f, secondLen, (u32)(limits[f.contig + 1] - limits[f.contig]), mean, stdDev, first, last);
    for (u32 k = first; k < last; ++k)
    {
        const RawMatch s = sm[k];
        const double secondErrorRate = (double)((float)s.errors / (float)secondLen);
        w += peMatchWeight(secondErrorRate, secondOpt) * peLibraryProb(f, s, mean, stdDev);
        if (secondErrorRate == secondOpt) --optImproper;
        else if (secondErrorRate == secondOpt + 0.01) --subImproper;
    }
    w += ((double)optImproper * peMatchWeight(secondOpt, secondOpt) + (double)subImproper * peMatchWeight(secondOpt + 0.01, secondOpt) + peResidualWeight(secondOpt)) * 0.0009;
    const double firstErrorRate = (double)((float)f.errors / (float)firstLen);
    w *= peMatchWeight(firstErrorRate, firstOpt);
    return w;
}

__device__ inline bool peFindPrimary(const RawMatch * fm, u32 nF, const RawMatch * sm, u32 nS, double firstOpt, double secondOpt, u32 secondOptCount, u32 secondSubCount,
                                     u32 firstLen, u32 secondLen, const u64 * limits, u32 mean, u32 stdDev, u32 & bestIdx, double & prob)
{
    double sum = 0.0, best = 0.0; bool found = false;
    for (u32 i = 0; i < nF; ++i)
    {
        const double w = pePairWeight(fm[i], sm, nS, firstOpt, secondOpt, secondOptCount, secondSubCount, firstLen, secondLen, limits, mean, stdDev);
        if (w > best) { best = w; bestIdx = i; found = true; }
        sum += w;
    }
    prob = best / sum;
    return found;
}

// Same for one wavefront: the weights of 64 first-mate matches are computed side by side, then folded in index order (the sum and
// the first maximum are order-dependent in fp64, so the fold is sequential); every lane ends with the same result.
__device__ inline bool peFindPrimaryWave(const RawMatch * fm, u32 nF, const RawMatch * sm, u32 nS, double firstOpt, double secondOpt, u32 secondOptCount, u32 secondSubCount,
                                         u32 firstLen, u32 secondLen, const u64 * limits, u32 mean, u32 stdDev, u32 & bestIdx, double & prob, u32 lane)
{
    double sum = 0.0, best = 0.0; bool found = false;
    for (u32 base = 0; base < nF; base += 64)
    {
        const u32 i = base + lane;
        double w = 0.0;
        if (i < nF) w = pePairWeight(fm[i], sm, nS, firstOpt, secondOpt, secondOptCount, secondSubCount, firstLen, secondLen, limits, mean, stdDev);
        const u32 cnt = nF - base < 64 ? nF - base : 64;
        for (u32 l = 0; l < cnt; ++l)
        {
            const double wl = __shfl(w, (int)l, 64);
            if (wl > best) { best = wl; bestIdx = base + l; found = true; }
            sum += wl;
        }
    }
    prob = best / sum;
    return found;
}

// ---- stage 3: one thread per pair of a bin; pairs with many unique matches go to a list for k_pe_pair_heavy -------------------------
__global__ void k_pe_pair(FinalizeArgs A, PeView V, u32 heavyMin, u32 * __restrict__ heavyList, u32 * __restrict__ heavyCount)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= V.nItems) return;
    const u32 bin = V.itemBin[t];
    const u32 b0 = V.binStart[bin], Pc = (V.binStart[bin + 1] - b0) / 2;
    if (t - b0 >= Pc) return;
    const PeLibrary L = V.lib[bin];
    if (!L.pairing) return;
    PeItem & a = V.items[t];
    PeItem & b = V.items[t + Pc];
    if (!a.nU2 || !b.nU2) return;
    if (a.nU2 + b.nU2 > heavyMin) { heavyList[atomicAdd(heavyCount, 1u)] = t; return; }
    const u32 ra = V.itemRead[t], rb = V.itemRead[t + Pc];
    const u32 aLen = (u32)(A.readOffs[ra + 1] - A.readOffs[ra]), bLen = (u32)(A.readOffs[rb + 1] - A.readOffs[rb]);
    const double aOpt = (double)((float)a.minErr / (float)aLen), bOpt = (double)((float)b.minErr / (float)bLen);
    const u64 * limits = A.limits + A.bins[bin].limOff;
    u32 ia = 0, ib = 0; double pa = 0.0, pb = 0.0;
    const bool fa = peFindPrimary(V.u2 + a.u2Off, a.nU2, V.u2 + b.u2Off, b.nU2, aOpt, bOpt, b.nOpt, b.nSet - b.nOpt, aLen, bLen, limits, L.length, L.dev, ia, pa);
    const bool fb = peFindPrimary(V.u2 + b.u2Off, b.nU2, V.u2 + a.u2Off, a.nU2, bOpt, aOpt, a.nOpt, a.nSet - a.nOpt, bLen, aLen, limits, L.length, L.dev, ib, pb);
    if (!fa || !fb) return;
    a.primary = ia; b.primary = ib; a.prob = pa; b.prob = pb; a.paired = 1; b.paired = 1;
}

// one wavefront per listed pair (repeat-rich mates: thousands of unique matches each)
__global__ __launch_bounds__(256) void k_pe_pair_heavy(FinalizeArgs A, PeView V, const u32 * __restrict__ heavyList, const u32 * __restrict__ heavyCount)
{
    const u32 lane = threadIdx.x & 63;
    const u32 gwave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nWaves = (gridDim.x * blockDim.x) >> 6;
    const u32 nList = *heavyCount;
    for (u32 li = gwave; li < nList; li += nWaves)
    {
        const u32 t = heavyList[li];
        const u32 bin = V.itemBin[t];
        const u32 b0 = V.binStart[bin], Pc = (V.binStart[bin + 1] - b0) / 2;
        const PeLibrary L = V.lib[bin];
        PeItem & a = V.items[t];
        PeItem & b = V.items[t + Pc];
        const u32 ra = V.itemRead[t], rb = V.itemRead[t + Pc];
        const u32 aLen = (u32)(A.readOffs[ra + 1] - A.readOffs[ra]), bLen = (u32)(A.readOffs[rb + 1] - A.readOffs[rb]);
        const double aOpt = (double)((float)a.minErr / (float)aLen), bOpt = (double)((float)b.minErr / (float)bLen);
        const u64 * limits = A.limits + A.bins[bin].limOff;
        u32 ia = 0, ib = 0; double pa = 0.0, pb = 0.0;
        const bool fa = peFindPrimaryWave(V.u2 + a.u2Off, a.nU2, V.u2 + b.u2Off, b.nU2, aOpt, bOpt, b.nOpt, b.nSet - b.nOpt, aLen, bLen, limits, L.length, L.dev, ia, pa, lane);
        const bool fb = peFindPrimaryWave(V.u2 + b.u2Off, b.nU2, V.u2 + a.u2Off, a.nU2, bOpt, aOpt, a.nOpt, a.nSet - a.nOpt, bLen, aLen, limits, L.length, L.dev, ib, pb, lane);
        if (fa && fb && lane == 0) { a.primary = ia; b.primary = ib; a.prob = pa; b.prob = pb; a.paired = 1; b.paired = 1; }
    }
}

// ---- stage 4: mate rescue -------------------------------------------------------------------------------------------
__global__ void k_pe_rescue_jobs(PeView V, u32 * __restrict__ jobs, u32 * __restrict__ jobCount)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= V.nItems || !V.verify) return;
    const u32 bin = V.itemBin[t];
    const u32 b0 = V.binStart[bin], Pc = (V.binStart[bin + 1] - b0) / 2;
    const u32 mate = t - b0 < Pc ? t + Pc : t - Pc;
    if (V.items[t].nOpt == 1 && V.items[mate].nU2 == 0) jobs[atomicAdd(jobCount, 1u)] = t;     // anchor item
}

// Unbanded semi-global affine DP of the mate against the insert window, one wavefront per job: lane = window column
// (64 columns per pass), the horizontal gap matrix by a max-plus prefix scan like alignMatchWave.
// scratch per wave: trace (m+1)*(n+1) bytes | rows S,V (2 x (n+1) ints, double buffered) | read m bytes | window n bytes
__host__ __device__ inline size_t peRescueScratchBytes(u32 maxReadLen, u32 maxWindow)
{
    return (size_t)(maxReadLen + 1) * (maxWindow + 1) + 4 * (size_t)(maxWindow + 1) * sizeof(int) + maxReadLen + maxWindow + 64;
}

// LDS bytes per wavefront for the rows + read + window of k_pe_rescue (the trace stays in the global slab)
__host__ __device__ inline size_t peRescueLdsBytes(u32 maxReadLen, u32 maxWindow)
{
    return ((4 * (size_t)(maxWindow + 1) * sizeof(int) + maxReadLen + maxWindow + 15) & ~(size_t)15);
}

__global__ __launch_bounds__(256) void k_pe_rescue(FinalizeArgs A, PeView V, const u32 * __restrict__ jobs, const u32 * __restrict__ jobCount,
                                                   u8 * __restrict__ scratch, size_t scratchStride, u32 maxWindow, u32 ldsPerWave)
{
    extern __shared__ __attribute__((aligned(16))) u8 rescueLds[];
    const u32 lane = threadIdx.x & 63;
    const u32 gwave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const u32 nWaves = (gridDim.x * blockDim.x) >> 6;
    u8 * my = scratch + (size_t)gwave * scratchStride;
    const int NEG = -(1 << 28), MISMATCH = -1000, GEXT = -999, GOPEN = -1001;
    const u32 nJobs = *jobCount;
    for (u32 jb = gwave; jb < nJobs; jb += nWaves)
    {
        const u32 t = jobs[jb];
        const u32 bin = V.itemBin[t];
        const u32 b0 = V.binStart[bin], Pc = (V.binStart[bin + 1] - b0) / 2;
        const u32 mt = t - b0 < Pc ? t + Pc : t - Pc;
        const PeLibrary L = V.lib[bin];
        const PeItem anchorIt = V.items[t];
        const RawMatch anchor = V.u2[anchorIt.u2Off + anchorIt.primary];     // nOpt == 1 and the mate has no match: the unique optimal match
        const u32 anchorRead = V.itemRead[t], mateRead = V.itemRead[mt];
        const u32 mateRev = anchor.strand ? 0 : 1;                             // the mate is searched on the opposite strand
        const int m = (int)(A.readOffs[mateRead + 1] - A.readOffs[mateRead]);
        const u8 * mseq = A.readSeqs + A.readOffs[mateRead];
        const BinDesc & bd = A.bins[bin];
        const u64 * limits = A.limits + bd.limOff;
        const u64 contigLength = limits[anchor.contig + 1] - limits[anchor.contig];
        u64 wb = 0, we = 0;
        if (mateRev)
        {   // --> ... mate   (RightMate, reference src/mapper_verifier.h:218-243)
            if ((u64)anchor.begin + L.length > 3ull * L.dev) wb = (u64)anchor.begin + L.length - 3ull * L.dev;
            if (wb > contigLength) wb = contigLength;
            we = (u64)anchor.begin + L.length + 3ull * L.dev; if (we > contigLength) we = contigLength;
        }
        else
        {   // mate ... <--   (LeftMate, :246-273)
            const u64 aend = mEnd(anchor);
            if (aend > (u64)L.length + 3ull * L.dev) wb = aend - L.length - 3ull * L.dev;
            if (aend + L.dev > L.length) we = aend - L.length + 3ull * L.dev;
            if (we > contigLength) we = contigLength;
        }
        if (we <= wb || we - wb > maxWindow) continue;
        const int n = (int)(we - wb);
        const u8 * ctg = A.contigs + bd.contigOff + limits[anchor.contig] + wb;
        u8 * trace = my;
        // rows, read and window: in LDS when the launch reserved room for them (ldsPerWave > 0), else behind the trace in the global slab
        int * rowS0 = ldsPerWave ? (int *)(rescueLds + (size_t)(threadIdx.x >> 6) * ldsPerWave)
                                 : (int *)(my + (((size_t)(m + 1) * (n + 1) + 3) & ~(size_t)3));
        int * rowV0 = rowS0 + (n + 1), * rowS1 = rowV0 + (n + 1), * rowV1 = rowS1 + (n + 1);
        u8 * sRead = (u8 *)(rowV1 + (n + 1));
        u8 * sCtg = sRead + m;
        for (int x = (int)lane; x < m; x += 64) sRead[x] = (u8)readBase(mseq, (u32)m, mateRev, (u32)x);
        for (int x = (int)lane; x < n; x += 64) sCtg[x] = ctg[x];
        for (int j = (int)lane; j <= n; j += 64) { rowS0[j] = 0; rowV0[j] = NEG; trace[j] = 0; }
        waveFence();
        int * pS = rowS0, * pV = rowV0, * cS = rowS1, * cV = rowV1;
        for (int i = 1; i <= m; ++i)
        {
            const u32 rc = sRead[i - 1];
            int carryScan = NEG, carryS = NEG, carryH = NEG;
            for (int base = 0; base <= n; base += 64)
            {
                const int j = base + (int)lane;
                const bool valid = j <= n;
                int v = NEG; u8 tr = 0;
                if (valid)
                {
                    const int ext = pV[j] + GEXT, opn = pS[j] + GOPEN;
                    if (ext >= opn) { v = ext; tr |= 4; } else v = opn;
                }
                int tt = NEG; u8 tb = 0;
                if (valid && j >= 1)
                {
                    const u32 cc = sCtg[j - 1];
                    tt = pS[j - 1] + ((cc == rc && cc < 4) ? 0 : MISMATCH); tb = 1;
                }
                if (v > tt) { tt = v; tb = 3; }
                // horizontal matrix: H[j] = max_{j'<j} (T[j'] + open + ext*(j-j'-1)); exclusive prefix max of U = T + 999 j
                const int u = valid ? tt - GEXT * j : NEG;
                const int incl = dppWaveMaxScan(u);
                const int excl = dppWaveShr1(incl, NEG);
                const int scan = excl > carryScan ? excl : carryScan;
                int h = NEG;
                if (valid && j >= 1) h = scan + GOPEN + GEXT * (j - 1);
                int sCell = tt;
                if (h > sCell) { sCell = h; tb = 2; }
                if (!valid) { sCell = NEG; h = NEG; v = NEG; }
                const int leftS = dppWaveShr1(sCell, carryS), leftH = dppWaveShr1(h, carryH);
                if (valid && j >= 1 && leftH + GEXT >= leftS + GOPEN) tr |= 8;
                const int lastIncl = __builtin_amdgcn_readlane(incl, 63);
                carryScan = lastIncl > carryScan ? lastIncl : carryScan;
                carryS = __builtin_amdgcn_readlane(sCell, 63); carryH = __builtin_amdgcn_readlane(h, 63);
                if (valid) { cS[j] = sCell; cV[j] = v; trace[(size_t)i * (n + 1) + j] = (u8)(tr | tb); }
            }
            if (ldsPerWave) waveFenceLds(); else waveFence();      // the rows live in LDS (or in the global slab)
            int * tS = pS; pS = cS; cS = tS; int * tV = pV; pV = cV; cV = tV;
        }
        waveFence();                                                // the trace is read back from the global slab
        // best cell of the last row (pS), leftmost on ties
        int bs = NEG, bj = -1;
        for (int base = 0; base <= n; base += 64)
        {
            const int j = base + (int)lane;
            const int val = j <= n ? pS[j] : NEG;
            int mx = val;
            for (int o = 32; o >= 1; o >>= 1) { const int other = __shfl_xor(mx, o, 64); mx = other > mx ? other : mx; }
            if (mx > bs) { bs = mx; bj = base + (int)(__ffsll((long long)__ballot(j <= n && val == mx)) - 1); }
        }
        // traceback (uniform), counting gap runs and gap characters of the clipped alignment
        // The walk is a chain of dependent reads of the trace in the global slab (~2 us each: it was most of this kernel's time).  So 64 rows
        // are fetched at once: lane L holds the 8 cells of row rowHi - L around the diagonal through the current cell, and the walk reads
        // them with v_readlane; a path that leaves the 8-cell corridor (a long gap) or the 64 rows refetches.
        int i = m, j = bj, state = 0, lastOp = -1;
        u32 gapOpens = 0, gapChars = 0;
        const bool corridor = n + 1 >= 8;
        int rowHi = -1, jHi = 0; u32 cellsLo = 0, cellsHi = 0;
        auto corridorStart = [&](int row) -> int { int c = jHi - (rowHi - row) - 4; if (c < 0) c = 0; if (c > n + 1 - 8) c = n + 1 - 8; return c; };
        while (i > 0)
        {
            u8 tr;
            if (corridor)
            {
                int c0 = rowHi >= 0 ? corridorStart(i) : 0;
                if (rowHi < 0 || i > rowHi || rowHi - i > 63 || j < c0 || j >= c0 + 8)
                {
                    rowHi = i; jHi = j;
                    const int row = i - (int)lane;
                    u64 cells = 0;
                    if (row >= 0) cells = load8(trace + (size_t)row * (n + 1) + corridorStart(row));
                    cellsLo = (u32)cells; cellsHi = (u32)(cells >> 32);
                    c0 = corridorStart(i);
                }
                const int L = __builtin_amdgcn_readfirstlane(rowHi - i), at = __builtin_amdgcn_readfirstlane(j - c0);
                const u32 word = at < 4 ? (u32)__builtin_amdgcn_readlane((int)cellsLo, L) : (u32)__builtin_amdgcn_readlane((int)cellsHi, L);
                tr = (u8)(word >> (8 * (at & 3)));
            }
            else tr = trace[(size_t)i * (n + 1) + j];
            int op;
            if (state == 0)
            {
                const u8 tb = tr & 3;
                if (tb == 1) { op = 0; --i; --j; }
                else if (tb == 3) { state = 1; continue; }
                else if (tb == 2) { state = 2; continue; }
                else break;
            }
            else if (state == 1) { op = 1; --i; if (!(tr & 4)) state = 0; }
            else { op = 2; --j; if (!(tr & 8)) state = 0; }
            if (op != 0) { ++gapChars; if (op != lastOp) ++gapOpens; }
            lastOp = op;
        }
        const int errors = bs / -999;
        const u32 gapExtensions = gapChars - gapOpens;
        const u32 events = (u32)errors + gapOpens - gapExtensions;
        const u32 maxErrors = readErrorsOf((u32)m, V.errorRate);
        u32 maxIndels = (u32)((float)m * V.indelRate); if (maxIndels > MAX_ERRORS) maxIndels = MAX_ERRORS;
        if (events <= maxErrors && gapChars <= maxIndels && lane == 0)
        {
            PeItem & mateIt = V.items[mt];
            RawMatch mm; mm.read = mateRead; mm.begin = (u32)(wb + (u64)j); mm.len = (u16)(bj - j); mm.errors = (u8)errors; mm.strand = (u8)mateRev;
            mm.bin = (u16)bin; mm.contig = anchor.contig;
            mateIt.mate = mm; mateIt.rescued = 1; mateIt.paired = 1;
            const u32 aLen = (u32)(A.readOffs[anchorRead + 1] - A.readOffs[anchorRead]);
            const double rate = (double)((float)anchor.errors / (float)aLen);
            const double pr = peMatchProb(rate, rate, anchorIt.nOpt, anchorIt.nSet - anchorIt.nOpt);
            mateIt.prob = pr;
            V.items[t].paired = 1; V.items[t].prob = pr;
            V.readHasRescue[mateRead] = 1;
        }
        waveFence();
    }
}

// ---- stage 5: routing + final per-read aggregation ------------------------------------------------------------------
// One thread per read.  Unmapped reads and — the bulk of a real batch — reads whose whole evidence is ONE entry (a single raw
// match, or a single rescued mate, in the only bin the pair selected) are finished here; the child and main mappers' sorts
// and compactions are identities on one entry.  A primary that needs the alignment DP, and every read with more entries,
// goes to the list of k_pe_finalize.
__global__ __launch_bounds__(1024) void k_pe_route(FinalizeArgs A, PeView V, const u32 * __restrict__ mStart, u32 * __restrict__ list, u32 * __restrict__ listCount,
                           u32 * __restrict__ alignList, u32 * __restrict__ alignCount, u32 * __restrict__ smallList, u32 * __restrict__ smallCount)
{
    const u32 read = blockIdx.x * blockDim.x + threadIdx.x;
    bool complex = false, align = false, small = false;
    if (read < A.nReads)
    {
        const u32 s0 = A.readStart[read], n = A.readStart[read + 1] - s0;
        const bool hasRescue = V.readHasRescue[read] != 0;
        if (n == 0 && !hasRescue)
        {
            ReadOut ro; initReadOut(ro, 3 * s0, read * A.cigarStride);
            A.out[read] = ro;
        }
        else
        {
            complex = true;
            u32 nSel = 0, bin = 0;
            for (u32 w = 0; w < V.W; ++w)
            {
                const u64 m = V.masks[(size_t)read * V.W + w];
                if (m) bin = w * 64 + (u32)(__ffsll((long long)m) - 1);
                nSel += (u32)__popcll(m);
            }
            if (nSel == 1 && n >= 2 && n <= SMALL_T && !hasRescue)
            {   // a handful of raw matches in the only selected bin: thread-level aggregation (k_pe_finalize_small)
                const u32 readLen = (u32)(A.readOffs[read + 1] - A.readOffs[read]);
                if (readLen <= THREAD_DP_MAX_LEN && readErrorsOf(readLen, A.errorRate) <= THREAD_DP_MAX_E) { complex = false; small = true; }
            }
            if (nSel == 1 && n <= 1)
            {
                const PeItem it = V.items[peFindItem(V, read, bin)];
                RawMatch prim; bool ok = false;
                if (n == 1) { prim = A.gp[s0]; ok = prim.bin == bin && it.primary == 0; }
                else if (it.rescued) { prim = it.mate; ok = true; }
                if (ok)
                {
                    const u32 readLen = (u32)(A.readOffs[read + 1] - A.readOffs[read]);
                    bool plain = prim.errors == 0 || (prim.errors == 1 && prim.len == readLen);
                    if (!plain && prim.errors == 2 && prim.len == readLen && readLen >= 8)
                    {
                        const BinDesc & bd = A.bins[bin];
                        const u8 * ctg = A.contigs + bd.contigOff + A.limits[bd.limOff + prim.contig] + prim.begin;
                        plain = alignIsUngapped2(ctg, (prim.strand ? A.readSeqsRc : A.readSeqs) + A.readOffs[read], (int)readLen);
                    }
                    if (plain)
                    {
                        ReadOut ro; initReadOut(ro, mStart[read], read * A.cigarStride);
                        A.cigars[ro.cigarBegin] = (readLen << 4) | 0; ro.cigarLen = 1;
                        ro.paired = it.paired ? 1 : 0; ro.pairProb = it.paired ? it.prob : 0.0;
                        writeSingleMatch(A, read, ro, prim);
                        complex = false;
                    }
                    else if (readLen <= THREAD_DP_MAX_LEN && prim.errors <= THREAD_DP_MAX_E) { complex = false; align = true; }
                }
            }
        }
    }
    const bool want[3] = {complex, align, small};
    u32 * const lists[3] = {list, alignList, smallList};
    u32 * const counts[3] = {listCount, alignCount, smallCount};
    blockAppend<3>(want, read, lists, counts);
}

// Reads with 2..SMALL_T raw matches, all in the only selected bin: one thread each (see finalizeSmallRead); the child's primary
// is the one the pairing stage left in PeItem::primary.
template <int EMAX, typename TRow>
__global__ __launch_bounds__(64) void k_pe_finalize_small(FinalizeArgs A, PeView V, const u32 * __restrict__ mStart, const u32 * __restrict__ smallList,
                                                          const u32 * __restrict__ smallCount)
{
    extern __shared__ u64 dpTraceLds[];
    TRow * trace = (TRow *)dpTraceLds + threadIdx.x;
    const u32 nList = *smallCount;
    for (u32 li = blockIdx.x * blockDim.x + threadIdx.x; li < nList; li += gridDim.x * blockDim.x)
    {
        const u32 read = smallList[li];
        u32 bin = 0;
        for (u32 w = 0; w < V.W; ++w) { const u64 m = V.masks[(size_t)read * V.W + w]; if (m) bin = w * 64 + (u32)(__ffsll((long long)m) - 1); }
        const PeItem it = V.items[peFindItem(V, read, bin)];
        ReadOut ro; initReadOut(ro, mStart[read], read * A.cigarStride);
        ro.paired = it.paired ? 1 : 0; ro.pairProb = it.paired ? it.prob : 0.0;
        finalizeSmallRead(A, read, ro, ThreadAligner<EMAX, TRow>{trace, blockDim.x}, true, [&](u32, u32) { return SmallPrimary{false, it.primary}; });
    }
}

// Single-entry reads whose primary needs the alignment DP (thread-level DP, all lanes busy)
template <int EMAX, typename TRow>
__global__ __launch_bounds__(64) void k_pe_align_single(FinalizeArgs A, PeView V, const u32 * __restrict__ mStart, const u32 * __restrict__ alignList,
                                                        const u32 * __restrict__ alignCount)
{
    extern __shared__ u64 dpTraceLds[];
    TRow * trace = (TRow *)dpTraceLds + threadIdx.x;
    const u32 nList = *alignCount;
    for (u32 li = blockIdx.x * blockDim.x + threadIdx.x; li < nList; li += gridDim.x * blockDim.x)
    {
        const u32 read = alignList[li];
        const u32 s0 = A.readStart[read], n = A.readStart[read + 1] - s0;
        const u32 readLen = (u32)(A.readOffs[read + 1] - A.readOffs[read]);
        u32 bin = 0;
        for (u32 w = 0; w < V.W; ++w) { const u64 m = V.masks[(size_t)read * V.W + w]; if (m) bin = w * 64 + (u32)(__ffsll((long long)m) - 1); }
        const PeItem it = V.items[peFindItem(V, read, bin)];
        RawMatch a = n == 1 ? A.gp[s0] : it.mate;
        ReadOut ro; initReadOut(ro, mStart[read], read * A.cigarStride);
        const BinDesc & bd = A.bins[bin];
        const u8 * ctg = A.contigs + bd.contigOff + A.limits[bd.limOff + a.contig] + a.begin;
        int cb, ce;
        const u32 nc = alignMatchThread<EMAX, TRow>(ctg, (int)a.len, A.readSeqs + A.readOffs[read], readLen, a.strand, (int)a.errors,
                                                    A.cigars + ro.cigarBegin, A.cigarStride, cb, ce, trace, blockDim.x);
        ro.cigarLen = nc < A.cigarStride ? nc : A.cigarStride;
        a.begin += (u32)cb; a.len = (u16)ce;
        ro.paired = it.paired ? 1 : 0; ro.pairProb = it.paired ? it.prob : 0.0;
        writeSingleMatch(A, read, ro, a);
    }
}

__global__ __launch_bounds__(256) void k_pe_finalize(FinalizeArgs A, PeView V, const u32 * __restrict__ mStart)
{
    const u32 lane = threadIdx.x & 63;
    const u32 gwave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const u32 nWaves = (gridDim.x * blockDim.x) >> 6;
    u8 * dp = A.dpScratch + (size_t)gwave * A.dpStride;
    const u32 nList = *A.complexCount;
    for (u32 li = gwave; li < nList; li += nWaves)
    {
        const u32 read = A.complexList[li];
        const u32 s0 = A.readStart[read], n = A.readStart[read + 1] - s0;
        const u32 readLen = (u32)(A.readOffs[read + 1] - A.readOffs[read]);
        const u8 * seq = A.readSeqs + A.readOffs[read];
        const u32 strata = readStrataOf(readLen, A.strataRate);
        const u32 ms = mStart[read], mCap = mStart[read + 1] - ms;
        ReadOut ro; initReadOut(ro, ms, read * A.cigarStride);
        const RawMatch * E = A.gp + s0;
        u32 * idxA = A.aux + (size_t)4 * s0;
        u32 * idxB = idxA + n;
        RawMatch * M = A.marr + ms;
        u32 * idxM = A.maux + (size_t)2 * ms;
        u32 * idxF = idxM + mCap;
        for (u32 i = lane; i < n; i += 64) idxA[i] = i;
        waveFence();
        waveSort(idxA, n, lane, [&](u32 a, u32 b) { return lessEnd(E[a], a, E[b], b); });
        u32 nM = 0, winErr = 256, winRank = 0, pairedRank = 0xFFFFFFFFu, p0 = 0;
        double pairProb = 0.0;
        for (u32 w = 0; w < V.W; ++w)
        {
            u64 mask = V.masks[(size_t)read * V.W + w];
            while (mask)
            {
                const u32 bin = w * 64 + (u32)(__ffsll((long long)mask) - 1);
                mask &= mask - 1;
                const PeItem it = V.items[peFindItem(V, read, bin)];
                const u32 binRank = A.binOrderRank[bin] + 1;
                u32 p1 = p0;
                if (p0 < n && E[idxA[p0]].bin == bin)
                {
                    u32 cntBin = 0;
                    for (u32 p = p0 + lane; p < n; p += 64) cntBin += E[idxA[p]].bin == bin;
                    p1 = p0 + waveSum32(cntBin);
                }
                bool hasPrimary = false; u32 primErr = 0;
                RawMatch prim; u32 primSlot = 0;
                if (p1 > p0)
                {
                    const ChildSeg seg = peChildPasses(E, idxA, idxB, p0, p1, lane, M, nM, true, strata);
                    u32 firstSeen = 0;
                    for (u32 base = 0; base < seg.n1; base += 64)
                    {
                        const u32 q = base + lane;
                        RawMatch a; bool first = false;
                        if (q < seg.n1)
                        {
                            a = E[idxB[q]];
                            first = q == 0;
                            if (!first) { const RawMatch b = E[idxB[q - 1]]; first = !(sameBinContigStrand(a, b) && a.begin == b.begin); }
                        }
                        const u64 balFirst = __ballot(first);
                        const u32 myFirstRank = firstSeen + (u32)__popcll(balFirst & ((1ull << lane) - 1));
                        const u64 balPrim = __ballot(first && myFirstRank == it.primary);
                        const u32 mySlot = wavePush(first, lane, nM);
                        if (first) M[mySlot] = a;
                        if (balPrim)
                        {
                            const u32 pl = __ffsll((long long)balPrim) - 1;
                            hasPrimary = true;
                            prim.begin = __shfl(a.begin, pl, 64); prim.len = (u16)__shfl((u32)a.len, pl, 64); prim.errors = (u8)__shfl((u32)a.errors, pl, 64);
                            prim.strand = (u8)__shfl((u32)a.strand, pl, 64); prim.contig = (u16)__shfl((u32)a.contig, pl, 64); prim.bin = (u16)bin; prim.read = read;
                            primSlot = __shfl(mySlot, pl, 64);
                        }
                        firstSeen += (u32)__popcll(balFirst);
                    }
                }
                else if (it.rescued)
                {
                    hasPrimary = true; prim = it.mate; primSlot = nM;
                    if (lane == 0) M[nM] = prim;
                    ++nM;
                }
                waveFence();
                if (hasPrimary)
                {
                    primErr = prim.errors;
                    const bool winner = primErr < winErr || (primErr == winErr && binRank > winRank);
                    const bool gapped = !(prim.errors == 0 || (prim.errors == 1 && prim.len == readLen));
                    if (gapped)
                    {
                        const BinDesc & bd = A.bins[bin];
                        const u8 * ctg = A.contigs + bd.contigOff + A.limits[bd.limOff + prim.contig] + prim.begin;
                        int cb = 0, ce = (int)prim.len;
                        u32 nc;
                        if (2 * (u32)prim.errors + 1 <= 64)
                            nc = alignMatchWave(ctg, (int)prim.len, seq, readLen, prim.strand, (int)prim.errors, dp, winner ? A.cigars + ro.cigarBegin : nullptr, A.cigarStride, cb, ce, lane);
                        else
                        {
                            nc = 0;
                            if (lane == 0)
                                nc = alignMatchDevice(ctg, (int)prim.len, seq, readLen, prim.strand, (int)prim.errors, dp, winner ? A.cigars + ro.cigarBegin : nullptr, A.cigarStride, cb, ce);
                            nc = __shfl(nc, 0, 64); cb = __shfl(cb, 0, 64); ce = __shfl(ce, 0, 64);
                        }
                        if (winner) ro.cigarLen = nc < A.cigarStride ? nc : A.cigarStride;
                        if (lane == 0) { M[primSlot].begin = prim.begin + (u32)cb; M[primSlot].len = (u16)ce; }
                    }
                    else if (winner)
                    {
                        if (lane == 0) A.cigars[ro.cigarBegin] = (readLen << 4) | 0;
                        ro.cigarLen = 1;
                    }
                    if (winner) { winErr = primErr; winRank = binRank; }
                    if (it.paired && binRank < pairedRank) { pairedRank = binRank; pairProb = it.prob; }   // first paired child in processing order
                }
                waveFence();
                p0 = p1;
            }
        }
        // ---- main mapper: aggregateMatches + rankMatches2 over the union (same as k_finalize_t)
        for (u32 i = lane; i < nM; i += 64) idxM[i] = i;
        waveFence();
        waveSort(idxM, nM, lane, [&](u32 a, u32 b) { return lessEnd(M[a], a, M[b], b); });
        u32 m1 = 0;
        for (u32 base = 0; base < nM; base += 64)
        {
            const u32 p = base + lane;
            bool first = false;
            if (p < nM)
            {
                const RawMatch a = M[idxM[p]];
                first = p == 0;
                if (!first) { const RawMatch b = M[idxM[p - 1]]; first = !(sameBinContigStrand(a, b) && mEnd(a) == mEnd(b)); }
            }
            const u32 slot = wavePush(first, lane, m1);
            if (first) idxF[slot] = idxM[p];
        }
        waveFence();
        waveSort(idxF, m1, lane, [&](u32 a, u32 b) { return lessBegin(M[a], a, M[b], b); });
        u32 m2 = 0;
        for (u32 base = 0; base < m1; base += 64)
        {
            const u32 q = base + lane;
            bool first = false;
            if (q < m1)
            {
                const RawMatch a = M[idxF[q]];
                first = q == 0;
                if (!first) { const RawMatch b = M[idxF[q - 1]]; first = !(sameBinContigStrand(a, b) && a.begin == b.begin); }
            }
            const u32 slot = wavePush(first, lane, m2);
            if (first) idxM[slot] = idxF[q];
        }
        waveFence();
        waveSort(idxM, m2, lane, [&](u32 a, u32 b) { return lessErrors(M[a], a, M[b], b); });
        const u32 minE = M[idxM[0]].errors;
        u32 cOpt = 0, cSet = 0;
        for (u32 q = lane; q < m2; q += 64) { const u32 e = M[idxM[q]].errors; cOpt += e == minE; cSet += e <= minE + strata; }
        const u32 nOpt = waveSum32(cOpt), nSet = waveSum32(cSet);
        const u32 pick = libcxxFirstUniform(nOpt);
        for (u32 q = lane; q < nSet; q += 64)
        {
            const RawMatch a = M[idxM[q]];
            OutMatch om; om.contig = A.bins[a.bin].contigIdBase + a.contig; om.begin = a.begin; om.length = a.len; om.errors = a.errors; om.isRev = a.strand;
            A.outSet[(size_t)ro.setBegin + q] = om;
        }
        if (lane == 0)
        {
            const RawMatch a = M[idxM[pick]];
            ro.contig = A.bins[a.bin].contigIdBase + a.contig; ro.begin = a.begin; ro.length = a.len; ro.errors = a.errors; ro.isRev = a.strand;
            ro.mapped = a.len != 0; ro.bestCount = nOpt; ro.subCount = nSet - nOpt;
            ro.paired = pairedRank != 0xFFFFFFFFu; ro.pairProb = pairProb;
            A.out[read] = ro;
        }
        waveFence();
    }
}

} // namespace dy

// Per-read aggregation on the device: one wavefront per read walks the reference's post-extension stages
//   child mapper : aggregateMatches (removeDuplicates in two in-place passes, incl. the stale tails that copyMatches
//                  forwards), rankMatches (primary among co-optimal), alignMatches (banded affine DP, CIGAR, shrink)
//                  reference src/mapper.h:772-800,826-906,1179-1246, src/bits_matches.h:669-758, src/mapper_aligner.h:143-236
//   main mapper  : copyMatches/copyCigars, aggregateMatches, rankMatches2
//                  reference src/d_mapper.h:127-215,610-721,828-838
// Sorting is replaced by rank counting under TOTAL orders (the reference's std::sort leaves equal keys unordered; the
// total orders used here are (contig,strand,end,errors,begin) and (contig,strand,begin,errors,end)), so the result is
// independent of the order in which the extension kernel appended matches.  Cost is O(n^2/64) per read with n matches.
#pragma once
#include <hip/hip_runtime.h>
#include "dy_types.h"
#include "map_kernels.h"

namespace dy {

struct ReadOut            // device mirror of dy_read_result (mapq filled on the host)
{
    u32 contig, begin;    // primary: global contig id, begin
    u16 length; u8 errors; u8 isRev;
    u32 mapped, bestCount, subCount, mapq, setBegin, cigarBegin, cigarLen;
};

struct OutMatch { u32 contig; u32 begin; u16 length; u8 errors; u8 isRev; };

struct FinalizeArgs
{
    const RawMatch * gp;          // matches grouped by read
    const u32 * readStart;        // nReads + 1
    const u64 * readOffs;         // read offsets (for lengths)
    const u8 *  readSeqs;
    u32 nReads;
    const BinDesc * bins;
    const u8 * contigs; const u64 * limits;
    const u32 * binOrderRank;
    float strataRate;
    // scratch
    u32 * aux;                    // 4 u32 per raw match
    RawMatch * marr;              // 3 per raw match: the main mapper's matchesByCoord of this read
    u32 * maux;                   // 2 u32 per marr entry
    u8 *  dpScratch; size_t dpStride;   // per wave
    // outputs
    ReadOut * out;
    OutMatch * outSet;            // 3 per raw match (only the first nSet of a read's region are meaningful)
    u32 * cigars; u32 cigarStride;
};

// libc++ std::default_random_engine (minstd_rand, seed 1) + uniform_int_distribution<size_t>(0, n-1), first draw
// (reference src/mapper.h:886-906; gold SAMs were produced with libc++)
__host__ __device__ inline u32 libcxxFirstUniform(u32 n)
{
    if (n <= 1) return 0;
    u32 w = 0; while ((1ull << w) < n) ++w;
    const u64 R = 2147483646ull;
    const u64 y0 = (R >> w) << w, mask = (1ull << w) - 1;
    u64 x = 1;
    for (;;)
    {
        u64 u;
        do { x = (x * 48271ull) % 2147483647ull; u = x - 1; } while (u >= y0);
        u64 v = u & mask;
        if (v < n) return (u32)v;
    }
}

__device__ __forceinline__ u32 mEnd(const RawMatch & m) { return m.begin + m.len; }
__device__ __forceinline__ bool sameBinContigStrand(const RawMatch & a, const RawMatch & b) { return a.bin == b.bin && a.contig == b.contig && a.strand == b.strand; }
// (bin, contig, strand) lexicographic == global contig id, strand
__device__ __forceinline__ int cmpLoc(const RawMatch & a, const RawMatch & b)
{
    if (a.bin != b.bin) return a.bin < b.bin ? -1 : 1;
    if (a.contig != b.contig) return a.contig < b.contig ? -1 : 1;
    if (a.strand != b.strand) return a.strand < b.strand ? -1 : 1;
    return 0;
}
// strict total orders; ia/ib break full ties
__device__ __forceinline__ bool lessEnd(const RawMatch & a, u32 ia, const RawMatch & b, u32 ib)
{
    int c = cmpLoc(a, b); if (c) return c < 0;
    u32 ea = mEnd(a), eb = mEnd(b); if (ea != eb) return ea < eb;
    if (a.errors != b.errors) return a.errors < b.errors;
    if (a.begin != b.begin) return a.begin < b.begin;
    return ia < ib;
}
__device__ __forceinline__ bool lessBegin(const RawMatch & a, u32 ia, const RawMatch & b, u32 ib)
{
    int c = cmpLoc(a, b); if (c) return c < 0;
    if (a.begin != b.begin) return a.begin < b.begin;
    if (a.errors != b.errors) return a.errors < b.errors;
    u32 ea = mEnd(a), eb = mEnd(b); if (ea != eb) return ea < eb;
    return ia < ib;
}
__device__ __forceinline__ bool lessErrors(const RawMatch & a, u32 ia, const RawMatch & b, u32 ib)
{
    if (a.errors != b.errors) return a.errors < b.errors;       // stable sort by errors over coordinate order
    return lessBegin(a, ia, b, ib);
}

enum { AUX_FIRST_END = 1, AUX_U2 = 2, AUX_TAIL1 = 4, AUX_TAIL2 = 8, AUX_PRIMARY = 16 };

__device__ __forceinline__ void waveFence()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ u32 waveMin32(u32 v) { for (int o = 32; o >= 1; o >>= 1) { u32 t = __shfl_xor(v, o, 64); v = t < v ? t : v; } return v; }
__device__ __forceinline__ u32 waveMax32(u32 v) { for (int o = 32; o >= 1; o >>= 1) { u32 t = __shfl_xor(v, o, 64); v = t > v ? t : v; } return v; }
__device__ __forceinline__ u32 waveSum32(u32 v) { for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64); return v; }

// Banded affine alignment of one match (contig horizontal, read vertical, free contig end gaps) + traceback.
// Mirrors globalAlignment(..., Score<int>(0,-1000,-999,-1001), AlignConfig<true,false,false,true>, -e, +e, Gotoh) followed by
// clipSemiGlobal (reference src/mapper_aligner.h:143-150,204-219); tie-breaks pinned by the 7 gapped gold records.
// scratch layout: trace (m+1)*(2e+1) bytes | 6 rows of (2e+1) ints | ops (m + n + 2 bytes)
// returns number of cigar ops written ((count<<4)|op); clipBegin/clipEnd as in the oracle.
__device__ inline u32 alignMatchDevice(const u8 * __restrict__ ctg, int n, const u8 * __restrict__ seq, u32 readLen, u32 strand, int e,
                                       u8 * scratch, u32 * cigarOut, u32 cigarCap, int & clipBegin, int & clipEnd)
{
    const int m = (int)readLen, B = 2 * e + 1;
    const int NEG = -(1 << 28), MISMATCH = -1000, GEXT = -999, GOPEN = -1001;
    u8 * trace = scratch;
    size_t traceBytes = ((size_t)(m + 1) * B + 3) & ~(size_t)3;
    int * rows = (int *)(scratch + traceBytes);
    int * pS = rows, * pH = rows + B, * pV = rows + 2 * B, * cS = rows + 3 * B, * cH = rows + 4 * B, * cV = rows + 5 * B;
    u8 * ops = (u8 *)(rows + 6 * B);
    for (int d = 0; d < B; ++d)
    {
        int j = d - e;
        pS[d] = (j >= 0 && j <= n) ? 0 : NEG; pH[d] = NEG; pV[d] = NEG;
        trace[d] = 0;
    }
    for (int i = 1; i <= m; ++i)
    {
        const u32 rc = readBase(seq, readLen, strand, (u32)(i - 1));
        for (int d = 0; d < B; ++d)
        {
            const int j = i + d - e;
            u8 tr = 0;
            if (j < 0 || j > n) { cS[d] = cH[d] = cV[d] = NEG; trace[(size_t)i * B + d] = 0; continue; }
            int v = NEG, h = NEG;
            if (d + 1 < B)
            {
                int ext = pV[d + 1] + GEXT, opn = pS[d + 1] + GOPEN;
                if (ext >= opn) { v = ext; tr |= 4; } else v = opn;
            }
            if (d >= 1 && j >= 1)
            {
                int ext = cH[d - 1] + GEXT, opn = cS[d - 1] + GOPEN;
                if (ext >= opn) { h = ext; tr |= 8; } else h = opn;
            }
            int best = NEG; u8 tb = 0;
            if (j >= 1)
            {
                const u32 cc = ctg[j - 1];
                best = pS[d] + ((cc == rc && cc < 4) ? 0 : MISMATCH); tb = 1;
            }
            if (v > best) { best = v; tb = 3; }
            if (h > best) { best = h; tb = 2; }
            cS[d] = best; cH[d] = h; cV[d] = v;
            trace[(size_t)i * B + d] = (u8)(tr | tb);
        }
        int * t;
        t = pS; pS = cS; cS = t; t = pH; pH = cH; cH = t; t = pV; pV = cV; cV = t;
    }
    // best cell of the last row, leftmost on ties (pS holds row m)
    int bj = -1, bs = NEG;
    for (int d = 0; d < B; ++d)
    {
        int j = m + d - e;
        if (j < 0 || j > n) continue;
        if (pS[d] > bs) { bs = pS[d]; bj = j; }
    }
    int i = m, j = bj, state = 0, nops = 0;
    while (i > 0)
    {
        const u8 tr = trace[(size_t)i * B + (j - i + e)];
        if (state == 0)
        {
            const u8 tb = tr & 3;
            if (tb == 1) { ops[nops++] = 0; --i; --j; }
            else if (tb == 3) state = 1;
            else if (tb == 2) state = 2;
            else break;
        }
        else if (state == 1) { ops[nops++] = 1; const bool ext = tr & 4; --i; if (!ext) state = 0; }
        else { ops[nops++] = 2; const bool ext = tr & 8; --j; if (!ext) state = 0; }
    }
    clipBegin = j; clipEnd = bj;
    u32 nc = 0;
    for (int k = nops - 1; k >= 0; )
    {
        u8 op = ops[k]; u32 run = 0;
        while (k >= 0 && ops[k] == op) { ++run; --k; }
        if (cigarOut && nc < cigarCap) cigarOut[nc] = (run << 4) | op;
        ++nc;
    }
    return nc;
}

__host__ __device__ inline size_t dpScratchBytes(u32 maxReadLen, u32 maxErrors)
{
    size_t B = 2 * (size_t)maxErrors + 1;
    return (((size_t)(maxReadLen + 1) * B + 3) & ~(size_t)3) + 6 * B * sizeof(int) + 2 * (size_t)maxReadLen + 2 * maxErrors + 16;
}

__global__ __launch_bounds__(256) void k_finalize(FinalizeArgs A)
{
    const u32 lane = threadIdx.x & 63;
    const u32 gwave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const u32 nWaves = (gridDim.x * blockDim.x) >> 6;
    __shared__ u32 sCount[4];
    u32 & mCount = sCount[threadIdx.x >> 6];
    u8 * dp = A.dpScratch + (size_t)gwave * A.dpStride;

    for (u32 read = gwave; read < A.nReads; read += nWaves)
    {
        const u32 s0 = A.readStart[read], n = A.readStart[read + 1] - s0;
        const u32 readLen = (u32)(A.readOffs[read + 1] - A.readOffs[read]);
        const u8 * seq = A.readSeqs + A.readOffs[read];
        ReadOut ro; ro.contig = 0; ro.begin = 0; ro.length = 0; ro.errors = MAX_ERRORS; ro.isRev = 0;
        ro.mapped = 0; ro.bestCount = 0; ro.subCount = 0; ro.mapq = 0; ro.setBegin = 3 * s0; ro.cigarBegin = read * A.cigarStride; ro.cigarLen = 0;
        if (lane == 0) A.out[read] = ro;
        if (n == 0) continue;
        const RawMatch * E = A.gp + s0;
        u32 * aux = A.aux + (size_t)4 * s0;            // [0]=flags [1]=idxA [2]=posB [3]=unused
        RawMatch * M = A.marr + (size_t)3 * s0;
        u32 * maux = A.maux + (size_t)6 * s0;

        // ---- pass A: child END pass — within-bin rank and "first of its (contig,strand,end) group"
        for (u32 i = lane; i < n; i += 64)
        {
            const RawMatch a = E[i];
            u32 idxA = 0; bool first = true;
            for (u32 j = 0; j < n; ++j)
            {
                const RawMatch b = E[j];
                if (b.bin != a.bin || j == i) continue;
                if (lessEnd(b, j, a, i)) { ++idxA; if (sameBinContigStrand(a, b) && mEnd(a) == mEnd(b)) first = false; }
            }
            aux[4 * i] = first ? AUX_FIRST_END : 0; aux[4 * i + 1] = idxA;
        }
        waveFence();
        // ---- pass B: stale tail of pass 1; child BEGIN pass over the survivors
        for (u32 i = lane; i < n; i += 64)
        {
            const RawMatch a = E[i];
            u32 flags = aux[4 * i];
            u32 n1 = 0, posB = 0; bool firstB = true;
            for (u32 j = 0; j < n; ++j)
            {
                const RawMatch b = E[j];
                if (b.bin != a.bin || !(aux[4 * j] & AUX_FIRST_END)) continue;
                ++n1;
                if (j != i && (flags & AUX_FIRST_END) && lessBegin(b, j, a, i)) { ++posB; if (sameBinContigStrand(a, b) && a.begin == b.begin) firstB = false; }
            }
            if (aux[4 * i + 1] >= n1) flags |= AUX_TAIL1;
            if ((flags & AUX_FIRST_END) && firstB) flags |= AUX_U2;
            aux[4 * i + 2] = posB;
            aux[4 * i + 3] = flags;         // staged: pass C must still see the old flags of others
        }
        waveFence();
        for (u32 i = lane; i < n; i += 64) aux[4 * i] = aux[4 * i + 3];
        waveFence();
        // ---- pass C: stale tail of pass 2; child primary = rnd-th co-optimal match in coordinate order
        for (u32 i = lane; i < n; i += 64)
        {
            const RawMatch a = E[i];
            u32 flags = aux[4 * i];
            u32 n2 = 0, minErr = 255;
            for (u32 j = 0; j < n; ++j)
            {
                const RawMatch b = E[j];
                if (b.bin != a.bin || !(aux[4 * j] & AUX_U2)) continue;
                ++n2; if (b.errors < minErr) minErr = b.errors;
            }
            if ((flags & AUX_FIRST_END) && aux[4 * i + 2] >= n2) flags |= AUX_TAIL2;
            if ((flags & AUX_U2) && a.errors == minErr)
            {
                u32 nOpt = 0, ord = 0;
                for (u32 j = 0; j < n; ++j)
                {
                    const RawMatch b = E[j];
                    if (b.bin != a.bin || !(aux[4 * j] & AUX_U2) || b.errors != minErr) continue;
                    ++nOpt; if (j != i && lessBegin(b, j, a, i)) ++ord;
                }
                if (ord == libcxxFirstUniform(nOpt)) flags |= AUX_PRIMARY;
            }
            aux[4 * i + 3] = flags;
        }
        waveFence();
        for (u32 i = lane; i < n; i += 64) aux[4 * i] = aux[4 * i + 3];
        if (lane == 0) mCount = 0;
        waveFence();
        // ---- pass D: copyCigars winner = last processed bin whose child primary has the minimal error count;
        //      alignMatches for gapped child primaries (shrinks the match); build the main mapper's match array
        u32 myMin = 255;
        for (u32 i = lane; i < n; i += 64) if (aux[4 * i] & AUX_PRIMARY) myMin = min(myMin, (u32)E[i].errors);
        const u32 globalMin = waveMin32(myMin);
        u32 myBestRank = 0;
        for (u32 i = lane; i < n; i += 64)
            if ((aux[4 * i] & AUX_PRIMARY) && E[i].errors == globalMin) myBestRank = max(myBestRank, A.binOrderRank[E[i].bin] + 1);
        const u32 winnerRank = waveMax32(myBestRank);     // > 0
        for (u32 base = 0; base < n; base += 64)
        {
            const u32 i = base + lane;
            RawMatch a; u32 flags = 0;
            if (i < n) { a = E[i]; flags = aux[4 * i]; }
            const bool isPrim = flags & AUX_PRIMARY;
            const bool gapped = isPrim && !(a.errors == 0 || (a.errors == 1 && a.len == readLen));
            const bool winner = isPrim && a.errors == globalMin && A.binOrderRank[a.bin] + 1 == winnerRank;
            if (winner && !gapped)
            {
                A.cigars[ro.cigarBegin] = (readLen << 4) | 0;
                A.out[read].cigarLen = 1;                   // rest of the record is written by lane 0 at the end
            }
            u64 todo = __ballot(gapped);
            while (todo)
            {
                const u32 l = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                if (lane == l)
                {
                    const BinDesc & bd = A.bins[a.bin];
                    const u8 * ctg = A.contigs + bd.contigOff + A.limits[bd.limOff + a.contig] + a.begin;
                    int cb, ce;
                    u32 nc = alignMatchDevice(ctg, (int)a.len, seq, readLen, a.strand, (int)a.errors, dp,
                                              winner ? A.cigars + ro.cigarBegin : nullptr, A.cigarStride, cb, ce);
                    if (winner) A.out[read].cigarLen = nc < A.cigarStride ? nc : A.cigarStride;
                    a.begin += (u32)cb; a.len = (u16)ce;   // sic: endPosition(contigGaps), reference src/mapper_aligner.h:213-218
                }
            }
            if (i < n)
            {
                if (flags & AUX_U2) { u32 p = atomicAdd(&mCount, 1u); M[p] = a; }
                const RawMatch orig = E[i];
                if (flags & AUX_TAIL2) { u32 p = atomicAdd(&mCount, 1u); M[p] = orig; }
                if (flags & AUX_TAIL1) { u32 p = atomicAdd(&mCount, 1u); M[p] = orig; }
            }
        }
        waveFence();
        const u32 nM = mCount;
        // ---- pass E: main END pass
        for (u32 i = lane; i < nM; i += 64)
        {
            const RawMatch a = M[i];
            bool first = true;
            for (u32 j = 0; j < nM && first; ++j)
            {
                if (j == i) continue;
                const RawMatch b = M[j];
                if (sameBinContigStrand(a, b) && mEnd(a) == mEnd(b) && lessEnd(b, j, a, i)) first = false;
            }
            maux[2 * i] = first ? 1u : 0u;
        }
        waveFence();
        // ---- pass F: main BEGIN pass over the survivors
        for (u32 i = lane; i < nM; i += 64)
        {
            const RawMatch a = M[i];
            bool keep = maux[2 * i] & 1;
            for (u32 j = 0; j < nM && keep; ++j)
            {
                if (j == i || !(maux[2 * j] & 1)) continue;
                const RawMatch b = M[j];
                if (sameBinContigStrand(a, b) && a.begin == b.begin && lessBegin(b, j, a, i)) keep = false;
            }
            maux[2 * i + 1] = keep ? 1u : 0u;
        }
        waveFence();
        // ---- pass G: rankMatches2 — sort by errors (stable over coordinate order), strata clip, primary pick
        u32 myMinE = 255;
        for (u32 i = lane; i < nM; i += 64) if (maux[2 * i + 1]) myMinE = min(myMinE, (u32)M[i].errors);
        const u32 minE = waveMin32(myMinE);
        const u32 strata = readStrataOf(readLen, A.strataRate);
        u32 myOpt = 0, mySet = 0;
        for (u32 i = lane; i < nM; i += 64)
            if (maux[2 * i + 1]) { myOpt += M[i].errors == minE; mySet += M[i].errors <= minE + strata; }
        const u32 nOpt = waveSum32(myOpt), nSet = waveSum32(mySet);
        const u32 pick = libcxxFirstUniform(nOpt);
        for (u32 i = lane; i < nM; i += 64)
        {
            if (!maux[2 * i + 1]) continue;
            const RawMatch a = M[i];
            if (a.errors > minE + strata) continue;
            u32 rank = 0;
            for (u32 j = 0; j < nM; ++j)
            {
                if (j == i || !maux[2 * j + 1]) continue;
                if (lessErrors(M[j], j, a, i)) ++rank;
            }
            OutMatch om; om.contig = A.bins[a.bin].contigIdBase + a.contig; om.begin = a.begin; om.length = a.len; om.errors = a.errors; om.isRev = a.strand;
            A.outSet[(size_t)ro.setBegin + rank] = om;
            if (rank == pick)
            {
                ReadOut * o = &A.out[read];
                o->contig = om.contig; o->begin = om.begin; o->length = om.length; o->errors = om.errors; o->isRev = om.isRev;
                o->mapped = a.len != 0;
            }
        }
        if (lane == 0)
        {
            ReadOut * o = &A.out[read];
            o->bestCount = nOpt; o->subCount = nSet - nOpt;
        }
        waveFence();
    }
}

// readStart-based regrouping of the raw pool
__global__ void k_group_matches(const RawMatch * __restrict__ pool, u32 nPool, const u32 * __restrict__ readStart, u32 * __restrict__ cursor,
                                RawMatch * __restrict__ gp)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nPool) return;
    const RawMatch m = pool[i];
    const u32 dst = readStart[m.read] + atomicAdd(&cursor[m.read], 1u);
    gp[dst] = m;
}

} // namespace dy

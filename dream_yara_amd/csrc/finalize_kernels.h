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
    u32 paired, reserved;   // paired-end: ReadsContext::paired of the main mapper
    double pairProb;        // paired-end: primaryMatchesProbs[read]
};

__device__ __forceinline__ void initReadOut(ReadOut & ro, u32 setBegin, u32 cigarBegin)
{
    ro.contig = 0; ro.begin = 0; ro.length = 0; ro.errors = MAX_ERRORS; ro.isRev = 0;
    ro.mapped = 0; ro.bestCount = 0; ro.subCount = 0; ro.mapq = 0; ro.setBegin = setBegin; ro.cigarBegin = cigarBegin; ro.cigarLen = 0;
    ro.paired = 0; ro.reserved = 0; ro.pairProb = 0.0;
}

struct OutMatch { u32 contig; u32 begin; u16 length; u8 errors; u8 isRev; };

struct FinalizeArgs
{
    const RawMatch * gp;          // matches grouped by read
    const u32 * readStart;        // nReads + 1
    const u64 * readOffs;         // read offsets (for lengths)
    const u8 *  readSeqs;
    const u8 *  readSeqsRc;       // reverse complements, same offsets
    u32 nReads;
    const BinDesc * bins;
    const u8 * contigs; const u64 * limits;
    const u32 * binOrderRank;
    float strataRate, errorRate;
    // scratch
    u32 * aux;                    // 4 u32 per raw match
    RawMatch * marr;              // 3 per raw match: the main mapper's matchesByCoord of this read
    u32 * maux;                   // 2 u32 per marr entry
    u8 *  dpScratch; size_t dpStride;   // per wave
    // outputs
    ReadOut * out;
    OutMatch * outSet;            // 3 per raw match (only the first nSet of a read's region are meaningful)
    u32 * cigars; u32 cigarStride;
    // reads that need the full aggregation (more than one raw match, or a gapped single match); the others were written by k_finalize_simple
    const u32 * complexList; const u32 * complexCount;
    u64 * sortKeys;         // 6 words per raw match (per read: [6 * readStart, ...)), or nullptr
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

__device__ __forceinline__ void waveFence()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// the same for data exchanged through LDS only: the wavefront's LDS operations have completed (its global stores may still be in flight —
// a full fence per DP row made the row loop wait ~2 us for the trace bytes it had just stored)
__device__ __forceinline__ void waveFenceLds()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ u32 waveMin32(u32 v) { for (int o = 32; o >= 1; o >>= 1) { u32 t = __shfl_xor(v, o, 64); v = t < v ? t : v; } return v; }
__device__ __forceinline__ u32 waveSum32(u32 v) { for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64); return v; }

// Sorts idx[0..cnt) (indices into some element array) ascending under the strict total order `less(ia, ib)`.
// Bitonic network with all compare-exchanges ascending (partner i ^ (k-1) on the first step of each merge), which is
// valid for any cnt: positions >= cnt behave as +infinity and never move.  One wavefront, idx in global memory.
template <typename TLess>
__device__ void waveSort(u32 * idx, u32 cnt, u32 lane, TLess less)
{
    if (cnt <= 1) return;
    if (cnt <= 64)
    {   // in registers: lane p holds position p
        const u32 INF = 0xFFFFFFFFu;
        u32 mine = lane < cnt ? idx[lane] : INF;
        for (u32 k = 2; (k >> 1) < cnt; k <<= 1)
        {
            for (u32 j = k >> 1, first = 1; j > 0; j >>= 1, first = 0)
            {
                const u32 partner = first ? (lane ^ (k - 1)) : (lane ^ j);
                const u32 other = __shfl(mine, partner, 64);
                const bool iAmLow = lane < partner;
                // ascending: low position keeps the smaller
                bool otherLess = other != INF && (mine == INF || less(other, mine));
                bool mineLess = mine != INF && (other == INF || less(mine, other));
                if (iAmLow ? otherLess : mineLess) mine = other;
            }
        }
        if (lane < cnt) idx[lane] = mine;
        waveFence();
        return;
    }
    for (u32 k = 2; (k >> 1) < cnt; k <<= 1)
    {
        for (u32 j = k >> 1, first = 1; j > 0; j >>= 1, first = 0)
        {
            for (u32 i = lane; i < cnt; i += 64)
            {
                const u32 l = first ? (i ^ (k - 1)) : (i ^ j);
                if (l > i && l < cnt)
                {
                    const u32 a = idx[i], b = idx[l];
                    if (less(b, a)) { idx[i] = b; idx[l] = a; }
                }
            }
            waveFence();
        }
    }
}

// wave-wide stream compaction helper: returns the slot of this lane's item (valid only if pred) and advances base
__device__ __forceinline__ u32 wavePush(bool pred, u32 lane, u32 & base)
{
    const u64 b = __ballot(pred);
    const u32 slot = base + (u32)__popcll(b & ((1ull << lane) - 1));
    base += (u32)__popcll(b);
    return slot;
}

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

// Is the banded DP's answer for a 2-error match of full read length certainly "<readLen>M, no shrink"?
// With free contig end gaps (TTop/TBottom), mismatch -1000, gap open -1001 / extend -999, diagonal preferred on ties and the
// LEFTMOST best cell of the last row, the main diagonal (score -2000) loses only to an alignment with exactly one inserted
// base and no mismatch (-1001; it skips the first or the last contig base), or — on the leftmost-tie rule — to one 2-base
// insertion without mismatches that leaves at least one trailing contig base unaligned (-2000, ends left of column n).
// Those exist only if a prefix of the read matches the window and the rest matches it shifted by one or two bases, which
// five LCPs rule out (conservatively).  `seq` is the read in the match's orientation.
__device__ inline bool alignIsUngapped2(const u8 * __restrict__ ctg, const u8 * __restrict__ seq, int m)
{
    if (countMismatches(ctg, seq, m) != 2) return false;
    const int lcpF = lcpForward(ctg, seq, m), lcpB = lcpBackward(ctg + m, seq + m, m);
    const int LB1 = lcpBackward(ctg + m - 1, seq + m, m - 1), LB2 = lcpBackward(ctg + m - 2, seq + m, m - 2);     // read[x+g] == contig[x]
    const int LF1 = lcpForward(ctg + 1, seq, m - 1), LF2 = lcpForward(ctg + 2, seq, m - 2);                        // read[x] == contig[x+g]
    if (lcpF + LB1 >= m - 1 || lcpF + LB2 >= m - 2) return false;
    if (LF1 + lcpB >= m - 1 || LF2 + lcpB >= m - 2) return false;
    if (LF1 + LB1 >= m - 2) return false;
    return true;
}

// Single-thread form for short reads and narrow bands (readLen <= 160, e <= 7): the three band rows live in registers
// (loops fully unrolled over the 15 possible diagonals), the trace is one 64-bit word per row (4 bits per cell) in
// private memory.  Same values, tie-breaks and traceback as alignMatchDevice; used by the one-match fast path.
enum { THREAD_DP_MAX_LEN = 160, THREAD_DP_MAX_E = 7 };
// EMAX bounds the error budget of every match the kernel will see (3: a 7-diagonal band, 28 trace bits per row; 7: 15
// diagonals, 60 bits).  The trace lives in LDS, one TRow per (row, thread) at trace[row * traceStride]: the traceback is a
// chain of ~readLen dependent loads, which in private (global) memory was most of the latency of these list kernels.
template <int EMAX, typename TRow>
__device__ inline u32 alignMatchThread(const u8 * __restrict__ ctg, int n, const u8 * __restrict__ seq, u32 readLen, u32 strand, int e,
                                       u32 * cigarOut, u32 cigarCap, int & clipBegin, int & clipEnd, TRow * __restrict__ trace, u32 traceStride)
{
    enum { BM = 2 * EMAX + 1 };
    static_assert(4 * BM <= 8 * sizeof(TRow), "trace row too narrow");
    const int m = (int)readLen, B = 2 * e + 1;
    const int NEG = -(1 << 28), MISMATCH = -1000, GEXT = -999, GOPEN = -1001;
    int pS[BM], pV[BM];
    u32 cw[BM];          // contig base under cell (i, d): ctg[i + d - e - 1], 5 outside the window (matches nothing)
#pragma unroll
    for (int d = 0; d < BM; ++d)
    {
        const int j = d - e;
        pS[d] = (d < B && j >= 0 && j <= n) ? 0 : NEG; pV[d] = NEG;
        const int idx = 1 + d - e - 1;
        cw[d] = (idx >= 0 && idx < n) ? ctg[idx] : 5u;
    }
    trace[0] = 0;
    // The read base and the incoming contig base of a row are loaded TWO rows ahead, as raw bytes: their latency hides behind the rows'
    // ~250 instructions each.  (One row ahead is not enough: the load counter is in-order, and the wait the compiler places at the loop
    // head for the loads issued before the loop also covers the load the row has just issued; complementing the byte right behind the
    // load would make the row wait for it on the spot.)
    // Both loads are unconditional (clamped index, validity applied at the use): with a conditional load in the loop the compiler's wait
    // for the bytes loaded before the loop would also cover the byte the row has just asked for.
    const u8 * rp = strand ? seq + (m - 1) : seq;
    const int rstep = strand ? -1 : 1;
    auto rawAt = [&](int row) -> u32 { return rp[(row <= m ? row - 1 : m - 1) * rstep]; };                       // read base of row `row` (1-based)
    auto incomingAt = [&](int row) -> u32 { const int idx = row + BM - 1 - e; return ctg[idx < 0 ? 0 : (idx < n ? idx : n - 1)]; };
    auto incomingValid = [&](int row) -> bool { const int idx = row + BM - 1 - e; return idx >= 0 && idx < n; };
    u32 raw1 = 0, raw2 = 0, inc1 = 5, inc2 = 5;
    if (m > 0 && n > 0) { raw1 = rawAt(1); inc1 = incomingAt(1); raw2 = rawAt(2); inc2 = incomingAt(2); }
    for (int i = 1; i <= m; ++i)
    {
        const u32 rc = (strand && raw1 < 4) ? 3 - raw1 : raw1, incoming = incomingValid(i) ? inc1 : 5u;
        raw1 = raw2; inc1 = inc2;
        if (n > 0) { raw2 = rawAt(i + 2); inc2 = incomingAt(i + 2); }
        int cS[BM], cV[BM];
        int leftS = NEG, leftH = NEG;
        TRow row = 0;
#pragma unroll
        for (int d = 0; d < BM; ++d)
        {
            const int j = i + d - e;
            // straight-line (selects instead of a branch region per cell: a lone wavefront pays for every exec-mask switch); NEG stays
            // far from overflow however many gap steps are added to it
            const bool valid = d < B && j >= 0 && j <= n;
            const bool useV = valid && d + 1 < B, useH = valid && d >= 1 && j >= 1, useD = valid && j >= 1;
            const int vExt = pV[d + 1 < BM ? d + 1 : d] + GEXT, vOpn = pS[d + 1 < BM ? d + 1 : d] + GOPEN;
            const bool vE = vExt >= vOpn;
            const int v = useV ? (vE ? vExt : vOpn) : NEG;
            const int hExt = leftH + GEXT, hOpn = leftS + GOPEN;
            const bool hE = hExt >= hOpn;
            const int h = useH ? (hE ? hExt : hOpn) : NEG;
            const u32 cc = cw[d];
            int s = useD ? pS[d] + ((cc == rc && cc < 4) ? 0 : MISMATCH) : NEG;
            u32 tb = useD ? 1u : 0u;
            if (v > s) { s = v; tb = 3; }
            if (h > s) { s = h; tb = 2; }
            if (!valid) { s = NEG; tb = 0; }
            const u32 tr = tb | ((useV && vE) ? 4u : 0u) | ((useH && hE) ? 8u : 0u);
            cS[d] = s; cV[d] = v; leftS = s; leftH = h;
            row |= (TRow)tr << (4 * d);
        }
        trace[(size_t)i * traceStride] = row;
#pragma unroll
        for (int d = 0; d < BM; ++d) { pS[d] = cS[d]; pV[d] = cV[d]; cw[d] = d + 1 < BM ? cw[d + 1 < BM ? d + 1 : d] : incoming; }
    }
    int bj = -1, bs = NEG;
#pragma unroll
    for (int d = 0; d < BM; ++d)
    {
        const int j = m + d - e;
        if (d < B && j >= 0 && j <= n && pS[d] > bs) { bs = pS[d]; bj = j; }
    }
    // traceback; cigar runs are produced back to front, so count first and fill from the end
    int i = m, j = bj, state = 0;
    u32 runs = 0; int lastOp = -1;
    {
        int ti = i, tj = j, ts = 0;
        while (ti > 0)
        {
            const u32 tr = (u32)(trace[(size_t)ti * traceStride] >> (4 * (tj - ti + e))) & 15;
            int op;
            if (ts == 0)
            {
                const u32 tb = tr & 3;
                if (tb == 1) { op = 0; --ti; --tj; }
                else if (tb == 3) { ts = 1; continue; }
                else if (tb == 2) { ts = 2; continue; }
                else break;
            }
            else if (ts == 1) { op = 1; --ti; if (!(tr & 4)) ts = 0; }
            else { op = 2; --tj; if (!(tr & 8)) ts = 0; }
            if (op != lastOp) { ++runs; lastOp = op; }
        }
        clipBegin = tj;
    }
    clipEnd = bj;
    if (cigarOut)
    {
        u32 slot = runs; lastOp = -1; u32 run = 0;
        while (i > 0)
        {
            const u32 tr = (u32)(trace[(size_t)i * traceStride] >> (4 * (j - i + e))) & 15;
            int op;
            if (state == 0)
            {
                const u32 tb = tr & 3;
                if (tb == 1) { op = 0; --i; --j; }
                else if (tb == 3) { state = 1; continue; }
                else if (tb == 2) { state = 2; continue; }
                else break;
            }
            else if (state == 1) { op = 1; --i; if (!(tr & 4)) state = 0; }
            else { op = 2; --j; if (!(tr & 8)) state = 0; }
            if (op != lastOp)
            {
                if (lastOp >= 0) { --slot; if (slot < cigarCap) cigarOut[slot] = (run << 4) | (u32)lastOp; }
                lastOp = op; run = 0;
            }
            ++run;
        }
        if (lastOp >= 0) { --slot; if (slot < cigarCap) cigarOut[slot] = (run << 4) | (u32)lastOp; }
    }
    return runs;
}

// Wave-parallel form of alignMatchDevice for bands of at most 64 diagonals: lane d owns diagonal d of the band, a row
// costs one coalesced contig load, two lane shifts and one max-plus prefix scan (the horizontal gap matrix obeys
// H[d] = max_{d'<d} (T[d'] + open + ext*(d-d'-1)) with T = max(diagonal, vertical), because open < ext).
// Values, tie-breaks and the traceback are identical to the sequential form.  All lanes must call it with the same arguments.
__device__ inline u32 alignMatchWave(const u8 * __restrict__ ctg, int n, const u8 * __restrict__ seq, u32 readLen, u32 strand, int e,
                                     u8 * scratch, u32 * cigarOut, u32 cigarCap, int & clipBegin, int & clipEnd, u32 lane)
{
    const int m = (int)readLen, B = 2 * e + 1, d = (int)lane;
    const int NEG = -(1 << 28), MISMATCH = -1000, GEXT = -999, GOPEN = -1001;
    u8 * trace = scratch;
    u8 * ops = scratch + (size_t)(m + 1) * B;
    u8 * sRead = ops + m + n + 2;            // strand-resolved read and the contig infix, staged once (coalesced)
    u8 * sCtg = sRead + m;
    for (int x = (int)lane; x < m; x += 64) sRead[x] = (u8)readBase(seq, readLen, strand, (u32)x);
    for (int x = (int)lane; x < n; x += 64) sCtg[x] = ctg[x];
    const bool inBand = d < B;
    int pS, pV = NEG;
    { const int j = d - e; pS = (inBand && j >= 0 && j <= n) ? 0 : NEG; if (inBand) trace[d] = 0; }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int i = 1; i <= m; ++i)
    {
        const u32 rc = sRead[i - 1];
        const int j = i + d - e;
        const bool valid = inBand && j >= 0 && j <= n;
        // vertical: from (i-1, j) = lane d+1 of the previous row
        const int upS = __shfl_down(pS, 1, 64), upV = __shfl_down(pV, 1, 64);
        int v = NEG; u8 tr = 0;
        if (valid && d + 1 < B)
        {
            const int ext = upV + GEXT, opn = upS + GOPEN;
            if (ext >= opn) { v = ext; tr |= 4; } else v = opn;
        }
        int t = NEG; u8 tb = 0;
        if (valid && j >= 1)
        {
            const u32 cc = sCtg[j - 1];
            t = pS + ((cc == rc && cc < 4) ? 0 : MISMATCH); tb = 1;
        }
        if (v > t) { t = v; tb = 3; }
        // horizontal: exclusive prefix max of U = T + 999*d over the lower diagonals of this row
        const int u = valid ? t - GEXT * d : NEG;
        int scan = __shfl_up(u, 1, 64); if (lane == 0) scan = NEG;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int other = __shfl_up(scan, o, 64); if ((int)lane >= o && other > scan) scan = other; }
        int h = NEG;
        if (valid && d >= 1 && j >= 1) h = scan + GOPEN + GEXT * (d - 1);
        int sCell = t;
        if (h > sCell) { sCell = h; tb = 2; }
        if (!valid) { sCell = NEG; h = NEG; v = NEG; tb = 0; tr = 0; }
        // extend-vs-open flag of the horizontal move needs the finished left neighbour
        const int leftS = __shfl_up(sCell, 1, 64), leftH = __shfl_up(h, 1, 64);
        if (valid && d >= 1 && j >= 1 && leftH + GEXT >= leftS + GOPEN) tr |= 8;
        if (inBand) trace[(size_t)i * B + d] = (u8)(tr | tb);
        pS = sCell; pV = v;
    }
    // best cell of the last row, leftmost on ties
    const bool lastValid = inBand && m + d - e >= 0 && m + d - e <= n;
    const int myBest = lastValid ? pS : NEG;
    int mx = myBest;
    for (int o = 32; o >= 1; o >>= 1) { const int other = __shfl_xor(mx, o, 64); mx = other > mx ? other : mx; }
    const u64 atMax = __ballot(lastValid && myBest == mx);
    const int bd = __ffsll((long long)atMax) - 1;
    const int bj = m + bd - e;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    int i = m, j = bj, state = 0, nops = 0;
    while (i > 0)
    {
        const u8 tr = trace[(size_t)i * B + (j - i + e)];
        if (state == 0)
        {
            const u8 tb = tr & 3;
            if (tb == 1) { if (lane == 0) ops[nops] = 0; ++nops; --i; --j; }
            else if (tb == 3) state = 1;
            else if (tb == 2) state = 2;
            else break;
        }
        else if (state == 1) { if (lane == 0) ops[nops] = 1; ++nops; const bool ext = tr & 4; --i; if (!ext) state = 0; }
        else { if (lane == 0) ops[nops] = 2; ++nops; const bool ext = tr & 8; --j; if (!ext) state = 0; }
    }
    clipBegin = j; clipEnd = bj;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    u32 nc = 0;
    for (int k = nops - 1; k >= 0; )
    {
        const u8 op = ops[k]; u32 run = 0;
        while (k >= 0 && ops[k] == op) { ++run; --k; }
        if (cigarOut && nc < cigarCap && lane == 0) cigarOut[nc] = (run << 4) | op;
        ++nc;
    }
    return nc;
}

__host__ __device__ inline size_t dpScratchBytes(u32 maxReadLen, u32 maxErrors)
{
    size_t B = 2 * (size_t)maxErrors + 1;
    return (((size_t)(maxReadLen + 1) * B + 3) & ~(size_t)3) + 6 * B * sizeof(int) + 4 * (size_t)maxReadLen + 4 * maxErrors + 32;
}

// Sub-wavefront form of alignMatchWave for the WIDE bands of short reads (more than 3 errors: 9..15 diagonals): the 16 lanes of a DPP row
// work on one match, lane d on diagonal d of the band (B = 2e+1 <= 15), four matches per wavefront.  The one-thread form
// (alignMatchThread<7>) needs 150 rows x 15 cells x ~30 dependent instructions per match — 1.2 ms for ONE wavefront however short the list
// — while these bands are rare (a percent of the reads), so the 5 x higher instruction count per match of the group form does not matter
// and its latency (a row = one cell per lane + a 4-step max-plus scan) does: 0.1 ms.  Strand-resolved read, contig infix and the trace
// (one byte per cell) live in the group's LDS region.  Values, tie-breaks and traceback are those of alignMatchDevice.  The lanes of a
// group must call it together with the same arguments; groups are independent (no wave-wide operation).
__host__ __device__ inline u32 groupDpLdsBytes(u32 maxReadLen) { return ((maxReadLen + 1) * 15 + 2 * maxReadLen + 16 + 3) & ~3u; }
// DPP row shifts (a row = 16 consecutive lanes): lane i takes the value of lane i + 1 / i - N of its row; a lane whose source lies outside
// the row keeps its own value
__device__ __forceinline__ int dppRowShl1(int x) { return __builtin_amdgcn_update_dpp(x, x, 0x101, 0xF, 0xF, false); }
// whole-wavefront forms (gfx9 DPP): lane i takes lane i - 1, lane 0 takes `first`; inclusive running maximum over the 64 lanes (shifts inside
// the 16-lane rows, then the last lane of a row broadcast into the rows behind it) — a dozen cycles each where a ds_bpermute shuffle takes ~100
__device__ __forceinline__ int dppWaveShr1(int x, int first) { return __builtin_amdgcn_update_dpp(first, x, 0x138, 0xF, 0xF, false); }
__device__ __forceinline__ int dppWaveMaxScan(int x)
{
    int t;
    t = __builtin_amdgcn_update_dpp(x, x, 0x111, 0xF, 0xF, false); x = t > x ? t : x;      // row_shr:1
    t = __builtin_amdgcn_update_dpp(x, x, 0x112, 0xF, 0xF, false); x = t > x ? t : x;      // row_shr:2
    t = __builtin_amdgcn_update_dpp(x, x, 0x114, 0xF, 0xF, false); x = t > x ? t : x;      // row_shr:4
    t = __builtin_amdgcn_update_dpp(x, x, 0x118, 0xF, 0xF, false); x = t > x ? t : x;      // row_shr:8
    t = __builtin_amdgcn_update_dpp(x, x, 0x142, 0xA, 0xF, false); x = t > x ? t : x;      // row_bcast:15 into rows 1 and 3
    t = __builtin_amdgcn_update_dpp(x, x, 0x143, 0xC, 0xF, false); x = t > x ? t : x;      // row_bcast:31 into rows 2 and 3
    return x;
}
template <int N> __device__ __forceinline__ int dppRowShr(int x) { return __builtin_amdgcn_update_dpp(x, x, 0x110 + N, 0xF, 0xF, false); }

__device__ inline u32 alignMatchGroup(const u8 * __restrict__ ctg, int n, const u8 * __restrict__ seq, u32 readLen, u32 strand, int e,
                                      u8 * lds, u32 * cigarOut, u32 cigarCap, int & clipBegin, int & clipEnd, u32 dLane)
{
    const int m = (int)readLen, B = 2 * e + 1, d = (int)dLane;
    const int NEG = -(1 << 28), MISMATCH = -1000, GEXT = -999, GOPEN = -1001;
    u8 * trace = lds;                                   // (m + 1) * B
    u8 * sRead = trace + (size_t)(m + 1) * B;           // m
    u8 * sCtg = sRead + m;                              // n <= m + e
    for (int x = d; x < m; x += 16) sRead[x] = (u8)readBase(seq, readLen, strand, (u32)x);
    for (int x = d; x < n; x += 16) sCtg[x] = ctg[x];
    const bool inBand = d < B;
    int pS, pV = NEG;
    { const int j = d - e; pS = (inBand && j >= 0 && j <= n) ? 0 : NEG; if (inBand) trace[d] = 0; }
    waveFence();
    // the read base and the contig base of the NEXT row are fetched from LDS while this row computes
    u32 rcNext = m > 0 ? sRead[0] : 0u;
    u32 ccNext = 5u;
    { const int j1 = 1 + d - e; if (inBand && j1 >= 1 && j1 <= n) ccNext = sCtg[j1 - 1]; }
    for (int i = 1; i <= m; ++i)
    {
        const u32 rc = rcNext, cc = ccNext;
        const int j = i + d - e;
        const bool valid = inBand && j >= 0 && j <= n;
        if (i < m) rcNext = sRead[i];
        ccNext = (inBand && j >= 0 && j < n) ? sCtg[j] : 5u;              // row i+1 looks at contig base (j+1)-1
        // vertical: from (i-1, j) = lane d+1 of the previous row
        const int upS = dppRowShl1(pS), upV = dppRowShl1(pV);
        int v = NEG; u32 tr = 0;
        if (valid && d + 1 < B)
        {
            const int ext = upV + GEXT, opn = upS + GOPEN;
            if (ext >= opn) { v = ext; tr |= 4; } else v = opn;
        }
        int t = NEG; u32 tb = 0;
        if (valid && j >= 1) { t = pS + ((cc == rc && cc < 4) ? 0 : MISMATCH); tb = 1; }
        if (v > t) { t = v; tb = 3; }
        // horizontal: exclusive prefix max of U = T + 999*d over the lower diagonals of this row
        const int u = valid ? t - GEXT * d : NEG;
        int scan = dppRowShr<1>(u); if (d == 0) scan = NEG;
        { const int other = dppRowShr<1>(scan); if (d >= 1 && other > scan) scan = other; }
        { const int other = dppRowShr<2>(scan); if (d >= 2 && other > scan) scan = other; }
        { const int other = dppRowShr<4>(scan); if (d >= 4 && other > scan) scan = other; }
        { const int other = dppRowShr<8>(scan); if (d >= 8 && other > scan) scan = other; }
        int h = NEG;
        if (valid && d >= 1 && j >= 1) h = scan + GOPEN + GEXT * (d - 1);
        int sCell = t;
        if (h > sCell) { sCell = h; tb = 2; }
        if (!valid) { sCell = NEG; h = NEG; v = NEG; tb = 0; tr = 0; }
        // extend-vs-open flag of the horizontal move needs the finished left neighbour
        const int leftS = dppRowShr<1>(sCell), leftH = dppRowShr<1>(h);
        if (valid && d >= 1 && j >= 1 && leftH + GEXT >= leftS + GOPEN) tr |= 8;
        if (inBand) trace[(size_t)i * B + d] = (u8)(tr | tb);
        pS = sCell; pV = v;
    }
    // best cell of the last row, leftmost on ties
    const bool lastValid = inBand && m + d - e >= 0 && m + d - e <= n;
    const int myBest = lastValid ? pS : NEG;
    int mx = myBest;
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) { const int other = __shfl_xor(mx, o, 16); mx = other > mx ? other : mx; }
    int bd = (lastValid && myBest == mx) ? d : 16;
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) { const int other = __shfl_xor(bd, o, 16); bd = other < bd ? other : bd; }
    const int bj = m + bd - e;
    waveFence();
    // traceback (every lane of the group walks it: the trace reads are LDS broadcasts); cigar runs come out back to front, so they are
    // counted first and filled from the end.  A diagonal outside the band ends the walk (it cannot happen with a trace written above;
    // the check keeps a damaged one from steering the LDS address anywhere).
    u32 runs = 0;
    {
        int ti = m, tj = bj, ts = 0, lastOp = -1;
        while (ti > 0 && (u32)(tj - ti + e) < (u32)B)
        {
            const u32 tr = trace[(size_t)ti * B + (tj - ti + e)];
            int op;
            if (ts == 0)
            {
                const u32 tb = tr & 3;
                if (tb == 1) { op = 0; --ti; --tj; }
                else if (tb == 3) { ts = 1; continue; }
                else if (tb == 2) { ts = 2; continue; }
                else break;
            }
            else if (ts == 1) { op = 1; --ti; if (!(tr & 4)) ts = 0; }
            else { op = 2; --tj; if (!(tr & 8)) ts = 0; }
            if (op != lastOp) { ++runs; lastOp = op; }
        }
        clipBegin = tj;
    }
    clipEnd = bj;
    if (cigarOut && d == 0)
    {
        int i = m, j = bj, state = 0, lastOp = -1;
        u32 slot = runs, run = 0;
        while (i > 0 && (u32)(j - i + e) < (u32)B)
        {
            const u32 tr = trace[(size_t)i * B + (j - i + e)];
            int op;
            if (state == 0)
            {
                const u32 tb = tr & 3;
                if (tb == 1) { op = 0; --i; --j; }
                else if (tb == 3) { state = 1; continue; }
                else if (tb == 2) { state = 2; continue; }
                else break;
            }
            else if (state == 1) { op = 1; --i; if (!(tr & 4)) state = 0; }
            else { op = 2; --j; if (!(tr & 8)) state = 0; }
            if (op != lastOp)
            {
                if (lastOp >= 0) { --slot; if (slot < cigarCap) cigarOut[slot] = (run << 4) | (u32)lastOp; }
                lastOp = op; run = 0;
            }
            ++run;
        }
        if (lastOp >= 0) { --slot; if (slot < cigarCap) cigarOut[slot] = (run << 4) | (u32)lastOp; }
    }
    waveFence();            // the region is reused by the group's next match
    return runs;
}

// ---- who works on one read in k_finalize_t: a wavefront (almost every read) or a whole 1024-thread block (reads with thousands of
// matches: repeats, low-complexity runs — one wavefront would sort them alone for seconds).  Same code, different primitives.
struct WaveGroup
{
    enum : u32 { SIZE = 64, BLOCK = 256, PER_BLOCK = 4, SORT_CAP = 1 };
    __device__ static __forceinline__ u32 tid() { return threadIdx.x & 63; }
    __device__ static __forceinline__ u32 slot() { return threadIdx.x >> 6; }
    __device__ static __forceinline__ u32 groupId() { return (blockIdx.x * blockDim.x + threadIdx.x) >> 6; }
    __device__ static __forceinline__ u32 groupCount() { return (gridDim.x * blockDim.x) >> 6; }
    __device__ static __forceinline__ bool alignerWave() { return true; }
    __device__ static __forceinline__ void fence() { waveFence(); }
    __device__ static __forceinline__ u32 sum(u32 v) { return waveSum32(v); }
    __device__ static __forceinline__ u32 minv(u32 v) { return waveMin32(v); }
    __device__ static __forceinline__ u32 push(bool pred, u32 & base) { return wavePush(pred, threadIdx.x & 63, base); }
    __device__ static __forceinline__ u32 count(bool pred) { return (u32)__popcll(__ballot(pred)); }
    // rank of this thread among the threads with pred (valid if pred); total = their number
    __device__ static __forceinline__ u32 rank(bool pred, u32 & total)
    {
        const u64 b = __ballot(pred);
        total = (u32)__popcll(b);
        return (u32)__popcll(b & ((1ull << (threadIdx.x & 63)) - 1));
    }
    __device__ static __forceinline__ int firstTrue(bool pred) { const u64 b = __ballot(pred); return b ? __ffsll((long long)b) - 1 : -1; }
    __device__ static __forceinline__ u32 bcast(u32 v, u32 src) { return __shfl(v, src, 64); }
};

struct BlockGroup
{   // every thread of the block must make the same sequence of calls
    enum : u32 { SIZE = 1024, BLOCK = 1024, PER_BLOCK = 1, NWAVES = 16, SORT_CAP = 4096 };
    __device__ static __forceinline__ u32 tid() { return threadIdx.x; }
    __device__ static __forceinline__ u32 slot() { return 0; }
    __device__ static __forceinline__ u32 groupId() { return blockIdx.x; }
    __device__ static __forceinline__ u32 groupCount() { return gridDim.x; }
    __device__ static __forceinline__ bool alignerWave() { return threadIdx.x < 64; }
    __device__ static __forceinline__ void fence() { __syncthreads(); }
    __device__ static __forceinline__ u32 sum(u32 v)
    {
        __shared__ u32 part[NWAVES];
        v = waveSum32(v);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
        __syncthreads();
        u32 t = 0;
#pragma unroll
        for (u32 k = 0; k < NWAVES; ++k) t += part[k];
        __syncthreads();
        return t;
    }
    __device__ static __forceinline__ u32 minv(u32 v)
    {
        __shared__ u32 part[NWAVES];
        v = waveMin32(v);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
        __syncthreads();
        u32 t = 0xFFFFFFFFu;
#pragma unroll
        for (u32 k = 0; k < NWAVES; ++k) t = min(t, part[k]);
        __syncthreads();
        return t;
    }
    __device__ static __forceinline__ u32 rank(bool pred, u32 & total)
    {
        __shared__ u32 part[NWAVES];
        const u64 b = __ballot(pred);
        const u32 w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) part[w] = (u32)__popcll(b);
        __syncthreads();
        u32 before = 0, t = 0;
#pragma unroll
        for (u32 k = 0; k < NWAVES; ++k) { const u32 c = part[k]; before += k < w ? c : 0; t += c; }
        __syncthreads();
        total = t;
        return before + (u32)__popcll(b & ((1ull << (threadIdx.x & 63)) - 1));
    }
    __device__ static __forceinline__ u32 push(bool pred, u32 & base) { u32 total; const u32 r = rank(pred, total); const u32 slot = base + r; base += total; return slot; }
    __device__ static __forceinline__ u32 count(bool pred) { u32 total; (void)rank(pred, total); return total; }
    __device__ static __forceinline__ int firstTrue(bool pred) { const u32 m = minv(pred ? threadIdx.x : 0xFFFFFFFFu); return m == 0xFFFFFFFFu ? -1 : (int)m; }
    __device__ static __forceinline__ u32 bcast(u32 v, u32 src)
    {
        __shared__ u32 x;
        if (threadIdx.x == src) x = v;
        __syncthreads();
        const u32 r = x;
        __syncthreads();
        return r;
    }
};

// waveSort's global-memory path for any group
template <typename G, typename TLess>
__device__ void groupSort(u32 * idx, u32 cnt, TLess less)
{
    if (G::SIZE == 64) { waveSort(idx, cnt, threadIdx.x & 63, less); return; }
    if (cnt <= 1) return;
    for (u32 k = 2; (k >> 1) < cnt; k <<= 1)
    {
        for (u32 j = k >> 1, first = 1; j > 0; j >>= 1, first = 0)
        {
            for (u32 i = G::tid(); i < cnt; i += G::SIZE)
            {
                const u32 l = first ? (i ^ (k - 1)) : (i ^ j);
                if (l > i && l < cnt)
                {
                    const u32 a = idx[i], b = idx[l];
                    if (less(b, a)) { idx[i] = b; idx[l] = a; }
                }
            }
            G::fence();
        }
    }
}

// The three sort orders of the aggregation as 128-bit keys (hi, lo) with the element index in the low 23 bits of lo:
// key(a, ia) < key(b, ib)  <=>  lessEnd / lessBegin / lessErrors (a, ia, b, ib).  MODE 0 = end, 1 = begin, 2 = errors.
template <int MODE>
__device__ __forceinline__ void sortKeyOf(const RawMatch & a, u32 i, u64 & hi, u64 & lo)
{
    const u64 loc = ((u64)a.bin << 17) | ((u64)a.contig << 1) | (u64)(a.strand & 1);      // 33 bits: (bin, contig, strand)
    const u32 end = a.begin + a.len;
    if (MODE == 0)      { hi = (loc << 31) | (end >> 1);     lo = ((u64)(end & 1) << 63)     | ((u64)a.errors << 55) | ((u64)a.begin << 23) | i; }
    else if (MODE == 1) { hi = (loc << 31) | (a.begin >> 1); lo = ((u64)(a.begin & 1) << 63) | ((u64)a.errors << 55) | ((u64)end << 23)     | i; }
    else                { hi = ((u64)a.errors << 56) | (loc << 23) | (a.begin >> 9); lo = ((u64)(a.begin & 511) << 55) | ((u64)end << 23) | i; }
}

// Sorts idx[0..cnt) (indices < nTotal into R) by the order MODE.  A block sorts keys instead of indices: up to SORT_CAP elements
// entirely in LDS (sKeys holds 2 * SORT_CAP words); more than that with the keys in the global area gKeys (2 * cnt words): the steps
// of the network whose partners lie inside a SORT_CAP-aligned tile run tile by tile in LDS, only the wider ones go through global
// memory (coalesced 16-byte elements instead of index -> record gathers).  The network is waveSort's (valid for any cnt: ascending
// compare-exchanges, mirror partner on the first step of a merge).  Everything else takes groupSort with the comparator of the same
// order; all give the one permutation the strict total order defines.
template <typename G, int MODE>
__device__ void groupSortKeyed(u32 * idx, u32 cnt, const RawMatch * R, u32 nTotal, u64 * sKeys, u64 * gKeys)
{
    if constexpr (G::SIZE != 64)
    {
        const u32 CAP = G::SORT_CAP;
        u64 * sHi = sKeys, * sLo = sKeys + CAP;
        auto exchangeLds = [&](u32 li, u32 ll) {
            const u64 ah = sHi[li], al = sLo[li], bh = sHi[ll], bl = sLo[ll];
            if (bh < ah || (bh == ah && bl < al)) { sHi[li] = bh; sLo[li] = bl; sHi[ll] = ah; sLo[ll] = al; }
        };
        if (cnt > 1 && cnt <= CAP && nTotal <= (1u << 23))
        {
            for (u32 i = G::tid(); i < cnt; i += G::SIZE) { const u32 e = idx[i]; u64 hi, lo; sortKeyOf<MODE>(R[e], e, hi, lo); sHi[i] = hi; sLo[i] = lo; }
            __syncthreads();
            for (u32 k = 2; (k >> 1) < cnt; k <<= 1)
                for (u32 j = k >> 1, first = 1; j > 0; j >>= 1, first = 0)
                {
                    for (u32 i = G::tid(); i < cnt; i += G::SIZE)
                    {
                        const u32 l = first ? (i ^ (k - 1)) : (i ^ j);
                        if (l > i && l < cnt) exchangeLds(i, l);
                    }
                    __syncthreads();
                }
            for (u32 i = G::tid(); i < cnt; i += G::SIZE) idx[i] = (u32)(sLo[i] & 0x7FFFFFu);
            __syncthreads();
            return;
        }
        if (cnt > CAP && gKeys && nTotal <= (1u << 23))
        {
            u64 * gHi = gKeys, * gLo = gKeys + cnt;
            for (u32 i = G::tid(); i < cnt; i += G::SIZE) { const u32 e = idx[i]; u64 hi, lo; sortKeyOf<MODE>(R[e], e, hi, lo); gHi[i] = hi; gLo[i] = lo; }
            __syncthreads();
            // steps kFrom.. of the network restricted to partners inside a tile: kFrom = 2 sorts every tile (merges up to CAP);
            // kFrom = 0 runs only the xor steps j = CAP/2 .. 1 that finish a merge wider than a tile
            auto tiles = [&](bool wholeSort) {
                for (u32 base = 0; base < cnt; base += CAP)
                {
                    const u32 len = cnt - base < CAP ? cnt - base : CAP;
                    for (u32 li = G::tid(); li < len; li += G::SIZE) { sHi[li] = gHi[base + li]; sLo[li] = gLo[base + li]; }
                    __syncthreads();
                    if (wholeSort)
                    {
                        for (u32 k = 2; k <= CAP; k <<= 1)
                            for (u32 j = k >> 1, first = 1; j > 0; j >>= 1, first = 0)
                            {
                                for (u32 li = G::tid(); li < len; li += G::SIZE)
                                {
                                    const u32 ll = first ? (li ^ (k - 1)) : (li ^ j);
                                    if (ll > li && ll < len) exchangeLds(li, ll);
                                }
                                __syncthreads();
                            }
                    }
                    else
                    {
                        for (u32 j = CAP >> 1; j > 0; j >>= 1)
                        {
                            for (u32 li = G::tid(); li < len; li += G::SIZE)
                            {
                                const u32 ll = li ^ j;
                                if (ll > li && ll < len) exchangeLds(li, ll);
                            }
                            __syncthreads();
                        }
                    }
                    for (u32 li = G::tid(); li < len; li += G::SIZE) { gHi[base + li] = sHi[li]; gLo[base + li] = sLo[li]; }
                    __syncthreads();
                }
            };
            tiles(true);
            for (u32 k = 2 * CAP; (k >> 1) < cnt; k <<= 1)
            {
                for (u32 j = k >> 1, first = 1; j >= CAP; j >>= 1, first = 0)
                {
                    for (u32 i = G::tid(); i < cnt; i += G::SIZE)
                    {
                        const u32 l = first ? (i ^ (k - 1)) : (i ^ j);
                        if (l > i && l < cnt)
                        {
                            const u64 ah = gHi[i], al = gLo[i], bh = gHi[l], bl = gLo[l];
                            if (bh < ah || (bh == ah && bl < al)) { gHi[i] = bh; gLo[i] = bl; gHi[l] = ah; gLo[l] = al; }
                        }
                    }
                    __syncthreads();
                }
                tiles(false);
            }
            for (u32 i = G::tid(); i < cnt; i += G::SIZE) idx[i] = (u32)(gLo[i] & 0x7FFFFFu);
            __syncthreads();
            return;
        }
    }
    if (MODE == 0)      groupSort<G>(idx, cnt, [&](u32 a, u32 b) { return lessEnd(R[a], a, R[b], b); });
    else if (MODE == 1) groupSort<G>(idx, cnt, [&](u32 a, u32 b) { return lessBegin(R[a], a, R[b], b); });
    else                groupSort<G>(idx, cnt, [&](u32 a, u32 b) { return lessErrors(R[a], a, R[b], b); });
}

// Reads of the complex list with minN <= #matches <= maxN, one group G per read.
template <typename G>
__global__ __launch_bounds__(G::BLOCK) void k_finalize_t(FinalizeArgs A, u32 minN, u32 maxN, u32 dpSlabBase)
{
    const u32 lane = G::tid();
    const u32 gwave = G::groupId();
    const u32 nWaves = G::groupCount();
    // alignment scratch: LDS when the banded matrices of this batch fit (they do for short reads), else the global slab
    enum { DP_LDS_BYTES = 4096, SMALL_N = 32 };
    __shared__ __attribute__((aligned(16))) u8 sDp[G::PER_BLOCK][DP_LDS_BYTES];
    u8 * dp = A.dpStride <= DP_LDS_BYTES ? sDp[G::slot()] : A.dpScratch + (size_t)(dpSlabBase + gwave) * A.dpStride;
    // reads with few matches (almost all of them) keep every working array in LDS; the rest use the global scratch
    __shared__ RawMatch sE[G::PER_BLOCK][SMALL_N];
    __shared__ RawMatch sM[G::PER_BLOCK][3 * SMALL_N];
    __shared__ u32 sIdx[G::PER_BLOCK][8 * SMALL_N];
    __shared__ u64 sKeys[2 * G::SORT_CAP];        // key sort of a block (groupSortKeyed); a token word for the wavefront variant

    const u32 nList = *A.complexCount;
    for (u32 li = gwave; li < nList; li += nWaves)
    {
        const u32 read = A.complexList[li];
        const u32 s0 = A.readStart[read], n = A.readStart[read + 1] - s0;
        if (n < minN || n > maxN) continue;
        const u32 readLen = (u32)(A.readOffs[read + 1] - A.readOffs[read]);
        const u8 * seq = A.readSeqs + A.readOffs[read];
        ReadOut ro; initReadOut(ro, 3 * s0, read * A.cigarStride);
        if (n == 0) { if (lane == 0) A.out[read] = ro; continue; }
        const bool small = G::SIZE == 64 && n <= SMALL_N;
        const u32 wv = G::slot();
        if (small && lane < n) sE[wv][lane] = A.gp[s0 + lane];
        const RawMatch * E = small ? sE[wv] : A.gp + s0;
        u32 * idxA = small ? sIdx[wv] : A.aux + (size_t)4 * s0;                 // n entries: all matches of the read in child END order
        u32 * idxB = idxA + n;                                                  // <= n entries: pass-1 survivors of the current bin
        RawMatch * M = small ? sM[wv] : A.marr + (size_t)3 * s0;                // the main mapper's matchesByCoord of this read
        u32 * idxM = small ? sIdx[wv] + 2 * SMALL_N : A.maux + (size_t)6 * s0;  // 3n
        u32 * idxF = idxM + (size_t)3 * n;                                      // 3n
        u64 * gKeys = A.sortKeys ? A.sortKeys + (size_t)6 * s0 : nullptr;       // 2 x 3n words: keys of the sort in progress (block variant, > SORT_CAP elements)

        // ---- child mappers: one (read, bin) segment at a time, bins in ascending order
        for (u32 i = lane; i < n; i += G::SIZE) idxA[i] = i;
        G::fence();
        groupSortKeyed<G, 0>(idxA, n, E, n, sKeys, gKeys);
        u32 nM = 0;
        u32 winErr = 256, winRank = 0;                 // copyCigars: last processed bin among those with the minimal errors
        u32 p0 = 0;
        while (p0 < n)
        {
            const u32 bin = E[idxA[p0]].bin;
            u32 cntBin = 0;
            for (u32 p = p0 + lane; p < n; p += G::SIZE) cntBin += E[idxA[p]].bin == bin;
            const u32 p1 = p0 + G::sum(cntBin);
            // pass 1 (ContigEnd): unique by (contig, strand, end); stale tail = sorted positions >= n1
            u32 n1 = 0;
            for (u32 base = p0; base < p1; base += G::SIZE)
            {
                const u32 p = base + lane;
                bool first = false;
                if (p < p1)
                {
                    const RawMatch a = E[idxA[p]];
                    first = p == p0;
                    if (!first) { const RawMatch b = E[idxA[p - 1]]; first = !(sameBinContigStrand(a, b) && mEnd(a) == mEnd(b)); }
                }
                const u32 slot = G::push(first, n1);
                if (first) idxB[slot] = idxA[p];
            }
            for (u32 base = p0 + n1; base < p1; base += G::SIZE)
            {
                const u32 p = base + lane;
                const u32 slot = G::push(p < p1, nM);
                if (p < p1) M[slot] = E[idxA[p]];
            }
            G::fence();
            // pass 2 (ContigBegin) over the n1 survivors
            groupSortKeyed<G, 1>(idxB, n1, E, n, sKeys, gKeys);
            u32 n2 = 0, myMin = 255;
            for (u32 base = 0; base < n1; base += G::SIZE)
            {
                const u32 q = base + lane;
                bool first = false;
                if (q < n1)
                {
                    const RawMatch a = E[idxB[q]];
                    first = q == 0;
                    if (!first) { const RawMatch b = E[idxB[q - 1]]; first = !(sameBinContigStrand(a, b) && a.begin == b.begin); }
                    if (first) myMin = min(myMin, (u32)a.errors);
                }
                n2 += G::count(first);
            }
            const u32 minErr = G::minv(myMin);
            for (u32 base = n2; base < n1; base += G::SIZE)
            {   // stale tail of pass 2
                const u32 q = base + lane;
                const u32 slot = G::push(q < n1, nM);
                if (q < n1) M[slot] = E[idxB[q]];
            }
            // rankMatches (child): primary = rnd-th co-optimal match in coordinate order among the unique ones
            u32 nOptB = 0;
            for (u32 base = 0; base < n1; base += G::SIZE)
            {
                const u32 q = base + lane;
                bool opt = false;
                if (q < n1)
                {
                    const RawMatch a = E[idxB[q]];
                    bool first = q == 0;
                    if (!first) { const RawMatch b = E[idxB[q - 1]]; first = !(sameBinContigStrand(a, b) && a.begin == b.begin); }
                    opt = first && a.errors == minErr;
                }
                nOptB += G::count(opt);
            }
            const u32 pickB = libcxxFirstUniform(nOptB);
            u32 optSeen = 0;
            const u32 binRank = A.binOrderRank[bin] + 1;
            for (u32 base = 0; base < n1; base += G::SIZE)
            {
                const u32 q = base + lane;
                RawMatch a; bool first = false;
                if (q < n1)
                {
                    a = E[idxB[q]];
                    first = q == 0;
                    if (!first) { const RawMatch b = E[idxB[q - 1]]; first = !(sameBinContigStrand(a, b) && a.begin == b.begin); }
                }
                const bool opt = first && a.errors == minErr;
                u32 optHere;
                const u32 myOptRank = optSeen + G::rank(opt, optHere);
                optSeen += optHere;
                const bool isPrim = opt && myOptRank == pickB;
                const int plFound = G::firstTrue(isPrim);
                if (plFound >= 0)
                {
                    const u32 pl = (u32)plFound;                        // the thread holding this bin's primary
                    const bool winner = minErr < winErr || (minErr == winErr && binRank > winRank);   // uniform
                    // broadcast the primary so that the whole wave can align it
                    const u32 pBegin = G::bcast(q < n1 ? a.begin : 0u, pl), pLen = G::bcast(q < n1 ? (u32)a.len : 0u, pl), pErr = G::bcast(q < n1 ? (u32)a.errors : 0u, pl);
                    const u32 pStrand = G::bcast(q < n1 ? (u32)a.strand : 0u, pl), pContig = G::bcast(q < n1 ? (u32)a.contig : 0u, pl);
                    const bool gapped = !(pErr == 0 || (pErr == 1 && pLen == readLen));
                    if (gapped)
                    {   // alignMatches: CIGAR + shrink
                        const BinDesc & bd = A.bins[bin];
                        const u8 * ctg = A.contigs + bd.contigOff + A.limits[bd.limOff + pContig] + pBegin;
                        int cb = 0, ce = (int)pLen;
                        u32 nc = 0;
                        if (2 * pErr + 1 <= 64)
                        {   // one wavefront aligns (lane = diagonal); a block hands the result round
                            if (G::alignerWave())
                                nc = alignMatchWave(ctg, (int)pLen, seq, readLen, pStrand, (int)pErr, dp, winner ? A.cigars + ro.cigarBegin : nullptr, A.cigarStride, cb, ce, threadIdx.x & 63);
                            if (G::SIZE != 64) { nc = G::bcast(nc, 0); cb = (int)G::bcast((u32)cb, 0); ce = (int)G::bcast((u32)ce, 0); }
                        }
                        else
                        {
                            if (lane == 0)
                                nc = alignMatchDevice(ctg, (int)pLen, seq, readLen, pStrand, (int)pErr, dp, winner ? A.cigars + ro.cigarBegin : nullptr, A.cigarStride, cb, ce);
                            nc = G::bcast(nc, 0); cb = (int)G::bcast((u32)cb, 0); ce = (int)G::bcast((u32)ce, 0);
                        }
                        if (winner) ro.cigarLen = nc < A.cigarStride ? nc : A.cigarStride;
                        if (lane == pl) { a.begin += (u32)cb; a.len = (u16)ce; }      // sic: endPosition(contigGaps), reference src/mapper_aligner.h:213-218
                    }
                    else if (winner)
                    {
                        if (lane == 0) A.cigars[ro.cigarBegin] = (readLen << 4) | 0;
                        ro.cigarLen = 1;
                    }
                    if (winner) { winErr = minErr; winRank = binRank; }
                }
                const u32 slot = G::push(first, nM);
                if (first) M[slot] = a;
            }
            G::fence();
            p0 = p1;
        }
        // ---- main mapper: aggregateMatches + rankMatches2 over the union
        for (u32 i = lane; i < nM; i += G::SIZE) idxM[i] = i;
        G::fence();
        groupSortKeyed<G, 0>(idxM, nM, M, nM, sKeys, gKeys);
        u32 m1 = 0;
        for (u32 base = 0; base < nM; base += G::SIZE)
        {
            const u32 p = base + lane;
            bool first = false;
            if (p < nM)
            {
                const RawMatch a = M[idxM[p]];
                first = p == 0;
                if (!first) { const RawMatch b = M[idxM[p - 1]]; first = !(sameBinContigStrand(a, b) && mEnd(a) == mEnd(b)); }
            }
            const u32 slot = G::push(first, m1);
            if (first) idxF[slot] = idxM[p];
        }
        G::fence();
        groupSortKeyed<G, 1>(idxF, m1, M, nM, sKeys, gKeys);
        u32 m2 = 0;
        for (u32 base = 0; base < m1; base += G::SIZE)
        {
            const u32 q = base + lane;
            bool first = false;
            if (q < m1)
            {
                const RawMatch a = M[idxF[q]];
                first = q == 0;
                if (!first) { const RawMatch b = M[idxF[q - 1]]; first = !(sameBinContigStrand(a, b) && a.begin == b.begin); }
            }
            const u32 slot = G::push(first, m2);
            if (first) idxM[slot] = idxF[q];
        }
        G::fence();
        groupSortKeyed<G, 2>(idxM, m2, M, nM, sKeys, gKeys);
        const u32 minE = M[idxM[0]].errors;
        const u32 strata = readStrataOf(readLen, A.strataRate);
        u32 cOpt = 0, cSet = 0;
        for (u32 q = lane; q < m2; q += G::SIZE) { const u32 e = M[idxM[q]].errors; cOpt += e == minE; cSet += e <= minE + strata; }
        const u32 nOpt = G::sum(cOpt), nSet = G::sum(cSet);
        const u32 pick = libcxxFirstUniform(nOpt);
        for (u32 q = lane; q < nSet; q += G::SIZE)
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
            A.out[read] = ro;
        }
        G::fence();
    }
}

// Fast path, one thread per read: a read without matches is unmapped; a read whose only match is ungapped needs no
// sorting, no alignment and has X0 = 1, X1 = 0.  Everything else is queued for k_finalize_t.
__device__ __forceinline__ void writeSingleMatch(const FinalizeArgs & A, u32 read, ReadOut & ro, const RawMatch & a)
{
    OutMatch om; om.contig = A.bins[a.bin].contigIdBase + a.contig; om.begin = a.begin; om.length = a.len; om.errors = a.errors; om.isRev = a.strand;
    A.outSet[(size_t)ro.setBegin] = om;
    ro.contig = om.contig; ro.begin = om.begin; ro.length = om.length; ro.errors = om.errors; ro.isRev = om.isRev;
    ro.mapped = om.length != 0; ro.bestCount = 1; ro.subCount = 0;
    A.out[read] = ro;
}

// list[count++] = value for the threads that want it, NL lists at once, ONE global atomic per block and list: same-address
// atomics are served one at a time (~10 ns each on this chip), so with one per wavefront a million-thread routing kernel
// spends 0.3 ms queueing for its list counters.  Every thread of the block must call this (it synchronises).
template <int NL>
__device__ __forceinline__ void blockAppend(const bool (&want)[NL], u32 value, u32 * const (&lists)[NL], u32 * const (&counts)[NL])
{
    __shared__ u32 waveOff[NL][16];
    __shared__ u32 blockBase[NL];
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nWaves = (blockDim.x + 63) >> 6;
    u64 mask[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l)
    {
        mask[l] = __ballot(want[l]);
        if (lane == 0) waveOff[l][wave] = (u32)__popcll(mask[l]);
    }
    __syncthreads();
    if (threadIdx.x < NL)
    {
        const u32 l = threadIdx.x;
        u32 total = 0;
        for (u32 w = 0; w < nWaves; ++w) { const u32 t = waveOff[l][w]; waveOff[l][w] = total; total += t; }
        blockBase[l] = total ? atomicAdd(counts[l], total) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int l = 0; l < NL; ++l)
        if (want[l]) lists[l][blockBase[l] + waveOff[l][wave] + (u32)__popcll(mask[l] & ((1ull << lane) - 1))] = value;
    __syncthreads();
}

// splitWide (the batch may hold matches with more than 3 errors): small-list reads with such a match go to smallWideList (k_finalize_small_group)
__global__ __launch_bounds__(1024) void k_finalize_simple(FinalizeArgs A, u32 * __restrict__ complexList, u32 * __restrict__ complexCount,
                                  u32 * __restrict__ alignList, u32 * __restrict__ alignCount, u32 * __restrict__ smallList, u32 * __restrict__ smallCount,
                                  u32 * __restrict__ smallWideList, u32 * __restrict__ smallWideCount, u32 splitWide)
{
    const u32 read = blockIdx.x * blockDim.x + threadIdx.x;
    u32 route = 0;      // 0 = done here, 1 = align list, 2 = small list, 3 = complex list, 4 = small list, wide band
    if (read < A.nReads)
    {
        const u32 s0 = A.readStart[read], n = A.readStart[read + 1] - s0;
        const u32 readLen = (u32)(A.readOffs[read + 1] - A.readOffs[read]);
        ReadOut ro; initReadOut(ro, 3 * s0, read * A.cigarStride);
        if (n == 0) A.out[read] = ro;
        else
        {
            route = 3;
            if (n == 1)
            {
                const RawMatch a = A.gp[s0];
                const bool gapped = !(a.errors == 0 || (a.errors == 1 && a.len == readLen));
                if (!gapped)
                {
                    A.cigars[ro.cigarBegin] = (readLen << 4) | 0;
                    ro.cigarLen = 1;
                    writeSingleMatch(A, read, ro, a);
                    route = 0;
                }
                else if (readLen <= THREAD_DP_MAX_LEN && a.errors <= THREAD_DP_MAX_E)
                    route = 1;      // alignment needed, but nothing to sort: k_align_check / k_align_single (all lanes busy there)
            }
            else if (n <= 4 && readLen <= THREAD_DP_MAX_LEN && readErrorsOf(readLen, A.errorRate) <= THREAD_DP_MAX_E)
            {
                route = 2;
                if (splitWide)
                    for (u32 i = 0; i < n; ++i) if (A.gp[s0 + i].errors > 3) route = 4;
            }
        }
    }
    const bool want[4] = {route == 1, route == 2, route == 3, route == 4};
    u32 * const lists[4] = {alignList, smallList, complexList, smallWideList};
    u32 * const counts[4] = {alignCount, smallCount, complexCount, smallWideCount};
    blockAppend<4>(want, read, lists, counts);
}

// First pass over the align list: a 2-substitution match needs no DP when alignIsUngapped2 holds; the rest goes to dpList.
__global__ __launch_bounds__(256) void k_align_check(FinalizeArgs A, const u32 * __restrict__ alignList, const u32 * __restrict__ alignCount,
                                                     u32 * __restrict__ dpList, u32 * __restrict__ dpCount, u32 * __restrict__ dpWideList, u32 * __restrict__ dpWideCount)
{
    const u32 nList = *alignCount;
    for (u32 base = blockIdx.x * blockDim.x; base < nList; base += gridDim.x * blockDim.x)
    {
        const u32 li = base + threadIdx.x;
        bool needDp = false, wide = false; u32 read = 0;
        if (li < nList)
        {
            read = alignList[li];
            const u32 s0 = A.readStart[read];
            const u32 readLen = (u32)(A.readOffs[read + 1] - A.readOffs[read]);
            const RawMatch a = A.gp[s0];
            needDp = true; wide = a.errors > 3;
            if (a.errors == 2 && a.len == readLen && readLen >= 8)
            {
                const BinDesc & bd = A.bins[a.bin];
                const u8 * ctg = A.contigs + bd.contigOff + A.limits[bd.limOff + a.contig] + a.begin;
                const u8 * seq = (a.strand ? A.readSeqsRc : A.readSeqs) + A.readOffs[read];
                if (alignIsUngapped2(ctg, seq, (int)readLen))
                {
                    ReadOut ro; initReadOut(ro, 3 * s0, read * A.cigarStride);
                    A.cigars[ro.cigarBegin] = (readLen << 4) | 0; ro.cigarLen = 1;
                    writeSingleMatch(A, read, ro, a);
                    needDp = false;
                }
            }
        }
        // up to 3 errors (7 diagonals): k_align_single<3>; more: k_align_group
        const bool want[2] = {needDp && !wide, needDp && wide};
        u32 * const lists[2] = {dpList, dpWideList};
        u32 * const counts[2] = {dpCount, dpWideCount};
        blockAppend<2>(want, read, lists, counts);
    }
}

// alignMatches for reads whose only match needs the DP: CIGAR + shrink (reference src/mapper_aligner.h:204-236)
template <int EMAX, typename TRow>
__global__ __launch_bounds__(64) void k_align_single(FinalizeArgs A, const u32 * __restrict__ alignList, const u32 * __restrict__ alignCount)
{
    extern __shared__ u64 dpTraceLds[];
    TRow * trace = (TRow *)dpTraceLds + threadIdx.x;
    const u32 nList = *alignCount;
    for (u32 li = blockIdx.x * blockDim.x + threadIdx.x; li < nList; li += gridDim.x * blockDim.x)
    {
        const u32 read = alignList[li];
        const u32 s0 = A.readStart[read];
        const u32 readLen = (u32)(A.readOffs[read + 1] - A.readOffs[read]);
        ReadOut ro; initReadOut(ro, 3 * s0, read * A.cigarStride);
        RawMatch a = A.gp[s0];
        const BinDesc & bd = A.bins[a.bin];
        const u8 * ctg = A.contigs + bd.contigOff + A.limits[bd.limOff + a.contig] + a.begin;
        int cb, ce;
        const u32 nc = alignMatchThread<EMAX, TRow>(ctg, (int)a.len, A.readSeqs + A.readOffs[read], readLen, a.strand, (int)a.errors,
                                                    A.cigars + ro.cigarBegin, A.cigarStride, cb, ce, trace, blockDim.x);
        ro.cigarLen = nc < A.cigarStride ? nc : A.cigarStride;
        a.begin += (u32)cb; a.len = (u16)ce;
        writeSingleMatch(A, read, ro, a);
    }
}

// Reads with 2..SMALL_T matches, one thread each: the same child/main aggregation as k_finalize_t with insertion sorts over
// a handful of entries in private memory (the wavefront version spends ~10^4 instructions per read on these).
enum { SMALL_T = 4 };
template <typename TLess>
__device__ __forceinline__ void insertionSortIdx(u32 * idx, u32 cnt, TLess less)
{
    for (u32 i = 1; i < cnt; ++i)
    {
        const u32 v = idx[i];
        u32 j = i;
        while (j > 0 && less(v, idx[j - 1])) { idx[j] = idx[j - 1]; --j; }
        idx[j] = v;
    }
}

// The child / main aggregation of one read with at most SMALL_T raw matches, in one thread.  `ro` comes initialised (set / cigar
// offsets, paired-end fields) and is written at the end.  primaryOf(bin, nOpt) names the child's primary: {true, i} = the i-th
// co-optimal entry (single-end: the child's own random pick), {false, i} = the i-th unique entry in coordinate order (paired-end:
// the pairing stage has chosen).
struct SmallPrimary { bool amongOptimal; u32 index; };
// alignFn(ctg, n, seq, readLen, strand, e, cigarOut, cigarCap, cb, ce) is the alignment DP; `writer` = this thread stores the results
// (the 16 lanes of a group run the aggregation redundantly around alignMatchGroup; one of them writes).
template <typename TAlign, typename TPrimaryOf>
__device__ inline void finalizeSmallRead(const FinalizeArgs & A, u32 read, ReadOut & ro, TAlign alignFn, bool writer, TPrimaryOf primaryOf)
{
    const u32 s0 = A.readStart[read], n = A.readStart[read + 1] - s0;
    const u32 readLen = (u32)(A.readOffs[read + 1] - A.readOffs[read]);
    const u8 * seq = A.readSeqs + A.readOffs[read];
    RawMatch E[SMALL_T], M[3 * SMALL_T];
    u32 idxA[SMALL_T], idxB[SMALL_T], idxM[3 * SMALL_T], idxF[3 * SMALL_T];
    for (u32 i = 0; i < n; ++i) { E[i] = A.gp[s0 + i]; idxA[i] = i; }
    insertionSortIdx(idxA, n, [&](u32 a, u32 b) { return lessEnd(E[a], a, E[b], b); });
    u32 nM = 0, winErr = 256, winRank = 0;
    u32 p0 = 0;
    while (p0 < n)
    {
        const u32 bin = E[idxA[p0]].bin;
        u32 p1 = p0 + 1;
        while (p1 < n && E[idxA[p1]].bin == bin) ++p1;
        u32 n1 = 0;
        for (u32 p = p0; p < p1; ++p)
        {
            const bool first = p == p0 || !(sameBinContigStrand(E[idxA[p]], E[idxA[p - 1]]) && mEnd(E[idxA[p]]) == mEnd(E[idxA[p - 1]]));
            if (first) idxB[n1++] = idxA[p];
        }
        for (u32 p = p0 + n1; p < p1; ++p) M[nM++] = E[idxA[p]];          // stale tail of pass 1
        insertionSortIdx(idxB, n1, [&](u32 a, u32 b) { return lessBegin(E[a], a, E[b], b); });
        u32 n2 = 0, minErr = 255, nOptB = 0;
        u32 firstMask = 0;
        for (u32 q = 0; q < n1; ++q)
        {
            const bool first = q == 0 || !(sameBinContigStrand(E[idxB[q]], E[idxB[q - 1]]) && E[idxB[q]].begin == E[idxB[q - 1]].begin);
            if (first) { firstMask |= 1u << q; ++n2; if (E[idxB[q]].errors < minErr) minErr = E[idxB[q]].errors; }
        }
        for (u32 q = n2; q < n1; ++q) M[nM++] = E[idxB[q]];               // stale tail of pass 2
        for (u32 q = 0; q < n1; ++q) if ((firstMask >> q & 1) && E[idxB[q]].errors == minErr) ++nOptB;
        const SmallPrimary prim = primaryOf(bin, nOptB);
        const u32 binRank = A.binOrderRank[bin] + 1;
        u32 optSeen = 0, uniqueSeen = 0;
        for (u32 q = 0; q < n1; ++q)
        {
            if (!(firstMask >> q & 1)) continue;
            RawMatch a = E[idxB[q]];
            bool isPrimary;
            if (prim.amongOptimal) isPrimary = a.errors == minErr && optSeen++ == prim.index;
            else isPrimary = uniqueSeen == prim.index;
            ++uniqueSeen;
            if (isPrimary)
            {   // the child's primary: CIGAR provenance (copyCigars) + alignMatches
                const u32 primErr = a.errors;
                const bool winner = primErr < winErr || (primErr == winErr && binRank > winRank);
                const bool gapped = !(a.errors == 0 || (a.errors == 1 && a.len == readLen));
                if (gapped)
                {
                    const BinDesc & bd = A.bins[a.bin];
                    const u8 * ctg = A.contigs + bd.contigOff + A.limits[bd.limOff + a.contig] + a.begin;
                    int cb, ce;
                    const u32 nc = alignFn(ctg, (int)a.len, seq, readLen, (u32)a.strand, (int)a.errors, winner ? A.cigars + ro.cigarBegin : nullptr, A.cigarStride, cb, ce);
                    if (winner) ro.cigarLen = nc < A.cigarStride ? nc : A.cigarStride;
                    a.begin += (u32)cb; a.len = (u16)ce;
                }
                else if (winner) { if (writer) A.cigars[ro.cigarBegin] = (readLen << 4) | 0; ro.cigarLen = 1; }
                if (winner) { winErr = primErr; winRank = binRank; }
            }
            M[nM++] = a;
        }
        p0 = p1;
    }
    // main mapper
    for (u32 i = 0; i < nM; ++i) idxM[i] = i;
    insertionSortIdx(idxM, nM, [&](u32 a, u32 b) { return lessEnd(M[a], a, M[b], b); });
    u32 m1 = 0;
    for (u32 p = 0; p < nM; ++p)
        if (p == 0 || !(sameBinContigStrand(M[idxM[p]], M[idxM[p - 1]]) && mEnd(M[idxM[p]]) == mEnd(M[idxM[p - 1]]))) idxF[m1++] = idxM[p];
    insertionSortIdx(idxF, m1, [&](u32 a, u32 b) { return lessBegin(M[a], a, M[b], b); });
    u32 m2 = 0;
    for (u32 q = 0; q < m1; ++q)
        if (q == 0 || !(sameBinContigStrand(M[idxF[q]], M[idxF[q - 1]]) && M[idxF[q]].begin == M[idxF[q - 1]].begin)) idxM[m2++] = idxF[q];
    insertionSortIdx(idxM, m2, [&](u32 a, u32 b) { return lessErrors(M[a], a, M[b], b); });
    const u32 minE = M[idxM[0]].errors, strata = readStrataOf(readLen, A.strataRate);
    u32 nOpt = 0, nSet = 0;
    for (u32 q = 0; q < m2; ++q) { const u32 e = M[idxM[q]].errors; nOpt += e == minE; nSet += e <= minE + strata; }
    const u32 pick = libcxxFirstUniform(nOpt);
    for (u32 q = 0; q < nSet; ++q)
    {
        const RawMatch a = M[idxM[q]];
        OutMatch om; om.contig = A.bins[a.bin].contigIdBase + a.contig; om.begin = a.begin; om.length = a.len; om.errors = a.errors; om.isRev = a.strand;
        if (writer) A.outSet[(size_t)ro.setBegin + q] = om;
        if (q == pick)
        {
            ro.contig = om.contig; ro.begin = om.begin; ro.length = om.length; ro.errors = om.errors; ro.isRev = om.isRev;
            ro.mapped = a.len != 0;
        }
    }
    ro.bestCount = nOpt; ro.subCount = nSet - nOpt;
    if (writer) A.out[read] = ro;
}

// the one-thread DP as finalizeSmallRead's alignFn
template <int EMAX, typename TRow>
struct ThreadAligner
{
    TRow * trace; u32 traceStride;
    __device__ __forceinline__ u32 operator()(const u8 * ctg, int n, const u8 * seq, u32 readLen, u32 strand, int e, u32 * cigarOut, u32 cigarCap, int & cb, int & ce) const
    { return alignMatchThread<EMAX, TRow>(ctg, n, seq, readLen, strand, e, cigarOut, cigarCap, cb, ce, trace, traceStride); }
};
// 16 lanes per match
struct GroupAligner
{
    u8 * lds; u32 d;
    __device__ __forceinline__ u32 operator()(const u8 * ctg, int n, const u8 * seq, u32 readLen, u32 strand, int e, u32 * cigarOut, u32 cigarCap, int & cb, int & ce) const
    { return alignMatchGroup(ctg, n, seq, readLen, strand, e, lds, cigarOut, cigarCap, cb, ce, d); }
};

template <int EMAX, typename TRow>
__global__ __launch_bounds__(64) void k_finalize_small(FinalizeArgs A, const u32 * __restrict__ smallList, const u32 * __restrict__ smallCount)
{
    extern __shared__ u64 dpTraceLds[];
    TRow * trace = (TRow *)dpTraceLds + threadIdx.x;
    const u32 nList = *smallCount;
    for (u32 li = blockIdx.x * blockDim.x + threadIdx.x; li < nList; li += gridDim.x * blockDim.x)
    {
        const u32 read = smallList[li];
        ReadOut ro; initReadOut(ro, 3 * A.readStart[read], read * A.cigarStride);
        finalizeSmallRead(A, read, ro, ThreadAligner<EMAX, TRow>{trace, blockDim.x}, true, [](u32, u32 nOptB) { return SmallPrimary{true, libcxxFirstUniform(nOptB)}; });
    }
}

// The reads of the small list that hold a match with more than 3 errors, 16 lanes per read (alignMatchGroup): `ldsPerGroup` bytes of
// dynamic LDS per group of the block
__global__ __launch_bounds__(256) void k_finalize_small_group(FinalizeArgs A, const u32 * __restrict__ smallList, const u32 * __restrict__ smallCount, u32 ldsPerGroup)
{
    extern __shared__ u64 dpTraceLds[];
    const u32 d = threadIdx.x & 15, groupInBlock = threadIdx.x >> 4, groupsPerBlock = blockDim.x >> 4;
    u8 * lds = (u8 *)dpTraceLds + (size_t)groupInBlock * ldsPerGroup;
    const u32 nList = min(*smallCount, A.nReads);          // a list never holds more than every read once
    for (u32 li = blockIdx.x * groupsPerBlock + groupInBlock; li < nList; li += gridDim.x * groupsPerBlock)
    {
        const u32 read = smallList[li];
        ReadOut ro; initReadOut(ro, 3 * A.readStart[read], read * A.cigarStride);
        finalizeSmallRead(A, read, ro, GroupAligner{lds, d}, d == 0, [](u32, u32 nOptB) { return SmallPrimary{true, libcxxFirstUniform(nOptB)}; });
    }
}

// k_align_single for the matches with more than 3 errors, 16 lanes per match
__global__ __launch_bounds__(256) void k_align_group(FinalizeArgs A, const u32 * __restrict__ alignList, const u32 * __restrict__ alignCount, u32 ldsPerGroup)
{
    extern __shared__ u64 dpTraceLds[];
    const u32 d = threadIdx.x & 15, groupInBlock = threadIdx.x >> 4, groupsPerBlock = blockDim.x >> 4;
    u8 * lds = (u8 *)dpTraceLds + (size_t)groupInBlock * ldsPerGroup;
    const u32 nList = min(*alignCount, A.nReads);
    for (u32 li = blockIdx.x * groupsPerBlock + groupInBlock; li < nList; li += gridDim.x * groupsPerBlock)
    {
        const u32 read = alignList[li];
        const u32 s0 = A.readStart[read];
        const u32 readLen = (u32)(A.readOffs[read + 1] - A.readOffs[read]);
        ReadOut ro; initReadOut(ro, 3 * s0, read * A.cigarStride);
        RawMatch a = A.gp[s0];
        const BinDesc & bd = A.bins[a.bin];
        const u8 * ctg = A.contigs + bd.contigOff + A.limits[bd.limOff + a.contig] + a.begin;
        int cb, ce;
        const u32 nc = alignMatchGroup(ctg, (int)a.len, A.readSeqs + A.readOffs[read], readLen, a.strand, (int)a.errors, lds,
                                       A.cigars + ro.cigarBegin, A.cigarStride, cb, ce, d);
        if (d == 0)
        {
            ro.cigarLen = nc < A.cigarStride ? nc : A.cigarStride;
            a.begin += (u32)cb; a.len = (u16)ce;
            writeSingleMatch(A, read, ro, a);
        }
    }
}

// readStart-based regrouping of the raw pool.  A record is only trusted as far as it is consistent with the per-read counts it was
// appended under: a read id outside the batch or more records of a read than were counted would index readStart / cursor / gp with
// an arbitrary value (a GPU memory fault aborts the whole process), so such a record raises `bad` and is dropped instead — the host
// turns the flag into DY_EHIP.  It cannot happen with the kernels in this library; the check costs one load per match.
// nPoolDev (may be NULL): the number of records, where only the device knows it yet (the host then passes the pool's capacity as nPool)
__global__ void k_group_matches(const RawMatch * __restrict__ pool, u32 nPool, const u32 * __restrict__ nPoolDev, u32 nReads, const u32 * __restrict__ readStart, u32 * __restrict__ cursor,
                                RawMatch * __restrict__ gp, u32 * __restrict__ bad)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (nPoolDev) { const u32 n = *nPoolDev; if (n < nPool) nPool = n; }
    if (i >= nPool) return;
    const RawMatch m = pool[i];
    if (m.read >= nReads) { atomicExch(bad, 1u); return; }
    const u32 begin = readStart[m.read], end = readStart[m.read + 1];
    const u32 k = atomicAdd(&cursor[m.read], 1u);
    if (k >= end - begin) { atomicExch(bad, 1u); return; }
    gp[begin + k] = m;
}

} // namespace dy

// K1 — Interleaved Bloom filter k-mer sweep for gfx950 (wave64), plus the k-mer insertion kernel used to build filters.
// Replaces SeqAnBloomFilter::whichBins / _addKmers (reference src/d_bloom_filter.h:161-224, 293-315) and the read loop of
// clasifyLoadedReads (reference src/d_mapper.h:345-371).
//
// Mapping: one wavefront per read; lane = k-mer.  A chunk of 64 consecutive k-mers needs 64+k-1 <= 95 bases; the two
// bit-planes of the 2-bit codes are built with wave ballots (bit t of a plane = base t), so a lane gets its k-mer with
// two funnel shifts, a bit reversal and a Morton interleave — no LDS, no per-base loop.  The look-ups are 64-bit gathers
// ((L-k+1)*h*W*8 B per strand if none were skipped — the figure the roofline is priced on); every sweep below issues only those
// that can still change a `count >= threshold` decision.  Per-bin counts: a bit-sliced adder butterfly across the wave for W = 1
// (waveBitSlicedSum: plane p = bit p of all 64 bin counts, in scalar registers; k_ibf_query_two, k_ibf_query<NP>, k_ibf_query_wide),
// LDS atomics on one counter per bin with a per-word suspect mask for W > 1 (k_ibf_query_lazy, k_ibf_query_eager2).
#pragma once
#include <hip/hip_runtime.h>
#include "dy_types.h"

namespace dy {

__device__ __forceinline__ u64 ballot64(bool p) { return __ballot(p); }

// Spread the low 32 bits of x to the even bit positions.
__device__ __forceinline__ u64 spreadBits(u64 x)
{
    x &= 0xFFFFFFFFull;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8))  & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4))  & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2))  & 0x3333333333333333ull;
    x = (x | (x << 1))  & 0x5555555555555555ull;
    return x;
}

__device__ __forceinline__ u64 waveOr64(u64 v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
    {
        u32 lo = __shfl_xor((u32)v, off, 64), hi = __shfl_xor((u32)(v >> 32), off, 64);
        v |= ((u64)hi << 32) | lo;
    }
    return v;
}

// Hash position (in 64-bit words) of k-mer hash `h` under hash function i (reference src/d_bloom_filter.h:286-291).
__device__ __forceinline__ u64 ibfWordIndex(const IbfGeom & g, u64 kmerHash, u32 i)
{
    // kmer_direct (reference src/d_kdx_filter.h:155-159): the hash is the position; hashes beyond the table (k-mers poisoned
    // by a leading N) land on the all-zero guard position nHashPos that dy_kdx_* allocate behind the table
    if (g.direct) return (kmerHash < g.nHashPos ? kmerHash : g.nHashPos) * g.W;
    u64 v = g.preCalc[i] * kmerHash;
    v ^= v >> 27;
    if (g.hashPosMask) v &= g.hashPosMask;
    else
    {   // v % nHashPos without a 64-bit division: hashPosMagic = floor(2^64 / nHashPos), so the quotient estimate is at most 2 short
        const u64 q = __umul64hi(v, g.hashPosMagic);
        v -= q * g.nHashPos;
        while (v >= g.nHashPos) v -= g.nHashPos;
    }
    return v * g.W;
}

// strand 0: code(idx) = seq[idx]; strand 1: complement of seq[len-1-idx] (N stays N)
__device__ __forceinline__ u32 strandCode(const u8 * __restrict__ seq, u32 len, u32 idx, u32 strand)
{
    if (idx >= len) return 0;
    u32 c = strand ? seq[len - 1 - idx] : seq[idx];
    return (strand && c < 4) ? 3 - c : c;
}

// k-mer hash of lane `lane` in the chunk starting at k-mer `base` of one strand (SeqAn Shape<Dna,SimpleShape>)
__device__ __forceinline__ u64 laneKmerHash(const u8 * __restrict__ seq, u32 len, u32 strand, u32 base, u32 lane, u32 k, u64 kMask, bool firstIsN)
{
    // bit planes of bases [base, base+128)
    u32 c0 = strandCode(seq, len, base + lane, strand), c1 = strandCode(seq, len, base + 64 + lane, strand);
    if (c0 > 3) c0 = 0;
    if (c1 > 3) c1 = 0;
    const u64 p0lo = ballot64(c0 & 1), p1lo = ballot64(c0 & 2), p0hi = ballot64(c1 & 1), p1hi = ballot64(c1 & 2);
    // bits [lane, lane+k) of the 128-bit planes
    u64 x0 = lane ? ((p0lo >> lane) | (p0hi << (64 - lane))) : p0lo;
    u64 x1 = lane ? ((p1lo >> lane) | (p1hi << (64 - lane))) : p1lo;
    x0 &= kMask; x1 &= kMask;
    // base t of the k-mer carries weight 4^(k-1-t): reverse the k bits, then interleave the planes
    const u64 r0 = __brevll(x0) >> (64 - k), r1 = __brevll(x1) >> (64 - k);
    u64 hash = spreadBits(r0) | (spreadBits(r1) << 1);
    const u32 i = base + lane;
    if (firstIsN && 2 * (k + i) < 64) hash += 1ull << (2 * (k + i));
    return hash;
}

// ---- bit-sliced per-bin counters: plane p holds bit p of the count of each of the 64 bins of one mask word ----
// NP planes count up to 2^NP - 1 k-mers: 7 for reads up to ~140 bp, 14 for the 8191-bp maximum

// Sum over the 64 lanes of a one-bit-per-bin mask: butterfly of ripple-carry adders, result replicated in every lane.
__device__ __forceinline__ void waveBitSlicedSum(u64 m, u64 (&s)[7])
{
    s[0] = m;
#pragma unroll
    for (int p = 1; p < 7; ++p) s[p] = 0;
#pragma unroll
    for (int l = 0; l < 6; ++l)
    {
        const int d = 1 << l;
        u64 carry = 0;
#pragma unroll
        for (int p = 0; p <= l; ++p)
        {
            const u64 o = ((u64)__shfl_xor((u32)(s[p] >> 32), d, 64) << 32) | __shfl_xor((u32)s[p], d, 64);
            const u64 x = s[p] ^ o;
            const u64 sum = x ^ carry;
            carry = (s[p] & o) | (carry & x);
            s[p] = sum;
        }
        s[l + 1] = carry;
    }
    // every lane now holds the same totals: move them to scalar registers (the per-strand counters then live on the SALU)
#pragma unroll
    for (int p = 0; p < 7; ++p)
        s[p] = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(s[p] >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)s[p]);
}
// planes += s (s has 7 planes)
template <int NP>
__device__ __forceinline__ void planesAdd(u64 (&planes)[NP], const u64 (&s)[7])
{
    u64 carry = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p)
    {
        const u64 b = p < 7 ? s[p] : 0;
        const u64 x = planes[p] ^ b;
        const u64 sum = x ^ carry;
        carry = (planes[p] & b) | (carry & x);
        planes[p] = sum;
    }
}
// mask of bins whose count >= c
template <int NP>
__device__ __forceinline__ u64 planesGe(const u64 (&planes)[NP], int c)
{
    if (c <= 0) return ~0ull;
    if (c >= (1 << NP)) return 0;
    u64 gt = 0, eq = ~0ull;
#pragma unroll
    for (int p = NP - 1; p >= 0; --p)
    {
        const u64 cb = ((c >> p) & 1) ? ~0ull : 0ull;
        gt |= eq & planes[p] & ~cb;
        eq &= ~(planes[p] ^ cb);
    }
    return gt | eq;
}
template <int NP>
__device__ __forceinline__ u32 planesGet(const u64 (&planes)[NP], u32 j)
{
    u32 v = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) v |= (u32)((planes[p] >> j) & 1) << p;
    return v;
}

// K1.  The answer per bin is only "count >= threshold", so the sweep needs as few gathers as it takes to decide every bin
// exactly (the gather rate, ~55 G/s on a 1 GiB table, is the hardware limit of this kernel):
//   phase A  looks up hash function 0 only.  Its per-bin counts are upper bounds of the real counts (the AND over all
//            hash functions can only clear bits), so a bin whose bound stays below the threshold is rejected for good;
//            the phase stops as soon as no bin can reach the threshold any more.  Counts are bit-sliced: one wave-wide
//            adder butterfly per 64 k-mers, no per-bin loop although every lane's word carries several bits.
//   phase B  verifies the few surviving candidate bins (normally the read's own bin) with the remaining hash
//            functions, 32 k-mers at a time, and stops per bin once count >= threshold or
//            count + (unverified k-mers that carry the bin's bit in word 0) < threshold.
// For a 100-bp read of one bin this is ~83 + ~70 + 64 gathers instead of 2*83*3.
template <int NP>
__global__ __launch_bounds__(256) void k_ibf_query(const u64 * __restrict__ words, IbfGeom g,
                                                   const u8 * __restrict__ seqs, const u64 * __restrict__ offs, u32 nReads,
                                                   u32 threshold, u64 * __restrict__ masks)
{
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 read = blockIdx.x * (blockDim.x >> 6) + wave;
    if (read >= nReads) return;
    const u8 * seq = seqs + offs[read];
    const u32 len = (u32)(offs[read + 1] - offs[read]);
    const u32 k = g.kmerSize;
    u64 * out = masks + (size_t)read * g.W;

    if (len < k || threshold == 0)
    {   // no k-mer: every count is 0; threshold 0 selects every bin (counts >= 0)
        for (u32 w = lane; w < g.W; w += 64)
        {
            const u32 left = g.nBins - w * 64;
            out[w] = threshold == 0 ? (left >= 64 ? ~0ull : ((1ull << left) - 1)) : 0ull;
        }
        return;
    }
    const u32 nKmers = len - k + 1;
    const u64 kMask = (1ull << k) - 1;          // k <= 32 (checked at load/create)

    for (u32 w = 0; w < g.W; ++w)
    {
        const u32 left = g.nBins - w * 64;
        const u64 binMask = left >= 64 ? ~0ull : ((1ull << left) - 1);
        u64 selW = 0;
        for (u32 strand = 0; strand < 2; ++strand)
        {
            const bool firstIsN = strandCode(seq, len, 0, strand) > 3;     // SeqAn hashInit takes text[0] uncast
            // ---- phase A: hash function 0, bit-sliced upper bounds
            u64 planes[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) planes[p] = 0;
            u64 cached0 = 0, cached1 = 0;       // word 0 of this lane's k-mer in chunks 0 and 1
            bool alive = true;
            for (u32 base = 0; base < nKmers && alive; base += 64)
            {
                const u64 hash = laneKmerHash(seq, len, strand, base, lane, k, kMask, firstIsN);
                const bool valid = base + lane < nKmers;
                const u64 m = valid ? words[ibfWordIndex(g, hash, 0) + w] : 0;
                if (base == 0) cached0 = m; else if (base == 64) cached1 = m;
                u64 s[7];
                waveBitSlicedSum(m, s);
                planesAdd(planes, s);
                const u32 remaining = nKmers - min(nKmers, base + 64);
                if (remaining) alive = (planesGe(planes, (int)threshold - (int)remaining) & binMask) != 0;
            }
            if (!alive) continue;               // no bin of this strand can reach the threshold
            // ---- phase B: exact decision for the candidate bins
            u64 und = planesGe(planes, (int)threshold) & binMask;
            while (und)
            {   // one candidate bin at a time (there is normally exactly one: the bin the read comes from)
                const u32 j = __ffsll((long long)und) - 1;
                und &= und - 1;
                const u32 bound0 = planesGet(planes, j);
                u32 exactCnt = 0, seenCnt = 0;
                bool open = true;
                for (u32 base = 0; base < nKmers && open; base += 64)
                {
                    const u64 hash = laneKmerHash(seq, len, strand, base, lane, k, kMask, firstIsN);
                    const bool valid = base + lane < nKmers;
                    u64 m0;
                    if (base == 0) m0 = cached0;
                    else if (base == 64) m0 = cached1;
                    else m0 = valid ? words[ibfWordIndex(g, hash, 0) + w] : 0;
                    const bool has0 = valid && ((m0 >> j) & 1);
                    // verify only as many k-mers at a time as are still needed to reach the threshold (29 of the 64 for a clean read)
                    u64 pending = ballot64(has0);
                    while (pending && open)
                    {
                        const u32 need = threshold - exactCnt;
                        const bool act = ((pending >> lane) & 1) && (u32)__popcll(pending & ((1ull << lane) - 1)) < need;
                        bool hit = act;
                        if (act)
                            for (u32 hf = 1; hf < g.nHashes; ++hf) hit &= (words[ibfWordIndex(g, hash, hf) + w] >> j) & 1;
                        const u64 actMask = ballot64(act);
                        exactCnt += (u32)__popcll(ballot64(hit));
                        seenCnt += (u32)__popcll(actMask);
                        pending &= ~actMask;
                        if (exactCnt >= threshold) { selW |= 1ull << j; open = false; }
                        else if (exactCnt + (bound0 - seenCnt) < threshold) open = false;
                    }
                }
            }
        }
        if (lane == 0) out[w] = selW;
    }
}

// K1 for the headline shape: 64 bins or fewer (W = 1) and 65..127 k-mers per read (reads of 82..144 bases at k = 18).  Same
// answers as k_ibf_query<7>; the second chunk of k-mers is fetched only as far as the decision needs it.
// tail k-mers per round: measured on 64 bins / 1 GiB / 100 bp (K1 per 1 M reads): 1 -> 3.81 ms, 2 -> 3.61, 3 -> 3.58, 4 -> 3.57, 5 -> 3.60, 8 -> 3.66, 12 -> 3.76,
// 19 (the whole tail at once) -> 3.94: the sweep is bound by the look-ups it makes, a round trip more costs less than four look-ups too many
#ifndef IBF_TAIL_STEP
#define IBF_TAIL_STEP 4
#endif
// (tailPre, from the host: tail k-mers requested with the first round — 4 where weak bins are the rule, 0 where the filter is so sparse
// that they are the exception: at 16 GiB / 64 bins the four look-ups per strand cost 3.8 % and save nothing)
__global__ __launch_bounds__(256) void k_ibf_query_two(const u64 * __restrict__ words, IbfGeom g,
                                                       const u8 * __restrict__ seqs, const u64 * __restrict__ offs, u32 nReads,
                                                       u32 threshold, u64 * __restrict__ masks, u32 slack, u32 tailPre)
{
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 read = blockIdx.x * (blockDim.x >> 6) + wave;
    if (read >= nReads) return;
    const u8 * seq = seqs + offs[read];
    const u32 len = (u32)(offs[read + 1] - offs[read]);
    const u32 k = g.kmerSize;
    const u64 binMask = g.nBins >= 64 ? ~0ull : ((1ull << g.nBins) - 1);
    if (len < k || threshold == 0)
    {
        if (lane == 0) masks[read] = threshold == 0 ? binMask : 0ull;
        return;
    }
    const u32 nKmers = len - k + 1;             // at most 127, checked by the host for the whole batch
    if (threshold > nKmers) { if (lane == 0) masks[read] = 0ull; return; }       // more hits needed than there are k-mers
    const u64 kMask = (1ull << k) - 1;
    // first round: the k-mers after which a bin without hits is out of reach, plus `slack` for the false hits a bin collects by then
    // (the host derives it from the filter's fill); at most 64 (one per lane), and the tail must fit the lanes too
    u32 n0 = min(min(nKmers, 64u), max(threshold, nKmers - threshold + slack));
    if (nKmers > 64 && n0 < nKmers - 64) n0 = nKmers - 64;
    const u32 tail = nKmers - n0;               // shorter reads of the batch simply have no tail
    // The kernel is a chain of dependent gather rounds, so both strands go through every stage together: one round for the first
    // 64 k-mers of both, one per tail group of both, then the verification.
    // After the first 64 k-mers a bin is STRONG (count >= threshold: its verification can succeed on those k-mers alone), WEAK (it
    // needs tail hits to reach the threshold) or dead.  Weak bins are false candidates almost always; the tail is gathered IBF_TAIL_STEP k-mers
    // at a time until they are all dead (bound = hits so far + k-mers not yet fetched).
    bool firstIsN[2]; u64 hashA[2], m0[2], hashB[2] = {0, 0}, m1[2] = {0, 0};
    u64 c0p[2][7]; u64 strong[2], weak[2]; u32 fetched[2] = {0, 0}; bool haveHashB[2] = {false, false};
    // Nearly every strand leaves the first round with a weak bin or two (64 bins: one of them reaches threshold - tail hits by chance), so
    // the first tailPre tail k-mers are requested WITH the first round instead of in a round of their own behind it.
    const u32 pre = min(tail, tailPre);
#pragma unroll
    for (int s = 0; s < 2; ++s)
    {
        firstIsN[s] = strandCode(seq, len, 0, s) > 3;
        hashA[s] = laneKmerHash(seq, len, s, 0, lane, k, kMask, firstIsN[s]);
        if (pre) { hashB[s] = laneKmerHash(seq, len, s, n0, lane, k, kMask, firstIsN[s]); haveHashB[s] = true; }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
    {
        m0[s] = lane < n0 ? words[ibfWordIndex(g, hashA[s], 0)] : 0;
        if (lane < pre) m1[s] = words[ibfWordIndex(g, hashB[s], 0)];
        fetched[s] = pre;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
    {
        waveBitSlicedSum(m0[s], c0p[s]);
        const u64 alive0 = planesGe(c0p[s], (int)threshold - (int)tail) & binMask;
        strong[s] = planesGe(c0p[s], (int)threshold) & binMask;
        weak[s] = alive0 & ~strong[s];
        // what the k-mers fetched ahead already decide
        u64 todo = pre ? weak[s] : 0;
        while (todo)
        {
            const u32 j = (u32)__ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const u32 c1 = (u32)__popcll(ballot64((m1[s] >> j) & 1));
            if (planesGet(c0p[s], j) + c1 + (tail - fetched[s]) < threshold) weak[s] &= ~(1ull << j);
        }
    }
    if (weak[0] | weak[1])
    {
#pragma unroll
        for (int s = 0; s < 2; ++s)
            if (weak[s] && !haveHashB[s]) { hashB[s] = laneKmerHash(seq, len, s, n0, lane, k, kMask, firstIsN[s]); haveHashB[s] = true; }
        while ((weak[0] && fetched[0] < tail) || (weak[1] && fetched[1] < tail))
        {
#pragma unroll
            for (int s = 0; s < 2; ++s)
                if (weak[s] && fetched[s] < tail)
                {
                    const u32 upTo = min(tail, fetched[s] + (u32)IBF_TAIL_STEP);
                    if (lane >= fetched[s] && lane < upTo) m1[s] = words[ibfWordIndex(g, hashB[s], 0)];
                    fetched[s] = upTo;
                }
#pragma unroll
            for (int s = 0; s < 2; ++s)
            {
                u64 todo = weak[s];
                while (todo)
                {
                    const u32 j = (u32)__ffsll((long long)todo) - 1;
                    todo &= todo - 1;
                    const u32 c1 = (u32)__popcll(ballot64((m1[s] >> j) & 1));
                    if (planesGet(c0p[s], j) + c1 + (tail - fetched[s]) < threshold) weak[s] &= ~(1ull << j);
                }
            }
        }
    }
    // ---- verification of strong and surviving weak bins (a surviving weak bin has count0 + count1 >= threshold, tail complete)
    u64 selW = 0;
#pragma unroll
    for (int s = 0; s < 2; ++s)
    {
        u64 und = strong[s] | weak[s];
        while (und)
        {
            const u32 j = __ffsll((long long)und) - 1;
            und &= und - 1;
            const u32 c0 = planesGet(c0p[s], j);
            u32 exactCnt = 0, seenCnt = 0;
            bool open = true;
            for (u32 chunk = 0; chunk < 2 && open; ++chunk)
            {
                if (chunk == 1 && fetched[s] < tail)
                {   // the verification of a strong bin did not reach the threshold on the first 64 k-mers: complete the tail
                    if (!haveHashB[s]) { hashB[s] = laneKmerHash(seq, len, s, n0, lane, k, kMask, firstIsN[s]); haveHashB[s] = true; }
                    if (lane >= fetched[s] && lane < tail) m1[s] = words[ibfWordIndex(g, hashB[s], 0)];
                    fetched[s] = tail;
                }
                const u64 hash = chunk ? hashB[s] : hashA[s];
                const u64 mw = chunk ? m1[s] : m0[s];
                // upper bound of the whole count: hits among the fetched k-mers + every k-mer not fetched yet
                const u32 bound0 = c0 + (u32)__popcll(ballot64((m1[s] >> j) & 1)) + (tail - fetched[s]);
                u64 pending = ballot64((mw >> j) & 1);
                while (pending && open)
                {
                    const u32 need = threshold - exactCnt;
                    const bool act = ((pending >> lane) & 1) && (u32)__popcll(pending & ((1ull << lane) - 1)) < need;
                    bool hit = act;
                    if (act)
                        for (u32 hf = 1; hf < g.nHashes; ++hf) hit &= (words[ibfWordIndex(g, hash, hf)] >> j) & 1;
                    const u64 actMask = ballot64(act);
                    exactCnt += (u32)__popcll(ballot64(hit));
                    seenCnt += (u32)__popcll(actMask);
                    pending &= ~actMask;
                    if (exactCnt >= threshold) { selW |= 1ull << j; open = false; }
                    else if (exactCnt + (bound0 - seenCnt) < threshold) open = false;
                }
            }
        }
    }
    if (lane == 0) masks[read] = selW;
}

// ---- K1 for filters with more than 64 bins (W >= 2 words per hash position) ------------------------------------------
// A hash position is W consecutive words (32 B at 256 bins, 128 B at 1024 bins), so G = min(pow2 >= W, 16) adjacent lanes
// serve one k-mer and a sub-pass of the wave covers 64/G k-mers with fully used 32..128-byte requests; every lane keeps
// its own bit-sliced counters (one word of one k-mer slot) and the 64/G slots are summed by a butterfly once per phase.
//   phase A  hash function 0 for every k-mer; the words are kept in LDS.  Upper bounds -> candidate bins.
//   phase B  one sweep over the k-mers with lazy gathers: a lane fetches hash function 1.. only while its word still
//            carries a candidate bit; exact counts; the sweep stops once every candidate is decided.
// Filters with more than 16 words are handled 16 words at a time (the words of a position are independent).
__device__ __forceinline__ u64 shflXor64(u64 v, int d)
{
    return ((u64)__shfl_xor((u32)(v >> 32), d, 64) << 32) | __shfl_xor((u32)v, d, 64);
}
// planes += m (a one-bit number per bin)
template <int NP>
__device__ __forceinline__ void planesAddBit(u64 (&planes)[NP], u64 m)
{
    u64 carry = m;
#pragma unroll
    for (int p = 0; p < NP; ++p) { const u64 t = planes[p] & carry; planes[p] ^= carry; carry = t; }
}
// planes += m0 + ... + m7 (eight one-bit numbers per bin): carry-save tree to a 4-bit number, then one ripple add
template <int NP>
__device__ __forceinline__ void planesAdd8(u64 (&planes)[NP], const u64 (&m)[8])
{
    auto fa = [](u64 a, u64 b, u64 c, u64 & sum, u64 & carry) { const u64 x = a ^ b; sum = x ^ c; carry = (a & b) | (x & c); };
    u64 s0, c0, s1, c1, s2, c2, ones, c3, t0, f0, twos, f1;
    fa(m[0], m[1], m[2], s0, c0);
    fa(m[3], m[4], m[5], s1, c1);
    s2 = m[6] ^ m[7]; c2 = m[6] & m[7];
    fa(s0, s1, s2, ones, c3);
    fa(c0, c1, c2, t0, f0);
    twos = t0 ^ c3; f1 = t0 & c3;
    const u64 fours = f0 ^ f1, eights = f0 & f1;
    const u64 add[4] = {ones, twos, fours, eights};
    u64 carry = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p)
    {
        const u64 v = p < 4 ? add[p] : 0;
        const u64 x = planes[p] ^ v;
        const u64 sum = x ^ carry;
        carry = (planes[p] & v) | (carry & x);
        planes[p] = sum;
    }
}
// sum of the per-lane counters over the lanes that differ in bits >= log2(G): every lane ends with its word's total.
// `width` = planes that can be non-zero in a single lane's counter; every butterfly step adds one.
template <int NP>
__device__ __forceinline__ void planesReduceSlots(u64 (&planes)[NP], u32 G, u32 width = NP)
{
    for (u32 d = G; d < 64; d <<= 1)
    {
        u64 carry = 0;
#pragma unroll
        for (int p = 0; p < NP; ++p)
        {
            if ((u32)p <= width)            // uniform: planes above the current width are zero on both sides
            {
                const u64 o = shflXor64(planes[p], (int)d);
                const u64 x = planes[p] ^ o;
                const u64 sum = x ^ carry;
                carry = (planes[p] & o) | (carry & x);
                planes[p] = sum;
            }
        }
        if (width < NP) ++width;
    }
}

template <int NP, bool SPARSE>
__global__ __launch_bounds__(256) void k_ibf_query_wide(const u64 * __restrict__ words, IbfGeom g,
                                                        const u8 * __restrict__ seqs, const u64 * __restrict__ offs, u32 nReads,
                                                        u32 threshold, u64 * __restrict__ masks, u32 gShift, u32 cacheKmers, u32 hashSlots, u32 sparseCap)
{
    extern __shared__ u64 ldsWide[];
    constexpr u32 UA = 8, UB = 4;   // sub-passes whose gathers are in flight together (phase A / phase B)
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 read = blockIdx.x * (blockDim.x >> 6) + wave;
    if (read >= nReads) return;
    const u8 * seq = seqs + offs[read];
    const u32 len = (u32)(offs[read + 1] - offs[read]);
    const u32 k = g.kmerSize;
    u64 * out = masks + (size_t)read * g.W;
    if (len < k || threshold == 0)
    {
        for (u32 w = lane; w < g.W; w += 64)
        {
            const u32 left = g.nBins - w * 64;
            out[w] = threshold == 0 ? (left >= 64 ? ~0ull : ((1ull << left) - 1)) : 0ull;
        }
        return;
    }
    const u32 nKmers = len - k + 1;
    const u64 kMask = (1ull << k) - 1;
    const u32 G = 1u << gShift, slots = 64u >> gShift, slot = lane >> gShift, wl = lane & (G - 1);
    // LDS per wave: the k-mer hashes of one strand (all of them when hashSlots >= nKmers, else a window of hashSlots = 128) and
    // the hash-0 words of the first cacheKmers k-mers
    // ... and, when sparseCap != 0, the scratch of the sparse verification: candidate mask per word (16 x u64), byte counters per
    // (word, bin) (16 x 64 bytes = 128 x u64) and a list of sparseCap 16-bit (k-mer, word) pairs
    const size_t perWave = hashSlots + (size_t)cacheKmers * G + (sparseCap ? 16 + 128 + sparseCap / 4 : 0);
    u64 * myHash = ldsWide + (size_t)wave * perWave;
    u64 * myW0 = myHash + hashSlots;
    u64 * spCand = myW0 + (size_t)cacheKmers * G;
    u32 * spCount = (u32 *)(spCand + 16);
    u16 * spList = (u16 *)(spCand + 16 + 128);
    const bool hashAll = hashSlots >= nKmers;
    const u32 CH = hashAll ? ((nKmers + 63) & ~63u) : hashSlots;        // k-mers per outer iteration
    u32 perLaneBits = 1; while ((1u << perLaneBits) <= (nKmers + slots - 1) / slots) ++perLaneBits;   // bits of a lane's own counter

    for (u32 w0 = 0; w0 < g.W; w0 += G)
    {
        const u32 w = w0 + wl;
        const bool wOk = w < g.W;
        const u32 left = wOk ? g.nBins - w * 64 : 0;
        const u64 binMask = left >= 64 ? ~0ull : ((1ull << left) - 1);
        u64 selW = 0;
        for (u32 strand = 0; strand < 2; ++strand)
        {
            const bool firstIsN = strandCode(seq, len, 0, strand) > 3;
            // ---- phase A
            u64 planes[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) planes[p] = 0;
            for (u32 base = 0; base < nKmers; base += CH)
            {
                __builtin_amdgcn_wave_barrier();
                for (u32 h0 = 0; h0 < CH && base + h0 < nKmers; h0 += 64)
                    myHash[h0 + lane] = laneKmerHash(seq, len, strand, base + h0, lane, k, kMask, firstIsN);
                __builtin_amdgcn_wave_barrier();
                const u32 inChunk = min(CH, nKmers - base);
                for (u32 s0 = 0; s0 * slots < inChunk; s0 += UA)
                {
                    u64 m[UA];
#pragma unroll
                    for (u32 u = 0; u < UA; ++u)
                    {
                        const u32 kl = (s0 + u) * slots + slot;
                        const bool valid = kl < inChunk && wOk;
                        m[u] = valid ? words[ibfWordIndex(g, myHash[kl < CH ? kl : 0], 0) + w] : 0;
                    }
#pragma unroll
                    for (u32 u = 0; u < UA; ++u)
                    {
                        const u32 kl = (s0 + u) * slots + slot;
                        if (kl < inChunk && base + kl < cacheKmers) myW0[(size_t)(base + kl) * G + wl] = m[u];
                    }
                    planesAdd8(planes, m);
                }
            }
            planesReduceSlots(planes, G, perLaneBits);
            const u64 cand = planesGe(planes, (int)threshold) & binMask;
            const u64 candLanes = ballot64(cand != 0);
            if (candLanes == 0) continue;
            // one candidate bin in the whole group (the usual case at high thresholds): its exact count is a popcount of ballots
            const bool single = (u32)__popcll(candLanes) == slots && ballot64((cand & (cand - 1)) != 0) == 0;
            // few candidate words: fetch the remaining hash functions together; many (low thresholds): one after the other, lazily
            const bool eager = __popcll(candLanes) <= (int)(2 * slots);
            // ---- phase B, sparse form for many candidates (low thresholds: dozens of false candidates, each carried by a handful of
            // k-mers): only 1 in ~6 (k-mer, word) slots holds a candidate bit, so the slots that do are compacted into a list and
            // verified 64 at a time; exact counts are byte counters in LDS.  Needs all hashes in LDS and counts below 256.
            if constexpr (SPARSE) if (!single && sparseCap && hashAll && nKmers < 256)
            {
                __builtin_amdgcn_wave_barrier();
                if (slot == 0) spCand[wl] = cand;
                for (u32 i = lane; i < 256; i += 64) spCount[i] = 0;
                u32 nPairs = 0;
                bool overflow = false;
                for (u32 s0 = 0; s0 * slots < nKmers; ++s0)
                {
                    const u32 kk = s0 * slots + slot;
                    u64 c0 = 0;
                    if (kk < nKmers && cand != 0)
                        c0 = (kk < cacheKmers ? myW0[(size_t)kk * G + wl] : words[ibfWordIndex(g, myHash[kk], 0) + w]) & cand;
                    const u64 act = ballot64(c0 != 0);
                    if (nPairs + (u32)__popcll(act) > sparseCap) { overflow = true; break; }
                    if (c0 != 0) spList[nPairs + (u32)__popcll(act & ((1ull << lane) - 1))] = (u16)((kk << 4) | wl);
                    nPairs += (u32)__popcll(act);
                }
                if (!overflow)
                {
                    __builtin_amdgcn_wave_barrier();
                    for (u32 p0 = 0; p0 < nPairs; p0 += 64)
                    {
                        const u32 pi = p0 + lane;
                        if (pi < nPairs)
                        {
                            const u32 e = spList[pi], kk = e >> 4, pw = e & 15;
                            const u64 hk = myHash[kk];
                            u64 c = (kk < cacheKmers ? myW0[(size_t)kk * G + pw] : words[ibfWordIndex(g, hk, 0) + w0 + pw]) & spCand[pw];
                            for (u32 hf = 1; hf < g.nHashes && c; ++hf) c &= words[ibfWordIndex(g, hk, hf) + w0 + pw];
                            while (c)
                            {
                                const u32 b = (u32)__ffsll((long long)c) - 1;
                                c &= c - 1;
                                atomicAdd(&spCount[(pw * 64 + b) >> 2], 1u << (8 * (b & 3)));
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    u64 rest = cand, acc = 0;
                    while (rest)
                    {
                        const u32 b = (u32)__ffsll((long long)rest) - 1;
                        rest &= rest - 1;
                        if (((spCount[(wl * 64 + b) >> 2] >> (8 * (b & 3))) & 0xffu) >= threshold) acc |= 1ull << b;
                    }
                    selW |= acc;
                    __builtin_amdgcn_wave_barrier();
                    continue;
                }
            }
            // ---- phase B
#pragma unroll
            for (int p = 0; p < NP; ++p) planes[p] = 0;
            u32 nextCheck = threshold + threshold / 4, exactOne = 0;
            bool done = false, acceptOne = false;
            for (u32 base = 0; base < nKmers && !done; base += CH)
            {
                if (!hashAll)
                {
                    __builtin_amdgcn_wave_barrier();
                    for (u32 h0 = 0; h0 < CH && base + h0 < nKmers; h0 += 64)
                        myHash[h0 + lane] = laneKmerHash(seq, len, strand, base + h0, lane, k, kMask, firstIsN);
                    __builtin_amdgcn_wave_barrier();
                }
                const u32 inChunk = min(CH, nKmers - base);
                for (u32 s0 = 0; s0 * slots < inChunk; s0 += UB)
                {
                    u64 c[UB], hk[UB];
#pragma unroll
                    for (u32 u = 0; u < UB; ++u)
                    {
                        const u32 kl = (s0 + u) * slots + slot;
                        const u32 kk = base + kl;
                        const bool valid = kl < inChunk && cand != 0;
                        hk[u] = myHash[kl < CH ? kl : 0];
                        c[u] = 0;
                        if (valid) c[u] = (kk < cacheKmers ? myW0[(size_t)kk * G + wl] : words[ibfWordIndex(g, hk[u], 0) + w]) & cand;
                    }
                    if (eager && g.nHashes == 3)
                    {
                        u64 v1[UB], v2[UB];
#pragma unroll
                        for (u32 u = 0; u < UB; ++u)
                        {
                            v1[u] = c[u] ? words[ibfWordIndex(g, hk[u], 1) + w] : 0;
                            v2[u] = c[u] ? words[ibfWordIndex(g, hk[u], 2) + w] : 0;
                        }
#pragma unroll
                        for (u32 u = 0; u < UB; ++u) c[u] &= v1[u] & v2[u];
                    }
                    else
                        for (u32 hf = 1; hf < g.nHashes; ++hf)
                        {
                            u64 v[UB];
#pragma unroll
                            for (u32 u = 0; u < UB; ++u) v[u] = c[u] ? words[ibfWordIndex(g, hk[u], hf) + w] : 0;
#pragma unroll
                            for (u32 u = 0; u < UB; ++u) c[u] &= v[u];
                        }
                    const u32 processed = min(nKmers, base + min(CH, (s0 + UB) * slots));
                    if (single)
                    {
#pragma unroll
                        for (u32 u = 0; u < UB; ++u) exactOne += (u32)__popcll(ballot64(c[u] != 0));
                        if (exactOne >= threshold) { acceptOne = true; done = true; break; }
                        if (exactOne + (nKmers - processed) < threshold) { done = true; break; }
                        continue;
                    }
#pragma unroll
                    for (u32 u = 0; u < UB; ++u)
                        if (ballot64(c[u] != 0)) planesAddBit(planes, c[u]);
                    if (processed >= nextCheck && processed < nKmers)
                    {   // every candidate decided?  exact so far >= threshold, or out of reach even if all remaining k-mers hit
                        u64 e[NP];
#pragma unroll
                        for (int p = 0; p < NP; ++p) e[p] = planes[p];
                        planesReduceSlots(e, G);
                        const u64 open = cand & ~planesGe(e, (int)threshold) & planesGe(e, (int)threshold - (int)(nKmers - processed));
                        if (ballot64(open != 0) == 0) { done = true; break; }
                        nextCheck = processed + max(threshold / 2, 32u);
                    }
                }
            }
            if (single) { if (acceptOne) selW |= cand; }
            else
            {
                planesReduceSlots(planes, G);
                selW |= planesGe(planes, (int)threshold) & cand;
            }
        }
        if (slot == 0 && wOk) out[w] = selW;
    }
}

// ---- K1 for more than 64 bins and reads of at most 255 k-mers ----------------------------------------------------------------------
// Measured (scripts/ubench_gather.hip, profiles/r02_gather_ceiling.json): the chip serves 48-60 G independent random requests per
// second whatever their width between 8 and 128 bytes.  A hash position of W <= 16 words is ONE request, so the cost of the sweep is
// the NUMBER of (k-mer, hash function) positions it touches — and, right behind it, vector instructions.  Both kernels below
//   * compute the position of a k-mer once per hash function, lane = k-mer (64 at a time), and share it through LDS: the G = pow2(W)
//     lanes that serve one position read it back instead of redoing the 64-bit hash arithmetic;
//   * count set bits (sparse: ~2 % of a column, plus the read's own bin) with LDS atomics, one u32 counter per bin, and let the atomic
//     that lifts a counter to the bar of interest mark the bin in a per-word SUSPECT mask — decisions then look at the handful of
//     suspects only, every lane for its own word, without any cross-lane reduction.
// Same answers as whichBins (reference src/d_bloom_filter.h:161-224) for every input: each decision is exact, only the order of the
// look-ups differs.
//
// k_ibf_query_lazy (thresholds high enough that false candidates are rare, e.g. 100 bp / threshold 29):
//   phase A  hash function 0 only, k-mers in order, words kept in LDS.  After n k-mers a bin is ALIVE if count0 + (nK - n) >= threshold
//            and STRONG if count0 >= threshold; the phase stops as soon as every alive bin is strong (after ~nK - threshold + slack
//            k-mers only the read's own bin is left) or nothing is alive.
//   phase B  the strong bins are verified lazily, k-mers in order: a position of hash function 1.. is fetched only while its word
//            still carries an undecided bin; accepted once exact >= threshold, rejected once exact + unverified carriers + k-mers
//            never looked at < threshold.  A single undecided bin has only as many carriers verified as it still needs, all
//            remaining hash functions in one round.  K-mers beyond phase A are fetched on demand.
//   100 bp / threshold 29 / 256 bins: ~62 + 62 + 2*29 = 182 positions per read instead of 2*83*3 = 498 (k_ibf_query_wide: 288).
// k_ibf_query_eager2 (low thresholds, e.g. 150 bp / threshold 7 / 1024 bins: dozens of false candidate bins after one hash function,
//   each of which k_ibf_query_lazy would have to refute through chains of dependent look-ups):
//   stage 1  hash functions 0 AND 1 of every k-mer, fetched together (no dependency between any two requests), AND-ed: the
//            survivors are the read's own bin plus ~0.5 false bits per k-mer; they are counted and the non-zero words kept in a list.
//   stage 2  the suspects (count >= threshold: the read's own bin, practically) are verified against the remaining hash functions on
//            their listed carriers, only as many as needed.
//   150 bp / threshold 7: 532 + ~7 positions per read instead of 798 (k_ibf_query_wide: ~1330 requests).
__device__ __forceinline__ u32 ldsByteGet(const u32 * __restrict__ c, u32 idx) { return (c[idx >> 2] >> (8 * (idx & 3))) & 0xffu; }
__device__ __forceinline__ void ldsByteInc(u32 * __restrict__ c, u32 idx) { atomicAdd(&c[idx >> 2], 1u << (8 * (idx & 3))); }
// 16-bit counters, two to an LDS word (counts stay below 256: no carry between the halves); the increment returns the old value
__device__ __forceinline__ u32 ldsU16Inc(u32 * __restrict__ c, u32 idx) { const u32 sh = (idx & 1) * 16; return (atomicAdd(&c[idx >> 1], 1u << sh) >> sh) & 0xFFFFu; }
__device__ __forceinline__ u32 ldsU16Get(const u32 * __restrict__ c, u32 idx) { return (c[idx >> 1] >> ((idx & 1) * 16)) & 0xFFFFu; }
// every set bit of `bits` (a word of bins [binBase, binBase + 64)) bumps its bin's counter; the bump that reaches `bar` marks the bin
__device__ __forceinline__ void countBitsMark(u32 * __restrict__ counters, u64 * __restrict__ suspectWord, u32 binBase, u64 bits, u32 bar)
{
    u32 lo = (u32)bits, hi = (u32)(bits >> 32);
    while (lo) { const u32 b = (u32)__ffs((int)lo) - 1; lo &= lo - 1; if (ldsU16Inc(counters, binBase + b) + 1 == bar) atomicOr((unsigned long long *)suspectWord, 1ull << b); }
    while (hi) { const u32 b = (u32)__ffs((int)hi) + 31; hi &= hi - 1; if (ldsU16Inc(counters, binBase + b) + 1 == bar) atomicOr((unsigned long long *)suspectWord, 1ull << b); }
}
__device__ __forceinline__ void countBitsBytes(u32 * __restrict__ counters, u32 binBase, u64 bits)
{
    while (bits) { const u32 b = (u32)__ffsll((long long)bits) - 1; bits &= bits - 1; ldsByteInc(counters, binBase + b); }
}

__global__ __launch_bounds__(256) void k_ibf_query_lazy(const u64 * __restrict__ words, IbfGeom g,
                                                        const u8 * __restrict__ seqs, const u64 * __restrict__ offs, u32 nReads,
                                                        u32 threshold, u64 * __restrict__ masks, u32 gShift, u32 nKcap, u32 slack)
{
    extern __shared__ u64 ldsLazy[];
    constexpr u32 UA = 8, UB = 4;   // sub-passes whose gathers are in flight together (phase A / phase B)
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 read = blockIdx.x * (blockDim.x >> 6) + wave;
    if (read >= nReads) return;
    const u8 * seq = seqs + offs[read];
    const u32 len = (u32)(offs[read + 1] - offs[read]);
    const u32 k = g.kmerSize;
    u64 * out = masks + (size_t)read * g.W;
    const u32 nKmers = len >= k ? len - k + 1 : 0;
    if (nKmers == 0 || threshold == 0 || threshold > nKmers)
    {   // no k-mer: every count is 0; threshold 0 selects every bin; more hits needed than there are k-mers: none
        for (u32 w = lane; w < g.W; w += 64)
        {
            const u32 left = g.nBins - w * 64;
            out[w] = threshold == 0 ? (left >= 64 ? ~0ull : ((1ull << left) - 1)) : 0ull;
        }
        return;
    }
    const u64 kMask = (1ull << k) - 1;
    const u32 G = 1u << gShift, slots = 64u >> gShift, slot = lane >> gShift, wl = lane & (G - 1);
    const u32 hashSlots = (nKcap + 31) & ~31u;
    // LDS per wave (it decides the waves per CU): k-mer hashes | positions of hash function 0 | positions of the round's window under
    // two further hash functions | hash-0 words w0c[kmer * G + word] | cnt0: u16 per (word, bin), hash-0 hits among the k-mers of phase A |
    // exc, seen: bytes per (word, bin), exact hits among the verified carriers / verified carriers of phase A | suspect mask per word
    const size_t perWave = 2 * (size_t)hashSlots + 128 + (size_t)nKcap * G + 16 * G + 16 * G + G;
    u64 * myHash = ldsLazy + (size_t)wave * perWave;
    u64 * myPos0 = myHash + hashSlots;
    u64 * myPosX = myPos0 + hashSlots;                // [2][64]
    u64 * w0c = myPosX + 128;
    u32 * cnt0 = (u32 *)(w0c + (size_t)nKcap * G);    // 64 G 16-bit counters = 32 G words
    u32 * exc = cnt0 + 32 * G;                        // 16 G words of 4 byte counters
    u32 * seen = exc + 16 * G;
    u64 * susp = (u64 *)(seen + 16 * G);              // G words

    for (u32 w0 = 0; w0 < g.W; w0 += G)
    {
        const u32 w = w0 + wl;
        const bool wOk = w < g.W;
        const u32 left = wOk ? g.nBins - w * 64 : 0;
        const u64 binMask = left >= 64 ? ~0ull : ((1ull << left) - 1);
        u64 selW = 0;
        for (u32 strand = 0; strand < 2; ++strand)
        {
            const bool firstIsN = strandCode(seq, len, 0, strand) > 3;
            __builtin_amdgcn_wave_barrier();
            for (u32 h0 = 0; h0 < nKmers; h0 += 64)
            {
                const u64 h = laneKmerHash(seq, len, strand, h0, lane, k, kMask, firstIsN);
                if (h0 + lane < nKmers) { myHash[h0 + lane] = h; myPos0[h0 + lane] = ibfWordIndex(g, h, 0); }
            }
            for (u32 i = lane; i < (32 + 32) * G + 2 * G; i += 64) cnt0[i] = 0;      // cnt0, exc, seen, susp are contiguous
            __builtin_amdgcn_wave_barrier();
            // ---- phase A
            // first stop where the false bins are expected to have died (slack = false hits a bin may collect by then), then every 8 k-mers
            const u32 n1 = min(nKmers, max(nKmers - threshold + slack, threshold));
            const u32 bar1 = threshold - (nKmers - n1);           // >= 1: a bin below it at any stop is out of reach
            u32 fetched = 0;
            u64 strong = 0;
            bool anyStrong = false;
            for (;;)
            {
                const u32 target = fetched == 0 ? n1 : min(nKmers, fetched + 8u);
                for (u32 kb = fetched; kb < target; kb += UA * slots)
                {
                    u64 m[UA];
#pragma unroll
                    for (u32 u = 0; u < UA; ++u)
                    {
                        const u32 kk = kb + u * slots + slot;
                        m[u] = (kk < target && wOk) ? words[myPos0[kk] + w] : 0;
                    }
#pragma unroll
                    for (u32 u = 0; u < UA; ++u)
                    {
                        const u32 kk = kb + u * slots + slot;
                        if (kk < target) w0c[(size_t)kk * G + wl] = m[u];
                        countBitsMark(cnt0, susp + wl, wl * 64, m[u] & binMask, bar1);
                    }
                }
                fetched = target;
                __builtin_amdgcn_wave_barrier();
                // every lane decides for its own word (the lanes of the other slots do the same work on the same data)
                const u32 rest = nKmers - fetched;
                u64 alive = 0; strong = 0;
                for (u64 todo = susp[wl]; todo; todo &= todo - 1)
                {
                    const u32 b = (u32)__ffsll((long long)todo) - 1;
                    const u32 c = ldsU16Get(cnt0, wl * 64 + b);
                    if (c + rest >= threshold) alive |= 1ull << b;
                    if (c >= threshold) strong |= 1ull << b;
                }
                anyStrong = ballot64(strong != 0) != 0;
                if (ballot64(alive != 0) == 0) { anyStrong = false; break; }
                if (fetched == nKmers || ballot64((alive & ~strong) != 0) == 0) break;
            }
            if (!anyStrong) continue;
            // ---- phase B: lazy verification of the strong bins.  K-mers [0, processed) are done with: every carrier of an undecided bin
            // among them has been verified.  bound(bin) = exact + unverified carriers of phase A + k-mers never looked at.
            u64 und = strong;                         // undecided bins of word wl (the same in every slot)
            u32 processed = 0;
            while (processed < nKmers)
            {
                const u64 undLanes = ballot64(und != 0 && slot == 0);
                if (undLanes == 0) break;
                // a single undecided bin: verify only as many of its carriers as it still needs, all remaining hash functions at once
                u32 need = 0xFFFFFFFFu;
                if ((undLanes & (undLanes - 1)) == 0)
                {
                    const u32 ownerLane = (u32)__ffsll((long long)undLanes) - 1;          // slot 0: ownerLane = the bin's word
                    const u32 lo = __shfl((u32)und, (int)ownerLane, 64), hi = __shfl((u32)(und >> 32), (int)ownerLane, 64);
                    const u64 uo = ((u64)hi << 32) | lo;
                    if ((uo & (uo - 1)) == 0) need = threshold - ldsByteGet(exc, ownerLane * 64 + (u32)__ffsll((long long)uo) - 1);
                }
                // a round looks at up to 64 k-mers; their positions under the next two hash functions, lane = k-mer
                const u32 span = min(64u, UB * slots);
                const u32 lastK = min(nKmers, processed + span);
                __builtin_amdgcn_wave_barrier();
                if (g.nHashes > 1 && processed + lane < lastK)
                {
                    const u64 h = myHash[processed + lane];
                    myPosX[lane] = ibfWordIndex(g, h, 1);
                    if (g.nHashes > 2) myPosX[64 + lane] = ibfWordIndex(g, h, 2);
                }
                __builtin_amdgcn_wave_barrier();
                u32 nextK = lastK;
                u64 c[UB];
                u32 taken = 0;
#pragma unroll
                for (u32 u = 0; u < UB; ++u)
                {
                    const u32 kk = processed + u * slots + slot;
                    u64 m0 = 0;
                    if (u * slots < span && kk < lastK && und != 0) m0 = kk < fetched ? w0c[(size_t)kk * G + wl] : words[myPos0[kk] + w];
                    c[u] = m0 & und;
                    if (need != 0xFFFFFFFFu)
                    {   // the carriers of the single bin in k-mer order = this sub-pass's lanes in lane order (one lane per k-mer carries its word)
                        const u64 car = ballot64(c[u] != 0);
                        const u32 pc = (u32)__popcll(car);
                        if (taken + (u32)__popcll(car & ((1ull << lane) - 1)) >= need) c[u] = 0;
                        if (taken < need && taken + pc >= need)
                        {   // the need-th carrier sits in this sub-pass: the round ends behind its k-mer
                            u64 t = car;
                            for (u32 i = taken + 1; i < need; ++i) t &= t - 1;
                            nextK = processed + u * slots + (((u32)__ffsll((long long)t) - 1) >> gShift) + 1;
                        }
                        taken += pc;
                    }
                }
#pragma unroll
                for (u32 u = 0; u < UB; ++u)
                    if (processed + u * slots + slot < fetched) countBitsBytes(seen, wl * 64, c[u]);
                if (need != 0xFFFFFFFFu && g.nHashes == 3)
                {
                    u64 v1[UB], v2[UB];
#pragma unroll
                    for (u32 u = 0; u < UB; ++u)
                    {
                        v1[u] = c[u] ? words[myPosX[u * slots + slot] + w] : 0;
                        v2[u] = c[u] ? words[myPosX[64 + u * slots + slot] + w] : 0;
                    }
#pragma unroll
                    for (u32 u = 0; u < UB; ++u) c[u] &= v1[u] & v2[u];
                }
                else
                    for (u32 hf = 1; hf < g.nHashes; ++hf)
                    {
                        u64 v[UB];
#pragma unroll
                        for (u32 u = 0; u < UB; ++u)
                        {
                            v[u] = 0;
                            if (c[u]) v[u] = words[(hf <= 2 ? myPosX[(hf - 1) * 64 + u * slots + slot] : ibfWordIndex(g, myHash[processed + u * slots + slot], hf)) + w];
                        }
#pragma unroll
                        for (u32 u = 0; u < UB; ++u) c[u] &= v[u];
                    }
#pragma unroll
                for (u32 u = 0; u < UB; ++u) countBitsBytes(exc, wl * 64, c[u]);
                processed = nextK;
                __builtin_amdgcn_wave_barrier();
                const u32 unlooked = nKmers - max(fetched, processed);
                for (u64 todo = und; todo; todo &= todo - 1)
                {
                    const u32 b = (u32)__ffsll((long long)todo) - 1;
                    const u32 e = ldsByteGet(exc, wl * 64 + b);
                    if (e >= threshold) { selW |= 1ull << b; und &= ~(1ull << b); }
                    else if (e + (ldsU16Get(cnt0, wl * 64 + b) - ldsByteGet(seen, wl * 64 + b)) + unlooked < threshold) und &= ~(1ull << b);
                }
            }
        }
        if (slot == 0 && wOk) out[w] = selW;
    }
}

// see the comment above k_ibf_query_lazy.  LDS per wave (kept small: it decides the waves per CU): k-mer hashes | positions of a 64-k-mer
// chunk under hash functions 0 and 1 | 16-bit counters | suspect masks | list of the surviving bits as (k-mer << 10 | word << 6 | bit).
__global__ __launch_bounds__(256) void k_ibf_query_eager2(const u64 * __restrict__ words, IbfGeom g,
                                                          const u8 * __restrict__ seqs, const u64 * __restrict__ offs, u32 nReads,
                                                          u32 threshold, u64 * __restrict__ masks, u32 gShift, u32 nKcap, u32 listCap)
{
    extern __shared__ u64 ldsEager[];
    constexpr u32 U = 8;            // sub-passes in flight (two gathers each): the kernel is a chain of dependent batches, the fewer the better
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 read = blockIdx.x * (blockDim.x >> 6) + wave;
    if (read >= nReads) return;
    const u8 * seq = seqs + offs[read];
    const u32 len = (u32)(offs[read + 1] - offs[read]);
    const u32 k = g.kmerSize;
    u64 * out = masks + (size_t)read * g.W;
    const u32 nKmers = len >= k ? len - k + 1 : 0;
    if (nKmers == 0 || threshold == 0 || threshold > nKmers)
    {
        for (u32 w = lane; w < g.W; w += 64)
        {
            const u32 left = g.nBins - w * 64;
            out[w] = threshold == 0 ? (left >= 64 ? ~0ull : ((1ull << left) - 1)) : 0ull;
        }
        return;
    }
    const u64 kMask = (1ull << k) - 1;
    const u32 G = 1u << gShift, slots = 64u >> gShift, slot = lane >> gShift, wl = lane & (G - 1);
    const u32 hashSlots = (nKcap + 31) & ~31u;
    const size_t perWave = (size_t)hashSlots + 128 + 16 * G + G + 1 + (listCap + 1) / 2;
    u64 * myHash = ldsEager + (size_t)wave * perWave;
    u64 * myPos = myHash + hashSlots;                 // [2][64]: the current chunk under hash functions 0 and 1
    u32 * cnt = (u32 *)(myPos + 128);                 // 64 G 16-bit counters: k-mers whose hash-0 AND hash-1 words both carry the bin
    u64 * susp = (u64 *)(cnt + 32 * G);               // G words
    u32 * listCount = (u32 *)(susp + G);              // entries appended so far (may run past listCap: the excess is dropped and the list void)
    u32 * list = listCount + 2;
    const u32 stage1 = g.nHashes >= 2 ? 2 : 1;

    for (u32 w0 = 0; w0 < g.W; w0 += G)
    {
        const u32 w = w0 + wl;
        const bool wOk = w < g.W;
        const u32 left = wOk ? g.nBins - w * 64 : 0;
        const u64 binMask = left >= 64 ? ~0ull : ((1ull << left) - 1);
        u64 selW = 0;
        for (u32 strand = 0; strand < 2; ++strand)
        {
            const bool firstIsN = strandCode(seq, len, 0, strand) > 3;
            __builtin_amdgcn_wave_barrier();
            for (u32 i = lane; i < 32 * G + 2 * G + 2; i += 64) cnt[i] = 0;           // cnt, susp, listCount are contiguous
            // ---- stage 1: every k-mer, hash functions 0 and 1 together; a chunk of 64 k-mers at a time gets its positions, lane = k-mer
            for (u32 c0 = 0; c0 < nKmers; c0 += 64)
            {
                __builtin_amdgcn_wave_barrier();
                {
                    const u64 h = laneKmerHash(seq, len, strand, c0, lane, k, kMask, firstIsN);
                    if (c0 + lane < nKmers)
                    {
                        myHash[c0 + lane] = h;
                        myPos[lane] = ibfWordIndex(g, h, 0);
                        if (stage1 == 2) myPos[64 + lane] = ibfWordIndex(g, h, 1);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                const u32 inChunk = min(64u, nKmers - c0);
                for (u32 kb = 0; kb < inChunk; kb += U * slots)
                {
                    u64 a[U], b1[U];
#pragma unroll
                    for (u32 u = 0; u < U; ++u)
                    {
                        const u32 kl = kb + u * slots + slot;
                        const bool valid = kl < inChunk && wOk;
                        a[u] = valid ? words[myPos[kl] + w] : 0;
                        b1[u] = (valid && stage1 == 2) ? words[myPos[64 + kl] + w] : ~0ull;
                    }
#pragma unroll
                    for (u32 u = 0; u < U; ++u)
                    {
                        const u32 kk = c0 + kb + u * slots + slot;
                        u64 v = a[u] & b1[u] & binMask;
                        while (v)
                        {
                            const u32 b = (u32)__ffsll((long long)v) - 1;
                            v &= v - 1;
                            if (ldsU16Inc(cnt, wl * 64 + b) + 1 == threshold) atomicOr((unsigned long long *)(susp + wl), 1ull << b);
                            const u32 pos = atomicAdd(listCount, 1u);
                            if (pos < listCap) list[pos] = (kk << 10) | (wl << 6) | b;
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (ballot64(susp[wl] != 0) == 0) continue;                      // no bin reaches the threshold even on two hash functions
            if (g.nHashes <= stage1)
            {   // nothing left to verify: the counts are exact
                selW |= susp[wl];
                continue;
            }
            // ---- stage 2: the suspects against the remaining hash functions, one suspect at a time (there is normally one)
            const u32 nList = *listCount;
            const bool overflow = nList > listCap;
            for (u32 ow = 0; ow < G; ++ow)
            {
                u64 todo = susp[ow];
                while (todo)
                {
                    const u32 b = (u32)__ffsll((long long)todo) - 1;
                    todo &= todo - 1;
                    const u32 carriers = ldsU16Get(cnt, ow * 64 + b);
                    const u32 want = (ow << 6) | b;
                    u32 exact = 0, seenC = 0;
                    bool decided = false, accepted = false;
                    if (!overflow)
                        for (u32 l0 = 0; l0 < nList && !decided; l0 += 64)
                        {
                            const u32 li = l0 + lane;
                            bool car = false; u32 e = 0;
                            if (li < nList) { e = list[li]; car = (e & 1023u) == want; }
                            u64 pending = ballot64(car);
                            while (pending && !decided)
                            {
                                const u32 need = threshold - exact;
                                const bool act = ((pending >> lane) & 1) && (u32)__popcll(pending & ((1ull << lane) - 1)) < need;
                                bool hit = act;
                                if (act)
                                {
                                    const u64 h = myHash[e >> 10];
                                    for (u32 hf = stage1; hf < g.nHashes; ++hf) hit = hit && ((words[ibfWordIndex(g, h, hf) + w0 + ow] >> b) & 1);
                                }
                                const u64 actMask = ballot64(act);
                                exact += (u32)__popcll(ballot64(hit));
                                seenC += (u32)__popcll(actMask);
                                pending &= ~actMask;
                                if (exact >= threshold) { decided = true; accepted = true; }
                                else if (exact + (carriers - seenC) < threshold) decided = true;
                            }
                        }
                    else
                    {   // the list overflowed (pathological: hundreds of surviving bits): plain sweep over every k-mer for this bin
                        for (u32 kb = 0; kb < nKmers && !decided; kb += 64)
                        {
                            const u32 kk = kb + lane;
                            bool hit = false;
                            if (kk < nKmers)
                            {
                                hit = true;
                                const u64 h = myHash[kk];
                                for (u32 hf = 0; hf < g.nHashes && hit; ++hf) hit = (words[ibfWordIndex(g, h, hf) + w0 + ow] >> b) & 1;
                            }
                            exact += (u32)__popcll(ballot64(hit));
                            const u32 restK = nKmers - min(nKmers, kb + 64);
                            if (exact >= threshold) { decided = true; accepted = true; }
                            else if (exact + restK < threshold) decided = true;
                        }
                    }
                    if (accepted && wl == ow) selW |= 1ull << b;
                }
            }
        }
        if (slot == 0 && wOk) out[w] = selW;
    }
}

// Set bits in an evenly spread sample of the table's words (the filter's fill decides where the sweeps place their first check)
__global__ void k_ibf_fill_sample(const u64 * __restrict__ words, u64 nTableWords, u32 nSamples, unsigned long long * __restrict__ total)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    u32 c = 0;
    if (t < nSamples) c = (u32)__popcll(words[(u64)(((unsigned __int128)t * nTableWords) / nSamples)]);
    for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(total, (unsigned long long)c);
}

// Insertion: thread per k-mer (reference src/d_bloom_filter.h:293-315). seq = Dna5 codes on the device.
__global__ void k_ibf_insert(u64 * __restrict__ words, IbfGeom g, const u8 * __restrict__ seq, u64 len, u32 bin)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 k = g.kmerSize;
    if (len < k || i >= len - k + 1) return;
    u64 hash = 0;
    for (u32 t = 0; t < k; ++t) { u32 c = seq[i + t]; hash = hash * 4 + (c > 3 ? 0 : c); }
    if (seq[0] > 3 && 2 * (k + i) < 64) hash += 1ull << (2 * (k + i));
    const u64 bitInWord = bin & 63, wordInPos = bin >> 6;
    if (g.direct && hash >= g.nHashPos) return;
    for (u32 hf = 0; hf < g.nHashes; ++hf)
        atomicOr((unsigned long long *)&words[ibfWordIndex(g, hash, hf) + wordInPos], 1ull << bitInWord);
}

// clearBins (reference src/d_bloom_filter.h:99-123, src/d_kdx_filter.h:74-99): one streaming read-modify-write pass over the table
__global__ void k_ibf_clear_bins(u64 * __restrict__ words, u64 nTableWords, u32 W, const u64 * __restrict__ clearMask)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < nTableWords; i += stride)
    {
        const u64 m = clearMask[i % W];
        if (m) words[i] &= ~m;
    }
}

} // namespace dy

// K1 — Interleaved Bloom filter k-mer sweep for gfx950 (wave64), plus the k-mer insertion kernel used to build filters.
// Replaces SeqAnBloomFilter::whichBins / _addKmers (reference src/d_bloom_filter.h:161-224, 293-315) and the read loop of
// clasifyLoadedReads (reference src/d_mapper.h:345-371).
//
// Mapping: one wavefront per read; lane = k-mer.  A chunk of 64 consecutive k-mers needs 64+k-1 <= 95 bases; the two
// bit-planes of the 2-bit codes are built with wave ballots (bit t of a plane = base t), so a lane gets its k-mer with
// two funnel shifts, a bit reversal and a Morton interleave — no LDS, no per-base loop.  Each lane then issues its h*W
// independent 64-bit gathers (the only HBM traffic that matters: (L-k+1)*h*W*8 B per strand), ANDs them, and the
// per-bin counts are popcounts of ballots over the bins present in the wave's OR-mask.
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
    u64 v = g.preCalc[i] * kmerHash;
    v ^= v >> 27;
    v = g.hashPosMask ? (v & g.hashPosMask) : (v % g.nHashPos);
    return v * g.W;
}

// strand 0: code(idx) = seq[idx]; strand 1: complement of seq[len-1-idx] (N stays N)
__device__ __forceinline__ u32 strandCode(const u8 * __restrict__ seq, u32 len, u32 idx, u32 strand)
{
    if (idx >= len) return 0;
    u32 c = strand ? seq[len - 1 - idx] : seq[idx];
    return (strand && c < 4) ? 3 - c : c;
}

// counts: LDS, one u16 per bin for this wave (64*W entries)
__global__ __launch_bounds__(256) void k_ibf_query(const u64 * __restrict__ words, IbfGeom g,
                                                   const u8 * __restrict__ seqs, const u64 * __restrict__ offs, u32 nReads,
                                                   u32 threshold, u64 * __restrict__ masks)
{
    extern __shared__ u16 ldsCounts[];
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 read = blockIdx.x * (blockDim.x >> 6) + wave;
    if (read >= nReads) return;
    u16 * counts = ldsCounts + (size_t)wave * 64 * g.W;
    const u8 * seq = seqs + offs[read];
    const u32 len = (u32)(offs[read + 1] - offs[read]);
    const u32 k = g.kmerSize;
    u64 * out = masks + (size_t)read * g.W;

    if (len < k)
    {   // no k-mer: every count is 0
        for (u32 w = lane; w < g.W; w += 64) out[w] = threshold == 0 ? ~0ull : 0ull;
        return;
    }
    const u32 nKmers = len - k + 1;
    const u64 kMask = (1ull << k) - 1;          // k <= 32 (checked at load/create)

    u64 selOwn = 0;   // selected bins: lane l owns mask word l (W <= 64)

    // Pruning that cannot change the result: once every bin is decided (count >= threshold, or count + remaining <
    // threshold) the remaining k-mers are not looked up.  The last `tail` k-mers (at most threshold-2, so that a bin
    // with a single stray hit is already decided) form their own chunk; for reads that really come from one bin this
    // skips about a third of the gathers of both strands.
    const u32 tail = threshold >= 2 ? min(min(threshold - 2, 64u), nKmers) : 0;
    const u32 head = nKmers - tail;

    for (u32 strand = 0; strand < 2; ++strand)
    {
        for (u32 b = lane; b < 64 * g.W; b += 64) counts[b] = 0;
        const bool firstIsN = strandCode(seq, len, 0, strand) > 3;     // SeqAn hashInit takes text[0] uncast
        u32 base = 0, limit = head;
        for (u32 phase = 0; phase < 2; ++phase)
        {
            if (phase == 1)
            {
                if (!tail) break;
                bool undecided = false;
                for (u32 w = 0; w < g.W; ++w)
                {
                    const u32 c = counts[w * 64 + lane];
                    undecided |= __ballot(c < threshold && c + tail >= threshold) != 0;
                }
                if (!undecided) break;
                base = head; limit = nKmers;
            }
            for (; base < limit; base += 64)
            {
                // bit planes of bases [base, base+128)
                u32 c0 = strandCode(seq, len, base + lane, strand), c1 = strandCode(seq, len, base + 64 + lane, strand);
                if (c0 > 3) c0 = 0;
                if (c1 > 3) c1 = 0;
                const u64 p0lo = ballot64(c0 & 1), p1lo = ballot64(c0 & 2), p0hi = ballot64(c1 & 1), p1hi = ballot64(c1 & 2);
                const u32 i = base + lane;
                const bool valid = i < limit;
                // bits [lane, lane+k) of the 128-bit planes
                u64 x0 = lane ? ((p0lo >> lane) | (p0hi << (64 - lane))) : p0lo;
                u64 x1 = lane ? ((p1lo >> lane) | (p1hi << (64 - lane))) : p1lo;
                x0 &= kMask; x1 &= kMask;
                // base t of the k-mer carries weight 4^(k-1-t): reverse the k bits, then interleave the planes
                u64 r0 = __brevll(x0) >> (64 - k), r1 = __brevll(x1) >> (64 - k);
                u64 hash = spreadBits(r0) | (spreadBits(r1) << 1);
                if (firstIsN && 2 * (k + i) < 64) hash += 1ull << (2 * (k + i));

                for (u32 w = 0; w < g.W; ++w)
                {
                    u64 m = 0;
                    if (valid)
                    {
                        m = ~0ull;
                        for (u32 hf = 0; hf < g.nHashes; ++hf) m &= words[ibfWordIndex(g, hash, hf) + w];
                    }
                    u64 cand = waveOr64(m);
                    while (cand)
                    {
                        u32 j = __ffsll((long long)cand) - 1;
                        cand &= cand - 1;
                        u32 cnt = __popcll(ballot64((m >> j) & 1));
                        if (lane == 0) counts[w * 64 + j] += (u16)cnt;
                    }
                }
            }
        }
        // whichBins tail: selected[bin] |= counts[bin] >= threshold   (reference src/d_bloom_filter.h:219-223)
        for (u32 w = 0; w < g.W; ++w)
        {
            u32 bin = w * 64 + lane;
            u64 sel = ballot64(bin < g.nBins && counts[bin] >= threshold);
            if (lane == (w & 63)) selOwn |= sel;     // W <= 64: word w lives in lane w
        }
    }
    if (lane < g.W) out[lane] = selOwn;
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
    for (u32 hf = 0; hf < g.nHashes; ++hf)
        atomicOr((unsigned long long *)&words[ibfWordIndex(g, hash, hf) + wordInPos], 1ull << bitInWord);
}

} // namespace dy

// Device-side FM-index primitives over the reference's own on-disk layout:
// rank over 32-byte blocks of lf.drv, LF-mapping with the sentinel correction, backward search step
// (reference src/index_fm.h:132-155) and the sparse-SA walk of SeqAn's CompressedSA (SAMPLING = 10,
// reference src/misc_types.h:174).
#pragma once
#include <hip/hip_runtime.h>
#include "dy_types.h"

namespace dy {

// Pointers into the device arena shared by all bins.
struct IndexView
{
    const BinDesc * bins;
    const uint4 *   rank;       // 2 x uint4 per block: {sym0.lo, sym0.hi, sym1.lo, sym1.hi}, {cntA, cntC, cntG, cntT}
    const uint2 *   ind;        // {bits, onesBefore}
    const u32 *     valOffset;
    const u16 *     valContig;
    const u8 *      contigs;
    const u64 *     limits;
    const u32 *     drp;
    const u32 *     saFullOffset;   // expanded SA (one entry per BWT row) or nullptr: then the sparse SA is walked
    const u16 *     saFullContig;   // nullptr when every bin has a single contig
    const uint2 *   prefixTable;    // per bin 4^q (lo,hi) ranges of every q-mer, or nullptr
    const uint4 *   prefixWide;     // the same table in 16-byte entries {lo, hi, next bases 0-15, next bases 16-31} (k_prefix_wide), or nullptr;
                                    // when it exists it REPLACES prefixTable
    const u32 *     isa;            // inverse of the expanded SA: row of the suffix at (contig, offset in the reversed contig), or nullptr
    const u8 *      ftext;          // the INDEXED text (Ns randomised), forward, laid out like `contigs`; position 0 of a contig is undefined
    u32             prefixQ;
    u32             nBins;
};

struct BinView
{
    const uint4 * rank;
    const uint2 * ind;
    const u32 *   valOffset;
    const u16 *   valContig;
    const u8 *    contigs;      // base of this bin's concatenated contigs
    const u64 *   limits;       // nContigs + 1
    const u32 *   drp;
    const u32 *   saFullOffset;
    const u16 *   saFullContig;
    u32 n, nContigs, drs;
    uint4 pst;                  // C[A..T]; read through sel4: a dynamically indexed array would put the whole view into scratch memory
};

__device__ __forceinline__ BinView makeBinView(const IndexView & ix, u32 bin)
{
    const BinDesc & d = ix.bins[bin];
    BinView b;
    b.rank = ix.rank + d.rankOff * 2;
    b.ind = ix.ind + d.indOff;
    b.valOffset = ix.valOffset + d.valOff;
    b.valContig = ix.valContig + d.valOff;
    b.contigs = ix.contigs + d.contigOff;
    b.limits = ix.limits + d.limOff;
    b.drp = ix.drp + d.drpOff;
    b.saFullOffset = ix.saFullOffset ? ix.saFullOffset + d.saFullOff : nullptr;
    b.saFullContig = ix.saFullContig ? ix.saFullContig + d.saFullOff : nullptr;
    b.n = d.n; b.nContigs = d.nContigs; b.drs = d.drs;
    b.pst = make_uint4(d.pst[0], d.pst[1], d.pst[2], d.pst[3]);
    return b;
}

__device__ __forceinline__ u32 sel4(uint4 v, u32 c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }

// number of symbols equal to c among the first `off` (0..63) symbols of a block (symbol t sits at bits (31 - t%32)*2 of word t/32)
__device__ __forceinline__ u32 countInBlock(uint4 sym, u32 c, u32 off)
{
    const u64 rep = 0x5555555555555555ull * c;
    u64 w0 = ((u64)sym.y << 32) | sym.x, w1 = ((u64)sym.w << 32) | sym.z;
    u64 x0 = ~(w0 ^ rep), x1 = ~(w1 ^ rep);
    u64 m0 = x0 & (x0 >> 1) & 0x5555555555555555ull, m1 = x1 & (x1 >> 1) & 0x5555555555555555ull;
    u32 k0 = off < 32 ? off : 32, k1 = off < 32 ? 0 : off - 32;
    if (k0 < 32) m0 = k0 ? (m0 & (~0ull << (64 - 2 * k0))) : 0;
    m1 = k1 ? (m1 & (~0ull << (64 - 2 * k1))) : 0;
    return __popcll(m0) + __popcll(m1);
}

// #sentinel rows < pos
__device__ __forceinline__ u32 sentinelsBefore(const BinView & b, u32 pos)
{
    if (b.nContigs == 1) return b.drp[0] < pos;
    u32 lo = 0, hi = b.nContigs;
    while (lo < hi) { u32 mid = (lo + hi) >> 1; if (b.drp[mid] < pos) lo = mid + 1; else hi = mid; }
    return lo;
}

// LF(pos, c) = C[c] + rank_c(bwt[0..pos))
__device__ __forceinline__ u32 lfMap(const BinView & b, u32 pos, u32 c)
{
    const uint4 * blk = b.rank + (size_t)(pos >> 6) * 2;
    uint4 sym = blk[0], cnt = blk[1];
    u32 r = sel4(cnt, c) + countInBlock(sym, c, pos & 63);
    if (c == b.drs) r -= sentinelsBefore(b, pos);
    return sel4(b.pst, c) + r;
}

__device__ __forceinline__ u32 bwtSymbol(const BinView & b, u32 row)
{
    uint4 sym = b.rank[(size_t)(row >> 6) * 2];
    u32 t = row & 63;
    u32 word = t < 32 ? (((t & 31) < 16) ? sym.y : sym.x) : (((t & 31) < 16) ? sym.w : sym.z);
    // symbol t%32 sits at bits (31 - t%32)*2 of the 64-bit word: t%32 < 16 -> high dword
    u32 sh = ((31 - (t & 31)) * 2) & 31;
    return (word >> sh) & 3;
}

// one backward-search step on both range ends; false when the range becomes empty.
// Once the range is narrow both ends fall into the same 64-symbol block: it is fetched once.
__device__ __forceinline__ bool fmStep(const BinView & b, u32 & lo, u32 & hi, u32 c)
{
    const uint4 * blkLo = b.rank + (size_t)(lo >> 6) * 2;
    const uint4 symLo = blkLo[0], cntLo = blkLo[1];
    uint4 symHi = symLo, cntHi = cntLo;
    if ((hi >> 6) != (lo >> 6))
    {
        const uint4 * blkHi = b.rank + (size_t)(hi >> 6) * 2;
        symHi = blkHi[0]; cntHi = blkHi[1];
    }
    u32 rl = sel4(cntLo, c) + countInBlock(symLo, c, lo & 63);
    u32 rh = sel4(cntHi, c) + countInBlock(symHi, c, hi & 63);
    if (c == b.drs) { rl -= sentinelsBefore(b, lo); rh -= sentinelsBefore(b, hi); }
    lo = sel4(b.pst, c) + rl;
    hi = sel4(b.pst, c) + rh;
    return lo < hi;
}

// XCD-aware block remap: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 share an L2), so give every XCD
// one contiguous eighth of the (bin-major) work list: its 4 MiB L2 then holds the rank blocks of the bin it is walking.
__device__ __forceinline__ u32 xcdRemapBlock(u32 blockId, u32 nBlocks)
{
    const u32 per = (nBlocks + 7) / 8;
    return (blockId % 8) * per + blockId / 8;
}

// SA[row] -> (contig, offset in the REVERSED contig) by walking LF to the next sampled row (SeqAn CompressedSA)
__device__ __forceinline__ void fmLocateSparse(const BinView & b, u32 row, u32 & contig, u32 & offset)
{
    u32 steps = 0;
    for (;;)
    {
        uint2 e = b.ind[row >> 5];
        u32 bit = 31 - (row & 31);
        if ((e.x >> bit) & 1)
        {
            u32 idx = e.y + (bit == 31 ? 0u : (u32)__popc(e.x >> (bit + 1)));
            contig = b.valContig[idx];
            offset = b.valOffset[idx] + steps;
            return;
        }
        row = lfMap(b, row, bwtSymbol(b, row));
        ++steps;
    }
}

// With 288 GB of HBM the sparse SA (one sample per 10 text positions on disk) is expanded once at load time to one
// entry per row, so a locate on the mapping path is a single load instead of up to 9 dependent LF steps.
__device__ __forceinline__ void fmLocate(const BinView & b, u32 row, u32 & contig, u32 & offset)
{
    if (b.saFullOffset)
    {
        offset = b.saFullOffset[row];
        contig = b.saFullContig ? b.saFullContig[row] : 0;
        return;
    }
    fmLocateSparse(b, row, contig, offset);
}

// Range table of all q-mers of one bin, built level by level at load time: entry(p·4 + c) = step(entry(p), c).
// On the mapping path one gather replaces the first q backward-search steps of a seed (the ones whose two range ends
// still sit in different rank blocks).
__global__ void k_prefix_level(IndexView ix, u32 bin, const uint2 * __restrict__ parent, uint2 * __restrict__ child, u32 nChild)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nChild) return;
    const BinView b = makeBinView(ix, bin);
    uint2 r = parent ? parent[i >> 2] : make_uint2(0u, b.n);
    if (r.x < r.y)
    {
        u32 lo = r.x, hi = r.y;
        r = fmStep(b, lo, hi, i & 3) ? make_uint2(lo, hi) : make_uint2(0u, 0u);
    }
    else r = make_uint2(0u, 0u);
    child[i] = r;
}

// Device-only companions of the expanded SA, filled once at load time: the inverse SA and the indexed text itself (recovered
// from the BWT: the symbol in front of the suffix at reversed offset x is forward base clen - x).  With them a seed whose
// q-mer range is narrow is finished by comparing the rest of the seed with the text instead of walking LF (k_seed_search).
__global__ void k_build_isa_ftext(IndexView ix, u32 bin, u32 * __restrict__ isa, u8 * __restrict__ ftext)
{
    const BinView b = makeBinView(ix, bin);
    const u32 row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= b.n || row < b.nContigs) return;
    u32 contig, x;
    fmLocate(b, row, contig, x);
    const u64 lim = b.limits[contig];
    const u32 clen = (u32)(b.limits[contig + 1] - lim);
    if (x >= clen) return;
    isa[ix.bins[bin].saFullOff + lim + x] = row;
    if (x >= 1) ftext[ix.bins[bin].contigOff + lim + (clen - x)] = (u8)bwtSymbol(b, row);
}

// "Direct" encodings (only used when the inverse SA / indexed text exist).  Rows stay below 0xFFFE0000 (bins above ~4.29 Gbp
// are refused at load), so a second word at or above a tag cannot be a range end:
//   q-mer table entry (x, DIRECT_QMER | contig): the q-mer occurs once; x = offset of its suffix in the reversed contig
//   seed hit          (x, DIRECT_HIT  | contig): one occurrence of the whole seed, located; x as for fmLocate
enum : u32 { DIRECT_QMER = 0xFFFE0000u, DIRECT_HIT = 0xFFFF0000u };
__device__ __forceinline__ u32 hitCount(uint2 r) { return r.y >= DIRECT_HIT ? 1u : r.y - r.x; }

// after the expanded SA exists: unique q-mers carry their position instead of their row (saves the locate on the mapping path)
__global__ void k_prefix_direct(IndexView ix, u32 bin, uint2 * __restrict__ table, u32 nEntries)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nEntries) return;
    const uint2 r = table[i];
    if (r.y - r.x != 1 || r.y >= DIRECT_QMER) return;
    const BinView b = makeBinView(ix, bin);
    u32 contig, x;
    fmLocate(b, r.x, contig, x);
    table[i] = make_uint2(x, DIRECT_QMER | contig);
}

// 8 bytes of 2-bit codes -> 16 bits, byte j in bits 2j..2j+1 (the caller has checked that every byte is < 4, or masks the result)
__device__ __forceinline__ u32 pack2bit8(u64 w)
{
    w &= 0x0303030303030303ull;
    w = (w | (w >> 6)) & 0x000F000F000F000Full;
    w = (w | (w >> 12)) & 0x000000FF000000FFull;
    return (u32)((w | (w >> 24)) & 0xFFFFu);
}

// The q-mer table in 16-byte entries: a unique q-mer also carries the (up to) 32 bases that FOLLOW its one occurrence in the indexed text,
// two bits each, so K2 verifies the rest of a seed from the table's own cache line instead of a second miss in the text.
// Bases past the contig's end stay 0: a seed that needs them starts before the contig does and K2 rejects it on x < rem first.
__global__ void k_prefix_wide(IndexView ix, u32 bin, const uint2 * __restrict__ narrow, uint4 * __restrict__ wide, u32 nEntries)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nEntries) return;
    const uint2 r = narrow[i];
    if (r.y < DIRECT_QMER) { wide[i] = make_uint4(r.x, r.y, 0u, 0u); return; }
    const u32 contig = r.y & 0xFFFFu, x = r.x;
    const u64 * limits = ix.limits + ix.bins[bin].limOff;
    const u64 lim = limits[contig];
    const u32 clen = (u32)(limits[contig + 1] - lim);
    const u8 * text = ix.ftext + ix.bins[bin].contigOff + lim + (clen - x);
    u64 next = 0;
    const u32 have = min(x, 32u);
    for (u32 j = 0; j < have; ++j) next |= (u64)(text[j] & 3u) << (2 * j);
    wide[i] = make_uint4(r.x, r.y, (u32)next, (u32)(next >> 32));
}

__global__ void k_expand_sa(IndexView ix, u32 bin, u32 * __restrict__ outOffset, u16 * __restrict__ outContig)
{
    BinView b = makeBinView(ix, bin);
    b.saFullOffset = nullptr;
    const u32 row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= b.n) return;
    u32 contig = 0, offset = 0;
    // the first nContigs rows are the suffixes "$c": they never occur in a hit range and carry no sample
    if (row >= b.nContigs) fmLocateSparse(b, row, contig, offset);
    const u64 base = ix.bins[bin].saFullOff;
    outOffset[base + row] = offset;
    if (outContig) outContig[base + row] = (u16)contig;
}

} // namespace dy

// Bit-parallel prefix edit distance for K3's extensions (reference src/find_extender.h:159-168,233-248: SeqAn's
// Myers<AlignTextBanded<FindPrefix, NMatchesNone_, NMatchesNone_>>), two forms with the same reports.
// Pure device functions over u32 / u64 (tests/myers_fuzz.cpp compiles them for the host and checks both against a plain DP).
#pragma once
#include "dy_types.h"

namespace dy {

// ---------------------------------------------------------------------------------------------------
// Bit-parallel prefix edit distance (Myers/Hyyro block algorithm, global start): stands in for SeqAn's
// Myers<AlignTextBanded<FindPrefix, NMatchesNone_, NMatchesNone_>> as used by find_extender.h:159-168,233-248.
// Reports, in increasing text end j, every prefix text[0..j) with ed(needle, prefix) <= k (j = 0 too when m <= k).
// N (code 4) matches nothing.  NW = 64-bit words covering the needle.
// ---------------------------------------------------------------------------------------------------
template <int NW, typename TNeedle, typename TText, typename TReport>
__device__ __forceinline__ void myersPrefixSearch(TNeedle needle, int m, TText text, int t, int k, TReport report)
{
    if (m <= k) report(-1, m);
    if (m == 0)
    {   // D[0][j] = j
        for (int j = 1; j <= t && j <= k; ++j) report(j - 1, j);
        return;
    }
    u64 peq[4][NW], vp[NW], vn[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) { peq[0][w] = peq[1][w] = peq[2][w] = peq[3][w] = 0; vp[w] = ~0ull; vn[w] = 0; }
    for (int i = 0; i < m; ++i)
    {
        u32 c = needle(i);
        if (c < 4)
        {
#pragma unroll
            for (int w = 0; w < NW; ++w)
                if ((i >> 6) == w) { u64 bit = 1ull << (i & 63); if (c == 0) peq[0][w] |= bit; else if (c == 1) peq[1][w] |= bit; else if (c == 2) peq[2][w] |= bit; else peq[3][w] |= bit; }
        }
    }
    const int lastWord = (m - 1) >> 6;
    const u64 lastBit = 1ull << ((m - 1) & 63);
    int score = m;
    const int jmax = t < m + k ? t : m + k;
    for (int j = 1; j <= jmax; ++j)
    {
        u32 c = text(j - 1);
        int hin = 1;                      // global start: D[0][j] = D[0][j-1] + 1
#pragma unroll
        for (int w = 0; w < NW; ++w)
        {
            if (w <= lastWord)
            {
                u64 eq = c == 0 ? peq[0][w] : c == 1 ? peq[1][w] : c == 2 ? peq[2][w] : c == 3 ? peq[3][w] : 0ull;
                const u64 pv = vp[w], mv = vn[w];
                const u64 hinNeg = hin < 0 ? 1ull : 0ull;
                const u64 xv = eq | mv;
                eq |= hinNeg;
                const u64 xh = (((eq & pv) + pv) ^ pv) | eq;
                u64 ph = mv | ~(xh | pv);
                u64 mh = pv & xh;
                const u64 top = (w == lastWord) ? lastBit : (1ull << 63);
                int hout = 0;
                if (ph & top) hout = 1;
                if (mh & top) hout = -1;
                ph <<= 1; mh <<= 1;
                mh |= hinNeg;
                if (hin > 0) ph |= 1ull;
                vp[w] = mh | ~(xv | ph);
                vn[w] = ph & xv;
                hin = hout;
            }
        }
        score += hin;
        if (score <= k) report(j - 1, score);
    }
}

// The same reports from a BANDED computation in one machine word TW — u32 for k <= 15, u64 for k <= 31 — (Hyyro's diagonal band): only the cells with |row - column| <= k
// can hold a distance <= k, so the bit-vectors cover a window of w = 2k+1 rows that slides down one row per column (after the first
// k+1 columns, while the band still touches row 1).  What lies outside the window is replaced by the bounds that hold for every edit
// distance matrix — the row entering at the bottom is one more than the row above it, the cell above the window is one more than its
// left neighbour — so the computed matrix is >= the true one everywhere and equal wherever the true value is <= k (an optimal path to such
// a cell never leaves the band).  The absolute score is carried on the window's bottom row; D[m][j] follows from it through the vertical
// deltas between row m and the bottom.  A column costs ~30 instructions whatever the needle's length (the block form: ~25 per 64 rows,
// plus building the needle's match masks first).
__device__ __forceinline__ int popcWord(u32 x) { return (int)__popc(x); }
__device__ __forceinline__ int popcWord(u64 x) { return (int)__popcll(x); }

template <typename TW, typename TNeedle, typename TText, typename TReport>
__device__ __forceinline__ void myersPrefixSearchBanded(TNeedle needle, int m, TText text, int t, int k, TReport report)
{
    if (m <= k) report(-1, m);
    if (m == 0)
    {   // D[0][j] = j
        for (int j = 1; j <= t && j <= k; ++j) report(j - 1, j);
        return;
    }
    const int w = 2 * k + 1;                                   // <= 31 (u32) / 63 (u64)
    const TW one = 1, mask = (one << w) - one, topBit = one << (w - 1);
    // match masks of the window's rows per character: bit b <-> row lo + b (needle index lo + b - 1); lo = 1 for the first k+1 columns
    TW B0 = 0, B1 = 0, B2 = 0, B3 = 0;
    for (int b = 0; b < w && b < m; ++b)
    {
        const u32 c = needle(b); const TW bit = one << b;
        if (c == 0) B0 |= bit; else if (c == 1) B1 |= bit; else if (c == 2) B2 |= bit; else if (c == 3) B3 |= bit;
    }
    TW pv = mask, mv = 0;
    int scoreBottom = w;                                       // Dc[lo + w - 1][j], column 0: D[i][0] = i
    int lo = 1;
    const int jmax = t < m + k ? t : m + k;
    for (int j = 1; j <= jmax; ++j)
    {
        int slide = 0;
        if (j > k + 1)
        {   // the window moves down one row: its new bottom row is one more than the row above it in column j-1
            ++lo; slide = 1;
            pv = (pv >> 1) | topBit; mv >>= 1;
            B0 >>= 1; B1 >>= 1; B2 >>= 1; B3 >>= 1;
            const int ni = lo + w - 2;                         // needle index of the new bottom row
            if (ni < m)
            {
                const u32 c = needle(ni);
                if (c == 0) B0 |= topBit; else if (c == 1) B1 |= topBit; else if (c == 2) B2 |= topBit; else if (c == 3) B3 |= topBit;
            }
        }
        const u32 c = text(j - 1);
        const TW eq = c == 0 ? B0 : c == 1 ? B1 : c == 2 ? B2 : c == 3 ? B3 : (TW)0;
        const TW xv = eq | mv;
        const TW xh = ((((eq & pv) + pv) ^ pv) | eq) & mask;
        TW ph = (mv | ~(xh | pv)) & mask;
        TW mh = pv & xh;
        scoreBottom += slide + (int)((ph >> (w - 1)) & one) - (int)((mh >> (w - 1)) & one);
        ph = ((ph << 1) | one) & mask;                          // the cell above the window's top: one more than its left neighbour
        mh = (mh << 1) & mask;
        pv = (mh | ~(xv | ph)) & mask;
        mv = ph & xv;
        const int bm = m - lo;                                 // bit of row m in this column's window
        if (bm >= 0 && bm < w)
        {
            const TW above = bm + 1 < w ? (mask >> (bm + 1)) << (bm + 1) : (TW)0;     // rows m+1 .. bottom
            const int score = scoreBottom - popcWord(pv & above) + popcWord(mv & above);
            if (score <= k) report(j - 1, score);
        }
    }
}

// What K3 calls: the banded form wherever one word holds the band.  NW = 64-bit words that cover the longest needle of the batch: reads of
// up to 256 bases (NW <= 4) carry at most 25 errors (the error rate is capped at 10 %), so their kernels never need the block form.
// CONTRACT of the small instantiations (NW <= 4): k <= PREFIX_SEARCH_SMALL_MAX_K.  They leave the block form out (its registers would cost
// K3 a wave per SIMD), so the host must not launch them with a larger error budget: runSearchExtend (dyhip.hip) dispatches on this very
// constant (`bandFits`), and tests/native/myers_fuzz.cpp checks the dispatcher against a plain DP up to k = 40 for NW = 16 and up to the
// bound for NW = 2 / 4.  The clamp below only keeps a violated contract from shifting by more than the word: it would lose matches.
enum { PREFIX_SEARCH_SMALL_MAX_K = 31 };
template <int NW, typename TNeedle, typename TText, typename TReport>
__device__ __forceinline__ void prefixSearch(TNeedle needle, int m, TText text, int t, int k, TReport report)
{
    if (k <= 15) myersPrefixSearchBanded<u32>(needle, m, text, t, k, report);
    else if (NW <= 4 || k <= 31) myersPrefixSearchBanded<u64>(needle, m, text, t, k < PREFIX_SEARCH_SMALL_MAX_K ? k : (int)PREFIX_SEARCH_SMALL_MAX_K, report);
    else myersPrefixSearch<NW>(needle, m, text, t, k, report);
}

} // namespace dy

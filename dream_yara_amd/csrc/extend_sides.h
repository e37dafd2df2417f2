// The two sides of a seed extension (reference src/find_extender.h:262-403): the byte-string compare primitives, the edit distance searches
// as K3 runs them, and their closed forms for a side with a single substitution.  No HIP dependency beyond the function qualifiers, so
// tests/native/extend_fuzz.cpp compiles this file for the host and checks the closed forms against the searches.
#pragma once
#include "dy_types.h"
#include "myers_device.h"

namespace dy {

// ---- 8-bytes-at-a-time comparison of Dna5 code strings (N = 4 never matches, reference src/basic_alphabet.h:47) ----
struct __attribute__((packed)) PackedU64 { u64 v; };
__device__ __forceinline__ u64 load8(const u8 * p) { return ((const PackedU64 *)p)->v; }
// bit 8*i set iff byte i differs or holds an N on either side
__device__ __forceinline__ u64 mismatchFlags(u64 a, u64 b)
{
    u64 x = (a ^ b) | ((a | b) & 0x0404040404040404ull);
    x |= (x >> 1) | (x >> 2);
    return x & 0x0101010101010101ull;
}
// length of the common prefix of a[0..n) and b[0..n).  32 bytes per step with all eight loads issued before the first compare:
// the loop is a chain of dependent memory round trips (the early exit keeps the compiler from hoisting loads), and K3 is bound
// by exactly that latency.
__device__ __forceinline__ int lcpForward(const u8 * __restrict__ a, const u8 * __restrict__ b, int n)
{
    int i = 0;
    for (; i + 32 <= n; i += 32)
    {
        const u64 a0 = load8(a + i), a1 = load8(a + i + 8), a2 = load8(a + i + 16), a3 = load8(a + i + 24);
        const u64 b0 = load8(b + i), b1 = load8(b + i + 8), b2 = load8(b + i + 16), b3 = load8(b + i + 24);
        const u64 f0 = mismatchFlags(a0, b0), f1 = mismatchFlags(a1, b1), f2 = mismatchFlags(a2, b2), f3 = mismatchFlags(a3, b3);
        if (f0) return i + (__ffsll((long long)f0) - 1) / 8;
        if (f1) return i + 8 + (__ffsll((long long)f1) - 1) / 8;
        if (f2) return i + 16 + (__ffsll((long long)f2) - 1) / 8;
        if (f3) return i + 24 + (__ffsll((long long)f3) - 1) / 8;
    }
    for (; i + 8 <= n; i += 8)
    {
        const u64 f = mismatchFlags(load8(a + i), load8(b + i));
        if (f) return i + (__ffsll((long long)f) - 1) / 8;
    }
    if (i < n && n >= 8)
    {   // the last 1..7 bytes as one 8-byte chunk ending at n (its first bytes are known to be equal) instead of a round trip per byte
        const u64 f = mismatchFlags(load8(a + n - 8), load8(b + n - 8));
        return f ? n - 8 + (__ffsll((long long)f) - 1) / 8 : n;
    }
    for (; i < n; ++i) { const u32 x = a[i], y = b[i]; if (!(x == y && x < 4)) break; }
    return i;
}
// length of the common suffix of a[-n..0) and b[-n..0) (a, b point one past the last byte)
__device__ __forceinline__ int lcpBackward(const u8 * __restrict__ a, const u8 * __restrict__ b, int n)
{
    int i = 0;
    for (; i + 32 <= n; i += 32)
    {
        const u64 a0 = load8(a - i - 8), a1 = load8(a - i - 16), a2 = load8(a - i - 24), a3 = load8(a - i - 32);
        const u64 b0 = load8(b - i - 8), b1 = load8(b - i - 16), b2 = load8(b - i - 24), b3 = load8(b - i - 32);
        const u64 f0 = mismatchFlags(a0, b0), f1 = mismatchFlags(a1, b1), f2 = mismatchFlags(a2, b2), f3 = mismatchFlags(a3, b3);
        if (f0) return i + __clzll((long long)f0) / 8;
        if (f1) return i + 8 + __clzll((long long)f1) / 8;
        if (f2) return i + 16 + __clzll((long long)f2) / 8;
        if (f3) return i + 24 + __clzll((long long)f3) / 8;
    }
    for (; i + 8 <= n; i += 8)
    {
        const u64 f = mismatchFlags(load8(a - i - 8), load8(b - i - 8));
        if (f) return i + __clzll((long long)f) / 8;
    }
    if (i < n && n >= 8)
    {
        const u64 f = mismatchFlags(load8(a - n), load8(b - n));
        return f ? n - 8 + __clzll((long long)f) / 8 : n;
    }
    for (; i < n; ++i) { const u32 x = a[-1 - i], y = b[-1 - i]; if (!(x == y && x < 4)) break; }
    return i;
}
// ---- left side: needle'[i] = seq[nb - i] (n elements), text'[j] = ctg[hb - j] (t elements); both run towards lower addresses ----
// _extendLeft's search (reference src/find_extender.h:330-362): the LAST end position with the smallest distance <= k.
// Returns that distance (k + 1: none) and sets end = text characters consumed.
template <int NW>
__device__ __forceinline__ u32 searchLeftSide(const u8 * __restrict__ ctg, u32 hb, const u8 * __restrict__ seq, u32 nb, int n, int t, u32 k, int & end)
{
    u32 minErr = k + 1;
    int endPos = 0;
    // the Myers loops ask for elements 0, 1, 2, ... with a wave-uniform index: one 8-byte load serves eight of them (a
    // byte load per column costs the texture addresser a full divergent access each time); elements whose 8-byte group
    // would start before the array are fetched singly
    u64 nw = 0, tw = 0;
    auto needleAt = [&](int i) -> u32 {
        if ((i & 7) == 0) { if (nb >= (u32)i + 7) nw = load8(seq + nb - i - 7); else { nw = 0; for (u32 x = 0; x < 8 && (u32)i + x <= nb; ++x) nw |= (u64)seq[nb - i - x] << (8 * (7 - x)); } }
        return (u32)(nw >> (8 * (7 - (i & 7)))) & 0xff; };
    auto textAt = [&](int j) -> u32 {
        if ((j & 7) == 0) { if (hb >= (u32)j + 7) tw = load8(ctg + hb - j - 7); else { tw = 0; for (u32 x = 0; x < 8 && (u32)j + x <= hb; ++x) tw |= (u64)ctg[hb - j - x] << (8 * (7 - x)); } }
        return (u32)(tw >> (8 * (7 - (j & 7)))) & 0xff; };
    prefixSearch<NW>(needleAt, n, textAt, t, (int)k, [&](int pos, int dist) { if ((u32)dist <= minErr) { minErr = (u32)dist; endPos = pos + 1; } });
    end = endPos;
    return minErr;
}
// The same answer without the search for a side whose only difference is ONE substitution at needle'[0] / text'[0] (the pair that
// ended the caller's LCP; k >= 1):
//   * that pair cannot be part of an exact alignment, so no end position has distance 0, and the diagonal end (n characters) has 1;
//   * one position further out has distance 1 only if the needle equals the text shifted by one (the text's first character inserted) —
//     one more LCP; positions beyond cost at least 2, and earlier ones lose the tie.
// false: not such a side (the caller runs the search).
__device__ __forceinline__ bool closeLeftSide(const u8 * __restrict__ ctg, u32 hb, const u8 * __restrict__ seq, u32 nb, int n, int t, int & end)
{
    if (n < 1 || t < n) return false;                                            // the window ends before the needle does
    if (lcpBackward(ctg + hb, seq + nb, n - 1) != n - 1) return false;           // needle'[1..n) == text'[1..n)
    end = n;
    if (t > n && lcpBackward(ctg + hb, seq + nb + 1, n) == n) end = n + 1;       // needle' == text'[1..n+1)
    return true;
}

// ---- right side: needle[i] = seq[nb + i] (nl elements, the last one held out), text[j] = ctg[hb + j] (hl elements) ----
// _extendRight's search (reference src/find_extender.h:369-395): the needle without its last base is searched in the text without its
// last character, and the held-out base is compared at every reported end.  Returns the cost (k + 1: none), end = characters consumed
// before the held-out one.
template <int NW>
__device__ __forceinline__ u32 searchRightSide(const u8 * __restrict__ ctg, u32 hb, const u8 * __restrict__ seq, u32 nb, int nl, int hl, u32 k, int & end)
{
    u32 minErr = k + 1;
    int endPos = 0;
    const u32 lastNeedle = seq[nb + nl - 1];
    u64 nw = 0, tw = 0;        // forward: an 8-byte load may run past the needle / window, never past the padded arrays
    auto needleAt = [&](int i) -> u32 { if ((i & 7) == 0) nw = load8(seq + nb + i); return (u32)(nw >> (8 * (i & 7))) & 0xff; };
    auto textAt = [&](int j) -> u32 { if ((j & 7) == 0) tw = load8(ctg + hb + j); return (u32)(tw >> (8 * (j & 7))) & 0xff; };
    prefixSearch<NW>(needleAt, nl - 1, textAt, hl - 1, (int)k,
                     [&](int pos, int dist) {
                         const int currentEnd = pos + 1;
                         u32 cur = (u32)dist;
                         const u32 h = ctg[hb + currentEnd];
                         if (!(h == lastNeedle && h < 4))
                             if (++cur > k) return;
                         if (cur <= minErr) { minErr = cur; endPos = currentEnd; }
                     });
    end = endPos;
    return minErr;
}
// Closed form, as on the left: one substitution at needle[0] / text[0], everything after it equal INCLUDING the held-out base on the
// diagonal (if that base differs the side has two errors: not closed here).  The end one further out costs 1 only if the searched needle
// equals the text shifted by one and the held-out base matches there too.
__device__ __forceinline__ bool closeRightSide(const u8 * __restrict__ ctg, u32 hb, const u8 * __restrict__ seq, u32 nb, int nl, int hl, int & end)
{
    const int n1 = nl - 1;                                                       // the searched needle
    if (n1 < 1 || hl < nl) return false;
    if (lcpForward(ctg + hb + 1, seq + nb + 1, n1 - 1) != n1 - 1) return false;
    const u32 lastNeedle = seq[nb + n1];
    const u32 h = ctg[hb + n1];
    if (!(h == lastNeedle && h < 4)) return false;
    end = n1;
    if (hl > nl && lcpForward(ctg + hb + 1, seq + nb, n1) == n1)
    {
        const u32 h1 = ctg[hb + n1 + 1];
        if (h1 == lastNeedle && h1 < 4) end = n1 + 1;
    }
    return true;
}

} // namespace dy

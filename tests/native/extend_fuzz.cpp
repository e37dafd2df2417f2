// Host fuzz of the closed forms of K3's extension sides (dream_yara_amd/csrc/extend_sides.h: closeLeftSide / closeRightSide) against the
// edit distance searches they stand in for (searchLeftSide / searchRightSide, the code K3's second pass runs), on sides that start with a
// mismatch: random, low-complexity, N-bearing, truncated windows, extra edits.  Wherever a closed form says "yes" its (cost 1, end) must be
// the search's answer.  Test infrastructure: compiled and run by tests/test_myers_fuzz.py.
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#define __device__
#define __forceinline__ inline
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
#include "../../dream_yara_amd/csrc/extend_sides.h"

using namespace dy;

static bool same(u8 a, u8 b) { return a == b && a < 4; }

int main(int argc, char ** argv)
{
    const long rounds = argc > 1 ? atol(argv[1]) : 400000;
    std::mt19937_64 rng(argc > 2 ? strtoull(argv[2], nullptr, 10) : 777ull);
    long closedLeft = 0, closedRight = 0, shiftedLeft = 0, shiftedRight = 0, refused = 0;
    const int PAD = 96;
    for (long it = 0; it < rounds; ++it)
    {
        const int alpha = it % 4 == 0 ? 2 : (it % 9 == 0 ? 1 : 4);          // low complexity: the shifted end also costs 1
        const u32 k = 1 + (u32)(rng() % 5);
        const int n = 1 + (int)(rng() % (it % 5 == 0 ? 150 : 40));           // needle of the side (right: including the held-out base)
        int t = n + (int)k;                                                  // the window as the extender sizes it ...
        if (it % 6 == 0) t = (int)(rng() % (n + k + 1));                     // ... or clipped by the contig's end
        if (t < 1) t = 1;
        std::vector<u8> needle(n), text(t);
        for (auto & c : needle) c = (u8)(rng() % alpha);
        for (int j = 0; j < t; ++j) text[j] = j < n ? needle[j] : (u8)(rng() % alpha);
        if (it % 17 == 0) needle[rng() % n] = 4;                             // an N matches nothing
        if (it % 19 == 0) text[rng() % t] = 4;
        // further edits of the text beyond the first pair, sometimes
        const int extra = it % 3 == 0 ? (int)(rng() % 3) : 0;
        for (int e = 0; e < extra && t > 2; ++e)
        {
            const size_t p = 1 + rng() % (t - 1);
            switch (rng() % 3)
            {
                case 0: text[p] = (u8)(rng() % 4); break;
                case 1: text.erase(text.begin() + p); text.push_back((u8)(rng() % alpha)); break;
                default: text.insert(text.begin() + p, (u8)(rng() % alpha)); text.pop_back(); break;
            }
        }
        // the pair that ended the caller's LCP
        if (same(needle[0], text[0])) text[0] = (u8)((needle[0] + 1 + rng() % 3) % 4);
        if (same(needle[0], text[0])) continue;                              // (needle[0] was an N and so on)

        {   // ---- right side: forward layout ----
            std::vector<u8> seq(PAD + n + PAD), ctg(PAD + t + PAD);
            for (auto & c : seq) c = (u8)(rng() % 5);
            for (auto & c : ctg) c = (u8)(rng() % 5);
            std::copy(needle.begin(), needle.end(), seq.begin() + PAD);
            std::copy(text.begin(), text.end(), ctg.begin() + PAD);
            int endClosed = -1, endSearch = -1;
            const bool closed = closeRightSide(ctg.data(), PAD, seq.data(), PAD, n, t, endClosed);
            const u32 cost = searchRightSide<2>(ctg.data(), PAD, seq.data(), PAD, n, t, k, endSearch);
            if (closed)
            {
                ++closedRight; shiftedRight += endClosed == n;
                if (cost != 1 || endClosed != endSearch)
                {
                    printf("RIGHT mismatch at round %ld: n=%d t=%d k=%u closed end=%d, search cost=%u end=%d\n", it, n, t, k, endClosed, cost, endSearch);
                    return 1;
                }
            }
            else ++refused;
        }
        {   // ---- left side: both strings run towards lower addresses ----
            std::vector<u8> seq(PAD + n + PAD), ctg(PAD + t + PAD);
            for (auto & c : seq) c = (u8)(rng() % 5);
            for (auto & c : ctg) c = (u8)(rng() % 5);
            const u32 nb = PAD + n - 1, hb = PAD + t - 1;
            for (int i = 0; i < n; ++i) seq[nb - i] = needle[i];
            for (int j = 0; j < t; ++j) ctg[hb - j] = text[j];
            int endClosed = -1, endSearch = -1;
            const bool closed = closeLeftSide(ctg.data(), hb, seq.data(), nb, n, t, endClosed);
            const u32 cost = searchLeftSide<2>(ctg.data(), hb, seq.data(), nb, n, t, k, endSearch);
            if (closed)
            {
                ++closedLeft; shiftedLeft += endClosed == n + 1;
                if (cost != 1 || endClosed != endSearch)
                {
                    printf("LEFT mismatch at round %ld: n=%d t=%d k=%u closed end=%d, search cost=%u end=%d\n", it, n, t, k, endClosed, cost, endSearch);
                    return 1;
                }
            }
            else ++refused;
        }
    }
    printf("ok: %ld left and %ld right sides closed (%ld / %ld at the shifted end), %ld left to the search\n", closedLeft, closedRight, shiftedLeft, shiftedRight, refused);
    return (closedLeft > rounds / 10 && closedRight > rounds / 10 && shiftedLeft > 100 && shiftedRight > 100) ? 0 : 2;
}

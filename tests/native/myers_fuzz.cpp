// Host fuzz of the two prefix edit distance searches of K3 (dream_yara_amd/csrc/myers_device.h) against a plain DP.
// Test infrastructure: compiled and run by tests/test_myers_fuzz.py.
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
#include "../../dream_yara_amd/csrc/myers_device.h"

using namespace dy;
typedef std::vector<std::pair<int, int> > Reports;

static Reports plainDp(const std::vector<u8> & needle, const std::vector<u8> & text, int k)
{
    const int m = (int)needle.size(), t = (int)text.size();
    Reports out;
    if (m <= k) out.push_back({-1, m});
    if (m == 0) { for (int j = 1; j <= t && j <= k; ++j) out.push_back({j - 1, j}); return out; }
    const int jmax = std::min(t, m + k);
    std::vector<int> prev(m + 1), cur(m + 1);
    for (int i = 0; i <= m; ++i) prev[i] = i;
    for (int j = 1; j <= jmax; ++j)
    {
        cur[0] = j;
        for (int i = 1; i <= m; ++i)
        {
            const bool eq = needle[i - 1] == text[j - 1] && needle[i - 1] < 4;
            cur[i] = std::min(std::min(prev[i] + 1, cur[i - 1] + 1), prev[i - 1] + (eq ? 0 : 1));
        }
        if (cur[m] <= k) out.push_back({j - 1, cur[m]});
        std::swap(prev, cur);
    }
    return out;
}

int main(int argc, char ** argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 200000;
    std::mt19937_64 rng(12345);
    long checkedBanded = 0, checkedBlock = 0, withReports = 0;
    for (int it = 0; it < rounds; ++it)
    {
        const int k = (int)(rng() % 32);
        int m = (int)(rng() % (it % 7 == 0 ? 200 : 40));
        if (it % 11 == 0) m = (int)(rng() % (2 * k + 3));                 // needles shorter than the band
        const int alpha = it % 5 == 0 ? 2 : (it % 13 == 0 ? 1 : 4);       // low-complexity texts make many co-optimal ends
        std::vector<u8> needle(m);
        for (auto & c : needle) c = (u8)(rng() % alpha);
        // the text: the needle with up to ~k+2 edits, then random bases; sometimes cut short
        std::vector<u8> text(needle);
        const int edits = (int)(rng() % (k + 3));
        for (int e = 0; e < edits && !text.empty(); ++e)
        {
            const size_t p = rng() % text.size();
            switch (rng() % 3)
            {
                case 0: text[p] = (u8)(rng() % alpha); break;
                case 1: text.insert(text.begin() + p, (u8)(rng() % alpha)); break;
                default: text.erase(text.begin() + p); break;
            }
        }
        const int extra = (int)(rng() % 24);
        for (int e = 0; e < extra; ++e) text.push_back((u8)(rng() % alpha));
        if (it % 9 == 0 && !text.empty()) text.resize(rng() % text.size());
        if (it % 17 == 0 && !needle.empty()) needle[rng() % needle.size()] = 4;    // N matches nothing
        if (it % 19 == 0 && !text.empty()) text[rng() % text.size()] = 4;
        const Reports want = plainDp(needle, text, k);
        withReports += !want.empty();
        auto nd = [&](int i) -> u32 { return needle[(size_t)i]; };
        auto tx = [&](int j) -> u32 { return text[(size_t)j]; };
        Reports gotBanded;
        if (k <= 15 && it % 2) myersPrefixSearchBanded<u32>(nd, (int)needle.size(), tx, (int)text.size(), k, [&](int pos, int dist) { gotBanded.push_back({pos, dist}); });
        else myersPrefixSearchBanded<u64>(nd, (int)needle.size(), tx, (int)text.size(), k, [&](int pos, int dist) { gotBanded.push_back({pos, dist}); });
        ++checkedBanded;
        if (gotBanded != want)
        {
            printf("BANDED MISMATCH it=%d k=%d m=%d t=%zu: want %zu reports, got %zu\n", it, k, m, text.size(), want.size(), gotBanded.size());
            for (auto & r : want) printf(" want (%d,%d)", r.first, r.second);
            printf("\n");
            for (auto & r : gotBanded) printf(" got (%d,%d)", r.first, r.second);
            printf("\n");
            return 1;
        }
        if (m <= 256)
        {
            Reports gotBlock;
            myersPrefixSearch<4>(nd, (int)needle.size(), tx, (int)text.size(), k, [&](int pos, int dist) { gotBlock.push_back({pos, dist}); });
            ++checkedBlock;
            if (gotBlock != want) { printf("BLOCK MISMATCH it=%d k=%d m=%d\n", it, k, m); return 1; }
        }
    }
    // the dispatcher K3 calls: every branch of prefixSearch<16> (one-word band up to 15 / 31 errors, block form beyond: budgets up to 40), and the
    // small instantiations up to their contract bound
    long checkedDispatch = 0;
    for (int it = 0; it < rounds / 4; ++it)
    {
        const int k = (int)(rng() % 41);
        const int m = (int)(rng() % (it % 3 == 0 ? 900 : 120));
        const int alpha = it % 5 == 0 ? 2 : 4;
        std::vector<u8> needle(m);
        for (auto & c : needle) c = (u8)(rng() % alpha);
        std::vector<u8> text(needle);
        const int edits = (int)(rng() % (k + 3));
        for (int e = 0; e < edits && !text.empty(); ++e)
        {
            const size_t p = rng() % text.size();
            switch (rng() % 3)
            {
                case 0: text[p] = (u8)(rng() % alpha); break;
                case 1: text.insert(text.begin() + p, (u8)(rng() % alpha)); break;
                default: text.erase(text.begin() + p); break;
            }
        }
        for (int e = (int)(rng() % 50); e > 0; --e) text.push_back((u8)(rng() % alpha));
        const Reports want = plainDp(needle, text, k);
        auto nd = [&](int i) -> u32 { return needle[(size_t)i]; };
        auto tx = [&](int j) -> u32 { return text[(size_t)j]; };
        Reports got;
        auto rep = [&](int pos, int dist) { got.push_back({pos, dist}); };
        if (m <= 1024) { prefixSearch<16>(nd, m, tx, (int)text.size(), k, rep); ++checkedDispatch; if (got != want) { printf("DISPATCH<16> MISMATCH it=%d k=%d m=%d\n", it, k, m); return 1; } }
        if (k <= PREFIX_SEARCH_SMALL_MAX_K && m <= 256)
        {
            got.clear(); prefixSearch<4>(nd, m, tx, (int)text.size(), k, rep); ++checkedDispatch;
            if (got != want) { printf("DISPATCH<4> MISMATCH it=%d k=%d m=%d\n", it, k, m); return 1; }
            if (m <= 128) { got.clear(); prefixSearch<2>(nd, m, tx, (int)text.size(), k, rep); ++checkedDispatch; if (got != want) { printf("DISPATCH<2> MISMATCH it=%d k=%d m=%d\n", it, k, m); return 1; } }
        }
    }
    printf("ok banded=%ld block=%ld dispatch=%ld with_reports=%ld\n", checkedBanded, checkedBlock, checkedDispatch, withReports);
    return 0;
}

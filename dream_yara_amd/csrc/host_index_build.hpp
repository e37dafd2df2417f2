// Host-side builder of one bin's 12 index files in the reference's on-disk format, so that the
// files stay interchangeable with the reference's mapper/indexer (reference src/d_indexer.cpp:278-333,389-396,
// src/store_seqs.h:222-253, src/misc_types.h:151-188).  Suffix sorting is prefix doubling over packed 64-bit keys
// (one key sort resolves 21 symbols at a time for single-contig bins), which finishes random genomes in one round.
#pragma once
#include "dy_types.h"
#include "host_formats.hpp"
#include <algorithm>
#include <functional>
#include <numeric>
#include <random>
#include <string>
#include <vector>

namespace dy {

struct FastaRecord { std::string name; std::vector<u8> codes; };

inline std::vector<FastaRecord> readFasta(const std::string & path)
{
    MappedFile f(path);
    std::vector<FastaRecord> recs;
    const u8 * p = f.data(), * end = p + f.size();
    while (p < end)
    {
        while (p < end && (*p == '\n' || *p == '\r')) ++p;
        if (p >= end) break;
        if (*p != '>') throw IoError("not a FASTA file: " + path);
        const u8 * s = ++p;
        while (p < end && *p != '\n') ++p;
        FastaRecord r;
        const u8 * e = p;
        while (e > s && (e[-1] == '\r')) --e;
        r.name.assign((const char *)s, (const char *)e);
        while (p < end && *p != '>')
        {
            u8 c = *p++;
            if (c == '\n' || c == '\r' || c == ' ') continue;
            r.codes.push_back(dna5Code((char)c));
        }
        recs.push_back(std::move(r));
    }
    return recs;
}

// Suffix array of t[0..n), symbols in [0, sigma).  Every contig ends with its own sentinel value, so no two suffixes
// compare equal and comparisons never run past a sentinel.
inline std::vector<u32> suffixArrayDoubling(const std::vector<u32> & t, u32 sigma)
{
    const u32 n = (u32)t.size();
    u32 bits = 1; while ((1u << bits) <= sigma) ++bits;       // symbols are shifted by +1; 0 pads beyond the end
    const u32 q = 63 / bits;                                   // symbols packed per key
    const u64 mask = (1ull << (q * bits)) - 1;
    auto sym = [&](u64 i) -> u64 { return i < n ? (u64)t[i] + 1 : 0; };
    std::vector<u64> key(n);
    {
        u64 k = 0;
        for (u32 j = 0; j < q; ++j) k = (k << bits) | sym(j);
        key[0] = k;
        for (u32 i = 1; i < n; ++i) { k = ((k << bits) & mask) | sym((u64)i + q - 1); key[i] = k; }
    }
    std::vector<u32> sa(n), rank(n), tmpRank(n);
    std::iota(sa.begin(), sa.end(), 0u);
    std::sort(sa.begin(), sa.end(), [&](u32 a, u32 b) { return key[a] < key[b]; });
    auto assignRanks = [&](auto && same) -> bool {
        bool allUnique = true;
        tmpRank[sa[0]] = 0;
        for (u32 i = 1; i < n; ++i)
        {
            bool eq = same(sa[i - 1], sa[i]);
            tmpRank[sa[i]] = eq ? tmpRank[sa[i - 1]] : i;
            allUnique &= !eq;
        }
        rank.swap(tmpRank);
        return allUnique;
    };
    bool done = assignRanks([&](u32 a, u32 b) { return key[a] == key[b]; });
    for (u32 h = q; !done; h *= 2)
    {
        auto second = [&](u32 i) -> u64 { return i + h < n ? (u64)rank[i + h] + 1 : 0; };
        for (u32 i = 0; i < n; ++i) key[i] = ((u64)rank[i] << 32) | second(i);
        // only ranges of equal rank need sorting
        u32 s = 0;
        while (s < n)
        {
            u32 e = s + 1;
            while (e < n && rank[sa[e]] == rank[sa[s]]) ++e;
            if (e - s > 1) std::sort(sa.begin() + s, sa.begin() + e, [&](u32 a, u32 b) { return key[a] < key[b]; });
            s = e;
        }
        done = assignRanks([&](u32 a, u32 b) { return key[a] == key[b]; });
    }
    return sa;
}

struct BuiltBin
{
    // contigs
    u64 maxLength = 0, count = 0, sum = 0;
    std::vector<u64> limits;
    std::vector<u8> concat;           // Dna5 codes (Ns kept)
    std::vector<u32> nameLimits; std::string names;
    // FM index
    u32 n = 0; u32 pst[5] = {0, 0, 0, 0, 0}; u8 drs = 0;
    std::vector<u32> drp;
    std::vector<u8> drv, ind, val;
};

// `sorter` (optional): suffix array of (text, alphabet size) — the device sort of libdyhip when the indexer runs on a GPU box
typedef std::function<std::vector<u32>(const std::vector<u32> &, u32)> SuffixSorter;
// `fmBuilder` (optional): everything behind the text — suffix array, BWT, rank dictionary, sentinel rows, sampled suffix array — into B
// (dy_fm_build of libdyhip: the whole FM construction on the device); it takes precedence over `sorter`
struct BuiltBin;
typedef std::function<void(const std::vector<u32> & text, u32 nContigs, const std::vector<u32> & starts, BuiltBin & B)> FmBuilder;
inline BuiltBin buildBin(const std::vector<FastaRecord> & recs, const SuffixSorter & sorter = SuffixSorter(), const FmBuilder & fmBuilder = FmBuilder())
{
    BuiltBin B;
    B.count = recs.size();
    if (B.count == 0) throw IoError("a bin must hold at least one contig");
    if (B.count > 65535) throw IoError("Maximum number of contigs exceeded. Recompile with -DDR_YARA_LARGE_CONTIGS=ON.");      // reference src/d_indexer.cpp:376-382
    B.limits.push_back(0); B.nameLimits.push_back(0);
    for (auto & r : recs)
    {
        B.concat.insert(B.concat.end(), r.codes.begin(), r.codes.end());
        B.limits.push_back(B.concat.size());
        B.maxLength = std::max<u64>(B.maxLength, r.codes.size());
        size_t sp = r.name.find_first_of(" \t");                     // trimSeqNames, reference src/store_seqs.h:194-216
        B.names += r.name.substr(0, sp);
        B.nameLimits.push_back((u32)B.names.size());
    }
    B.sum = B.concat.size();
    if (B.sum + B.count >= 0xFFFFFFF0ull) throw IoError("bins above 4 Gbp are not supported");
    const u32 cnt = (u32)B.count, n = (u32)(B.sum + B.count);
    B.n = n;

    // text = reverse(randomizeNs(contig_c)) $c ...   symbols: $c -> c, base b -> cnt + b
    std::vector<u32> T(n);
    std::vector<u32> start(cnt + 1);
    {
        std::mt19937 rng(0xDEADBEEF);                                // one generator per bin, reference src/store_seqs.h:244-252
        u32 p = 0;
        for (u32 c = 0; c < cnt; ++c)
        {
            const u64 lo = B.limits[c], len = B.limits[c + 1] - lo;
            start[c] = p;
            std::vector<u8> tmp(B.concat.begin() + lo, B.concat.begin() + lo + len);
            for (auto & x : tmp) if (x > 3) x = (u8)(rng() % 4);
            for (u64 i = 0; i < len; ++i) T[p++] = cnt + tmp[len - 1 - i];
            T[p++] = c;
        }
        start[cnt] = p;
    }
    if (fmBuilder) { fmBuilder(T, cnt, start, B); return B; }
    std::vector<u32> sa = sorter ? sorter(T, cnt + 4) : suffixArrayDoubling(T, cnt + 4);

    // row -> (contig, offset); BWT with the sentinel substitute; sentinel rows
    std::vector<u32> contigOf(n);
    for (u32 c = 0; c < cnt; ++c) for (u32 p = start[c]; p < start[c + 1]; ++p) contigOf[p] = c;
    std::vector<u8> bwt(n), isSent(n, 0);
    u32 occ[4] = {0, 0, 0, 0};
    for (u32 r = 0; r < n; ++r)
    {
        u32 pos = sa[r], c = contigOf[pos];
        if (pos == start[c]) { isSent[r] = 1; B.drp.push_back(r); }
        else { bwt[r] = (u8)(T[pos - 1] - cnt); occ[bwt[r]]++; }
    }
    B.drs = 0;
    for (u8 c = 1; c < 4; ++c) if (occ[c] < occ[B.drs]) B.drs = c;   // least frequent symbol stands in for the sentinels
    B.pst[0] = cnt; B.pst[1] = B.pst[0] + occ[0]; B.pst[2] = B.pst[1] + occ[1]; B.pst[3] = B.pst[2] + occ[2]; B.pst[4] = n;
    // lf.drv: 32 B per 64 symbols
    const size_t nblk = ((size_t)n + 63) / 64;
    B.drv.assign(nblk * 32, 0);
    u32 run[4] = {0, 0, 0, 0};
    for (size_t b = 0; b < nblk; ++b)
    {
        u64 sym[2] = {0, 0};
        memcpy(B.drv.data() + b * 32 + 16, run, 16);
        for (u32 t = 0; t < 64 && b * 64 + t < n; ++t)
        {
            size_t r = b * 64 + t;
            u8 s = isSent[r] ? B.drs : bwt[r];
            sym[t >> 5] |= (u64)s << ((31 - (t & 31)) * 2);
            run[s]++;
        }
        memcpy(B.drv.data() + b * 32, sym, 16);
    }
    // sa.ind / sa.val: sample offsets divisible by 10, never the "$c" rows
    const size_t nind = ((size_t)n + 31) / 32;
    B.ind.assign(nind * 8, 0);
    u32 before = 0;
    for (size_t e = 0; e < nind; ++e)
    {
        u32 bitsWord = 0;
        memcpy(B.ind.data() + e * 8 + 4, &before, 4);
        for (u32 t = 0; t < 32 && e * 32 + t < n; ++t)
        {
            size_t r = e * 32 + t;
            u32 pos = sa[r], c = contigOf[pos], off = pos - start[c];
            bool sentinelSuffix = pos + 1 == start[c + 1];
            if (!sentinelSuffix && off % 10 == 0)
            {
                bitsWord |= 1u << (31 - t);
                // Pair<contig id, u32 offset>, packed: the id type follows the bin's own contig count (reference src/d_indexer.cpp:366-383)
                u8 recBytes[6]; const size_t idBytes = cnt <= 255 ? 1 : 2;
                recBytes[0] = (u8)c; if (idBytes == 2) recBytes[1] = (u8)(c >> 8);
                memcpy(recBytes + idBytes, &off, 4);
                B.val.insert(B.val.end(), recBytes, recBytes + idBytes + 4);
                ++before;
            }
        }
        memcpy(B.ind.data() + e * 8, &bitsWord, 4);
    }
    return B;
}

inline void writeAll(const std::string & path, const void * p, size_t n)
{
    FILE * f = fopen(path.c_str(), "wb");
    if (!f) throw IoError("cannot write " + path);
    bool ok = n == 0 || fwrite(p, 1, n, f) == n;
    fclose(f);
    if (!ok) throw IoError("short write to " + path);
}

inline void saveBin(const BuiltBin & B, const std::string & prefix)
{
    u64 sz[3] = {B.maxLength, B.count, B.sum};
    writeAll(prefix + ".txt.size", sz, 24);
    writeAll(prefix + ".txt.limits", B.limits.data(), B.limits.size() * 8);
    {
        const u64 len = B.concat.size();
        const size_t words = (size_t)((len + 20) / 21);
        std::vector<u64> packed(1 + words, 0);
        packed[0] = len;
        for (u64 i = 0; i < len; ++i) packed[1 + i / 21] |= (u64)B.concat[i] << (60 - 3 * (i % 21));
        writeAll(prefix + ".txt.concat", packed.data(), packed.size() * 8);
    }
    writeAll(prefix + ".rid.limits", B.nameLimits.data(), B.nameLimits.size() * 4);
    writeAll(prefix + ".rid.concat", B.names.data(), B.names.size());
    writeAll(prefix + ".sa.len", &B.n, 4);
    writeAll(prefix + ".sa.ind", B.ind.data(), B.ind.size());
    writeAll(prefix + ".sa.val", B.val.data(), B.val.size());
    writeAll(prefix + ".lf.pst", B.pst, 20);
    writeAll(prefix + ".lf.drv", B.drv.data(), B.drv.size());
    writeAll(prefix + ".lf.drs", &B.drs, 1);
    if (B.count <= 255) writeAll(prefix + ".lf.drp", B.drp.data(), B.drp.size() * 4);         // Naive dictionary: the sentinel rows
    else
    {   // Levels bool dictionary over the rows, laid out like sa.ind (see host_formats.hpp appendBin)
        const size_t nind = ((size_t)B.n + 31) / 32;
        std::vector<u8> lv(nind * 8, 0);
        std::vector<u32> bitsOf(nind, 0);
        for (u32 r : B.drp) bitsOf[r >> 5] |= 1u << (31 - (r & 31));
        u32 seen = 0;
        for (size_t e = 0; e < nind; ++e)
        {
            memcpy(lv.data() + e * 8, &bitsOf[e], 4); memcpy(lv.data() + e * 8 + 4, &seen, 4);
            seen += (u32)__builtin_popcount(bitsOf[e]);
        }
        writeAll(prefix + ".lf.drp", lv.data(), lv.size());
    }
}

} // namespace dy

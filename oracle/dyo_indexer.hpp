// ORACLE — TEST INFRASTRUCTURE ONLY (see dyo_formats.hpp header).
// CPU restatement of the per-bin FM-index builder and the IBF builder.
//   indexer : reference src/d_indexer.cpp:278-333,389-396 (randomizeNs -> reverse -> FM index -> save .sa/.lf)
//             + src/store_seqs.h:222-253 (N randomisation with mt19937(0xDEADBEEF), forward order, before reversal)
//   filter  : reference src/d_bloom_filter.h:127-147,293-315 (addFastaFile/_addKmers/_insertKmer)
// The SeqAn internals (suffix sorting, rank-dictionary layout, sparse SA) are restated from SURVEY.md
// Appendix A and pinned by byte-equality with the 64 gold bins (tests/test_oracle_indexer.py).
#pragma once
#include "dyo_formats.hpp"
#include <random>

namespace dyo {

// ----------------------------------------------------------------------------
// SA-IS (induced sorting) over an integer alphabet. s[n-1] must be the unique smallest symbol.
// ----------------------------------------------------------------------------
namespace sais {

inline void getBuckets(const int32_t * s, int32_t n, int32_t K, std::vector<int32_t> & bkt, bool end)
{
    std::fill(bkt.begin(), bkt.end(), 0);
    for (int32_t i = 0; i < n; ++i) bkt[s[i]]++;
    int32_t sum = 0;
    for (int32_t i = 0; i < K; ++i) { sum += bkt[i]; bkt[i] = end ? sum : sum - bkt[i]; }
}

inline void induce(const int32_t * s, int32_t * SA, int32_t n, int32_t K, const std::vector<bool> & isS, std::vector<int32_t> & bkt)
{
    getBuckets(s, n, K, bkt, false);
    for (int32_t i = 0; i < n; ++i)
    {
        int32_t j = SA[i] - 1;
        if (SA[i] > 0 && !isS[j]) SA[bkt[s[j]]++] = j;
    }
    getBuckets(s, n, K, bkt, true);
    for (int32_t i = n - 1; i >= 0; --i)
    {
        int32_t j = SA[i] - 1;
        if (SA[i] > 0 && isS[j]) SA[--bkt[s[j]]] = j;
    }
}

inline void build(const int32_t * s, int32_t * SA, int32_t n, int32_t K)
{
    if (n == 1) { SA[0] = 0; return; }
    std::vector<bool> isS(n);
    isS[n - 1] = true;
    for (int32_t i = n - 2; i >= 0; --i)
        isS[i] = s[i] < s[i + 1] || (s[i] == s[i + 1] && isS[i + 1]);
    auto isLMS = [&](int32_t i) { return i > 0 && isS[i] && !isS[i - 1]; };

    std::vector<int32_t> bkt(K);
    getBuckets(s, n, K, bkt, true);
    std::fill(SA, SA + n, -1);
    for (int32_t i = 1; i < n; ++i) if (isLMS(i)) SA[--bkt[s[i]]] = i;
    induce(s, SA, n, K, isS, bkt);

    // compact sorted LMS substrings
    int32_t n1 = 0;
    for (int32_t i = 0; i < n; ++i) if (isLMS(SA[i])) SA[n1++] = SA[i];
    std::fill(SA + n1, SA + n, -1);
    int32_t name = 0, prev = -1;
    for (int32_t i = 0; i < n1; ++i)
    {
        int32_t pos = SA[i];
        bool diff = false;
        if (prev < 0) diff = true;
        else
            for (int32_t d = 0; d < n; ++d)
            {
                if (s[pos + d] != s[prev + d] || isS[pos + d] != isS[prev + d]) { diff = true; break; }
                if (d > 0 && (isLMS(pos + d) || isLMS(prev + d))) break;
            }
        if (diff) { ++name; prev = pos; }
        SA[n1 + pos / 2] = name - 1;
    }
    for (int32_t i = n - 1, j = n - 1; i >= n1; --i) if (SA[i] >= 0) SA[j--] = SA[i];

    int32_t * SA1 = SA;
    int32_t * s1 = SA + n - n1;
    if (name < n1)
        build(s1, SA1, n1, name);
    else
        for (int32_t i = 0; i < n1; ++i) SA1[s1[i]] = i;

    getBuckets(s, n, K, bkt, true);
    for (int32_t i = 1, j = 0; i < n; ++i) if (isLMS(i)) s1[j++] = i;
    for (int32_t i = 0; i < n1; ++i) SA1[i] = s1[SA1[i]];
    std::fill(SA + n1, SA + n, -1);
    for (int32_t i = n1 - 1; i >= 0; --i)
    {
        int32_t j = SA[i]; SA[i] = -1;
        SA[--bkt[s[j]]] = j;
    }
    induce(s, SA, n, K, isS, bkt);
}

} // namespace sais

// ----------------------------------------------------------------------------
// Build the 12-file index of one bin from (names, Dna5 codes per contig).
// ----------------------------------------------------------------------------
struct BinIndexFiles
{
    Contigs contigs;
    FMIndex fm;
    std::vector<u8> saValFile;     // packed (contig, offset) records
    std::vector<u8> drpFile;       // lf.drp: position list (<= 255 contigs) or Levels bool dictionary (see loadFMIndex)
};

inline BinIndexFiles buildBinIndex(const std::vector<std::string> & names, const std::vector<std::vector<u8> > & seqs)
{
    BinIndexFiles out;
    Contigs & C = out.contigs;
    C.count = seqs.size();
    if (C.count > 65535) throw std::runtime_error("Maximum number of contigs exceeded. Recompile with -DDR_YARA_LARGE_CONTIGS=ON.");   // reference src/d_indexer.cpp:376-382
    C.limits.assign(1, 0);
    for (size_t i = 0; i < seqs.size(); ++i)
    {
        // names are trimmed at the first whitespace (reference src/store_seqs.h:194-216)
        std::string nm = names[i];
        size_t sp = nm.find_first_of(" \t");
        if (sp != std::string::npos) nm.resize(sp);
        C.names.push_back(nm);
        C.concat.insert(C.concat.end(), seqs[i].begin(), seqs[i].end());
        C.limits.push_back(C.concat.size());
        C.maxLength = std::max<u64>(C.maxLength, seqs[i].size());
    }
    C.sum = C.concat.size();

    // randomizeNs: one generator per bin, contigs in order, forward direction (src/store_seqs.h:222-253)
    std::mt19937 rng(0xDEADBEEF);
    std::vector<std::vector<u8> > txt(seqs);
    for (auto & s : txt)
        for (auto & c : s)
            if (c == 4) c = (u8)(rng() % 4);
    // reverse (src/d_indexer.cpp:296)
    for (auto & s : txt) std::reverse(s.begin(), s.end());

    // generalized text: s0 $0 s1 $1 ... ; $i < $j for i<j, all sentinels < A
    const int32_t cnt = (int32_t)C.count;
    const int32_t n = (int32_t)(C.sum + C.count);
    std::vector<int32_t> T(n + 1), SA(n + 1);
    std::vector<u32> startOf(cnt);      // start of contig c in T
    {
        int32_t p = 0;
        for (int32_t c = 0; c < cnt; ++c)
        {
            startOf[c] = (u32)p;
            for (u8 ch : txt[c]) T[p++] = 1 + cnt + ch;
            T[p++] = 1 + c;
        }
        T[p] = 0;  // global terminator for SA-IS only
    }
    sais::build(T.data(), SA.data(), n + 1, cnt + 5);
    // drop the global terminator row (SA[0] == n)
    // rows 0..cnt-1 are the suffixes "$c" in contig order; then proper suffixes.
    FMIndex & fm = out.fm;
    fm.n = (u32)n;
    std::vector<u8> bwt(n);
    std::vector<u8> isSent(n, 0);
    // map text position -> (contig, offset)
    std::vector<u32> contigOf(n);
    {
        int32_t p = 0;
        for (int32_t c = 0; c < cnt; ++c) { for (size_t k = 0; k <= txt[c].size(); ++k) contigOf[p++] = (u32)c; }
    }
    std::vector<u32> rowContig(n), rowOffset(n);
    for (int32_t r = 0; r < n; ++r)
    {
        int32_t pos = SA[r + 1];
        u32 c = contigOf[pos];
        u32 off = (u32)pos - startOf[c];
        rowContig[r] = c; rowOffset[r] = off;
        if (off == 0) { isSent[r] = 1; bwt[r] = 0; }            // char before contig start = sentinel
        else bwt[r] = (u8)(T[pos - 1] - 1 - cnt);
    }
    for (int32_t r = 0; r < n; ++r) if (isSent[r]) fm.drp.push_back((u32)r);
    // prefix sums: sentinels first
    u32 occ[4] = {0, 0, 0, 0};
    for (int32_t r = 0; r < n; ++r) if (!isSent[r]) occ[bwt[r]]++;
    // the sentinel substitute is the least frequent BWT symbol (first on ties) — pinned by the 64 gold lf.drs files
    fm.drs = 0;
    for (u8 c = 1; c < 4; ++c) if (occ[c] < occ[fm.drs]) fm.drs = c;
    fm.pst[0] = (u32)cnt; fm.pst[1] = fm.pst[0] + occ[0]; fm.pst[2] = fm.pst[1] + occ[1]; fm.pst[3] = fm.pst[2] + occ[2]; fm.pst[4] = (u32)n;
    // rank blocks (sentinel slots hold the substitute symbol and ARE counted; lf() corrects)
    size_t nblk = ((size_t)n + 63) / 64;
    fm.drv.resize(nblk);
    u32 run[4] = {0, 0, 0, 0};
    for (size_t b = 0; b < nblk; ++b)
    {
        RankBlock & blk = fm.drv[b];
        blk.sym[0] = blk.sym[1] = 0;
        for (unsigned c = 0; c < 4; ++c) blk.cnt[c] = run[c];
        for (unsigned t = 0; t < 64; ++t)
        {
            size_t r = b * 64 + t;
            if (r >= (size_t)n) break;
            u8 s = isSent[r] ? fm.drs : bwt[r];
            blk.sym[t >> 5] |= (u64)s << ((31 - (t & 31)) * 2);
            run[s]++;
        }
    }
    // sparse SA: sample rows with offset % 10 == 0 that are not a "$c" row
    size_t nind = ((size_t)n + 31) / 32;
    fm.ind.assign(nind, IndEntry{0, 0});
    u32 before = 0;
    for (size_t e = 0; e < nind; ++e)
    {
        fm.ind[e].before = before;
        for (unsigned t = 0; t < 32; ++t)
        {
            size_t r = e * 32 + t;
            if (r >= (size_t)n) break;
            bool sentinelSuffix = rowOffset[r] == txt[rowContig[r]].size();
            if (!sentinelSuffix && rowOffset[r] % 10 == 0)
            {
                fm.ind[e].bits |= 1u << (31 - t);
                fm.valContig.push_back((u16)rowContig[r]);
                fm.valOffset.push_back(rowOffset[r]);
                ++before;
            }
        }
    }
    // the contig-id type follows the bin's own contig count (reference src/d_indexer.cpp:366-383)
    fm.contigIdBytes = cnt <= 255 ? 1 : 2;
    const size_t rec = fm.contigIdBytes + 4;
    out.saValFile.resize(fm.valOffset.size() * rec);
    for (size_t i = 0; i < fm.valOffset.size(); ++i)
    {
        out.saValFile[i * rec] = (u8)fm.valContig[i];
        if (fm.contigIdBytes == 2) out.saValFile[i * rec + 1] = (u8)(fm.valContig[i] >> 8);
        memcpy(out.saValFile.data() + i * rec + fm.contigIdBytes, &fm.valOffset[i], 4);
    }
    if (cnt <= 255)
    {
        out.drpFile.resize(fm.drp.size() * 4);
        memcpy(out.drpFile.data(), fm.drp.data(), out.drpFile.size());
    }
    else
    {   // Levels bool dictionary over the rows, laid out like sa.ind
        out.drpFile.assign(nind * 8, 0);
        u32 seen = 0;
        for (size_t e = 0; e < nind; ++e)
        {
            u32 bits = 0;
            for (unsigned t = 0; t < 32 && e * 32 + t < (size_t)n; ++t) if (isSent[e * 32 + t]) bits |= 1u << (31 - t);
            memcpy(out.drpFile.data() + e * 8, &bits, 4);
            memcpy(out.drpFile.data() + e * 8 + 4, &seen, 4);
            seen += (u32)__builtin_popcount(bits);
        }
    }
    return out;
}

inline void saveBinIndex(const BinIndexFiles & b, const std::string & prefix)
{
    saveContigs(b.contigs, prefix);
    const FMIndex & fm = b.fm;
    writeFileBytes(prefix + ".sa.len", &fm.n, 4);
    writeFileBytes(prefix + ".sa.ind", fm.ind.data(), fm.ind.size() * 8);
    writeFileBytes(prefix + ".sa.val", b.saValFile.data(), b.saValFile.size());
    writeFileBytes(prefix + ".lf.pst", fm.pst, 20);
    writeFileBytes(prefix + ".lf.drv", fm.drv.data(), fm.drv.size() * 32);
    writeFileBytes(prefix + ".lf.drs", &fm.drs, 1);
    writeFileBytes(prefix + ".lf.drp", b.drpFile.data(), b.drpFile.size());
}

// ----------------------------------------------------------------------------
// IBF k-mer hashing / insertion
// ----------------------------------------------------------------------------
// Rolling 2-bit hash of SeqAn's Shape<Dna,SimpleShape> over a Dna5 text (reference src/d_bloom_filter.h:168-176,
// 304-315; SeqAn hashInit/hashNext restated in SURVEY.md B.3 — UNPINNED by any reference test: N maps to A except
// that hashInit takes the very first character uncast, so an N at position 0 contributes 4).
// Calls f(i, hash) for every k-mer start i.
template <typename F>
inline void forEachKmerHash(const u8 * text, size_t len, unsigned k, F && f)
{
    if (len < k) return;
    auto ordDna = [](u8 c) -> u64 { return c > 3 ? 0 : c; };
    u64 leftFactor = 1; for (unsigned i = 1; i < k; ++i) leftFactor *= 4;
    // hashInit: hValue = ordValue(text[0]) [uncast], then text[1..k-2] cast to Dna; leftChar = 0
    u64 h = text[0];
    for (unsigned i = 1; i + 1 < k; ++i) h = h * 4 + ordDna(text[i]);
    u64 leftChar = 0;
    for (size_t i = 0; i + k <= len; ++i)
    {
        h = (h - leftChar * leftFactor) * 4 + ordDna(text[i + k - 1]);
        leftChar = ordDna(text[i]);
        f(i, h);
    }
}

// reference src/d_bloom_filter.h:286-291
inline u64 ibfHashPos(const IBF & f, u64 kmerHash, unsigned i)
{
    if (f.direct) return kmerHash * f.blockBitSize;          // reference src/d_kdx_filter.h:159,258
    u64 v = f.preCalc[i] * kmerHash;
    v ^= v >> IBF::SHIFT;
    v %= f.noOfHashPos;
    v *= f.blockBitSize;
    return v;
}

// reference src/d_bloom_filter.h:293-315
inline void ibfAddKmers(IBF & f, const u8 * text, size_t len, u32 binNo)
{
    forEachKmerHash(text, len, f.kmerSize, [&](size_t, u64 h) {
        if (f.direct && h >= f.noOfHashPos) return;
        for (unsigned i = 0; i < f.noOfHashFunc; ++i)
        {
            u64 bit = ibfHashPos(f, h, i) + binNo;
            f.words[bit >> 6] |= 1ull << (bit & 63);
        }
    });
}

} // namespace dyo

// ORACLE — TEST INFRASTRUCTURE ONLY (see dyo_formats.hpp header).
// CPU restatement of DREAM-Yara's mapping hot path, function by function:
//   IBF sweep            reference src/d_bloom_filter.h:161-224, src/d_mapper.h:83-92,309-387
//   per-bin pipeline     reference src/d_mapper.h:252-303 (same staging as src/mapper.h:1299-1352)
//   seeds                reference src/mapper_collector.h:143-177
//   FM search            reference src/index_fm.h:132-155 + SeqAn find(...Backtracking<>)
//   classify / rank      reference src/mapper_classifier.h:111-155, src/mapper_ranker.h:122-177
//   extension            reference src/mapper_extender.h:147-308, src/find_extender.h:87-403
//   dedupe / strata      reference src/bits_matches.h:504-528,669-758, src/mapper.h:772-925
//   DREAM merge          reference src/d_mapper.h:127-245,610-721,828-838
// PINNED by: tests/golden/gold/64-viral-reads.{t1,rec.t1,tag.t1}.sam (SE, -e 3, strata 0 and 10).
// UNPINNED: paired-end, -y full, L=150/e=7 tail behaviour (no reference gold exists).
#pragma once
#include <stdexcept>
#include "dyo_formats.hpp"
#include "dyo_indexer.hpp"
#include <cmath>
#include <map>
#include <functional>

namespace dyo {

// ----------------------------------------------------------------------------
// Options (reference src/mapper.h:49-131, src/d_mapper.cpp:270-394)
// ----------------------------------------------------------------------------
enum Sensitivity { LOW = 0, HIGH = 1, FULL = 2 };

struct Options
{
    float    errorRate = 0.05f;
    float    strataRate = 0.00f;
    u64      strataCount = 0xFFFFFFFFull;   // (uint64_t)-1u
    int      sensitivity = HIGH;
    unsigned hitsThreshold = 300;
    unsigned readsCount = 100000;
    unsigned threads = 1;
    bool     useFilter = true;
    float    indelRate = 0.25f;
    unsigned libraryLength = 0, libraryDev = 0;
    bool     verifyMatches = true;
    // NOT a reference option — checker only: evaluate the pairing weights with the device's fixed IEEE sequences (dyo_detmath.hpp) instead of
    // libm's std::pow / std::erfc as the reference does (src/bits_matches.h:898-966).  tests/test_oracle_pe_math.py shows both give
    // the same pairing decisions; the default is the reference's libm.
    bool     detMath = false;
    std::string readGroup = "none";
    std::string version = "1.0.2 []";
    std::string commandLine;
};

static const unsigned MAX_MATCH_ERRORS = 127;   // MemberBits<Match,Errors> = 7

// reference src/misc_options.h:258-263
inline unsigned getReadErrors(const Options & o, unsigned readLen)
{
    return std::min<unsigned>((unsigned)(readLen * o.errorRate), MAX_MATCH_ERRORS);
}
// reference src/misc_options.h:283-287
inline unsigned getReadStrata(const Options & o, unsigned readLen)
{
    unsigned s = o.strataCount != 0xFFFFFFFFull ? (unsigned)o.strataCount : (unsigned)(readLen * o.strataRate);
    return std::min<unsigned>(s, MAX_MATCH_ERRORS);
}
// reference src/misc_options.h:270-275
inline unsigned getReadIndels(const Options & o, unsigned readLen)
{
    return std::min<unsigned>((unsigned)(readLen * o.indelRate), MAX_MATCH_ERRORS);
}
// reference src/d_mapper.h:83-92
inline u16 getThreshold(const Options & o, u32 kmerSize, u16 readLen)
{
    u16 maxError = (u16)(o.errorRate * readLen);
    if (kmerSize * (1u + maxError) > readLen) return 0;
    return (u16)(readLen - kmerSize * (1u + maxError) + 1);
}

// ----------------------------------------------------------------------------
// IBF query (reference src/d_bloom_filter.h:161-224): OR into `selected` every bin whose k-mer count >= threshold.
// ----------------------------------------------------------------------------
inline void ibfWhichBins(const IBF & f, std::vector<bool> & selected, const u8 * text, size_t len, u16 threshold,
                         std::vector<u16> * countsOut = nullptr)
{
    std::vector<u16> counts(f.noOfBins, 0);
    forEachKmerHash(text, len, f.kmerSize, [&](size_t, u64 h) {
        if (f.direct && h >= f.noOfHashPos) return;              // src/d_kdx_filter.h:155-156
        u64 idx[8];
        for (unsigned i = 0; i < f.noOfHashFunc; ++i) idx[i] = ibfHashPos(f, h, i);
        for (u32 batch = 0; batch < f.binIntWidth; ++batch)
        {
            u64 tmp = f.getInt(idx[0]);
            for (unsigned i = 1; i < f.noOfHashFunc; ++i) tmp &= f.getInt(idx[i]);
            // :196-211 == add every set bit j of tmp to counts[64*batch + j]
            while (tmp)
            {
                unsigned j = (unsigned)__builtin_ctzll(tmp);
                tmp &= tmp - 1;
                u32 bin = batch * 64 + j;
                if (bin < f.noOfBins) ++counts[bin];
            }
            for (unsigned i = 0; i < f.noOfHashFunc; ++i) idx[i] += 64;
        }
    });
    for (u32 b = 0; b < f.noOfBins; ++b)
        if (counts[b] >= threshold) selected[b] = true;
    if (countsOut) *countsOut = counts;
}

// ----------------------------------------------------------------------------
// Match record (reference src/bits_matches.h:116-127) — unpacked
// ----------------------------------------------------------------------------
struct Match
{
    u32 readId = 0;
    u32 contigId = 0;
    bool isRev = false;
    u64 contigBegin = 0;
    u32 contigLen = 0;      // the reference's 13-bit "contigEnd" member holds the LENGTH
    u32 errors = MAX_MATCH_ERRORS;
    u64 contigEnd() const { return contigBegin + contigLen; }
    bool valid() const { return contigLen != 0; }
};

// reference src/bits_matches.h:504-528 (ContigSize bits = 32 here; only the lexicographic order matters).
// The reference sorts with std::sort on (contig, strand, end|begin, errors) and so leaves matches with equal keys in an
// unspecified order; the restatement fixes that order with the remaining coordinate as a last key (a TOTAL order), which
// is what the device path uses too.  The gold SAMs are reproduced either way.
struct KeyEnd   { bool operator()(const Match & a, const Match & b) const {
    if (a.contigId != b.contigId) return a.contigId < b.contigId;
    if (a.isRev != b.isRev) return a.isRev < b.isRev;
    if (a.contigEnd() != b.contigEnd()) return a.contigEnd() < b.contigEnd();
    if (a.errors != b.errors) return a.errors < b.errors;
    return a.contigBegin < b.contigBegin; } };
struct KeyBegin { bool operator()(const Match & a, const Match & b) const {
    if (a.contigId != b.contigId) return a.contigId < b.contigId;
    if (a.isRev != b.isRev) return a.isRev < b.isRev;
    if (a.contigBegin != b.contigBegin) return a.contigBegin < b.contigBegin;
    if (a.errors != b.errors) return a.errors < b.errors;
    return a.contigEnd() < b.contigEnd(); } };
inline bool dupEnd(const Match & a, const Match & b)   { return a.contigId == b.contigId && a.isRev == b.isRev && a.contigEnd() == b.contigEnd(); }
inline bool dupBegin(const Match & a, const Match & b) { return a.contigId == b.contigId && a.isRev == b.isRev && a.contigBegin == b.contigBegin; }

// reference src/bits_matches.h:669-725: two in-place compaction passes over one read's matches.
// Returns the number of unique matches; the array keeps its stale tail (src/d_mapper.h:136-152 forwards it).
template <typename TDup>
inline size_t compactUnique(Match * m, size_t n, TDup dup)
{
    size_t nw = 0, old = 0;
    while (old != n)
    {
        m[nw] = m[old];
        ++old;
        while (old != n && dup(m[nw], m[old])) ++old;
        ++nw;
    }
    return nw;
}
inline size_t removeDuplicates(std::vector<Match> & m)
{
    std::stable_sort(m.begin(), m.end(), KeyEnd());
    size_t n1 = compactUnique(m.data(), m.size(), dupEnd);
    std::stable_sort(m.begin(), m.begin() + n1, KeyBegin());
    return compactUnique(m.data(), n1, dupBegin);
}

// ----------------------------------------------------------------------------
// Bin = contigs + FM index
// ----------------------------------------------------------------------------
struct Bin
{
    Contigs contigs;
    FMIndex fm;
};

struct Hit { u32 lo, hi; u32 seedId; u8 errors; };   // reference src/bits_hits.h:59-65

// exact / Hamming-backtracking search of one seed (reference src/index_fm.h:132-155)
inline void fmSearch(const FMIndex & fm, const u8 * seed, unsigned len, unsigned maxErr, u32 seedId, std::vector<Hit> & hits)
{
    struct Frame { u32 lo, hi; u16 depth; u8 errs; };
    Frame stack[4 * 256 + 4];
    int sp = 0;
    stack[sp++] = Frame{0, fm.n, 0, 0};
    while (sp)
    {
        Frame f = stack[--sp];
        // exact descent once the budget is exhausted
        while (f.depth < len && f.errs == maxErr)
        {
            u8 c = seed[f.depth];
            if (c > 3) { f.lo = f.hi = 0; break; }
            f.lo = fm.lf(f.lo, c); f.hi = fm.lf(f.hi, c);
            if (f.lo >= f.hi) break;
            ++f.depth;
        }
        if (f.lo >= f.hi) continue;
        if (f.depth == len) { hits.push_back(Hit{f.lo, f.hi, seedId, f.errs}); continue; }
        // children in reverse so that A is expanded first (order is immaterial to the result set)
        for (int c = 3; c >= 0; --c)
        {
            u32 lo = fm.lf(f.lo, (u8)c), hi = fm.lf(f.hi, (u8)c);
            if (lo >= hi) continue;
            u8 e = f.errs + ((u8)c != seed[f.depth] ? 1 : 0);
            stack[sp++] = Frame{lo, hi, (u16)(f.depth + 1), e};
        }
    }
}

// Edit-distance backtracking of one seed — `-y full`, reference src/mapper.h:607-649 with TSeedsDistance = EditDistance
// (src/d_mapper.cpp:417-436): find(index, seeds, errors, delegate, Backtracking<EditDistance>).  PARITY UNPINNED: SeqAn's enumeration
// lives in the absent submodule and no fixture uses -y full.  Defined here as: depth-first over the FM trie (children A, C, G, T), every
// node carrying the band |i - depth| <= maxErr of the DP column D[i] = ed(seed[0..i), path); a node is reported, with errors = D[len],
// when D[len] <= maxErr, and expanded while some band cell is <= maxErr.  Nested nodes are separate hits (their occurrences end at
// different text positions, and the END of an occurrence is what the extension anchors on, src/mapper_extender.h:263-285).
// What the reads' final records depend on is the set of anchors — every text position where an occurrence within maxErr edits of the
// seed ends — which any complete enumeration yields; multiplicities only enter the seed ranking, whose order cannot change the matches
// within the reported strata (the early exit waits for rank * (E + 1) >= minErrors + strata, by which every such match has been
// reached through some seed whatever the order) but can change which representative of a duplicated location survives.
inline void fmSearchEdit(const FMIndex & fm, const u8 * seed, unsigned len, unsigned maxErr, u32 seedId, std::vector<Hit> & hits)
{
    const unsigned E = maxErr, BW = 2 * E + 1, INF = E + 1;
    if (maxErr > 2) throw std::logic_error("fmSearchEdit: the band of a frame holds 2 * 2 + 1 cells (seed errors are 1 or 2, reference src/mapper_classifier.h:141-149)");
    struct Frame { u32 lo, hi; u8 alt; u8 band[5]; };
    std::vector<Frame> st(len + E + 2);
    int level = 0;
    st[0].lo = 0; st[0].hi = fm.n; st[0].alt = 0;
    for (unsigned t = 0; t < BW; ++t) st[0].band[t] = t >= E ? (u8)(t - E) : (u8)INF;     // D[i] = i at depth 0, i = t - E
    while (level >= 0)
    {
        Frame & f = st[level];
        if (f.alt == 0 && level > 0)
        {   // first visit: report when the whole seed is consumed within the budget
            const int t = (int)len - level + (int)E;
            if (t >= 0 && t < (int)BW && f.band[t] <= E) hits.push_back(Hit{f.lo, f.hi, seedId, f.band[t]});
        }
        if (f.alt == 4 || (unsigned)level == len + E) { --level; continue; }
        const u8 c = f.alt++;
        // child band: cell t' stands for i' = level + 1 - E + t'
        u8 nb[5]; unsigned best = INF;
        for (unsigned t = 0; t < BW; ++t)
        {
            const int i = level + 1 - (int)E + (int)t;
            unsigned v = INF;
            if (i >= 0 && i <= (int)len)
            {
                if (i == 0) v = (unsigned)level + 1;                                        // nothing of the seed against level + 1 text characters
                else
                {
                    const unsigned diag = f.band[t] + ((seed[i - 1] == c && c < 4) ? 0u : 1u);           // parent cell i - 1 sits at the same t
                    const unsigned up = t + 1 < BW ? (unsigned)f.band[t + 1] + 1 : INF;                 // parent cell i: the text character stays unmatched
                    const unsigned left = t > 0 ? (unsigned)nb[t - 1] + 1 : INF;                          // child cell i - 1: the seed character stays unmatched
                    v = std::min(diag, std::min(up, left));
                }
                if (v > INF) v = INF;
            }
            nb[t] = (u8)v;
            best = std::min(best, v);
        }
        if (best > E) continue;
        const u32 lo = fm.lf(f.lo, c), hi = fm.lf(f.hi, c);
        if (lo >= hi) continue;
        Frame & g = st[level + 1];
        g.lo = lo; g.hi = hi; g.alt = 0;
        for (unsigned t = 0; t < BW; ++t) g.band[t] = nb[t];
        ++level;
    }
}

// ----------------------------------------------------------------------------
// Banded prefix edit distance (stands in for SeqAn Myers<AlignTextBanded<FindPrefix,NMatchesNone_,NMatchesNone_>>,
// SURVEY.md C.2): for every text prefix length j>=0 with ed(needle, text[0..j)) <= k call f(j-1, dist) in
// increasing j (j = 0, the empty prefix, is reported as position -1 when the needle is no longer than k — pinned by
// gold read bin.53.22, whose only error is a mismatch at the last base).  N matches nothing.
// ----------------------------------------------------------------------------
template <typename TGetN, typename TGetT, typename F>
inline void prefixEditSearch(TGetN needle, int m, TGetT text, int t, int k, F && f)
{
    if (m < 0 || t < 0) return;
    const int INF = 1 << 20;
    // column-wise DP over text positions, rows = needle; band |i-j| <= k
    std::vector<int> prev(m + 1), cur(m + 1);
    for (int i = 0; i <= m; ++i) prev[i] = i <= k ? i : INF;
    if (m <= k) f(-1, m);
    for (int j = 1; j <= t; ++j)
    {
        int lo = std::max(1, j - k), hi = std::min(m, j + k);
        cur[0] = j <= k ? j : INF;
        if (lo > 1) cur[lo - 1] = INF;
        for (int i = lo; i <= hi; ++i)
        {
            int sub = prev[i - 1] + (dnaEq(needle(i - 1), text(j - 1)) ? 0 : 1);
            int del = (i <= j - 1 + k ? prev[i] : INF) + 1;     // text char unmatched
            int ins = cur[i - 1] + 1;                            // needle char unmatched
            cur[i] = std::min(sub, std::min(del, ins));
        }
        if (hi < m) cur[hi + 1] = INF;
        if (hi == m && cur[m] <= k) f(j - 1, cur[m]);
        std::swap(prev, cur);
        if (j - k > m) break;
    }
}

// ----------------------------------------------------------------------------
// Seed extension (reference src/find_extender.h:87-403), Hamming seeds.
// contig: real (N-containing) sequence; read: the read sequence of the seed's strand.
// Returns true and fills (matchBegin, matchEnd, matchErrors) when the delegate would be called.
// ----------------------------------------------------------------------------
inline bool extendSeed(const u8 * contig, u64 contigLen, const u8 * read, unsigned readLen,
                       u64 hayBegin, u64 hayEnd, unsigned needleBegin, unsigned needleEnd, unsigned maxErrors,
                       u64 & matchBegin, u64 & matchEnd, unsigned & matchErrors, bool editSeeds = false)
{
    matchErrors = 0;
    if (hayEnd - hayBegin != needleEnd - needleBegin) return false;
    // Hamming seeds: _checkSeed :87-106, then the left extension starts at the seed's begin.  Edit-distance seeds (-y full): no check,
    // the left extension starts at the seed's END and re-aligns the seed region too (:314-322: haystackLeftEnd / needleLeftEnd).
    u64 leftAnchorHay = hayEnd; unsigned leftAnchorNeedle = needleEnd;
    if (!editSeeds)
    {
        u64 h = hayBegin; unsigned q = needleBegin;
        for (; h != hayEnd && matchErrors <= maxErrors; ++h, ++q)
            matchErrors += !dnaEq(contig[h], read[q]);
        if (!(h == hayEnd && matchErrors <= maxErrors)) return false;
        leftAnchorHay = hayBegin; leftAnchorNeedle = needleBegin;
    }
    matchBegin = leftAnchorHay;
    // ---- extend left :334-366, _extendLeft :112-174
    if (leftAnchorNeedle > 0)
    {
        u64 leftOffset = needleBegin + (maxErrors - matchErrors);
        u64 leftBegin = hayBegin > leftOffset ? hayBegin - leftOffset : 0;
        const u64 hayBeginSeed = hayBegin; const unsigned needleBeginSeed = needleBegin;
        hayBegin = leftAnchorHay; needleBegin = leftAnchorNeedle;          // the anchor of the reversed views below
        // reversed views: hay[i] = contig[hayBegin-1-i], i < hayLen; needle[i] = read[needleBegin-1-i]
        int hayLen = (int)(hayBegin - leftBegin), ndlLen = (int)needleBegin;
        int lcp = 0;
        while (lcp < hayLen && lcp < ndlLen && dnaEq(contig[hayBegin - 1 - lcp], read[needleBegin - 1 - lcp])) ++lcp;
        if (lcp == ndlLen)
            matchBegin -= lcp;
        else
        {
            unsigned remaining = maxErrors - matchErrors;
            unsigned minErrors = remaining + 1;
            int endPos = 0;
            if (!remaining) return false;
            int hl = hayLen - lcp, nl = ndlLen - lcp;
            u64 hb = hayBegin - 1 - lcp; unsigned nb = needleBegin - 1 - lcp;
            prefixEditSearch([&](int i) { return read[nb - i]; }, nl, [&](int j) { return contig[hb - j]; }, hl, (int)remaining,
                             [&](int pos, int dist) {
                                 if ((unsigned)dist <= minErrors) { minErrors = (unsigned)dist; endPos = pos + 1; }
                             });
            matchErrors += minErrors;
            matchBegin -= (u64)(endPos + lcp);
            if (matchErrors > maxErrors) return false;
        }
        hayBegin = hayBeginSeed; needleBegin = needleBeginSeed;
    }
    // ---- extend right :369-396, _extendRight :180-254
    matchEnd = hayEnd;
    if (needleEnd < readLen)
    {
        u64 rightOffset = readLen - needleBegin + (maxErrors - matchErrors);
        u64 rightEnd = contigLen;
        if (rightEnd > hayBegin + rightOffset) rightEnd = hayBegin + rightOffset;
        int hayLen = (int)(rightEnd - hayEnd), ndlLen = (int)(readLen - needleEnd);
        int lcp = 0;
        while (lcp < hayLen && lcp < ndlLen && dnaEq(contig[hayEnd + lcp], read[needleEnd + lcp])) ++lcp;
        if (lcp == ndlLen)
            matchEnd += lcp;
        else if (lcp == hayLen)
        {
            matchErrors += ndlLen - hayLen;
            matchEnd += lcp;
            if (matchErrors > maxErrors) return false;
        }
        else
        {
            unsigned remaining = maxErrors - matchErrors;
            unsigned minErrors = remaining + 1;
            int endPos = 0;
            if (!remaining) return false;
            int hl = hayLen - lcp, nl = ndlLen - lcp;
            u64 hb = hayEnd + lcp; unsigned nb = needleEnd + lcp;
            u8 lastNeedle = read[nb + nl - 1];
            // last base held out on both sides :223-227
            prefixEditSearch([&](int i) { return read[nb + i]; }, nl - 1, [&](int j) { return contig[hb + j]; }, hl - 1, (int)remaining,
                             [&](int pos, int dist) {
                                 int currentEnd = pos + 1;
                                 unsigned cur = (unsigned)dist;
                                 if (!dnaEq(contig[hb + currentEnd], lastNeedle))
                                     if (++cur > remaining) return;
                                 if (cur <= minErrors) { minErrors = cur; endPos = currentEnd; }
                             });
            matchErrors += minErrors;
            matchEnd += (u64)(endPos + lcp + 1);
            if (matchErrors > maxErrors) return false;
        }
    }
    return true;
}

// ----------------------------------------------------------------------------
// Per-bin mapping of one read (fwd + revcomp). Appends every extension result to `out`.
// Restates the read-local view of _mapReadsImpl (reference src/d_mapper.h:252-294): all stages are
// data-parallel over reads, so running them read by read yields the same per-read match multiset.
// ----------------------------------------------------------------------------
struct ReadCtx { u8 seedErrors[2] = {0, 0}; u8 minErrors = 255; bool mapped = false; };

struct SeedSet
{
    // seeds of one read sequence in seedId order (descending offset, reference src/mapper_collector.h:166-171)
    unsigned count = 0, length = 0;
    unsigned offset(unsigned id) const { return (count - 1 - id) * length; }
};

inline SeedSet collectSeedsFor(unsigned readLen, unsigned readErrors, unsigned seedErrors)
{
    SeedSet s;
    s.count = (unsigned)std::ceil((readErrors + 1) / (seedErrors + 1.0));
    s.length = readLen / s.count;
    return s;
}

struct BinMapStats { u64 seeds = 0, hits = 0, located = 0, extended = 0, matches = 0; };

class ReadBinMapper
{
public:
    ReadBinMapper(const Bin & bin, const Options & opt) : bin(bin), opt(opt) {}

    // seqs[0] = forward, seqs[1] = reverse complement
    void map(const u8 * const seqs[2], unsigned readLen, u32 readId, std::vector<Match> & out, BinMapStats * st = nullptr)
    {
        stats = st;
        ReadCtx ctx;
        const unsigned readErrors = getReadErrors(opt, readLen);
        const unsigned readStrata = getReadStrata(opt, readLen);
        std::vector<Hit> hits[3][2];      // [bucket][strand]
        SeedSet seeds[3];
        bool have[3] = {false, false, false};

        // ---- round 0 ------------------------------------------------------
        seeds[0] = collectSeedsFor(readLen, readErrors, 0); have[0] = true;          // collectSeeds<0>
        findSeeds(seqs, seeds[0], 0, hits[0]);                                        // findSeeds<0>(0)
        // classifyReads (reference src/mapper_classifier.h:111-155)
        {
            u64 total = countHits(hits[0][0]) + countHits(hits[0][1]);
            if (total > opt.hitsThreshold)
            {
                unsigned se = (total < 2ull * 200 * opt.hitsThreshold) ? 1 : 2;
                ctx.seedErrors[0] = ctx.seedErrors[1] = (u8)se;
                for (int s = 0; s < 2; ++s) for (auto & h : hits[0][s]) h.lo = h.hi = 0;   // clearHits: ranges zeroed
            }
        }
        for (unsigned b = 1; b <= 2; ++b)                                             // collectSeeds<1>,<2>; findSeeds<0>(1),(2)
            if (ctx.seedErrors[0] == b)
            {
                seeds[b] = collectSeedsFor(readLen, readErrors, b); have[b] = true;
                findSeeds(seqs, seeds[b], 0, hits[b]);
            }
        // NOTE: bucket-0 seeds exist for every read (seedErrors was 0 at collect time), also for re-classified ones,
        // whose hits are now empty ranges.
        for (unsigned b = 0; b <= 2; ++b)
            if (have[b]) extendHits(seqs, readLen, readId, seeds[b], hits[b], 0, readErrors, readStrata, ctx, out);
        // ---- round 1 ------------------------------------------------------
        for (unsigned b = 0; b <= 2; ++b) { have[b] = false; hits[b][0].clear(); hits[b][1].clear(); }
        for (unsigned b = 1; b <= 2; ++b)
            if (!ctx.mapped && ctx.seedErrors[0] == b)
            {
                seeds[b] = collectSeedsFor(readLen, readErrors, b); have[b] = true;
                findSeeds(seqs, seeds[b], 1, hits[b]);
            }
        for (unsigned b = 1; b <= 2; ++b)
            if (have[b]) extendHits(seqs, readLen, readId, seeds[b], hits[b], 1, readErrors, readStrata, ctx, out);
        // ---- round 2 ------------------------------------------------------
        if (opt.sensitivity > LOW)
        {
            for (unsigned b = 0; b <= 2; ++b) { have[b] = false; hits[b][0].clear(); hits[b][1].clear(); }
            if (!ctx.mapped && ctx.seedErrors[0] == 2)
            {
                seeds[2] = collectSeedsFor(readLen, readErrors, 2); have[2] = true;
                findSeeds(seqs, seeds[2], 2, hits[2]);
                extendHits(seqs, readLen, readId, seeds[2], hits[2], 2, readErrors, readStrata, ctx, out);
            }
        }
    }

private:
    const Bin & bin;
    const Options & opt;
    BinMapStats * stats = nullptr;

    static u64 countHits(const std::vector<Hit> & h) { u64 n = 0; for (auto & x : h) n += x.hi - x.lo; return n; }

    void findSeeds(const u8 * const seqs[2], const SeedSet & ss, unsigned errors, std::vector<Hit> hits[2])
    {
        for (int s = 0; s < 2; ++s)
            for (unsigned id = 0; id < ss.count; ++id)
            {
                if (errors > 0 && opt.sensitivity == FULL) fmSearchEdit(bin.fm, seqs[s] + ss.offset(id), ss.length, errors, id, hits[s]);
                else fmSearch(bin.fm, seqs[s] + ss.offset(id), ss.length, errors, id, hits[s]);
                if (stats) stats->seeds++;
            }
        if (stats) stats->hits += hits[0].size() + hits[1].size();
    }

    // reference src/mapper_extender.h:147-208 (+ ranks from src/mapper_ranker.h:122-177)
    void extendHits(const u8 * const seqs[2], unsigned readLen, u32 readId, const SeedSet & ss, std::vector<Hit> hits[2],
                    unsigned seedErrors, unsigned readErrors, unsigned readStrata, ReadCtx & ctx, std::vector<Match> & out)
    {
        // rank seeds of each strand by hit count (stable: ties keep ascending seedId)
        std::vector<unsigned> rank[2];
        for (int s = 0; s < 2; ++s)
        {
            std::vector<u64> cnt(ss.count, 0);
            for (auto & h : hits[s]) cnt[h.seedId] += h.hi - h.lo;
            rank[s].resize(ss.count);
            for (unsigned i = 0; i < ss.count; ++i) rank[s][i] = i;
            std::stable_sort(rank[s].begin(), rank[s].end(), [&](unsigned a, unsigned b) { return cnt[a] < cnt[b]; });
        }
        for (unsigned seedRank = 0; seedRank < ss.count && !ctx.mapped; ++seedRank)
        {
            for (int s = 0; s < 2; ++s)
            {
                unsigned seedId = rank[s][seedRank];
                for (auto & h : hits[s])
                    if (h.seedId == seedId) extendHit(seqs[s], readLen, readId, s == 1, ss, h, readErrors, ctx, out);
            }
            if ((unsigned)ctx.minErrors + readStrata <= seedRank * (seedErrors + 1)) ctx.mapped = true;
        }
    }

    // reference src/mapper_extender.h:215-286
    void extendHit(const u8 * read, unsigned readLen, u32 readId, bool isRev, const SeedSet & ss, const Hit & h,
                   unsigned maxErrors, ReadCtx & ctx, std::vector<Match> & out)
    {
        unsigned needleBegin = ss.offset(h.seedId), needleEnd = needleBegin + ss.length;
        for (u32 row = h.lo; row < h.hi; ++row)
        {
            u32 contig, x;
            bin.fm.locate(row, contig, x);
            if (stats) stats->located++;
            u64 clen = bin.contigs.length(contig);
            if (clen - x < ss.length) continue;                 // suffixLength < seedLength
            u64 hayBegin = clen - x - ss.length;                // invert the reversed-text coordinate
            u64 mb, me; unsigned errs;
            if (stats) stats->extended++;
            if (!extendSeed(bin.contigs.seq(contig), clen, read, readLen, hayBegin, hayBegin + ss.length,
                            needleBegin, needleEnd, maxErrors, mb, me, errs, opt.sensitivity == FULL))
                continue;
            Match m;
            m.readId = readId; m.contigId = contig; m.isRev = isRev;
            m.contigBegin = mb; m.contigLen = (u32)(me - mb); m.errors = errs;
            out.push_back(m);
            if (stats) stats->matches++;
            if (errs < ctx.minErrors) ctx.minErrors = (u8)errs;  // setMinErrors, src/bits_context.h:128-133
        }
    }
};

// ----------------------------------------------------------------------------
// libc++ std::default_random_engine + uniform_int_distribution(0, n-1), fresh engine per read
// (reference src/mapper.h:886-906; SURVEY.md B.1 — gold was produced with libc++).
// ----------------------------------------------------------------------------
inline u64 libcxxFirstUniform(u64 n)
{
    if (n <= 1) return 0;
    // minstd_rand: x <- 48271 x mod 2147483647, seed 1; engine range [1, 2147483646] => R = 2147483646, log2(R) -> 30 bits
    u64 x = 1;
    unsigned w = 0; while ((1ull << w) < n) ++w;         // bits needed for [0, n-1]  (n not nec. power of 2)
    // __independent_bits_engine with w <= 30: single draw, low... libc++ takes (x - min) & mask of w0 bits after
    // rejecting x - min >= y0 where y0 = (R >> w0) << w0
    const u64 R = 2147483646ull;
    unsigned w0 = w;
    u64 y0 = (R >> w0) << w0;
    u64 mask = w0 >= 64 ? ~0ull : ((1ull << w0) - 1);
    for (;;)
    {
        u64 u;
        do { x = (x * 48271ull) % 2147483647ull; u = x - 1; } while (u >= y0);
        u64 v = u & mask;
        if (v < n) return v;
    }
}

// ----------------------------------------------------------------------------
// Banded affine alignment of a primary match -> CIGAR, with semi-global clipping
// (reference src/mapper_aligner.h:143-150,191-236).
// Score(match 0, mismatch -1000, gap extend -999, gap open -1001), contig horizontal with free end gaps.
// ----------------------------------------------------------------------------
struct CigarOp { char op; u32 count; };

inline std::string cigarToString(const std::vector<CigarOp> & c)
{
    std::string s;
    for (auto & e : c) { s += std::to_string(e.count); s += e.op; }
    return s;
}

// Returns cigar; updates begin offset (clip at begin) and the new end position (source coordinate of the clipped end).
inline std::vector<CigarOp> alignMatch(const u8 * contig, int n, const u8 * read, int m, int errors,
                                        int & clipBegin, int & clipEnd)
{
    std::vector<CigarOp> cigar;
    auto push = [&](char op) { if (!cigar.empty() && cigar.back().op == op) cigar.back().count++; else cigar.push_back(CigarOp{op, 1}); };
    clipBegin = 0; clipEnd = n;
    if (errors == 0 || (errors == 1 && n == m))
    {
        cigar.push_back(CigarOp{'M', (u32)m});
        return cigar;
    }
    const int NEG = -(1 << 28);
    const int MATCH = 0, MISMATCH = -1000, GEXT = -999, GOPEN = -1001;
    const int lower = -errors, upper = errors;     // band on j - i
    // full matrices (reads are short): S = best, H = gap in read (horizontal move, consumes contig), V = gap in contig
    std::vector<int> S((size_t)(m + 1) * (n + 1), NEG), H(S), V(S);
    std::vector<u8> trS(S.size(), 0), trH(S.size(), 0), trV(S.size(), 0);
    auto at = [&](int i, int j) { return (size_t)i * (n + 1) + j; };
    auto inBand = [&](int i, int j) { return j - i >= lower && j - i <= upper; };
    enum { DIAG = 1, HOR = 2, VER = 3 };
    for (int j = 0; j <= n && inBand(0, j); ++j) { S[at(0, j)] = 0; trS[at(0, j)] = 0; }   // free leading contig gaps (TTop)
    for (int i = 1; i <= m; ++i)
    {
        for (int j = std::max(0, i + lower); j <= std::min(n, i + upper); ++j)
        {
            size_t c = at(i, j);
            // vertical: consumes read char i-1 (insertion w.r.t. contig)
            if (inBand(i - 1, j))
            {
                int ext = V[at(i - 1, j)] + GEXT, opn = S[at(i - 1, j)] + GOPEN;
                if (ext >= opn) { V[c] = ext; trV[c] = 1; } else { V[c] = opn; trV[c] = 0; }
            }
            if (j > 0 && inBand(i, j - 1))
            {
                int ext = H[at(i, j - 1)] + GEXT, opn = S[at(i, j - 1)] + GOPEN;
                if (ext >= opn) { H[c] = ext; trH[c] = 1; } else { H[c] = opn; trH[c] = 0; }
            }
            int best = NEG; u8 tb = 0;
            if (j > 0 && inBand(i - 1, j - 1))
            {
                best = S[at(i - 1, j - 1)] + (dnaEq(contig[j - 1], read[i - 1]) ? MATCH : MISMATCH); tb = DIAG;
            }
            if (V[c] > best) { best = V[c]; tb = VER; }
            if (H[c] > best) { best = H[c]; tb = HOR; }
            if (j == 0) { /* first column: not free (TLeft=false) -> only vertical gaps */ }
            S[c] = best; trS[c] = tb;
        }
    }
    // free trailing contig gaps (TBottom): best cell of the last row, leftmost on ties
    int bj = -1, bs = NEG;
    for (int j = std::max(0, m + lower); j <= std::min(n, m + upper); ++j)
        if (S[at(m, j)] > bs) { bs = S[at(m, j)]; bj = j; }
    // traceback
    std::vector<char> ops;
    int i = m, j = bj, state = 0;   // 0 = S, 1 = V, 2 = H
    while (i > 0)
    {
        size_t c = at(i, j);
        if (state == 0)
        {
            if (trS[c] == DIAG) { ops.push_back('M'); --i; --j; }
            else if (trS[c] == VER) state = 1;
            else if (trS[c] == HOR) state = 2;
            else break;
        }
        else if (state == 1) { ops.push_back('I'); bool ext = trV[c]; --i; if (!ext) state = 0; }
        else { ops.push_back('D'); bool ext = trH[c]; --j; if (!ext) state = 0; }
    }
    clipBegin = j; clipEnd = bj;
    for (auto it = ops.rbegin(); it != ops.rend(); ++it) push(*it);
    return cigar;
}

// ----------------------------------------------------------------------------
// Mate rescue (reference src/find_verifier.h:80-140): UNBANDED semi-global affine alignment of the mate against the
// insert window (contig horizontal, free leading/trailing contig gaps), same scores and tie-breaks as alignMatch.
// On success returns true with the clipped window-relative [begin,end) and errors = score / -999.
// ----------------------------------------------------------------------------
inline bool verifyMate(const u8 * window, int n, const u8 * read, int m, unsigned maxErrors, unsigned maxIndels,
                       int & matchBegin, int & matchEnd, int & errorsOut)
{
    if (n <= 0) return false;
    const int NEG = -(1 << 28), MISMATCH = -1000, GEXT = -999, GOPEN = -1001;
    std::vector<int> S((size_t)(m + 1) * (n + 1), NEG), H(S), V(S);
    std::vector<u8> trS(S.size(), 0), trH(S.size(), 0), trV(S.size(), 0);
    auto at = [&](int i, int j) { return (size_t)i * (n + 1) + j; };
    enum { DIAG = 1, HOR = 2, VER = 3 };
    for (int j = 0; j <= n; ++j) S[at(0, j)] = 0;
    for (int i = 1; i <= m; ++i)
        for (int j = 0; j <= n; ++j)
        {
            size_t c = at(i, j);
            {
                int ext = V[at(i - 1, j)] + GEXT, opn = S[at(i - 1, j)] + GOPEN;
                if (ext >= opn) { V[c] = ext; trV[c] = 1; } else { V[c] = opn; trV[c] = 0; }
            }
            if (j > 0)
            {
                int ext = H[at(i, j - 1)] + GEXT, opn = S[at(i, j - 1)] + GOPEN;
                if (ext >= opn) { H[c] = ext; trH[c] = 1; } else { H[c] = opn; trH[c] = 0; }
            }
            int best = NEG; u8 tb = 0;
            if (j > 0) { best = S[at(i - 1, j - 1)] + (dnaEq(window[j - 1], read[i - 1]) ? 0 : MISMATCH); tb = DIAG; }
            if (V[c] > best) { best = V[c]; tb = VER; }
            if (H[c] > best) { best = H[c]; tb = HOR; }
            S[c] = best; trS[c] = tb;
        }
    int bj = -1, bs = NEG;
    for (int j = 0; j <= n; ++j) if (S[at(m, j)] > bs) { bs = S[at(m, j)]; bj = j; }
    int i = m, j = bj, state = 0;
    unsigned gapOpens = 0, gapChars = 0;
    int lastOp = -1;
    while (i > 0)
    {
        size_t c = at(i, j);
        int op;
        if (state == 0)
        {
            if (trS[c] == DIAG) { op = 0; --i; --j; }
            else if (trS[c] == VER) { state = 1; continue; }
            else if (trS[c] == HOR) { state = 2; continue; }
            else break;
        }
        else if (state == 1) { op = 1; bool ext = trV[c]; --i; if (!ext) state = 0; }
        else { op = 2; bool ext = trH[c]; --j; if (!ext) state = 0; }
        if (op != 0) { ++gapChars; if (op != lastOp) ++gapOpens; }
        lastOp = op;
    }
    int errors = bs / -999;
    unsigned gapExtensions = gapChars - gapOpens;
    unsigned events = (unsigned)errors + gapOpens - gapExtensions;
    if (events <= maxErrors && gapChars <= maxIndels)
    {
        matchBegin = j; matchEnd = bj; errorsOut = errors;
        return true;
    }
    return false;
}

} // namespace dyo

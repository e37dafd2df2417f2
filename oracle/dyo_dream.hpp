// ORACLE — TEST INFRASTRUCTURE ONLY (see dyo_formats.hpp header).
// CPU restatement of the DREAM driver: IBF classification, bin loop, merge, ranking and SAM text.
//   runDisMapper / clasifyLoadedReads / loadFilteredReads   reference src/d_mapper.h:309-443,860-890
//   copyMatches / copyCigars / rankMatches2 / finalize       reference src/d_mapper.h:127-245,610-721,828-838
//   rankMatches (child)                                      reference src/mapper.h:826-925
//   SAM records                                              reference src/mapper_writer.h:124-215,221-225,296-345,593-634
#pragma once
#include "dyo_mapper.hpp"
#include "dyo_detmath.hpp"
#include <ostream>
#include <sstream>
#include <thread>
#include <atomic>
#include <numeric>

namespace dyo {

struct ReadBatch
{
    std::vector<std::string> names;
    std::vector<std::vector<u8> > fwd, rev;     // Dna5 codes
    std::vector<std::string> qual;              // forward-strand quality characters
    size_t size() const { return fwd.size(); }
};

inline void appendRead(ReadBatch & b, const std::string & name, const std::string & seq, const std::string & qual)
{
    std::vector<u8> f(seq.size()), r(seq.size());
    for (size_t i = 0; i < seq.size(); ++i) f[i] = dna5Code(seq[i]);
    for (size_t i = 0; i < seq.size(); ++i) { u8 c = f[seq.size() - 1 - i]; r[i] = c < 4 ? (u8)(3 - c) : (u8)4; }
    std::string q(seq.size(), 'I');
    if (!qual.empty()) for (size_t i = 0; i < seq.size() && i < qual.size(); ++i) q[i] = (char)std::min<int>(qual[i], 33 + 62);
    for (size_t i = 0; i < seq.size(); ++i) if (f[i] == 4) q[i] = '!';     // Dna5Q: N carries no quality
    b.names.push_back(name); b.fwd.push_back(f); b.rev.push_back(r); b.qual.push_back(q);
}

struct DreamIndex
{
    std::vector<Bin> bins;
    std::vector<u32> contigOffsets;            // first global contig id of each bin
    std::vector<std::string> contigNames;      // all bins, bin order
    std::vector<u64> contigLengths;
};

inline DreamIndex loadDreamIndex(const std::string & dir, u32 numberOfBins)
{
    DreamIndex d;
    d.bins.resize(numberOfBins);
    u32 off = 0;
    for (u32 b = 0; b < numberOfBins; ++b)
    {
        std::string prefix = dir + "/" + std::to_string(b);
        d.bins[b].contigs = loadContigs(prefix);
        d.bins[b].fm = loadFMIndex(prefix, d.bins[b].contigs.count);
        d.contigOffsets.push_back(off);
        off += (u32)d.bins[b].contigs.count;
        for (u32 c = 0; c < d.bins[b].contigs.count; ++c)
        {
            d.contigNames.push_back(d.bins[b].contigs.names[c]);
            d.contigLengths.push_back(d.bins[b].contigs.length(c));
        }
    }
    return d;
}

// ---- MAPQ (reference src/bits_matches.h:898-949, src/mapper_writer.h:221-225) ----
inline double getResidualWeight(double errorRate) { double p = std::pow(10.0, std::min(2.0, 100.0 * errorRate - 7.0)); return p / (1.0 - p); }
inline double getMatchWeight(double errorRate, double optimalRate) { return (1.0 - errorRate) / std::pow(10.0, 300.0 * (errorRate - optimalRate)); }
inline double getMatchProb(double errorRate, double optimalRate, u64 optimalCount, u64 subCount)
{
    double w = optimalCount * getMatchWeight(optimalRate, optimalRate) + subCount * getMatchWeight(optimalRate + 0.01, optimalRate) +
               getResidualWeight(optimalRate);
    return getMatchWeight(errorRate, optimalRate) / w;
}
inline unsigned getMapq(double p) { return (unsigned)std::round(-10.0 * std::log10(1.0 - std::min(p, 0.9999999))); }

// Per-read result of the whole path (what the SAM writer consumes).
struct ReadResult
{
    bool mapped = false;
    Match primary;
    std::vector<Match> matches;      // matchesSet[readId]: within strata, sorted by (errors, contig, strand, begin)
    u32 bestCount = 0, subCount = 0;
    std::vector<CigarOp> cigar;
    bool paired = false;             // paired-end only: ReadsContext::paired of the main mapper
    double pairProb = 0.0;           // paired-end only: primaryMatchesProbs[read]
};

struct DreamStats { u64 filteredReads = 0; BinMapStats bin; std::vector<u64> readsPerBin; };

// sort one read's unique matches by errors (stable => coordinate order inside a stratum) and clip
inline void rankRead(std::vector<Match> & uniq, unsigned strata, ReadResult & r)
{
    std::stable_sort(uniq.begin(), uniq.end(), [](const Match & a, const Match & b) { return a.errors < b.errors; });
    r.matches.clear(); r.mapped = false; r.bestCount = r.subCount = 0;
    if (uniq.empty()) return;
    u32 best = uniq[0].errors;
    size_t nOpt = 0, nSet = 0;
    while (nOpt < uniq.size() && uniq[nOpt].errors <= best) ++nOpt;
    while (nSet < uniq.size() && uniq[nSet].errors <= best + strata) ++nSet;
    size_t pick = (size_t)libcxxFirstUniform(nOpt);
    r.primary = uniq[pick];
    r.mapped = r.primary.valid();
    r.matches.assign(uniq.begin(), uniq.begin() + nSet);
    r.bestCount = (u32)nOpt; r.subCount = (u32)(nSet - nOpt);
}

class DreamMapper
{
public:
    DreamMapper(const DreamIndex & idx, const IBF * filter, const Options & opt) : idx(idx), filter(filter), opt(opt) {}

    // reference src/d_mapper.h:309-387 — origReadIdMap[bin] = ascending read ids
    std::vector<std::vector<u32> > classify(const ReadBatch & reads) const
    {
        u32 numBins = (u32)idx.bins.size();
        std::vector<std::vector<u32> > map(numBins);
        size_t n = reads.size();
        if (!opt.useFilter || !filter)
        {
            for (u32 b = 0; b < numBins; ++b) { map[b].resize(n); std::iota(map[b].begin(), map[b].end(), 0u); }
            return map;
        }
        u64 lengthSum = 0;
        for (size_t i = 0; i < n; ++i) lengthSum += 2 * reads.fwd[i].size();
        u16 avgReadLen = (u16)(lengthSum / (n * 2));
        u16 threshold = getThreshold(opt, filter->kmerSize, avgReadLen);
        if (threshold == 0)   // :320-333 (every read to every bin; the reference then classifies again -> duplicates, B.12)
            for (u32 b = 0; b < numBins; ++b) for (size_t i = 0; i < n; ++i) map[b].push_back((u32)i);
        std::vector<std::vector<bool> > sel(n);
        unsigned T = std::max(1u, opt.threads);
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; ++t)
            th.emplace_back([&, t]() {
                for (size_t i = t; i < n; i += T)
                {
                    sel[i].assign(numBins, false);
                    ibfWhichBins(*filter, sel[i], reads.fwd[i].data(), reads.fwd[i].size(), threshold);
                    ibfWhichBins(*filter, sel[i], reads.rev[i].data(), reads.rev[i].size(), threshold);
                }
            });
        for (auto & x : th) x.join();
        for (size_t i = 0; i < n; ++i)
            for (u32 b = 0; b < numBins; ++b)
                if (sel[i][b]) map[b].push_back((u32)i);
        return map;
    }

    // Maps one batch; returns one ReadResult per read.
    std::vector<ReadResult> mapBatch(const ReadBatch & reads, DreamStats * stats = nullptr) const
    {
        size_t n = reads.size();
        std::vector<std::vector<u32> > origReadIdMap = classify(reads);
        u32 numBins = (u32)idx.bins.size();
        if (stats) { stats->readsPerBin.assign(numBins, 0); for (u32 b = 0; b < numBins; ++b) { stats->readsPerBin[b] = origReadIdMap[b].size(); stats->filteredReads += origReadIdMap[b].size(); } }

        // sortedBins :843-854 (descending load; ties by bin number — the reference's std::sort leaves ties unspecified)
        std::vector<u32> order(numBins);
        std::iota(order.begin(), order.end(), 0u);
        std::stable_sort(order.begin(), order.end(), [&](u32 a, u32 b) { return origReadIdMap[a].size() > origReadIdMap[b].size(); });

        std::vector<std::vector<Match> > mainMatches(n);          // per original read, in arrival order
        std::vector<u8> mainMinErrors(n, 255);
        std::vector<std::vector<CigarOp> > collectedCigars(n);

        struct ChildOut { std::vector<Match> all; bool hasPrimary = false; Match primary; std::vector<CigarOp> cigar; };

        for (u32 binNo : order)
        {
            const std::vector<u32> & ids = origReadIdMap[binNo];
            if (ids.empty()) continue;
            const Bin & bin = idx.bins[binNo];
            std::vector<ChildOut> outs(ids.size());
            unsigned T = std::max(1u, opt.threads);
            std::vector<BinMapStats> tstats(T);
            std::atomic<size_t> next(0);
            auto work = [&](unsigned t) {
                ReadBinMapper mapper(bin, opt);
                for (;;)
                {
                    size_t k = next.fetch_add(1);
                    if (k >= ids.size()) break;
                    u32 orig = ids[k];
                    unsigned L = (unsigned)reads.fwd[orig].size();
                    const u8 * seqs[2] = {reads.fwd[orig].data(), reads.rev[orig].data()};
                    ChildOut & o = outs[k];
                    mapper.map(seqs, L, orig, o.all, stats ? &tstats[t] : nullptr);
                    if (o.all.empty()) continue;
                    // aggregateMatches + rankMatches (child)
                    size_t nu = removeDuplicates(o.all);
                    std::vector<Match> uniq(o.all.begin(), o.all.begin() + nu);
                    // positions of uniq elements inside o.all are 0..nu-1 (coordinate order)
                    std::vector<size_t> pos(nu);
                    std::iota(pos.begin(), pos.end(), (size_t)0);
                    std::stable_sort(pos.begin(), pos.end(), [&](size_t a, size_t b) { return uniq[a].errors < uniq[b].errors; });
                    size_t nOpt = 0;
                    while (nOpt < nu && uniq[pos[nOpt]].errors <= uniq[pos[0]].errors) ++nOpt;
                    size_t p = pos[(size_t)libcxxFirstUniform(nOpt)];
                    // alignMatches: CIGAR of the primary, shrink the match in place (src/mapper_aligner.h:191-219)
                    Match & pm = o.all[p];
                    const u8 * rs = pm.isRev ? reads.rev[orig].data() : reads.fwd[orig].data();
                    int cb, ce;
                    o.cigar = alignMatch(bin.contigs.seq(pm.contigId) + pm.contigBegin, (int)pm.contigLen, rs, (int)L, (int)pm.errors, cb, ce);
                    if (!(pm.errors == 0 || (pm.errors == 1 && pm.contigLen == L)))
                    {
                        pm.contigBegin += (u64)cb;
                        pm.contigLen = (u32)ce;      // sic: endPosition(contigGaps), see src/mapper_aligner.h:213-218
                    }
                    o.hasPrimary = true; o.primary = pm;
                }
            };
            if (T == 1) work(0);
            else { std::vector<std::thread> th; for (unsigned t = 0; t < T; ++t) th.emplace_back(work, t); for (auto & x : th) x.join(); }
            if (stats) for (auto & s : tstats) { stats->bin.seeds += s.seeds; stats->bin.hits += s.hits; stats->bin.located += s.located; stats->bin.extended += s.extended; stats->bin.matches += s.matches; }
            // copyMatches / copyCigars :127-215
            u32 coff = idx.contigOffsets[binNo];
            for (size_t k = 0; k < ids.size(); ++k)
            {
                u32 orig = ids[k];
                ChildOut & o = outs[k];
                for (Match m : o.all) { m.contigId += coff; m.readId = orig; mainMatches[orig].push_back(m); }
                if (o.hasPrimary)
                {
                    if (o.primary.errors < mainMinErrors[orig]) mainMinErrors[orig] = (u8)o.primary.errors;
                    if (mainMinErrors[orig] == o.primary.errors) collectedCigars[orig] = o.cigar;
                }
            }
        }
        // finalizeMainMapper :828-838 -> aggregateMatches + rankMatches2
        std::vector<ReadResult> res(n);
        for (size_t i = 0; i < n; ++i)
        {
            std::vector<Match> & all = mainMatches[i];
            size_t nu = removeDuplicates(all);
            std::vector<Match> uniq(all.begin(), all.begin() + nu);
            rankRead(uniq, getReadStrata(opt, (unsigned)reads.fwd[i].size()), res[i]);
            res[i].cigar = collectedCigars[i];
        }
        return res;
    }

    // =====================================================================================================
    // Paired-end (reference src/mapper.h:927-1172, src/bits_matches.h:816-1041, src/mapper_verifier.h, src/find_verifier.h,
    // src/d_mapper.h:186-215,394-443).  UNPINNED: the reference ships no paired-end gold.
    // Batch layout like the reference: reads [0,P) are the first mates, [P,2P) the second mates.
    // =====================================================================================================
    static bool orientationProper(const Match & one, const Match & two)
    {
        bool oneBeforeTwo = one.contigBegin < two.contigBegin;
        return (!one.isRev && two.isRev && oneBeforeTwo) || (!two.isRev && one.isRev && !oneBeforeTwo);
    }
    static u32 libraryLengthOf(const Match & a, const Match & b)
    {
        if (a.valid() && b.valid() && a.contigId == b.contigId)
            return (u32)(b.contigEnd() > a.contigBegin ? b.contigEnd() - a.contigBegin : a.contigEnd() - b.contigBegin);
        return 0xFFFFFFFFu;
    }
    static u32 libraryDeviationOf(const Match & a, const Match & b, unsigned mean)
    {
        if (a.valid() && b.valid() && a.contigId == b.contigId)
        {
            int d = (int)libraryLengthOf(a, b) - (int)mean;
            return (u32)(d < 0 ? -d : d);
        }
        return 0xFFFFFFFFu;
    }
    // getResidualWeight / getMatchWeight / getMatchProb / getLibraryProb — reference src/bits_matches.h:898-966 (std::pow, std::erfc)
    double pow10Of(double x) const { return opt.detMath ? detPow10(x) : std::pow(10.0, x); }
    double peResidualWeight(double errorRate) const { double p = pow10Of(std::min(2.0, 100.0 * errorRate - 7.0)); return p / (1.0 - p); }
    double peMatchWeight(double errorRate, double optimalRate) const { return (1.0 - errorRate) / pow10Of(300.0 * (errorRate - optimalRate)); }
    double peMatchProb(double errorRate, double optimalRate, u64 optimalCount, u64 subCount) const
    {
        double w = optimalCount * peMatchWeight(optimalRate, optimalRate) + subCount * peMatchWeight(optimalRate + 0.01, optimalRate) + peResidualWeight(optimalRate);
        return peMatchWeight(errorRate, optimalRate) / w;
    }
    double libraryProbOf(const Match & one, const Match & two, unsigned mean, unsigned stdDev) const
    {
        bool proper = orientationProper(one, two) && libraryDeviationOf(one, two, mean) < 6 * stdDev;
        if (!proper) return 0.0009;
        double dev = (double)libraryDeviationOf(one, two, mean);
        double score = dev / stdDev;
        return std::max(0.001, opt.detMath ? detErfc(score / 1.41421356237) : std::erfc(score / 1.41421356237));
    }

    // findProperMates (src/bits_matches.h:819-867): [first,last) into `mates` (sorted by coordinate)
    std::pair<size_t, size_t> findProperMates(const std::vector<Match> & mates, const Match & match, unsigned mateLength, u64 contigLength,
                                              unsigned mean, unsigned stdDev) const
    {
        Match leq = match; leq.isRev = !match.isRev; Match geq = leq;
        leq.errors = 0; geq.errors = MAX_MATCH_ERRORS;
        int ml = (int)mateLength;
        unsigned deltaMinus = (unsigned)std::max(0, (int)mean - (int)(6 * stdDev) - ml);
        unsigned deltaPlus = (unsigned)std::max(0, (int)mean + (int)(6 * stdDev) - ml);
        auto addPos = [&](Match & m, unsigned delta) { m.contigBegin = (m.contigBegin + m.contigLen + delta < contigLength) ? m.contigBegin + delta : contigLength - m.contigLen; };
        auto subPos = [&](Match & m, unsigned delta) { m.contigBegin = m.contigBegin > delta ? m.contigBegin - delta : 0; };
        if (!match.isRev) { addPos(leq, deltaMinus); addPos(geq, deltaPlus); }
        else { subPos(leq, deltaPlus); subPos(geq, deltaMinus); }
        auto keyB = [](const Match & a, const Match & b) {
            if (a.contigId != b.contigId) return a.contigId < b.contigId;
            if (a.isRev != b.isRev) return a.isRev < b.isRev;
            if (a.contigBegin != b.contigBegin) return a.contigBegin < b.contigBegin;
            return a.errors < b.errors; };
        auto keyE = [](const Match & a, const Match & b) {
            if (a.contigId != b.contigId) return a.contigId < b.contigId;
            if (a.isRev != b.isRev) return a.isRev < b.isRev;
            if (a.contigEnd() != b.contigEnd()) return a.contigEnd() < b.contigEnd();
            return a.errors < b.errors; };
        size_t first = std::lower_bound(mates.begin(), mates.end(), leq, keyB) - mates.begin();
        size_t last = std::upper_bound(mates.begin(), mates.end(), geq, keyE) - mates.begin();
        if (first > last) return {0, 0};
        return {first, last};
    }

    // findPrimaryMatch (src/bits_matches.h:972-1041): index into firstMatches (or npos) and its probability
    std::pair<size_t, double> findPrimaryMatch(const std::vector<Match> & firstMatches, const std::vector<Match> & secondMatches,
                                               double firstOptimalRate, double secondOptimalRate, u64 secondOptimalCount, u64 secondSubCount,
                                               unsigned firstLen, unsigned secondLen, u64 contigLengthOf(const void *, u32), const void * cl,
                                               unsigned mean, unsigned stdDev) const
    {
        double sum = 0, best = 0; size_t bestIdx = (size_t)-1;
        for (size_t i = 0; i < firstMatches.size(); ++i)
        {
            const Match & fm = firstMatches[i];
            double w = 0;
            u64 optImproper = secondOptimalCount, subImproper = secondSubCount;
            auto range = findProperMates(secondMatches, fm, secondLen, contigLengthOf(cl, fm.contigId), mean, stdDev);
            for (size_t k = range.first; k < range.second; ++k)
            {
                const Match & sm = secondMatches[k];
                double secondErrorRate = (double)((float)sm.errors / (float)secondLen);
                w += peMatchWeight(secondErrorRate, secondOptimalRate) * libraryProbOf(fm, sm, mean, stdDev);
                if (secondErrorRate == secondOptimalRate) --optImproper;
                else if (secondErrorRate == secondOptimalRate + 0.01) --subImproper;
            }
            w += ((double)optImproper * peMatchWeight(secondOptimalRate, secondOptimalRate) +
                  (double)subImproper * peMatchWeight(secondOptimalRate + 0.01, secondOptimalRate) + peResidualWeight(secondOptimalRate)) * 0.0009;
            double firstErrorRate = (double)((float)fm.errors / (float)firstLen);
            w *= peMatchWeight(firstErrorRate, firstOptimalRate);
            if (w > best) { best = w; bestIdx = i; }
            sum += w;
        }
        return {bestIdx, best / sum};
    }

    static u64 contigLenThunk(const void * p, u32 c) { return ((const Bin *)p)->contigs.length(c); }

    std::vector<ReadResult> mapBatchPaired(const ReadBatch & reads, DreamStats * stats = nullptr) const
    {
        const size_t n = reads.size(), P = n / 2;
        const u32 numBins = (u32)idx.bins.size();
        // clasifyLoadedReads for pairs: the four sequences of a pair vote together (src/d_mapper.h:350-358)
        std::vector<std::vector<u32> > pairMap(numBins);
        if (!opt.useFilter || !filter)
            for (u32 b = 0; b < numBins; ++b) { pairMap[b].resize(P); std::iota(pairMap[b].begin(), pairMap[b].end(), 0u); }
        else
        {
            u64 lengthSum = 0;
            for (size_t i = 0; i < n; ++i) lengthSum += 2 * reads.fwd[i].size();
            u16 avg = (u16)(lengthSum / (n * 2));
            u16 thr = getThreshold(opt, filter->kmerSize, avg);
            // the votes of different pairs are independent: computed on opt.threads threads, appended in pair order
            std::vector<std::vector<bool> > sels(P);
            const unsigned T = std::max(1u, opt.threads);
            std::vector<std::thread> th;
            for (unsigned t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    for (size_t p = t; p < P; p += T)
                    {
                        std::vector<bool> & sel = sels[p];
                        sel.assign(numBins, thr == 0);
                        if (thr)
                            for (size_t r : {p, p + P})
                            {
                                if (reads.fwd[r].size() < filter->kmerSize) continue;
                                ibfWhichBins(*filter, sel, reads.fwd[r].data(), reads.fwd[r].size(), thr);
                                ibfWhichBins(*filter, sel, reads.rev[r].data(), reads.rev[r].size(), thr);
                            }
                    }
                });
            for (auto & x : th) x.join();
            for (size_t p = 0; p < P; ++p)
                for (u32 b = 0; b < numBins; ++b) if (sels[p][b]) pairMap[b].push_back((u32)p);
        }
        if (stats) { stats->readsPerBin.assign(numBins, 0); for (u32 b = 0; b < numBins; ++b) { stats->readsPerBin[b] = 2 * pairMap[b].size(); stats->filteredReads += 2 * pairMap[b].size(); } }
        std::vector<u32> order(numBins);
        std::iota(order.begin(), order.end(), 0u);
        std::stable_sort(order.begin(), order.end(), [&](u32 a, u32 b) { return pairMap[a].size() > pairMap[b].size(); });

        std::vector<std::vector<Match> > mainMatches(n);
        std::vector<u8> mainMinErrors(n, 255);
        std::vector<std::vector<CigarOp> > collectedCigars(n);
        std::vector<bool> mainPaired(n, false);
        std::vector<double> mainProbs(n, 0.0);

        for (u32 binNo : order)
        {
            const std::vector<u32> & pids = pairMap[binNo];
            const size_t Pc = pids.size();
            if (!Pc) continue;
            const Bin & bin = idx.bins[binNo];
            auto origOf = [&](size_t r) { return r < Pc ? (size_t)pids[r] : (size_t)pids[r - Pc] + P; };
            struct Child { std::vector<Match> all; size_t nu = 0; std::vector<size_t> byErr; size_t nOpt = 0, nSet = 0; size_t primary = (size_t)-1;
                           bool mapped = false, paired = false; double prob = 0; bool rescued = false; Match mate; std::vector<CigarOp> cigar; };
            std::vector<Child> ch(2 * Pc);
            // the child mapper's per-read stage has no cross-read state: reads go to opt.threads threads (pairing below stays sequential)
            BinMapStats bst;
            const unsigned Tc = (unsigned)std::min<size_t>(std::max(1u, opt.threads), std::max<size_t>(1, 2 * Pc / 64));
            std::vector<BinMapStats> tst(Tc);
            std::vector<std::thread> cth;
            for (unsigned t = 0; t < Tc; ++t) cth.emplace_back([&, t] {
            ReadBinMapper mapper(bin, opt);
            for (size_t r = t; r < 2 * Pc; r += Tc)
            {
                size_t orig = origOf(r);
                unsigned L = (unsigned)reads.fwd[orig].size();
                const u8 * seqs[2] = {reads.fwd[orig].data(), reads.rev[orig].data()};
                Child & c = ch[r];
                mapper.map(seqs, L, (u32)orig, c.all, stats ? &tst[t] : nullptr);
                if (c.all.empty()) continue;
                c.nu = removeDuplicates(c.all);
                c.byErr.resize(c.nu); std::iota(c.byErr.begin(), c.byErr.end(), (size_t)0);
                std::stable_sort(c.byErr.begin(), c.byErr.end(), [&](size_t a, size_t b) { return c.all[a].errors < c.all[b].errors; });
                unsigned strata = getReadStrata(opt, L);
                while (c.nOpt < c.nu && c.all[c.byErr[c.nOpt]].errors <= c.all[c.byErr[0]].errors) ++c.nOpt;
                while (c.nSet < c.nu && c.all[c.byErr[c.nSet]].errors <= c.all[c.byErr[0]].errors + strata) ++c.nSet;
                c.primary = c.byErr[(size_t)libcxxFirstUniform(c.nOpt)];
                c.mapped = true;
            }
            });
            for (auto & x : cth) x.join();
            for (auto & x : tst) { bst.seeds += x.seeds; bst.hits += x.hits; bst.located += x.located; bst.extended += x.extended; bst.matches += x.matches; }
            if (stats) { stats->bin.seeds += bst.seeds; stats->bin.hits += bst.hits; stats->bin.located += bst.located; stats->bin.extended += bst.extended; stats->bin.matches += bst.matches; }
            // ---- library mean / deviation from unique proper pairs (src/mapper.h:929-985)
            unsigned libraryLength = 0, libraryDev = 0;
            bool pairing = true;
            if (!opt.libraryLength || !opt.libraryDev)
            {
                std::vector<unsigned> lens;
                for (size_t j = 0; j < Pc; ++j)
                {
                    const Child & a = ch[j], & b = ch[j + Pc];
                    if (a.nOpt == 1 && b.nOpt == 1)
                    {
                        const Match & fm = a.all[a.byErr[0]], & sm = b.all[b.byErr[0]];
                        if (fm.contigId == sm.contigId && orientationProper(fm, sm)) lens.push_back(libraryLengthOf(fm, sm));
                    }
                }
                if (lens.empty()) pairing = false;
                else
                {
                    std::vector<unsigned> tmp(lens);
                    std::nth_element(tmp.begin(), tmp.begin() + tmp.size() / 2, tmp.end());
                    unsigned median = tmp[tmp.size() / 2];
                    std::vector<unsigned> kept;
                    for (unsigned v : lens) if (!((double)v > 6.0 * median)) kept.push_back(v);
                    if (kept.empty()) pairing = false;
                    else
                    {
                        unsigned sum = 0; for (unsigned v : kept) sum += v;
                        float mean = std::max(sum / static_cast<float>(kept.size()), 1.0f);
                        float sq = 0.0f;
                        for (unsigned v : kept) { float d = (float)v - mean; sq += d * d; }
                        float dev = std::max(std::sqrt(sq / static_cast<float>(kept.size())), 1.0f);
                        libraryLength = (unsigned)mean; libraryDev = (unsigned)dev;
                    }
                }
            }
            if (pairing)
            {
                if (opt.libraryLength) libraryLength = opt.libraryLength;
                if (opt.libraryDev) libraryDev = opt.libraryDev;
                // ---- enumerate feasible pairs (src/mapper.h:989-1061)
                for (size_t j = 0; j < Pc; ++j)
                {
                    Child & a = ch[j], & b = ch[j + Pc];
                    if (!a.nu || !b.nu) continue;
                    std::vector<Match> am(a.all.begin(), a.all.begin() + a.nu), bm(b.all.begin(), b.all.begin() + b.nu);
                    unsigned aLen = (unsigned)reads.fwd[origOf(j)].size(), bLen = (unsigned)reads.fwd[origOf(j + Pc)].size();
                    double aOpt = (double)((float)a.all[a.byErr[0]].errors / (float)aLen), bOpt = (double)((float)b.all[b.byErr[0]].errors / (float)bLen);
                    auto fa = findPrimaryMatch(am, bm, aOpt, bOpt, b.nOpt, b.nSet - b.nOpt, aLen, bLen, contigLenThunk, &bin, libraryLength, libraryDev);
                    auto fb = findPrimaryMatch(bm, am, bOpt, aOpt, a.nOpt, a.nSet - a.nOpt, bLen, aLen, contigLenThunk, &bin, libraryLength, libraryDev);
                    if (fa.first == (size_t)-1 || fb.first == (size_t)-1) continue;
                    a.primary = fa.first; b.primary = fb.first;
                    a.prob = fa.second; b.prob = fb.second;
                    a.paired = b.paired = true;
                }
            }
            // ---- verifyMatches: rescue the unmapped mate of a uniquely anchored read (src/mapper_verifier.h:141-273)
            if (opt.verifyMatches)
            {
                for (size_t r = 0; r < 2 * Pc; ++r)
                {
                    Child & anchorC = ch[r];
                    size_t mr = r < Pc ? r + Pc : r - Pc;
                    Child & mateC = ch[mr];
                    if (anchorC.nOpt != 1 || mateC.mapped) continue;
                    const Match & anchor = anchorC.all[anchorC.byErr[0]];
                    size_t mateOrig = origOf(mr);
                    bool mateRev = !anchor.isRev;                      // getMateSeqId: the mate is searched on the opposite strand
                    const std::vector<u8> & mateSeq = mateRev ? reads.rev[mateOrig] : reads.fwd[mateOrig];
                    u64 contigLength = bin.contigs.length(anchor.contigId);
                    u64 wb = 0, we = 0;
                    if (mateRev)
                    {   // --> ... mate
                        if (anchor.contigBegin + libraryLength > 3ull * libraryDev) wb = anchor.contigBegin + libraryLength - 3ull * libraryDev;
                        wb = std::min(wb, contigLength);
                        we = std::min<u64>(anchor.contigBegin + libraryLength + 3ull * libraryDev, contigLength);
                    }
                    else
                    {   // mate ... <--
                        if (anchor.contigEnd() > (u64)libraryLength + 3ull * libraryDev) wb = anchor.contigEnd() - libraryLength - 3ull * libraryDev;
                        if (anchor.contigEnd() + libraryDev > libraryLength) we = anchor.contigEnd() - libraryLength + 3ull * libraryDev;
                        we = std::min(we, contigLength);
                    }
                    if (we <= wb) continue;
                    int mb, me, errs;
                    unsigned L = (unsigned)mateSeq.size();
                    if (!verifyMate(bin.contigs.seq(anchor.contigId) + wb, (int)(we - wb), mateSeq.data(), (int)L, getReadErrors(opt, L), getReadIndels(opt, L), mb, me, errs))
                        continue;
                    mateC.rescued = true;
                    mateC.mate.readId = (u32)mateOrig; mateC.mate.contigId = anchor.contigId; mateC.mate.isRev = mateRev;
                    mateC.mate.contigBegin = wb + (u64)mb; mateC.mate.contigLen = (u32)(me - mb); mateC.mate.errors = (u32)errs;
                }
                for (size_t mr = 0; mr < 2 * Pc; ++mr)
                {
                    Child & mateC = ch[mr];
                    if (!mateC.rescued) continue;
                    Child & anchorC = ch[mr < Pc ? mr + Pc : mr - Pc];
                    mateC.mapped = true; mateC.paired = true; anchorC.paired = true;
                    const Match & ap = anchorC.all[anchorC.primary];
                    unsigned aLen = (unsigned)reads.fwd[origOf(mr < Pc ? mr + Pc : mr - Pc)].size();
                    double rate = (double)((float)ap.errors / (float)aLen);
                    anchorC.prob = peMatchProb(rate, rate, anchorC.nOpt, anchorC.nSet - anchorC.nOpt);
                    mateC.prob = anchorC.prob;
                }
            }
            // ---- alignMatches + copyMatches + copyCigars
            u32 coff = idx.contigOffsets[binNo];
            for (size_t r = 0; r < 2 * Pc; ++r)
            {
                Child & c = ch[r];
                size_t orig = origOf(r);
                unsigned L = (unsigned)reads.fwd[orig].size();
                Match * pm = nullptr;
                if (c.rescued) pm = &c.mate;
                else if (c.mapped) pm = &c.all[c.primary];
                if (pm)
                {
                    const u8 * rs = pm->isRev ? reads.rev[orig].data() : reads.fwd[orig].data();
                    int cb, ce;
                    c.cigar = alignMatch(bin.contigs.seq(pm->contigId) + pm->contigBegin, (int)pm->contigLen, rs, (int)L, (int)pm->errors, cb, ce);
                    if (!(pm->errors == 0 || (pm->errors == 1 && pm->contigLen == L))) { pm->contigBegin += (u64)cb; pm->contigLen = (u32)ce; }
                }
                for (Match m : c.all) { m.contigId += coff; m.readId = (u32)orig; mainMatches[orig].push_back(m); }
                if (c.rescued) { Match m = c.mate; m.contigId += coff; mainMatches[orig].push_back(m); }
                if (pm)
                {
                    if (pm->errors < mainMinErrors[orig]) mainMinErrors[orig] = (u8)pm->errors;
                    if (!mainPaired[orig] && c.paired) { mainPaired[orig] = true; mainProbs[orig] = c.prob; }
                    if (mainMinErrors[orig] == pm->errors) collectedCigars[orig] = c.cigar;
                }
            }
        }
        std::vector<ReadResult> res(n);
        for (size_t i = 0; i < n; ++i)
        {
            std::vector<Match> & all = mainMatches[i];
            size_t nu = removeDuplicates(all);
            std::vector<Match> uniq(all.begin(), all.begin() + nu);
            rankRead(uniq, getReadStrata(opt, (unsigned)reads.fwd[i].size()), res[i]);
            res[i].cigar = collectedCigars[i];
            res[i].paired = mainPaired[i]; res[i].pairProb = mainProbs[i];
        }
        return res;
    }

    // SAM records of a paired batch (src/mapper_writer.h:268-293,347-400,515-571); reads [0,P) first mates, [P,2P) second mates
    void writeRecordsPaired(std::ostream & os, const ReadBatch & reads, const std::vector<ReadResult> & res) const
    {
        const size_t n = reads.size(), P = n / 2;
        for (size_t i = 0; i < n; ++i)
        {
            const ReadResult & r = res[i];
            const size_t mate = i < P ? i + P : i - P;
            const ReadResult & mr = res[mate];
            std::string qname = reads.names[i];
            size_t cut = qname.find_first_of(" \t/");
            if (cut != std::string::npos) qname.resize(cut);
            bool rev = r.mapped && r.primary.isRev;
            const std::vector<u8> & s = rev ? reads.rev[i] : reads.fwd[i];
            std::string seq(s.size(), 'N'), qual(reads.qual[i]);
            for (size_t k = 0; k < s.size(); ++k) seq[k] = DNA5_CHARS[s[k]];
            if (rev) std::reverse(qual.begin(), qual.end());
            unsigned flag = 1 | (i < P ? 64 : 128);
            if (!mr.mapped) flag |= 8;
            std::string rname = "*", rnext = "*"; u64 pos = 0, pnext = 0; long long tlen = 0; unsigned mapq = 0; std::string cigar = "*";
            if (!r.mapped)
            {
                flag |= 4;
                if (mr.mapped)
                {
                    rname = idx.contigNames[mr.primary.contigId]; pos = mr.primary.contigBegin + 1;
                    rnext = "="; pnext = mr.primary.contigBegin + 1;
                    if (mr.primary.isRev) flag |= 32;
                }
                os << qname << "\t" << flag << "\t" << rname << "\t" << pos << "\t0\t*\t" << rnext << "\t" << pnext << "\t0\t" << seq << "\t" << qual
                   << "\tRG:Z:" << opt.readGroup << "\n";
                continue;
            }
            if (rev) flag |= 16;
            rname = idx.contigNames[r.primary.contigId]; pos = r.primary.contigBegin + 1; cigar = cigarToString(r.cigar);
            if (mr.mapped)
            {
                rnext = mr.primary.contigId == r.primary.contigId ? "=" : idx.contigNames[mr.primary.contigId];
                pnext = mr.primary.contigBegin + 1;
                if (mr.primary.isRev) flag |= 32;
                if (r.paired)
                {
                    if (orientationProper(r.primary, mr.primary)) flag |= 2;
                    if (r.primary.contigId == mr.primary.contigId)
                        tlen = r.primary.contigBegin < mr.primary.contigBegin ? (long long)mr.primary.contigEnd() - (long long)r.primary.contigBegin
                                                                                : (long long)mr.primary.contigBegin - (long long)r.primary.contigEnd();
                }
            }
            else { rnext = "="; pnext = r.primary.contigBegin + 1; }
            if (r.paired) mapq = getMapq(r.pairProb);
            else { double errorRate = (double)((float)r.primary.errors / (float)s.size()); mapq = getMapq(getMatchProb(errorRate, errorRate, r.bestCount, r.subCount)); }
            os << qname << "\t" << flag << "\t" << rname << "\t" << pos << "\t" << mapq << "\t" << cigar << "\t" << rnext << "\t" << pnext << "\t" << tlen
               << "\t" << seq << "\t" << qual << "\tNM:i:" << r.primary.errors << "\tX0:i:" << r.bestCount << "\tX1:i:" << r.subCount << "\tRG:Z:" << opt.readGroup;
            std::string xa; bool skipped = false;
            for (const Match & m : r.matches)
            {
                if (!skipped && dupBegin(m, r.primary)) { skipped = true; continue; }
                xa += idx.contigNames[m.contigId]; xa += ','; xa += std::to_string(m.contigBegin + 1); xa += ',';
                xa += std::to_string(m.contigEnd() + 1); xa += ','; xa += m.isRev ? '-' : '+'; xa += ','; xa += std::to_string(m.errors); xa += ';';
            }
            if (!xa.empty()) os << "\tXA:Z:" << xa;
            os << "\n";
        }
    }

    // ---- SAM ----
    void writeHeader(std::ostream & os) const
    {
        os << "@HD\tVN:1.4\tSO:unsorted\n";
        os << "@PG\tID:Yara\tPN:Yara\tVN:" << opt.version << "\tCL:" << opt.commandLine << "\n";
        os << "@RG\tID:" << opt.readGroup << "\tSM:" << opt.readGroup << "\tPG:Yara\n";
        for (size_t c = 0; c < idx.contigNames.size(); ++c)
            os << "@SQ\tSN:" << idx.contigNames[c] << "\tLN:" << idx.contigLengths[c] << "\n";
    }

    void writeRecords(std::ostream & os, const ReadBatch & reads, const std::vector<ReadResult> & res) const
    {
        for (size_t i = 0; i < reads.size(); ++i)
        {
            const ReadResult & r = res[i];
            std::string qname = reads.names[i];
            size_t cut = qname.find_first_of(" \t/");           // readUntil(IsSpace || IsSlash), src/mapper_writer.h:445-446
            if (cut != std::string::npos) qname.resize(cut);
            bool rev = r.mapped && r.primary.isRev;
            const std::vector<u8> & s = rev ? reads.rev[i] : reads.fwd[i];
            std::string seq(s.size(), 'N'), qual(reads.qual[i]);
            for (size_t k = 0; k < s.size(); ++k) seq[k] = DNA5_CHARS[s[k]];
            if (rev) std::reverse(qual.begin(), qual.end());
            if (!r.mapped)
            {
                os << qname << "\t4\t*\t0\t0\t*\t*\t0\t0\t" << seq << "\t" << qual << "\tRG:Z:" << opt.readGroup << "\n";
                continue;
            }
            double errorRate = (double)((float)r.primary.errors / (float)s.size());
            unsigned mapq = getMapq(getMatchProb(errorRate, errorRate, r.bestCount, r.subCount));
            os << qname << "\t" << (rev ? 16 : 0) << "\t" << idx.contigNames[r.primary.contigId] << "\t" << (r.primary.contigBegin + 1)
               << "\t" << mapq << "\t" << cigarToString(r.cigar) << "\t*\t0\t0\t" << seq << "\t" << qual
               << "\tNM:i:" << r.primary.errors << "\tX0:i:" << r.bestCount << "\tX1:i:" << r.subCount << "\tRG:Z:" << opt.readGroup;
            // XA (TAG mode is the only reachable one): every non-primary entry of matchesSet
            std::string xa;
            bool skipped = false;
            for (const Match & m : r.matches)
            {
                if (!skipped && dupBegin(m, r.primary)) { skipped = true; continue; }     // findMatch :787-800
                xa += idx.contigNames[m.contigId]; xa += ',';
                xa += std::to_string(m.contigBegin + 1); xa += ',';
                xa += std::to_string(m.contigEnd() + 1); xa += ',';
                xa += m.isRev ? '-' : '+'; xa += ',';
                xa += std::to_string(m.errors); xa += ';';
            }
            if (!xa.empty()) os << "\tXA:Z:" << xa;
            os << "\n";
        }
    }

private:
    const DreamIndex & idx;
    const IBF * filter;
    const Options & opt;
};

} // namespace dyo

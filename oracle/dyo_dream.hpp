// ORACLE — TEST INFRASTRUCTURE ONLY (see dyo_formats.hpp header).
// CPU restatement of the DREAM driver: IBF classification, bin loop, merge, ranking and SAM text.
//   runDisMapper / clasifyLoadedReads / loadFilteredReads   reference src/d_mapper.h:309-443,860-890
//   copyMatches / copyCigars / rankMatches2 / finalize       reference src/d_mapper.h:127-245,610-721,828-838
//   rankMatches (child)                                      reference src/mapper.h:826-925
//   SAM records                                              reference src/mapper_writer.h:124-215,221-225,296-345,593-634
#pragma once
#include "dyo_mapper.hpp"
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

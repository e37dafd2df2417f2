// ORACLE — TEST INFRASTRUCTURE ONLY. C API over the CPU restatement so that tests/ (ctypes) can compare the device
// path with it on the same inputs.  Record layouts deliberately mirror include/dyhip.h (dy_match, dy_read_result).
#include "dyo_dream.hpp"

using namespace dyo;

namespace {
struct OMatch { u32 contig; u32 begin; u16 length; u8 errors; u8 is_rev; };
struct ORead { OMatch primary; u32 mapped, best_count, sub_count, mapq, set_begin, cigar_begin, cigar_len, paired, reserved; double pair_prob; };
static_assert(sizeof(ORead) == 56, "layout must match the numpy dtype");
struct OResult { std::vector<ORead> reads; std::vector<OMatch> matches; std::vector<u32> cigars; DreamStats stats; };
thread_local std::string g_err;
OMatch conv(const Match & m) { OMatch o; o.contig = m.contigId; o.begin = (u32)m.contigBegin; o.length = (u16)m.contigLen; o.errors = (u8)m.errors; o.is_rev = m.isRev; return o; }
ReadBatch makeBatch(const u8 * seqs, const u64 * offs, u32 n)
{
    ReadBatch b;
    for (u32 i = 0; i < n; ++i)
    {
        size_t len = (size_t)(offs[i + 1] - offs[i]);
        std::vector<u8> f(seqs + offs[i], seqs + offs[i] + len), r(len);
        for (size_t k = 0; k < len; ++k) { u8 c = f[len - 1 - k]; r[k] = c < 4 ? (u8)(3 - c) : (u8)4; }
        b.names.push_back(std::to_string(i)); b.fwd.push_back(f); b.rev.push_back(r); b.qual.emplace_back(len, 'I');
    }
    return b;
}
}

extern "C" {

const char * dyo_last_error() { return g_err.c_str(); }

void * dyo_ibf_load(const char * path)
{
    try { IBF * f = new IBF; f->load(path); return f; } catch (std::exception const & e) { g_err = e.what(); return nullptr; }
}
// read-only view of filter words already in memory (tests: a filter copied back from the device); the caller keeps them alive
void * dyo_ibf_attach(const u64 * words, u64 nBits, int direct)
{
    try { IBF * f = new IBF; f->attach(words, nBits, direct != 0); return f; } catch (std::exception const & e) { g_err = e.what(); return nullptr; }
}
void * dyo_kdx_load(const char * path)
{
    try { IBF * f = new IBF; f->load(path, true); return f; } catch (std::exception const & e) { g_err = e.what(); return nullptr; }
}
int dyo_kdx_build(const char * path, u32 bins, u32 k, const u8 * seqs, const u64 * offs, u32 nSeqs)
{
    try
    {
        IBF f; f.createDirect(bins, k);
        for (u32 i = 0; i < nSeqs && i < bins; ++i)
        {
            size_t len = (size_t)(offs[i + 1] - offs[i]);
            if (len >= k) ibfAddKmers(f, seqs + offs[i], len, i);
        }
        f.save(path);
        return 0;
    }
    catch (std::exception const & e) { g_err = e.what(); return -1; }
}
void dyo_ibf_free(void * f) { delete (IBF *)f; }
void dyo_ibf_info(void * f, u64 * out) { IBF * i = (IBF *)f; out[0] = i->noOfBits; out[1] = i->noOfBins; out[2] = i->noOfHashFunc; out[3] = i->kmerSize; out[4] = i->binIntWidth; out[5] = i->noOfHashPos; }

// whichBins(fwd) | whichBins(revcomp) per read, W words each
static void ibfQueryRange(IBF & f, const u8 * seqs, const u64 * offs, u32 lo, u32 hi, u32 step, u16 threshold, u64 * masks)
{
    for (u32 i = lo; i < hi; i += step)
    {
        size_t len = (size_t)(offs[i + 1] - offs[i]);
        std::vector<u8> rc(len);
        for (size_t k = 0; k < len; ++k) { u8 c = seqs[offs[i] + len - 1 - k]; rc[k] = c < 4 ? (u8)(3 - c) : (u8)4; }
        std::vector<bool> sel(f.noOfBins, false);
        if (len >= f.kmerSize)
        {
            ibfWhichBins(f, sel, seqs + offs[i], len, threshold);
            ibfWhichBins(f, sel, rc.data(), len, threshold);
        }
        else if (threshold == 0) sel.assign(f.noOfBins, true);
        for (u32 w = 0; w < f.binIntWidth; ++w) masks[(size_t)i * f.binIntWidth + w] = 0;
        for (u32 b = 0; b < f.noOfBins; ++b) if (sel[b]) masks[(size_t)i * f.binIntWidth + b / 64] |= 1ull << (b % 64);
    }
}
int dyo_ibf_query_mt(void * fv, const u8 * seqs, const u64 * offs, u32 n, u16 threshold, u64 * masks, unsigned threads)
{
    try
    {
        IBF & f = *(IBF *)fv;
        const unsigned T = std::max(1u, threads);
        std::vector<std::thread> th;
        for (unsigned t = 1; t < T; ++t) th.emplace_back([&, t] { ibfQueryRange(f, seqs, offs, t, n, T, threshold, masks); });
        ibfQueryRange(f, seqs, offs, 0, n, T, threshold, masks);
        for (auto & x : th) x.join();
        return 0;
    }
    catch (std::exception const & e) { g_err = e.what(); return -1; }
}
int dyo_ibf_query(void * fv, const u8 * seqs, const u64 * offs, u32 n, u16 threshold, u64 * masks)
{
    return dyo_ibf_query_mt(fv, seqs, offs, n, threshold, masks, 1);
}

// build a filter file: one bin per sequence (codes), for tests
int dyo_ibf_build(const char * path, u32 bins, u32 hashes, u32 k, u64 bitsWithoutTrailer, const u8 * seqs, const u64 * offs, u32 nSeqs)
{
    try
    {
        IBF f; f.create(bins, hashes, k, bitsWithoutTrailer + 256);
        for (u32 i = 0; i < nSeqs && i < bins; ++i)
        {
            size_t len = (size_t)(offs[i + 1] - offs[i]);
            if (len >= k) ibfAddKmers(f, seqs + offs[i], len, i);
        }
        f.save(path);
        return 0;
    }
    catch (std::exception const & e) { g_err = e.what(); return -1; }
}

// alignMatch on raw codes (tests of the device's DP shortcuts): returns the number of CIGAR ops written as (len << 4) | op
int dyo_align_match(const u8 * contig, int n, const u8 * read, int m, int errors, u32 * cigarOut, int cap, int * clipBegin, int * clipEnd)
{
    std::vector<CigarOp> c = alignMatch(contig, n, read, m, errors, *clipBegin, *clipEnd);
    int k = 0;
    for (auto & op : c) { if (k < cap) cigarOut[k] = (op.count << 4) | (op.op == 'M' ? 0u : op.op == 'I' ? 1u : 2u); ++k; }
    return k;
}

// Seed search of one bin on its own (tests of the search definitions): every occurrence reported for `seed` with at most maxErr
// errors — Hamming (edit = 0, reference Backtracking<HammingDistance>) or edit distance (edit = 1, -y full) — as (contig, END position
// in the forward contig, errors) triples.  The end is what the extension anchors on (reference src/mapper_extender.h:263-285).
// Returns the number of occurrences; at most `cap` are written.
u64 dyo_seed_occurrences(void * idxv, u32 bin, const u8 * seed, u32 len, u32 maxErr, int edit, u32 * out, u64 cap)
{
    DreamIndex & idx = *(DreamIndex *)idxv;
    const Bin & b = idx.bins[bin];
    std::vector<Hit> hits;
    if (edit && maxErr) fmSearchEdit(b.fm, seed, len, maxErr, 0, hits); else fmSearch(b.fm, seed, len, maxErr, 0, hits);
    u64 n = 0;
    for (auto & h : hits)
        for (u32 row = h.lo; row < h.hi; ++row)
        {
            u32 contig, x;
            b.fm.locate(row, contig, x);
            if (n < cap) { out[3 * n] = contig; out[3 * n + 1] = (u32)(b.contigs.length(contig) - x); out[3 * n + 2] = h.errors; }
            ++n;
        }
    return n;
}

void * dyo_index_load(const char * dir, u32 nBins)
{
    try { std::string d(dir); while (d.size() > 1 && d.back() == '/') d.pop_back(); return new DreamIndex(loadDreamIndex(d, nBins)); }
    catch (std::exception const & e) { g_err = e.what(); return nullptr; }
}
void dyo_index_free(void * p) { delete (DreamIndex *)p; }

// index one bin from codes (one contig per sequence) and write its 12 files under prefix
int dyo_index_build(const char * prefix, const u8 * seqs, const u64 * offs, u32 nContigs)
{
    try
    {
        std::vector<std::string> names; std::vector<std::vector<u8> > cs;
        for (u32 i = 0; i < nContigs; ++i) { names.push_back("contig" + std::to_string(i)); cs.emplace_back(seqs + offs[i], seqs + offs[i + 1]); }
        saveBinIndex(buildBinIndex(names, cs), prefix);
        return 0;
    }
    catch (std::exception const & e) { g_err = e.what(); return -1; }
}

static void * mapBatchImpl(void * idxv, void * ibfv, const u8 * seqs, const u64 * offs, u32 n, float errorRate, float strataRate, int sensitivity,
                           unsigned hitsThreshold, unsigned threads, bool paired, unsigned libraryLength, unsigned libraryDev, int verify, bool detMath = false)
{
    try
    {
        DreamIndex & idx = *(DreamIndex *)idxv;
        Options opt; opt.errorRate = errorRate; opt.strataRate = strataRate; opt.sensitivity = sensitivity; opt.hitsThreshold = hitsThreshold;
        opt.threads = threads; opt.useFilter = ibfv != nullptr;
        opt.libraryLength = libraryLength; opt.libraryDev = libraryDev; opt.verifyMatches = verify != 0; opt.detMath = detMath;
        ReadBatch batch = makeBatch(seqs, offs, n);
        DreamMapper mapper(idx, (IBF *)ibfv, opt);
        OResult * R = new OResult;
        std::vector<ReadResult> res = paired ? mapper.mapBatchPaired(batch, &R->stats) : mapper.mapBatch(batch, &R->stats);
        R->reads.resize(n);
        for (u32 i = 0; i < n; ++i)
        {
            ORead & o = R->reads[i];
            const ReadResult & r = res[i];
            memset(&o, 0, sizeof(o));
            o.mapped = r.mapped; o.best_count = r.mapped ? r.bestCount : 0; o.sub_count = r.mapped ? r.subCount : 0;
            o.primary.errors = 127;
            o.set_begin = (u32)R->matches.size(); o.cigar_begin = (u32)R->cigars.size();
            o.paired = r.paired; o.pair_prob = r.pairProb;
            if (r.mapped)
            {
                o.primary = conv(r.primary);
                double rate = (double)((float)r.primary.errors / (float)batch.fwd[i].size());
                o.mapq = r.paired ? getMapq(r.pairProb) : getMapq(getMatchProb(rate, rate, r.bestCount, r.subCount));
                for (auto & m : r.matches) R->matches.push_back(conv(m));
                for (auto & c : r.cigar) R->cigars.push_back((c.count << 4) | (c.op == 'M' ? 0u : c.op == 'I' ? 1u : 2u));
                o.cigar_len = (u32)r.cigar.size();
            }
        }
        return R;
    }
    catch (std::exception const & e) { g_err = e.what(); return nullptr; }
}
void * dyo_map_batch(void * idxv, void * ibfv, const u8 * seqs, const u64 * offs, u32 n, float errorRate, float strataRate, int sensitivity,
                     unsigned hitsThreshold, unsigned threads)
{
    return mapBatchImpl(idxv, ibfv, seqs, offs, n, errorRate, strataRate, sensitivity, hitsThreshold, threads, false, 0, 0, 1);
}
// paired-end: reads [0,n/2) are the first mates, [n/2,n) the second mates
void * dyo_map_batch_paired(void * idxv, void * ibfv, const u8 * seqs, const u64 * offs, u32 n, float errorRate, float strataRate, int sensitivity,
                            unsigned hitsThreshold, unsigned libraryLength, unsigned libraryDev, int verify)
{
    return mapBatchImpl(idxv, ibfv, seqs, offs, n, errorRate, strataRate, sensitivity, hitsThreshold, 1, true, libraryLength, libraryDev, verify);
}
void * dyo_map_batch_paired_mt(void * idxv, void * ibfv, const u8 * seqs, const u64 * offs, u32 n, float errorRate, float strataRate, int sensitivity,
                               unsigned hitsThreshold, unsigned libraryLength, unsigned libraryDev, int verify, unsigned threads, int detMath)
{
    return mapBatchImpl(idxv, ibfv, seqs, offs, n, errorRate, strataRate, sensitivity, hitsThreshold, std::max(1u, threads), true, libraryLength, libraryDev, verify, detMath != 0);
}
void dyo_result_free(void * r) { delete (OResult *)r; }
u64 dyo_result_n_matches(void * r) { return ((OResult *)r)->matches.size(); }
u64 dyo_result_n_cigars(void * r) { return ((OResult *)r)->cigars.size(); }
void dyo_result_copy(void * rv, void * reads, void * matches, u32 * cigars)
{
    OResult * r = (OResult *)rv;
    memcpy(reads, r->reads.data(), r->reads.size() * sizeof(ORead));
    if (!r->matches.empty()) memcpy(matches, r->matches.data(), r->matches.size() * sizeof(OMatch));
    if (!r->cigars.empty()) memcpy(cigars, r->cigars.data(), r->cigars.size() * 4);
}
void dyo_result_stats(void * rv, u64 * out)
{
    OResult * r = (OResult *)rv;
    out[0] = r->stats.filteredReads; out[1] = r->stats.bin.matches; out[2] = r->stats.bin.seeds; out[3] = r->stats.bin.hits; out[4] = r->stats.bin.located;
}

} // extern "C"

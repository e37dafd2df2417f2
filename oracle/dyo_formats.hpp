// ORACLE — TEST INFRASTRUCTURE ONLY.
// CPU restatement of DREAM-Yara's on-disk formats (reference at /root/reference, read-only).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, link or run
// anything under oracle/.  The product (dream_yara_amd/) never includes this file.
//
// Formats follow SURVEY.md Appendix A (verified against tests/golden/gold/64-viral-binned-indices):
//   <b>.txt.size / .txt.limits / .txt.concat / .rid.limits / .rid.concat   (reference src/store_seqs.h:128-158,
//                                                                          src/misc_options.h:293-331)
//   <b>.sa.len / .sa.ind / .sa.val / .lf.pst / .lf.drv / .lf.drs / .lf.drp (reference src/index_fm.h:68-103,
//                                                                          src/misc_types.h:151-188)
//   .filter = sdsl::bit_vector serialisation + 256-bit trailer              (reference src/d_bloom_filter.h:255-277)
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <stdexcept>
#include <fstream>
#include <algorithm>

namespace dyo {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

// ----------------------------------------------------------------------------
// Alphabet: Dna5 codes A=0 C=1 G=2 T=3 N=4 (everything non-ACGTU -> N)
// ----------------------------------------------------------------------------
inline u8 dna5Code(char c)
{
    switch (c)
    {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': case 'U': case 'u': return 3;
        default: return 4;
    }
}
static const char DNA5_CHARS[] = "ACGTN";

// N never equals anything, N included (reference src/basic_alphabet.h:47,142-146).
inline bool dnaEq(u8 a, u8 b) { return a == b && a < 4; }

inline std::vector<u8> readFileBytes(const std::string & path)
{
    FILE * f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<u8> buf((size_t)sz);
    if (sz && fread(buf.data(), 1, (size_t)sz, f) != (size_t)sz) { fclose(f); throw std::runtime_error("short read " + path); }
    fclose(f);
    return buf;
}

inline void writeFileBytes(const std::string & path, const void * p, size_t n)
{
    FILE * f = fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + path);
    if (n && fwrite(p, 1, n, f) != n) { fclose(f); throw std::runtime_error("short write " + path); }
    fclose(f);
}

// ----------------------------------------------------------------------------
// Contigs of one bin (real sequence, Ns kept).
// ----------------------------------------------------------------------------
struct Contigs
{
    std::vector<std::string> names;
    std::vector<u64>         limits;   // count+1 prefix sums
    std::vector<u8>          concat;   // Dna5 codes of all contigs back to back
    u64 maxLength = 0, count = 0, sum = 0;

    u64 length(u32 c) const { return limits[c + 1] - limits[c]; }
    const u8 * seq(u32 c) const { return concat.data() + limits[c]; }
};

// 3 bit/base, 21 bases per u64, base 0 in bits 62..60; word 0 of the file is the length (Appendix A.2).
inline void unpackDna5(const u8 * file, size_t fileBytes, std::vector<u8> & out)
{
    if (fileBytes < 8) throw std::runtime_error("txt.concat too short");
    u64 len; memcpy(&len, file, 8);
    out.resize(len);
    const u8 * p = file + 8;
    size_t words = (fileBytes - 8) / 8;
    if (words * 21 < len) throw std::runtime_error("txt.concat truncated");
    for (u64 i = 0; i < len; ++i)
    {
        u64 w; memcpy(&w, p + (i / 21) * 8, 8);
        out[i] = (u8)((w >> (60 - 3 * (i % 21))) & 7);
    }
}

inline std::vector<u8> packDna5(const std::vector<u8> & codes)
{
    u64 len = codes.size();
    size_t words = (len + 20) / 21;
    std::vector<u8> file(8 + words * 8, 0);
    memcpy(file.data(), &len, 8);
    for (size_t wi = 0; wi < words; ++wi)
    {
        u64 w = 0;
        for (unsigned k = 0; k < 21; ++k)
        {
            u64 i = (u64)wi * 21 + k;
            if (i >= len) break;
            w |= (u64)codes[i] << (60 - 3 * k);
        }
        memcpy(file.data() + 8 + wi * 8, &w, 8);
    }
    return file;
}

inline Contigs loadContigs(const std::string & prefix)
{
    Contigs c;
    std::vector<u8> sz = readFileBytes(prefix + ".txt.size");
    if (sz.size() != 24) throw std::runtime_error("bad txt.size");
    memcpy(&c.maxLength, sz.data(), 8); memcpy(&c.count, sz.data() + 8, 8); memcpy(&c.sum, sz.data() + 16, 8);
    std::vector<u8> lim = readFileBytes(prefix + ".txt.limits");
    c.limits.resize(lim.size() / 8);
    memcpy(c.limits.data(), lim.data(), c.limits.size() * 8);
    std::vector<u8> cc = readFileBytes(prefix + ".txt.concat");
    unpackDna5(cc.data(), cc.size(), c.concat);
    std::vector<u8> rl = readFileBytes(prefix + ".rid.limits");
    std::vector<u8> rc = readFileBytes(prefix + ".rid.concat");
    std::vector<u32> rlim(rl.size() / 4);
    memcpy(rlim.data(), rl.data(), rlim.size() * 4);
    for (size_t i = 0; i + 1 < rlim.size(); ++i)
        c.names.emplace_back((const char *)rc.data() + rlim[i], (const char *)rc.data() + rlim[i + 1]);
    if (c.limits.size() != c.count + 1 || c.names.size() != c.count) throw std::runtime_error("contig count mismatch in " + prefix);
    return c;
}

inline void saveContigs(const Contigs & c, const std::string & prefix)
{
    u64 sz[3] = {c.maxLength, c.count, c.sum};
    writeFileBytes(prefix + ".txt.size", sz, 24);
    writeFileBytes(prefix + ".txt.limits", c.limits.data(), c.limits.size() * 8);
    std::vector<u8> packed = packDna5(c.concat);
    writeFileBytes(prefix + ".txt.concat", packed.data(), packed.size());
    std::vector<u32> rl(1, 0);
    std::string rc;
    for (auto & n : c.names) { rc += n; rl.push_back((u32)rc.size()); }
    writeFileBytes(prefix + ".rid.limits", rl.data(), rl.size() * 4);
    writeFileBytes(prefix + ".rid.concat", rc.data(), rc.size());
}

// ----------------------------------------------------------------------------
// FM index of one bin over the reversed, N-randomised contigs (Appendix A.3).
// ----------------------------------------------------------------------------
struct RankBlock { u64 sym[2]; u32 cnt[4]; };          // 32 B per 64 BWT symbols
struct IndEntry  { u32 bits; u32 before; };            // 8 B per 32 rows

struct FMIndex
{
    u32 n = 0;                       // text length incl. sentinels
    u32 pst[5] = {0, 0, 0, 0, 0};    // C[A],C[C],C[G],C[T],n
    std::vector<RankBlock> drv;
    u8 drs = 0;                      // sentinel substitute symbol
    std::vector<u32> drp;            // BWT rows holding a sentinel (ascending)
    std::vector<IndEntry> ind;
    std::vector<u16> valContig;      // unpacked SA samples (u8 on disk for bins of up to 255 contigs, u16 above)
    std::vector<u32> valOffset;
    unsigned contigIdBytes = 1;      // 1 (u8 contigs) or 2 (u16 contigs), src/misc_types.h:184-188

    inline u8 sym(u32 row) const
    {
        const RankBlock & b = drv[row >> 6];
        unsigned t = row & 63;
        return (u8)((b.sym[t >> 5] >> ((31 - (t & 31)) * 2)) & 3);
    }

    // #c in bwt[0..pos)   (Appendix A.4)
    inline u32 rank(u8 c, u32 pos) const
    {
        const RankBlock & b = drv[pos >> 6];
        unsigned off = pos & 63;
        u32 r = b.cnt[c];
        static const u64 REP[4] = {0x0000000000000000ull, 0x5555555555555555ull, 0xAAAAAAAAAAAAAAAAull, 0xFFFFFFFFFFFFFFFFull};
        for (unsigned w = 0; w < 2 && off > 0; ++w)
        {
            unsigned k = off >= 32 ? 32 : off;
            u64 x = ~(b.sym[w] ^ REP[c]);
            u64 m = x & (x >> 1) & 0x5555555555555555ull;
            if (k < 32) m &= ~0ull << (64 - 2 * k);
            r += (u32)__builtin_popcountll(m);
            off -= k;
        }
        if (c == drs)
            r -= (u32)(std::lower_bound(drp.begin(), drp.end(), pos) - drp.begin());
        return r;
    }

    inline u32 lf(u32 pos, u8 c) const { return pst[c] + rank(c, pos); }

    inline bool isSentinelRow(u32 row) const { return std::binary_search(drp.begin(), drp.end(), row); }

    // sparse-SA lookup (SAMPLING = 10): returns (contig, offset in REVERSED contig)
    inline void locate(u32 row, u32 & contig, u32 & offset) const
    {
        u32 steps = 0;
        for (;;)
        {
            const IndEntry & e = ind[row >> 5];
            unsigned bit = 31 - (row & 31);
            if ((e.bits >> bit) & 1)
            {
                u32 idx = e.before + (u32)__builtin_popcount(bit == 31 ? 0u : (e.bits >> (bit + 1)));
                contig = valContig[idx];
                offset = valOffset[idx] + steps;
                return;
            }
            row = lf(row, sym(row));
            ++steps;
        }
    }
};

inline FMIndex loadFMIndex(const std::string & prefix, u64 contigCount)
{
    FMIndex fm;
    std::vector<u8> b;
    b = readFileBytes(prefix + ".sa.len"); if (b.size() != 4) throw std::runtime_error("bad sa.len"); memcpy(&fm.n, b.data(), 4);
    b = readFileBytes(prefix + ".lf.pst"); if (b.size() != 20) throw std::runtime_error("bad lf.pst"); memcpy(fm.pst, b.data(), 20);
    b = readFileBytes(prefix + ".lf.drv");
    fm.drv.resize(b.size() / 32); memcpy(fm.drv.data(), b.data(), fm.drv.size() * 32);
    // rank(.., n) with n % 64 == 0 needs one block past the end: synthesise it.
    if ((u64)fm.drv.size() * 64 <= fm.n)
    {
        RankBlock extra; extra.sym[0] = extra.sym[1] = 0;
        const RankBlock & last = fm.drv.back();
        for (unsigned c = 0; c < 4; ++c) extra.cnt[c] = last.cnt[c];
        for (unsigned t = 0; t < 64; ++t) extra.cnt[(last.sym[t >> 5] >> ((31 - (t & 31)) * 2)) & 3]++;
        fm.drv.push_back(extra);
    }
    b = readFileBytes(prefix + ".lf.drs"); if (b.size() != 1) throw std::runtime_error("bad lf.drs"); fm.drs = b[0];
    b = readFileBytes(prefix + ".lf.drp");
    // The indexer picks the contig-id type from the bin's OWN contig count (reference src/d_indexer.cpp:366-383): up to 255 contigs ->
    // uint8_t ids, sentinel rows as a plain position list (Naive dictionary, src/misc_types.h:165-167); up to 65535 -> uint16_t ids and
    // a Levels bool dictionary over the n rows — the very type of sa.ind (RankDictionary<bool, Levels<void, YaraFMConfig>>), hence the
    // same layout: 8 bytes per 32 rows, [u32 bits, row j at bit 31-j][u32 set bits before].  More than 65535 contigs per bin need the
    // reference's DR_YARA_LARGE_CONTIGS build.  PARITY UNPINNED: the gold bins hold one contig each.
    if (contigCount > 65535) throw std::runtime_error("Maximum number of contigs exceeded. Recompile with -DDR_YARA_LARGE_CONTIGS=ON.");
    if (contigCount <= 255)
    {
        if (b.size() != contigCount * 4) throw std::runtime_error("unexpected lf.drp size");
        fm.drp.resize(b.size() / 4); memcpy(fm.drp.data(), b.data(), fm.drp.size() * 4);
    }
    else
    {
        if (b.size() != ((size_t)fm.n + 31) / 32 * 8) throw std::runtime_error("unexpected lf.drp size (Levels form)");
        for (size_t e = 0; e * 8 < b.size(); ++e)
        {
            u32 bits; memcpy(&bits, b.data() + e * 8, 4);
            for (unsigned t = 0; t < 32; ++t) if ((bits >> (31 - t)) & 1) fm.drp.push_back((u32)(e * 32 + t));
        }
        if (fm.drp.size() != contigCount) throw std::runtime_error("lf.drp does not hold one sentinel per contig");
    }
    b = readFileBytes(prefix + ".sa.ind");
    fm.ind.resize(b.size() / 8); memcpy(fm.ind.data(), b.data(), fm.ind.size() * 8);
    b = readFileBytes(prefix + ".sa.val");
    // sa.val: packed Pair<TContigsSize, uint32_t> (src/misc_types.h:184-188): 5 bytes with u8 ids, 6 bytes with u16 ids
    fm.contigIdBytes = contigCount <= 255 ? 1 : 2;
    const size_t rec = fm.contigIdBytes + 4;
    if (b.size() % rec) throw std::runtime_error("unexpected sa.val record size");
    size_t cnt = b.size() / rec;
    fm.valContig.resize(cnt); fm.valOffset.resize(cnt);
    for (size_t i = 0; i < cnt; ++i)
    {
        u16 c = b[i * rec];
        if (fm.contigIdBytes == 2) c |= (u16)b[i * rec + 1] << 8;
        fm.valContig[i] = c;
        memcpy(&fm.valOffset[i], b.data() + i * rec + fm.contigIdBytes, 4);
    }
    return fm;
}

// ----------------------------------------------------------------------------
// Interleaved Bloom filter file (sdsl bit_vector: u64 bit count, then words; Appendix A.6)
// ----------------------------------------------------------------------------
struct IBF
{
    u64 noOfBits = 0;
    u32 noOfBins = 0;
    u32 noOfHashFunc = 0;
    u32 kmerSize = 0;
    u32 binIntWidth = 0;      // W
    u32 blockBitSize = 0;     // 64 W
    u64 noOfHashPos = 0;
    bool direct = false;      // k-mer direct-addressing filter (reference src/d_kdx_filter.h): position = the k-mer's own hash
    std::vector<u64> preCalc;
    std::vector<u64> words;   // ceil(bits/64)
    const u64 * view = nullptr; // read-only words owned by the caller (tests hand over a filter copied back from the device)
    const u64 * data() const { return view ? view : words.data(); }

    static const u64 SEED = 0x90b45d39fb6da1faull;   // src/d_bloom_filter.h:330
    static const unsigned SHIFT = 27;                // :329

    // reference src/d_bloom_filter.h:245-254
    void init()
    {
        binIntWidth = (noOfBins + 63) / 64;
        blockBitSize = binIntWidth * 64;
        noOfHashPos = (noOfBits - 256) / blockBitSize;
        // kmer_direct: the reference bounds the hash by (bits-256)/W (src/d_kdx_filter.h:215), 64x the number of blocks the
        // vector holds, i.e. it reads past the vector for larger hashes; here such k-mers (only those poisoned by a leading N,
        // SURVEY B.3) select no bin.
        if (direct) noOfHashFunc = 1;
        preCalc.resize(noOfHashFunc);
        for (u32 i = 0; i < noOfHashFunc; ++i) preCalc[i] = (u64)i ^ ((u64)kmerSize * SEED);
    }
    // bit i of the vector = word[i/64] >> (i%64) & 1; get_int(idx,64) on an unaligned idx spans two words.
    inline u64 getInt(u64 idx) const
    {
        u64 w = idx >> 6; unsigned o = idx & 63;
        const u64 * d = data();
        if (!o) return d[w];
        return (d[w] >> o) | (d[w + 1] << (64 - o));
    }
    // reference src/d_build_filter.cpp:305-312: 4^k positions of `bins` bits (+ trailer); defined for bins % 64 == 0
    void createDirect(u32 bins, u32 k)
    {
        direct = true;
        create(bins, 1, k, ((u64)1 << (2 * k)) * (((u64)bins + 63) / 64 * 64) + 256);
    }
    void create(u32 bins, u32 hashes, u32 k, u64 bits)
    {
        noOfBins = bins; noOfHashFunc = hashes; kmerSize = k; noOfBits = bits;
        words.assign((bits + 63) / 64, 0);
        init();
    }
    // reference src/d_bloom_filter.h:267-277
    void setMetadata()
    {
        u64 ms = noOfBits - 256;   // bits; 256-aligned iff noOfBits is (bs*2^33+256 is)
        if (ms % 64) throw std::runtime_error("unaligned IBF trailer not supported");
        words[ms / 64] = noOfBins; words[ms / 64 + 1] = direct ? 0 : noOfHashFunc; words[ms / 64 + 2] = kmerSize;     // kdx: src/d_kdx_filter.h:231-238
    }
    void save(const std::string & path)
    {
        setMetadata();
        FILE * f = fopen(path.c_str(), "wb");
        if (!f) throw std::runtime_error("cannot write " + path);
        fwrite(&noOfBits, 8, 1, f);
        fwrite(words.data(), 8, words.size(), f);
        fclose(f);
    }
    // a filter whose words already sit in memory (same layout as the file after its 8-byte bit count, trailer included)
    void attach(const u64 * w, u64 bits, bool asDirect = false)
    {
        direct = asDirect; view = w; noOfBits = bits;
        u64 ms = noOfBits - 256;
        noOfBins = (u32)getInt(ms); noOfHashFunc = (u32)getInt(ms + 64); kmerSize = (u32)getInt(ms + 128);
        init();
    }
    void load(const std::string & path, bool asDirect = false)
    {
        direct = asDirect;
        FILE * f = fopen(path.c_str(), "rb");
        if (!f) throw std::runtime_error("File \"" + path + "\" could not be read.");
        if (fread(&noOfBits, 8, 1, f) != 1) { fclose(f); throw std::runtime_error("bad filter"); }
        words.resize((noOfBits + 63) / 64);
        if (fread(words.data(), 8, words.size(), f) != words.size()) { fclose(f); throw std::runtime_error("truncated filter"); }
        fclose(f);
        u64 ms = noOfBits - 256;
        noOfBins = (u32)getInt(ms); noOfHashFunc = (u32)getInt(ms + 64); kmerSize = (u32)getInt(ms + 128);
        init();
    }
};

// ----------------------------------------------------------------------------
// FASTA / FASTQ reader (names keep the full header line minus the marker).
// ----------------------------------------------------------------------------
struct SeqRecords
{
    std::vector<std::string> names;
    std::vector<std::string> seqs;    // raw characters
    std::vector<std::string> quals;   // empty for FASTA
};

class SeqReader
{
public:
    explicit SeqReader(const std::string & path) : in(path)
    {
        if (!in) throw std::runtime_error("Error while opening reads file.");
        havePending = (bool)std::getline(in, pending);
    }
    // reads up to maxRecords records; returns number read
    size_t read(SeqRecords & out, size_t maxRecords)
    {
        size_t n = 0;
        while (n < maxRecords && havePending)
        {
            strip(pending);
            if (pending.empty()) { havePending = (bool)std::getline(in, pending); continue; }
            if (pending[0] == '>')
            {
                out.names.push_back(pending.substr(1));
                std::string s;
                while ((havePending = (bool)std::getline(in, pending)))
                {
                    strip(pending);
                    if (!pending.empty() && pending[0] == '>') break;
                    s += pending;
                }
                out.seqs.push_back(s); out.quals.emplace_back();
            }
            else if (pending[0] == '@')
            {
                out.names.push_back(pending.substr(1));
                std::string s, plus, q;
                std::getline(in, s); strip(s);
                std::getline(in, plus);
                std::getline(in, q); strip(q);
                out.seqs.push_back(s); out.quals.push_back(q);
                havePending = (bool)std::getline(in, pending);
            }
            else throw std::runtime_error("unrecognised sequence file format");
            ++n;
        }
        return n;
    }
private:
    static void strip(std::string & s) { while (!s.empty() && (s.back() == '\r' || s.back() == '\n' || s.back() == ' ')) s.pop_back(); }
    std::ifstream in;
    std::string pending;
    bool havePending = false;
};

} // namespace dyo

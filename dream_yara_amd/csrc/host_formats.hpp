// Host-side readers for the reference's on-disk inputs (no SeqAn / sdsl): the 12 per-bin index files,
// the .filter file and FASTA/FASTQ reads.  Everything is parsed straight into the flat arenas that are
// uploaded to HBM (see DESIGN.md "Data layout").  File formats: SURVEY.md Appendix A; producer side in the
// reference: src/index_fm.h:68-82, src/store_seqs.h:143-158, src/misc_options.h:293-307, src/d_bloom_filter.h:150-154.
#pragma once
#include "dy_types.h"
#include <string>
#include <vector>
#include <thread>
#include <algorithm>
#include <stdexcept>
#include <cstdio>
#include <cstring>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#ifdef DY_HAVE_ZLIB
#include <zlib.h>
#endif

namespace dy {

struct IoError : std::runtime_error { using std::runtime_error::runtime_error; };

// Read-only memory map of a whole file.
class MappedFile
{
public:
    explicit MappedFile(const std::string & path)
    {
        fd_ = ::open(path.c_str(), O_RDONLY);
        if (fd_ < 0) throw IoError("cannot open " + path);
        struct stat st;
        if (fstat(fd_, &st) != 0) { ::close(fd_); throw IoError("cannot stat " + path); }
        size_ = (size_t)st.st_size;
        if (size_)
        {
            ptr_ = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
            if (ptr_ == MAP_FAILED) { ::close(fd_); throw IoError("cannot map " + path); }
        }
    }
    ~MappedFile() { if (ptr_ && ptr_ != MAP_FAILED) munmap(ptr_, size_); if (fd_ >= 0) ::close(fd_); }
    MappedFile(const MappedFile &) = delete;
    MappedFile & operator=(const MappedFile &) = delete;
    const u8 * data() const { return (const u8 *)ptr_; }
    size_t size() const { return size_; }
private:
    int fd_ = -1; void * ptr_ = nullptr; size_t size_ = 0;
};

template <typename T> inline T loadLE(const u8 * p) { T v; memcpy(&v, p, sizeof(T)); return v; }

// Host image of the whole index (all bins), laid out exactly like the device arena.
struct HostIndex
{
    std::vector<BinDesc> bins;
    std::vector<u8>  rank;          // 32 B blocks
    std::vector<u8>  ind;           // 8 B entries
    std::vector<u32> valOffset;     // SA sample offsets
    std::vector<u16> valContig;     // SA sample contig ids
    std::vector<u8>  contigs;       // 1 B/base Dna5 codes, all bins back to back
    std::vector<u64> limits;        // per bin: nContigs+1 prefix sums
    std::vector<u32> drp;           // sentinel rows
    std::vector<std::string> contigNames;   // global order
    std::vector<u64> contigLengths;
};

// withFm = false: contigs and names only (a bin whose FM structures live on another rank), BinDesc::n stays 0
inline void appendBin(HostIndex & H, const std::string & prefix, bool withFm = true)
{
    BinDesc d; memset(&d, 0, sizeof(d));
    MappedFile size(prefix + ".txt.size");
    if (size.size() != 24) throw IoError("Error while opening reference file.");
    u64 nContigs = loadLE<u64>(size.data() + 8), sum = loadLE<u64>(size.data() + 16);
    // more than 65535 contigs per bin need the reference's DR_YARA_LARGE_CONTIGS build (src/d_indexer.cpp:376-382)
    if (nContigs == 0) throw IoError("Error while opening reference file.");
    if (nContigs > 65535) throw IoError("Maximum number of contigs exceeded. Recompile with -DDR_YARA_LARGE_CONTIGS=ON. (" + prefix + ")");
    d.nContigs = (u32)nContigs;
    d.contigIdBase = (u32)H.contigNames.size();

    // contigs: txt.limits (u64 prefix sums) + txt.concat (word 0 = length, then 21 bases / u64, 3 bit each, MSB first)
    {
        MappedFile lim(prefix + ".txt.limits"), cc(prefix + ".txt.concat");
        if (lim.size() != (nContigs + 1) * 8 || cc.size() < 8) throw IoError("Error while opening reference file.");
        d.limOff = (u32)H.limits.size();
        for (u64 i = 0; i <= nContigs; ++i) H.limits.push_back(loadLE<u64>(lim.data() + i * 8));
        u64 len = loadLE<u64>(cc.data());
        if (len != sum || (cc.size() - 8) / 8 * 21 < len) throw IoError("txt.concat does not match txt.size in " + prefix);
        d.contigOff = H.contigs.size();
        H.contigs.resize(d.contigOff + len);
        u8 * out = H.contigs.data() + d.contigOff;
        const u8 * w = cc.data() + 8;
        for (u64 base = 0; base < len; base += 21, w += 8)
        {
            u64 word = loadLE<u64>(w);
            u64 lim21 = len - base < 21 ? len - base : 21;
            for (u64 k = 0; k < lim21; ++k) out[base + k] = (u8)((word >> (60 - 3 * k)) & 7);
        }
        for (u64 i = 0; i < nContigs; ++i) H.contigLengths.push_back(H.limits[d.limOff + i + 1] - H.limits[d.limOff + i]);
    }
    // names: rid.limits (u32) + rid.concat
    {
        MappedFile rl(prefix + ".rid.limits"), rc(prefix + ".rid.concat");
        if (rl.size() != (nContigs + 1) * 4) throw IoError("Error while opening reference file.");
        for (u64 i = 0; i < nContigs; ++i)
        {
            u32 a = loadLE<u32>(rl.data() + i * 4), b = loadLE<u32>(rl.data() + i * 4 + 4);
            if (b < a || b > rc.size()) throw IoError("corrupt rid.limits in " + prefix);
            H.contigNames.emplace_back((const char *)rc.data() + a, (const char *)rc.data() + b);
        }
    }
    // FM index
    if (withFm)
    {
        MappedFile len(prefix + ".sa.len"), pst(prefix + ".lf.pst"), drv(prefix + ".lf.drv"), drs(prefix + ".lf.drs"),
                   drp(prefix + ".lf.drp"), ind(prefix + ".sa.ind"), val(prefix + ".sa.val");
        if (len.size() != 4 || pst.size() != 20 || drs.size() != 1 || drv.size() % 32 || ind.size() % 8)
            throw IoError("Error while opening reference index file.");
        d.n = loadLE<u32>(len.data());
        if (d.n != sum + nContigs) throw IoError("sa.len does not match txt.size in " + prefix);
        if (d.n >= 0xFFFE0000u) throw IoError("bins above 4.29 Gbp are not supported (" + prefix + ")");      // rows must stay below the DIRECT_* tags
        for (int c = 0; c < 4; ++c) d.pst[c] = loadLE<u32>(pst.data() + 4 * c);
        d.drs = drs.data()[0];
        // The indexer types an index by the bin's OWN contig count (reference src/d_indexer.cpp:366-383): up to 255 contigs -> uint8_t ids
        // and the sentinel rows as a plain position list (Naive dictionary, src/misc_types.h:165-167); up to 65535 -> uint16_t ids and a
        // Levels bool dictionary over the rows — the type of sa.ind, hence its layout: 8 bytes per 32 rows, [u32 bits, row j at bit
        // 31-j][u32 set bits before].  Either way the device gets the ascending list of sentinel rows.  (The reference MAPPER types all
        // bins by the contig count summed over the bins, src/d_mapper.cpp:437-472, so it cannot read a mixed set its own indexer wrote;
        // here every bin is read the way it was written.)  Parity unpinned: the gold bins hold one contig each.
        d.drpOff = (u32)H.drp.size();
        if (nContigs <= 255)
        {
            if (drp.size() != nContigs * 4) throw IoError("unexpected lf.drp size in " + prefix);
            for (u64 i = 0; i < nContigs; ++i) H.drp.push_back(loadLE<u32>(drp.data() + 4 * i));
        }
        else
        {
            if (drp.size() != ((size_t)d.n + 31) / 32 * 8) throw IoError("unexpected lf.drp size in " + prefix);
            u64 found = 0;
            for (size_t e = 0; e * 8 < drp.size(); ++e)
            {
                const u32 bits = loadLE<u32>(drp.data() + e * 8);
                for (u32 t = 0; t < 32; ++t) if ((bits >> (31 - t)) & 1) { H.drp.push_back((u32)(e * 32 + t)); ++found; }
            }
            if (found != nContigs) throw IoError("lf.drp does not hold one sentinel per contig in " + prefix);
        }
        // rank blocks; rank(.., n) with n % 64 == 0 touches one block past the file: synthesise it
        d.rankOff = H.rank.size() / 32;
        size_t nblk = drv.size() / 32;
        if (nblk * 64 < d.n) throw IoError("lf.drv too short in " + prefix);
        H.rank.insert(H.rank.end(), drv.data(), drv.data() + drv.size());
        if (nblk * 64 <= d.n)
        {
            const u8 * last = drv.data() + (nblk - 1) * 32;
            u64 s0 = loadLE<u64>(last), s1 = loadLE<u64>(last + 8);
            u32 cnt[4];
            for (int c = 0; c < 4; ++c) cnt[c] = loadLE<u32>(last + 16 + 4 * c);
            for (int t = 0; t < 64; ++t) cnt[((t < 32 ? s0 : s1) >> ((31 - (t & 31)) * 2)) & 3]++;
            u8 extra[32]; memset(extra, 0, 32); memcpy(extra + 16, cnt, 16);
            H.rank.insert(H.rank.end(), extra, extra + 32);
        }
        d.indOff = H.ind.size() / 8;
        if (ind.size() / 8 * 32 < d.n) throw IoError("sa.ind too short in " + prefix);
        H.ind.insert(H.ind.end(), ind.data(), ind.data() + ind.size());
        // sa.val: packed Pair<contig id, u32 offset> (reference src/misc_types.h:184-188): 5 bytes with u8 ids, 6 bytes with u16 ids
        const size_t idBytes = nContigs <= 255 ? 1 : 2, rec = idBytes + 4;
        if (val.size() % rec) throw IoError("unexpected sa.val record size in " + prefix);
        d.valOff = H.valOffset.size();
        size_t nval = val.size() / rec;
        for (size_t i = 0; i < nval; ++i)
        {
            H.valContig.push_back(idBytes == 1 ? (u16)val.data()[i * rec] : loadLE<u16>(val.data() + i * rec));
            H.valOffset.push_back(loadLE<u32>(val.data() + i * rec + idBytes));
        }
    }
    H.bins.push_back(d);
}

// .filter file = sdsl::bit_vector: u64 bit count, then ceil(bits/64) words.
struct FilterFile
{
    MappedFile map;
    u64 nBits = 0;
    const u8 * words = nullptr;
    size_t nWords = 0;
    explicit FilterFile(const std::string & path) : map(path)
    {
        if (map.size() < 8 + 32) throw IoError("File \"" + path + "\" could not be read.");
        nBits = loadLE<u64>(map.data());
        nWords = (size_t)((nBits + 63) / 64);
        if (nBits < 256 || map.size() < 8 + nWords * 8) throw IoError("File \"" + path + "\" could not be read.");
        words = map.data() + 8;
    }
    u64 getInt(u64 bit) const   // sdsl get_int(bit, 64)
    {
        u64 w = bit >> 6; unsigned o = bit & 63;
        u64 lo = loadLE<u64>(words + w * 8);
        if (!o) return lo;
        u64 hi = w + 1 < nWords ? loadLE<u64>(words + (w + 1) * 8) : 0;
        return (lo >> o) | (hi << (64 - o));
    }
};

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

// Streaming FASTA/FASTQ reader producing SoA batches (codes + offsets) ready for dy_reads_upload.
struct ReadsBatch
{
    std::vector<std::string> names;
    std::vector<u8>  codes;     // forward reads back to back
    std::vector<u64> offs;      // n+1
    std::vector<char> quals;    // same layout as codes; 'I' for FASTA, '!' under N
    size_t size() const { return names.size(); }
    void clear() { names.clear(); codes.clear(); offs.assign(1, 0); quals.clear(); }
};

class ReadsFile
{
public:
    // plain or gzip-compressed FASTA/FASTQ (reference src/d_mapper.cpp:120-138 accepts .gz through SeqAn's stream layer);
    // a compressed file is inflated into memory once
    explicit ReadsFile(const std::string & path) : map_(path), p_(map_.data()), end_(map_.data() + map_.size())
    {
        if (map_.size() >= 2 && map_.data()[0] == 0x1f && map_.data()[1] == 0x8b)
        {
#ifdef DY_HAVE_ZLIB
            gzFile f = gzopen(path.c_str(), "rb");
            if (!f) throw IoError("Error while opening reads file.");
            std::vector<u8> chunk(1 << 22);
            for (;;)
            {
                int n = gzread(f, chunk.data(), (unsigned)chunk.size());
                if (n < 0) { gzclose(f); throw IoError("corrupt gzip stream in " + path); }
                if (n == 0) break;
                inflated_.insert(inflated_.end(), chunk.begin(), chunk.begin() + n);
            }
            gzclose(f);
            p_ = inflated_.data(); end_ = inflated_.data() + inflated_.size();
#else
            throw IoError("gzip-compressed input needs a build with zlib");
#endif
        }
        else if (map_.size() >= 3 && map_.data()[0] == 'B' && map_.data()[1] == 'Z' && map_.data()[2] == 'h')
            throw IoError("bzip2-compressed input is not supported (no libbz2 on this system)");
    }
    // Reads up to maxReads records into b; returns their number.  Two phases over byte spans of the mapped file (no temporaries):
    // a sequential pass finds the record boundaries (one memchr per line) and the offsets, then `threads` workers convert disjoint
    // record ranges (256-entry table for the bases) into the pre-sized arrays.
    size_t next(ReadsBatch & b, size_t maxReads, unsigned threads = 1)
    {
        b.clear();
        spans_.clear();
        size_t bases = 0;
        while (spans_.size() < maxReads)
        {
            skipBlank();
            if (p_ >= end_) break;
            const char marker = (char)*p_;
            if (marker != '>' && marker != '@') throw IoError("unrecognised reads file format");
            RecSpan r;
            const u8 * s, * e;
            span(s, e);
            r.nameS = s + 1; r.nameE = e;
            r.fasta = marker == '>';
            r.len = 0;
            if (r.fasta)
            {   // the sequence may span several lines: remember the region, count the bases
                r.seqS = p_;
                while (p_ < end_ && *p_ != '>') { span(s, e); r.len += (size_t)(e - s); }
                r.seqE = p_;
                r.qualS = r.qualE = nullptr;
            }
            else
            {
                span(s, e); r.seqS = s; r.seqE = e; r.len = (size_t)(e - s);
                span(s, e);                               // '+'
                span(s, e); r.qualS = s; r.qualE = e;
            }
            bases += r.len;
            b.offs.push_back(bases);
            spans_.push_back(r);
        }
        const size_t n = spans_.size();
        b.names.resize(n); b.codes.resize(bases); b.quals.resize(bases);
        auto convert = [&](size_t lo, size_t hi) {
            static const Dna5Table lut;
            for (size_t i = lo; i < hi; ++i)
            {
                const RecSpan & r = spans_[i];
                b.names[i].assign((const char *)r.nameS, (const char *)r.nameE);
                u8 * codes = b.codes.data() + b.offs[i];
                char * quals = b.quals.data() + b.offs[i];
                if (r.fasta)
                {
                    size_t at = 0;
                    for (const u8 * q = r.seqS; q < r.seqE;)
                    {
                        const u8 * nl = (const u8 *)memchr(q, '\n', (size_t)(r.seqE - q));
                        const u8 * e = nl ? nl : r.seqE;
                        const u8 * nextLine = nl ? nl + 1 : r.seqE;
                        while (e > q && (e[-1] == '\r' || e[-1] == ' ')) --e;
                        for (const u8 * c = q; c < e; ++c) codes[at++] = lut.code[*c];
                        q = nextLine;
                    }
                    for (size_t k = 0; k < r.len; ++k) quals[k] = codes[k] == 4 ? '!' : 'I';
                }
                else
                {
                    const size_t ql = std::min(r.len, (size_t)(r.qualE - r.qualS));
                    for (size_t k = 0; k < r.len; ++k) codes[k] = lut.code[r.seqS[k]];
                    for (size_t k = 0; k < ql; ++k) quals[k] = (char)(r.qualS[k] > 33 + 62 ? 33 + 62 : r.qualS[k]);   // Dna5Q keeps 6 quality bits
                    for (size_t k = ql; k < r.len; ++k) quals[k] = 'I';
                    for (size_t k = 0; k < r.len; ++k) if (codes[k] == 4) quals[k] = '!';                               // N carries no quality
                }
            }
        };
        const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, n / 8192 + 1));
        if (T <= 1) convert(0, n);
        else
        {
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < T; ++t) pool.emplace_back(convert, n * t / T, n * (t + 1) / T);
            convert(0, n / T);
            for (auto & th : pool) th.join();
        }
        return n;
    }
private:
    struct Dna5Table { u8 code[256]; Dna5Table() { for (int c = 0; c < 256; ++c) code[c] = dna5Code((char)c); } };
    void skipBlank() { while (p_ < end_ && (*p_ == '\n' || *p_ == '\r')) ++p_; }
    // [s, e) = the next line without its terminator and trailing CR / blanks
    void span(const u8 *& s, const u8 *& e)
    {
        s = p_;
        const u8 * nl = (const u8 *)memchr(p_, '\n', (size_t)(end_ - p_));
        e = nl ? nl : end_;
        p_ = nl ? nl + 1 : end_;
        while (e > s && (e[-1] == '\r' || e[-1] == ' ')) --e;
    }
    struct RecSpan { const u8 * nameS, * nameE, * seqS, * seqE, * qualS, * qualE; size_t len; bool fasta; };
    std::vector<RecSpan> spans_;
    MappedFile map_;
    std::vector<u8> inflated_;
    const u8 * p_; const u8 * end_;
};

} // namespace dy

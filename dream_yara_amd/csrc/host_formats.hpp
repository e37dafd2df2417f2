// Host-side readers for the reference's on-disk inputs (no SeqAn / sdsl): the 12 per-bin index files,
// the .filter file and FASTA/FASTQ reads.  Everything is parsed straight into the flat arenas that are
// uploaded to HBM (see DESIGN.md "Data layout").  File formats: SURVEY.md Appendix A; producer side in the
// reference: src/index_fm.h:68-82, src/store_seqs.h:143-158, src/misc_options.h:293-307, src/d_bloom_filter.h:150-154.
#pragma once
#include <string_view>
#include <atomic>
#include <cerrno>
#include <chrono>
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
#include <dlfcn.h>
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
    // Maps the pages of [p, p + len) in one call (MADV_POPULATE_READ, Linux 5.14+; ignored where unknown): a first touch per 4 KB page
    // costs a sequential parser more than the parsing
    // (the page-table work of a 200 MB window takes longer than parsing it with a dozen threads: the window is cut into slices, one call each)
    void populate(const u8 * p, size_t len, unsigned threads = 1) const
    {
        if (!ptr_ || ptr_ == MAP_FAILED || !len || p < data() || p >= data() + size_) return;
        const size_t page = 4096;
        const size_t lo = (size_t)(p - data()) & ~(page - 1);
        size_t hi = (size_t)(p - data()) + len; if (hi > size_) hi = size_;
        const size_t nPages = (hi - lo + page - 1) / page;
        const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, nPages / 1024));
        if (T <= 1) { (void)madvise((u8 *)ptr_ + lo, hi - lo, 22 /* MADV_POPULATE_READ */); return; }
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < T; ++t)
        {
            const size_t a = lo + nPages * t / T * page, b = std::min(hi, lo + nPages * (t + 1) / T * page);
            if (b > a) pool.emplace_back([this, a, b] { (void)madvise((u8 *)ptr_ + a, b - a, 22 /* MADV_POPULATE_READ */); });
        }
        for (auto & th : pool) th.join();
    }
    // Copies file bytes [off, off + len) into dst with several threads (pread: page cache -> private memory, no page-table work)
    bool readInto(u8 * dst, size_t off, size_t len, unsigned threads) const
    {
        if (fd_ < 0 || off + len > size_) return false;
        const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, len / ((size_t)4 << 20) + 1));
        std::atomic<bool> ok(true);
        auto part = [&](unsigned t) {
            size_t lo = len * t / T; const size_t hi = len * (t + 1) / T;
            while (lo < hi)
            {
                const ssize_t r = ::pread(fd_, dst + lo, hi - lo, (off_t)(off + lo));
                if (r < 0 && errno == EINTR) continue;
                if (r <= 0) { ok = false; return; }
                lo += (size_t)r;
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < T; ++t) pool.emplace_back(part, t);
        part(0);
        for (auto & th : pool) th.join();
        return ok;
    }
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
// One bin's 12 files -> a HostIndex of its own (offsets relative to this bin; merged by loadHostIndex).
inline void parseBin(HostIndex & H, const std::string & prefix, bool withFm = true)
{
    BinDesc d; memset(&d, 0, sizeof(d));
    MappedFile size(prefix + ".txt.size");
    if (size.size() != 24) throw IoError("Error while opening reference file.");
    u64 nContigs = loadLE<u64>(size.data() + 8), sum = loadLE<u64>(size.data() + 16);
    // more than 65535 contigs per bin need the reference's DR_YARA_LARGE_CONTIGS build (src/d_indexer.cpp:376-382)
    if (nContigs == 0) throw IoError("Error while opening reference file.");
    if (nContigs > 65535) throw IoError("Maximum number of contigs exceeded. Recompile with -DDR_YARA_LARGE_CONTIGS=ON. (" + prefix + ")");
    d.nContigs = (u32)nContigs;
    d.contigIdBase = 0;

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

// All bins of an index directory, parsed by several threads (a bin's files are independent: unpacking 3-bit contigs and splitting the
// packed SA samples of 64 x 4 Mbp takes 0.4 s on one thread), then laid out back to back exactly like the device arena.
inline void loadHostIndex(HostIndex & H, const std::string & dir, u32 nBins, const uint8_t * owned, unsigned threads)
{
    std::vector<HostIndex> parts(nBins);
    std::vector<std::string> errors(nBins);
    std::atomic<u32> next(0);
    auto work = [&] {
        for (;;)
        {
            const u32 b = next.fetch_add(1);
            if (b >= nBins) break;
            try { parseBin(parts[b], dir + "/" + std::to_string(b), !owned || owned[b]); }
            catch (std::exception const & e) { errors[b] = e.what(); if (errors[b].empty()) errors[b] = "error"; }
        }
    };
    const unsigned T = std::max(1u, std::min<unsigned>(threads, nBins));
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < T; ++t) pool.emplace_back(work);
    work();
    for (auto & th : pool) th.join();
    for (u32 b = 0; b < nBins; ++b) if (!errors[b].empty()) throw IoError(errors[b]);
    // offsets of every bin in the merged arrays
    struct Off { size_t rank, ind, val, contigs, limits, drp, names; };
    std::vector<Off> off(nBins + 1);
    Off acc = {0, 0, 0, 0, 0, 0, 0};
    for (u32 b = 0; b < nBins; ++b)
    {
        off[b] = acc;
        const HostIndex & P = parts[b];
        acc.rank += P.rank.size(); acc.ind += P.ind.size(); acc.val += P.valOffset.size(); acc.contigs += P.contigs.size();
        acc.limits += P.limits.size(); acc.drp += P.drp.size(); acc.names += P.contigNames.size();
    }
    off[nBins] = acc;
    if (acc.limits > 0xFFFFFFFFull || acc.drp > 0xFFFFFFFFull || acc.names > 0xFFFFFFFFull) throw IoError("too many contigs in the index");
    H.rank.resize(acc.rank); H.ind.resize(acc.ind); H.valOffset.resize(acc.val); H.valContig.resize(acc.val); H.contigs.resize(acc.contigs);
    H.limits.resize(acc.limits); H.drp.resize(acc.drp); H.contigNames.resize(acc.names); H.contigLengths.resize(acc.names);
    H.bins.resize(nBins);
    next = 0;
    auto place = [&] {
        for (;;)
        {
            const u32 b = next.fetch_add(1);
            if (b >= nBins) break;
            HostIndex & P = parts[b];
            const Off & o = off[b];
            BinDesc d = P.bins[0];
            d.rankOff = o.rank / 32; d.indOff = o.ind / 8; d.valOff = o.val; d.contigOff = o.contigs; d.limOff = (u32)o.limits; d.drpOff = (u32)o.drp;
            d.contigIdBase = (u32)o.names;
            H.bins[b] = d;
            if (!P.rank.empty()) memcpy(H.rank.data() + o.rank, P.rank.data(), P.rank.size());
            if (!P.ind.empty()) memcpy(H.ind.data() + o.ind, P.ind.data(), P.ind.size());
            if (!P.valOffset.empty()) { memcpy(H.valOffset.data() + o.val, P.valOffset.data(), P.valOffset.size() * 4); memcpy(H.valContig.data() + o.val, P.valContig.data(), P.valContig.size() * 2); }
            if (!P.contigs.empty()) memcpy(H.contigs.data() + o.contigs, P.contigs.data(), P.contigs.size());
            std::copy(P.limits.begin(), P.limits.end(), H.limits.begin() + o.limits);
            std::copy(P.drp.begin(), P.drp.end(), H.drp.begin() + o.drp);
            for (size_t i = 0; i < P.contigNames.size(); ++i) { H.contigNames[o.names + i] = std::move(P.contigNames[i]); H.contigLengths[o.names + i] = P.contigLengths[i]; }
            P = HostIndex();
        }
    };
    pool.clear();
    for (unsigned t = 1; t < T; ++t) pool.emplace_back(place);
    place();
    for (auto & th : pool) th.join();
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
    std::vector<char> nameChars;   // read names back to back (one allocation per batch, not one per read)
    std::vector<u64> nameOffs;     // n+1
    std::vector<u8>  codes;        // forward reads back to back
    std::vector<u64> offs;         // n+1
    std::vector<char> quals;       // same layout as codes; 'I' for FASTA, '!' under N
    size_t size() const { return nameOffs.size() - 1; }
    std::string_view name(size_t i) const { return std::string_view(nameChars.data() + nameOffs[i], (size_t)(nameOffs[i + 1] - nameOffs[i])); }
    // (the byte arrays keep their size: the reader resizes them to the batch's exact totals, and a clear() + resize() would zero-fill
    // a few hundred MB per batch for nothing)
    void clear() { nameOffs.assign(1, 0); offs.assign(1, 0); }
    // the reads of `other` behind these (file pairs: first mates, then second mates)
    void append(const ReadsBatch & other)
    {
        const u64 nameShift = nameChars.size(), baseShift = codes.size();
        nameChars.insert(nameChars.end(), other.nameChars.begin(), other.nameChars.end());
        codes.insert(codes.end(), other.codes.begin(), other.codes.end());
        quals.insert(quals.end(), other.quals.begin(), other.quals.end());
        for (size_t i = 1; i < other.nameOffs.size(); ++i) nameOffs.push_back(other.nameOffs[i] + nameShift);
        for (size_t i = 1; i < other.offs.size(); ++i) offs.push_back(other.offs[i] + baseShift);
    }
};

// bzip2 streams through the system's libbz2.so.1.0, bound at run time (the image carries the library but not its header; the three entry
// points and bz_stream below are bzip2 1.0's documented, frozen ABI).  Concatenated streams (pbzip2) are decoded one after the other.
inline void bunzip2Into(const u8 * in, size_t nIn, std::vector<u8> & out, const std::string & path)
{
    struct BzStream
    {
        char * next_in; unsigned avail_in, total_in_lo32, total_in_hi32;
        char * next_out; unsigned avail_out, total_out_lo32, total_out_hi32;
        void * state; void * (*bzalloc)(void *, int, int); void (*bzfree)(void *, void *); void * opaque;
    };
    enum { BZ_OK = 0, BZ_STREAM_END = 4 };
    void * lib = dlopen("libbz2.so.1.0", RTLD_NOW);
    if (!lib) lib = dlopen("libbz2.so.1", RTLD_NOW);
    if (!lib) throw IoError("bzip2-compressed input needs libbz2.so.1.0 on this system");
    auto init = (int (*)(BzStream *, int, int))dlsym(lib, "BZ2_bzDecompressInit");
    auto step = (int (*)(BzStream *))dlsym(lib, "BZ2_bzDecompress");
    auto done = (int (*)(BzStream *))dlsym(lib, "BZ2_bzDecompressEnd");
    if (!init || !step || !done) { dlclose(lib); throw IoError("libbz2 lacks the BZ2_bzDecompress entry points"); }
    std::vector<u8> chunk(1 << 22);
    size_t at = 0;
    while (at < nIn)
    {
        BzStream z;
        memset(&z, 0, sizeof(z));
        if (init(&z, 0, 0) != BZ_OK) { dlclose(lib); throw IoError("libbz2 could not start a stream"); }
        int rc = BZ_OK;
        while (rc == BZ_OK)
        {
            if (z.avail_in == 0 && at < nIn)
            {
                const size_t take = std::min<size_t>(nIn - at, 1u << 30);
                z.next_in = (char *)in + at; z.avail_in = (unsigned)take; at += take;
            }
            z.next_out = (char *)chunk.data(); z.avail_out = (unsigned)chunk.size();
            const unsigned inBefore = z.avail_in;
            rc = step(&z);
            const size_t got = chunk.size() - z.avail_out;
            out.insert(out.end(), chunk.begin(), chunk.begin() + got);
            if (rc == BZ_OK && got == 0 && z.avail_in == inBefore && at >= nIn) rc = -7;      // truncated: no progress and no input left
        }
        at -= z.avail_in;                              // what the stream did not consume belongs to the next one
        done(&z);
        if (rc != BZ_STREAM_END) { dlclose(lib); throw IoError("corrupt bzip2 stream in " + path); }
    }
    dlclose(lib);
}

class ReadsFile
{
public:
    // plain, gzip- or bzip2-compressed FASTA/FASTQ (reference src/d_mapper.cpp:120-138 accepts .gz / .bz2 through SeqAn's stream layer);
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
        {
            bunzip2Into(map_.data(), map_.size(), inflated_, path);
            p_ = inflated_.data(); end_ = inflated_.data() + inflated_.size();
        }
    }
    // Reads up to maxReads records into b; returns their number.  Byte spans of the mapped file, no temporaries:
    //   * four-line FASTQ (the usual case) with several threads: the threads index the line ends of a window sized from the previous
    //     batch, every record is then lines 4r .. 4r+3 — checked ('@' / '+' markers) — and the threads convert disjoint record ranges;
    //     anything unexpected (blank lines, wrapped sequences, a window that holds too few lines) falls back to
    //   * the general form: one sequential pass finds the record boundaries (a memchr per line; wrapped FASTA, blank lines), then the
    //     threads convert.
    size_t next(ReadsBatch & b, size_t maxReads, unsigned threads = 1)
    {
        b.clear();
        nSpans_ = 0;                                   // (spans_ keeps its storage: re-creating a million spans would zero-fill 64 MB per batch)
        skipBlank();
        if (p_ >= end_ || maxReads == 0) { b.nameChars.clear(); b.codes.clear(); b.quals.clear(); return 0; }
        if (*p_ != '>' && *p_ != '@') throw IoError("unrecognised reads file format");
        auto tick = [] { return std::chrono::steady_clock::now(); };
        auto lap = [&](double & acc, std::chrono::steady_clock::time_point & t) { const auto n = tick(); acc += std::chrono::duration<double>(n - t).count(); t = n; };
        auto tp = tick();
        const bool threaded = threads > 1 && *p_ == '@' && bytesPerRecord_ > 0;
        if (!(threaded && copyWindow && inflated_.empty()))
            map_.populate(p_, bytesPerRecord_ > 0 ? (size_t)(bytesPerRecord_ * (double)maxReads * 1.05) + 65536 : (size_t)64 << 20, threads);
        lap(phase[0], tp);
        if (!(threads > 1 && *p_ == '@' && bytesPerRecord_ > 0 && indexFastqWindow(maxReads, threads)))
        {
            const u8 * first = p_;
            spans_.clear();
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
                spans_.push_back(r);
            }
            if (!spans_.empty()) bytesPerRecord_ = (double)(p_ - first) / (double)spans_.size();
            nSpans_ = spans_.size();
        }
        lap(phase[1], tp);
        const size_t n = nSpans_;
        b.offs.resize(n + 1); b.nameOffs.resize(n + 1);
        u64 bases = 0, nameBytes = 0;
        {   // offsets of the reads and of their names: per-thread partial sums, then every thread fills its range
            const unsigned TP = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, n / 65536 + 1));
            std::vector<u64> pb(TP + 1, 0), pn(TP + 1, 0);
            auto sums = [&](unsigned t) {
                u64 sb = 0, sn = 0;
                for (size_t i = n * t / TP; i < n * (t + 1) / TP; ++i) { sb += spans_[i].len; sn += (u64)(spans_[i].nameE - spans_[i].nameS); }
                pb[t + 1] = sb; pn[t + 1] = sn;
            };
            auto fill = [&](unsigned t) {
                u64 sb = pb[t], sn = pn[t];
                for (size_t i = n * t / TP; i < n * (t + 1) / TP; ++i) { b.offs[i] = sb; b.nameOffs[i] = sn; sb += spans_[i].len; sn += (u64)(spans_[i].nameE - spans_[i].nameS); }
            };
            auto runAll = [&](auto && f) {
                std::vector<std::thread> pool;
                for (unsigned t = 1; t < TP; ++t) pool.emplace_back(f, t);
                f(0u);
                for (auto & th : pool) th.join();
            };
            runAll(sums);
            for (unsigned t = 0; t < TP; ++t) { pb[t + 1] += pb[t]; pn[t + 1] += pn[t]; }
            runAll(fill);
            bases = pb[TP]; nameBytes = pn[TP];
        }
        b.offs[n] = bases; b.nameOffs[n] = nameBytes;
        b.nameChars.resize(nameBytes); b.codes.resize(bases); b.quals.resize(bases);
        lap(phase[2], tp);
        auto convert = [&](size_t lo, size_t hi) {
            static const Dna5Table lut;
            for (size_t i = lo; i < hi; ++i)
            {
                const RecSpan & r = spans_[i];
                memcpy(b.nameChars.data() + b.nameOffs[i], r.nameS, (size_t)(r.nameE - r.nameS));
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
                {   // one pass: code, quality clamped to 6 bits (Dna5Q), '!' under N (N carries no quality), 'I' where the quality line is short
                    const size_t ql = std::min(r.len, (size_t)(r.qualE - r.qualS));
                    for (size_t k = 0; k < ql; ++k)
                    {
                        const u8 c = lut.code[r.seqS[k]], q = r.qualS[k];
                        codes[k] = c;
                        quals[k] = c == 4 ? '!' : (char)(q > 33 + 62 ? 33 + 62 : q);
                    }
                    for (size_t k = ql; k < r.len; ++k) { const u8 c = lut.code[r.seqS[k]]; codes[k] = c; quals[k] = c == 4 ? '!' : 'I'; }
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
        lap(phase[3], tp);
        return n;
    }
    bool copyWindow = false;           // true: parse a private copy of each window made by parallel preads (measured: no gain)
    // for the device-side parser (dy_reads_from_fastq): is this an uncompressed file that starts like FASTQ, its size, raw reads, and where the
    // host parser would continue
    bool plainFastq() const { return inflated_.empty() && map_.size() > 0 && map_.data()[0] == '@'; }
    size_t fileSize() const { return map_.size(); }
    bool readRaw(u8 * dst, size_t off, size_t len, unsigned threads) const { return map_.readInto(dst, off, len, threads); }
    void seek(size_t off) { p_ = map_.data() + std::min(off, map_.size()); }
    double phase[4] = {0, 0, 0, 0};    // seconds spent in: page population, record boundaries, offsets + allocation, conversion (observability: -v)
private:
    struct Dna5Table { u8 code[256]; Dna5Table() { for (int c = 0; c < 256; ++c) code[c] = dna5Code((char)c); } };
    struct RecSpan { const u8 * nameS, * nameE, * seqS, * seqE, * qualS, * qualE; size_t len; bool fasta; };
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
    // Four-line FASTQ, threaded: fills spans_ with up to maxReads records starting at p_ and advances p_, or leaves both untouched and
    // returns false (the caller then takes the general path from the same position).
    bool indexFastqWindow(size_t maxReads, unsigned threads)
    {
        const size_t remaining = (size_t)(end_ - p_);
        size_t window = (size_t)(bytesPerRecord_ * (double)maxReads * 1.02) + 4096;
        for (int attempt = 0; attempt < 2; ++attempt, window += window / 3)      // records longer than the previous batch's: one wider try
            if (indexFastqWindow(maxReads, threads, std::min(window, remaining))) return true;
        return false;
    }
    bool indexFastqWindow(size_t maxReads, unsigned threads, size_t window)
    {
        const size_t remaining = (size_t)(end_ - p_);
        const u8 * base = p_;
        if (copyWindow && inflated_.empty())
        {   // the window as a private copy made by several threads: touching the mapped file's pages for the first time (one fault per 4 KB,
            // or one populate call) competes with the page-cache work of the output writer, a parallel pread does not
            if (window_.size() < window + 64) window_.resize(window + window / 8 + 64);
            if (map_.readInto(window_.data(), (size_t)(p_ - map_.data()), window, threads)) base = window_.data();
        }
        const u8 * const origin = p_;
        // line ends of the window, chunk by chunk
        const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, window / (1 << 20) + 1));
        if (ends_.size() < T) ends_.resize(T);                     // kept across batches: no allocation, no first touch of fresh pages per batch
        std::vector<std::vector<u32> > & ends = ends_;
        for (unsigned t = T; t < ends.size(); ++t) ends[t].clear();
        if (window > 0xFFFFFFF0ull) return false;
        auto scan = [&](unsigned t) {
            const size_t lo = window * t / T, hi = window * (t + 1) / T;
            std::vector<u32> & v = ends[t];
            v.clear();
            v.reserve((hi - lo) / 48 + 16);
            const u8 * q = base + lo, * stop = base + hi;
            while (q < stop)
            {
                const u8 * nl = (const u8 *)memchr(q, '\n', (size_t)(stop - q));
                if (!nl) break;
                v.push_back((u32)(nl - base));
                q = nl + 1;
            }
        };
        {
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < T; ++t) pool.emplace_back(scan, t);
            scan(0);
            for (auto & th : pool) th.join();
        }
        size_t nLines = 0;
        for (auto & v : ends) nLines += v.size();
        const bool atEof = window == remaining;
        const bool tail = atEof && window > 0 && base[window - 1] != '\n';       // a last line without its newline
        size_t nRec = (nLines + (tail ? 1 : 0)) / 4;
        if (nRec > maxReads) nRec = maxReads;
        if (nRec == 0 || (nRec < maxReads && !atEof)) return false;                 // the window came out too small: general path
        lineEnds_.resize(nRec * 4);
        {   // every scanner copies its own line ends into place
            std::vector<size_t> start(T + 1, 0);
            for (unsigned t = 0; t < T; ++t) start[t + 1] = start[t] + ends[t].size();
            const size_t want = nRec * 4;
            auto place = [&](unsigned t) {
                if (start[t] >= want) return;
                const size_t n = std::min(ends[t].size(), want - start[t]);
                if (n) memcpy(lineEnds_.data() + start[t], ends[t].data(), n * sizeof(u32));
            };
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < T; ++t) pool.emplace_back(place, t);
            place(0);
            for (auto & th : pool) th.join();
            size_t at = std::min(start[T], want);
            if (at < want) lineEnds_[at++] = (u32)window;                            // the unterminated last line
            if (at < want) return false;
        }
        if (spans_.size() < nRec) spans_.resize(nRec);
        std::atomic<bool> ok(true);
        auto build = [&](size_t lo, size_t hi) {
            for (size_t r = lo; r < hi; ++r)
            {
                const u8 * l0 = base + (r ? lineEnds_[4 * r - 1] + 1 : 0);
                const u8 * e0 = base + lineEnds_[4 * r], * e1 = base + lineEnds_[4 * r + 1], * e2 = base + lineEnds_[4 * r + 2], * e3 = base + lineEnds_[4 * r + 3];
                const u8 * l1 = e0 + 1, * l2 = e1 + 1, * l3 = e2 + 1;
                if (l0 >= e0 || *l0 != '@' || l2 > e2 || (l2 < e2 ? *l2 != '+' : true) || l1 > e1) { ok = false; return; }
                auto trim = [](const u8 * s, const u8 * e) { while (e > s && (e[-1] == '\r' || e[-1] == ' ')) --e; return e; };
                RecSpan & R = spans_[r];
                R.nameS = l0 + 1; R.nameE = trim(l0 + 1, e0);
                R.seqS = l1; R.seqE = trim(l1, e1); R.len = (size_t)(R.seqE - R.seqS);
                R.qualS = l3; R.qualE = trim(l3, e3);
                R.fasta = false;
                if (R.len == 0) { ok = false; return; }                              // a blank line where the sequence should be: not this layout
            }
        };
        {
            const unsigned TB = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, nRec / 8192 + 1));
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < TB; ++t) pool.emplace_back(build, nRec * t / TB, nRec * (t + 1) / TB);
            build(0, nRec / TB);
            for (auto & th : pool) th.join();
        }
        if (!ok) return false;
        const size_t used = (size_t)lineEnds_[nRec * 4 - 1] + 1;
        p_ = origin + (used < window ? used : window);
        bytesPerRecord_ = (double)(p_ - origin) / (double)nRec;
        nSpans_ = nRec;
        return true;
    }
    std::vector<RecSpan> spans_;
    size_t nSpans_ = 0;                // records of the current batch in spans_
    std::vector<u32> lineEnds_;
    std::vector<std::vector<u32> > ends_;
    std::vector<u8> window_;           // private copy of the window being parsed (copyWindow)
    double bytesPerRecord_ = 0;        // of the previous batch: sizes the window of the threaded FASTQ path
    MappedFile map_;
    std::vector<u8> inflated_;
    const u8 * p_; const u8 * end_;
};

} // namespace dy

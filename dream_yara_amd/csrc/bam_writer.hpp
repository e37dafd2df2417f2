// BAM (BGZF) output for dream_yara_mapper — the reference picks SAM or BAM by the output file's extension
// (reference src/d_mapper.cpp:140-153, src/mapper_writer.h via seqan::BamFileOut).  Host-side I/O only.
// Layout per the SAM/BAM specification v1: BGZF blocks (gzip members with a "BC" extra field, <= 64 KiB of payload,
// an empty block as EOF marker) around: magic, header text, reference dictionary, alignment records.
#pragma once
#include <charconv>
#include <zlib.h>
#include <cstdint>
#include <cstring>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>
#include <thread>
#include <algorithm>

namespace dy {

class BgzfWriter
{
public:
    explicit BgzfWriter(std::ostream & os, int level = -1) : os_(os), level_(level) { buf_.reserve(kBlock); }      // level: zlib's 0..9, -1 = its default (6)
    void write(const void * data, size_t n)
    {
        const uint8_t * p = (const uint8_t *)data;
        while (n)
        {
            size_t take = std::min(n, kBlock - buf_.size());
            buf_.insert(buf_.end(), p, p + take);
            p += take; n -= take;
            if (buf_.size() == kBlock) flushBlock();
        }
    }
    template <typename T> void put(T v) { write(&v, sizeof(T)); }      // little-endian host
    // Same stream as write(data, n), but the 64 KiB blocks are deflated by `threads` workers (BGZF blocks are independent).
    void writeParallel(const uint8_t * data, size_t n, unsigned threads)
    {
        if (!buf_.empty()) { size_t take = std::min(n, kBlock - buf_.size()); write(data, take); data += take; n -= take; }
        const size_t nBlocks = n / kBlock;
        if (nBlocks < 2 || threads < 2) { write(data, n); return; }
        std::vector<std::vector<uint8_t>> outBlocks(nBlocks);
        std::vector<std::thread> pool;
        const unsigned T = (unsigned)std::min<size_t>(threads, nBlocks);
        std::vector<std::string> errors(T);
        for (unsigned t = 0; t < T; ++t)
            pool.emplace_back([&, t] {
                try { for (size_t b = t; b < nBlocks; b += T) compressBlock(data + b * kBlock, kBlock, outBlocks[b], level_); }
                catch (std::exception const & e) { errors[t] = e.what(); }
            });
        for (auto & th : pool) th.join();
        for (auto & e : errors) if (!e.empty()) throw std::runtime_error(e);
        for (auto & ob : outBlocks) os_.write((const char *)ob.data(), (std::streamsize)ob.size());
        write(data + nBlocks * kBlock, n - nBlocks * kBlock);
    }
    void finish()
    {
        if (!buf_.empty()) flushBlock();
        static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        os_.write((const char *)eof, 28);
        os_.flush();
    }
private:
    static constexpr size_t kBlock = 0xff00;       // payload per block, leaves room for incompressible data
    static void compressBlock(const uint8_t * in, size_t n, std::vector<uint8_t> & out, int level)
    {
        out.resize(0x10000 + 64);
        z_stream zs; memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::runtime_error("zlib: deflateInit2 failed");
        zs.next_in = const_cast<uint8_t *>(in); zs.avail_in = (uInt)n;
        zs.next_out = out.data() + 18; zs.avail_out = (uInt)(out.size() - 18 - 8);
        if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { deflateEnd(&zs); throw std::runtime_error("zlib: deflate failed"); }
        const size_t clen = zs.total_out;
        deflateEnd(&zs);
        const size_t bsize = clen + 18 + 8;
        if (bsize > 0x10000) throw std::runtime_error("BGZF block too large");
        const uint8_t head[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, (uint8_t)((bsize - 1) & 0xff), (uint8_t)((bsize - 1) >> 8)};
        memcpy(out.data(), head, 18);
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), in, (uInt)n), isize = (uint32_t)n;
        memcpy(out.data() + 18 + clen, &crc, 4); memcpy(out.data() + 18 + clen + 4, &isize, 4);
        out.resize(bsize);
    }
    void flushBlock()
    {
        std::vector<uint8_t> out;
        compressBlock(buf_.data(), buf_.size(), out, level_);
        os_.write((const char *)out.data(), (std::streamsize)out.size());
        buf_.clear();
    }
    std::ostream & os_;
    int level_ = -1;
    std::vector<uint8_t> buf_;
};

// One alignment line, format-neutral; the SAM text and the BAM record are both produced from it.
struct AlnRecord
{
    std::string qname;
    uint32_t flag = 0;
    int32_t  rid = -1, pos = -1;            // 0-based
    uint32_t mapq = 0;
    const uint32_t * cigar = nullptr; uint32_t nCigar = 0;      // (len << 4) | op, op 0=M 1=I 2=D (BAM numbering)
    int32_t  rnext = -1, pnext = -1, tlen = 0;
    std::string seq, qual;                  // as printed in SAM (already oriented); qual in Phred+33
    bool hasCounts = false; uint32_t nm = 0, x0 = 0, x1 = 0;
    std::string rg, xa;
};

inline uint16_t bamReg2Bin(int64_t beg, int64_t end)
{
    --end;
    if (beg >> 14 == end >> 14) return (uint16_t)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (uint16_t)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (uint16_t)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (uint16_t)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (uint16_t)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

inline void bamWriteHeader(BgzfWriter & w, const std::string & text, const std::vector<std::string> & names, const std::vector<uint64_t> & lengths)
{
    w.write("BAM\1", 4);
    w.put<int32_t>((int32_t)text.size());
    w.write(text.data(), text.size());
    w.put<int32_t>((int32_t)names.size());
    for (size_t i = 0; i < names.size(); ++i)
    {
        w.put<int32_t>((int32_t)names[i].size() + 1);
        w.write(names[i].c_str(), names[i].size() + 1);
        w.put<int32_t>((int32_t)lengths[i]);
    }
}

// appends block_size + the record to `out`
inline void bamEncodeRecord(const AlnRecord & r, std::vector<uint8_t> & out)
{
    const uint32_t lSeq = (uint32_t)r.seq.size();
    const size_t total = 4 + 32 + (r.qname.size() + 1) + (size_t)r.nCigar * 4 + (lSeq + 1) / 2 + lSeq + (r.hasCounts ? 3 * 7 : 0) +
                         (3 + r.rg.size() + 1) + (r.xa.empty() ? 0 : 3 + r.xa.size() + 1);
    const size_t sizeAt = out.size();
    out.resize(sizeAt + total);                 // one growth per record, then plain stores
    uint8_t * w = out.data() + sizeAt;
    auto put = [&](const void * p, size_t n) { memcpy(w, p, n); w += n; };
    auto put32 = [&](uint32_t v) { put(&v, 4); };
    int64_t refLen = 0;
    for (uint32_t k = 0; k < r.nCigar; ++k) if ((r.cigar[k] & 15) != 1) refLen += r.cigar[k] >> 4;
    const int64_t end = r.pos < 0 ? 0 : (refLen ? r.pos + refLen : r.pos + 1);
    put32((uint32_t)(total - 4));
    put32((uint32_t)r.rid); put32((uint32_t)r.pos);
    const uint8_t lName = (uint8_t)(r.qname.size() + 1), mq = (uint8_t)r.mapq;
    const uint16_t bin = bamReg2Bin(r.pos, end), nOp = (uint16_t)r.nCigar, flag = (uint16_t)r.flag;
    put(&lName, 1); put(&mq, 1); put(&bin, 2); put(&nOp, 2); put(&flag, 2);
    put32(lSeq); put32((uint32_t)r.rnext); put32((uint32_t)r.pnext); put32((uint32_t)r.tlen);
    put(r.qname.c_str(), r.qname.size() + 1);
    if (r.nCigar) put(r.cigar, (size_t)r.nCigar * 4);
    auto code = [](char c) -> uint8_t { switch (c) { case 'A': return 1; case 'C': return 2; case 'G': return 4; case 'T': return 8; default: return 15; } };
    for (uint32_t i = 0; i < lSeq; i += 2)
    {
        uint8_t b = (uint8_t)(code(r.seq[i]) << 4);
        if (i + 1 < lSeq) b |= code(r.seq[i + 1]);
        *w++ = b;
    }
    if (r.qual.size() == lSeq) for (uint32_t i = 0; i < lSeq; ++i) *w++ = (uint8_t)(r.qual[i] - 33);
    else { memset(w, 0xff, lSeq); w += lSeq; }
    auto tagInt = [&](const char * t, uint32_t v) { put(t, 2); put("i", 1); put32(v); };      // typed 'i' like the reference (src/mapper_writer.h:177-191)
    auto tagStr = [&](const char * t, const std::string & v) { put(t, 2); put("Z", 1); put(v.c_str(), v.size() + 1); };
    if (r.hasCounts) { tagInt("NM", r.nm); tagInt("X0", r.x0); tagInt("X1", r.x1); }
    tagStr("RG", r.rg);
    if (!r.xa.empty()) tagStr("XA", r.xa);
}

inline void samWriteRecord(std::string & out, const AlnRecord & r, const std::vector<std::string> & names)
{
    auto num = [&](long long v) { char b[24]; const auto r = std::to_chars(b, b + sizeof b, v); out.append(b, (size_t)(r.ptr - b)); };
    out += r.qname; out += '\t'; num(r.flag); out += '\t';
    if (r.rid >= 0) out += names[(size_t)r.rid]; else out += '*';
    out += '\t'; num((long long)r.pos + 1); out += '\t'; num(r.mapq); out += '\t';
    if (r.nCigar) for (uint32_t k = 0; k < r.nCigar; ++k) { num(r.cigar[k] >> 4); out += "MID"[r.cigar[k] & 15]; }
    else out += '*';
    out += '\t';
    if (r.rnext < 0) out += '*'; else if (r.rnext == r.rid) out += '='; else out += names[(size_t)r.rnext];
    out += '\t'; num((long long)r.pnext + 1); out += '\t'; num(r.tlen); out += '\t';
    out += r.seq; out += '\t'; out += r.qual;
    if (r.hasCounts) { out += "\tNM:i:"; num(r.nm); out += "\tX0:i:"; num(r.x0); out += "\tX1:i:"; num(r.x1); }
    out += "\tRG:Z:"; out += r.rg;
    if (!r.xa.empty()) { out += "\tXA:Z:"; out += r.xa; }
    out += '\n';
}

} // namespace dy

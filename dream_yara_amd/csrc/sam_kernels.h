// Text in, text out on the device: four-line FASTQ -> reads batch, and per-read results -> SAM records.
//   FASTQ side replaces readRecords() of SeqAn's SeqFileIn as the mapper uses it      reference src/mapper.h:445-469 (loadReads), src/store_seqs.h
//   SAM side replaces MatchesWriter / writeMatches for single-end reads                reference src/mapper_writer.h:176-225, 296-345, 593-634
// Why on the device: the host share of a GPU box is a few cores; parsing (215 B per read) and formatting (275 B per read) cost them more time
// than the whole mapping costs the device.  Here the host only moves bytes: file -> pinned buffer -> device, device -> pinned buffer -> file.
// The records are byte-identical to what the host path (host_formats.hpp ReadsFile + bam_writer.hpp samWriteRecord) produces; anything the
// kernels do not handle the same way (blank lines, CR line ends, wrapped sequences, qualities of another length) raises a flag and the
// caller falls back to the host path.
#pragma once
#include <hip/hip_runtime.h>
#include "dy_types.h"
#include "finalize_kernels.h"       // ReadOut, OutMatch

namespace dy {

// per-byte equality flags (bit 7 of every byte that equals c), exact
__device__ __forceinline__ u64 byteEqFlags(u64 w, u8 c)
{
    const u64 x = w ^ (0x0101010101010101ull * c);
    return ~(((x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | x | 0x7F7F7F7F7F7F7F7Full);
}

struct FqStatus          // one per parsed window, fetched by the host in one copy
{
    u32 nLines, nRecords;
    u64 consumed;        // bytes of the window the records occupy
    u64 totalBases;
    u32 maxLen, minLen;
    u32 bad;             // not plain four-line FASTQ (see the reasons in k_fq_records)
    u32 pad;
};
struct FqRec { u32 nameStart, seqStart, qualStart; u16 nameLen /* up to the first blank, tab or slash */, len; };
static_assert(sizeof(FqRec) == 16, "record table stride");

enum { FQ_BLOCK_BYTES = 4096 };    // 256 threads x 16 bytes

// newline count of every 4 KB block (text is readable up to the next multiple of 16 bytes: the buffer is padded)
__global__ __launch_bounds__(256) void k_fq_count_lines(const u8 * __restrict__ text, u64 n, u32 * __restrict__ blockCount)
{
    __shared__ u32 waveSum[4];
    const u64 at = (u64)blockIdx.x * FQ_BLOCK_BYTES + (u64)threadIdx.x * 16;
    u32 c = 0;
    if (at < n)
    {
        const uint4 v = *(const uint4 *)(text + at);
        u64 lo = (u64)v.x | ((u64)v.y << 32), hi = (u64)v.z | ((u64)v.w << 32);
        u64 f0 = byteEqFlags(lo, '\n'), f1 = byteEqFlags(hi, '\n');
        if (at + 16 > n)
        {   // bytes beyond the window do not count
            const u32 valid = (u32)(n - at);
            if (valid <= 8) { f1 = 0; f0 &= valid == 8 ? ~0ull : ((1ull << (8 * valid)) - 1); }
            else f1 &= (1ull << (8 * (valid - 8))) - 1;
        }
        c = (u32)__popcll(f0) + (u32)__popcll(f1);
    }
    for (int o = 32; o >= 1; o >>= 1) c += (u32)__shfl_xor((int)c, o, 64);
    if ((threadIdx.x & 63) == 0) waveSum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blockCount[blockIdx.x] = waveSum[0] + waveSum[1] + waveSum[2] + waveSum[3];
}

// positions of the newlines, in order; lines beyond `cap` are not stored (the caller only needs the first 4 x maxReads)
__global__ __launch_bounds__(256) void k_fq_write_lines(const u8 * __restrict__ text, u64 n, const u32 * __restrict__ blockOff, u32 cap, u32 * __restrict__ lineEnds)
{
    __shared__ u32 waveSum[4];
    const u64 at = (u64)blockIdx.x * FQ_BLOCK_BYTES + (u64)threadIdx.x * 16;
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u64 f0 = 0, f1 = 0;
    if (at < n)
    {
        const uint4 v = *(const uint4 *)(text + at);
        const u64 lo = (u64)v.x | ((u64)v.y << 32), hi = (u64)v.z | ((u64)v.w << 32);
        f0 = byteEqFlags(lo, '\n'); f1 = byteEqFlags(hi, '\n');
        if (at + 16 > n)
        {
            const u32 valid = (u32)(n - at);
            if (valid <= 8) { f1 = 0; f0 &= valid == 8 ? ~0ull : ((1ull << (8 * valid)) - 1); }
            else f1 &= (1ull << (8 * (valid - 8))) - 1;
        }
    }
    const u32 c = (u32)__popcll(f0) + (u32)__popcll(f1);
    u32 incl = c;
    for (int o = 1; o < 64; o <<= 1) { const u32 t = (u32)__shfl_up((int)incl, o, 64); if (lane >= (u32)o) incl += t; }
    if (lane == 63) waveSum[wave] = incl;
    __syncthreads();
    u32 base = blockOff[blockIdx.x];
    for (u32 w = 0; w < wave; ++w) base += waveSum[w];
    u32 slot = base + incl - c;
    while (f0) { const int b = __ffsll((long long)f0) - 1; f0 &= f0 - 1; if (slot < cap) lineEnds[slot] = (u32)(at + (u32)(b >> 3)); ++slot; }
    while (f1) { const int b = __ffsll((long long)f1) - 1; f1 &= f1 - 1; if (slot < cap) lineEnds[slot] = (u32)(at + 8 + (u32)(b >> 3)); ++slot; }
}

// how many records the window holds (at most maxReads), and where they end
__global__ void k_fq_status(const u32 * __restrict__ blockOff, u32 nBlocks, const u8 * __restrict__ text, u64 n, u32 atEof, u32 maxReads, u32 cap,
                            u32 * __restrict__ lineEnds, FqStatus * __restrict__ st)
{
    u32 nLines = blockOff[nBlocks];
    if (atEof && n && text[n - 1] != '\n') { if (nLines < cap) lineEnds[nLines] = (u32)n; ++nLines; }       // a last line without its newline
    u32 nRec = nLines / 4;
    if (nRec > maxReads) nRec = maxReads;
    FqStatus s; s.nLines = nLines; s.nRecords = nRec; s.consumed = 0; s.totalBases = 0; s.maxLen = 0; s.minLen = 0xFFFFFFFFu; s.bad = 0; s.pad = 0;
    if (nRec) { const u64 e = lineEnds[4 * nRec - 1]; s.consumed = e < n ? e + 1 : n; }
    *st = s;
}

// One thread per record: its three spans, checked.  bad bits: 1 no '@', 2 no '+', 4 empty or over-long sequence (reference src/mapper.h:453-454),
// 8 quality line of another length, 16 CR / blank before a line end (the host parser trims those), 32 empty name line.
__global__ __launch_bounds__(256) void k_fq_records(const u8 * __restrict__ text, const u32 * __restrict__ lineEnds, FqStatus * __restrict__ st,
                                                    FqRec * __restrict__ recs, u64 * __restrict__ lensWide, u16 * __restrict__ lens16)
{
    __shared__ u32 sMax, sMin; __shared__ unsigned long long sSum; __shared__ u32 sBad;
    if (threadIdx.x == 0) { sMax = 0; sMin = 0xFFFFFFFFu; sSum = 0; sBad = 0; }
    __syncthreads();
    const u32 nRec = st->nRecords;
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r == nRec) lensWide[r] = 0;
    if (r < nRec)
    {
        const u32 s0 = r ? lineEnds[4 * r - 1] + 1 : 0;
        const u32 e0 = lineEnds[4 * r], e1 = lineEnds[4 * r + 1], e2 = lineEnds[4 * r + 2], e3 = lineEnds[4 * r + 3];
        const u32 s1 = e0 + 1, s2 = e1 + 1, s3 = e2 + 1;
        u32 bad = 0;
        if (s0 >= e0 || text[s0] != '@') bad |= 1;
        if (s2 >= e2 || text[s2] != '+') bad |= 2;
        const u32 len = e1 - s1;
        if (len == 0 || len > 8191) bad |= 4;
        if (e3 - s3 != len) bad |= 8;
        auto endsBlank = [&](u32 s, u32 e) { return e > s && (text[e - 1] == '\r' || text[e - 1] == ' '); };
        if (endsBlank(s0, e0) || endsBlank(s1, e1) || endsBlank(s2, e2) || endsBlank(s3, e3)) bad |= 16;
        if (e0 - s0 < 2) bad |= 32;
        u32 cut = 0;
        if (!bad) { const u32 nameLen = e0 - s0 - 1; while (cut < nameLen) { const u8 c = text[s0 + 1 + cut]; if (c == ' ' || c == '\t' || c == '/') break; ++cut; } if (cut > 0xFFFF) bad |= 32; }
        FqRec R; R.nameStart = s0 + 1; R.seqStart = s1; R.qualStart = s3; R.nameLen = (u16)cut; R.len = (u16)(bad ? 0 : len);
        recs[r] = R; lensWide[r] = bad ? 0 : len; lens16[r] = (u16)(bad ? 0 : len);
        if (bad) atomicOr(&sBad, bad);
        else { atomicMax(&sMax, len); atomicMin(&sMin, len); atomicAdd(&sSum, (unsigned long long)len); }
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        if (sBad) atomicOr(&st->bad, sBad);
        if (sSum) { atomicMax(&st->maxLen, sMax); atomicMin(&st->minLen, sMin); atomicAdd((unsigned long long *)&st->totalBases, sSum); }
    }
}

__device__ __forceinline__ u32 dna5CodeOf(u32 c)          // host_formats.hpp dna5Code
{
    const u32 u = c & 0xDFu;                               // upper case
    return u == 'A' ? 0u : u == 'C' ? 1u : u == 'G' ? 2u : (u == 'T' || u == 'U') ? 3u : 4u;
}

// Dna5 codes of the reads and their reverse complements (16 lanes per read, 8 bases per lane and step)
__global__ __launch_bounds__(256) void k_fq_codes(const u8 * __restrict__ text, const FqRec * __restrict__ recs, u32 nRec, const u64 * __restrict__ offs,
                                                  u8 * __restrict__ seqs, u8 * __restrict__ seqsRc)
{
    const u32 g = threadIdx.x & 15;
    const u32 group = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, nGroups = (gridDim.x * blockDim.x) >> 4;
    for (u32 r = group; r < nRec; r += nGroups)
    {
        const FqRec R = recs[r];
        const u8 * src = text + R.seqStart;
        const u64 o = offs[r];
        const u32 len = R.len;
        for (u32 p = g * 8; p < len; p += 128)
        {
            const u32 nb = len - p < 8 ? len - p : 8;
            u64 fw = 0;
            for (u32 k = 0; k < nb; ++k) fw |= (u64)dna5CodeOf(src[p + k]) << (8 * k);
            if (nb == 8)
            {
                ((PackedU64rc *)(seqs + o + p))->v = fw;
                ((PackedU64rc *)(seqsRc + o + len - 8 - p))->v = revcomp8(fw);
            }
            else for (u32 k = 0; k < nb; ++k)
            {
                const u8 c = (u8)(fw >> (8 * k));
                seqs[o + p + k] = c; seqsRc[o + len - 1 - (p + k)] = c < 4 ? (u8)(3 - c) : c;
            }
        }
    }
}

// ---- SAM ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 decDigits(u64 v) { u32 d = 1; while (v >= 10) { v /= 10; ++d; } return d; }
__device__ __forceinline__ u8 * putDec(u8 * p, u64 v)
{
    const u32 d = decDigits(v);
    for (u32 k = d; k-- > 0;) { p[k] = (u8)('0' + v % 10); v /= 10; }
    return p + d;
}
__device__ __forceinline__ u8 * putBytes(u8 * p, const u8 * s, u32 n) { for (u32 k = 0; k < n; ++k) p[k] = s[k]; return p + n; }

// MAPQ of single-end reads (reference src/bits_matches.h:898-949, src/mapper_writer.h:221-225; dyhip.hip mapqOf is the host form).  The host
// evaluates pow / log10 with libm, the device with its own routines (accurate to an ulp or two): the results can only differ where the value
// is within such a distance of a rounding boundary x.5 — those reads raise `uncertain` and the caller takes the host path for the batch.
__device__ __forceinline__ u32 mapqOnDevice(u32 errors, u32 readLen, u64 best, u64 sub, u32 * __restrict__ uncertain)
{
    const double rate = (double)((float)errors / (float)readLen);
    auto residual = [](double r) { const double p = pow(10.0, fmin(2.0, 100.0 * r - 7.0)); return p / (1.0 - p); };
    auto weight = [](double r, double opt) { return (1.0 - r) / pow(10.0, 300.0 * (r - opt)); };
    const double w = (double)best * weight(rate, rate) + (double)sub * weight(rate + 0.01, rate) + residual(rate);
    const double prob = weight(rate, rate) / w;
    const double x = -10.0 * log10(1.0 - fmin(prob, 0.9999999));
    const double fl = floor(x);
    if (!(x >= 0.0) || fabs(x - fl - 0.5) < 1e-6) atomicExch(uncertain, 1u);
    return (u32)round(x);
}

struct SamArgs
{
    const u8 * text; const FqRec * recs; const u64 * offs;     // the FASTQ window of the batch and its record table
    const ReadOut * out; const OutMatch * outSet; const u32 * cigars;
    const u8 * contigNames; const u32 * contigNameOffs;        // all contigs of the index, global ids
    const u8 * readGroup; u32 readGroupLen;
    u32 nReads;
};

// bytes of every record and its MAPQ (thread per read); sizes[nReads] = 0 for the scan
__global__ __launch_bounds__(256) void k_sam_sizes(SamArgs A, u32 * __restrict__ mapq, u64 * __restrict__ sizes, u32 * __restrict__ uncertain, u32 * __restrict__ mappedCount)
{
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r == A.nReads) sizes[r] = 0;
    if (r >= A.nReads) return;
    const FqRec R = A.recs[r];
    const ReadOut o = A.out[r];
    u64 n = (u64)R.nameLen + 1;
    if (!o.mapped)
    {   // qname 4 * 0 0 * * 0 0 seq qual RG
        n += 2 + 2 + 2 + 2 + 2 + 2 + 2 + 2;                                     // "4\t" "*\t" "0\t" "0\t" "*\t" "*\t" "0\t" "0\t"
        n += (u64)R.len * 2 + 1;
        n += 6 + A.readGroupLen + 1;                                              // "\tRG:Z:" rg "\n"
        mapq[r] = 0; sizes[r] = n;
        return;
    }
    const u32 q = mapqOnDevice(o.errors, R.len, o.bestCount, o.subCount, uncertain);
    mapq[r] = q;
    n += decDigits(o.isRev ? 16u : 0u) + 1;
    n += (A.contigNameOffs[o.contig + 1] - A.contigNameOffs[o.contig]) + 1;
    n += decDigits((u64)o.begin + 1) + 1 + decDigits(q) + 1;
    if (o.cigarLen) for (u32 k = 0; k < o.cigarLen; ++k) n += decDigits(A.cigars[o.cigarBegin + k] >> 4) + 1; else n += 1;
    n += 1 + 6;                                                                   // "\t" "*\t0\t0\t"
    n += (u64)R.len * 2 + 1;
    n += 6 + decDigits(o.errors) + 6 + decDigits(o.bestCount) + 6 + decDigits(o.subCount);
    n += 6 + A.readGroupLen;
    // XA: every entry of matchesSet except the primary (the first entry with the primary's contig / strand / begin)
    const u32 nset = o.bestCount + o.subCount;
    bool skipped = false; u64 xa = 0;
    for (u32 k = 0; k < nset; ++k)
    {
        const OutMatch m = A.outSet[(size_t)o.setBegin + k];
        if (!skipped && m.contig == o.contig && m.isRev == o.isRev && m.begin == o.begin) { skipped = true; continue; }
        xa += (A.contigNameOffs[m.contig + 1] - A.contigNameOffs[m.contig]) + 1 + decDigits((u64)m.begin + 1) + 1 + decDigits((u64)m.begin + m.length + 1) + 1 + 1 + 1 + decDigits(m.errors) + 1;
    }
    if (xa) n += 6 + xa;
    n += 1;
    sizes[r] = n;
    atomicAdd(mappedCount, 1u);
}

// the records (16 lanes per read: lane 0 writes the short fields, all lanes move name, bases and qualities)
__global__ __launch_bounds__(256) void k_sam_write(SamArgs A, const u32 * __restrict__ mapq, const u64 * __restrict__ recOff, u8 * __restrict__ sam)
{
    const u32 g = threadIdx.x & 15;
    const u32 group = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, nGroups = (gridDim.x * blockDim.x) >> 4;
    for (u32 r = group; r < A.nReads; r += nGroups)
    {
        const FqRec R = A.recs[r];
        const ReadOut o = A.out[r];
        u8 * p = sam + recOff[r];
        const u8 * name = A.text + R.nameStart;
        for (u32 k = g; k < R.nameLen; k += 16) p[k] = name[k];
        p += R.nameLen;
        // position of SEQ inside the record: everything in front of it is written by lane 0
        u32 head = 0;
        if (g == 0)
        {
            u8 * q = p;
            *q++ = '\t';
            if (!o.mapped) { const u8 s[] = {'4', '\t', '*', '\t', '0', '\t', '0', '\t', '*', '\t', '*', '\t', '0', '\t', '0', '\t'}; q = putBytes(q, s, 16); }
            else
            {
                q = putDec(q, o.isRev ? 16u : 0u); *q++ = '\t';
                q = putBytes(q, A.contigNames + A.contigNameOffs[o.contig], A.contigNameOffs[o.contig + 1] - A.contigNameOffs[o.contig]); *q++ = '\t';
                q = putDec(q, (u64)o.begin + 1); *q++ = '\t';
                q = putDec(q, mapq[r]); *q++ = '\t';
                if (o.cigarLen) for (u32 k = 0; k < o.cigarLen; ++k) { const u32 c = A.cigars[o.cigarBegin + k]; q = putDec(q, c >> 4); *q++ = (u8)("MID"[c & 15]); }
                else *q++ = '*';
                const u8 s[] = {'\t', '*', '\t', '0', '\t', '0', '\t'};
                q = putBytes(q, s, 7);
            }
            head = (u32)(q - p);
        }
        head = (u32)__shfl((int)head, (int)(threadIdx.x & 63 & ~15u), 64);
        u8 * seq = p + head, * qual = seq + R.len + 1;
        // SEQ / QUAL of the mapped strand: "ACGTN"[code] (lower case and IUPAC codes normalised like the host parser does), qualities clamped to
        // 6 bits, '!' under N (reference src/store_seqs.h Dna5Q)
        const u8 * st = A.text + R.seqStart, * qt = A.text + R.qualStart;
        const bool rev = o.mapped && o.isRev;
        for (u32 k = g; k < R.len; k += 16)
        {
            const u32 code = dna5CodeOf(st[k]);
            const u8 qc = qt[k];
            const u8 b = rev ? (u8)("TGCAN"[code]) : (u8)("ACGTN"[code]);
            const u8 ql = code == 4 ? (u8)'!' : (qc > 33 + 62 ? (u8)(33 + 62) : qc);
            const u32 at = rev ? R.len - 1 - k : k;
            seq[at] = b; qual[at] = ql;
        }
        if (g == 0)
        {
            u8 * q = seq + R.len; *q = '\t';
            q = qual + R.len;
            if (o.mapped)
            {
                const u8 nm[] = {'\t', 'N', 'M', ':', 'i', ':'}, x0[] = {'\t', 'X', '0', ':', 'i', ':'}, x1[] = {'\t', 'X', '1', ':', 'i', ':'};
                q = putBytes(q, nm, 6); q = putDec(q, o.errors);
                q = putBytes(q, x0, 6); q = putDec(q, o.bestCount);
                q = putBytes(q, x1, 6); q = putDec(q, o.subCount);
            }
            const u8 rg[] = {'\t', 'R', 'G', ':', 'Z', ':'};
            q = putBytes(q, rg, 6); q = putBytes(q, A.readGroup, A.readGroupLen);
            if (o.mapped)
            {
                const u32 nset = o.bestCount + o.subCount;
                bool skipped = false, any = false;
                for (u32 k = 0; k < nset; ++k)
                {
                    const OutMatch m = A.outSet[(size_t)o.setBegin + k];
                    if (!skipped && m.contig == o.contig && m.isRev == o.isRev && m.begin == o.begin) { skipped = true; continue; }
                    if (!any) { const u8 xa[] = {'\t', 'X', 'A', ':', 'Z', ':'}; q = putBytes(q, xa, 6); any = true; }
                    q = putBytes(q, A.contigNames + A.contigNameOffs[m.contig], A.contigNameOffs[m.contig + 1] - A.contigNameOffs[m.contig]); *q++ = ',';
                    q = putDec(q, (u64)m.begin + 1); *q++ = ',';
                    q = putDec(q, (u64)m.begin + m.length + 1); *q++ = ',';
                    *q++ = m.isRev ? '-' : '+'; *q++ = ',';
                    q = putDec(q, m.errors); *q++ = ';';
                }
            }
            *q++ = '\n';
        }
    }
}

} // namespace dy

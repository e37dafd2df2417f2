// dream_yara_mapper — drop-in command line for the reference's mapper (reference src/d_mapper.cpp:114-394,554-585),
// written against the C ABI of libdyhip (include/dyhip.h) exactly like a maintainer's binding would be.
// Host work here is I/O only: option parsing, FASTA/FASTQ batches in, SAM text out (reference src/mapper_writer.h).
#include "../../include/dyhip.h"
#include "host_formats.hpp"
#include "bam_writer.hpp"
#include "multi_gpu.hpp"
#include <charconv>
#include <condition_variable>
#include <mutex>
#include <cerrno>
#include <cstdlib>
#include <fcntl.h>
#include <unistd.h>
#include <iostream>
#include <fstream>
#include <sstream>
#include <algorithm>
#include <chrono>
#include <memory>
#include <thread>
#include <future>
#include <atomic>

using namespace dy;

namespace {

struct CliOptions
{
    std::string indicesDir, readsFile, readsFile2, outputFile, filterFile, filterType = "bloom", readGroup = "none";
    dy_params params;
    unsigned readsBatch = 100000;
    unsigned hostThreads = 0;          // 0 = min(32, hardware threads)
    bool verbose = false, skipSamHeader = false;
    std::string commandLine;
    // ---- not in the reference (its -t threads share one address space): devices of this node
    unsigned gpus = 1;                  // --gpus N
    std::vector<int> devices;           // --devices 0,1,...  (default 0..N-1; a device may be listed twice: two contexts on it, used by the tests)
    std::string shard = "reads";        // --shard reads | bins   (multi_gpu.hpp)
    std::string transport = "auto";     // --transport auto | rccl | copy   (bins mode)
    std::string outputFormat = "sam";   // -f: format of the standard output
    bool uncompressedBam = false;       // -u
    std::string binAssignment = "lpt";  // --bin-assignment lpt | mod   (bins mode: which device owns which bin)
    bool inputInPlace = false;          // --input-in-place: parse the memory-mapped reads file where it lies instead of a private copy of each window
    bool hostIo = false;                // --host-io: parse FASTQ and format SAM on the host threads even where the device could (single-end, four-line FASTQ in, SAM out)
    bool teardown = false;              // --teardown: release every device / host object and shut the runtime down in order before exiting
    int bamLevel = -1;                  // --bam-level 0..9: deflate level of BAM output (default: zlib's, 6)
};

[[noreturn]] void fail(const std::string & msg) { throw std::runtime_error(msg); }

void check(int rc) { if (rc != DY_OK) fail(dy_last_error()); }

CliOptions parse(int argc, char ** argv)
{
    CliOptions o;
    dy_params_default(&o.params);
    std::vector<std::string> pos;
    for (int i = 1; i < argc; ++i)
    {
        std::string a = argv[i];
        auto value = [&]() -> std::string { if (i + 1 >= argc) fail("option requires an argument: " + a); return argv[++i]; };
        auto uintValue = [&](unsigned lo, unsigned hi) -> unsigned {
            std::string v = value(); size_t used = 0; unsigned long x = 0;
            try { x = std::stoul(v, &used); } catch (...) { used = 0; }
            if (used != v.size() || x < lo || x > hi) fail("the given value '" + v + "' for option " + a + " is out of range");
            return (unsigned)x;
        };
        if (a == "-e" || a == "--error-rate") o.params.error_rate = (float)(uintValue(0, 10) / 100.0);          // src/d_mapper.cpp:339-342
        else if (a == "-s" || a == "--strata-rate") o.params.strata_rate = (float)(uintValue(0, 10) / 100.0);   // :344-346
        else if (a == "-y" || a == "--sensitivity")
        {
            std::string v = value();
            if (v == "low") o.params.sensitivity = 0; else if (v == "high") o.params.sensitivity = 1; else if (v == "full") o.params.sensitivity = 2;
            else fail("the given value '" + v + "' is not in the list of allowed values for sensitivity");
        }
        else if (a == "-fi" || a == "--bloom-filter") o.filterFile = value();
        else if (a == "-ft" || a == "--filter-type")
        {
            o.filterType = value();
            if (o.filterType != "bloom" && o.filterType != "none" && o.filterType != "kmer_direct") fail("unknown filter type " + o.filterType);
        }
        else if (a == "-o" || a == "--output-file") o.outputFile = value();
        else if (a == "-f" || a == "--output-format")
        {   // format of the standard output; with -o the file name's extension decides (reference src/d_mapper.cpp:144-148,306-312, src/d_mapper.h:768-785)
            o.outputFormat = value();
            if (o.outputFormat != "sam" && o.outputFormat != "bam") fail("the given value '" + o.outputFormat + "' is not in the list of allowed values for output-format (sam, bam)");
        }
        else if (a == "-u" || a == "--uncompressed-bam") o.uncompressedBam = true;                              // :150-153,314-316: BAM on standard output without compression
        else if (a == "-sa" || a == "--secondary-alignments")
        {   // kept by the reference only to say so (:165-168,323-327)
            (void)value();
            fail("The 'secondary-alignments' option has been renamed to 'secondary-matches'!");
        }
        else if (a == "-ra" || a == "--rabema-alignments")
            // the reference then takes the CIGARs from a unit-cost global alignment (LinearGaps, src/mapper.h:1203, src/mapper_aligner.h:155-160)
            fail("--rabema-alignments (CIGARs from a unit-cost global alignment instead of the affine one) is not implemented");
        else if (a == "-rg" || a == "--read-group") o.readGroup = value();
        else if (a == "-sm" || a == "--secondary-matches")
        {   // validated, then without effect: the reference reads the value under the legacy option's name (:320), so it stays `tag`
            const std::string v = value();
            if (v != "tag" && v != "record" && v != "omit") fail("the given value '" + v + "' is not in the list of allowed values for secondary-matches (tag, record, omit)");
        }
        else if (a == "-t" || a == "--threads") o.hostThreads = uintValue(1, 2048);      // the device replaces the mapping threads; these format the output
        else if (a == "-rb" || a == "--reads-batch") o.readsBatch = uintValue(1000, 5000000);
        else if (a == "-sk" || a == "--skip-sam-headers") o.skipSamHeader = true;
        else if (a == "-ll" || a == "--library-length") o.params.library_length = uintValue(1, 1000000000u);      // src/d_mapper.cpp:209-216
        else if (a == "-ld" || a == "--library-deviation") o.params.library_dev = uintValue(1, 1000000000u);
        else if (a == "-i" || a == "--indel-rate") o.params.indel_rate = (float)(uintValue(0, 50) / 100.0);         // :355-357
        else if (a == "-ni" || a == "--no-indels") o.params.verify_matches = 0;                                      // :359
        else if (a == "-v" || a == "--verbose" || a == "-vv" || a == "--very-verbose") o.verbose = true;
        else if (a == "--gpus") o.gpus = uintValue(1, 64);
        else if (a == "--devices")
        {
            std::stringstream ss(value()); std::string tok;
            while (std::getline(ss, tok, ',')) { try { o.devices.push_back(std::stoi(tok)); } catch (...) { fail("--devices takes a comma-separated list of device ordinals"); } }
        }
        else if (a == "--teardown") o.teardown = true;
        else if (a == "--host-io") o.hostIo = true;
        else if (a == "--input-in-place") o.inputInPlace = true;
        else if (a == "--bam-level") o.bamLevel = (int)uintValue(0, 9);
        else if (a == "--shard") { o.shard = value(); if (o.shard != "reads" && o.shard != "bins") fail("--shard takes reads or bins"); }
        else if (a == "--bin-assignment") { o.binAssignment = value(); if (o.binAssignment != "lpt" && o.binAssignment != "mod") fail("--bin-assignment takes lpt or mod"); }
        else if (a == "--transport") { o.transport = value(); if (o.transport != "auto" && o.transport != "rccl" && o.transport != "copy") fail("--transport takes auto, rccl or copy"); }
        else if (a.size() > 1 && a[0] == '-') fail("unknown option " + a);
        else pos.push_back(a);
    }
    if (pos.size() < 2) fail("usage: dream_yara_mapper [OPTIONS] <INDEX DIR> <READS>");
    if (pos.size() > 3) fail("Too many arguments!");                                                              // :300-302
    o.indicesDir = pos[0]; o.readsFile = pos[1];
    if (pos.size() == 3) { o.readsFile2 = pos[2]; o.params.paired = 1; }
    if (o.indicesDir.empty() || o.indicesDir.back() != '/') o.indicesDir += '/';                                // appendTrailingSlash :280
    if (o.filterFile.empty()) o.filterFile = o.indicesDir + "bloom.filter";                                     // :371-375
    if (o.readsBatch > 2097151) o.readsBatch = 2097151;
    if (!o.devices.empty()) o.gpus = (unsigned)o.devices.size();
    else for (unsigned d = 0; d < o.gpus; ++d) o.devices.push_back((int)d);
    for (int i = 0; i < argc; ++i) { if (i) o.commandLine += ' '; o.commandLine += argv[i]; }
    return o;
}

void appendUint(std::string & s, unsigned long long v) { char buf[24]; const auto r = std::to_chars(buf, buf + sizeof buf, v); s.append(buf, (size_t)(r.ptr - buf)); }

} // namespace

int main(int argc, char ** argv)
{
    try
    {
        CliOptions opt = parse(argc, argv);
        auto t0 = std::chrono::steady_clock::now();
        const unsigned N = opt.gpus;
        // bins are sharded when there is something to shard over — or when a transport is named, which runs the whole exchange sequence
        // on one device too (a self send/receive through RCCL: the only RCCL rehearsal a one-GPU box allows)
        const bool shardBins = opt.shard == "bins" && (opt.gpus > 1 || opt.transport != "auto");
        const unsigned hostThreads = opt.hostThreads ? opt.hostThreads : std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
        // ---- input side first: the first round of batches is parsed while the devices load the filter and the index
        ReadsFile in(opt.readsFile);
        std::unique_ptr<ReadsFile> in2;
        if (opt.params.paired) in2.reset(new ReadsFile(opt.readsFile2));
        in.copyWindow = !opt.inputInPlace; if (in2) in2->copyWindow = !opt.inputInPlace;
        if (opt.params.paired && opt.readsBatch > 1048575) opt.readsBatch = 1048575;      // both mates share the 21-bit read id space
        // A batch is mapped in PARTS: one part = the whole batch on one device (--shard reads: consecutive batches go to consecutive
        // devices) or, with --shard bins, the slice of the batch that is at home on a device.  A part covers reads [lo, hi) (single-end)
        // or pairs [lo, hi) = reads [lo, hi) + [P + lo, P + hi) (paired).  Three ROUNDS of up to N batches are in flight: one being
        // parsed, one on the devices, one being formatted and written (in input order).
        struct Part { size_t lo = 0, hi = 0; unsigned dev = 0; dy_reads * reads = nullptr; dy_result * res = nullptr; std::vector<u8> codes; std::vector<u64> offs; };
        struct Slot { ReadsBatch batch; size_t nPairs = 0; std::vector<Part> parts; };
        struct Round { std::vector<Slot> slots; size_t used = 0; };
        const unsigned perRound = shardBins ? 1u : N;
        Round rounds[3];
        for (auto & r : rounds) r.slots.resize(perRound);
        unsigned long long total = 0, mapped = 0;
        double mapSeconds = 0, parseSeconds = 0, formatSeconds = 0, writeSeconds = 0;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(now() - t).count(); };
        auto readInto = [&](Slot & sl) -> bool {
            ReadsBatch & batch = sl.batch;
            if (!in.next(batch, opt.readsBatch, hostThreads)) return false;
            if (in2)
            {   // readRecords of a file pair: first mates, then second mates (reference src/file_pair.h:81-88)
                ReadsBatch second;
                if (in2->next(second, batch.size(), hostThreads) != batch.size()) fail("the two reads files hold different numbers of records");
                batch.append(second);
            }
            sl.nPairs = batch.size() / 2;
            return true;
        };
        auto readRound = [&](Round & rd) -> bool {
            const auto p0 = now();
            rd.used = 0;
            while (rd.used < rd.slots.size() && readInto(rd.slots[rd.used])) ++rd.used;
            parseSeconds += since(p0);
            return rd.used > 0;
        };
        // Device I/O (single-end, uncompressed four-line FASTQ in, SAM out, one device): the reads file goes to the device as bytes and the SAM
        // records come back as bytes (dy_reads_from_fastq / dy_result_sam) — the host threads only move them.  Anything else, and any batch the
        // device parser refuses, takes the host path below.
        const bool bam = opt.outputFile.empty() ? opt.outputFormat == "bam"
                                                : (opt.outputFile.size() >= 4 && opt.outputFile.compare(opt.outputFile.size() - 4, 4, ".bam") == 0);
        const bool deviceIo = !opt.hostIo && !opt.params.paired && !bam && N == 1 && opt.shard != "bins" && in.plainFastq();
        std::future<bool> firstRead;
        if (!deviceIo) firstRead = std::async(std::launch::async, [&] { return readRound(rounds[0]); });
        // one context (+ filter + index) per listed device
        struct Device { int ordinal = 0; dy_ctx * ctx = nullptr; dy_ibf * ibf = nullptr; dy_index * idx = nullptr; };
        std::vector<Device> devs(N);
        uint32_t nBins = 0;
        if (!(opt.filterType == "bloom" || opt.filterType == "kmer_direct"))
        {
            opt.params.use_filter = 0;
            for (;; ++nBins) { std::ifstream f(opt.indicesDir + std::to_string(nBins) + ".txt.size"); if (!f) break; }
            if (!nBins) fail("Error while opening reference file.");
        }
        // bin -> device, static per job: longest-processing-time-first on the bins' text sizes (multi_gpu.hpp assignBins; the per-batch
        // counterpart in the reference is sortedBins(), src/d_mapper.h:843-854), or plain bin mod N (--bin-assignment mod)
        auto binOwners = [&](uint32_t bins) {
            std::vector<uint64_t> sizes(bins, 1);
            if (opt.binAssignment == "lpt")
                for (uint32_t b = 0; b < bins; ++b)
                {
                    std::ifstream f(opt.indicesDir + std::to_string(b) + ".txt.size", std::ios::binary);
                    uint64_t v[3] = {0, 0, 1};                                     // contigsMaxLength, contigs, contigsSum (reference src/misc_options.h:293-331)
                    if (f) f.read((char *)v, sizeof v);
                    sizes[b] = v[2];
                }
            return assignBins(sizes, N);
        };
        {
            std::vector<std::string> errors(N);
            auto setup = [&](unsigned d) {
                try
                {
                    Device & D = devs[d];
                    D.ordinal = opt.devices[d];
                    check(dy_ctx_create(D.ordinal, &D.ctx));
                    uint32_t bins = nBins;
                    if (opt.params.use_filter)
                    {   // reference src/d_mapper.h:910-936
                        check(opt.filterType == "bloom" ? dy_ibf_load(D.ctx, opt.filterFile.c_str(), &D.ibf) : dy_kdx_load(D.ctx, opt.filterFile.c_str(), &D.ibf));
                        dy_ibf_info info; check(dy_ibf_get_info(D.ibf, &info));
                        bins = info.n_bins;                                                                         // readFilterMetadata :532-548
                    }
                    if (shardBins)
                    {
                        const std::vector<uint32_t> owner = binOwners(bins);
                        std::vector<uint8_t> owned(bins);
                        for (uint32_t b = 0; b < bins; ++b) owned[b] = owner[b] == d;
                        check(dy_index_load_sharded(D.ctx, opt.indicesDir.c_str(), bins, owned.data(), &D.idx));
                    }
                    else check(dy_index_load(D.ctx, opt.indicesDir.c_str(), bins, &D.idx));
                }
                catch (std::exception const & e) { errors[d] = e.what(); }
            };
            std::vector<std::thread> th;
            for (unsigned d = 1; d < N; ++d) th.emplace_back(setup, d);
            setup(0);
            for (auto & t : th) t.join();
            for (auto & e : errors) if (!e.empty()) fail(e);
            nBins = dy_index_bins(devs[0].idx);
        }
        dy_index * idx = devs[0].idx;
        auto t1 = std::chrono::steady_clock::now();
        // bins mode: one ShardRank per device + the exchange between them
        std::unique_ptr<Exchange> exchange;
        std::vector<ShardRank> ranks;
        if (shardBins)
        {
            bool distinct = true;
            for (unsigned i = 0; i < N; ++i) for (unsigned j = 0; j < i; ++j) distinct &= opt.devices[i] != opt.devices[j];
            if (opt.transport == "rccl" && !distinct) fail("--transport rccl needs distinct devices");
            exchange.reset(new Exchange(opt.devices, opt.transport == "rccl" || (opt.transport == "auto" && distinct)));
            ranks.resize(N);
            const uint32_t W = (nBins + 63) / 64;
            for (unsigned d = 0; d < N; ++d)
            {
                ShardRank & R = ranks[d];
                R.rank = d; R.nRanks = N; R.device = devs[d].ordinal; R.ctx = devs[d].ctx; R.ibf = devs[d].ibf; R.idx = devs[d].idx; R.nBins = nBins; R.W = W;
                R.ownerMasks.assign((size_t)N * W, 0);
                const std::vector<uint32_t> owner = binOwners(nBins);
                for (uint32_t b = 0; b < nBins; ++b) R.ownerMasks[(size_t)owner[b] * W + b / 64] |= 1ull << (b % 64);
            }
        }

        // output format: by extension with -o (reference src/d_mapper.cpp:140-153): .bam -> BGZF-compressed BAM, anything else -> SAM text; -f for the standard output
        if (opt.uncompressedBam && opt.outputFile.empty() && bam) opt.bamLevel = 0;                               // reference src/d_mapper.h:771-776
        // SAM into a file goes through a plain descriptor (no stream buffer in between: the chunks are hundreds of MB)
        std::ofstream ofs;
        int samFd = -1; uint64_t samOffset = 0;
        if (!opt.outputFile.empty())
        {
            if (bam) { ofs.open(opt.outputFile, std::ios::binary); if (!ofs) fail("Error while opening output file."); }
            else { samFd = ::open(opt.outputFile.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666); if (samFd < 0) fail("Error while opening output file."); }
        }
        std::ostream & os = opt.outputFile.empty() ? std::cout : ofs;
        auto writeAll = [&](int fd, const char * p, size_t n, uint64_t at) {
            while (n)
            {
                const ssize_t w = ::pwrite(fd, p, n, (off_t)at);
                if (w < 0) { if (errno == EINTR) continue; fail("Error while writing the output file."); }
                p += w; n -= (size_t)w; at += (uint64_t)w;
            }
        };
        std::vector<std::string> contigNames; std::vector<uint64_t> contigLengths;
        for (uint32_t c = 0; c < dy_index_contigs(idx); ++c) { contigNames.push_back(dy_index_contig_name(idx, c)); contigLengths.push_back(dy_index_contig_length(idx, c)); }
        std::string headerText;
        if (!opt.skipSamHeader)
        {   // fillHeader, reference src/mapper_writer.h:124-155; order as in the gold files
            headerText += "@HD\tVN:1.4\tSO:unsorted\n";
            headerText += "@PG\tID:Yara\tPN:Yara\tVN:1.0.2 []\tCL:" + opt.commandLine + "\n";
            headerText += "@RG\tID:" + opt.readGroup + "\tSM:" + opt.readGroup + "\tPG:Yara\n";
            for (size_t c = 0; c < contigNames.size(); ++c) headerText += "@SQ\tSN:" + contigNames[c] + "\tLN:" + std::to_string(contigLengths[c]) + "\n";
        }
        BgzfWriter bgzf(os, opt.bamLevel);
        if (bam) bamWriteHeader(bgzf, headerText, contigNames, contigLengths);
        else if (samFd >= 0) { writeAll(samFd, headerText.data(), headerText.size(), 0); samOffset = headerText.size(); }
        else os << headerText;

        // batch index of part-local read j
        auto batchIndex = [&](const Slot & sl, const Part & pt, size_t j) -> size_t {
            if (!opt.params.paired) return pt.lo + j;
            const size_t np = pt.hi - pt.lo;
            return j < np ? pt.lo + j : sl.nPairs + pt.lo + (j - np);
        };
        // ---- alignment records of part-local reads [jlo, jhi) (reference src/mapper_writer.h:296-345, 593-634; pairs :268-293,347-400,515-571)
        auto formatRange = [&](const Slot & sl, const Part & pt, size_t jlo, size_t jhi, std::string & sam, std::vector<uint8_t> & bamOut, unsigned long long & nMapped) {
            const ReadsBatch & batch = sl.batch;
            const size_t np = opt.params.paired ? pt.hi - pt.lo : 0;
            const dy_read_result * rr = dy_result_reads(pt.res);
            const dy_match * ms = dy_result_matches(pt.res);
            const uint32_t * cg = dy_result_cigars(pt.res);
            if (jhi > jlo)
            {
                const size_t i0 = batchIndex(sl, pt, jlo);
                const size_t guess = (jhi - jlo) * (size_t)(2 * (batch.offs[i0 + 1] - batch.offs[i0]) + 96);
                if (bam) bamOut.reserve(guess); else sam.reserve(guess);
            }
            AlnRecord a;                    // reused: its strings keep their capacity
            for (size_t j = jlo; j < jhi; ++j)
            {
                const size_t i = batchIndex(sl, pt, j);
                const dy_read_result & r = rr[j];
                a.flag = 0; a.rid = a.pos = -1; a.mapq = 0; a.cigar = nullptr; a.nCigar = 0; a.rnext = a.pnext = -1; a.tlen = 0;
                a.hasCounts = false; a.nm = a.x0 = a.x1 = 0; a.xa.clear();
                const std::string_view name = batch.name(i);
                const size_t cut = name.find_first_of(" \t/");
                a.qname.assign(name.data(), cut == std::string_view::npos ? name.size() : cut);
                const u8 * codes = batch.codes.data() + batch.offs[i];
                const char * quals = batch.quals.data() + batch.offs[i];
                const size_t len = (size_t)(batch.offs[i + 1] - batch.offs[i]);
                const bool rev = r.mapped && r.primary.is_rev;
                if (r.mapped)
                {
                    a.rid = (int32_t)r.primary.contig; a.pos = (int32_t)r.primary.begin; a.mapq = r.mapq;
                    a.cigar = cg + r.cigar_begin; a.nCigar = r.cigar_len;
                    if (rev) a.flag |= 16;
                }
                else a.flag |= 4;
                if (opt.params.paired)
                {
                    const dy_read_result & mr = rr[j < np ? j + np : j - np];
                    a.flag |= 1u | (j < np ? 64u : 128u);
                    if (!mr.mapped) a.flag |= 8;
                    if (mr.mapped && mr.primary.is_rev) a.flag |= 32;
                    if (r.mapped && mr.mapped && r.paired)
                    {
                        const bool oneBeforeTwo = r.primary.begin < mr.primary.begin;
                        if ((!r.primary.is_rev && mr.primary.is_rev && oneBeforeTwo) || (!mr.primary.is_rev && r.primary.is_rev && !oneBeforeTwo)) a.flag |= 2;
                        if (r.primary.contig == mr.primary.contig)
                            a.tlen = (int32_t)(oneBeforeTwo ? (long long)mr.primary.begin + mr.primary.length - (long long)r.primary.begin
                                                            : (long long)mr.primary.begin - ((long long)r.primary.begin + r.primary.length));
                    }
                    if (!r.mapped && mr.mapped) { a.rid = (int32_t)mr.primary.contig; a.pos = (int32_t)mr.primary.begin; }     // an unmapped mate sits at its mate's position
                    if (mr.mapped) { a.rnext = (int32_t)mr.primary.contig; a.pnext = (int32_t)mr.primary.begin; }
                    else if (r.mapped) { a.rnext = a.rid; a.pnext = a.pos; }
                }
                a.seq.resize(len); a.qual.resize(len);
                if (!rev) for (size_t k = 0; k < len; ++k) { a.seq[k] = "ACGTN"[codes[k]]; a.qual[k] = quals[k]; }
                else for (size_t k = 0; k < len; ++k) { u8 c = codes[len - 1 - k]; a.seq[k] = "ACGTN"[c < 4 ? 3 - c : 4]; a.qual[k] = quals[len - 1 - k]; }
                if (r.mapped) { a.hasCounts = true; a.nm = r.primary.errors; a.x0 = r.best_count; a.x1 = r.sub_count; }
                a.rg = opt.readGroup;
                if (r.mapped)
                {   // XA: every entry of matchesSet except the primary (first entry with the primary's contig/strand/begin)
                    bool skipped = false;
                    const uint32_t nset = r.best_count + r.sub_count;
                    for (uint32_t k = 0; k < nset; ++k)
                    {
                        const dy_match & m = ms[r.set_begin + k];
                        if (!skipped && m.contig == r.primary.contig && m.is_rev == r.primary.is_rev && m.begin == r.primary.begin) { skipped = true; continue; }
                        a.xa += contigNames[m.contig]; a.xa += ',';
                        appendUint(a.xa, (unsigned long long)m.begin + 1); a.xa += ',';
                        appendUint(a.xa, (unsigned long long)m.begin + m.length + 1); a.xa += ',';
                        a.xa += m.is_rev ? '-' : '+'; a.xa += ',';
                        appendUint(a.xa, m.errors); a.xa += ';';
                    }
                }
                if (bam) bamEncodeRecord(a, bamOut);
                else samWriteRecord(sam, a, contigNames);
                nMapped += r.mapped;
            }
        };
        // A batch's records are formatted by the host threads into chunks (in output order), then written by one thread while the next
        // batch is being formatted.
        struct Chunk { const Part * pt; size_t jlo, jhi; std::string sam; std::vector<uint8_t> bam; unsigned long long nMapped = 0; };
        struct Formatted { std::vector<Chunk> chunks; unsigned long long reads = 0; };
        auto formatSlot = [&](Slot & sl, Formatted & out) {
            // output order = input order: the first mates of every part, then the second mates of every part (single-end: one pass)
            const size_t first = out.chunks.size();
            const unsigned passes = opt.params.paired ? 2 : 1;
            for (unsigned pass = 0; pass < passes; ++pass)
                for (const Part & pt : sl.parts)
                {
                    const size_t np = pt.hi - pt.lo;
                    const size_t base = pass * np;
                    const size_t pieces = std::max<size_t>(1, std::min<size_t>(hostThreads, np / 4096 + 1));
                    for (size_t c = 0; c < pieces; ++c) { Chunk ch; ch.pt = &pt; ch.jlo = base + np * c / pieces; ch.jhi = base + np * (c + 1) / pieces; out.chunks.push_back(std::move(ch)); }
                }
            const auto f0 = now();
            {
                std::atomic<size_t> next(first);
                auto work = [&] { for (;;) { const size_t c = next.fetch_add(1); if (c >= out.chunks.size()) break; Chunk & ch = out.chunks[c]; formatRange(sl, *ch.pt, ch.jlo, ch.jhi, ch.sam, ch.bam, ch.nMapped); } };
                const unsigned T = (unsigned)std::min<size_t>(hostThreads, out.chunks.size() - first);
                std::vector<std::thread> pool;
                for (unsigned t = 1; t < T; ++t) pool.emplace_back(work);
                work();
                for (auto & th : pool) th.join();
            }
            formatSeconds += since(f0);
            out.reads += sl.batch.size();
        };
        auto formatRound = [&](Round & rd) { Formatted f; for (size_t i = 0; i < rd.used; ++i) formatSlot(rd.slots[i], f); return f; };
        auto writeFormatted = [&](Formatted f) {
            const auto w0 = now();
            if (!bam && samFd >= 0)
            {   // one writer: buffered writes to one file serialise on the inode lock (scripts/ubench/write_paths.cpp: 11 GB/s from one thread, 9 from eight)
                for (Chunk & ch : f.chunks) { writeAll(samFd, ch.sam.data(), ch.sam.size(), samOffset); samOffset += ch.sam.size(); mapped += ch.nMapped; }
            }
            else
            for (Chunk & ch : f.chunks)
            {
                if (bam) bgzf.writeParallel(ch.bam.data(), ch.bam.size(), hostThreads);
                else os.write(ch.sam.data(), (std::streamsize)ch.sam.size());
                mapped += ch.nMapped;
            }
            total += f.reads;
            writeSeconds += since(w0);
        };
        auto releaseRound = [&](Round & rd) {          // device objects go back while no thread drives the contexts
            for (size_t i = 0; i < rd.used; ++i)
                for (Part & pt : rd.slots[i].parts)
                {
                    if (pt.res) { dy_result_free(pt.res); pt.res = nullptr; }
                    if (pt.reads) { dy_reads_free(pt.reads); pt.reads = nullptr; }
                }
        };
        auto thresholdOf = [&](const ReadsBatch & batch, dy_ibf * ibf) -> uint16_t {
            // threshold from the batch-average read length (reference src/d_mapper.h:313-315)
            uint16_t avg = (uint16_t)(batch.offs.back() * 2 / (batch.size() * 2));
            dy_ibf_info info; check(dy_ibf_get_info(ibf, &info));
            uint16_t thr = dy_ibf_threshold(opt.params.error_rate, info.kmer_size, avg);
            if (thr == 0) std::cerr << "[WARNING!] 0 k-mer is required to filter a read!\nAll reads will pass filteration and be mapped everywhere.\n";
            return thr;
        };
        // one whole batch on one device (single GPU, and --shard reads)
        auto mapWhole = [&](Slot & sl, unsigned d) {
            Device & D = devs[d];
            sl.parts.assign(1, Part());
            Part & pt = sl.parts[0];
            pt.dev = d; pt.lo = 0; pt.hi = opt.params.paired ? sl.nPairs : sl.batch.size();
            const ReadsBatch & batch = sl.batch;
            check(dy_reads_upload(D.ctx, batch.codes.data(), batch.offs.data(), (uint32_t)batch.size(), &pt.reads));
            if (D.ibf) check(dy_ibf_query(D.ibf, pt.reads, thresholdOf(batch, D.ibf), nullptr));
            check(dy_map_reads(D.ctx, D.idx, pt.reads, &opt.params, &pt.res));
        };
        // one batch over all devices, bins sharded: device d is home to the d-th slice of the reads (pairs)
        auto mapSharded = [&](Slot & sl) {
            const ReadsBatch & batch = sl.batch;
            const size_t units = opt.params.paired ? sl.nPairs : batch.size();
            const uint16_t thr = devs[0].ibf ? thresholdOf(batch, devs[0].ibf) : 0;
            sl.parts.assign(N, Part());
            std::vector<std::string> errors(N);
            auto rankWork = [&](unsigned d) {
                try
                {
                    Part & pt = sl.parts[d];
                    pt.dev = d; pt.lo = units * d / N; pt.hi = units * (d + 1) / N;
                    // the slice as its own batch: reads [lo, hi) (+ their mates behind them)
                    pt.codes.clear(); pt.offs.assign(1, 0);
                    const unsigned halves = opt.params.paired ? 2 : 1;
                    for (unsigned h = 0; h < halves; ++h)
                        for (size_t u = pt.lo; u < pt.hi; ++u)
                        {
                            const size_t i = h * sl.nPairs + u;
                            pt.codes.insert(pt.codes.end(), batch.codes.begin() + batch.offs[i], batch.codes.begin() + batch.offs[i + 1]);
                            pt.offs.push_back(pt.codes.size());
                        }
                    Device & D = devs[d];
                    check(dy_reads_upload(D.ctx, pt.codes.data(), pt.offs.data(), (uint32_t)(pt.offs.size() - 1), &pt.reads));
                    if (D.ibf) check(dy_ibf_query(D.ibf, pt.reads, thr, nullptr));
                    else check(dy_reads_select_all_bins(D.ctx, pt.reads, nBins));
                    pt.res = ranks[d].mapBatch(*exchange, pt.reads, opt.params);
                }
                catch (std::exception const & e) { errors[d] = e.what(); exchange->abort(); }
            };
            std::vector<std::thread> th;
            for (unsigned d = 1; d < N; ++d) th.emplace_back(rankWork, d);
            rankWork(0);
            for (auto & t : th) t.join();
            for (auto & e : errors) if (!e.empty() && e != "another rank failed") fail(e);
            for (auto & e : errors) if (!e.empty()) fail(e);
        };
        auto mapRound = [&](Round & rd) {
            const auto m0 = now();
            if (shardBins) mapSharded(rd.slots[0]);
            else
            {
                std::vector<std::string> errors(rd.used);
                auto work = [&](size_t i) { try { mapWhole(rd.slots[i], (unsigned)i); } catch (std::exception const & e) { errors[i] = e.what(); } };
                std::vector<std::thread> th;
                for (size_t i = 1; i < rd.used; ++i) th.emplace_back(work, i);
                work(0);
                for (auto & t : th) t.join();
                for (auto & e : errors) if (!e.empty()) fail(e);
            }
            mapSeconds += since(m0);
        };
        // Four stages in flight: one round being parsed, one on the devices, one being formatted, one being written (in input order: at
        // most one formatter and one writer at a time).  A round's device objects go back when its records are formatted.
        std::future<Formatted> formatter;
        std::future<void> writer;
        int formatting = -1;
        double waitFormat = 0, waitWrite = 0, waitRead = 0, releaseSeconds = 0;
        auto retireFormatter = [&] {
            auto w0 = now();
            Formatted f = formatter.get();
            waitFormat += since(w0); w0 = now();
            releaseRound(rounds[formatting]); formatting = -1;
            releaseSeconds += since(w0); w0 = now();
            if (writer.valid()) writer.get();
            waitWrite += since(w0);
            writer = std::async(std::launch::async, [&writeFormatted, f = std::move(f)]() mutable { writeFormatted(std::move(f)); });
        };
        double ioRead = 0, ioWrite = 0, ioParse = 0, ioMap = 0, ioFormat = 0, ioWaitRead = 0, ioWaitWrite = 0;
        if (deviceIo)
        {   // ---- three threads, two pinned slots each way: read window j+1 | parse, map, format batch k on the device | write batch k-1
            Device & D = devs[0];
            const size_t fileSize = in.fileSize();
            std::mutex m; std::condition_variable cv;
            // reader: request j = (start, bytes) -> pinned slot j & 1 filled.  Request j + 1 is issued once window j has been parsed (its bytes are on
            // the device and the batch's end in the file is known), so the read of the next window runs beside the mapping of this batch.
            size_t reqStart[2] = {0, 0}, reqBytes[2] = {0, 0}; int issued = 0, filled = 0; bool stop = false;
            void * inSlot[2] = {nullptr, nullptr};
            std::string readerError, writerError;
            std::thread reader([&] {
                try
                {
                    for (int j = 0;; ++j)
                    {
                        size_t start, bytes;
                        { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return issued > j || stop; }); if (issued <= j) return; start = reqStart[j & 1]; bytes = reqBytes[j & 1]; }
                        const auto r0 = now();
                        void * buf = nullptr;
                        check(dy_host_buffer(D.ctx, (uint32_t)(j & 1), bytes, &buf));
                        if (bytes && !in.readRaw((u8 *)buf, start, bytes, 2)) fail("Error while reading the reads file.");
                        { std::lock_guard<std::mutex> lk(m); ioRead += since(r0); inSlot[j & 1] = buf; filled = j + 1; }
                        cv.notify_all();
                    }
                }
                catch (std::exception const & e) { std::lock_guard<std::mutex> lk(m); readerError = e.what(); cv.notify_all(); }
            });
            // writer: the SAM text of batch k lies in pinned slot 2 + k % 3 (three slots: the writer — the slowest stage, one thread copying into the
            // page cache — then runs back to back while the device works up to two batches ahead)
            const char * outText[3] = {nullptr, nullptr, nullptr}; uint64_t outBytes[3] = {0, 0, 0}; int produced = 0, written = 0;
            std::thread writerThread([&] {
                try
                {
                    for (int k = 0;; ++k)
                    {
                        const char * t; uint64_t n;
                        { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return produced > k || stop; }); if (produced <= k) return; t = outText[k % 3]; n = outBytes[k % 3]; }
                        const auto w0 = now();
                        if (samFd >= 0) { writeAll(samFd, t, (size_t)n, samOffset); samOffset += n; }
                        else os.write(t, (std::streamsize)n);
                        { std::lock_guard<std::mutex> lk(m); ioWrite += since(w0); written = k + 1; }
                        cv.notify_all();
                    }
                }
                catch (std::exception const & e) { std::lock_guard<std::mutex> lk(m); writerError = e.what(); cv.notify_all(); }
            });
            auto request = [&](size_t start, size_t bytes) { { std::lock_guard<std::mutex> lk(m); reqStart[issued & 1] = start; reqBytes[issued & 1] = bytes; ++issued; } cv.notify_all(); };
            size_t pos = 0; bool fellBack = false; std::string failure;
            try
            {
                // first window: sized from the record length seen in the first 256 KB of the file
                size_t window = std::min<size_t>(fileSize, (size_t)opt.readsBatch * 512 + 65536);
                {
                    std::vector<u8> probe(std::min<size_t>(fileSize, (size_t)256 << 10));
                    size_t lines = 0;
                    if (!probe.empty() && in.readRaw(probe.data(), 0, probe.size(), 1)) for (u8 c : probe) lines += c == '\n';
                    if (lines >= 8) window = std::min<size_t>(fileSize, (size_t)((double)probe.size() / (double)(lines / 4) * opt.readsBatch * 1.05) + 65536);
                }
                request(0, window);
                int k = 0;
                for (int j = 0; pos < fileSize; ++j)
                {
                    const auto dw = now();
                    { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return filled > j || !readerError.empty(); }); if (!readerError.empty()) fail(readerError); }
                    auto d0 = now();
                    ioWaitRead += since(dw); 
                    const size_t bytes = reqBytes[j & 1];
                    const bool atEof = pos + bytes >= fileSize;
                    dy_reads * reads = nullptr; uint64_t consumed = 0;
                    const int rc = dy_reads_from_fastq(D.ctx, inSlot[j & 1], bytes, opt.readsBatch, atEof ? 1u : 0u, &reads, &consumed);
                    if (rc == DY_EUNSUPPORTED) { fellBack = true; break; }                            // the rest of the file goes through the host parser
                    check(rc);
                    const uint32_t n = dy_reads_count(reads);
                    if (n == 0 && atEof) { dy_reads_free(reads); break; }                             // nothing but an incomplete record is left: the host parser reports it
                    if (n < opt.readsBatch && !atEof)
                    {   // the window held fewer complete records than a batch: once more, larger
                        dy_reads_free(reads);
                        window = std::min(fileSize - pos, bytes + bytes / 2 + 65536);
                        request(pos, window);
                        continue;
                    }
                    ioParse += since(d0); d0 = now();
                    pos += consumed;
                    window = std::min(fileSize - pos, (size_t)((double)consumed / (double)n * opt.readsBatch * 1.03) + 65536);
                    if (pos < fileSize) request(pos, window);                                         // read while this batch is mapped
                    if (D.ibf)
                    {
                        const uint16_t avg = (uint16_t)(dy_reads_total_bases(reads) * 2 / ((uint64_t)n * 2));          // reference src/d_mapper.h:313-315
                        dy_ibf_info info; check(dy_ibf_get_info(D.ibf, &info));
                        const uint16_t thr = dy_ibf_threshold(opt.params.error_rate, info.kmer_size, avg);
                        if (thr == 0) std::cerr << "[WARNING!] 0 k-mer is required to filter a read!\nAll reads will pass filteration and be mapped everywhere.\n";
                        check(dy_ibf_query(D.ibf, reads, thr, nullptr));
                    }
                    dy_result * res = nullptr;
                    int rm = dy_map_reads_async(D.ctx, D.idx, reads, &opt.params, &res);
                    if (rm != DY_OK) { dy_reads_free(reads); check(rm); }
                    ioMap += since(d0); d0 = now();
                    { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return written >= k - 2 || !writerError.empty(); }); if (!writerError.empty()) fail(writerError); }   // slot of batch k - 3 is free
                    ioWaitWrite += since(d0); d0 = now();
                    const char * text = nullptr; uint64_t nBytes = 0, nMapped = 0;
                    const int rs = dy_result_sam(res, reads, D.idx, opt.readGroup.c_str(), (uint32_t)(2 + k % 3), &text, &nBytes, &nMapped);
                    dy_result_free(res); dy_reads_free(reads);
                    if (rs == DY_EUNSUPPORTED) { pos -= consumed; fellBack = true; break; }           // (a mapping quality on a rounding boundary: the host formats from here on)
                    check(rs);
                    total += n; mapped += nMapped;
                    ioFormat += since(d0);
                    { std::lock_guard<std::mutex> lk(m); outText[k % 3] = text; outBytes[k % 3] = nBytes; produced = k + 1; }
                    cv.notify_all();
                    ++k;
                }
                { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return written >= k || !writerError.empty(); }); }
            }
            catch (std::exception const & e) { failure = e.what(); }
            { std::lock_guard<std::mutex> lk(m); stop = true; }
            cv.notify_all();
            reader.join(); writerThread.join();
            if (!failure.empty()) fail(failure);
            if (!writerError.empty()) fail(writerError);
            mapSeconds += ioParse + ioMap + ioFormat; parseSeconds += ioRead; writeSeconds += ioWrite;
            if (fellBack || pos < fileSize) { in.seek(pos); firstRead = std::async(std::launch::async, [&] { return readRound(rounds[0]); }); }
        }
        int cur = 0;
        bool have = firstRead.valid() ? firstRead.get() : false;
        while (have)
        {
            const int nxt = (cur + 1) % 3;
            if (formatting == nxt) retireFormatter();                                   // the round about to be refilled must be formatted
            std::future<bool> reader = std::async(std::launch::async, [&, nxt] { return readRound(rounds[nxt]); });
            mapRound(rounds[cur]);
            if (formatting >= 0) retireFormatter();
            formatter = std::async(std::launch::async, [&, cur] { return formatRound(rounds[cur]); });
            formatting = cur;
            { const auto w0 = now(); have = reader.get(); waitRead += since(w0); }
            cur = nxt;
        }
        if (formatting >= 0) retireFormatter();
        if (writer.valid()) writer.get();
        if (bam) bgzf.finish();
        os.flush();
        if (samFd >= 0 && ::close(samFd) != 0) fail("Error while writing the output file.");
        if (opt.verbose)
        {
            dy_map_stats s; dy_map_get_stats(devs[0].ctx, &s);
            std::cerr << "Devices:\t\t\t" << N << (N > 1 ? (shardBins ? (exchange->rccl() ? " (bins sharded, RCCL)" : " (bins sharded, device-to-device copies)") : " (reads sharded)") : "") << "\n"
                      << "Loading time:\t\t\t" << std::chrono::duration<double>(t1 - t0).count() << " sec\n"
                      << "Mapping time:\t\t\t" << mapSeconds << " sec\n"
                      << "Parse / format / write time (overlapped with mapping):\t" << parseSeconds << " / " << formatSeconds << " / " << writeSeconds << " sec\n"
                      << "Device I/O path: upload + parse / sweep + map / format + download " << ioParse << " / " << ioMap << " / " << ioFormat << " sec, waited for the reader / writer " << ioWaitRead << " / " << ioWaitWrite << " sec\n"
                      << "Main loop waited for parse / format / write / release:\t" << waitRead << " / " << waitFormat << " / " << waitWrite << " / " << releaseSeconds << " sec\n"
                      << "Since start:\t\t\t" << std::chrono::duration<double>(now() - t0).count() << " sec\n"
                      << "Total reads:\t\t\t" << total << "\nMapped reads:\t\t\t" << mapped << "\n"
                      << "Last batch (device 0): items " << s.items << ", raw matches " << s.raw_matches << ", hard items " << s.hard_items
                      << ", ms ibf/items/search/extend/finalize " << s.ms_ibf << "/" << s.ms_items << "/" << s.ms_search << "/" << s.ms_extend << "/" << s.ms_finalize << "\n";
        }
        if (!opt.teardown)
        {   // every byte is out: the tens of GB on the devices and in host vectors go back with the process (releasing them one by one and
            // shutting the HIP runtime down in order costs 0.4 s, a quarter of a 16 M-read run); --teardown keeps the orderly exit
            std::cout.flush(); std::cerr.flush();
            std::_Exit(0);
        }
        ranks.clear(); exchange.reset();
        for (auto & D : devs)
        {
            if (D.idx) dy_index_free(D.idx);
            if (D.ibf) dy_ibf_free(D.ibf);
            if (D.ctx) dy_ctx_free(D.ctx);
        }
    }
    catch (std::exception const & e)
    {
        std::cerr << "dream_yara_mapper: " << e.what() << std::endl;     // reference src/d_mapper.cpp:574-582
        return 1;
    }
    return 0;
}

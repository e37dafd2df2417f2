// dream_yara_mapper — drop-in command line for the reference's mapper (reference src/d_mapper.cpp:114-394,554-585),
// written against the C ABI of libdyhip (include/dyhip.h) exactly like a maintainer's binding would be.
// Host work here is I/O only: option parsing, FASTA/FASTQ batches in, SAM text out (reference src/mapper_writer.h).
#include "../../include/dyhip.h"
#include "host_formats.hpp"
#include "bam_writer.hpp"
#include <charconv>
#include <iostream>
#include <fstream>
#include <sstream>
#include <algorithm>
#include <chrono>
#include <memory>
#include <thread>
#include <future>

using namespace dy;

namespace {

struct CliOptions
{
    std::string indicesDir, readsFile, readsFile2, outputFile, filterFile, filterType = "bloom", readGroup = "none";
    dy_params params;
    unsigned readsBatch = 100000;
    unsigned hostThreads = 0;          // 0 = min(16, hardware threads)
    bool verbose = false, skipSamHeader = false;
    std::string commandLine;
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
        else if (a == "-rg" || a == "--read-group") o.readGroup = value();
        else if (a == "-sm" || a == "--secondary-matches") value();      // read under a legacy name by the reference: no effect (:320)
        else if (a == "-t" || a == "--threads") o.hostThreads = uintValue(1, 2048);      // the device replaces the mapping threads; these format the output
        else if (a == "-rb" || a == "--reads-batch") o.readsBatch = uintValue(1000, 5000000);
        else if (a == "-sk" || a == "--skip-sam-headers") o.skipSamHeader = true;
        else if (a == "-ll" || a == "--library-length") o.params.library_length = uintValue(1, 1000000000u);      // src/d_mapper.cpp:209-216
        else if (a == "-ld" || a == "--library-deviation") o.params.library_dev = uintValue(1, 1000000000u);
        else if (a == "-i" || a == "--indel-rate") o.params.indel_rate = (float)(uintValue(0, 50) / 100.0);         // :355-357
        else if (a == "-ni" || a == "--no-indels") o.params.verify_matches = 0;                                      // :359
        else if (a == "-v" || a == "--verbose" || a == "-vv" || a == "--very-verbose") o.verbose = true;
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
        dy_ctx * ctx = nullptr; check(dy_ctx_create(0, &ctx));
        dy_ibf * ibf = nullptr;
        uint32_t nBins = 0;
        if (opt.filterType == "bloom" || opt.filterType == "kmer_direct")
        {   // reference src/d_mapper.h:910-936
            check(opt.filterType == "bloom" ? dy_ibf_load(ctx, opt.filterFile.c_str(), &ibf) : dy_kdx_load(ctx, opt.filterFile.c_str(), &ibf));
            dy_ibf_info info; check(dy_ibf_get_info(ibf, &info));
            nBins = info.n_bins;                                                                                // readFilterMetadata :532-548
        }
        else
        {
            opt.params.use_filter = 0;
            for (;; ++nBins) { std::ifstream f(opt.indicesDir + std::to_string(nBins) + ".txt.size"); if (!f) break; }
            if (!nBins) fail("Error while opening reference file.");
        }
        dy_index * idx = nullptr; check(dy_index_load(ctx, opt.indicesDir.c_str(), nBins, &idx));
        auto t1 = std::chrono::steady_clock::now();

        // output format by extension (reference src/d_mapper.cpp:140-153): .bam -> BGZF-compressed BAM, anything else -> SAM text
        const bool bam = opt.outputFile.size() >= 4 && opt.outputFile.compare(opt.outputFile.size() - 4, 4, ".bam") == 0;
        std::ofstream ofs;
        if (!opt.outputFile.empty()) { ofs.open(opt.outputFile, bam ? std::ios::binary : std::ios::out); if (!ofs) fail("Error while opening output file."); }
        std::ostream & os = opt.outputFile.empty() ? std::cout : ofs;
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
        BgzfWriter bgzf(os);
        const unsigned hostThreads = opt.hostThreads ? opt.hostThreads : std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        if (bam) bamWriteHeader(bgzf, headerText, contigNames, contigLengths);
        else os << headerText;

        ReadsFile in(opt.readsFile);
        std::unique_ptr<ReadsFile> in2;
        if (opt.params.paired) in2.reset(new ReadsFile(opt.readsFile2));
        if (opt.params.paired && opt.readsBatch > 1048575) opt.readsBatch = 1048575;      // both mates share the 21-bit read id space
        // Three batches are in flight: one being parsed, one on the device, one being formatted and written (in input order).
        struct Slot { ReadsBatch batch; dy_reads * reads = nullptr; dy_result * res = nullptr; size_t nPairs = 0; };
        Slot slots[3];
        unsigned long long total = 0, mapped = 0;
        double mapSeconds = 0, parseSeconds = 0, formatSeconds = 0, writeSeconds = 0;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(now() - t).count(); };
        auto readInto = [&](Slot & sl) -> bool {
            ReadsBatch & batch = sl.batch;
            const auto p0 = now();
            if (!in.next(batch, opt.readsBatch, hostThreads)) return false;
            if (in2)
            {   // readRecords of a file pair: first mates, then second mates (reference src/file_pair.h:81-88)
                ReadsBatch second;
                if (in2->next(second, batch.size(), hostThreads) != batch.size()) fail("the two reads files hold different numbers of records");
                const u64 shift = batch.codes.size();
                batch.names.insert(batch.names.end(), second.names.begin(), second.names.end());
                batch.codes.insert(batch.codes.end(), second.codes.begin(), second.codes.end());
                batch.quals.insert(batch.quals.end(), second.quals.begin(), second.quals.end());
                for (size_t i = 1; i < second.offs.size(); ++i) batch.offs.push_back(second.offs[i] + shift);
            }
            sl.nPairs = batch.size() / 2;
            parseSeconds += since(p0);
            return true;
        };
        auto formatAndWrite = [&](Slot & sl) {
            const ReadsBatch & batch = sl.batch;
            const size_t nPairs = sl.nPairs;
            const dy_read_result * rr = dy_result_reads(sl.res);
            const dy_match * ms = dy_result_matches(sl.res);
            const uint32_t * cg = dy_result_cigars(sl.res);
            // ---- alignment records (reference src/mapper_writer.h:296-345, 593-634; pairs :268-293,347-400,515-571)
            const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(hostThreads, batch.size() / 4096 + 1));
            std::vector<std::string> samChunk(T); std::vector<std::vector<uint8_t>> bamChunk(T); std::vector<unsigned long long> mappedChunk(T, 0);
            auto formatChunk = [&](unsigned c) {
            const size_t lo = batch.size() * c / T, hi = batch.size() * (c + 1) / T;
            if (bam && hi > lo) bamChunk[c].reserve((hi - lo) * (size_t)(2 * (batch.offs[lo + 1] - batch.offs[lo]) + 96));
            if (!bam && hi > lo) samChunk[c].reserve((hi - lo) * (size_t)(2 * (batch.offs[lo + 1] - batch.offs[lo]) + 96));
            AlnRecord a;                    // reused: its strings keep their capacity
            for (size_t i = lo; i < hi; ++i)
            {
                const dy_read_result & r = rr[i];
                a.flag = 0; a.rid = a.pos = -1; a.mapq = 0; a.cigar = nullptr; a.nCigar = 0; a.rnext = a.pnext = -1; a.tlen = 0;
                a.hasCounts = false; a.nm = a.x0 = a.x1 = 0; a.xa.clear();
                const std::string & name = batch.names[i];
                size_t cut = name.find_first_of(" \t/");
                a.qname.assign(name, 0, cut == std::string::npos ? name.size() : cut);
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
                    const dy_read_result & mr = rr[i < nPairs ? i + nPairs : i - nPairs];
                    a.flag |= 1u | (i < nPairs ? 64u : 128u);
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
                if (bam) bamEncodeRecord(a, bamChunk[c]);
                else samWriteRecord(samChunk[c], a, contigNames);
                mappedChunk[c] += r.mapped;
            }
            };
            const auto f0 = now();
            {
                std::vector<std::thread> pool;
                for (unsigned c = 1; c < T; ++c) pool.emplace_back(formatChunk, c);
                formatChunk(0);
                for (auto & th : pool) th.join();
            }
            formatSeconds += since(f0);
            const auto w0 = now();
            for (unsigned c = 0; c < T; ++c)
            {
                if (bam) bgzf.writeParallel(bamChunk[c].data(), bamChunk[c].size(), hostThreads);
                else os.write(samChunk[c].data(), (std::streamsize)samChunk[c].size());
                mapped += mappedChunk[c];
            }
            total += batch.size();
            writeSeconds += since(w0);
        };
        auto release = [&](Slot & sl) {          // device objects go back on the thread that drives the context
            if (sl.res) { dy_result_free(sl.res); sl.res = nullptr; }
            if (sl.reads) { dy_reads_free(sl.reads); sl.reads = nullptr; }
        };
        std::future<void> writer;                // at most one: keeps the output in input order
        int writing = -1;
        int cur = 0;
        bool have = readInto(slots[cur]);
        while (have)
        {
            Slot & sl = slots[cur];
            const int nxt = (cur + 1) % 3;
            if (writing == nxt) { writer.get(); release(slots[nxt]); writing = -1; }      // the slot about to be refilled must be written out
            std::future<bool> reader = std::async(std::launch::async, [&, nxt] { return readInto(slots[nxt]); });
            const ReadsBatch & batch = sl.batch;
            auto m0 = std::chrono::steady_clock::now();
            check(dy_reads_upload(ctx, batch.codes.data(), batch.offs.data(), (uint32_t)batch.size(), &sl.reads));
            if (ibf)
            {
                // threshold from the batch-average read length (reference src/d_mapper.h:313-315)
                uint16_t avg = (uint16_t)(batch.offs.back() * 2 / (batch.size() * 2));
                dy_ibf_info info; check(dy_ibf_get_info(ibf, &info));
                uint16_t thr = dy_ibf_threshold(opt.params.error_rate, info.kmer_size, avg);
                if (thr == 0) std::cerr << "[WARNING!] 0 k-mer is required to filter a read!\nAll reads will pass filteration and be mapped everywhere.\n";
                check(dy_ibf_query(ibf, sl.reads, thr, nullptr));
            }
            check(dy_map_reads(ctx, idx, sl.reads, &opt.params, &sl.res));
            mapSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - m0).count();
            if (writing >= 0) { writer.get(); release(slots[writing]); writing = -1; }
            writer = std::async(std::launch::async, [&, cur] { formatAndWrite(slots[cur]); });
            writing = cur;
            have = reader.get();
            cur = nxt;
        }
        if (writing >= 0) { writer.get(); release(slots[writing]); }
        if (bam) bgzf.finish();
        os.flush();
        if (opt.verbose)
        {
            dy_map_stats s; dy_map_get_stats(ctx, &s);
            std::cerr << "Loading time:\t\t\t" << std::chrono::duration<double>(t1 - t0).count() << " sec\n"
                      << "Mapping time:\t\t\t" << mapSeconds << " sec\n"
                      << "Parse / format / write time (overlapped with mapping):\t" << parseSeconds << " / " << formatSeconds << " / " << writeSeconds << " sec\n"
                      << "Total reads:\t\t\t" << total << "\nMapped reads:\t\t\t" << mapped << "\n"
                      << "Last batch: items " << s.items << ", raw matches " << s.raw_matches << ", hard items " << s.hard_items
                      << ", ms ibf/items/search/extend/finalize " << s.ms_ibf << "/" << s.ms_items << "/" << s.ms_search << "/" << s.ms_extend << "/" << s.ms_finalize << "\n";
        }
        dy_index_free(idx);
        if (ibf) dy_ibf_free(ibf);
        dy_ctx_free(ctx);
    }
    catch (std::exception const & e)
    {
        std::cerr << "dream_yara_mapper: " << e.what() << std::endl;     // reference src/d_mapper.cpp:574-582
        return 1;
    }
    return 0;
}

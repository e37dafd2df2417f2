// ORACLE — TEST INFRASTRUCTURE ONLY. CPU mapper with the dream_yara_mapper command line
// (reference src/d_mapper.cpp:114-394,554-585). Used by tests/ to pin the restatement against the gold SAMs
// and by bench.py's cpu_baseline leg. Never linked into the product.
#include "dyo_dream.hpp"
#include <iostream>
#include <chrono>
#include <memory>

using namespace dyo;

int main(int argc, char ** argv)
{
    Options opt;
    std::string filterFile, outFile, filterType = "bloom";
    std::vector<std::string> pos;
    bool verbose = false;
    try
    {
        for (int i = 1; i < argc; ++i)
        {
            std::string a = argv[i];
            auto next = [&]() -> std::string { if (i + 1 >= argc) throw std::runtime_error("missing value for " + a); return argv[++i]; };
            if (a == "-e" || a == "--error-rate") opt.errorRate = (float)(std::stoul(next()) / 100.0);
            else if (a == "-s" || a == "--strata-rate") opt.strataRate = (float)(std::stoul(next()) / 100.0);
            else if (a == "-y" || a == "--sensitivity") { std::string v = next(); opt.sensitivity = v == "low" ? LOW : v == "high" ? HIGH : FULL; }
            else if (a == "-fi" || a == "--bloom-filter") filterFile = next();
            else if (a == "-ft" || a == "--filter-type") filterType = next();
            else if (a == "-o" || a == "--output-file") outFile = next();
            else if (a == "-t" || a == "--threads") opt.threads = (unsigned)std::stoul(next());
            else if (a == "-rb" || a == "--reads-batch") opt.readsCount = (unsigned)std::stoul(next());
            else if (a == "-rg" || a == "--read-group") opt.readGroup = next();
            else if (a == "-sm" || a == "--secondary-matches") next();     // parsed under a legacy name by the reference: no-op
            else if (a == "-ll" || a == "--library-length") opt.libraryLength = (unsigned)std::stoul(next());
            else if (a == "-ld" || a == "--library-deviation") opt.libraryDev = (unsigned)std::stoul(next());
            else if (a == "-i" || a == "--indel-rate") opt.indelRate = (float)(std::stoul(next()) / 100.0);
            else if (a == "-ni" || a == "--no-indels") opt.verifyMatches = false;
            else if (a == "-v" || a == "--verbose") verbose = true;
            else if (a.size() > 1 && a[0] == '-') throw std::runtime_error("unknown option " + a);
            else pos.push_back(a);
        }
        if (pos.size() != 2 && pos.size() != 3) throw std::runtime_error("usage: oracle_mapper [OPTIONS] <INDEX DIR> <READS> [<READS2>]");
        const bool paired = pos.size() == 3;
        for (int i = 0; i < argc; ++i) { if (i) opt.commandLine += ' '; opt.commandLine += argv[i]; }
        std::string dir = pos[0];
        while (dir.size() > 1 && dir.back() == '/') dir.pop_back();
        if (filterFile.empty()) filterFile = dir + "/bloom.filter";

        auto t0 = std::chrono::steady_clock::now();
        IBF filter;
        u32 numberOfBins = 0;
        if (filterType == "bloom") { filter.load(filterFile); numberOfBins = filter.noOfBins; }
        else if (filterType == "kmer_direct") { filter.load(filterFile, true); numberOfBins = filter.noOfBins; }
        else if (filterType == "none")
        {
            opt.useFilter = false;
            while (FILE * f = fopen((dir + "/" + std::to_string(numberOfBins) + ".txt.size").c_str(), "rb")) { fclose(f); ++numberOfBins; }
        }
        else throw std::runtime_error("filter type not restated: " + filterType);
        DreamIndex idx = loadDreamIndex(dir, numberOfBins);
        auto t1 = std::chrono::steady_clock::now();

        std::ofstream ofs;
        if (!outFile.empty()) { ofs.open(outFile); if (!ofs) throw std::runtime_error("Error while opening output file."); }
        std::ostream & os = outFile.empty() ? std::cout : ofs;

        DreamMapper mapper(idx, opt.useFilter ? &filter : nullptr, opt);
        mapper.writeHeader(os);
        SeqReader reader(pos[1]);
        std::unique_ptr<SeqReader> reader2;
        if (paired) reader2.reset(new SeqReader(pos[2]));
        u64 total = 0, mapped = 0;
        DreamStats stats;
        for (;;)
        {
            SeqRecords recs;
            size_t got = reader.read(recs, opt.readsCount);
            if (!got) break;
            if (paired && reader2->read(recs, got) != got) throw std::runtime_error("the two reads files hold different numbers of records");
            ReadBatch batch;
            for (size_t i = 0; i < recs.seqs.size(); ++i)
            {
                if (recs.seqs[i].size() > 8191) throw std::runtime_error("Maximum read length exceeded.");
                appendRead(batch, recs.names[i], recs.seqs[i], recs.quals[i]);
            }
            std::vector<ReadResult> res = paired ? mapper.mapBatchPaired(batch, verbose ? &stats : nullptr) : mapper.mapBatch(batch, verbose ? &stats : nullptr);
            if (paired) mapper.writeRecordsPaired(os, batch, res); else mapper.writeRecords(os, batch, res);
            total += batch.size();
            for (auto & r : res) mapped += r.mapped;
        }
        auto t2 = std::chrono::steady_clock::now();
        if (verbose)
        {
            std::cerr << "load_s\t" << std::chrono::duration<double>(t1 - t0).count() << "\n";
            std::cerr << "map_s\t" << std::chrono::duration<double>(t2 - t1).count() << "\n";
            std::cerr << "reads\t" << total << "\nmapped\t" << mapped << "\n";
            std::cerr << "filtered_reads\t" << stats.filteredReads << "\nseeds\t" << stats.bin.seeds << "\nhits\t" << stats.bin.hits
                      << "\nlocated\t" << stats.bin.located << "\nextended\t" << stats.bin.extended << "\nmatches\t" << stats.bin.matches << "\n";
        }
    }
    catch (std::exception const & e)
    {
        std::cerr << "dream_yara_mapper: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}

// ORACLE — TEST INFRASTRUCTURE ONLY. CPU IBF builder (reference src/d_build_filter.cpp:198-272).
// usage: oracle_build_filter -b BINS -k K -nh H (-bs GiB | --bits N) -o out.filter --split genomes.fa
//   --split: record i of the FASTA is bin i (the reference test splits 64-viral-genomes.fa the same way,
//            tests/run_tests.py:83-96).  --bits is an oracle-only extension to make small test filters.
#include "dyo_indexer.hpp"
#include <iostream>
using namespace dyo;
int main(int argc, char ** argv)
{
    try
    {
        u32 bins = 64, k = 20, nh = 3; u64 bits = 0; std::string out, split, ftype = "bloom";
        for (int i = 1; i < argc; ++i)
        {
            std::string a = argv[i];
            auto next = [&]() -> std::string { if (i + 1 >= argc) throw std::runtime_error("missing value"); return argv[++i]; };
            if (a == "-b") bins = (u32)std::stoul(next());
            else if (a == "-k") k = (u32)std::stoul(next());
            else if (a == "-nh") nh = (u32)std::stoul(next());
            else if (a == "-bs") bits = std::stoull(next()) * 8589934592ull + 256;
            else if (a == "--bits") bits = std::stoull(next()) + 256;
            else if (a == "-o") out = next();
            else if (a == "-ft") ftype = next();
            else if (a == "--split") split = next();
            else throw std::runtime_error("unknown option " + a);
        }
        const bool direct = ftype == "kmer_direct";
        if ((!bits && !direct) || out.empty() || split.empty()) throw std::runtime_error("need -bs/--bits, -o, --split");
        IBF f;
        if (direct) f.createDirect(bins, k); else f.create(bins, nh, k, bits);
        SeqReader rd(split); SeqRecords r; rd.read(r, ~(size_t)0);
        for (size_t i = 0; i < r.seqs.size() && i < bins; ++i)
        {
            std::vector<u8> s(r.seqs[i].size());
            for (size_t j = 0; j < s.size(); ++j) s[j] = dna5Code(r.seqs[i][j]);
            if (s.size() < k) continue;
            ibfAddKmers(f, s.data(), s.size(), (u32)i);
        }
        f.save(out);
    }
    catch (std::exception const & e) { std::cerr << "dream_yara_build_filter: " << e.what() << std::endl; return 1; }
    return 0;
}

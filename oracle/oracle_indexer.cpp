// ORACLE — TEST INFRASTRUCTURE ONLY. CPU indexer: <fasta of one bin> -> 12 index files
// (reference src/d_indexer.cpp:389-396). usage: oracle_indexer <in.fasta> <out prefix>
//                                                oracle_indexer --split <genomes.fa> <out dir>   (one record per bin)
#include "dyo_indexer.hpp"
#include <iostream>
using namespace dyo;
static void indexRecords(const SeqRecords & r, size_t lo, size_t hi, const std::string & prefix)
{
    std::vector<std::string> names; std::vector<std::vector<u8> > seqs;
    for (size_t i = lo; i < hi; ++i)
    {
        names.push_back(r.names[i]);
        std::vector<u8> s(r.seqs[i].size());
        for (size_t k = 0; k < s.size(); ++k) s[k] = dna5Code(r.seqs[i][k]);
        seqs.push_back(s);
    }
    saveBinIndex(buildBinIndex(names, seqs), prefix);
}
int main(int argc, char ** argv)
{
    try
    {
        if (argc == 4 && std::string(argv[1]) == "--split")
        {
            SeqReader rd(argv[2]); SeqRecords r; rd.read(r, ~(size_t)0);
            for (size_t i = 0; i < r.seqs.size(); ++i) indexRecords(r, i, i + 1, std::string(argv[3]) + "/" + std::to_string(i));
            return 0;
        }
        if (argc != 3) { std::cerr << "usage: oracle_indexer <in.fasta> <out prefix>\n"; return 1; }
        SeqReader rd(argv[1]); SeqRecords r; rd.read(r, ~(size_t)0);
        indexRecords(r, 0, r.seqs.size(), argv[2]);
    }
    catch (std::exception const & e) { std::cerr << "dream_yara_indexer: " << e.what() << std::endl; return 1; }
    return 0;
}

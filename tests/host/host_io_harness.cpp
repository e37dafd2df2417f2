// Test harness for the host-side I/O of dream_yara_mapper (no GPU): built and driven by tests/test_host_io.py.
//   host_io_harness parse <reads file> <max per batch> [threads] -> one line per read: name<TAB>ACGTN string<TAB>qualities
//   host_io_harness write <out.sam> <out.bam> <n> <threads> -> the same n synthetic records as SAM text and as BGZF BAM
#include "../../dream_yara_amd/csrc/host_formats.hpp"
#include "../../dream_yara_amd/csrc/bam_writer.hpp"
#include <chrono>
#include <fstream>
#include <iostream>
#include <random>

using namespace dy;

int main(int argc, char ** argv)
{
    try
    {
        const std::string mode = argc > 1 ? argv[1] : "";
        if (mode == "parse" && (argc == 4 || argc == 5))
        {
            ReadsFile in(argv[2]);
            ReadsBatch b;
            const size_t per = std::stoul(argv[3]);
            const unsigned threads = argc == 5 ? (unsigned)std::stoul(argv[4]) : 1;
            size_t batches = 0;
            while (in.next(b, per, threads))
            {
                ++batches;
                for (size_t i = 0; i < b.size(); ++i)
                {
                    std::string seq, qual;
                    for (u64 p = b.offs[i]; p < b.offs[i + 1]; ++p) { seq += "ACGTN"[b.codes[p]]; qual += b.quals[p]; }
                    std::cout << b.name(i) << '\t' << seq << '\t' << qual << '\n';
                }
            }
            std::cerr << batches << " batches\n";
            return 0;
        }
        if (mode == "parsebench" && argc == 5)
        {   // host_io_harness parsebench <reads file> <max per batch> <threads>: time of the parse alone, and a checksum of what it produced
            ReadsFile in(argv[2]);
            ReadsBatch b;
            const size_t per = std::stoul(argv[3]);
            const unsigned threads = (unsigned)std::stoul(argv[4]);
            size_t reads = 0; u64 sum = 0;
            const auto t0 = std::chrono::steady_clock::now();
            while (in.next(b, per, threads))
            {
                reads += b.size();
                sum += b.offs.back() * 31 + b.name(b.size() - 1).size() + b.codes[b.codes.size() / 2] + (u64)b.quals[b.quals.size() / 3];
            }
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::cout << reads << " reads in " << dt << " s = " << reads / dt / 1e6 << " M reads/s, checksum " << sum
                      << " | populate " << in.phase[0] << " boundaries " << in.phase[1] << " offsets+alloc " << in.phase[2] << " convert " << in.phase[3] << "\n";
            return 0;
        }
        if (mode == "write" && argc == 6)
        {
            const size_t n = std::stoul(argv[4]);
            const unsigned threads = (unsigned)std::stoul(argv[5]);
            std::vector<std::string> names = {"chrA", "chrB", "a_rather_long_contig_name.1"};
            std::vector<uint64_t> lens = {100000, 250, 70000000};
            std::string header = "@HD\tVN:1.4\tSO:unsorted\n@RG\tID:none\tSM:none\tPG:Yara\n";
            for (size_t c = 0; c < names.size(); ++c) header += "@SQ\tSN:" + names[c] + "\tLN:" + std::to_string(lens[c]) + "\n";
            std::ofstream sam(argv[2]), bamf(argv[3], std::ios::binary);
            sam << header;
            BgzfWriter bgzf(bamf);
            bamWriteHeader(bgzf, header, names, lens);
            std::mt19937 rng(7);
            std::string samText; std::vector<uint8_t> bamBytes;
            std::vector<std::vector<uint32_t>> cigars(n);
            for (size_t i = 0; i < n; ++i)
            {
                AlnRecord a;
                a.qname = "read" + std::to_string(i);
                const unsigned L = 30 + rng() % 120;
                a.seq.resize(L); a.qual.resize(L);
                for (unsigned k = 0; k < L; ++k) { a.seq[k] = "ACGTN"[rng() % 5]; a.qual[k] = (char)(33 + rng() % 60); }
                const bool mapped = rng() % 10 != 0;
                if (mapped)
                {
                    a.rid = (int32_t)(rng() % 3); a.pos = (int32_t)(rng() % (lens[a.rid] > 200 ? lens[a.rid] - 200 : 1)); a.mapq = rng() % 71;
                    a.flag = (rng() & 1) ? 16 : 0;
                    unsigned left = L; auto & cg = cigars[i];
                    if (rng() % 4 == 0 && L > 20) { unsigned m1 = 5 + rng() % (L - 15), g = 1 + rng() % 3; cg = {(m1 << 4) | 0u, (g << 4) | (1u + rng() % 2), 0}; left = L - m1 - ((cg[1] & 15) == 1 ? g : 0); cg[2] = (left << 4) | 0u; }
                    else cg = {(L << 4) | 0u};
                    a.cigar = cg.data(); a.nCigar = (uint32_t)cg.size();
                    a.hasCounts = true; a.nm = rng() % 8; a.x0 = 1 + rng() % 3; a.x1 = rng() % 4;
                    if (rng() % 5 == 0) a.xa = "chrB,12,112,+,2;chrA,99,199,-,3;";
                    if (rng() % 3 == 0) { a.flag |= 1 | 64; a.rnext = (rng() & 1) ? a.rid : (int32_t)(rng() % 3); a.pnext = (int32_t)(rng() % 200); a.tlen = (int32_t)(rng() % 1000) - 500; }
                }
                else a.flag = 4;
                a.rg = "none";
                samWriteRecord(samText, a, names);
                bamEncodeRecord(a, bamBytes);
            }
            sam << samText;
            bgzf.writeParallel(bamBytes.data(), bamBytes.size(), threads);
            bgzf.finish();
            return 0;
        }
        std::cerr << "usage: host_io_harness parse <reads> <per-batch> | write <out.sam> <out.bam> <n> <threads>\n";
        return 2;
    }
    catch (std::exception const & e) { std::cerr << "host_io_harness: " << e.what() << std::endl; return 1; }
}

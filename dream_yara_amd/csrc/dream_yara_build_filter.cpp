// dream_yara_build_filter — builds the Interleaved Bloom filter on the GPU (reference src/d_build_filter.cpp:106-323).
//   dream_yara_build_filter [-b BINS] [-k K] [-nh H] [-bs GiB] [-o OUT.filter] <DIR with 0.fna 1.fna ...>
// Every k-mer insertion is one atomic OR issued by k_ibf_insert; the file written is the reference's sdsl bit_vector.
#include "../../include/dyhip.h"
#include "host_index_build.hpp"
#include <iostream>

using namespace dy;

int main(int argc, char ** argv)
{
    try
    {
        unsigned bins = 64, k = 20, nh = 3; unsigned long long bs = 1, rawBits = 0;
        std::string out, dir, ftype = "bloom";
        for (int i = 1; i < argc; ++i)
        {
            std::string a = argv[i];
            auto value = [&]() -> std::string { if (i + 1 >= argc) throw IoError("option requires an argument: " + a); return argv[++i]; };
            if (a == "-b" || a == "--number-of-bins") bins = (unsigned)std::stoul(value());
            else if (a == "-k" || a == "--kmer-size") k = (unsigned)std::stoul(value());
            else if (a == "-nh" || a == "--num-hash") nh = (unsigned)std::stoul(value());
            else if (a == "-bs" || a == "--bloom-size") bs = std::stoull(value());
            else if (a == "--bits") rawBits = std::stoull(value());          // extension: exact size in bits (small test filters)
            else if (a == "-o" || a == "--output-file") out = value();
            else if (a == "-t" || a == "--threads") value();
            else if (a == "-ft" || a == "--filter-type") { ftype = value(); if (ftype != "bloom" && ftype != "kmer_direct") throw IoError("filter type must be bloom or kmer_direct"); }
            else if (a == "-v" || a == "--verbose") {}
            else if (a.size() > 1 && a[0] == '-') throw IoError("unknown option " + a);
            else dir = a;
        }
        if (dir.empty()) throw IoError("usage: dream_yara_build_filter [OPTIONS] <REFERENCE FILES DIR>");
        if (dir.back() != '/') dir += '/';
        if (out.empty()) out = dir + "bloom.filter";
        if (k < 14 || k > 32) throw IoError("kmer-size must be within [14,32]");
        if (nh < 2 || nh > 5) throw IoError("num-hash must be within [2,5]");
        if (!rawBits && (bs & (bs - 1))) { std::cerr << "[ERROR] --bloom-size (-bs) parameter should be a power of 2!" << std::endl; return 1; }
        unsigned long long bits = rawBits ? rawBits + 256 : bs * 8589934592ull + 256;     // reference src/d_build_filter.cpp:203
        // commonExtension (reference src/d_misc_options.h:196-222)
        std::string ext;
        for (const char * e : {".fa", ".fasta", ".fna", ".ffn", ".frn", ".faa", ".fas"})
        {
            unsigned c = 0;
            for (; c < bins; ++c) { FILE * f = fopen((dir + std::to_string(c) + e).c_str(), "rb"); if (!f) break; fclose(f); }
            if (c == bins) { ext = e; break; }
        }
        if (ext.empty()) { std::cerr << "The given directory:\n\t" << dir << "\ndoes not contain the fasta files of all the bins!\n"; return 1; }
        dy_ctx * ctx = nullptr; dy_ibf * f = nullptr;
        auto check = [](int rc) { if (rc != DY_OK) throw IoError(dy_last_error()); };
        check(dy_ctx_create(0, &ctx));
        if (ftype == "kmer_direct")
        {   // reference src/d_build_filter.cpp:305-312: 4^k positions; the 32-bit shift there limits k to 15
            if (k > 15) throw IoError("kmer_direct: kmer-size must be within [14,15]");
            check(dy_kdx_create(ctx, bins, k, &f));
        }
        else check(dy_ibf_create(ctx, bins, nh, k, bits, &f));
        for (unsigned b = 0; b < bins; ++b)
            for (auto & r : readFasta(dir + std::to_string(b) + ext))
                check(dy_ibf_add(f, r.codes.data(), r.codes.size(), b));
        check(dy_ibf_save(f, out.c_str()));
        dy_ibf_free(f); dy_ctx_free(ctx);
    }
    catch (std::exception const & e) { std::cerr << "dream_yara_build_filter: " << e.what() << std::endl; return 1; }
    return 0;
}

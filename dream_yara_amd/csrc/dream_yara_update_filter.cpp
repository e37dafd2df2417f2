// dream_yara_update_filter — replaces the k-mers of some bins of an existing filter (reference src/d_update_filter.cpp:96-294).
//   dream_yara_update_filter [-ft bloom|kmer_direct] [-t N] [-v] <FILTER FILE> <4.fna> <7.fna> ...
// A file's bin number is its base name up to the first '.' (reference src/d_misc_options.h, getBinNoFromFile).  The bins are
// cleared by one streaming pass over the table on the device (k_ibf_clear_bins), refilled by k_ibf_insert, and the filter is
// written back in place.
#include "../../include/dyhip.h"
#include "host_index_build.hpp"
#include <iostream>
#include <map>

using namespace dy;

int main(int argc, char ** argv)
{
    try
    {
        std::string filterFile, ftype = "bloom";
        std::map<uint32_t, std::string> binFiles;
        for (int i = 1; i < argc; ++i)
        {
            std::string a = argv[i];
            auto value = [&]() -> std::string { if (i + 1 >= argc) throw IoError("option requires an argument: " + a); return argv[++i]; };
            if (a == "-ft" || a == "--filter-type") { ftype = value(); if (ftype != "bloom" && ftype != "kmer_direct") throw IoError("filter type must be bloom or kmer_direct"); }
            else if (a == "-t" || a == "--threads") value();
            else if (a == "-v" || a == "--verbose") {}
            else if (a.size() > 1 && a[0] == '-') throw IoError("unknown option " + a);
            else if (filterFile.empty()) filterFile = a;
            else
            {
                std::string base = a.substr(a.find_last_of('/') == std::string::npos ? 0 : a.find_last_of('/') + 1);
                std::string stem = base.substr(0, base.find('.'));
                if (stem.empty() || stem.find_first_not_of("0123456789") != std::string::npos)
                { std::cerr << "File: " << a << "\ndoesn't have a valid name\n"; return 1; }
                binFiles[(uint32_t)std::stoul(stem)] = a;
            }
        }
        if (filterFile.empty() || binFiles.empty()) throw IoError("usage: dream_yara_update_filter [OPTIONS] <BLOOM-FILTER FILE> <4.fna> <7.fna>");
        auto check = [](int rc) { if (rc != DY_OK) throw IoError(dy_last_error()); };
        dy_ctx * ctx = nullptr; dy_ibf * f = nullptr;
        check(dy_ctx_create(0, &ctx));
        check(ftype == "bloom" ? dy_ibf_load(ctx, filterFile.c_str(), &f) : dy_kdx_load(ctx, filterFile.c_str(), &f));
        dy_ibf_info info; check(dy_ibf_get_info(f, &info));
        std::vector<uint32_t> bins;
        for (auto & kv : binFiles)
        {
            if (kv.first >= info.n_bins)
            {
                std::cerr << "The provided bloom filter has only " << info.n_bins << " Bins.\nRetry after removing " << kv.second << " from arguments!" << std::endl;
                return 1;
            }
            bins.push_back(kv.first);
        }
        check(dy_ibf_clear_bins(f, bins.data(), (uint32_t)bins.size()));
        for (auto & kv : binFiles)
            for (auto & r : readFasta(kv.second))
                check(dy_ibf_add(f, r.codes.data(), r.codes.size(), kv.first));
        check(dy_ibf_save(f, filterFile.c_str()));
        dy_ibf_free(f); dy_ctx_free(ctx);
    }
    catch (std::exception const & e) { std::cerr << "dream_yara_update_filter: " << e.what() << std::endl; return 1; }
    return 0;
}

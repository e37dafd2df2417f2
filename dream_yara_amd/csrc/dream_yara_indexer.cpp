// dream_yara_indexer — builds the per-bin index files (reference src/d_indexer.cpp:127-225,389-473).
//   dream_yara_indexer [-v] [-t THREADS] -o <OUTPUT PREFIX DIR/> <0.fna> <1.fna> ...
// File names must be the bin number (reference src/d_indexer.cpp:178-190).  With a device visible the FM construction of every bin runs on it
// (dy_fm_build: suffix array, BWT, rank dictionary, sentinel rows, sampled suffix array) — one context, the bins' host work (FASTA, N
// randomisation, file writing) spread over the -t threads around it; `--sorter sa` keeps only the suffix sort there, `--sorter host` nothing;
// the files are byte-identical every way.
#include "../../include/dyhip.h"
#include "host_index_build.hpp"
#include <iostream>
#include <thread>
#include <mutex>
#include <atomic>
#include <map>

using namespace dy;

int main(int argc, char ** argv)
{
    try
    {
        std::string outPrefix, sorterName = "auto"; unsigned threads = 1; bool verbose = false;
        std::map<unsigned, std::string> files;
        for (int i = 1; i < argc; ++i)
        {
            std::string a = argv[i];
            auto value = [&]() -> std::string { if (i + 1 >= argc) throw IoError("option requires an argument: " + a); return argv[++i]; };
            if (a == "-o" || a == "--output-prefix") outPrefix = value();
            else if (a == "-t" || a == "--threads") threads = (unsigned)std::stoul(value());
            else if (a == "-td" || a == "--tmp-dir") value();
            else if (a == "-v" || a == "--verbose") verbose = true;
            else if (a == "--sorter") { sorterName = value(); if (sorterName != "auto" && sorterName != "host" && sorterName != "gpu" && sorterName != "sa") throw IoError("--sorter takes auto, host, gpu or sa"); }
            else if (a.size() > 1 && a[0] == '-') throw IoError("unknown option " + a);
            else
            {
                size_t slash = a.find_last_of('/');
                std::string base = slash == std::string::npos ? a : a.substr(slash + 1);
                size_t dot = base.find('.');
                std::string stem = base.substr(0, dot);
                if (stem.empty() || stem.find_first_not_of("0123456789") != std::string::npos)
                { std::cerr << "File: " << a << "\ndoesn't have a valid name\n"; return 1; }
                files[(unsigned)std::stoul(stem)] = a;
            }
        }
        if (outPrefix.empty() || files.empty()) throw IoError("usage: dream_yara_indexer -o <OUTPUT PREFIX> <0.fna> <1.fna> ...");
        std::vector<std::pair<unsigned, std::string> > work(files.begin(), files.end());
        // the device sorter: one context shared by the worker threads (one sort at a time; the rest of a bin's work overlaps with it)
        dy_ctx * ctx = nullptr;
        std::mutex gpuMutex;
        if (sorterName != "host" && (dy_device_count() > 0 || sorterName == "gpu" || sorterName == "sa"))
            if (dy_ctx_create(0, &ctx) != DY_OK) { if (sorterName != "auto") throw IoError(dy_last_error()); ctx = nullptr; }
        SuffixSorter sorter;
        if (ctx)
            sorter = [&](const std::vector<u32> & text, u32 sigma) {
                std::vector<u32> sa(text.size());
                std::lock_guard<std::mutex> g(gpuMutex);
                if (dy_suffix_array(ctx, text.data(), text.size(), sigma, sa.data()) != DY_OK) throw IoError(dy_last_error());
                return sa;
            };
        // with a device the whole FM construction of a bin runs there (dy_fm_build: suffix array, BWT, rank dictionary, sentinel rows, sampled
        // suffix array); `--sorter sa` keeps only the suffix sort on the device (the round-2 split), the files are byte-identical every way
        FmBuilder fmBuilder;
        if (ctx && sorterName != "sa")
            fmBuilder = [&](const std::vector<u32> & text, u32 nContigs, const std::vector<u32> & starts, BuiltBin & B) {
                const size_t n = text.size(), idBytes = nContigs <= 255 ? 1 : 2;
                B.drv.assign((n + 63) / 64 * 32, 0); B.ind.assign((n + 31) / 32 * 8, 0); B.val.assign((n / 10 + nContigs + 64) * (idBytes + 4), 0); B.drp.assign(nContigs, 0);
                uint64_t nVal = 0;
                std::lock_guard<std::mutex> g(gpuMutex);
                if (dy_fm_build(ctx, text.data(), n, nContigs, starts.data(), B.drv.data(), B.ind.data(), B.val.data(), &nVal, B.drp.data(), B.pst, &B.drs) != DY_OK)
                    throw IoError(dy_last_error());
                B.val.resize((size_t)nVal * (idBytes + 4));
            };
        if (verbose) std::cerr << "index construction: " << (fmBuilder ? "device 0" : ctx ? "suffix sort on device 0, the rest on the host" : "host") << "\n";
        std::atomic<size_t> next(0);
        std::string firstError;
        std::mutex errMutex;
        auto worker = [&]() {
            for (;;)
            {
                size_t k = next.fetch_add(1);
                if (k >= work.size()) break;
                try
                {
                    if (verbose) { std::lock_guard<std::mutex> g(errMutex); std::cerr << "[bin " << work[k].first << "] Building reference index ...\n"; }
                    saveBin(buildBin(readFasta(work[k].second), sorter, fmBuilder), outPrefix + std::to_string(work[k].first));
                }
                catch (std::exception const & e) { std::lock_guard<std::mutex> g(errMutex); if (firstError.empty()) firstError = e.what(); }
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < std::max(1u, threads); ++t) pool.emplace_back(worker);
        for (auto & t : pool) t.join();
        if (ctx) dy_ctx_free(ctx);
        if (!firstError.empty()) throw IoError(firstError);
    }
    catch (std::exception const & e) { std::cerr << "dream_yara_indexer: " << e.what() << std::endl; return 1; }
    return 0;
}

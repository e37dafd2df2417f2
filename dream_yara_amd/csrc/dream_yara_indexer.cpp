// dream_yara_indexer — builds the per-bin index files (reference src/d_indexer.cpp:127-225,389-473).
//   dream_yara_indexer [-v] [-t THREADS] -o <OUTPUT PREFIX DIR/> <0.fna> <1.fna> ...
// File names must be the bin number (reference src/d_indexer.cpp:178-190). Host tool: suffix sorting is not on the mapping path.
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
        std::string outPrefix; unsigned threads = 1; bool verbose = false;
        std::map<unsigned, std::string> files;
        for (int i = 1; i < argc; ++i)
        {
            std::string a = argv[i];
            auto value = [&]() -> std::string { if (i + 1 >= argc) throw IoError("option requires an argument: " + a); return argv[++i]; };
            if (a == "-o" || a == "--output-prefix") outPrefix = value();
            else if (a == "-t" || a == "--threads") threads = (unsigned)std::stoul(value());
            else if (a == "-td" || a == "--tmp-dir") value();
            else if (a == "-v" || a == "--verbose") verbose = true;
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
                    saveBin(buildBin(readFasta(work[k].second)), outPrefix + std::to_string(work[k].first));
                }
                catch (std::exception const & e) { std::lock_guard<std::mutex> g(errMutex); if (firstError.empty()) firstError = e.what(); }
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < std::max(1u, threads); ++t) pool.emplace_back(worker);
        for (auto & t : pool) t.join();
        if (!firstError.empty()) throw IoError(firstError);
    }
    catch (std::exception const & e) { std::cerr << "dream_yara_indexer: " << e.what() << std::endl; return 1; }
    return 0;
}

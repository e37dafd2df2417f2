// Host-only checks of csrc/multi_gpu.hpp's Exchange (no device needed: the copy transport is never reached).
//   1. exchangeRows: every rank gets the rows the others have for it, over several generations and row sizes (incl. all-zero rows);
//   2. a rank that cannot prepare its side of an exchange (injected: its receive buffer "cannot be allocated") makes EVERY rank throw
//      before any transfer is queued — nobody is left waiting for data that never comes;
//   3. ThreadBarrier::abort wakes waiting ranks.
#include "../../dream_yara_amd/csrc/multi_gpu.hpp"
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>

using namespace dy;

int main()
{
    const unsigned N = 4;
    std::vector<int> devices(N, 0);
    {   // 1. rows
        Exchange ex(devices, false);
        std::atomic<int> bad(0);
        auto work = [&](unsigned r) {
            for (unsigned round = 0; round < 6; ++round)
            {
                const size_t rowBytes = round % 2 ? 32 : 72;
                std::vector<uint8_t> mine(N * rowBytes), theirs(N * rowBytes, 0xEE);
                for (unsigned d = 0; d < N; ++d) for (size_t k = 0; k < rowBytes; ++k) mine[d * rowBytes + k] = round == 4 ? 0 : (uint8_t)(r * 16 + d + k + round);
                ex.exchangeRows(r, mine.data(), rowBytes, theirs.data());
                for (unsigned s = 0; s < N; ++s) for (size_t k = 0; k < rowBytes; ++k)
                    if (theirs[s * rowBytes + k] != (round == 4 ? 0 : (uint8_t)(s * 16 + r + k + round))) bad++;
            }
        };
        std::vector<std::thread> th;
        for (unsigned r = 0; r < N; ++r) th.emplace_back(work, r);
        for (auto & t : th) t.join();
        if (bad) { printf("rows differ: %d\n", bad.load()); return 1; }
        printf("rows ok\n");
    }
    {   // 2. injected failure at rank 1: all ranks throw, none hangs, nothing was queued (zero-byte segments: no device call is reached)
        Exchange ex(devices, false);
        ex.injectReserveFailure(1);
        std::atomic<int> threw(0), passed(0);
        auto work = [&](unsigned r) {
            RankBuf recv;                       // zero-byte exchange: reserve(16) would touch the device, so the buffer is pre-sized by hand
            static char dummy[N][64];
            recv.p = dummy[r]; recv.cap = 64;
            std::vector<size_t> sb(N, 0), rb(N, 0);
            try { ex.allToAll(r, nullptr, nullptr, sb, recv, rb); passed++; }
            catch (GpuError const &) { threw++; }
            recv.p = nullptr; recv.cap = 0;
        };
        std::vector<std::thread> th;
        for (unsigned r = 0; r < N; ++r) th.emplace_back(work, r);
        for (auto & t : th) t.join();
        if (threw != (int)N || passed != 0) { printf("collective failure: %d threw, %d passed\n", threw.load(), passed.load()); return 1; }
        printf("collective failure ok\n");
    }
    {   // 3. abort wakes the waiters
        Exchange ex(devices, false);
        std::atomic<int> woke(0);
        std::vector<std::thread> th;
        for (unsigned r = 0; r + 1 < N; ++r) th.emplace_back([&] { try { ex.barrier(); } catch (GpuError const &) { woke++; } });
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
        ex.abort();
        for (auto & t : th) t.join();
        if (woke != (int)N - 1) { printf("abort woke %d\n", woke.load()); return 1; }
        printf("abort ok\n");
    }
    return 0;
}

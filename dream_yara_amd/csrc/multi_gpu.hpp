// Multi-GPU plumbing of dream_yara_mapper (host side, C++17): one dy_ctx + one host thread per device, all in ONE process.
//   --shard reads (default): every device holds the filter and all bins; whole batches go to the devices round-robin — no exchange.
//   --shard bins            : the reference's unit of distribution (one child Mapper per bin, src/d_mapper.h:876-884): device d holds
//                             the FM structures of the bins with bin % N == d; per batch every device takes a slice of the reads
//                             (its "home" reads), and the two exchanges of SURVEY.md §8(e) run between the devices:
//                               C1  reads -> owners of their selected bins      (loadFilteredReads, src/d_mapper.h:394-443)
//                               C2  raw matches (+ pairing records) -> home     (copyMatches / copyCigars, src/d_mapper.h:127-215)
//                             through RCCL (ncclSend / ncclRecv groups over xGMI, one communicator per device, librccl loaded on
//                             demand), or — when two ranks share a device (tests) or --transport copy — by direct device-to-device copies.
// The library (libdyhip) stays collective-free: it only packs and unpacks (include/dyhip.h, dy_shard_* / dy_map_bins / dy_finalize_raw).
#pragma once
#include "../../include/dyhip.h"
#include <hip/hip_runtime_api.h>
#include <dlfcn.h>
#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <unistd.h>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

namespace dy {

struct GpuError : std::runtime_error { using std::runtime_error::runtime_error; };
inline void hipCheck(hipError_t e, const char * what) { if (e != hipSuccess) throw GpuError(std::string(what) + ": " + hipGetErrorString(e)); }

class ThreadBarrier
{
public:
    explicit ThreadBarrier(unsigned n) : n_(n) {}
    void wait()
    {
        std::unique_lock<std::mutex> lk(m_);
        if (aborted_) throw GpuError("another rank failed");
        const unsigned gen = gen_;
        if (++count_ == n_) { count_ = 0; ++gen_; cv_.notify_all(); }
        else cv_.wait(lk, [&] { return gen != gen_ || aborted_; });
        if (aborted_) throw GpuError("another rank failed");
    }
    // a rank that fails must not leave the others waiting for it
    void abort() { std::lock_guard<std::mutex> lk(m_); aborted_ = true; cv_.notify_all(); }
private:
    std::mutex m_; std::condition_variable cv_; unsigned n_, count_ = 0, gen_ = 0; bool aborted_ = false;
};

// grow-only device buffer owned by one rank
struct RankBuf
{
    void * p = nullptr; size_t cap = 0; int device = 0;
    void reserve(size_t bytes)
    {
        if (bytes <= cap) return;
        hipCheck(hipSetDevice(device), "hipSetDevice");
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        hipCheck(hipMalloc(&p, bytes + bytes / 4 + 256), "hipMalloc");
        cap = bytes + bytes / 4 + 256;
    }
    ~RankBuf() { if (p) { (void)hipSetDevice(device); (void)hipFree(p); } }
};

// the few RCCL entry points used, resolved from librccl.so at run time (a single-GPU run never loads it)
struct RcclApi
{
    typedef void * comm_t;
    int (*CommInitAll)(comm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
    const char * (*GetErrorString)(int) = nullptr;
    void * handle = nullptr;
    enum { Uint8 = 1 };                     // ncclUint8 (rccl.h: ncclInt8 = 0, ncclUint8 = 1)
    void load()
    {
        for (const char * name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) { handle = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (handle) break; }
        if (!handle) throw GpuError(std::string("cannot load librccl.so: ") + dlerror());
        auto sym = [&](const char * n) { void * s = dlsym(handle, n); if (!s) throw GpuError(std::string("librccl.so lacks ") + n); return s; };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll"); CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart"); GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        Send = (decltype(Send))sym("ncclSend"); Recv = (decltype(Recv))sym("ncclRecv"); GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
    }
    void check(int rc, const char * what) const { if (rc != 0) throw GpuError(std::string(what) + ": " + (GetErrorString ? GetErrorString(rc) : "RCCL error")); }
};

// Bin -> rank assignment, static per job (the FM structures stay resident): longest-processing-time-first on the bins' text sizes
// (<dir>/<b>.txt.size: contigsSum), the static counterpart of sortedBins() (reference src/d_mapper.h:843-854), which orders the bins of
// every batch by their read counts: the largest bin goes to the least loaded rank (ties: lowest rank), so equal bins fall out as
// bin mod N.  sizes empty or all equal -> bin mod N.
inline std::vector<uint32_t> assignBins(const std::vector<uint64_t> & sizes, unsigned nRanks)
{
    const uint32_t nBins = (uint32_t)sizes.size();
    std::vector<uint32_t> order(nBins), owner(nBins, 0);
    for (uint32_t b = 0; b < nBins; ++b) order[b] = b;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return sizes[x] > sizes[y]; });
    std::vector<uint64_t> load(nRanks, 0);
    for (uint32_t b : order)
    {
        unsigned best = 0;
        for (unsigned r = 1; r < nRanks; ++r) if (load[r] < load[best]) best = r;
        owner[b] = best; load[best] += std::max<uint64_t>(sizes[b], 1);
    }
    return owner;
}

// all-to-all(v) of bytes between the N rank threads of this process.  The sizes travel through host memory (the threads share an
// address space: one barrier, no collective), the segments through ONE RCCL group per rank and exchange — stream-ordered, nothing
// waits for them — or, when two ranks share a device (tests) or --transport copy, by device-to-device copies.
class Exchange
{
public:
    Exchange(const std::vector<int> & devices, bool useRccl) : n_((unsigned)devices.size()), devices_(devices), barrier_((unsigned)devices.size()),
        sendPtr_(devices.size(), nullptr), useRccl_(useRccl)
    {
        if (useRccl_)
        {
            rccl_.load();
            comms_.assign(n_, nullptr);
            rccl_.check(rccl_.CommInitAll(comms_.data(), (int)n_, devices_.data()), "ncclCommInitAll");
        }
    }
    ~Exchange() { if (useRccl_) for (auto c : comms_) if (c) (void)rccl_.CommDestroy(c); }
    bool rccl() const { return useRccl_; }
    unsigned ranks() const { return n_; }

    // Every rank gives the N rows it has for ranks 0..N-1 (host memory, rowBytes each) and gets the rows the others have for it.
    // Rows live in one of two generations of a shared matrix: a rank can be at most one exchange ahead of the slowest one (each
    // exchange contains a barrier), so the rows of exchange g are read before anybody writes generation g + 2.
    void exchangeRows(unsigned rank, const void * mine, size_t rowBytes, void * theirs)
    {
        std::vector<uint8_t> & M = rows_[gen_[rank] & 1];
        {
            std::lock_guard<std::mutex> lk(m_);
            if (M.size() < (size_t)n_ * n_ * rowBytes) M.resize((size_t)n_ * n_ * rowBytes);
        }
        barrier_.wait();                                    // the matrix has its size (and the previous user of this generation is done with it)
        memcpy(M.data() + (size_t)rank * n_ * rowBytes, mine, (size_t)n_ * rowBytes);
        barrier_.wait();                                    // every rank has published
        for (unsigned s = 0; s < n_; ++s) memcpy((uint8_t *)theirs + s * rowBytes, M.data() + ((size_t)s * n_ + rank) * rowBytes, rowBytes);
        ++gen_[rank];
    }

    // A failure of one rank between the two halves of an exchange must not leave its peers waiting for data that never comes: every rank
    // reports whether it is ready to transfer (its receive buffer is allocated), and ALL ranks throw if any is not — before any transfer
    // is queued.
    void agree(unsigned rank, bool ok)
    {
        (void)rank;
        { std::lock_guard<std::mutex> lk(m_); if (!ok) ++failed_; }
        barrier_.wait();
        const bool bad = failed_ != 0;
        barrier_.wait();                                    // everybody has read the verdict
        if (bad) throw GpuError("a rank could not prepare its side of the exchange");
    }

    // Called by every rank thread with its own arguments.  `send`: device buffer, the segments for ranks 0..N-1 back to back, sendBytes[d] each;
    // `recv`: the segments of ranks 0..N-1 back to back, recvBytes[s] = bytes from rank s (both known from exchangeRows).  The transfer is
    // queued on `st`; with RCCL nothing waits for it (work queued on `st` afterwards sees the data, and the send buffer may be overwritten
    // by work queued on `st`).  An error inside the RCCL section cannot be made collective any more — peers may already be waiting on the
    // device — so it ends the process (exit code 1): never retried.
    void allToAll(unsigned rank, hipStream_t st, const void * send, const std::vector<size_t> & sendBytes, RankBuf & recv, const std::vector<size_t> & recvBytes)
    {
        size_t total = 0;
        for (unsigned s = 0; s < n_; ++s) total += recvBytes[s];
        bool ok = true;
        try { recv.reserve(total + 16); } catch (std::exception const &) { ok = false; }
        if (failNextReserve_ == (int)rank) { ok = false; failNextReserve_ = -1; }        // fault injection (tests/native/exchange_fail.cpp)
        sendPtr_[rank] = send;
        { std::lock_guard<std::mutex> lk(m_); if (sendOffs_.size() != (size_t)n_ * n_) sendOffs_.assign((size_t)n_ * n_, 0); }
        agree(rank, ok);
        if (useRccl_)
        {
            try
            {
                rccl_.check(rccl_.GroupStart(), "ncclGroupStart");
                size_t so = 0, ro = 0;
                for (unsigned p = 0; p < n_; ++p)
                {
                    if (sendBytes[p]) rccl_.check(rccl_.Send((const char *)send + so, sendBytes[p], RcclApi::Uint8, (int)p, comms_[rank], st), "ncclSend");
                    if (recvBytes[p]) rccl_.check(rccl_.Recv((char *)recv.p + ro, recvBytes[p], RcclApi::Uint8, (int)p, comms_[rank], st), "ncclRecv");
                    so += sendBytes[p]; ro += recvBytes[p];
                }
                rccl_.check(rccl_.GroupEnd(), "ncclGroupEnd");
            }
            catch (std::exception const & e)
            {
                fprintf(stderr, "dream_yara_mapper: rank %u: %s — peers may already wait for this transfer on their devices; exiting\n", rank, e.what());
                fflush(stderr);
                barrier_.abort();
                _exit(1);
            }
        }
        else
        {   // pull: every rank copies its segment out of each source's buffer (peer copy over xGMI, or inside one device)
            size_t acc = 0;
            for (unsigned d = 0; d < n_; ++d) { sendOffs_[(size_t)rank * n_ + d] = acc; acc += sendBytes[d]; }
            barrier_.wait();                                // every rank has published its buffer and offsets, and its packing is queued ...
            hipCheck(hipStreamSynchronize(st), "hipStreamSynchronize (exchange)");
            barrier_.wait();                                // ... and done
            size_t ro = 0;
            for (unsigned s = 0; s < n_; ++s)
            {
                if (recvBytes[s]) hipCheck(hipMemcpyAsync((char *)recv.p + ro, (const char *)sendPtr_[s] + sendOffs_[(size_t)s * n_ + rank], recvBytes[s], hipMemcpyDefault, st), "hipMemcpyAsync (exchange)");
                ro += recvBytes[s];
            }
            hipCheck(hipStreamSynchronize(st), "hipStreamSynchronize (exchange)");
            barrier_.wait();                                // the send buffers may be reused from here on
        }
    }
    void barrier() { barrier_.wait(); }
    void abort() { barrier_.abort(); }
    void injectReserveFailure(int rank) { failNextReserve_ = rank; }

private:
    unsigned n_;
    std::vector<int> devices_;
    ThreadBarrier barrier_;
    std::vector<const void *> sendPtr_;
    std::vector<size_t> sendOffs_;
    std::vector<uint8_t> rows_[2];
    unsigned gen_[64] = {};
    unsigned failed_ = 0;
    int failNextReserve_ = -1;
    std::mutex m_;
    bool useRccl_;
    RcclApi rccl_;
    std::vector<RcclApi::comm_t> comms_;
};

// pinned host memory for the rows a rank fetches from its device (two small copies per batch)
struct PinnedRows
{
    void * p = nullptr; size_t cap = 0;
    void reserve(size_t bytes) { if (bytes <= cap) return; if (p) (void)hipHostFree(p); p = nullptr; cap = 0; hipCheck(hipHostMalloc(&p, bytes + 256, hipHostMallocDefault), "hipHostMalloc"); cap = bytes + 256; }
    ~PinnedRows() { if (p) (void)hipHostFree(p); }
};

// One rank of a bins-sharded job: the per-batch sequence of include/dyhip.h ("bins sharded over ranks").
struct ShardRank
{
    unsigned rank = 0, nRanks = 1; int device = 0;
    dy_ctx * ctx = nullptr; dy_ibf * ibf = nullptr; dy_index * idx = nullptr;
    uint32_t nBins = 0, W = 0;
    std::vector<uint64_t> ownerMasks;        // nRanks x W
    RankBuf rowsDev, recv1, recv2;
    PinnedRows rowsHost;

    static void check(int rc) { if (rc != DY_OK) throw GpuError(dy_last_error()); }

    // `reads`: this rank's home reads, bin masks already on them (dy_ibf_query, or every bin for -ft none).  Returns the home result.
    // Host synchronisations per batch: one per exchange (the sizes), those inside dy_map_bins, one at the end.
    dy_result * mapBatch(Exchange & ex, dy_reads * reads, const dy_params & prm)
    {
        hipStream_t st = (hipStream_t)dy_ctx_stream(ctx);
        const bool paired = prm.paired != 0;
        for (RankBuf * b : {&rowsDev, &recv1, &recv2}) b->device = device;
        hipCheck(hipSetDevice(device), "hipSetDevice");
        const size_t row2 = dy_shard_c2_row_bytes(nBins), rowMax = std::max(row2, sizeof(dy_seg_header));
        rowsDev.reserve(nRanks * rowMax); rowsHost.reserve(2 * nRanks * rowMax);
        uint8_t * mine = (uint8_t *)rowsHost.p, * theirs = mine + nRanks * rowMax;
        std::vector<size_t> sb(nRanks), rb(nRanks);

        // [C1] reads -> owners of their selected bins
        void * send = nullptr;
        check(dy_shard_c1_pack(ctx, reads, nRanks, ownerMasks.data(), paired ? 1u : 0u, rowsDev.p, &send));
        hipCheck(hipMemcpyAsync(mine, rowsDev.p, nRanks * sizeof(dy_seg_header), hipMemcpyDeviceToHost, st), "hipMemcpyAsync (headers)");
        hipCheck(hipStreamSynchronize(st), "hipStreamSynchronize (headers)");
        ex.exchangeRows(rank, mine, sizeof(dy_seg_header), theirs);
        std::vector<dy_seg_header> in(nRanks);
        memcpy(in.data(), theirs, nRanks * sizeof(dy_seg_header));
        for (unsigned d = 0; d < nRanks; ++d) { sb[d] = (size_t)((const dy_seg_header *)mine)[d].bytes; rb[d] = (size_t)in[d].bytes; }
        ex.allToAll(rank, st, send, sb, recv1, rb);
        dy_reads * sub = nullptr;
        check(dy_shard_c1_unpack(ctx, recv1.p, nRanks, in.data(), nBins, paired ? 1u : 0u, &sub));
        int rc = dy_map_bins(ctx, idx, sub, &prm, nullptr, nullptr);
        if (rc != DY_OK) { dy_reads_free(sub); check(rc); }

        // [C2] raw matches (+ pairing records) -> the read's home rank; the rows also carry the items per bin (sortedBins(), reference src/d_mapper.h:843-854)
        rc = dy_shard_c2_pack(ctx, rowsDev.p, &send);
        dy_reads_free(sub);
        check(rc);
        hipCheck(hipMemcpyAsync(mine, rowsDev.p, nRanks * row2, hipMemcpyDeviceToHost, st), "hipMemcpyAsync (rows)");
        hipCheck(hipStreamSynchronize(st), "hipStreamSynchronize (rows)");
        ex.exchangeRows(rank, mine, row2, theirs);
        for (unsigned d = 0; d < nRanks; ++d)
        {
            dy_seg_header h; memcpy(&h, mine + d * row2, sizeof(h)); sb[d] = (size_t)h.bytes;
            memcpy(&h, theirs + d * row2, sizeof(h)); rb[d] = (size_t)h.bytes;
        }
        ex.allToAll(rank, st, send, sb, recv2, rb);
        dy_result * res = nullptr;
        check(dy_shard_finalize(ctx, idx, reads, &prm, recv2.p, nRanks, theirs, &res));
        rc = dy_result_fetch(res);
        if (rc != DY_OK) { dy_result_free(res); check(rc); }
        return res;
    }
};

} // namespace dy

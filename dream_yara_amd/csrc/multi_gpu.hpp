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
#include <condition_variable>
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

// all-to-all(v) of bytes between the N rank threads of this process
class Exchange
{
public:
    Exchange(const std::vector<int> & devices, bool useRccl) : n_((unsigned)devices.size()), devices_(devices), barrier_((unsigned)devices.size()),
        sendPtr_(devices.size(), nullptr), counts_(devices.size() * devices.size(), 0), sums_(), useRccl_(useRccl)
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

    // Called by every rank thread with its own arguments.  `send`: device buffer, the segments for ranks 0..N-1 back to back, sendBytes[d] each.
    // `recv` receives the segments of ranks 0..N-1 back to back; recvBytes[s] = bytes that came from rank s.
    void allToAll(unsigned rank, hipStream_t st, const void * send, const std::vector<size_t> & sendBytes, RankBuf & recv, std::vector<size_t> & recvBytes)
    {
        sendPtr_[rank] = send;
        for (unsigned d = 0; d < n_; ++d) counts_[rank * n_ + d] = sendBytes[d];
        barrier_.wait();                                    // every rank has published its buffer and counts
        recvBytes.assign(n_, 0);
        size_t total = 0;
        for (unsigned s = 0; s < n_; ++s) { recvBytes[s] = counts_[s * n_ + rank]; total += recvBytes[s]; }
        recv.reserve(total + 16);
        if (useRccl_)
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
        else
        {   // pull: every rank copies its segment out of each source's buffer (peer copy over xGMI, or inside one device)
            size_t ro = 0;
            for (unsigned s = 0; s < n_; ++s)
            {
                size_t so = 0;
                for (unsigned d = 0; d < rank; ++d) so += counts_[s * n_ + d];
                if (recvBytes[s]) hipCheck(hipMemcpyAsync((char *)recv.p + ro, (const char *)sendPtr_[s] + so, recvBytes[s], hipMemcpyDefault, st), "hipMemcpyAsync (exchange)");
                ro += recvBytes[s];
            }
        }
        hipCheck(hipStreamSynchronize(st), "hipStreamSynchronize (exchange)");
        barrier_.wait();                                    // the send buffers may be reused from here on
    }

    // element-wise sum over the ranks (host values; every rank gets the result)
    void allReduceSum(unsigned rank, std::vector<uint32_t> & v)
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            if (sums_.size() != v.size()) sums_.assign(v.size(), 0);
            for (size_t i = 0; i < v.size(); ++i) sums_[i] += v[i];
        }
        barrier_.wait();
        v = sums_;
        barrier_.wait();
        if (rank == 0) sums_.assign(sums_.size(), 0);
        barrier_.wait();
    }
    void barrier() { barrier_.wait(); }
    void abort() { barrier_.abort(); }

private:
    unsigned n_;
    std::vector<int> devices_;
    ThreadBarrier barrier_;
    std::vector<const void *> sendPtr_;
    std::vector<size_t> counts_;
    std::vector<uint32_t> sums_;
    std::mutex m_;
    bool useRccl_;
    RcclApi rccl_;
    std::vector<RcclApi::comm_t> comms_;
};

// One rank of a bins-sharded job: the per-batch sequence of include/dyhip.h ("bins sharded over ranks").
struct ShardRank
{
    unsigned rank = 0, nRanks = 1; int device = 0;
    dy_ctx * ctx = nullptr; dy_ibf * ibf = nullptr; dy_index * idx = nullptr;
    uint32_t nBins = 0, W = 0;
    std::vector<uint64_t> ownerMasks;        // nRanks x W
    RankBuf ids, lens, masks, seqs, rIds, rLens, rMasks, rSeqs, raw, rRaw, items, rItems;

    static void check(int rc) { if (rc != DY_OK) throw GpuError(dy_last_error()); }

    // `reads`: this rank's home reads, bin masks already on them (dy_ibf_query, or every bin for -ft none).  Returns the home result.
    dy_result * mapBatch(Exchange & ex, dy_reads * reads, const dy_params & prm)
    {
        hipStream_t st = (hipStream_t)dy_ctx_stream(ctx);
        const bool paired = prm.paired != 0;
        std::vector<uint32_t> nTo(nRanks); std::vector<uint64_t> bTo(nRanks);
        check(dy_shard_plan(ctx, reads, nRanks, ownerMasks.data(), paired ? 1u : 0u, nTo.data(), bTo.data()));
        size_t nTot = 0, bTot = 0;
        for (unsigned d = 0; d < nRanks; ++d) { nTot += nTo[d]; bTot += bTo[d]; }
        for (RankBuf * b : {&ids, &lens, &masks, &seqs, &rIds, &rLens, &rMasks, &rSeqs, &raw, &rRaw, &items, &rItems}) b->device = device;
        ids.reserve(nTot * 4 + 16); lens.reserve(nTot * 4 + 16); masks.reserve(nTot * W * 8 + 16); seqs.reserve(bTot + 16);
        check(dy_shard_pack(ctx, reads, ids.p, lens.p, masks.p, seqs.p));
        std::vector<size_t> sb(nRanks), rb;
        for (unsigned d = 0; d < nRanks; ++d) sb[d] = (size_t)nTo[d] * 4;
        std::vector<size_t> rIdBytes;
        ex.allToAll(rank, st, ids.p, sb, rIds, rIdBytes);
        ex.allToAll(rank, st, lens.p, sb, rLens, rb);
        for (unsigned d = 0; d < nRanks; ++d) sb[d] = (size_t)nTo[d] * W * 8;
        ex.allToAll(rank, st, masks.p, sb, rMasks, rb);
        for (unsigned d = 0; d < nRanks; ++d) sb[d] = (size_t)bTo[d];
        ex.allToAll(rank, st, seqs.p, sb, rSeqs, rb);
        size_t nRecv = 0, basesRecv = 0;
        std::vector<uint32_t> seg(nRanks + 1, 0);
        for (unsigned s = 0; s < nRanks; ++s) { nRecv += rIdBytes[s] / 4; seg[s + 1] = (uint32_t)nRecv; basesRecv += rb[s]; }
        dy_reads * sub = nullptr;
        if (paired) check(dy_reads_from_device_paired(ctx, rSeqs.p, rLens.p, rMasks.p, (uint32_t)nRecv, basesRecv, nBins, nRanks, seg.data(), &sub));
        else check(dy_reads_from_device(ctx, rSeqs.p, rLens.p, rMasks.p, (uint32_t)nRecv, basesRecv, nBins, &sub));
        std::vector<uint32_t> loads(nBins, 0); uint64_t nRaw = 0;
        check(dy_map_bins(ctx, idx, sub, &prm, loads.data(), &nRaw));
        raw.reserve(nRaw * DY_RAW_MATCH_BYTES + 16);
        std::vector<uint64_t> rawTo(nRanks, 0), itemsTo(nRanks, 0);
        check(dy_raw_pack(ctx, nRanks, seg.data(), rIds.p, raw.p, rawTo.data()));
        size_t nItemsIn = 0;
        if (paired)
        {
            size_t nItems = 0; for (uint32_t l : loads) nItems += l;
            items.reserve(nItems * DY_PE_ITEM_BYTES + 16);
            check(dy_pe_items_pack(ctx, nRanks, seg.data(), rIds.p, items.p, itemsTo.data()));
            for (unsigned d = 0; d < nRanks; ++d) sb[d] = (size_t)itemsTo[d] * DY_PE_ITEM_BYTES;
            ex.allToAll(rank, st, items.p, sb, rItems, rb);
            for (size_t b : rb) nItemsIn += b / DY_PE_ITEM_BYTES;
        }
        dy_reads_free(sub);
        for (unsigned d = 0; d < nRanks; ++d) sb[d] = (size_t)rawTo[d] * DY_RAW_MATCH_BYTES;
        ex.allToAll(rank, st, raw.p, sb, rRaw, rb);
        size_t nRawIn = 0; for (size_t b : rb) nRawIn += b / DY_RAW_MATCH_BYTES;
        ex.allReduceSum(rank, loads);                          // sortedBins() over the whole job (reference src/d_mapper.h:843-854)
        dy_result * res = nullptr;
        check(dy_finalize_raw(ctx, idx, reads, &prm, rRaw.p, nRawIn, nItemsIn ? rItems.p : nullptr, nItemsIn, loads.data(), &res));
        check(dy_result_fetch(res));
        return res;
    }
};

} // namespace dy

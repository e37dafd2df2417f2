// libdyhip.so — C ABI (include/dyhip.h) over the gfx950 kernels. Host code is plumbing only: file parsing, HBM
// allocation, kernel launches on one HIP stream, result mirroring. No compute falls back to the host.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <hipcub/hipcub.hpp>
#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <vector>
#include <algorithm>
#include <atomic>
#include <cstddef>
#include <thread>

#include "../../include/dyhip.h"
#include "dy_types.h"
#include "host_formats.hpp"
#include "ibf_kernels.h"
#include "map_kernels.h"
#include "finalize_kernels.h"
#include "pe_kernels.h"
#include "shard_kernels.h"
#include "sa_kernels.h"
#include "sam_kernels.h"

using namespace dy;

namespace {

thread_local std::string g_lastError;

struct DyError
{
    int code; std::string msg;
    DyError(int c, std::string m) : code(c), msg(std::move(m)) {}
};

#define HIP_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    throw DyError(e_ == hipErrorOutOfMemory ? DY_ENOMEM : DY_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

template <typename F> int guarded(F && f)
{
    try { f(); return DY_OK; }
    catch (DyError const & e) { g_lastError = e.msg; return e.code; }
    catch (IoError const & e) { g_lastError = e.what(); return DY_EIO; }
    catch (std::bad_alloc const &) { g_lastError = "host allocation failed"; return DY_ENOMEM; }
    catch (std::exception const & e) { g_lastError = e.what(); return DY_EINVAL; }
}

// grow-only device buffer
struct DevBuf
{
    void * p = nullptr; size_t cap = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf & operator=(const DevBuf &) = delete;
    void reserve(size_t bytes)
    {
        if (bytes <= cap) return;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        HIP_CHECK(hipMalloc(&p, want));
        cap = want;
    }
    // no growth slack: objects of tens of GiB that never grow (the filter)
    void reserveExact(size_t bytes)
    {
        if (bytes <= cap) return;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        HIP_CHECK(hipMalloc(&p, bytes + 256));
        cap = bytes + 256;
    }
    template <typename T> T * as() const { return (T *)p; }
    void moveTo(DevBuf & o) { if (o.p) (void)hipFree(o.p); o.p = p; o.cap = cap; p = nullptr; cap = 0; }
};

// grow-only pinned host buffer (D2H of the results runs at PCIe speed instead of through the pageable-memory staging path)
struct PinnedBuf
{
    void * p = nullptr; size_t cap = 0;
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf & operator=(const PinnedBuf &) = delete;
    void reserve(size_t bytes)
    {
        if (bytes <= cap) return;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        HIP_CHECK(hipHostMalloc(&p, want, hipHostMallocDefault));
        cap = want;
    }
    template <typename T> T * as() const { return (T *)p; }
    void moveTo(PinnedBuf & o) { if (o.p) (void)hipHostFree(o.p); o.p = p; o.cap = cap; p = nullptr; cap = 0; }
};

int deviceCount()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

} // namespace

// ctx->counters (u32 slots, zeroed before K3): 0 items k_extend's fast pass left to the full form, 1 pool overflow flag, 2 items of the wavefront-per-item list, 3 complex reads,
// 4 align list, 5 small list, 6 DP list / rescue jobs, 7 bad raw record (dy_finalize_raw), 8 PE child list, 9 PE align list, 10 PE small list,
// 11 PE heavy pairs, 12 busy-but-not-re-seeded items (= slot 2 + 10, see extendEasyItem), 13 inconsistent record met by k_group_matches,
// 14-15 the 64-bit raw-match count, 16 small list / 17 DP list of the wide bands (a match with more than 3 errors)
enum { COUNTER_BYTES = 128, COUNTER_BAD_GROUP = 13, COUNTER_POOL64 = 14, COUNTER_SMALL_WIDE = 16, COUNTER_DP_WIDE = 17 };

// Explicit knobs (dy_ctx_set_tuning): which of several result-identical code paths / table sets a context uses.  The defaults are
// the measured best; the parity tests force the other paths through this, nothing on the mapping path reads the environment.
struct Tuning
{
    u64 sparseSa = 0;           // 1 = keep the on-disk SA sampling (no expanded SA)
    u64 prefixQ = 12;           // q of the q-mer range table (0 = none); capped by the free HBM at load
    u64 markers = 0;            // 1 = roctx ranges around the stages (dy:sweep, dy:items, dy:search, dy:extend, dy:finalize) for rocprofv3 --marker-trace
    u64 extendMode = 3;         // K3: 3 = 1 or 2 by the share of items the last fast pass could not finish (see launchExtend); 1 fast pass + a second launch
                                // over the listed rest (k_extend MODE 1 + 0), 0 fast pass + the block's own rest in full (MODE 2), 2 one full pass (MODE 0)
    u64 prefixWide = 1;         // 16-byte table entries that carry the bases after a unique q-mer (k_prefix_wide), when they fit
    u64 noIsa = 0;              // 1 = no inverse SA / indexed text (pure LF walk)
    u64 finalizeBlockMin = 2048;// reads with more raw matches are aggregated by a 1024-thread block
    u64 finalizeNoKeys = 0;     // 1 = block aggregation sorts indices instead of 128-bit keys
    u64 peHeavyMin = 256;       // pairs with more unique matches are paired by a wavefront
    u64 peRescueGlobal = 0;     // 1 = mate-rescue DP rows in global memory instead of LDS
    u64 ibfNarrow = 0;          // 1 = lane-per-k-mer sweep also for W > 1
    u64 ibfFullTail = 0;        // 1 = k_ibf_query<7> instead of k_ibf_query_two
    u64 ibfLdsKb = 12;          // LDS per wave of k_ibf_query_wide
    u64 ibfDense = 0;           // 1 = no sparse verification in k_ibf_query_wide
    u64 ibfKernel = 0;          // sweep for W > 1 and reads of at most 255 k-mers: 0 = by expected false candidates, 1 = k_ibf_query_lazy, 2 = k_ibf_query_eager2, 3 = k_ibf_query_wide
    u64 ibfOwnStream = 0;       // 1 = dy_ibf_query runs on the context's second stream (a caller may queue the NEXT batch's sweep before mapping the current one)
    u64 ibfSlack = 0;           // first check of the sweeps = nK - threshold + slack k-mers (0 = derived from the filter's fill)
    u64 poolCap = 0;            // initial capacity of the raw-match pool (0 = 2 x items); it grows on overflow either way
    u64 dpSmallBlocks = 4096;   // grid of the thread-level aggregation kernels (k_finalize_small, k_pe_*): list-driven, grid-stride
    u64 dpAlignBlocks = 8192;   // grid of k_align_single
    u64 itemsSorted = 1;        // work list: 1 = (bin, read) pairs in read order + stable radix sort by bin, 0 = bin-major count matrix + scan + scatter
    u64 asyncMap = 0;           // 1 = single-end batches after a context's first are queued whole (work list and pool sized from bounds, the real sizes read on the device, ONE fetch at the end);
                                // 0 = the host waits for the work list's size and the pool counter.  Measured (config 2): 5.24 ms per step queued whole, 5.12 ms waiting — see DESIGN.md section 8
    u64 sweepChunk = 0;         // reads per chunk of a deferred sweep queued in front of each host wait (16384 = ~60 us of K1, about one round trip); 0 = no chunks:
                                // the whole sweep runs when its masks are first needed
};

struct dy_ibf; struct dy_reads;
// dy_ibf_query_deferred: the sweep of the batch mapped next, launched chunk by chunk in front of the host's waits
struct PendingSweep { dy_ibf * f = nullptr; dy_reads * reads = nullptr; u32 threshold = 0, next = 0, slot = 0; };
struct SweepTiming { enum { MAX_CHUNKS = 24 }; hipEvent_t ev[2 * MAX_CHUNKS] = {}; u32 nChunks = 0; bool complete = false; };

// Stage ranges for a profiler (reference: the Stats timers of src/mapper.h:238-279 are the only observability it has).
// roctx is bound at run time and only when the tuning key `markers` asks for it: the library links no profiler.
struct Markers
{
    void * lib = nullptr;
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    bool tried = false;
    void load()
    {
        if (tried) return;
        tried = true;
        for (const char * name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"})
        {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) return;
        push = (int (*)(const char *))dlsym(lib, "roctxRangePushA");
        pop = (int (*)())dlsym(lib, "roctxRangePop");
        if (!push || !pop) { push = nullptr; pop = nullptr; }
    }
};
static Markers g_markers;
static void markBegin(bool enabled, const char * name) { if (!enabled) return; g_markers.load(); if (g_markers.push) g_markers.push(name); }
static void markEnd(bool enabled) { if (enabled && g_markers.pop) g_markers.pop(); }
struct StageRange
{
    bool on;
    StageRange(bool enabled, const char * name) : on(false)
    {
        if (!enabled) return;
        g_markers.load();
        if (g_markers.push) { g_markers.push(name); on = true; }
    }
    ~StageRange() { if (on) g_markers.pop(); }
};

struct dy_ctx
{
    int device = 0;
    PendingSweep pending;
    SweepTiming sweepTiming[2];
    hipEvent_t evFetch = nullptr;
    Tuning tune;
    hipStream_t stream = nullptr;
    hipEvent_t ev[8] = {};
    hipStream_t ibfStream = nullptr;   // K1 runs here: the sweep of the NEXT batch overlaps K2..K4 of the current one (dy_map_reads waits for the batch's own sweep)
    hipStream_t side[3] = {};          // K4's list kernels are latency-bound on a handful of waves each: they run side by side
    hipEvent_t evFork = nullptr, evJoin[3] = {};
    dy_map_stats stats = {};
    // workspaces (grow-only, reused across batches)
    DevBuf peHeavy, peItems, peU2, peLib, peLens, peIdx, peHasRescue, peJobs, peScratch, peMCounts, peMStart,
           waveCounts, waveOffsets, complexList, alignList, smallList, binTotals, binStart, binOrderRank, itemRead, itemBin, hits0, pool, counters, readMatchCount, readStart,
           cursor, gp, aux, marr, maux, dp, scanTmp,
           shLen, shLenOff, shOdd, shOddOff, shOwner, shTotals, hardList, slowList, dpList, itemKeys, itemPerm, orderHist, orderCursor, editSlab;
    size_t poolCap = 0;
    DevBuf devItems;                    // u32[4]: [0] the work list's size as the kernels see it, [1] as counted, [2] it did not fit (k_items_clamp)
    double extendLeftShare = 0.0;       // items the last two-launch K3 left to its second pass / items (extend_mode 3)
    u32 extendOnePassBatches = 0, extendModeUsed = 1;
    double itemsPerRead = 0.0;          // of the last batch mapped: sizes the next batch's work list when nothing waits for its count
    PinnedBuf hostUp;                   // staging of dy_reads_upload
    PinnedBuf hostCounters, hostItems;
    PinnedBuf hostSlots[8];             // dy_host_buffer: pinned host memory the caller reads files into / writes files from (FASTQ in, SAM out)
    DevBuf fqBlockCount, fqBlockOff, fqLineEnds, fqLensWide, fqLens16, fqStatus, samSizes, samOff, samMapq, samText, samFlags, samGroup;
    PinnedBuf hostStatus;               // small device -> host fetches (counters, flags): pinned, so that a fetch is one copy + one synchronisation
    PinnedBuf sparePinned[9];           // host arrays of fetched results, handed back by dy_result_free (hipHostMalloc costs milliseconds)
    DevBuf sortKeys;                    // k_finalize_t<BlockGroup>: 128-bit sort keys of reads with more than SORT_CAP matches
    DevBuf packCntSet, packCntOps, packOffSet, packOffOps, packReads, packSet, packOps;     // dy_result_fetch: dense copies made on the device
    DevBuf spare[12];                   // result buffers handed back by dy_result_free (hipMalloc / hipFree cost ~0.1 ms each and hipFree drains the device)
    // bins sharded over ranks: state between dy_shard_c1_unpack, dy_map_bins and dy_shard_c2_pack (packed exchanges, shard_kernels.h)
    std::vector<u64> shOwnerHost;       // the ownership table resident in shOwner
    DevBuf shPairs, shScan, shSegOff, shSend, shIds, shFlags, shFlatLens, shFlatMasks, shCounts, shSegOff2, shSend2, shLoads, shRawFlat;
    SegTable shArrival = {};            // where each source rank's reads start in the arrival order of the owner-side batch
    u32 shRanks = 0; const dy_reads * shBatch = nullptr;
    u32 lastRaw = 0; bool rawValid = false;
    u32 lastItems = 0, lastBins = 0; bool lastPaired = false, lastHasWorkList = false; DevBuf shOrig;      // paired dy_map_bins: the batch's arrival indices for dy_shard_c2_pack
    const u32 * devItemCount = nullptr; // device copy of the last work list's size
};

// result buffers are recycled through the context: take the smallest spare that is large enough, else allocate
static void takeBuffer(dy_ctx * ctx, DevBuf & d, size_t bytes)
{
    int best = -1;
    for (int i = 0; i < 12; ++i)
        if (ctx->spare[i].p && ctx->spare[i].cap >= bytes && (best < 0 || ctx->spare[i].cap < ctx->spare[best].cap)) best = i;
    if (best >= 0) ctx->spare[best].moveTo(d);
    else d.reserve(bytes);
}
static void giveBuffer(dy_ctx * ctx, DevBuf & d)
{
    if (!d.p) return;
    int slot = -1;
    for (int i = 0; i < 12; ++i) if (!ctx->spare[i].p) { slot = i; break; }
    if (slot < 0) { slot = 0; for (int i = 1; i < 12; ++i) if (ctx->spare[i].cap < ctx->spare[slot].cap) slot = i; if (ctx->spare[slot].cap >= d.cap) return; }
    d.moveTo(ctx->spare[slot]);
}

static void takePinned(dy_ctx * ctx, PinnedBuf & d, size_t bytes)
{
    int best = -1;
    for (int i = 0; i < 9; ++i)
        if (ctx->sparePinned[i].p && ctx->sparePinned[i].cap >= bytes && (best < 0 || ctx->sparePinned[i].cap < ctx->sparePinned[best].cap)) best = i;
    if (best >= 0) ctx->sparePinned[best].moveTo(d);
    else d.reserve(bytes);
}
static void givePinned(dy_ctx * ctx, PinnedBuf & d)
{
    if (!d.p) return;
    int slot = -1;
    for (int i = 0; i < 9; ++i) if (!ctx->sparePinned[i].p) { slot = i; break; }
    if (slot < 0) { slot = 0; for (int i = 1; i < 9; ++i) if (ctx->sparePinned[i].cap < ctx->sparePinned[slot].cap) slot = i; if (ctx->sparePinned[slot].cap >= d.cap) return; }
    d.moveTo(ctx->sparePinned[slot]);
}

struct dy_ibf
{
    dy_ctx * ctx = nullptr;
    IbfGeom g = {};
    DevBuf words;
    size_t nWords = 0;
    double fill = 0.0; bool fillKnown = false;     // fraction of set bits (sampled), refreshed after the filter changed
};

// Fraction of set bits of the table, from an evenly spread sample of 2^18 words (dy_ibf_add / clear invalidate it).
static double ibfFill(dy_ibf * f)
{
    if (f->fillKnown) return f->fill;
    dy_ctx * ctx = f->ctx;
    const u64 nTable = f->g.nHashPos * f->g.W;
    const u32 nSamples = (u32)std::min<u64>(nTable, 1u << 18);
    ctx->shTotals.reserve(64);
    HIP_CHECK(hipMemsetAsync(ctx->shTotals.p, 0, 8, ctx->stream));
    hipLaunchKernelGGL(k_ibf_fill_sample, dim3((nSamples + 255) / 256), dim3(256), 0, ctx->stream, f->words.as<u64>(), nTable, nSamples, ctx->shTotals.as<unsigned long long>());
    unsigned long long total = 0;
    HIP_CHECK(hipMemcpyAsync(&total, ctx->shTotals.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    f->fill = (double)total / (64.0 * nSamples);
    f->fillKnown = true;
    return f->fill;
}

// The sweeps first look at nK - threshold + slack k-mers (a bin without hits is out of reach after nK - threshold + 1): slack = the
// number of false hits that less than 1 in 20 reads sees any of the bins collect among that many k-mers (Poisson tail on the fill).
// Only the speed depends on it, never a result.
static u32 ibfSlack(dy_ibf * f, u32 nKmers, u32 threshold)
{
    if (f->ctx->tune.ibfSlack) return (u32)f->ctx->tune.ibfSlack;
    if (threshold == 0 || threshold > nKmers) return 1;
    const double lambda = ibfFill(f) * (double)(nKmers - threshold + 1);
    double term = std::exp(-lambda), cdf = 0.0;        // P(X = 0)
    for (u32 s = 1; s < threshold; ++s)
    {
        cdf += term;                                   // P(X < s)
        if ((double)f->g.nBins * (1.0 - cdf) < 0.05) return s;
        term *= lambda / (double)s;
    }
    return threshold;
}

// Bins per strand expected to pass the threshold on hash function 0 alone although the read is not theirs (Poisson tail on the fill):
// well below one -> k_ibf_query_lazy verifies just the read's own bin; dozens -> k_ibf_query_eager2 (two hash functions up front).
static double ibfExpectedFalseCandidates(dy_ibf * f, u32 nKmers, u32 threshold)
{
    if (threshold == 0 || threshold > nKmers) return 0.0;
    const double lambda = ibfFill(f) * (double)nKmers;
    double term = std::exp(-lambda), cdf = 0.0;
    for (u32 s = 0; s < threshold; ++s) { cdf += term; term *= lambda / (double)(s + 1); }
    return (double)f->g.nBins * std::max(0.0, 1.0 - cdf);
}

struct dy_reads
{
    dy_ctx * ctx = nullptr;
    u32 n = 0; u64 totalBases = 0; u32 maxLen = 0; u32 minLen = 0;
    DevBuf seqs, seqsRc, offs, masks;
    std::shared_ptr<std::vector<u32>> lens = std::make_shared<std::vector<u32>>();   // host copy of the read lengths (MAPQ needs them); shared with results
    u32 maskW = 0;       // words per read currently held in masks (0 = none)
    bool recycle = false; // buffers came from / go back to the context's spare pool (batches built per step by dy_reads_from_device)
    u32 maskBins = 0;
    hipEvent_t evK1[2] = {nullptr, nullptr};   // begin / end of the last dy_ibf_query on these reads (on the context's sweep stream)
    bool k1Pending = false;                    // the masks are being written by a sweep that nobody has waited for yet
    DevBuf orig;          // dy_shard_c1_unpack (paired): arrival index of every read (the batch is re-ordered to [first mates | second mates])
    bool pairLayout = false;
    DevBuf fqText, fqRecs; u64 fqBytes = 0; bool fromFastq = false;     // dy_reads_from_fastq: the FASTQ window and its record table stay resident for dy_result_sam
};

struct dy_index
{
    dy_ctx * ctx = nullptr;
    std::vector<BinDesc> bins;
    std::vector<std::string> contigNames;
    std::vector<u64> contigLengths;
    DevBuf dBins, rank, ind, valOffset, valContig, contigs, limits, drp, saFullOffset, saFullContig, prefixTable, prefixWide, isa, ftext;
    bool hasIsa = false, hasWide = false;
    bool hasFullSa = false, multiContig = false, sharded = false;
    mutable DevBuf dNames, dNameOffs; mutable bool namesOnDevice = false;      // contig names for dy_result_sam (uploaded at the first use)
    std::vector<u64> ownedMask;     // bins whose FM structures are resident (all of them unless loaded sharded)
    DevBuf dOwnedMask;
    u32 prefixQ = 0;
    IndexView view() const
    {
        IndexView v;
        v.bins = dBins.as<BinDesc>(); v.rank = rank.as<uint4>(); v.ind = ind.as<uint2>(); v.valOffset = valOffset.as<u32>();
        v.valContig = valContig.as<u16>(); v.contigs = contigs.as<u8>(); v.limits = limits.as<u64>(); v.drp = drp.as<u32>();
        v.saFullOffset = hasFullSa ? saFullOffset.as<u32>() : nullptr;
        v.saFullContig = (hasFullSa && multiContig) ? saFullContig.as<u16>() : nullptr;
        v.prefixTable = (prefixQ && !hasWide) ? prefixTable.as<uint2>() : nullptr; v.prefixQ = prefixQ;
        v.prefixWide = (prefixQ && hasWide) ? prefixWide.as<uint4>() : nullptr;
        v.isa = hasIsa ? isa.as<u32>() : nullptr; v.ftext = hasIsa ? ftext.as<u8>() : nullptr;
        v.nBins = (u32)bins.size();
        return v;
    }
};

struct dy_result
{
    dy_ctx * ctx = nullptr;
    u32 nReads = 0; u32 nPool = 0; u32 cigarStride = 0; size_t setCap = 0;
    std::shared_ptr<const std::vector<u32>> readLens;
    DevBuf out, outSet, cigars;
    bool fetched = false;
    PinnedBuf reads, matches, cigarOps;      // dy_read_result[nReads], dy_match[nMatches], u32[nOps] after dy_result_fetch
    size_t nMatches = 0, nOps = 0;
};

// dy_result_fetch, device side: the per-read records with dense matchesSet / CIGAR offsets (the finalize kernels write both at
// capacity-sized strides), so that only what the host arrays hold crosses PCIe
__global__ void k_result_counts(const ReadOut * __restrict__ out, u32 n, u32 * __restrict__ cntSet, u32 * __restrict__ cntOps)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    cntSet[i] = i < n ? out[i].bestCount + out[i].subCount : 0;
    cntOps[i] = i < n && out[i].mapped ? out[i].cigarLen : 0;
}
__global__ void k_result_pack(const ReadOut * __restrict__ out, const OutMatch * __restrict__ outSet, const u32 * __restrict__ cigars, u32 n,
                              const u32 * __restrict__ offSet, const u32 * __restrict__ offOps,
                              ReadOut * __restrict__ packReads, OutMatch * __restrict__ packSet, u32 * __restrict__ packOps)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ReadOut o = out[i];
    const u32 nset = o.bestCount + o.subCount, nops = o.mapped ? o.cigarLen : 0;
    const u32 s = offSet[i], c = offOps[i];
    for (u32 k = 0; k < nset; ++k) packSet[s + k] = outSet[(size_t)o.setBegin + k];
    for (u32 k = 0; k < nops; ++k) packOps[c + k] = cigars[(size_t)o.cigarBegin + k];
    o.setBegin = s; o.cigarBegin = c; o.cigarLen = nops;
    packReads[i] = o;
}
static_assert(sizeof(ReadOut) == sizeof(dy_read_result) && sizeof(OutMatch) == sizeof(dy_match), "device mirrors of the ABI records");
static_assert(offsetof(ReadOut, mapq) == offsetof(dy_read_result, mapq) && offsetof(ReadOut, setBegin) == offsetof(dy_read_result, set_begin) &&
              offsetof(ReadOut, cigarBegin) == offsetof(dy_read_result, cigar_begin) && offsetof(ReadOut, pairProb) == offsetof(dy_read_result, pair_prob) &&
              offsetof(ReadOut, errors) == offsetof(dy_read_result, primary.errors), "device mirrors of the ABI records");

template <typename T> static void upload(DevBuf & d, const std::vector<T> & h)
{
    d.reserve(h.size() * sizeof(T) + 64);
    if (!h.empty()) HIP_CHECK(hipMemcpy(d.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
}

__global__ void k_fill_masks(u64 * masks, u32 nReads, u32 W, u32 nBins)
{
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (u64)nReads * W) return;
    const u32 w = (u32)(t % W);
    const u32 left = nBins - w * 64;
    masks[t] = left >= 64 ? ~0ull : ((1ull << left) - 1);
}

template <int NW>
static void launchExtend(dy_ctx * ctx, const IndexView & ix, const ReadsView & rv, u32 nItems, u32 S0, const MapParams & mp, const MatchSink & sink, u32 maxSeedLen)
{
    const dim3 grid(((nItems + 255) / 256 + 7) / 8 * 8);
    u32 * counters = ctx->counters.as<u32>();
    u32 mode = mp.sensitivity == 2 ? 2u : (u32)ctx->tune.extendMode;          // -y full: edit-distance seeds, no fast pass (tuning extend_mode)
    if (mode == 3)
    {   // The fast pass pays when few items need the second one (10 % at the default error profile: even; 19 % at 150 bp / 5 %: 1.38 -> 1.13 ms)
        // and costs when most do (reads with 1-3 % substitutions: 1.08 against 0.85 ms per 1 M reads).  The share the last fast pass left over
        // decides for the batches that follow; after 15 one-pass batches a two-launch batch measures it again.
        mode = 1;
        if (ctx->extendLeftShare > 0.30 && ctx->extendOnePassBatches > 0) mode = 2;
    }
    ctx->extendModeUsed = mode;
    if (mode == 0)          // fast pass, then each block redoes what it left (counters[0]: how many items that was)
        hipLaunchKernelGGL((k_extend<NW, 2>), grid, dim3(256), 0, ctx->stream, ix, rv, ctx->itemRead.as<u32>(), ctx->itemBin.as<u16>(),
                           ctx->devItemCount, S0, mp, ctx->hits0.as<uint2>(), sink, counters + 2, ctx->hardList.as<u32>(), ctx->itemPerm.as<u32>(), counters, (u32 *)nullptr, 1u);
    else if (mode == 1)
    {   // fast pass, then the listed rest in a launch of its own (counters[0] = its size; the second grid covers the worst case, the
        // blocks the list does not need return at once)
        hipLaunchKernelGGL((k_extend<NW, 1>), grid, dim3(256), 0, ctx->stream, ix, rv, ctx->itemRead.as<u32>(), ctx->itemBin.as<u16>(),
                           ctx->devItemCount, S0, mp, ctx->hits0.as<uint2>(), sink, counters + 2, ctx->hardList.as<u32>(), ctx->itemPerm.as<u32>(), counters, ctx->slowList.as<u32>(), 1u);
        hipLaunchKernelGGL((k_extend<NW, 0>), grid, dim3(256), 0, ctx->stream, ix, rv, ctx->itemRead.as<u32>(), ctx->itemBin.as<u16>(),
                           counters, S0, mp, ctx->hits0.as<uint2>(), sink, counters + 2, ctx->hardList.as<u32>(), ctx->slowList.as<u32>(), (u32 *)nullptr, (u32 *)nullptr, 0u);
    }
    else
        hipLaunchKernelGGL((k_extend<NW, 0>), grid, dim3(256), 0, ctx->stream, ix, rv, ctx->itemRead.as<u32>(), ctx->itemBin.as<u16>(),
                           ctx->devItemCount, S0, mp, ctx->hits0.as<uint2>(), sink, counters + 2, ctx->hardList.as<u32>(), ctx->itemPerm.as<u32>(), (u32 *)nullptr, (u32 *)nullptr, 1u);
    // re-seeded items: list-driven, spread thinly over a full grid (normally a tiny fraction of the items)
    // -y full: the edit-distance trie walk keeps one frame per level in a slab, one slice per wavefront of this grid
    const u32 hardBlocks = 2048;
    EditFrame * slab = nullptr; u32 stride = 0;
    if (mp.sensitivity == 2)
    {
        stride = maxSeedLen + 8;
        ctx->editSlab.reserve((size_t)hardBlocks * 4 * stride * sizeof(EditFrame) + 64);
        slab = ctx->editSlab.as<EditFrame>();
    }
    hipLaunchKernelGGL(k_extend_hard<NW>, dim3(hardBlocks), dim3(256), 0, ctx->stream, ix, rv, ctx->itemRead.as<u32>(),
                       ctx->itemBin.as<u16>(), S0, mp, ctx->hits0.as<uint2>(), sink, ctx->counters.as<u32>() + 2, ctx->hardList.as<u32>(), slab, stride);
}

// The bin masks of `reads` may still be in the making on the sweep stream: the main stream waits for them (device side, no host stall).
static void sweepFlush(dy_ctx * ctx);
static void waitForMasks(dy_ctx * ctx, dy_reads * reads)
{
    if (ctx->pending.reads == reads) sweepFlush(ctx);       // a deferred sweep of these reads: whatever is left of it runs now, on this stream
    if (reads->k1Pending) HIP_CHECK(hipStreamWaitEvent(ctx->stream, reads->evK1[1], 0));
}
// ... and once the main stream has been synchronised behind that wait, the sweep's duration can be read
static void collectSweepTime(dy_ctx * ctx, dy_reads * reads)
{
    if (!reads->k1Pending) return;
    if (hipEventQuery(reads->evK1[1]) != hipSuccess) return;      // a sweep queued ahead for the NEXT use of these reads (nobody has waited for it yet): leave it alone
    float ms = 0;
    HIP_CHECK(hipEventSynchronize(reads->evK1[1]));
    HIP_CHECK(hipEventElapsedTime(&ms, reads->evK1[0], reads->evK1[1]));
    ctx->stats.ms_ibf = ms; reads->k1Pending = false;
}

static void fetchSync(dy_ctx * ctx);
static void collectDeferredSweepTime(dy_ctx * ctx);
template <typename TIn, typename TOut> static void exclusiveSum(dy_ctx * ctx, const TIn * in, TOut * out, size_t n);

// Work list (bin-major): (read,bin) items of the masks on `reads` -> ctx->itemRead / itemBin / binStart (+ the local sortedBins() rank).
// asyncCap = 0: the host waits for the list's size and sizes everything exactly (returns the size).  asyncCap > 0 (sorted list only): nothing
// waits — the arrays hold asyncCap entries, the kernels take the size from ctx->devItems[0] (k_items_clamp; [1] = what was counted, devFlags[0] = it did
// not fit) and the caller checks the flag when the whole step has run (returns asyncCap: the bound every later launch is sized by).
static u32 buildWorkList(dy_ctx * ctx, dy_reads * reads, u32 nBins, u32 asyncCap = 0)
{
        hipStream_t st = ctx->stream;
        const u32 nReads = reads->n, W = (nBins + 63) / 64;
        waitForMasks(ctx, reads);
        ctx->binStart.reserve((size_t)(nBins + 1) * 4); ctx->binOrderRank.reserve((size_t)nBins * 4);
        ctx->devItems.reserve(64);
        u32 nItems = 0;
        if (ctx->tune.itemsSorted && asyncCap)
        {
            const u64 nWords = (u64)nReads * W;
            const u32 cap = asyncCap;
            ctx->waveCounts.reserve((size_t)(nReads + 1) * 4); ctx->waveOffsets.reserve((size_t)(nReads + 1) * 4);
            HIP_CHECK(hipMemsetAsync(ctx->waveCounts.p, 0, (size_t)(nReads + 1) * 4, st));
            if (nWords) hipLaunchKernelGGL(k_items_word_count, dim3((u32)((nWords + 255) / 256)), dim3(256), 0, st, reads->masks.as<u64>(), nWords, W, nBins, ctx->waveCounts.as<u32>());
            exclusiveSum(ctx, ctx->waveCounts.as<u32>(), ctx->waveOffsets.as<u32>(), (size_t)nReads + 1);
            HIP_CHECK(hipMemsetAsync(ctx->devItems.p, 0, 16, st));
            hipLaunchKernelGGL(k_items_clamp, dim3(1), dim3(1), 0, st, ctx->waveOffsets.as<u32>() + nReads, cap, ctx->devItems.as<u32>(), ctx->devItems.as<u32>() + 2);
            ctx->devItemCount = ctx->devItems.as<u32>();
            ctx->itemRead.reserve((size_t)cap * 4 + 64); ctx->itemBin.reserve((size_t)cap * 2 + 64);
            ctx->itemKeys.reserve((size_t)cap * 4 + 64); ctx->itemPerm.reserve((size_t)cap * 4 + 64);
            // slots behind the list sort to its end: key = nBins
            HIP_CHECK(hipMemsetD16Async((hipDeviceptr_t)ctx->itemKeys.p, (unsigned short)nBins, cap, st));
            if (nWords) hipLaunchKernelGGL(k_items_emit, dim3((u32)((nWords + 255) / 256)), dim3(256), 0, st, reads->masks.as<u64>(), nWords, W, nBins, ctx->waveOffsets.as<u32>(),
                                           ctx->itemKeys.as<u16>(), ctx->itemPerm.as<u32>(), cap);
            int bits = 1; while ((1u << bits) < nBins + 1) ++bits;
            size_t sortBytes = 0;
            HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, sortBytes, ctx->itemKeys.as<u16>(), ctx->itemBin.as<u16>(), ctx->itemPerm.as<u32>(), ctx->itemRead.as<u32>(), (int)cap, 0, bits, st));
            ctx->scanTmp.reserve(sortBytes + 64);
            HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(ctx->scanTmp.p, sortBytes, ctx->itemKeys.as<u16>(), ctx->itemBin.as<u16>(), ctx->itemPerm.as<u32>(), ctx->itemRead.as<u32>(), (int)cap, 0, bits, st));
            hipLaunchKernelGGL(k_items_bin_bounds, dim3((nBins + 1 + 255) / 256), dim3(256), 0, st, ctx->itemBin.as<u16>(), ctx->devItems.as<u32>(), nBins, ctx->binStart.as<u32>());
            hipLaunchKernelGGL(k_items_bin_rank, dim3((nBins + 255) / 256), dim3(256), 0, st, ctx->binStart.as<u32>(), nBins, ctx->binOrderRank.as<u32>());
            return cap;
        }
        if (ctx->tune.itemsSorted)
        {   // (bin, read) pairs in read order, stable radix sort by bin (map_kernels.h)
            const u64 nWords = (u64)nReads * W;
            ctx->waveCounts.reserve((size_t)(nReads + 1) * 4); ctx->waveOffsets.reserve((size_t)(nReads + 1) * 4);
            HIP_CHECK(hipMemsetAsync(ctx->waveCounts.p, 0, (size_t)(nReads + 1) * 4, st));
            if (nWords) hipLaunchKernelGGL(k_items_word_count, dim3((u32)((nWords + 255) / 256)), dim3(256), 0, st, reads->masks.as<u64>(), nWords, W, nBins, ctx->waveCounts.as<u32>());
            size_t tmpBytes = 0;
            HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmpBytes, ctx->waveCounts.as<u32>(), ctx->waveOffsets.as<u32>(), (int)(nReads + 1), st));
            ctx->scanTmp.reserve(tmpBytes + 64);
            HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(ctx->scanTmp.p, tmpBytes, ctx->waveCounts.as<u32>(), ctx->waveOffsets.as<u32>(), (int)(nReads + 1), st));
            ctx->devItemCount = ctx->waveOffsets.as<u32>() + nReads;
            ctx->hostItems.reserve(64);
            HIP_CHECK(hipMemcpyAsync(ctx->hostItems.p, ctx->waveOffsets.as<u32>() + nReads, 4, hipMemcpyDeviceToHost, st));
            if ((u64)nReads * nBins > 0x7FFFFFFFull)
            {   // the list is indexed with 31 bits and the 32-bit scan could have wrapped (every read selecting hundreds of bins: an unfiltered
                // run over many bins): count again in 64 bits
                ctx->binTotals.reserve(16);
                size_t rb = 0;
                HIP_CHECK(hipcub::DeviceReduce::Sum(nullptr, rb, ctx->waveCounts.as<u32>(), ctx->binTotals.as<unsigned long long>(), (int)nReads, st));
                ctx->scanTmp.reserve(rb + 64);
                HIP_CHECK(hipcub::DeviceReduce::Sum(ctx->scanTmp.p, rb, ctx->waveCounts.as<u32>(), ctx->binTotals.as<unsigned long long>(), (int)nReads, st));
                unsigned long long exact = 0;
                HIP_CHECK(hipMemcpyAsync(&exact, ctx->binTotals.p, 8, hipMemcpyDeviceToHost, st));
                HIP_CHECK(hipStreamSynchronize(st));
                if (exact > 0x7FFFFFFFull) throw DyError(DY_EUNSUPPORTED, "more than 2^31 (read, bin) pairs in one batch: map fewer reads per batch");
            }
            fetchSync(ctx);
            nItems = *ctx->hostItems.as<u32>();
            collectSweepTime(ctx, reads);
            ctx->itemRead.reserve((size_t)nItems * 4 + 64); ctx->itemBin.reserve((size_t)nItems * 2 + 64);
            ctx->itemKeys.reserve((size_t)nItems * 4 + 64); ctx->itemPerm.reserve((size_t)nItems * 4 + 64);      // unsorted keys / values (free until the K3 work order)
            if (nItems)
            {
                hipLaunchKernelGGL(k_items_emit, dim3((u32)((nWords + 255) / 256)), dim3(256), 0, st, reads->masks.as<u64>(), nWords, W, nBins, ctx->waveOffsets.as<u32>(),
                                   ctx->itemKeys.as<u16>(), ctx->itemPerm.as<u32>(), nItems);
                int bits = 1; while ((1u << bits) < nBins) ++bits;
                size_t sortBytes = 0;
                HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, sortBytes, ctx->itemKeys.as<u16>(), ctx->itemBin.as<u16>(), ctx->itemPerm.as<u32>(), ctx->itemRead.as<u32>(),
                                                             (int)nItems, 0, bits, st));
                ctx->scanTmp.reserve(sortBytes + 64);
                HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(ctx->scanTmp.p, sortBytes, ctx->itemKeys.as<u16>(), ctx->itemBin.as<u16>(), ctx->itemPerm.as<u32>(), ctx->itemRead.as<u32>(),
                                                             (int)nItems, 0, bits, st));
            }
            HIP_CHECK(hipMemsetAsync(ctx->devItems.p, 0, 16, st));
            hipLaunchKernelGGL(k_items_clamp, dim3(1), dim3(1), 0, st, ctx->waveOffsets.as<u32>() + nReads, nItems, ctx->devItems.as<u32>(), ctx->devItems.as<u32>() + 2);
            ctx->devItemCount = ctx->devItems.as<u32>();
            hipLaunchKernelGGL(k_items_bin_bounds, dim3((nBins + 1 + 255) / 256), dim3(256), 0, st, ctx->itemBin.as<u16>(), ctx->devItems.as<u32>(), nBins, ctx->binStart.as<u32>());
            hipLaunchKernelGGL(k_items_bin_rank, dim3((nBins + 255) / 256), dim3(256), 0, st, ctx->binStart.as<u32>(), nBins, ctx->binOrderRank.as<u32>());
            return nItems;
        }
        const u32 nWaves = (nReads + 63) / 64;
        const size_t nCells = (size_t)nBins * nWaves;
        ctx->waveCounts.reserve((nCells + 1) * 4); ctx->waveOffsets.reserve((nCells + 1) * 4);
        HIP_CHECK(hipMemsetAsync(ctx->waveCounts.as<u32>() + nCells, 0, 4, st));
        hipLaunchKernelGGL(k_items_count, dim3((nWaves * 64 + 255) / 256), dim3(256), 0, st, reads->masks.as<u64>(), nReads, W, nBins, ctx->waveCounts.as<u32>());
        {
            size_t tmpBytes = 0;
            HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmpBytes, ctx->waveCounts.as<u32>(), ctx->waveOffsets.as<u32>(), (int)(nCells + 1), st));
            ctx->scanTmp.reserve(tmpBytes + 64);
            HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(ctx->scanTmp.p, tmpBytes, ctx->waveCounts.as<u32>(), ctx->waveOffsets.as<u32>(), (int)(nCells + 1), st));
        }
        hipLaunchKernelGGL(k_items_bin_starts, dim3((nBins + 1 + 255) / 256), dim3(256), 0, st, ctx->waveOffsets.as<u32>(), ctx->waveOffsets.as<u32>() + nCells,
                           nWaves, nBins, ctx->binStart.as<u32>());
        hipLaunchKernelGGL(k_items_bin_rank, dim3((nBins + 255) / 256), dim3(256), 0, st, ctx->binStart.as<u32>(), nBins, ctx->binOrderRank.as<u32>());
        ctx->devItemCount = ctx->waveOffsets.as<u32>() + nCells;
        ctx->hostItems.reserve(64);
        HIP_CHECK(hipMemcpyAsync(ctx->hostItems.p, ctx->waveOffsets.as<u32>() + nCells, 4, hipMemcpyDeviceToHost, st));
        fetchSync(ctx);
        nItems = *ctx->hostItems.as<u32>();
        collectSweepTime(ctx, reads);
        ctx->itemRead.reserve((size_t)nItems * 4 + 64); ctx->itemBin.reserve((size_t)nItems * 2 + 64);
        if (nItems)
            hipLaunchKernelGGL(k_items_scatter, dim3((nWaves * 64 + 255) / 256), dim3(256), 0, st, reads->masks.as<u64>(), nReads, W, nBins,
                               ctx->waveOffsets.as<u32>(), ctx->itemRead.as<u32>(), ctx->itemBin.as<u16>());
        return nItems;
}

// Work list + K2 + K3: (read,bin) items of the masks on `reads` -> raw matches in ctx->pool, per-read counts in ctx->readMatchCount.
// async: nothing waits — the work list is sized from the previous batch's items per read, the pool keeps its capacity, nItemsOut / nPoolOut are
// those BOUNDS (every later launch and allocation is sized by them, the kernels read the real sizes on the device) and the caller checks
// devItems[2] / the pool's overflow flag once the whole step has run, mapping the batch again the waiting way if either is raised.
static void runSearchExtend(dy_ctx * ctx, const dy_index * idx, dy_reads * reads, const dy_params * p, u32 S0, u32 & nItemsOut, u32 & nPoolOut, bool async = false)
{
        hipStream_t st = ctx->stream;
        const u32 nReads = reads->n, nBins = (u32)idx->bins.size();
        HIP_CHECK(hipEventRecord(ctx->ev[2], st));
        u32 asyncCap = 0;
        if (async) asyncCap = (u32)std::min<double>(2.0e9, (double)nReads * ctx->itemsPerRead * 1.25 + 4096.0);
        u32 nItemsBuilt = 0;
        { StageRange range(ctx->tune.markers != 0, "dy:items"); nItemsBuilt = buildWorkList(ctx, reads, nBins, asyncCap); }
        const u32 nItems = nItemsBuilt;
        const u32 * nItemsDev = ctx->devItemCount;
        ctx->stats.items = async ? 0 : nItems;
        HIP_CHECK(hipEventRecord(ctx->ev[3], st));
        const bool marks = ctx->tune.markers != 0;
        markBegin(marks, "dy:search");

        // ---- K2: exact search of the round-0 seeds
        const IndexView ix = idx->view();
        ReadsView rv; rv.seqs = reads->seqs.as<u8>(); rv.seqsRc = reads->seqsRc.as<u8>(); rv.offs = reads->offs.as<u64>(); rv.nReads = nReads;
        MapParams mp; mp.errorRate = p->error_rate; mp.strataRate = p->strata_rate; mp.sensitivity = p->sensitivity; mp.hitsThreshold = p->hits_threshold;
        ctx->hits0.reserve((size_t)nItems * 2 * S0 * sizeof(uint2) + 64);
        if (nItems)
        {
            u64 threads = (u64)nItems * 2 * S0;
            hipLaunchKernelGGL(k_seed_search, dim3((u32)(((threads + 255) / 256 + 7) / 8 * 8)), dim3(256), 0, st, ix, rv, ctx->itemRead.as<u32>(), ctx->itemBin.as<u16>(),
                               nItemsDev, S0, mp.errorRate, ctx->hits0.as<uint2>());
        }
        // ---- K3 work order: (bin, seeds found)
        ctx->itemKeys.reserve((size_t)nItems * 4 + 64); ctx->itemPerm.reserve((size_t)nItems * 4 + 64);
        {
            const size_t nKeys = (size_t)nBins * 8 + 1;
            ctx->orderHist.reserve(nKeys * 4 + 64); ctx->orderCursor.reserve(nKeys * 4 + 64);
            if (nItems)
            {
                HIP_CHECK(hipMemsetAsync(ctx->orderHist.p, 0, nKeys * 4, st));
                hipLaunchKernelGGL(k_item_order_count, dim3((nItems + 1023) / 1024), dim3(1024), 0, st, ctx->hits0.as<uint2>(), ctx->itemBin.as<u16>(), nItemsDev, S0,
                                   ctx->itemKeys.as<u32>(), ctx->orderHist.as<u32>());
                size_t tb = 0;
                HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, ctx->orderHist.as<u32>(), ctx->orderCursor.as<u32>(), (int)nKeys, st));
                ctx->scanTmp.reserve(tb + 64);
                HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(ctx->scanTmp.p, tb, ctx->orderHist.as<u32>(), ctx->orderCursor.as<u32>(), (int)nKeys, st));
                hipLaunchKernelGGL(k_item_order_scatter, dim3((nItems + 1023) / 1024), dim3(1024), 0, st, ctx->itemKeys.as<u32>(), nItemsDev, ctx->orderCursor.as<u32>(),
                                   ctx->itemPerm.as<u32>());
            }
        }
        HIP_CHECK(hipEventRecord(ctx->ev[4], st));
        markEnd(marks);
        StageRange extendRange(marks, "dy:extend");

        // ---- K3: classify, rank, locate, extend -> raw match pool (grown and re-run on overflow)
        ctx->counters.reserve(COUNTER_BYTES);
        ctx->readMatchCount.reserve((size_t)(nReads + 1) * 4);
        ctx->hardList.reserve((size_t)nItems * 4 + 64);
        ctx->slowList.reserve((size_t)nItems * 4 + 64);
        if (ctx->tune.poolCap) { if (!ctx->poolCap) ctx->poolCap = (size_t)ctx->tune.poolCap; }
        else if (ctx->poolCap < (size_t)nItems + 1024) ctx->poolCap = (size_t)nItems * 2 + 1024;
        u32 hostCounters[COUNTER_BYTES / 4] = {0};   // layout: see COUNTER_* below
        const int nw = (int)((reads->maxLen + 63) / 64);
        u64 poolTotal = 0;
        for (;;)
        {
            ctx->pool.reserve(ctx->poolCap * sizeof(RawMatch));
            HIP_CHECK(hipMemsetAsync(ctx->counters.p, 0, COUNTER_BYTES, st));
            HIP_CHECK(hipMemsetAsync(ctx->readMatchCount.p, 0, (size_t)(nReads + 1) * 4, st));
            MatchSink sink; sink.pool = ctx->pool.as<RawMatch>(); sink.poolCount = (unsigned long long *)(ctx->counters.as<u32>() + COUNTER_POOL64); sink.poolCap = (u32)std::min<size_t>(ctx->poolCap, 0xFFFFFFF0u);
            sink.overflow = ctx->counters.as<u32>() + 1; sink.readMatchCount = ctx->readMatchCount.as<u32>();
            if (nItems)
            {
                // the longest seed a read of the batch can be cut into: a short read with few errors may be ONE seed (ceil((e + 1) / 3) = 1)
                const u32 maxSeedLen = reads->maxLen;
                // the kernels for reads of up to 256 bases hold the extension's edit-distance band in one word (myers_device.h): at most 31 errors
                const bool bandFits = readErrorsOf(reads->maxLen, p->error_rate) <= PREFIX_SEARCH_SMALL_MAX_K;      // the contract of prefixSearch<NW <= 4>
                if (nw <= 2 && bandFits) launchExtend<2>(ctx, ix, rv, nItems, S0, mp, sink, maxSeedLen);
                else if (nw <= 4 && bandFits) launchExtend<4>(ctx, ix, rv, nItems, S0, mp, sink, maxSeedLen);
                else if (nw <= 16) launchExtend<16>(ctx, ix, rv, nItems, S0, mp, sink, maxSeedLen);
                else launchExtend<128>(ctx, ix, rv, nItems, S0, mp, sink, maxSeedLen);
                HIP_CHECK(hipGetLastError());
            }
            if (async) break;                    // the counters are looked at when the step has run
            ctx->hostCounters.reserve(COUNTER_BYTES);
            HIP_CHECK(hipMemcpyAsync(ctx->hostCounters.p, ctx->counters.p, COUNTER_BYTES, hipMemcpyDeviceToHost, st));
            fetchSync(ctx);
            memcpy(hostCounters, ctx->hostCounters.p, COUNTER_BYTES);
            memcpy(&poolTotal, hostCounters + COUNTER_POOL64, 8);
            if (!hostCounters[1]) break;
            // every append was counted, stored or not: the count of the failed pass IS the capacity the batch needs (the extension is
            // deterministic); + 1/4 only spares a second retry when the caller maps similar batches with the same context
            if (poolTotal >= 0xFFFFFFF0ull) throw DyError(DY_EUNSUPPORTED, "more than 2^32 raw matches in one batch: use a smaller reads batch");
            ctx->poolCap = (size_t)poolTotal + (size_t)poolTotal / 4 + 1024;
            ctx->stats.pool_retries++;
        }
        HIP_CHECK(hipEventRecord(ctx->ev[5], st));
        const u32 nPool = async ? (u32)std::min<size_t>(ctx->poolCap, 0xFFFFFFF0u) : (u32)poolTotal;
        if (!async)
        {
            ctx->stats.raw_matches = nPool; ctx->stats.hard_items = hostCounters[2] - hostCounters[12]; ctx->stats.second_pass_items = hostCounters[0];     // re-seeded items, as the reference counts them
            if (ctx->extendModeUsed == 1 && nItems)
            {
                ctx->extendLeftShare = (double)hostCounters[0] / (double)nItems;
                ctx->extendOnePassBatches = ctx->extendLeftShare > 0.30 ? 15u : 0u;
            }
            else if (ctx->extendModeUsed == 2 && ctx->extendOnePassBatches) --ctx->extendOnePassBatches;
            if (nReads) ctx->itemsPerRead = std::max(0.25, (double)nItems / (double)nReads);
        }
        nItemsOut = nItems; nPoolOut = nPool;
}

// Group raw matches by read (counts in ctx->readMatchCount) and run the per-read aggregation (K4 / paired-end stages).
// Paired-end: the bin-local stages (child, library, pairing, rescue — the reference's child mapper, src/mapper.h:826-1172) run where the
// bin's FM index lives (PE_OWNER) and the per-read merge over the bins (copyMatches / copyCigars / main mapper, src/d_mapper.h:127-245)
// where the read lives (PE_HOME); one GPU does both.  PE_OWNER alone: `res` may be NULL.  PE_HOME alone: the caller has filled
// ctx->peItems / peHasRescue from the owners' records and the work list of the home reads is in place.
enum { PE_OWNER = 1, PE_HOME = 2 };
static void runFinalize(dy_ctx * ctx, const dy_index * idx, dy_reads * reads, const dy_params * p, dy_result * res, const RawMatch * pool, u32 nPool,
                        u32 nItems, u32 maxMatchErr, u32 peStages = PE_OWNER | PE_HOME, const u32 * nPoolDev = nullptr /* the real count, where nPool is only its bound */)
{
        hipStream_t st = ctx->stream;
        StageRange finalizeRange(ctx->tune.markers != 0, "dy:finalize");
        const u32 nReads = reads->n, nBins = (u32)idx->bins.size(), W = (nBins + 63) / 64;
        const bool paired = p->paired != 0;
        const IndexView ix = idx->view();
        ReadsView rv; rv.seqs = reads->seqs.as<u8>(); rv.seqsRc = reads->seqsRc.as<u8>(); rv.offs = reads->offs.as<u64>(); rv.nReads = nReads;
        // ---- group by read, then per-read aggregation
        ctx->readStart.reserve((size_t)(nReads + 1) * 4);
        size_t tmpBytes = 0;
        HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmpBytes, ctx->readMatchCount.as<u32>(), ctx->readStart.as<u32>(), (int)(nReads + 1), st));
        ctx->scanTmp.reserve(tmpBytes + 64);
        HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(ctx->scanTmp.p, tmpBytes, ctx->readMatchCount.as<u32>(), ctx->readStart.as<u32>(), (int)(nReads + 1), st));
        ctx->cursor.reserve((size_t)nReads * 4 + 64);
        HIP_CHECK(hipMemsetAsync(ctx->cursor.p, 0, (size_t)nReads * 4, st));
        ctx->gp.reserve((size_t)nPool * sizeof(RawMatch) + 64);
        if (nPool)
            hipLaunchKernelGGL(k_group_matches, dim3((nPool + 255) / 256), dim3(256), 0, st, pool, nPool, nPoolDev, nReads, ctx->readStart.as<u32>(),
                               ctx->cursor.as<u32>(), ctx->gp.as<RawMatch>(), ctx->counters.as<u32>() + COUNTER_BAD_GROUP);
        ctx->aux.reserve((size_t)nPool * 16 + 64);
        const bool home = !paired || (peStages & PE_HOME), owner = !paired || (peStages & PE_OWNER);
        const size_t setCap = paired ? (size_t)nPool * 3 + nItems : (size_t)nPool * 3;
        if (home)
        {
            res->setCap = setCap;
            ctx->marr.reserve(setCap * sizeof(RawMatch) + 64);
            ctx->maux.reserve(setCap * 2 * 4 + 64);
            takeBuffer(ctx, res->outSet, setCap * sizeof(OutMatch) + 64);
        }
        const u32 finBlocks = 256 * 8, finWaves = finBlocks * 4;
        // reads with more matches than this are aggregated by a whole 1024-thread block instead of one wavefront (repeats, low-complexity runs)
        const u32 blockMin = (u32)ctx->tune.finalizeBlockMin;
        const u32 hugeBlocks = 512;
        const size_t dpStride = (dpScratchBytes(reads->maxLen, maxMatchErr) + 15) & ~(size_t)15;
        ctx->dp.reserve(dpStride * (finWaves + hugeBlocks) + 64);
        FinalizeArgs fa;
        fa.gp = ctx->gp.as<RawMatch>(); fa.readStart = ctx->readStart.as<u32>(); fa.readOffs = rv.offs; fa.readSeqs = rv.seqs; fa.readSeqsRc = rv.seqsRc; fa.nReads = nReads;
        fa.bins = ix.bins; fa.contigs = ix.contigs; fa.limits = ix.limits; fa.binOrderRank = ctx->binOrderRank.as<u32>(); fa.strataRate = p->strata_rate; fa.errorRate = p->error_rate;
        fa.aux = ctx->aux.as<u32>(); fa.marr = ctx->marr.as<RawMatch>(); fa.maux = ctx->maux.as<u32>(); fa.dpScratch = ctx->dp.as<u8>(); fa.dpStride = dpStride;
        fa.out = home ? res->out.as<ReadOut>() : nullptr; fa.outSet = home ? res->outSet.as<OutMatch>() : nullptr;
        fa.cigars = home ? res->cigars.as<u32>() : nullptr; fa.cigarStride = home ? res->cigarStride : 0;
        fa.sortKeys = nullptr;
        if (!paired && !ctx->tune.finalizeNoKeys) { ctx->sortKeys.reserve((size_t)nPool * 48 + 64); fa.sortKeys = ctx->sortKeys.as<u64>(); }
        ctx->complexList.reserve((size_t)nReads * 4 + 64);
        u32 * complexCount = ctx->counters.as<u32>() + 3;          // zeroed with the other counters before K3
        fa.complexList = ctx->complexList.as<u32>(); fa.complexCount = complexCount;
        if (!paired)
        {
            ctx->alignList.reserve((size_t)nReads * 4 + 64);
            u32 * alignCount = ctx->counters.as<u32>() + 4;
            // Matches with up to 3 errors (bands of up to 7 diagonals) are aligned by the one-thread DP with 32-bit trace rows, the rare wider
            // ones by 16 lanes each (alignMatchGroup): the lists are split by it when the batch's error budget exceeds 3.
            const bool splitWide = maxMatchErr > 3;
            ctx->smallList.reserve((size_t)nReads * 8 + 64);
            u32 * smallCount = ctx->counters.as<u32>() + 5, * smallWideCount = ctx->counters.as<u32>() + COUNTER_SMALL_WIDE;
            u32 * smallWideList = ctx->smallList.as<u32>() + nReads;
            hipLaunchKernelGGL(k_finalize_simple, dim3((nReads + 1023) / 1024), dim3(1024), 0, st, fa, ctx->complexList.as<u32>(), complexCount,
                               ctx->alignList.as<u32>(), alignCount, ctx->smallList.as<u32>(), smallCount, smallWideList, smallWideCount, splitWide ? 1u : 0u);
            ctx->dpList.reserve((size_t)nReads * 8 + 64);
            u32 * dpCount = ctx->counters.as<u32>() + 6, * dpWideCount = ctx->counters.as<u32>() + COUNTER_DP_WIDE;
            u32 * dpWideList = ctx->dpList.as<u32>() + nReads;
            // the list kernels touch disjoint reads: fork after the routing kernel, join before the stage ends
            HIP_CHECK(hipEventRecord(ctx->evFork, st));
            HIP_CHECK(hipStreamWaitEvent(ctx->side[0], ctx->evFork, 0));
            HIP_CHECK(hipStreamWaitEvent(ctx->side[1], ctx->evFork, 0));
            // thread-level DP kernels: 64-thread blocks, trace rows in LDS ((maxLen + 1) rows x 64 threads x 4 bytes); group kernels: 256-thread
            // blocks = 16 groups, each with its trace / sequences region
            const u32 dpLen = std::min<u32>(reads->maxLen, THREAD_DP_MAX_LEN);
            const u32 traceRows = dpLen + 1;
            const u32 dpThreads = 64;
            const size_t traceLds = (size_t)traceRows * dpThreads * 4;
            const u32 groupLds = groupDpLdsBytes(dpLen);
            // three chains: the thread-level aggregation on side[0], the alignment lists on side[1], the wavefront-level aggregation on the main
            // stream itself (a fourth stream shared a hardware queue with another one — HIP deals its streams over four — and on reads with
            // 1-3 % errors, where every chain is long, two of them ran one after the other: 568 us instead of 405)
            hipLaunchKernelGGL((k_finalize_small<3, u32>), dim3((u32)ctx->tune.dpSmallBlocks), dim3(dpThreads), traceLds, ctx->side[0], fa, ctx->smallList.as<u32>(), smallCount);
            hipLaunchKernelGGL(k_finalize_t<BlockGroup>, dim3(hugeBlocks), dim3(1024), 0, ctx->side[0], fa, blockMin + 1, 0xFFFFFFFFu, finWaves);
            hipLaunchKernelGGL(k_align_check, dim3(1024), dim3(256), 0, ctx->side[1], fa, ctx->alignList.as<u32>(), alignCount, ctx->dpList.as<u32>(), dpCount, dpWideList, dpWideCount);
            if (splitWide) hipLaunchKernelGGL(k_align_group, dim3(1024), dim3(256), 16 * groupLds, ctx->side[1], fa, dpWideList, dpWideCount, groupLds);
            hipLaunchKernelGGL((k_align_single<3, u32>), dim3((u32)ctx->tune.dpAlignBlocks), dim3(dpThreads), traceLds, ctx->side[1], fa, ctx->dpList.as<u32>(), dpCount);
            for (int q = 0; q < 2; ++q) HIP_CHECK(hipEventRecord(ctx->evJoin[q], ctx->side[q]));
            hipLaunchKernelGGL(k_finalize_t<WaveGroup>, dim3(finBlocks), dim3(256), 0, st, fa, 0u, blockMin, 0u);
            if (splitWide) hipLaunchKernelGGL(k_finalize_small_group, dim3(1024), dim3(256), 16 * groupLds, st, fa, smallWideList, smallWideCount, groupLds);
            // (Measured and dropped: starting a deferred sweep of the NEXT batch here, on the main stream, beside the three chains — a few hundred
            // wavefronts that spend ~0.2 ms on per-wave latency.  5.099 vs 5.088 ms per step: the sweep's half a million workgroups win the
            // dispatch, the chains run in its tail like every other kernel queued beside a sweep.)
            for (int q = 0; q < 2; ++q) HIP_CHECK(hipStreamWaitEvent(st, ctx->evJoin[q], 0));
        }
        else
        {
            // ---- paired-end stages (pe_kernels.h)
            ctx->peItems.reserve((size_t)nItems * sizeof(PeItem) + 64);
            ctx->peU2.reserve((size_t)nPool * sizeof(RawMatch) + 64);
            ctx->peLib.reserve((size_t)nBins * sizeof(PeLibrary) + 64);
            ctx->peLens.reserve((size_t)nItems * 4 + 64); ctx->peIdx.reserve((size_t)nItems * 4 + 64);
            ctx->peHasRescue.reserve((size_t)nReads * 4 + 64); ctx->peJobs.reserve((size_t)nItems * 4 + 64);
            if (owner)
            {
                HIP_CHECK(hipMemsetAsync(ctx->peItems.p, 0, (size_t)nItems * sizeof(PeItem), st));
                HIP_CHECK(hipMemsetAsync(ctx->peHasRescue.p, 0, (size_t)nReads * 4, st));
            }
            PeView V;
            V.items = ctx->peItems.as<PeItem>(); V.u2 = ctx->peU2.as<RawMatch>(); V.itemRead = ctx->itemRead.as<u32>(); V.itemBin = ctx->itemBin.as<u16>();
            V.binStart = ctx->binStart.as<u32>(); V.masks = reads->masks.as<u64>(); V.W = W; V.nBins = nBins; V.nPairs = nReads / 2; V.nItems = nItems;
            V.lib = ctx->peLib.as<PeLibrary>(); V.libLens = ctx->peLens.as<u32>(); V.libIdx = ctx->peIdx.as<u32>();
            V.optLibLength = p->library_length; V.optLibDev = p->library_dev; V.verify = p->verify_matches;
            V.errorRate = p->error_rate; V.indelRate = p->indel_rate; V.strataRate = p->strata_rate; V.readHasRescue = ctx->peHasRescue.as<u32>();
            if (owner)
            {
            ctx->dpList.reserve((size_t)nReads * 4 + 64);         // reads with more than one raw match (list of k_pe_child)
            u32 * childCount = ctx->counters.as<u32>() + 8;
            hipLaunchKernelGGL(k_pe_child_simple, dim3((nReads + 1023) / 1024), dim3(1024), 0, st, fa, V, ctx->dpList.as<u32>(), childCount);
            hipLaunchKernelGGL(k_pe_child, dim3(finBlocks), dim3(256), 0, st, fa, V, ctx->dpList.as<u32>(), childCount);
            hipLaunchKernelGGL(k_pe_library, dim3(nBins), dim3(1024), 0, st, V);
            if (nItems)
            {   // pairs whose mates have many unique matches (repeats) are paired by a wavefront each
                const u32 heavyMin = (u32)ctx->tune.peHeavyMin;
                ctx->peHeavy.reserve((size_t)nItems * 4 + 64);
                u32 * heavyCount = ctx->counters.as<u32>() + 11;
                hipLaunchKernelGGL(k_pe_pair, dim3((nItems + 255) / 256), dim3(256), 0, st, fa, V, heavyMin, ctx->peHeavy.as<u32>(), heavyCount);
                hipLaunchKernelGGL(k_pe_pair_heavy, dim3(1024), dim3(256), 0, st, fa, V, ctx->peHeavy.as<u32>(), heavyCount);
            }
            if (p->verify_matches && nItems)
            {
                std::vector<PeLibrary> hostLib(nBins);
                HIP_CHECK(hipMemcpyAsync(hostLib.data(), ctx->peLib.p, (size_t)nBins * sizeof(PeLibrary), hipMemcpyDeviceToHost, st));
                HIP_CHECK(hipStreamSynchronize(st));
                u32 maxWindow = 0;
                for (auto & l : hostLib) maxWindow = std::max(maxWindow, 6 * l.dev);
                if (maxWindow)
                {
                    u32 * jobCount = ctx->counters.as<u32>() + 6;
                    hipLaunchKernelGGL(k_pe_rescue_jobs, dim3((nItems + 255) / 256), dim3(256), 0, st, V, ctx->peJobs.as<u32>(), jobCount);
                    const u32 rescueBlocks = 1024, rescueWaves = rescueBlocks * 4;
                    const size_t stride = (peRescueScratchBytes(reads->maxLen, maxWindow) + 63) & ~(size_t)63;
                    ctx->peScratch.reserve(stride * rescueWaves + 64);
                    // the DP rows of a wavefront live in LDS when they fit (they do for ordinary libraries: 6 KB per wavefront at 100 bp / sigma 40)
                    const size_t ldsWave = peRescueLdsBytes(reads->maxLen, maxWindow);
                    const u32 ldsPerWave = ldsWave * 4 <= 60 * 1024 && !ctx->tune.peRescueGlobal ? (u32)ldsWave : 0u;
                    hipLaunchKernelGGL(k_pe_rescue, dim3(rescueBlocks), dim3(256), (size_t)ldsPerWave * 4, st, fa, V, ctx->peJobs.as<u32>(), jobCount, ctx->peScratch.as<u8>(), stride,
                                       maxWindow, ldsPerWave);
                }
            }
            }
            if (home)
            {
            // room per read in the main mapper's match array: 3 per raw match + one per selected bin (a rescued mate)
            ctx->peMCounts.reserve((size_t)(nReads + 1) * 4 + 64); ctx->peMStart.reserve((size_t)(nReads + 1) * 4 + 64);
            hipLaunchKernelGGL(k_pe_mcounts, dim3((nReads + 1 + 255) / 256), dim3(256), 0, st, ctx->readMatchCount.as<u32>(), reads->masks.as<u64>(), W, nReads, ctx->peMCounts.as<u32>());
            {
                size_t tb = 0;
                HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, ctx->peMCounts.as<u32>(), ctx->peMStart.as<u32>(), (int)(nReads + 1), st));
                ctx->scanTmp.reserve(tb + 64);
                HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(ctx->scanTmp.p, tb, ctx->peMCounts.as<u32>(), ctx->peMStart.as<u32>(), (int)(nReads + 1), st));
            }
            ctx->alignList.reserve((size_t)nReads * 4 + 64);
            u32 * peAlignCount = ctx->counters.as<u32>() + 9;
            ctx->smallList.reserve((size_t)nReads * 4 + 64);
            u32 * peSmallCount = ctx->counters.as<u32>() + 10;
            hipLaunchKernelGGL(k_pe_route, dim3((nReads + 1023) / 1024), dim3(1024), 0, st, fa, V, ctx->peMStart.as<u32>(), ctx->complexList.as<u32>(), complexCount,
                               ctx->alignList.as<u32>(), peAlignCount, ctx->smallList.as<u32>(), peSmallCount);
            {
                const bool narrowBand = maxMatchErr <= 3;
                const u32 traceRows = std::min<u32>(reads->maxLen, THREAD_DP_MAX_LEN) + 1;
                const u32 dpThreads = (size_t)traceRows * 64 * (narrowBand ? 4 : 8) > 60 * 1024 ? 32 : 64;
                const size_t traceLds = (size_t)traceRows * dpThreads * (narrowBand ? 4 : 8);
                // like the single-end stage: the list kernels touch disjoint reads and run side by side
                HIP_CHECK(hipEventRecord(ctx->evFork, st));
                HIP_CHECK(hipStreamWaitEvent(ctx->side[0], ctx->evFork, 0));
                HIP_CHECK(hipStreamWaitEvent(ctx->side[1], ctx->evFork, 0));
                if (narrowBand)
                {
                    hipLaunchKernelGGL((k_pe_align_single<3, u32>), dim3((u32)ctx->tune.dpSmallBlocks), dim3(dpThreads), traceLds, ctx->side[0], fa, V, ctx->peMStart.as<u32>(), ctx->alignList.as<u32>(), peAlignCount);
                    hipLaunchKernelGGL((k_pe_finalize_small<3, u32>), dim3((u32)ctx->tune.dpSmallBlocks), dim3(dpThreads), traceLds, ctx->side[1], fa, V, ctx->peMStart.as<u32>(), ctx->smallList.as<u32>(), peSmallCount);
                }
                else
                {
                    hipLaunchKernelGGL((k_pe_align_single<7, u64>), dim3((u32)ctx->tune.dpSmallBlocks), dim3(dpThreads), traceLds, ctx->side[0], fa, V, ctx->peMStart.as<u32>(), ctx->alignList.as<u32>(), peAlignCount);
                    hipLaunchKernelGGL((k_pe_finalize_small<7, u64>), dim3((u32)ctx->tune.dpSmallBlocks), dim3(dpThreads), traceLds, ctx->side[1], fa, V, ctx->peMStart.as<u32>(), ctx->smallList.as<u32>(), peSmallCount);
                }
                HIP_CHECK(hipEventRecord(ctx->evJoin[0], ctx->side[0]));
                HIP_CHECK(hipEventRecord(ctx->evJoin[1], ctx->side[1]));
            }
            hipLaunchKernelGGL(k_pe_finalize, dim3(finBlocks), dim3(256), 0, st, fa, V, ctx->peMStart.as<u32>());
            HIP_CHECK(hipStreamWaitEvent(st, ctx->evJoin[0], 0));
            HIP_CHECK(hipStreamWaitEvent(st, ctx->evJoin[1], 0));
            }
        }
        HIP_CHECK(hipGetLastError());
}

// MAPQ — reference src/bits_matches.h:898-949, src/mapper_writer.h:221-225 (fp64, host libm like the reference)
static double residualWeight(double errorRate) { double p = std::pow(10.0, std::min(2.0, 100.0 * errorRate - 7.0)); return p / (1.0 - p); }
static double matchWeight(double errorRate, double optimalRate) { return (1.0 - errorRate) / std::pow(10.0, 300.0 * (errorRate - optimalRate)); }
static unsigned mapqOf(unsigned errors, unsigned readLen, u64 best, u64 sub)
{
    double rate = (double)((float)errors / (float)readLen);
    double w = best * matchWeight(rate, rate) + sub * matchWeight(rate + 0.01, rate) + residualWeight(rate);
    double prob = matchWeight(rate, rate) / w;
    return (unsigned)std::round(-10.0 * std::log10(1.0 - std::min(prob, 0.9999999)));
}

// a rescued mate may carry up to maxErrors + maxIndels 'errors' (score / -999, reference src/find_verifier.h:112-139)
static u32 maxMatchErrorsOf(const dy_reads * reads, const dy_params * p, u32 maxErr)
{
    if (!(p->paired && p->verify_matches)) return maxErr;
    const u32 mi = (u32)((float)reads->maxLen * p->indel_rate);
    return std::min<u32>(MAX_ERRORS, maxErr + std::min<u32>(mi, MAX_ERRORS));
}

// ---------------------------------------------------------------------------------------------------
static std::unique_ptr<dy_result> makeResult(dy_ctx * ctx, const dy_reads * reads, u32 maxMatchErr)
{
    std::unique_ptr<dy_result> res(new dy_result);
    res->ctx = ctx; res->nReads = reads->n; res->cigarStride = 2 * maxMatchErr + 1;
    takeBuffer(ctx, res->out, (size_t)reads->n * sizeof(ReadOut) + 64);
    takeBuffer(ctx, res->cigars, (size_t)reads->n * res->cigarStride * 4 + 64);
    return res;
}

template <typename TIn, typename TOut>
static void exclusiveSum(dy_ctx * ctx, const TIn * in, TOut * out, size_t n)
{
    size_t tmpBytes = 0;
    HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmpBytes, in, out, (int)n, ctx->stream));
    ctx->scanTmp.reserve(tmpBytes + 64);
    HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(ctx->scanTmp.p, tmpBytes, in, out, (int)n, ctx->stream));
}

extern "C" {

int dy_abi_version(void) { return DY_ABI_VERSION; }
const char * dy_last_error(void) { return g_lastError.c_str(); }
int dy_device_count(void) { return deviceCount(); }

int dy_ctx_create(int device, dy_ctx ** out)
{
    return guarded([&] {
        if (!out) throw DyError(DY_EINVAL, "dy_ctx_create: out is NULL");
        int n = deviceCount();
        if (n <= 0) throw DyError(DY_ENODEVICE, "no HIP device visible: libdyhip has no CPU fallback");
        if (device < 0 || device >= n) throw DyError(DY_EINVAL, "device ordinal out of range");
        HIP_CHECK(hipSetDevice(device));
        std::unique_ptr<dy_ctx> c(new dy_ctx);
        c->device = device;
        HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        HIP_CHECK(hipStreamCreateWithFlags(&c->ibfStream, hipStreamNonBlocking));
        for (auto & e : c->ev) HIP_CHECK(hipEventCreate(&e));
        for (auto & q : c->side) HIP_CHECK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&c->evFork, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&c->evFetch, hipEventDisableTiming));
        for (auto & e : c->evJoin) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        *out = c.release();
    });
}

void dy_ctx_free(dy_ctx * ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->ibfStream) { (void)hipStreamSynchronize(ctx->ibfStream); (void)hipStreamDestroy(ctx->ibfStream); }
    if (ctx->stream) { (void)hipStreamSynchronize(ctx->stream); }
    for (auto & e : ctx->ev) if (e) (void)hipEventDestroy(e);
    for (auto & q : ctx->side) if (q) (void)hipStreamDestroy(q);
    if (ctx->evFork) (void)hipEventDestroy(ctx->evFork);
    if (ctx->evFetch) (void)hipEventDestroy(ctx->evFetch);
    for (auto & T : ctx->sweepTiming) for (auto & e : T.ev) if (e) (void)hipEventDestroy(e);
    for (auto & e : ctx->evJoin) if (e) (void)hipEventDestroy(e);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int dy_ctx_sync(dy_ctx * ctx)
{
    return guarded([&] {
        if (!ctx) throw DyError(DY_EINVAL, "ctx is NULL");
        HIP_CHECK(hipStreamSynchronize(ctx->ibfStream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

void * dy_ctx_stream(dy_ctx * ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int dy_ctx_set_tuning(dy_ctx * ctx, const char * key, uint64_t value)
{
    return guarded([&] {
        if (!ctx || !key) throw DyError(DY_EINVAL, "dy_ctx_set_tuning: NULL argument");
        Tuning & t = ctx->tune;
        const struct { const char * name; u64 * slot; u64 lo, hi; } keys[] = {
            {"sparse_sa", &t.sparseSa, 0, 1}, {"prefix_q", &t.prefixQ, 0, 12}, {"prefix_wide", &t.prefixWide, 0, 1}, {"extend_mode", &t.extendMode, 0, 3}, {"markers", &t.markers, 0, 1}, {"no_isa", &t.noIsa, 0, 1},
            {"finalize_block_min", &t.finalizeBlockMin, 1, 0xFFFFFFF0ull}, {"finalize_no_keys", &t.finalizeNoKeys, 0, 1},
            {"pe_heavy_min", &t.peHeavyMin, 1, 0xFFFFFFF0ull}, {"pe_rescue_global", &t.peRescueGlobal, 0, 1},
            {"ibf_narrow", &t.ibfNarrow, 0, 1}, {"ibf_full_tail", &t.ibfFullTail, 0, 1}, {"ibf_lds_kb", &t.ibfLdsKb, 4, 60}, {"ibf_dense", &t.ibfDense, 0, 1},
            {"ibf_kernel", &t.ibfKernel, 0, 3}, {"ibf_slack", &t.ibfSlack, 0, 255}, {"ibf_own_stream", &t.ibfOwnStream, 0, 1},
            {"pool_cap", &t.poolCap, 0, 0xFFFFFFF0ull}, {"dp_small_blocks", &t.dpSmallBlocks, 64, 65536}, {"dp_align_blocks", &t.dpAlignBlocks, 64, 65536}, {"items_sorted", &t.itemsSorted, 0, 1}, {"sweep_chunk", &t.sweepChunk, 0, 2097151}, {"async_map", &t.asyncMap, 0, 1},
        };
        for (auto & k : keys)
            if (!strcmp(k.name, key))
            {
                if (value < k.lo || value > k.hi) throw DyError(DY_EINVAL, std::string("dy_ctx_set_tuning: value out of range for ") + key);
                *k.slot = value;
                if (k.slot == &t.poolCap) ctx->poolCap = 0;      // takes effect at the next batch
                return;
            }
        throw DyError(DY_EINVAL, std::string("dy_ctx_set_tuning: unknown key ") + key);
    });
}

// ---- IBF ----------------------------------------------------------------------------------------------
static void initGeom(IbfGeom & g, u32 bins, u32 hashes, u32 k, u64 bits)
{
    if (bins == 0 || hashes == 0 || hashes > 8) throw DyError(DY_EINVAL, "IBF: number of bins/hash functions out of range");
    if (k == 0 || k > 32) throw DyError(DY_EUNSUPPORTED, "IBF: k-mer size must be in [1,32]");
    if (bits < 256 + 64) throw DyError(DY_EINVAL, "IBF: filter too small");
    if (bits % 64) throw DyError(DY_EUNSUPPORTED, "IBF: bit count must be a multiple of 64");
    memset(&g, 0, sizeof(g));
    g.nBits = bits; g.nBins = bins; g.nHashes = hashes; g.kmerSize = k;
    g.W = (bins + 63) / 64;                                    // reference src/d_bloom_filter.h:247-249
    if (g.W > 64) throw DyError(DY_EUNSUPPORTED, "IBF: more than 4096 bins are not supported");
    g.nHashPos = (bits - 256) / (64ull * g.W);
    if (g.nHashPos == 0) throw DyError(DY_EINVAL, "IBF: filter too small for the number of bins");
    g.hashPosMask = (g.nHashPos & (g.nHashPos - 1)) == 0 ? g.nHashPos - 1 : 0;
    if (g.nHashPos == 1) g.hashPosMask = 0;
    g.hashPosMagic = g.nHashPos ? ~0ull / g.nHashPos : 0;
    for (u32 i = 0; i < hashes; ++i) g.preCalc[i] = (u64)i ^ ((u64)k * 0x90b45d39fb6da1faull);    // :251-253
}

int dy_ibf_create(dy_ctx * ctx, uint32_t n_bins, uint32_t n_hashes, uint32_t kmer_size, uint64_t n_bits, dy_ibf ** out)
{
    return guarded([&] {
        if (!ctx || !out) throw DyError(DY_EINVAL, "dy_ibf_create: NULL argument");
        HIP_CHECK(hipSetDevice(ctx->device));
        std::unique_ptr<dy_ibf> f(new dy_ibf);
        f->ctx = ctx;
        initGeom(f->g, n_bins, n_hashes, kmer_size, n_bits);
        f->nWords = (size_t)(n_bits / 64);
        f->words.reserveExact(f->nWords * 8 + 8);
        HIP_CHECK(hipMemsetAsync(f->words.p, 0, f->nWords * 8 + 8, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        *out = f.release();
    });
}

int dy_ibf_load(dy_ctx * ctx, const char * path, dy_ibf ** out)
{
    return guarded([&] {
        if (!ctx || !path || !out) throw DyError(DY_EINVAL, "dy_ibf_load: NULL argument");
        HIP_CHECK(hipSetDevice(ctx->device));
        FilterFile ff(path);
        u64 ms = ff.nBits - 256;                                // reference src/d_bloom_filter.h:255-265
        u32 bins = (u32)ff.getInt(ms), hashes = (u32)ff.getInt(ms + 64), k = (u32)ff.getInt(ms + 128);
        std::unique_ptr<dy_ibf> f(new dy_ibf);
        f->ctx = ctx;
        initGeom(f->g, bins, hashes, k, ff.nBits);
        f->nWords = ff.nWords;
        f->words.reserveExact(f->nWords * 8 + 8);
        HIP_CHECK(hipMemcpy(f->words.p, ff.words, f->nWords * 8, hipMemcpyHostToDevice));
        *out = f.release();
    });
}

// ---- k-mer direct-addressing filter (-ft kmer_direct): same handle, same query / add / save entry points
static void makeDirect(dy_ctx * ctx, dy_ibf * f)
{
    f->g.direct = 1; f->g.nHashes = 1;
    const u64 positions = f->g.kmerSize >= 32 ? ~0ull : (1ull << (2 * f->g.kmerSize));
    if (f->g.nHashPos > positions) f->g.nHashPos = positions;
    // zeroed guard position behind the table (it overlays the trailer, whose fields live in IbfGeom)
    const size_t guard = (size_t)f->g.nHashPos * f->g.W;
    HIP_CHECK(hipMemsetAsync(f->words.as<u64>() + guard, 0, (size_t)f->g.W * 8, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
}

int dy_kdx_create(dy_ctx * ctx, uint32_t n_bins, uint32_t kmer_size, dy_ibf ** out)
{
    return guarded([&] {
        if (!ctx || !out) throw DyError(DY_EINVAL, "dy_kdx_create: NULL argument");
        if (kmer_size == 0 || kmer_size > 16) throw DyError(DY_EUNSUPPORTED, "kmer_direct: k-mer size must be in [1,16]");
        HIP_CHECK(hipSetDevice(ctx->device));
        const u64 W = ((u64)n_bins + 63) / 64;
        const u64 bits = (1ull << (2 * kmer_size)) * 64 * W + 256;      // reference src/d_build_filter.cpp:307-309 (bins a multiple of 64)
        std::unique_ptr<dy_ibf> f(new dy_ibf);
        f->ctx = ctx;
        initGeom(f->g, n_bins, 1, kmer_size, bits);
        f->nWords = (size_t)(bits / 64);
        f->words.reserveExact((f->nWords + W + 8) * 8);
        HIP_CHECK(hipMemsetAsync(f->words.p, 0, (f->nWords + W + 8) * 8, ctx->stream));
        makeDirect(ctx, f.get());
        *out = f.release();
    });
}

int dy_kdx_load(dy_ctx * ctx, const char * path, dy_ibf ** out)
{
    return guarded([&] {
        if (!ctx || !path || !out) throw DyError(DY_EINVAL, "dy_kdx_load: NULL argument");
        HIP_CHECK(hipSetDevice(ctx->device));
        FilterFile ff(path);
        u64 ms = ff.nBits - 256;                                // reference src/d_kdx_filter.h:218-227
        u32 bins = (u32)ff.getInt(ms), k = (u32)ff.getInt(ms + 128);
        std::unique_ptr<dy_ibf> f(new dy_ibf);
        f->ctx = ctx;
        initGeom(f->g, bins, 1, k, ff.nBits);
        f->nWords = ff.nWords;
        f->words.reserveExact((f->nWords + f->g.W + 8) * 8);
        HIP_CHECK(hipMemcpy(f->words.p, ff.words, f->nWords * 8, hipMemcpyHostToDevice));
        makeDirect(ctx, f.get());
        *out = f.release();
    });
}

int dy_ibf_add(dy_ibf * f, const uint8_t * seq, uint64_t len, uint32_t bin)
{
    return guarded([&] {
        if (f) HIP_CHECK(hipStreamSynchronize(f->ctx->ibfStream));      // no sweep may be reading the filter
        if (!f || (!seq && len)) throw DyError(DY_EINVAL, "dy_ibf_add: NULL argument");
        if (bin >= f->g.nBins) throw DyError(DY_EINVAL, "dy_ibf_add: bin out of range");
        if (len < f->g.kmerSize) return;                        // reference src/d_bloom_filter.h:142-143
        dy_ctx * ctx = f->ctx;
        HIP_CHECK(hipSetDevice(ctx->device));
        DevBuf d; d.reserve(len);
        HIP_CHECK(hipMemcpyAsync(d.p, seq, len, hipMemcpyHostToDevice, ctx->stream));
        u64 nk = len - f->g.kmerSize + 1;
        u32 blocks = (u32)((nk + 255) / 256);
        f->fillKnown = false;
        hipLaunchKernelGGL(k_ibf_insert, dim3(blocks), dim3(256), 0, ctx->stream, f->words.as<u64>(), f->g, d.as<u8>(), len, bin);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

int dy_ibf_clear_bins(dy_ibf * f, const uint32_t * bins, uint32_t n_bins)
{
    return guarded([&] {
        if (f) HIP_CHECK(hipStreamSynchronize(f->ctx->ibfStream));      // no sweep may be reading the filter
        if (!f || (!bins && n_bins)) throw DyError(DY_EINVAL, "dy_ibf_clear_bins: NULL argument");
        std::vector<u64> mask(f->g.W, 0);
        for (u32 i = 0; i < n_bins; ++i)
        {
            if (bins[i] >= f->g.nBins) throw DyError(DY_EINVAL, "dy_ibf_clear_bins: bin out of range");
            mask[bins[i] / 64] |= 1ull << (bins[i] % 64);
        }
        if (!n_bins) return;
        dy_ctx * ctx = f->ctx;
        HIP_CHECK(hipSetDevice(ctx->device));
        DevBuf d; upload(d, mask);
        const u64 nTable = f->g.nHashPos * f->g.W;
        f->fillKnown = false;
        hipLaunchKernelGGL(k_ibf_clear_bins, dim3(256 * 32), dim3(256), 0, ctx->stream, f->words.as<u64>(), nTable, f->g.W, d.as<u64>());
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

int dy_ibf_save(dy_ibf * f, const char * path)
{
    return guarded([&] {
        if (f) HIP_CHECK(hipStreamSynchronize(f->ctx->ibfStream));      // no sweep may be reading the filter
        if (!f || !path) throw DyError(DY_EINVAL, "dy_ibf_save: NULL argument");
        HIP_CHECK(hipSetDevice(f->ctx->device));
        std::vector<u64> host(f->nWords);
        HIP_CHECK(hipMemcpy(host.data(), f->words.p, f->nWords * 8, hipMemcpyDeviceToHost));
        size_t ms = (size_t)((f->g.nBits - 256) / 64);          // _setMetadata, reference src/d_bloom_filter.h:267-277
        host[ms] = f->g.nBins; host[ms + 1] = f->g.direct ? 0 : f->g.nHashes; host[ms + 2] = f->g.kmerSize; host[ms + 3] = 0;
        FILE * fp = fopen(path, "wb");
        if (!fp) throw IoError(std::string("cannot write ") + path);
        bool ok = fwrite(&f->g.nBits, 8, 1, fp) == 1 && fwrite(host.data(), 8, host.size(), fp) == host.size();
        fclose(fp);
        if (!ok) throw IoError(std::string("short write to ") + path);
    });
}

int dy_ibf_read_words(dy_ibf * f, uint64_t first_word, uint64_t n_words, uint64_t * words_host)
{
    return guarded([&] {
        if (f) HIP_CHECK(hipStreamSynchronize(f->ctx->ibfStream));      // no sweep may be reading the filter
        if (!f || (!words_host && n_words)) throw DyError(DY_EINVAL, "dy_ibf_read_words: NULL argument");
        if (first_word > f->nWords || n_words > f->nWords - first_word) throw DyError(DY_EINVAL, "dy_ibf_read_words: range exceeds the filter");
        if (!n_words) return;
        HIP_CHECK(hipSetDevice(f->ctx->device));
        HIP_CHECK(hipStreamSynchronize(f->ctx->stream));
        // large copies go chunk by chunk so that the runtime's pageable staging stays small
        const u64 CH = (u64)1 << 27;
        for (u64 at = 0; at < n_words; at += CH)
            HIP_CHECK(hipMemcpy(words_host + at, f->words.as<u64>() + first_word + at, (size_t)std::min(CH, n_words - at) * 8, hipMemcpyDeviceToHost));
        const u64 ms = (f->g.nBits - 256) / 64;                 // _setMetadata, reference src/d_bloom_filter.h:267-277
        const u64 meta[4] = {f->g.nBins, f->g.direct ? 0u : f->g.nHashes, f->g.kmerSize, 0};
        for (u64 i = 0; i < 4; ++i)
            if (ms + i >= first_word && ms + i < first_word + n_words) words_host[ms + i - first_word] = meta[i];
    });
}

int dy_ibf_get_info(const dy_ibf * f, dy_ibf_info * info)
{
    if (!f || !info) { g_lastError = "NULL argument"; return DY_EINVAL; }
    info->n_bits = f->g.nBits; info->n_bins = f->g.nBins; info->n_hashes = f->g.nHashes; info->kmer_size = f->g.kmerSize;
    info->words_per_pos = f->g.W; info->n_hash_pos = f->g.nHashPos;
    return DY_OK;
}

void dy_ibf_free(dy_ibf * f) { if (f) { if (f->ctx->pending.f == f) { f->ctx->pending.f = nullptr; f->ctx->pending.reads = nullptr; } (void)hipSetDevice(f->ctx->device); (void)hipStreamSynchronize(f->ctx->ibfStream); (void)hipStreamSynchronize(f->ctx->stream); delete f; } }

uint16_t dy_ibf_threshold(float error_rate, uint32_t kmer_size, uint16_t read_len)
{
    uint16_t maxError = (uint16_t)(error_rate * read_len);
    if (kmer_size * (1u + maxError) > read_len) return 0;
    return (uint16_t)(read_len - kmer_size * (1u + maxError) + 1);
}

// ---- reads --------------------------------------------------------------------------------------------
int dy_reads_upload(dy_ctx * ctx, const uint8_t * seqs, const uint64_t * offs, uint32_t n_reads, dy_reads ** out)
{
    return guarded([&] {
        if (!ctx || !offs || !out || (!seqs && n_reads && offs[n_reads])) throw DyError(DY_EINVAL, "dy_reads_upload: NULL argument");
        if (n_reads > 2097151u) throw DyError(DY_EINVAL, "a batch holds at most 2097151 reads (21-bit read id)");
        HIP_CHECK(hipSetDevice(ctx->device));
        std::unique_ptr<dy_reads> r(new dy_reads);
        r->ctx = ctx; r->n = n_reads; r->totalBases = offs[n_reads];
        u32 mx = 0, mn = 0xFFFFFFFFu;
        r->lens->resize(n_reads);
        u32 * lens = r->lens->data();
        for (u32 i = 0; i < n_reads; ++i)
        {
            if (offs[i + 1] < offs[i]) throw DyError(DY_EINVAL, "read offsets must be non-decreasing");
            u64 l = offs[i + 1] - offs[i];
            if (l > 8191) throw DyError(DY_EINVAL, "Maximum read length exceeded.");      // reference src/mapper.h:453-454
            mx = std::max<u32>(mx, (u32)l); mn = std::min<u32>(mn, (u32)l);
            lens[i] = (u32)l;
        }
        r->maxLen = mx; r->minLen = n_reads ? mn : 0;
        // device buffers come from (and go back to) the context's pool: a batch per step would otherwise pay hipMalloc / hipFree
        r->recycle = true;
        takeBuffer(ctx, r->seqs, r->totalBases + 64);
        takeBuffer(ctx, r->offs, (size_t)(n_reads + 1) * 8);
        takeBuffer(ctx, r->seqsRc, r->totalBases + 64);
        // The caller's memory is pageable: a few threads copy it chunk by chunk into a pinned staging buffer while the chunks already
        // staged are on their way to the device (the runtime's own pageable path stages with one thread).
        const size_t total = r->totalBases, CH = (size_t)8 << 20, nChunks = (total + CH - 1) / CH;
        if (nChunks >= 4)
        {
            ctx->hostUp.reserve(total + 64);
            u8 * stage = ctx->hostUp.as<u8>();
            std::vector<std::atomic<int>> ready(nChunks);
            for (auto & f : ready) f.store(0, std::memory_order_relaxed);
            const unsigned T = (unsigned)std::min<size_t>(4, nChunks);
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < T; ++t)
                pool.emplace_back([&, t] {
                    for (size_t c = t; c < nChunks; c += T)
                    {
                        const size_t at = c * CH, len = std::min(CH, total - at);
                        memcpy(stage + at, seqs + at, len);
                        ready[c].store(1, std::memory_order_release);
                    }
                });
            hipError_t err = hipSuccess;
            for (size_t c = 0; c < nChunks; ++c)
            {
                while (!ready[c].load(std::memory_order_acquire)) std::this_thread::yield();
                const size_t at = c * CH, len = std::min(CH, total - at);
                if (err == hipSuccess) err = hipMemcpyAsync(r->seqs.as<u8>() + at, stage + at, len, hipMemcpyHostToDevice, ctx->stream);
            }
            for (auto & th : pool) th.join();
            HIP_CHECK(err);
        }
        else if (total) HIP_CHECK(hipMemcpyAsync(r->seqs.p, seqs, total, hipMemcpyHostToDevice, ctx->stream));
        HIP_CHECK(hipMemcpyAsync(r->offs.p, offs, (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        if (n_reads) hipLaunchKernelGGL(k_revcomp_reads, dim3(std::min<u32>((n_reads + 15) / 16, 65536u)), dim3(256), 0, ctx->stream, r->seqs.as<u8>(), r->seqsRc.as<u8>(), r->offs.as<u64>(), n_reads);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        *out = r.release();
    });
}

uint32_t dy_reads_count(const dy_reads * r) { return r ? r->n : 0; }
void dy_reads_free(dy_reads * r)
{
    if (!r) return;
    (void)hipSetDevice(r->ctx->device);
    if (r->k1Pending) (void)hipEventSynchronize(r->evK1[1]);      // a sweep nobody waited for still writes the masks
    if (r->ctx->shBatch == r) r->ctx->shBatch = nullptr;
    if (r->ctx->pending.reads == r) { (void)hipStreamSynchronize(r->ctx->stream); r->ctx->pending.reads = nullptr; r->ctx->pending.f = nullptr; }     // chunks already queued write its masks
    for (auto & e : r->evK1) if (e) (void)hipEventDestroy(e);
    if (r->recycle) { giveBuffer(r->ctx, r->seqs); giveBuffer(r->ctx, r->seqsRc); giveBuffer(r->ctx, r->offs); giveBuffer(r->ctx, r->masks); giveBuffer(r->ctx, r->orig);
                      giveBuffer(r->ctx, r->fqText); giveBuffer(r->ctx, r->fqRecs); }
    delete r;
}

// The sweep of reads [first, first + count) of `reads` on stream sq (K1: whichBins of both strands, reference src/d_bloom_filter.h:161-224 through
// src/d_mapper.h:350-358).  The kernels index a read by its wavefront, so a sub-range is the same launch on shifted offsets / masks.
static void launchSweep(dy_ibf * f, dy_reads * reads, u32 threshold, u32 first, u32 count, hipStream_t sq)
{
            dy_ctx * ctx = f->ctx;
            StageRange sweepRange(ctx->tune.markers != 0, "dy:sweep");
            const u32 W = f->g.W;
            const u64 * offs = reads->offs.as<u64>() + first;
            u64 * masks = reads->masks.as<u64>() + (size_t)first * W;
            const bool longReads = reads->maxLen >= 127 + f->g.kmerSize;
            const u32 nKmax = reads->maxLen >= f->g.kmerSize ? reads->maxLen - f->g.kmerSize + 1 : 0;
            u32 gShift = 0;
            while ((1u << gShift) < W && gShift < 4) ++gShift;
            const u32 G = 1u << gShift;
            // LDS per wave of the two sweeps for W > 1 (see ibf_kernels.h)
            const size_t hs = ((size_t)nKmax + 31) & ~(size_t)31;
            const size_t lazyWave = (2 * hs + 128 + (size_t)nKmax * G + 33 * G) * 8;
            const u32 listCap = 384;                            // surviving bits of a strand: one per k-mer of the read's own bin + ~0.5 false ones per k-mer
            const size_t eagerWave = (hs + 128 + 17 * G + 1 + (listCap + 1) / 2) * 8;
            u32 kernel = 3;
            if (W > 1 && !ctx->tune.ibfNarrow && nKmax >= 1 && nKmax <= 255)
            {
                kernel = (u32)ctx->tune.ibfKernel;
                if (kernel == 0) kernel = ibfExpectedFalseCandidates(f, nKmax, threshold) < 0.5 ? 1 : 2;
                if ((kernel == 1 && lazyWave > 60 * 1024) || (kernel == 2 && eagerWave > 60 * 1024)) kernel = 3;
            }
            if (kernel == 1 || kernel == 2)
            {
                u32 slack = ibfSlack(f, nKmax, threshold);                      // (the fill estimate may use the stream: the timed region starts behind it)
                // the lazy sweep gains from a slightly later first stop than the "false bins have died" point (256 bins / 16 GiB, 100 bp:
                // slack 8 -> 4.32-4.42 ms, 9 -> 4.29, 10 -> 4.23-4.25, 11 -> 4.36 per 1 M reads): more of the own bin's carriers are at hand
                if (kernel == 1 && !ctx->tune.ibfSlack) slack = std::min<u32>(slack + 2, (u32)threshold);
                const size_t perWave = kernel == 1 ? lazyWave : eagerWave;
                const u32 wavesPerBlock = perWave * 4 <= 32 * 1024 ? 4 : perWave * 2 <= 32 * 1024 ? 2 : 1;
                const dim3 grid((count + wavesPerBlock - 1) / wavesPerBlock), block(64 * wavesPerBlock);
                ctx->stats.ibf_kernel = kernel == 1 ? 4 : 5; ctx->stats.ibf_slack = slack;
                if (kernel == 1)
                    hipLaunchKernelGGL(k_ibf_query_lazy, grid, block, perWave * wavesPerBlock, sq, f->words.as<u64>(), f->g, reads->seqs.as<u8>(), offs,
                                       count, (u32)threshold, masks, gShift, nKmax, slack);
                else
                    hipLaunchKernelGGL(k_ibf_query_eager2, grid, block, perWave * wavesPerBlock, sq, f->words.as<u64>(), f->g, reads->seqs.as<u8>(), offs,
                                       count, (u32)threshold, masks, gShift, nKmax, listCap);
            }
            else if (W == 1 || ctx->tune.ibfNarrow)
            {
                const u32 wavesPerBlock = 4;
                const dim3 grid((count + wavesPerBlock - 1) / wavesPerBlock), block(64 * wavesPerBlock);
                if (W == 1 && nKmax > 64 && nKmax <= 127 && !ctx->tune.ibfFullTail)
                {
                    const u32 slack = ibfSlack(f, nKmax, threshold);
                    // tail k-mers requested with the first round: worth their look-ups only where a strand is expected to leave the first
                    // round with a weak bin (a bin that reaches threshold - tail hits among the first 64 k-mers by chance): 4 at the
                    // fill of a 1 GiB filter (0.97 weak bins per strand), none at 16 GiB
                    u32 tailPre = 0;
                    {
                        const u32 n0 = std::min<u32>(std::min<u32>(nKmax, 64u), std::max<u32>(threshold, nKmax - std::min<u32>(threshold, nKmax) + slack));
                        const u32 tail = nKmax - n0;
                        if (tail && threshold > tail && ibfExpectedFalseCandidates(f, n0, threshold - tail) >= 0.5) tailPre = 4;
                    }
                        ctx->stats.ibf_kernel = 1; ctx->stats.ibf_slack = slack; ctx->stats.ibf_tail_pre = tailPre;
                    hipLaunchKernelGGL(k_ibf_query_two, grid, block, 0, sq, f->words.as<u64>(), f->g, reads->seqs.as<u8>(), offs,
                                       count, (u32)threshold, masks, slack, tailPre);
                }
                else if (!longReads)
                {
                    ctx->stats.ibf_kernel = 2;
                    hipLaunchKernelGGL(k_ibf_query<7>, grid, block, 0, sq, f->words.as<u64>(), f->g, reads->seqs.as<u8>(), offs,
                                       count, (u32)threshold, masks);
                }
                else
                {
                    ctx->stats.ibf_kernel = 3;
                    hipLaunchKernelGGL(k_ibf_query<14>, grid, block, 0, sq, f->words.as<u64>(), f->g, reads->seqs.as<u8>(), offs,
                                       count, (u32)threshold, masks);
                }
            }
            else
            {   // G = lanes per k-mer; LDS per wave = 64 hashes + the hash-0 words of as many k-mers as fit in ~19 KB
                const u32 nK = nKmax;
                ctx->stats.ibf_kernel = 6;
                const u32 hashSlots = nK <= 512 ? ((nK + 63) & ~63u) : 128;                   // all hashes of a strand, or a window
                // measured on the 1024-bin geometry: 19 KB per wave 41.3 ms, 12 KB 32.1 ms, 8 KB 34.4 ms (LDS limits the waves per CU; words not cached are gathered again)
                const u32 ldsBudget = (u32)ctx->tune.ibfLdsKb * 1024;
                // sparse verification pays when false candidates are expected in numbers, i.e. at thresholds far below the k-mer count
                const u32 sparseCap = (nK < 256 && nK <= hashSlots && 4u * threshold <= nK && !ctx->tune.ibfDense) ? 1024 : 0;
                const u32 fixedWords = hashSlots + (sparseCap ? 16 + 128 + sparseCap / 4 : 0);
                const u32 cacheKmers = ldsBudget > fixedWords * 8 ? std::min<u32>(nK, (u32)((ldsBudget - fixedWords * 8) / (G * 8))) : 0;
                const size_t perWave = ((size_t)fixedWords + (size_t)cacheKmers * G) * 8;
                const u32 wavesPerBlock = perWave > 8 * 1024 ? 2 : 4;
                const dim3 grid((count + wavesPerBlock - 1) / wavesPerBlock), block(64 * wavesPerBlock);
#define DY_LAUNCH_WIDE(NPV, SP) hipLaunchKernelGGL((k_ibf_query_wide<NPV, SP>), grid, block, perWave * wavesPerBlock, sq, f->words.as<u64>(), f->g, \
                                   reads->seqs.as<u8>(), offs, count, (u32)threshold, masks, gShift, cacheKmers, hashSlots, sparseCap)
                if (nK <= 127) { if (sparseCap) DY_LAUNCH_WIDE(7, true); else DY_LAUNCH_WIDE(7, false); }
                else if (nK <= 255) { if (sparseCap) DY_LAUNCH_WIDE(8, true); else DY_LAUNCH_WIDE(8, false); }
                else DY_LAUNCH_WIDE(14, false);
#undef DY_LAUNCH_WIDE
            }
            HIP_CHECK(hipGetLastError());
}

// ---- deferred sweep: K1 of the batch mapped NEXT, cut into chunks that are queued in front of the host's waits ----------------------
// A sweep holds every wave slot of the chip, so a second stream cannot run beside it: what CAN be overlapped are the waits of
// the host — for a count it has to read (work-list size, pool counter, the rows of an exchange) or for a peer.  A chunk of the pending
// sweep queued right before such a wait keeps the device busy while the host is away; the kernels the host queues afterwards run behind it.
static void sweepChunk(dy_ctx * ctx, u32 maxReads)
{
    PendingSweep & P = ctx->pending;
    if (!P.reads || P.next >= P.reads->n) return;
    const u32 count = std::min<u32>(maxReads ? maxReads : (u32)std::max<u64>(ctx->tune.sweepChunk, 1024), P.reads->n - P.next);
    SweepTiming & T = ctx->sweepTiming[P.slot];
    const bool timed = T.nChunks < SweepTiming::MAX_CHUNKS;
    if (timed) HIP_CHECK(hipEventRecord(T.ev[2 * T.nChunks], ctx->stream));
    launchSweep(P.f, P.reads, P.threshold, P.next, count, ctx->stream);
    if (timed) { HIP_CHECK(hipEventRecord(T.ev[2 * T.nChunks + 1], ctx->stream)); ++T.nChunks; }
    P.next += count;
}
static void sweepFlush(dy_ctx * ctx)
{
    PendingSweep & P = ctx->pending;
    if (P.reads && P.next < P.reads->n) sweepChunk(ctx, P.reads->n - P.next);
    if (P.reads) ctx->sweepTiming[P.slot].complete = true;
    P.reads = nullptr; P.f = nullptr;
}
// device time of a completed deferred sweep (its chunks' events), once everything queued has run
static void collectDeferredSweepTime(dy_ctx * ctx)
{
    for (SweepTiming & T : ctx->sweepTiming)
    {
        if (!T.complete || !T.nChunks) continue;
        if (hipEventQuery(T.ev[2 * T.nChunks - 1]) != hipSuccess) continue;
        double total = 0;
        for (u32 c = 0; c < T.nChunks; ++c) { float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, T.ev[2 * c], T.ev[2 * c + 1])); total += ms; }
        ctx->stats.ms_ibf = total;
        T.complete = false; T.nChunks = 0;
    }
}
// the host is about to wait for what it has just queued on the stream (a small fetch): a chunk of the pending sweep goes behind it first
static void fetchSync(dy_ctx * ctx)
{
    if (ctx->tune.sweepChunk && ctx->pending.reads && ctx->pending.next < ctx->pending.reads->n)
    {
        HIP_CHECK(hipEventRecord(ctx->evFetch, ctx->stream));
        sweepChunk(ctx, 0);
        HIP_CHECK(hipEventSynchronize(ctx->evFetch));
    }
    else HIP_CHECK(hipStreamSynchronize(ctx->stream));
}

int dy_ibf_query(dy_ibf * f, dy_reads * reads, uint16_t threshold, uint64_t * bin_masks_host)
{
    return guarded([&] {
        if (!f || !reads) throw DyError(DY_EINVAL, "dy_ibf_query: NULL argument");
        if (f->ctx != reads->ctx) throw DyError(DY_EINVAL, "filter and reads belong to different contexts");
        dy_ctx * ctx = f->ctx;
        HIP_CHECK(hipSetDevice(ctx->device));
        const u32 W = f->g.W;
        if (reads->recycle && !reads->masks.p) takeBuffer(ctx, reads->masks, (size_t)reads->n * W * 8 + 8);
        else reads->masks.reserve((size_t)reads->n * W * 8 + 8);
        reads->maskW = W; reads->maskBins = f->g.nBins;
        // The sweep runs on the context's second stream: a caller that queues the sweep of its NEXT batch before mapping the current one
        // gets the two overlapped (dy_map_reads makes the main stream wait for the batch's own sweep).  Everything this call reads was
        // put in place by synchronous calls (dy_reads_upload, dy_ibf_load / add), so no wait is needed here — except for a mapping that
        // may still be reading the previous masks of the SAME reads object on the main stream.
        // (tuning key ibf_own_stream; off by default: the hand-over between two streams costs ~25 us per batch, and a sweep that already
        // holds every wave slot overlaps with nothing — DESIGN.md section 8)
        const bool ownStream = ctx->tune.ibfOwnStream != 0;
        hipStream_t sq = ownStream ? ctx->ibfStream : ctx->stream;
        for (auto & e : reads->evK1) if (!e) HIP_CHECK(hipEventCreate(&e));
        if (ownStream)
        {
            HIP_CHECK(hipEventRecord(ctx->evFork, ctx->stream));
            HIP_CHECK(hipStreamWaitEvent(sq, ctx->evFork, 0));
        }
        if (reads->n)
        {
            if (ctx->pending.reads == reads) { ctx->pending.reads = nullptr; ctx->pending.f = nullptr; }      // superseded
            HIP_CHECK(hipEventRecord(reads->evK1[0], sq));
            launchSweep(f, reads, (u32)threshold, 0, reads->n, sq);
            HIP_CHECK(hipEventRecord(reads->evK1[1], sq));
        }
        if (bin_masks_host)
        {
            if (reads->n) HIP_CHECK(hipMemcpyAsync(bin_masks_host, reads->masks.p, (size_t)reads->n * W * 8, hipMemcpyDeviceToHost, sq));
            HIP_CHECK(hipStreamSynchronize(sq));
            if (reads->n) { float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, reads->evK1[0], reads->evK1[1])); ctx->stats.ms_ibf = ms; }
            reads->k1Pending = false;
        }
        else if (reads->n) reads->k1Pending = true;
    });
}


int dy_ibf_query_deferred(dy_ibf * f, dy_reads * reads, uint16_t threshold)
{
    return guarded([&] {
        if (!f || !reads) throw DyError(DY_EINVAL, "dy_ibf_query_deferred: NULL argument");
        if (f->ctx != reads->ctx) throw DyError(DY_EINVAL, "filter and reads belong to different contexts");
        dy_ctx * ctx = f->ctx;
        HIP_CHECK(hipSetDevice(ctx->device));
        if (ctx->pending.reads && ctx->pending.reads != reads) sweepFlush(ctx);        // one pending sweep per context
        if (reads->k1Pending) { HIP_CHECK(hipStreamWaitEvent(ctx->stream, reads->evK1[1], 0)); reads->k1Pending = false; }   // an earlier sweep on the other stream writes the same masks
        const u32 W = f->g.W;
        if (reads->recycle && !reads->masks.p) takeBuffer(ctx, reads->masks, (size_t)reads->n * W * 8 + 8);
        else reads->masks.reserve((size_t)reads->n * W * 8 + 8);
        reads->maskW = W; reads->maskBins = f->g.nBins;
        PendingSweep & P = ctx->pending;
        P.f = f; P.reads = reads; P.threshold = threshold; P.next = 0; P.slot ^= 1u;
        SweepTiming & T = ctx->sweepTiming[P.slot];
        T.nChunks = 0; T.complete = false;
        for (auto & e : T.ev) if (!e) HIP_CHECK(hipEventCreate(&e));
        (void)ibfFill(f);                                                                // the sampled fill (a synchronising estimate) is taken now, not inside a chunk
    });
}

int dy_ctx_background_chunk(dy_ctx * ctx, uint32_t max_reads)
{
    return guarded([&] {
        if (!ctx) throw DyError(DY_EINVAL, "ctx is NULL");
        HIP_CHECK(hipSetDevice(ctx->device));
        sweepChunk(ctx, max_reads);
    });
}

// suffix array of text[0, n) on the device: returns the buffer that holds it (one of idxA / idxB) — shared by dy_suffix_array and dy_fm_build
struct SaBuffers { DevBuf dText, keysA, keysB, idxA, idxB, rank, flags, tmp, ties; };
static u32 * suffixArrayOnDevice(dy_ctx * ctx, const uint32_t * text, u32 N, uint32_t sigma, SaBuffers & B)
{
        hipStream_t st = ctx->stream;
        u32 bits = 1; while ((1ull << bits) <= sigma) ++bits;              // symbols are shifted by +1
        const u32 q = 63 / bits;
        B.dText.reserve((size_t)N * 4); B.keysA.reserve((size_t)N * 8); B.keysB.reserve((size_t)N * 8); B.idxA.reserve((size_t)N * 4); B.idxB.reserve((size_t)N * 4);
        B.rank.reserve((size_t)N * 4); B.flags.reserve((size_t)N * 4); B.ties.reserve(64);
        HIP_CHECK(hipMemcpyAsync(B.dText.p, text, (size_t)N * 4, hipMemcpyHostToDevice, st));
        const dim3 grid((N + 255) / 256), block(256);
        hipLaunchKernelGGL(k_sa_first_keys, grid, block, 0, st, B.dText.as<u32>(), N, bits, q, B.keysA.as<u64>(), B.idxA.as<u32>());
        size_t sortBytes = 0, scanBytes = 0;
        HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, sortBytes, B.keysA.as<u64>(), B.keysB.as<u64>(), B.idxA.as<u32>(), B.idxB.as<u32>(), (int)N, 0, 64, st));
        HIP_CHECK(hipcub::DeviceScan::InclusiveScan(nullptr, scanBytes, B.flags.as<u32>(), B.flags.as<u32>(), MaxU32(), (int)N, st));
        B.tmp.reserve(std::max(sortBytes, scanBytes) + 64);
        ctx->hostStatus.reserve(64);
        int keyBits = (int)std::min<u32>(64, q * bits);
        // A round = sort by (rank, rank + h), regroup.  Whether a round left ties is known on the device; the host looks every round for the
        // first two (a random genome is done after one or two) and then only every other round: a round too many only re-sorts unique keys.
        u32 round = 0;
        for (u32 h = q;; h = h > 0x40000000u ? 0x80000000u : h * 2, ++round)
        {
            size_t sb = sortBytes;
            HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(B.tmp.p, sb, B.keysA.as<u64>(), B.keysB.as<u64>(), B.idxA.as<u32>(), B.idxB.as<u32>(), (int)N, 0, keyBits, st));
            HIP_CHECK(hipMemsetAsync(B.ties.p, 0, 4, st));
            hipLaunchKernelGGL(k_sa_group_flags, grid, block, 0, st, B.keysB.as<u64>(), N, B.flags.as<u32>(), B.ties.as<u32>());
            if (round < 2 || (round & 1))
            {
                HIP_CHECK(hipMemcpyAsync(ctx->hostStatus.p, B.ties.p, 4, hipMemcpyDeviceToHost, st));
                HIP_CHECK(hipStreamSynchronize(st));
                if (*ctx->hostStatus.as<u32>() == 0 || h >= N) break;       // all ranks unique (h >= n cannot happen with sentinel-terminated texts)
            }
            size_t cb = scanBytes;
            HIP_CHECK(hipcub::DeviceScan::InclusiveScan(B.tmp.p, cb, B.flags.as<u32>(), B.flags.as<u32>(), MaxU32(), (int)N, st));
            hipLaunchKernelGGL(k_sa_scatter_ranks, grid, block, 0, st, B.idxB.as<u32>(), B.flags.as<u32>(), N, B.rank.as<u32>());
            hipLaunchKernelGGL(k_sa_doubled_keys, grid, block, 0, st, B.rank.as<u32>(), N, h, B.keysA.as<u64>(), B.idxA.as<u32>());
            keyBits = 64;
        }
        HIP_CHECK(hipGetLastError());
        return B.idxB.as<u32>();
}

int dy_suffix_array(dy_ctx * ctx, const uint32_t * text, uint64_t n, uint32_t sigma, uint32_t * sa_out)
{
    return guarded([&] {
        if (!ctx || (n && (!text || !sa_out))) throw DyError(DY_EINVAL, "dy_suffix_array: NULL argument");
        if (n >= 0x7FFFFFF0ull) throw DyError(DY_EUNSUPPORTED, "dy_suffix_array: texts of 2^31 symbols or more are not supported");
        if (sigma == 0 || sigma > 0x40000000u) throw DyError(DY_EINVAL, "dy_suffix_array: alphabet size out of range");
        if (!n) return;
        HIP_CHECK(hipSetDevice(ctx->device));
        SaBuffers B;
        const u32 * sa = suffixArrayOnDevice(ctx, text, (u32)n, sigma, B);
        HIP_CHECK(hipMemcpyAsync(sa_out, sa, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

int dy_fm_build(dy_ctx * ctx, const uint32_t * text, uint64_t n, uint32_t n_contigs, const uint32_t * starts, uint8_t * drv_out, uint8_t * ind_out, uint8_t * val_out,
                uint64_t * n_val, uint32_t * drp_out, uint32_t * pst_out, uint8_t * drs_out)
{
    return guarded([&] {
        if (!ctx || !text || !starts || !drv_out || !ind_out || !val_out || !n_val || !drp_out || !pst_out || !drs_out) throw DyError(DY_EINVAL, "dy_fm_build: NULL argument");
        if (n == 0 || n >= 0x7FFFFFF0ull) throw DyError(DY_EUNSUPPORTED, "dy_fm_build: texts of 2^31 symbols or more are not supported");
        if (n_contigs == 0 || n_contigs > 65535 || starts[0] != 0 || starts[n_contigs] != n) throw DyError(DY_EINVAL, "dy_fm_build: contig starts must run from 0 to n (1..65535 contigs)");
        HIP_CHECK(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        const u32 N = (u32)n;
        SaBuffers B;
        const u32 * sa = suffixArrayOnDevice(ctx, text, N, n_contigs + 4, B);
        const u32 nBlocks = (N + 63) / 64, nEntries = (N + 31) / 32, idBytes = n_contigs <= 255 ? 1u : 2u;
        DevBuf dStarts, sym, sampled, occ, counts, before, drv, bitsA, cntA, bitsZ, cntZ, beforeA, beforeZ, ind, val, drp;
        dStarts.reserve((size_t)(n_contigs + 1) * 4 + 64); sym.reserve(N + 64); sampled.reserve(N + 64); occ.reserve(64);
        counts.reserve((size_t)(nBlocks + 1) * sizeof(Count4) + 64); before.reserve((size_t)(nBlocks + 1) * sizeof(Count4) + 64); drv.reserve((size_t)nBlocks * 32 + 64);
        for (DevBuf * d : {&bitsA, &cntA, &bitsZ, &cntZ, &beforeA, &beforeZ}) d->reserve((size_t)(nEntries + 1) * 4 + 64);
        ind.reserve((size_t)nEntries * 8 + 64); val.reserve(((size_t)N / 10 + n_contigs + 64) * (idBytes + 4)); drp.reserve((size_t)n_contigs * 4 + 64);
        HIP_CHECK(hipMemcpyAsync(dStarts.p, starts, (size_t)(n_contigs + 1) * 4, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipMemsetAsync(occ.p, 0, 32, st));
        hipLaunchKernelGGL(k_fm_rows, dim3((N + 255) / 256), dim3(256), 0, st, sa, B.dText.as<u32>(), N, n_contigs, dStarts.as<u32>(), sym.as<u8>(), sampled.as<u8>(), occ.as<u32>());
        u32 hostOcc[5];
        HIP_CHECK(hipMemcpyAsync(hostOcc, occ.p, 20, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (hostOcc[4] != n_contigs) throw DyError(DY_EINVAL, "dy_fm_build: the text does not hold one sentinel per contig");
        u32 drs = 0;
        for (u32 c = 1; c < 4; ++c) if (hostOcc[c] < hostOcc[drs]) drs = c;          // least frequent symbol stands in for the sentinels (gold: SURVEY.md A.3)
        pst_out[0] = n_contigs; pst_out[1] = pst_out[0] + hostOcc[0]; pst_out[2] = pst_out[1] + hostOcc[1]; pst_out[3] = pst_out[2] + hostOcc[2]; pst_out[4] = N;
        *drs_out = (u8)drs;
        hipLaunchKernelGGL(k_fm_block_counts, dim3((nBlocks + 1 + 255) / 256), dim3(256), 0, st, sym.as<u8>(), N, drs, nBlocks, counts.as<Count4>());
        {
            size_t tb = 0; Count4 zero; zero.c[0] = zero.c[1] = zero.c[2] = zero.c[3] = 0;
            HIP_CHECK(hipcub::DeviceScan::ExclusiveScan(nullptr, tb, counts.as<Count4>(), before.as<Count4>(), Count4Sum(), zero, (int)(nBlocks + 1), st));
            ctx->scanTmp.reserve(tb + 64);
            HIP_CHECK(hipcub::DeviceScan::ExclusiveScan(ctx->scanTmp.p, tb, counts.as<Count4>(), before.as<Count4>(), Count4Sum(), zero, (int)(nBlocks + 1), st));
        }
        hipLaunchKernelGGL(k_fm_drv, dim3((nBlocks + 255) / 256), dim3(256), 0, st, sym.as<u8>(), N, drs, nBlocks, before.as<Count4>(), drv.as<u8>());
        hipLaunchKernelGGL(k_fm_bit_words, dim3((nEntries + 1 + 255) / 256), dim3(256), 0, st, sym.as<u8>(), sampled.as<u8>(), N, nEntries, bitsA.as<u32>(), cntA.as<u32>(),
                           bitsZ.as<u32>(), cntZ.as<u32>());
        exclusiveSum(ctx, cntA.as<u32>(), beforeA.as<u32>(), (size_t)nEntries + 1);
        exclusiveSum(ctx, cntZ.as<u32>(), beforeZ.as<u32>(), (size_t)nEntries + 1);
        hipLaunchKernelGGL(k_fm_samples, dim3((nEntries + 255) / 256), dim3(256), 0, st, sa, N, n_contigs, dStarts.as<u32>(), nEntries, bitsA.as<u32>(), beforeA.as<u32>(),
                           bitsZ.as<u32>(), beforeZ.as<u32>(), ind.as<u8>(), val.as<u8>(), idBytes, drp.as<u32>());
        HIP_CHECK(hipGetLastError());
        u32 nSamples = 0;
        HIP_CHECK(hipMemcpyAsync(&nSamples, beforeA.as<u32>() + nEntries, 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        HIP_CHECK(hipMemcpyAsync(drv_out, drv.p, (size_t)nBlocks * 32, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipMemcpyAsync(ind_out, ind.p, (size_t)nEntries * 8, hipMemcpyDeviceToHost, st));
        if (nSamples) HIP_CHECK(hipMemcpyAsync(val_out, val.p, (size_t)nSamples * (idBytes + 4), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipMemcpyAsync(drp_out, drp.p, (size_t)n_contigs * 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        *n_val = nSamples;
    });
}

int dy_reads_select_all_bins(dy_ctx * ctx, dy_reads * reads, uint32_t n_bins)
{
    return guarded([&] {
        if (!ctx || !reads) throw DyError(DY_EINVAL, "dy_reads_select_all_bins: NULL argument");
        if (reads->ctx != ctx) throw DyError(DY_EINVAL, "reads belong to a different context");
        if (n_bins == 0 || n_bins > 65535) throw DyError(DY_EINVAL, "number of bins out of range");
        HIP_CHECK(hipSetDevice(ctx->device));
        const u32 W = (n_bins + 63) / 64;
        if (reads->recycle && !reads->masks.p) takeBuffer(ctx, reads->masks, (size_t)reads->n * W * 8 + 8);
        else reads->masks.reserve((size_t)reads->n * W * 8 + 8);
        reads->maskW = W; reads->maskBins = n_bins;
        if (reads->n) hipLaunchKernelGGL(k_fill_masks, dim3((u32)(((u64)reads->n * W + 255) / 256)), dim3(256), 0, ctx->stream, reads->masks.as<u64>(), reads->n, W, n_bins);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

// ---- index --------------------------------------------------------------------------------------------
static int loadIndex(dy_ctx * ctx, const char * dir, uint32_t n_bins, const uint8_t * owned, dy_index ** out)
{
    return guarded([&] {
        if (!ctx || !dir || !out) throw DyError(DY_EINVAL, "dy_index_load: NULL argument");
        if (n_bins == 0 || n_bins > 4096) throw DyError(DY_EINVAL, "number of bins out of range");
        HIP_CHECK(hipSetDevice(ctx->device));
        std::string d(dir);
        while (d.size() > 1 && d.back() == '/') d.pop_back();
        HostIndex H;
        loadHostIndex(H, d, n_bins, owned, std::min(16u, std::max(1u, std::thread::hardware_concurrency())));
        std::unique_ptr<dy_index> idx(new dy_index);
        idx->ctx = ctx;
        idx->sharded = owned != nullptr;
        idx->ownedMask.assign((n_bins + 63) / 64, 0);
        for (u32 b = 0; b < n_bins; ++b) if (!owned || owned[b]) idx->ownedMask[b / 64] |= 1ull << (b % 64);
        upload(idx->dOwnedMask, idx->ownedMask);
        idx->bins = H.bins; idx->contigNames = H.contigNames; idx->contigLengths = H.contigLengths;
        u64 totalRows = 0;
        for (auto & bd : H.bins) { bd.saFullOff = totalRows; totalRows += bd.n; idx->multiContig |= bd.nContigs > 1; }
        idx->bins = H.bins;
        upload(idx->dBins, H.bins); upload(idx->rank, H.rank); upload(idx->ind, H.ind); upload(idx->valOffset, H.valOffset);
        upload(idx->valContig, H.valContig); upload(idx->contigs, H.contigs); upload(idx->limits, H.limits); upload(idx->drp, H.drp);
        // expand the sparse SA when it fits comfortably (tuning key sparse_sa keeps the on-disk sampling, for comparison)
        size_t freeB = 0, totalB = 0;
        HIP_CHECK(hipMemGetInfo(&freeB, &totalB));
        const size_t need = totalRows * (idx->multiContig ? 6 : 4) + 4096;
        if (!ctx->tune.sparseSa && need < freeB / 2)
        {
            idx->saFullOffset.reserve(totalRows * 4 + 64);
            if (idx->multiContig) idx->saFullContig.reserve(totalRows * 2 + 64);
            IndexView v = idx->view();
            for (u32 b = 0; b < n_bins; ++b)
                if (H.bins[b].n)
                hipLaunchKernelGGL(k_expand_sa, dim3((H.bins[b].n + 255) / 256), dim3(256), 0, ctx->stream, v, b, idx->saFullOffset.as<u32>(),
                                   idx->multiContig ? idx->saFullContig.as<u16>() : nullptr);
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(hipStreamSynchronize(ctx->stream));
            idx->hasFullSa = true;
        }
        // q-mer range table: as many levels as fit in a quarter of the free HBM, at most 12 (tuning key prefix_q lowers it, 0 = off)
        {
            HIP_CHECK(hipMemGetInfo(&freeB, &totalB));
            u32 q = (u32)std::min<u64>(ctx->tune.prefixQ, 12);
            u32 nOwned = 0;
            for (u32 b = 0; b < n_bins; ++b) nOwned += H.bins[b].n ? 1 : 0;
            while (q > 0 && ((size_t)std::max(nOwned, 1u) << (2 * q)) * sizeof(uint2) > freeB / 4) --q;
            if (q >= 4)
            {
                const size_t perBin = (size_t)1 << (2 * q);
                idx->prefixTable.reserve(perBin * std::max(nOwned, 1u) * sizeof(uint2) + 64);
                DevBuf ping, pong;
                ping.reserve((perBin / 4) * sizeof(uint2) + 64); pong.reserve((perBin / 4) * sizeof(uint2) + 64);
                { u32 slot = 0; for (u32 b = 0; b < n_bins; ++b) idx->bins[b].prefixOff = perBin * (H.bins[b].n ? slot++ : 0); }
                upload(idx->dBins, idx->bins);
                IndexView v = idx->view();
                for (u32 b = 0; b < n_bins; ++b)
                {
                    if (!H.bins[b].n) continue;
                    const uint2 * parent = nullptr;
                    for (u32 level = 1; level <= q; ++level)
                    {
                        const u32 nChild = 1u << (2 * level);
                        uint2 * child = level == q ? idx->prefixTable.as<uint2>() + idx->bins[b].prefixOff : ((level & 1) ? ping.as<uint2>() : pong.as<uint2>());
                        hipLaunchKernelGGL(k_prefix_level, dim3((nChild + 255) / 256), dim3(256), 0, ctx->stream, v, b, parent, child, nChild);
                        parent = child;
                    }
                }
                HIP_CHECK(hipGetLastError());
                HIP_CHECK(hipStreamSynchronize(ctx->stream));
                idx->prefixQ = q;
            }
        }
        // inverse SA + indexed text (tuning key no_isa keeps the pure LF walk, for comparison)
        HIP_CHECK(hipMemGetInfo(&freeB, &totalB));
        if (idx->hasFullSa && idx->prefixQ && !ctx->tune.noIsa && totalRows * 4 + H.contigs.size() + 4096 < freeB / 4)
        {
            idx->isa.reserve(totalRows * 4 + 64);
            idx->ftext.reserve(H.contigs.size() + 64);
            HIP_CHECK(hipMemcpyAsync(idx->ftext.p, idx->contigs.p, H.contigs.size(), hipMemcpyDeviceToDevice, ctx->stream));
            HIP_CHECK(hipMemsetAsync(idx->isa.p, 0, totalRows * 4, ctx->stream));
            IndexView v = idx->view();
            for (u32 b = 0; b < n_bins; ++b)
                if (idx->bins[b].n)
                    hipLaunchKernelGGL(k_build_isa_ftext, dim3((idx->bins[b].n + 255) / 256), dim3(256), 0, ctx->stream, v, b, idx->isa.as<u32>(), idx->ftext.as<u8>());
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(hipStreamSynchronize(ctx->stream));
            idx->hasIsa = true;
            // unique q-mers: position instead of row
            {
                IndexView v2 = idx->view();
                const u32 nEntries = 1u << (2 * idx->prefixQ);
                for (u32 b = 0; b < n_bins; ++b)
                    if (idx->bins[b].n)
                        hipLaunchKernelGGL(k_prefix_direct, dim3((nEntries + 255) / 256), dim3(256), 0, ctx->stream, v2, b,
                                           idx->prefixTable.as<uint2>() + idx->bins[b].prefixOff, nEntries);
                HIP_CHECK(hipGetLastError());
                HIP_CHECK(hipStreamSynchronize(ctx->stream));
            }
            // 16-byte entries with the bases after each unique q-mer, when twice the table still leaves three quarters of the free HBM
            HIP_CHECK(hipMemGetInfo(&freeB, &totalB));
            u32 nOwned = 0;
            for (u32 b = 0; b < n_bins; ++b) nOwned += idx->bins[b].n ? 1 : 0;
            const size_t wideBytes = ((size_t)std::max(nOwned, 1u) << (2 * idx->prefixQ)) * sizeof(uint4);
            if (ctx->tune.prefixWide && wideBytes < freeB / 4)
            {
                idx->prefixWide.reserveExact(wideBytes + 64);
                IndexView v3 = idx->view();
                const u32 nEntries = 1u << (2 * idx->prefixQ);
                for (u32 b = 0; b < n_bins; ++b)
                    if (idx->bins[b].n)
                        hipLaunchKernelGGL(k_prefix_wide, dim3((nEntries + 255) / 256), dim3(256), 0, ctx->stream, v3, b,
                                           idx->prefixTable.as<uint2>() + idx->bins[b].prefixOff, idx->prefixWide.as<uint4>() + idx->bins[b].prefixOff, nEntries);
                HIP_CHECK(hipGetLastError());
                HIP_CHECK(hipStreamSynchronize(ctx->stream));
                idx->hasWide = true;
                DevBuf drop; idx->prefixTable.moveTo(drop);           // the narrow table is not read again
            }
        }
        *out = idx.release();
    });
}

int dy_index_load(dy_ctx * ctx, const char * dir, uint32_t n_bins, dy_index ** out) { return loadIndex(ctx, dir, n_bins, nullptr, out); }
int dy_index_load_sharded(dy_ctx * ctx, const char * dir, uint32_t n_bins, const uint8_t * owned, dy_index ** out)
{
    if (!owned) { g_lastError = "dy_index_load_sharded: owned is NULL"; return DY_EINVAL; }
    return loadIndex(ctx, dir, n_bins, owned, out);
}

uint32_t dy_index_bins(const dy_index * idx) { return idx ? (uint32_t)idx->bins.size() : 0; }
uint32_t dy_index_contigs(const dy_index * idx) { return idx ? (uint32_t)idx->contigNames.size() : 0; }
const char * dy_index_contig_name(const dy_index * idx, uint32_t c) { return (idx && c < idx->contigNames.size()) ? idx->contigNames[c].c_str() : nullptr; }
uint64_t dy_index_contig_length(const dy_index * idx, uint32_t c) { return (idx && c < idx->contigLengths.size()) ? idx->contigLengths[c] : 0; }
int dy_index_get_info(const dy_index * idx, dy_index_info * info)
{
    if (!idx || !info) { g_lastError = "NULL argument"; return DY_EINVAL; }
    memset(info, 0, sizeof(*info));
    info->n_bins = (uint32_t)idx->bins.size(); info->n_contigs = (uint32_t)idx->contigNames.size(); info->prefix_q = idx->prefixQ;
    u64 rows = 0; u32 resident = 0;
    for (auto & b : idx->bins) { rows += b.n; resident += b.n ? 1 : 0; }
    info->n_bins_resident = resident; info->total_rows = rows;
    info->bytes_rank = idx->rank.cap; info->bytes_sparse_sa = idx->ind.cap + idx->valOffset.cap + idx->valContig.cap; info->bytes_contigs = idx->contigs.cap;
    info->bytes_expanded_sa = idx->hasFullSa ? idx->saFullOffset.cap + idx->saFullContig.cap : 0;
    info->bytes_prefix_table = idx->prefixQ ? (idx->hasWide ? idx->prefixWide.cap : idx->prefixTable.cap) : 0;
    info->bytes_isa_and_text = idx->hasIsa ? idx->isa.cap + idx->ftext.cap : 0;
    return DY_OK;
}
void dy_index_free(dy_index * idx) { if (idx) { (void)hipSetDevice(idx->ctx->device); delete idx; } }

// ---- mapping ------------------------------------------------------------------------------------------
void dy_params_default(dy_params * p)
{
    if (!p) return;
    p->error_rate = 0.05f; p->strata_rate = 0.0f; p->sensitivity = 1; p->hits_threshold = 300; p->use_filter = 1;
    p->paired = 0; p->library_length = 0; p->library_dev = 0; p->verify_matches = 1; p->indel_rate = 0.25f;
}

int dy_map_reads_async(dy_ctx * ctx, const dy_index * idx, dy_reads * reads, const dy_params * p, dy_result ** out)
{
    return guarded([&] {
        if (!ctx || !idx || !reads || !p || !out) throw DyError(DY_EINVAL, "dy_map_reads: NULL argument");
        if (idx->ctx != ctx || reads->ctx != ctx) throw DyError(DY_EINVAL, "index/reads belong to a different context");
        if (p->sensitivity > 2) throw DyError(DY_EINVAL, "sensitivity must be 0 (low), 1 (high) or 2 (full)");
        if (idx->sharded) throw DyError(DY_EINVAL, "this index holds a shard of the bins: use dy_map_bins / dy_finalize_raw");
        if (!(p->error_rate >= 0.0f && p->error_rate <= 0.10f)) throw DyError(DY_EINVAL, "error rate must be within [0, 0.10]");
        HIP_CHECK(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        const u32 nReads = reads->n, nBins = (u32)idx->bins.size(), W = (nBins + 63) / 64;
        const u32 maxErr = readErrorsOf(reads->maxLen, p->error_rate);
        const u32 S0 = seedCountOf(maxErr, 0);
        if (S0 > MAX_SEEDS) throw DyError(DY_EUNSUPPORTED, "more than 32 seeds per read (read length x error rate too large)");
        if (nBins > 65535) throw DyError(DY_EUNSUPPORTED, "too many bins");

        if (!p->use_filter)
        {
            if (reads->recycle && !reads->masks.p) takeBuffer(ctx, reads->masks, (size_t)nReads * W * 8 + 8);
            else reads->masks.reserve((size_t)nReads * W * 8 + 8);
            reads->maskW = W; reads->maskBins = nBins;
            if (nReads) hipLaunchKernelGGL(k_fill_masks, dim3((u32)(((u64)nReads * W + 255) / 256)), dim3(256), 0, st, reads->masks.as<u64>(), nReads, W, nBins);
        }
        else if (reads->maskW != W || reads->maskBins != nBins)
            throw DyError(DY_EINVAL, "dy_ibf_query must run on these reads first (with a filter of the index's bin count)");
        const bool paired = p->paired != 0;
        if (paired && (nReads & 1)) throw DyError(DY_EINVAL, "a paired batch holds an even number of reads (first mates, then second mates)");
        waitForMasks(ctx, reads);
        if (paired && nReads)
            hipLaunchKernelGGL(k_pe_or_masks, dim3((u32)(((u64)(nReads / 2) * W + 255) / 256)), dim3(256), 0, st, reads->masks.as<u64>(), nReads / 2, W);

        std::unique_ptr<dy_result> res(new dy_result);
        const u32 maxMatchErr = maxMatchErrorsOf(reads, p, maxErr);
        res->ctx = ctx; res->nReads = nReads; res->cigarStride = 2 * maxMatchErr + 1;
        takeBuffer(ctx, res->out, (size_t)nReads * sizeof(ReadOut) + 64);
        takeBuffer(ctx, res->cigars, (size_t)nReads * res->cigarStride * 4 + 64);
        ctx->stats.items = ctx->stats.raw_matches = ctx->stats.hard_items = ctx->stats.pool_retries = 0;
        if (nReads == 0) { *out = res.release(); return; }

        // Tuning key async_map: single-end batches are queued from the work list to the last aggregation kernel without the host waiting for a
        // size (the waiting way has two round trips — work-list size, pool counter — that leave the device idle for 85 us per step): sizes are
        // bounded from the previous batch, the kernels read the real ones on the device, and the ONE fetch at the end brings the flags that say
        // whether the bounds held.  If not (or on a context's first batch, or for paired batches) the batch is mapped the waiting way.
        // OFF by default: the idle time goes (0.195 -> 0.160 ms) but every small kernel of the chain then takes 2-3x as long when it is
        // dispatched behind its predecessor instead of into an idle queue, and the padded sort / larger grids cost the rest: 5.24 vs 5.12 ms.
        const bool canAsync = !paired && ctx->tune.itemsSorted && ctx->tune.asyncMap && ctx->itemsPerRead > 0.0 && ctx->poolCap > 0;
        res->readLens = reads->lens;       // for the host-side MAPQ (shared, not copied)
        for (int attempt = 0;; ++attempt)
        {
            const bool async = canAsync && attempt == 0;
            u32 nItems = 0, nPool = 0;
            runSearchExtend(ctx, idx, reads, p, S0, nItems, nPool, async);
            res->nPool = nPool;
            runFinalize(ctx, idx, reads, p, res.get(), ctx->pool.as<RawMatch>(), nPool, nItems, maxMatchErr, PE_OWNER | PE_HOME, async ? ctx->counters.as<u32>() + COUNTER_POOL64 : nullptr);
            HIP_CHECK(hipEventRecord(ctx->ev[6], st));
            ctx->hostStatus.reserve(COUNTER_BYTES + 64);
            HIP_CHECK(hipMemcpyAsync(ctx->hostStatus.p, ctx->counters.p, COUNTER_BYTES, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipMemcpyAsync(ctx->hostStatus.as<u8>() + COUNTER_BYTES, ctx->devItems.p, 16, hipMemcpyDeviceToHost, st));
            fetchSync(ctx);
            const u32 * c = ctx->hostStatus.as<u32>(), * items = c + COUNTER_BYTES / 4;
            if (async)
            {
                u64 poolTotal; memcpy(&poolTotal, c + COUNTER_POOL64, 8);
                if (c[1] || items[2])
                {   // a bound did not hold: once more, waiting for the sizes (the pool keeps what this pass learnt)
                    if (c[1])
                    {
                        if (poolTotal >= 0xFFFFFFF0ull) throw DyError(DY_EUNSUPPORTED, "more than 2^32 raw matches in one batch: use a smaller reads batch");
                        ctx->poolCap = (size_t)poolTotal + (size_t)poolTotal / 4 + 1024;
                    }
                    ctx->stats.pool_retries++;
                    continue;
                }
                ctx->stats.items = items[0]; ctx->stats.raw_matches = poolTotal; ctx->stats.hard_items = c[2] - c[12];
                ctx->itemsPerRead = std::max(0.25, (double)items[0] / (double)nReads);
            }
            if (c[COUNTER_BAD_GROUP]) throw DyError(DY_EHIP, "internal error: a raw match record is inconsistent with the per-read counts");
            break;
        }
        collectDeferredSweepTime(ctx);
        float ms;
        HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3])); ctx->stats.ms_items = ms;
        HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev[3], ctx->ev[4])); ctx->stats.ms_search = ms;
        HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5])); ctx->stats.ms_extend = ms;
        HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev[5], ctx->ev[6])); ctx->stats.ms_finalize = ms;
        *out = res.release();
    });
}

int dy_result_fetch(dy_result * r)
{
    return guarded([&] {
        if (!r) throw DyError(DY_EINVAL, "result is NULL");
        if (r->fetched) return;
        dy_ctx * ctx = r->ctx;
        HIP_CHECK(hipSetDevice(ctx->device));
        const u32 n = r->nReads;
        hipStream_t st = ctx->stream;
        // dense offsets of every read's matchesSet and CIGAR, then one packing pass on the device
        ctx->packCntSet.reserve((size_t)(n + 1) * 4); ctx->packCntOps.reserve((size_t)(n + 1) * 4);
        ctx->packOffSet.reserve((size_t)(n + 1) * 4); ctx->packOffOps.reserve((size_t)(n + 1) * 4);
        hipLaunchKernelGGL(k_result_counts, dim3(n / 256 + 1), dim3(256), 0, st, r->out.as<ReadOut>(), n, ctx->packCntSet.as<u32>(), ctx->packCntOps.as<u32>());
        exclusiveSum(ctx, ctx->packCntSet.as<u32>(), ctx->packOffSet.as<u32>(), (size_t)n + 1);
        exclusiveSum(ctx, ctx->packCntOps.as<u32>(), ctx->packOffOps.as<u32>(), (size_t)n + 1);
        u32 totals[2] = {0, 0};
        HIP_CHECK(hipMemcpyAsync(&totals[0], ctx->packOffSet.as<u32>() + n, 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipMemcpyAsync(&totals[1], ctx->packOffOps.as<u32>() + n, 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        r->nMatches = totals[0]; r->nOps = totals[1];
        ctx->packReads.reserve((size_t)n * sizeof(ReadOut) + 64); ctx->packSet.reserve(r->nMatches * sizeof(OutMatch) + 64); ctx->packOps.reserve(r->nOps * 4 + 64);
        if (n)
            hipLaunchKernelGGL(k_result_pack, dim3((n + 255) / 256), dim3(256), 0, st, r->out.as<ReadOut>(), r->outSet.as<OutMatch>(), r->cigars.as<u32>(), n,
                               ctx->packOffSet.as<u32>(), ctx->packOffOps.as<u32>(), ctx->packReads.as<ReadOut>(), ctx->packSet.as<OutMatch>(), ctx->packOps.as<u32>());
        takePinned(ctx, r->reads, (size_t)n * sizeof(dy_read_result) + 64);
        takePinned(ctx, r->matches, r->nMatches * sizeof(dy_match) + 64);
        takePinned(ctx, r->cigarOps, r->nOps * 4 + 64);
        if (n) HIP_CHECK(hipMemcpyAsync(r->reads.p, ctx->packReads.p, (size_t)n * sizeof(ReadOut), hipMemcpyDeviceToHost, st));
        if (r->nMatches) HIP_CHECK(hipMemcpyAsync(r->matches.p, ctx->packSet.p, r->nMatches * sizeof(OutMatch), hipMemcpyDeviceToHost, st));
        if (r->nOps) HIP_CHECK(hipMemcpyAsync(r->cigarOps.p, ctx->packOps.p, r->nOps * 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        // MAPQ (fp64, host libm: reference src/mapper_writer.h:221-225) is a pure function of (errors, read length, best, sub): a small
        // memo covers nearly every read (3 pow + 1 log10 each otherwise); read ranges go to a few threads
        dy_read_result * reads = r->reads.as<dy_read_result>();
        const std::vector<u32> & lens = *r->readLens;
        auto fillMapq = [&](u32 lo, u32 hi) {
            enum { MEMO_E = 16, MEMO_C = 8 };
            u32 memoLen = 0; int memo[MEMO_E][MEMO_C][MEMO_C];
            for (auto & a : memo) for (auto & b : a) for (auto & c : b) c = -1;
            for (u32 i = lo; i < hi; ++i)
            {
                dy_read_result & d = reads[i];
                if (!d.mapped) d.mapq = 0;
                else if (d.paired) d.mapq = (unsigned)std::round(-10.0 * std::log10(1.0 - std::min(d.pair_prob, 0.9999999)));
                else
                {
                    const u32 len = lens[i];
                    if (len != memoLen) { memoLen = len; for (auto & a : memo) for (auto & b : a) for (auto & c : b) c = -1; }
                    if (d.primary.errors < MEMO_E && d.best_count < MEMO_C && d.sub_count < MEMO_C)
                    {
                        int & m = memo[d.primary.errors][d.best_count][d.sub_count];
                        if (m < 0) m = (int)mapqOf(d.primary.errors, len, d.best_count, d.sub_count);
                        d.mapq = (unsigned)m;
                    }
                    else d.mapq = mapqOf(d.primary.errors, len, d.best_count, d.sub_count);
                }
            }
        };
        const u32 T = std::min<u32>(4, n / 65536 + 1);
        if (T <= 1) fillMapq(0, n);
        else
        {
            std::vector<std::thread> pool;
            for (u32 t = 1; t < T; ++t) pool.emplace_back(fillMapq, (u32)((u64)n * t / T), (u32)((u64)n * (t + 1) / T));
            fillMapq(0, (u32)((u64)n / T));
            for (auto & th : pool) th.join();
        }
        r->fetched = true;
    });
}

int dy_map_reads(dy_ctx * ctx, const dy_index * idx, dy_reads * reads, const dy_params * p, dy_result ** out)
{
    int rc = dy_map_reads_async(ctx, idx, reads, p, out);
    if (rc != DY_OK) return rc;
    rc = dy_result_fetch(*out);
    if (rc != DY_OK) { dy_result_free(*out); *out = nullptr; }
    return rc;
}

const dy_read_result * dy_result_reads(const dy_result * r) { return (r && r->fetched) ? r->reads.as<dy_read_result>() : nullptr; }
const dy_match * dy_result_matches(const dy_result * r) { return (r && r->fetched) ? r->matches.as<dy_match>() : nullptr; }
uint64_t dy_result_n_matches(const dy_result * r) { return (r && r->fetched) ? r->nMatches : 0; }
const uint32_t * dy_result_cigars(const dy_result * r) { return (r && r->fetched) ? r->cigarOps.as<u32>() : nullptr; }
void dy_result_free(dy_result * r)
{
    if (!r) return;
    (void)hipSetDevice(r->ctx->device);
    giveBuffer(r->ctx, r->out); giveBuffer(r->ctx, r->outSet); giveBuffer(r->ctx, r->cigars);
    givePinned(r->ctx, r->reads); givePinned(r->ctx, r->matches); givePinned(r->ctx, r->cigarOps);
    delete r;
}

int dy_map_get_stats(const dy_ctx * ctx, dy_map_stats * s)
{
    if (!ctx || !s) { g_lastError = "NULL argument"; return DY_EINVAL; }
    *s = ctx->stats;
    return DY_OK;
}

// ---- text in, text out (sam_kernels.h) ----------------------------------------------------------------------
int dy_host_buffer(dy_ctx * ctx, uint32_t slot, uint64_t bytes, void ** out)
{
    return guarded([&] {
        if (!ctx || !out) throw DyError(DY_EINVAL, "dy_host_buffer: NULL argument");
        if (slot >= 8) throw DyError(DY_EINVAL, "dy_host_buffer: slots 0..7");
        HIP_CHECK(hipSetDevice(ctx->device));
        ctx->hostSlots[slot].reserve((size_t)bytes + 64);
        *out = ctx->hostSlots[slot].p;
    });
}

uint64_t dy_reads_total_bases(const dy_reads * r) { return r ? r->totalBases : 0; }

int dy_reads_from_fastq(dy_ctx * ctx, const void * text, uint64_t n_bytes, uint32_t max_reads, uint32_t at_eof, dy_reads ** out, uint64_t * consumed)
{
    return guarded([&] {
        if (!ctx || !out || !consumed || (n_bytes && !text)) throw DyError(DY_EINVAL, "dy_reads_from_fastq: NULL argument");
        if (max_reads == 0 || max_reads > 2097151u) throw DyError(DY_EINVAL, "a batch holds at most 2097151 reads (21-bit read id)");
        if (n_bytes >= 0xFFFFFF00ull) throw DyError(DY_EINVAL, "dy_reads_from_fastq: windows of 4 GiB or more are not supported");
        HIP_CHECK(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        *consumed = 0;
        std::unique_ptr<dy_reads> r(new dy_reads);
        r->ctx = ctx; r->recycle = true; r->fromFastq = true; r->fqBytes = n_bytes;
        takeBuffer(ctx, r->fqText, (size_t)n_bytes + 64);
        if (n_bytes) HIP_CHECK(hipMemcpyAsync(r->fqText.p, text, (size_t)n_bytes, hipMemcpyHostToDevice, st));
        // line ends of the window: count per 4 KB block, scan, write (only the first 4 x max_reads are kept)
        const u32 nBlocks = (u32)((n_bytes + FQ_BLOCK_BYTES - 1) / FQ_BLOCK_BYTES);
        const u32 cap = 4 * max_reads + 4;
        ctx->fqBlockCount.reserve((size_t)(nBlocks + 1) * 4 + 64); ctx->fqBlockOff.reserve((size_t)(nBlocks + 1) * 4 + 64);
        ctx->fqLineEnds.reserve((size_t)cap * 4 + 64); ctx->fqStatus.reserve(sizeof(FqStatus) + 64);
        ctx->fqLensWide.reserve((size_t)(max_reads + 1) * 8 + 64); ctx->fqLens16.reserve((size_t)(max_reads + 1) * 2 + 64);
        takeBuffer(ctx, r->fqRecs, (size_t)max_reads * sizeof(FqRec) + 64);
        HIP_CHECK(hipMemsetAsync(ctx->fqBlockCount.as<u32>() + nBlocks, 0, 4, st));
        if (nBlocks) hipLaunchKernelGGL(k_fq_count_lines, dim3(nBlocks), dim3(256), 0, st, r->fqText.as<u8>(), (u64)n_bytes, ctx->fqBlockCount.as<u32>());
        exclusiveSum(ctx, ctx->fqBlockCount.as<u32>(), ctx->fqBlockOff.as<u32>(), (size_t)nBlocks + 1);
        if (nBlocks) hipLaunchKernelGGL(k_fq_write_lines, dim3(nBlocks), dim3(256), 0, st, r->fqText.as<u8>(), (u64)n_bytes, ctx->fqBlockOff.as<u32>(), cap, ctx->fqLineEnds.as<u32>());
        hipLaunchKernelGGL(k_fq_status, dim3(1), dim3(1), 0, st, ctx->fqBlockOff.as<u32>(), nBlocks, r->fqText.as<u8>(), (u64)n_bytes, at_eof ? 1u : 0u, max_reads, cap,
                           ctx->fqLineEnds.as<u32>(), ctx->fqStatus.as<FqStatus>());
        hipLaunchKernelGGL(k_fq_records, dim3((max_reads + 1 + 255) / 256), dim3(256), 0, st, r->fqText.as<u8>(), ctx->fqLineEnds.as<u32>(), ctx->fqStatus.as<FqStatus>(),
                           r->fqRecs.as<FqRec>(), ctx->fqLensWide.as<u64>(), ctx->fqLens16.as<u16>());
        HIP_CHECK(hipGetLastError());
        // one fetch: the status and the read lengths (the host keeps them for MAPQ)
        ctx->hostStatus.reserve(sizeof(FqStatus) + (size_t)max_reads * 2 + 64);
        HIP_CHECK(hipMemcpyAsync(ctx->hostStatus.p, ctx->fqStatus.p, sizeof(FqStatus), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipMemcpyAsync(ctx->hostStatus.as<u8>() + sizeof(FqStatus), ctx->fqLens16.p, (size_t)max_reads * 2, hipMemcpyDeviceToHost, st));
        fetchSync(ctx);
        FqStatus fs; memcpy(&fs, ctx->hostStatus.p, sizeof(fs));
        if (fs.bad)
        {
            std::string why;
            if (fs.bad & 1) why += " a record does not start with '@';";
            if (fs.bad & 2) why += " a separator line does not start with '+';";
            if (fs.bad & 4) why += " an empty or over-long sequence line (or blank / wrapped lines);";
            if (fs.bad & 8) why += " a quality line of another length than its sequence;";
            if (fs.bad & 16) why += " carriage returns or blanks at line ends;";
            if (fs.bad & 32) why += " an empty read name;";
            dy_reads_free(r.release());
            throw DyError(DY_EUNSUPPORTED, "dy_reads_from_fastq: not plain four-line FASTQ (" + why + " ) — parse this input on the host");
        }
        const u32 n = fs.nRecords;
        r->n = n; r->totalBases = fs.totalBases; r->maxLen = n ? fs.maxLen : 0; r->minLen = n ? fs.minLen : 0;
        r->lens->resize(n);
        { const u16 * l16 = (const u16 *)(ctx->hostStatus.as<u8>() + sizeof(FqStatus)); u32 * lens = r->lens->data(); for (u32 i = 0; i < n; ++i) lens[i] = l16[i]; }
        takeBuffer(ctx, r->seqs, r->totalBases + 64); takeBuffer(ctx, r->seqsRc, r->totalBases + 64);
        takeBuffer(ctx, r->offs, (size_t)(n + 1) * 8);
        exclusiveSum(ctx, ctx->fqLensWide.as<u64>(), r->offs.as<u64>(), (size_t)n + 1);
        if (n) hipLaunchKernelGGL(k_fq_codes, dim3(std::min<u32>((n + 15) / 16, 65536u)), dim3(256), 0, st, r->fqText.as<u8>(), r->fqRecs.as<FqRec>(), n, r->offs.as<u64>(),
                                  r->seqs.as<u8>(), r->seqsRc.as<u8>());
        HIP_CHECK(hipGetLastError());
        *consumed = fs.consumed;
        *out = r.release();
    });
}

int dy_result_sam(dy_result * res, const dy_reads * reads, const dy_index * idx, const char * read_group, uint32_t out_slot, const char ** text, uint64_t * n_bytes,
                  uint64_t * n_mapped)
{
    return guarded([&] {
        if (!res || !reads || !idx || !read_group || !text || !n_bytes) throw DyError(DY_EINVAL, "dy_result_sam: NULL argument");
        if (out_slot >= 8) throw DyError(DY_EINVAL, "dy_result_sam: slots 0..7");
        dy_ctx * ctx = res->ctx;
        if (reads->ctx != ctx || idx->ctx != ctx) throw DyError(DY_EINVAL, "result, reads and index belong to different contexts");
        if (!reads->fromFastq) throw DyError(DY_EINVAL, "dy_result_sam needs a batch of dy_reads_from_fastq (it copies names and qualities from the FASTQ text)");
        if (res->nReads != reads->n) throw DyError(DY_EINVAL, "dy_result_sam: the result does not belong to these reads");
        HIP_CHECK(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        const u32 n = reads->n;
        *text = nullptr; *n_bytes = 0; if (n_mapped) *n_mapped = 0;
        if (!n) return;
        if (!idx->namesOnDevice)
        {   // contig names of all bins, global ids (reference src/d_mapper.h:727-760 collects them for the header the same way)
            std::vector<u8> chars; std::vector<u32> offs(idx->contigNames.size() + 1, 0);
            for (size_t c = 0; c < idx->contigNames.size(); ++c) { chars.insert(chars.end(), idx->contigNames[c].begin(), idx->contigNames[c].end()); offs[c + 1] = (u32)chars.size(); }
            upload(idx->dNames, chars); upload(idx->dNameOffs, offs);
            idx->namesOnDevice = true;
        }
        const size_t rgLen = strlen(read_group);
        ctx->samGroup.reserve(rgLen + 64);
        if (rgLen) HIP_CHECK(hipMemcpyAsync(ctx->samGroup.p, read_group, rgLen, hipMemcpyHostToDevice, st));
        ctx->samSizes.reserve((size_t)(n + 1) * 8 + 64); ctx->samOff.reserve((size_t)(n + 1) * 8 + 64); ctx->samMapq.reserve((size_t)n * 4 + 64); ctx->samFlags.reserve(64);
        HIP_CHECK(hipMemsetAsync(ctx->samFlags.p, 0, 16, st));
        SamArgs A;
        A.text = reads->fqText.as<u8>(); A.recs = reads->fqRecs.as<FqRec>(); A.offs = reads->offs.as<u64>();
        A.out = res->out.as<ReadOut>(); A.outSet = res->outSet.as<OutMatch>(); A.cigars = res->cigars.as<u32>();
        A.contigNames = idx->dNames.as<u8>(); A.contigNameOffs = idx->dNameOffs.as<u32>();
        A.readGroup = ctx->samGroup.as<u8>(); A.readGroupLen = (u32)rgLen; A.nReads = n;
        hipLaunchKernelGGL(k_sam_sizes, dim3((n + 1 + 255) / 256), dim3(256), 0, st, A, ctx->samMapq.as<u32>(), ctx->samSizes.as<u64>(), ctx->samFlags.as<u32>(), ctx->samFlags.as<u32>() + 1);
        exclusiveSum(ctx, ctx->samSizes.as<u64>(), ctx->samOff.as<u64>(), (size_t)n + 1);
        ctx->hostStatus.reserve(64);
        HIP_CHECK(hipMemcpyAsync(ctx->hostStatus.p, ctx->samOff.as<u64>() + n, 8, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipMemcpyAsync(ctx->hostStatus.as<u8>() + 8, ctx->samFlags.p, 8, hipMemcpyDeviceToHost, st));
        fetchSync(ctx);
        u64 total; u32 flags[2];
        memcpy(&total, ctx->hostStatus.p, 8); memcpy(flags, ctx->hostStatus.as<u8>() + 8, 8);
        if (flags[0]) throw DyError(DY_EUNSUPPORTED, "dy_result_sam: a mapping quality lies on a rounding boundary of the device's arithmetic — format this batch on the host (dy_result_fetch)");
        ctx->samText.reserve((size_t)total + 64);
        hipLaunchKernelGGL(k_sam_write, dim3(std::min<u32>((n + 15) / 16, 65536u)), dim3(256), 0, st, A, ctx->samMapq.as<u32>(), ctx->samOff.as<u64>(), ctx->samText.as<u8>());
        HIP_CHECK(hipGetLastError());
        ctx->hostSlots[out_slot].reserve((size_t)total + 64);
        HIP_CHECK(hipMemcpyAsync(ctx->hostSlots[out_slot].p, ctx->samText.p, (size_t)total, hipMemcpyDeviceToHost, st));
        fetchSync(ctx);
        *text = ctx->hostSlots[out_slot].as<char>(); *n_bytes = total; if (n_mapped) *n_mapped = flags[1];
    });
}

// ---- bins sharded over ranks --------------------------------------------------------------------------
// Per-context state of the packed exchanges (shard_kernels.h): nothing below synchronises with the host — the sizes stay on the device
// in the header rows until the CALLER fetches them (one stream synchronisation per exchange).
static SegTable segTableC1(const dy_seg_header * h, u32 nRanks, u32 W, u64 & nReads, u64 & nBases, u32 & maxLen, u32 & minLen)
{
    SegTable T; memset(&T, 0, sizeof(T));
    T.nRanks = nRanks;
    u64 off = 0; nReads = 0; nBases = 0; maxLen = 0; minLen = 0xFFFFFFFFu;
    for (u32 s = 0; s < nRanks; ++s)
    {
        // bases travel at 4 bits each, every read from a byte of its own: (bases + 1) / 2 .. (bases + reads) / 2 bytes
        if (h[s].bytes < c1SegBytes(h[s].n_reads, (h[s].n_bases + 1) / 2, W) || h[s].bytes > c1SegBytes(h[s].n_reads, (h[s].n_bases + h[s].n_reads) / 2, W))
            throw DyError(DY_EINVAL, "dy_shard_c1_unpack: a segment header does not describe its own size");
        T.off[s] = off; T.start[s] = (u32)nReads;
        off += h[s].bytes; nReads += h[s].n_reads; nBases += h[s].n_bases;
        if (nReads > 2097151u) throw DyError(DY_EINVAL, "a batch holds at most 2097151 reads (21-bit read id)");
        if (h[s].n_reads) { maxLen = std::max(maxLen, (u32)(h[s].aux & 0xFFFFFFFFu)); minLen = std::min(minLen, (u32)(h[s].aux >> 32)); }
    }
    T.start[nRanks] = (u32)nReads;
    if (!nReads) minLen = 0;
    return T;
}

int dy_shard_c1_pack(dy_ctx * ctx, dy_reads * reads, uint32_t n_ranks, const uint64_t * owner_masks, uint32_t paired, void * headers_dev, void ** send_dev)
{
    return guarded([&] {
        if (!ctx || !reads || !owner_masks || !headers_dev || !send_dev) throw DyError(DY_EINVAL, "dy_shard_c1_pack: NULL argument");
        if (reads->ctx != ctx) throw DyError(DY_EINVAL, "reads belong to a different context");
        if (n_ranks == 0 || n_ranks > MAX_RANKS) throw DyError(DY_EINVAL, "1..64 ranks");
        if (!reads->maskW) throw DyError(DY_EINVAL, "dy_ibf_query must run on these reads first");
        if (paired && (reads->n & 1)) throw DyError(DY_EINVAL, "a paired batch holds an even number of reads (first mates, then second mates)");
        HIP_CHECK(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        const u32 n = reads->n, W = reads->maskW, nWaves = (n + 63) / 64;
        waitForMasks(ctx, reads);
        // the four sequences of a pair vote together (reference src/d_mapper.h:350-358): both mates go wherever either is selected
        if (paired && n) hipLaunchKernelGGL(k_pe_or_masks, dim3((u32)(((u64)(n / 2) * W + 255) / 256)), dim3(256), 0, st, reads->masks.as<u64>(), n / 2, W);
        // ownership is static per job: upload it when it changed
        const size_t ownerWords = (size_t)n_ranks * W;
        if (ctx->shOwnerHost.size() != ownerWords || memcmp(ctx->shOwnerHost.data(), owner_masks, ownerWords * 8))
        {
            HIP_CHECK(hipStreamSynchronize(st));                    // an earlier pack may still read the previous table
            ctx->shOwnerHost.assign(owner_masks, owner_masks + ownerWords);
            ctx->shOwner.reserve(ownerWords * 8 + 64);
            HIP_CHECK(hipMemcpy(ctx->shOwner.p, ctx->shOwnerHost.data(), ownerWords * 8, hipMemcpyHostToDevice));
        }
        const size_t nPairs = (size_t)n_ranks * nWaves + 1;
        ctx->shPairs.reserve(nPairs * sizeof(CountPair) + 64); ctx->shScan.reserve(nPairs * sizeof(CountPair) + 64);
        ctx->shSegOff.reserve((MAX_RANKS + 1) * 8);
        hipLaunchKernelGGL(k_c1_count, dim3(std::max(1u, (nWaves + 3) / 4)), dim3(256), 0, st, reads->masks.as<u64>(), reads->offs.as<u64>(), n, W, ctx->shOwner.as<u64>(),
                           n_ranks, nWaves, ctx->shPairs.as<CountPair>());
        {
            size_t tb = 0; CountPair zero; zero.bases = 0; zero.cnt = 0; zero.packed = 0;
            HIP_CHECK(hipcub::DeviceScan::ExclusiveScan(nullptr, tb, ctx->shPairs.as<CountPair>(), ctx->shScan.as<CountPair>(), CountPairSum(), zero, (int)nPairs, st));
            ctx->scanTmp.reserve(tb + 64);
            HIP_CHECK(hipcub::DeviceScan::ExclusiveScan(ctx->scanTmp.p, tb, ctx->shPairs.as<CountPair>(), ctx->shScan.as<CountPair>(), CountPairSum(), zero, (int)nPairs, st));
        }
        const u64 lenRange = (u64)reads->maxLen | ((u64)reads->minLen << 32);
        hipLaunchKernelGGL(k_c1_headers, dim3(1), dim3(MAX_RANKS), 0, st, ctx->shScan.as<CountPair>(), n_ranks, nWaves, W, lenRange, (SegHeader *)headers_dev, ctx->shSegOff.as<u64>());
        // room for the worst case (every read selected by every rank): the sizes are not known on the host, and 288 GB make the bound cheap
        ctx->shSend.reserve((size_t)n_ranks * (c1SegBytes(n, (reads->totalBases + n) / 2 + 1, W) + 16) + 64);
        if (nWaves)
            hipLaunchKernelGGL(k_c1_pack, dim3((nWaves + 3) / 4), dim3(256), 0, st, reads->masks.as<u64>(), reads->offs.as<u64>(), reads->seqs.as<u8>(), n, W, ctx->shOwner.as<u64>(),
                               n_ranks, nWaves, ctx->shScan.as<CountPair>(), (const SegHeader *)headers_dev, ctx->shSegOff.as<u64>(), ctx->shSend.as<u8>());
        HIP_CHECK(hipGetLastError());
        *send_dev = ctx->shSend.p;
    });
}

int dy_shard_c1_unpack(dy_ctx * ctx, const void * recv_dev, uint32_t n_ranks, const dy_seg_header * recv_headers, uint32_t n_bins, uint32_t paired, dy_reads ** out)
{
    return guarded([&] {
        if (!ctx || !recv_headers || !out) throw DyError(DY_EINVAL, "dy_shard_c1_unpack: NULL argument");
        if (n_ranks == 0 || n_ranks > MAX_RANKS) throw DyError(DY_EINVAL, "1..64 ranks");
        if (n_bins == 0 || n_bins > 65535) throw DyError(DY_EINVAL, "number of bins out of range");
        const u32 W = (n_bins + 63) / 64;
        u64 nReads64 = 0, nBases = 0; u32 maxLen = 0, minLen = 0;
        const SegTable T = segTableC1(recv_headers, n_ranks, W, nReads64, nBases, maxLen, minLen);
        const u32 n = (u32)nReads64;
        if (n && !recv_dev) throw DyError(DY_EINVAL, "dy_shard_c1_unpack: NULL buffer");
        if (maxLen > 8191) throw DyError(DY_EINVAL, "Maximum read length exceeded.");
        if (paired) for (u32 s = 0; s < n_ranks; ++s) if (recv_headers[s].n_reads & 1)
            throw DyError(DY_EINVAL, "every source segment of a paired batch holds an even number of reads (first mates, then second mates)");
        HIP_CHECK(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        std::unique_ptr<dy_reads> r(new dy_reads);
        r->ctx = ctx; r->n = n; r->totalBases = nBases; r->maxLen = maxLen; r->minLen = minLen;
        r->recycle = true; r->pairLayout = paired != 0;
        takeBuffer(ctx, r->seqs, nBases + 64); takeBuffer(ctx, r->seqsRc, nBases + 64);
        takeBuffer(ctx, r->offs, (size_t)(n + 1) * 8);
        takeBuffer(ctx, r->masks, (size_t)n * W * 8 + 8);
        r->maskW = W; r->maskBins = n_bins;
        ctx->shIds.reserve((size_t)n * 4 + 64); ctx->shLen.reserve((size_t)(n + 1) * 8);
        ctx->shOdd.reserve((size_t)(n + 1) * 4 + 64); ctx->shOddOff.reserve((size_t)(n + 1) * 4 + 64);
        ctx->shFlags.reserve(64);
        HIP_CHECK(hipMemsetAsync(ctx->shFlags.p, 0, 16, st));
        const u8 * recv = (const u8 *)recv_dev;
        const dim3 perRead((n + 1 + 255) / 256), block(256);
        if (!paired)
        {
            hipLaunchKernelGGL(k_c1_unpack_scalars, perRead, block, 0, st, recv, T, W, n, ctx->shIds.as<u32>(), ctx->shLen.as<u64>(), ctx->shOdd.as<u32>(), r->masks.as<u64>(), ctx->shFlags.as<u32>());
            exclusiveSum(ctx, ctx->shLen.as<u64>(), r->offs.as<u64>(), (size_t)n + 1);
            exclusiveSum(ctx, ctx->shOdd.as<u32>(), ctx->shOddOff.as<u32>(), (size_t)n + 1);
            if (n) hipLaunchKernelGGL(k_c1_unpack_bases, dim3(std::min<u32>((n + 15) / 16, 65536u)), block, 0, st, recv, T, W, (const u32 *)nullptr, n, r->offs.as<u64>(), ctx->shOddOff.as<u32>(),
                                      r->offs.as<u64>(), nBases, r->seqs.as<u8>(), r->seqsRc.as<u8>(), ctx->shFlags.as<u32>());
        }
        else
        {
            takeBuffer(ctx, r->orig, (size_t)n * 4 + 64);
            ctx->shFlatLens.reserve((size_t)(n + 1) * 8); ctx->shFlatMasks.reserve((size_t)n * W * 8 + 64); ctx->shLenOff.reserve((size_t)(n + 1) * 8);
            if (n) hipLaunchKernelGGL(k_pair_layout, dim3((n + 255) / 256), block, 0, st, T, n, r->orig.as<u32>());
            hipLaunchKernelGGL(k_c1_unpack_scalars, perRead, block, 0, st, recv, T, W, n, ctx->shIds.as<u32>(), ctx->shFlatLens.as<u64>(), ctx->shOdd.as<u32>(), ctx->shFlatMasks.as<u64>(), ctx->shFlags.as<u32>());
            exclusiveSum(ctx, ctx->shFlatLens.as<u64>(), ctx->shLenOff.as<u64>(), (size_t)n + 1);                // offsets in arrival order
            exclusiveSum(ctx, ctx->shOdd.as<u32>(), ctx->shOddOff.as<u32>(), (size_t)n + 1);
            hipLaunchKernelGGL(k_permute_scalars, perRead, block, 0, st, r->orig.as<u32>(), n, W, ctx->shFlatLens.as<u64>(), ctx->shFlatMasks.as<u64>(), ctx->shLen.as<u64>(), r->masks.as<u64>());
            exclusiveSum(ctx, ctx->shLen.as<u64>(), r->offs.as<u64>(), (size_t)n + 1);                           // offsets of the re-ordered batch
            if (n) hipLaunchKernelGGL(k_c1_unpack_bases, dim3(std::min<u32>((n + 15) / 16, 65536u)), block, 0, st, recv, T, W, r->orig.as<u32>(), n, ctx->shLenOff.as<u64>(), ctx->shOddOff.as<u32>(),
                                      r->offs.as<u64>(), nBases, r->seqs.as<u8>(), r->seqsRc.as<u8>(), ctx->shFlags.as<u32>());
        }
        HIP_CHECK(hipGetLastError());
        // what dy_shard_c2_pack needs of this batch: where each source's reads start in arrival order
        ctx->shArrival = T;
        ctx->shRanks = n_ranks; ctx->shBatch = r.get(); ctx->rawValid = false;
        *out = r.release();
    });
}

int dy_map_bins(dy_ctx * ctx, const dy_index * idx, dy_reads * reads, const dy_params * p, uint32_t * bin_loads, uint64_t * n_raw)
{
    return guarded([&] {
        if (!ctx || !idx || !reads || !p) throw DyError(DY_EINVAL, "dy_map_bins: NULL argument");
        if (idx->ctx != ctx || reads->ctx != ctx) throw DyError(DY_EINVAL, "index/reads belong to a different context");
        if (p->sensitivity > 2) throw DyError(DY_EINVAL, "sensitivity must be 0 (low), 1 (high) or 2 (full)");
        if (!(p->error_rate >= 0.0f && p->error_rate <= 0.10f)) throw DyError(DY_EINVAL, "error rate must be within [0, 0.10]");
        const bool paired = p->paired != 0;
        if (paired && !reads->pairLayout) throw DyError(DY_EINVAL, "a paired dy_map_bins needs a batch built by dy_shard_c1_unpack");
        HIP_CHECK(hipSetDevice(ctx->device));
        const u32 nReads = reads->n, nBins = (u32)idx->bins.size(), W = (nBins + 63) / 64;
        if (reads->maskW != W || reads->maskBins != nBins) throw DyError(DY_EINVAL, "the reads carry no bin masks for this index");
        const u32 maxErr = readErrorsOf(reads->maxLen, p->error_rate);
        const u32 S0 = seedCountOf(maxErr, 0);
        if (S0 > MAX_SEEDS) throw DyError(DY_EUNSUPPORTED, "more than 32 seeds per read (read length x error rate too large)");
        ctx->stats.items = ctx->stats.raw_matches = ctx->stats.hard_items = ctx->stats.pool_retries = 0;
        ctx->rawValid = false; ctx->lastRaw = 0; ctx->lastItems = 0; ctx->lastPaired = paired; ctx->lastBins = nBins; ctx->lastHasWorkList = false;
        if (ctx->shBatch != reads) ctx->shRanks = 0;            // not a batch of dy_shard_c1_unpack: dy_shard_c2_pack has no arrival table for it
        if (bin_loads) for (u32 b = 0; b < nBins; ++b) bin_loads[b] = 0;
        if (n_raw) *n_raw = 0;
        if (!nReads) { ctx->rawValid = true; return; }
        // never search a bin whose FM structures are not resident
        hipLaunchKernelGGL(k_and_masks, dim3((u32)(((u64)nReads * W + 255) / 256)), dim3(256), 0, ctx->stream, reads->masks.as<u64>(), nReads, W, idx->dOwnedMask.as<u64>());
        u32 nItems = 0, nPool = 0;
        runSearchExtend(ctx, idx, reads, p, S0, nItems, nPool);
        if (paired)
        {   // the child mapper's pairing stages are bin-local: they run here, on the owner of the bin (reference src/mapper.h:927-1172)
            runFinalize(ctx, idx, reads, p, nullptr, ctx->pool.as<RawMatch>(), nPool, nItems, maxMatchErrorsOf(reads, p, maxErr), PE_OWNER);
            ctx->shOrig.reserve((size_t)nReads * 4 + 64);
            HIP_CHECK(hipMemcpyAsync(ctx->shOrig.p, reads->orig.p, (size_t)nReads * 4, hipMemcpyDeviceToDevice, ctx->stream));
        }
        // one pinned fetch (every pageable copy is a round trip of its own): the flags, and the work list's bin boundaries if asked for
        ctx->hostStatus.reserve((size_t)(nBins + 1) * 4 + 64);
        u32 * status = ctx->hostStatus.as<u32>();
        status[0] = status[1] = 0;
        if (paired) HIP_CHECK(hipMemcpyAsync(status, ctx->counters.as<u32>() + COUNTER_BAD_GROUP, 4, hipMemcpyDeviceToHost, ctx->stream));      // k_group_matches ran (pairing stages)
        if (ctx->shBatch == reads) HIP_CHECK(hipMemcpyAsync(status + 1, ctx->shFlags.p, 4, hipMemcpyDeviceToHost, ctx->stream));
        if (bin_loads) HIP_CHECK(hipMemcpyAsync(status + 16, ctx->binStart.p, (size_t)(nBins + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (paired || ctx->shBatch == reads || bin_loads) fetchSync(ctx);
        if (status[1]) throw DyError(DY_EINVAL, "dy_shard_c1_unpack: the received segments are inconsistent with their headers");
        if (status[0]) throw DyError(DY_EHIP, "internal error: a raw match record is inconsistent with the per-read counts");
        if (bin_loads) for (u32 b = 0; b < nBins; ++b) bin_loads[b] = status[16 + b + 1] - status[16 + b];
        if (n_raw) *n_raw = nPool;
        ctx->lastRaw = nPool; ctx->lastItems = nItems; ctx->rawValid = true; ctx->lastHasWorkList = true;
        float ms;
        HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3])); ctx->stats.ms_items = ms;
        HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev[3], ctx->ev[4])); ctx->stats.ms_search = ms;
        HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5])); ctx->stats.ms_extend = ms;
    });
}

size_t dy_shard_c2_row_bytes(uint32_t n_bins) { return sizeof(SegHeader) + (((size_t)n_bins * 4 + 7) & ~(size_t)7); }

int dy_shard_c2_pack(dy_ctx * ctx, void * rows_dev, void ** send_dev)
{
    return guarded([&] {
        if (!ctx || !rows_dev || !send_dev) throw DyError(DY_EINVAL, "dy_shard_c2_pack: NULL argument");
        if (!ctx->rawValid || !ctx->shRanks) throw DyError(DY_EINVAL, "dy_map_bins on a batch of dy_shard_c1_unpack must run first");
        HIP_CHECK(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        const u32 nRanks = ctx->shRanks, n = ctx->lastRaw, nItems = ctx->lastItems, nBins = ctx->lastBins;
        const bool paired = ctx->lastPaired;
        const SegTable & T = ctx->shArrival;
        ctx->shCounts.reserve(4 * MAX_RANKS * 4 + 64);
        u32 * counts = ctx->shCounts.as<u32>(), * itemCounts = counts + MAX_RANKS, * cursors = counts + 2 * MAX_RANKS;     // cursors: raw, then pairing records
        HIP_CHECK(hipMemsetAsync(counts, 0, 2 * MAX_RANKS * 4, st));
        const u32 * orig = paired ? ctx->shOrig.as<u32>() : nullptr;
        const u32 * nPoolDev = ctx->counters.as<u32>() + COUNTER_POOL64, * nItemsDev = ctx->devItemCount;
        const u32 rawBlocks = std::min<u32>((n + 255) / 256, 512u), itemBlocks        // (every block ends in same-address atomics: few, fat blocks)
                   = std::min<u32>((nItems + 255) / 256, 2048u);
        if (n) hipLaunchKernelGGL(k_raw_seg_count, dim3(rawBlocks), dim3(256), 0, st, ctx->pool.as<RawMatch>(), nPoolDev, T, orig, counts);
        if (paired && nItems)
            hipLaunchKernelGGL(k_pe_items_seg_count, dim3(itemBlocks), dim3(256), 0, st, ctx->peItems.as<PeItem>(), ctx->itemRead.as<u32>(), nItemsDev, T, orig, itemCounts);
        ctx->shSegOff2.reserve((MAX_RANKS + 1) * 8);
        hipLaunchKernelGGL(k_c2_rows, dim3(nRanks), dim3(256), 0, st, counts, paired ? itemCounts : (const u32 *)nullptr, nRanks, ctx->lastHasWorkList ? ctx->binStart.as<u32>() : (const u32 *)nullptr,
                           nBins, (u32)dy_shard_c2_row_bytes(nBins), (u8 *)rows_dev, ctx->shSegOff2.as<u64>(), cursors);
        ctx->shSend2.reserve((size_t)n * sizeof(RawMatch) + (paired ? (size_t)nItems * sizeof(PeItemRec) : 0) + (size_t)nRanks * 16 + 64);
        if (n) hipLaunchKernelGGL(k_raw_seg_scatter, dim3(rawBlocks), dim3(256), 0, st, ctx->pool.as<RawMatch>(), nPoolDev, T, orig, ctx->shIds.as<u32>(), cursors,
                                  ctx->shSegOff2.as<u64>(), ctx->shSend2.as<u8>());
        if (paired && nItems)
            hipLaunchKernelGGL(k_pe_items_seg_scatter, dim3(itemBlocks), dim3(256), 0, st, ctx->peItems.as<PeItem>(), ctx->itemRead.as<u32>(), ctx->itemBin.as<u16>(), nItemsDev, T, orig,
                               ctx->shIds.as<u32>(), cursors + MAX_RANKS, ctx->shSegOff2.as<u64>(), counts, ctx->shSend2.as<u8>());
        HIP_CHECK(hipGetLastError());
        *send_dev = ctx->shSend2.p;
    });
}

int dy_shard_finalize(dy_ctx * ctx, const dy_index * idx, dy_reads * reads, const dy_params * p, const void * recv_dev, uint32_t n_ranks, const void * recv_rows,
                      dy_result ** out)
{
    return guarded([&] {
        if (!ctx || !idx || !reads || !p || !recv_rows || !out) throw DyError(DY_EINVAL, "dy_shard_finalize: NULL argument");
        if (idx->ctx != ctx || reads->ctx != ctx) throw DyError(DY_EINVAL, "index/reads belong to a different context");
        if (n_ranks == 0 || n_ranks > MAX_RANKS) throw DyError(DY_EINVAL, "1..64 ranks");
        const bool paired = p->paired != 0;
        if (paired && (reads->n & 1)) throw DyError(DY_EINVAL, "a paired batch holds an even number of reads (first mates, then second mates)");
        HIP_CHECK(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        const u32 nReads = reads->n, nBins = (u32)idx->bins.size(), W = (nBins + 63) / 64;
        if (paired && (reads->maskW != W || reads->maskBins != nBins)) throw DyError(DY_EINVAL, "the reads carry no bin masks for this index (dy_ibf_query + dy_shard_c1_pack first)");
        // the rows of all sources: segment sizes, and the items per bin of the whole job (sortedBins(), reference src/d_mapper.h:843-854)
        const size_t rowBytes = dy_shard_c2_row_bytes(nBins);
        SegTable T; memset(&T, 0, sizeof(T)); T.nRanks = n_ranks;
        std::vector<u32> loads(nBins, 0);
        u64 off = 0, nRaw = 0, nPeItems = 0;
        for (u32 s = 0; s < n_ranks; ++s)
        {
            SegHeader h; memcpy(&h, (const u8 *)recv_rows + s * rowBytes, sizeof(h));
            if (h.bytes != c2SegBytes(h.a, h.b)) throw DyError(DY_EINVAL, "dy_shard_finalize: a segment header does not describe its own size");
            T.off[s] = off; T.start[s] = (u32)nRaw; T.start2[s] = (u32)nPeItems;
            off += h.bytes; nRaw += h.a; nPeItems += h.b;
            if (nRaw >= 0xFFFFFFF0ull) throw DyError(DY_EUNSUPPORTED, "more than 2^32 raw matches in one batch: use a smaller reads batch");
            if (nPeItems >= 0xFFFFFFF0ull) throw DyError(DY_EINVAL, "dy_shard_finalize: too many pairing records");
            const u32 * l = (const u32 *)((const u8 *)recv_rows + s * rowBytes + sizeof(SegHeader));
            for (u32 b = 0; b < nBins; ++b) loads[b] = (u32)std::min<u64>((u64)loads[b] + l[b], 0xFFFFFFFFull);
        }
        T.start[n_ranks] = (u32)nRaw; T.start2[n_ranks] = (u32)nPeItems;
        if (!paired && nPeItems) throw DyError(DY_EINVAL, "dy_shard_finalize: pairing records given for a single-end batch");
        if ((nRaw || nPeItems) && !recv_dev) throw DyError(DY_EINVAL, "dy_shard_finalize: NULL buffer");
        const u32 nPool = (u32)nRaw;
        const u32 maxErr = readErrorsOf(reads->maxLen, p->error_rate);
        const u32 maxMatchErr = maxMatchErrorsOf(reads, p, maxErr);
        std::unique_ptr<dy_result> res = makeResult(ctx, reads, maxMatchErr);
        res->nPool = nPool;
        ctx->stats.raw_matches = nPool;
        if (!nReads) { *out = res.release(); return; }
        ctx->counters.reserve(COUNTER_BYTES);
        ctx->readMatchCount.reserve((size_t)(nReads + 1) * 4);
        HIP_CHECK(hipMemsetAsync(ctx->counters.p, 0, COUNTER_BYTES, st));
        HIP_CHECK(hipMemsetAsync(ctx->readMatchCount.p, 0, (size_t)(nReads + 1) * 4, st));
        HIP_CHECK(hipEventRecord(ctx->ev[5], st));
        u32 nItems = 0;
        const u8 * recv = (const u8 *)recv_dev;
        if (paired)
        {   // the home reads' own work list (their OR-ed masks) indexes the records of the owners: copyCigars, reference src/d_mapper.h:186-215
            nItems = buildWorkList(ctx, reads, nBins);
            ctx->peItems.reserve((size_t)nItems * sizeof(PeItem) + 64);
            ctx->peHasRescue.reserve((size_t)nReads * 4 + 64);
            HIP_CHECK(hipMemsetAsync(ctx->peItems.p, 0, (size_t)nItems * sizeof(PeItem), st));
            HIP_CHECK(hipMemsetAsync(ctx->peHasRescue.p, 0, (size_t)nReads * 4, st));
            if (nPeItems)
            {
                PeView V = {};
                V.items = ctx->peItems.as<PeItem>(); V.itemRead = ctx->itemRead.as<u32>(); V.itemBin = ctx->itemBin.as<u16>(); V.binStart = ctx->binStart.as<u32>();
                V.nBins = nBins; V.nItems = nItems; V.readHasRescue = ctx->peHasRescue.as<u32>();
                hipLaunchKernelGGL(k_pe_items_unpack, dim3((u32)((nPeItems + 255) / 256)), dim3(256), 0, st, recv, T, (u32)nPeItems, V, nReads, ctx->counters.as<u32>() + 7);
            }
        }
        // sortedBins() over the loads of the whole job
        ctx->binOrderRank.reserve((size_t)nBins * 4); ctx->shLoads.reserve((size_t)nBins * 4 + 64);
        HIP_CHECK(hipMemcpyAsync(ctx->shLoads.p, loads.data(), (size_t)nBins * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_bin_rank_from_loads, dim3((nBins + 255) / 256), dim3(256), 0, st, ctx->shLoads.as<u32>(), nBins, ctx->binOrderRank.as<u32>());
        ctx->shRawFlat.reserve((size_t)nPool * sizeof(RawMatch) + 64);
        if (nPool)
            hipLaunchKernelGGL(k_c2_unpack_raw, dim3((nPool + 255) / 256), dim3(256), 0, st, recv, T, nPool, nReads, ctx->shRawFlat.as<RawMatch>(), ctx->readMatchCount.as<u32>(),
                               ctx->counters.as<u32>() + 7);
        runFinalize(ctx, idx, reads, p, res.get(), ctx->shRawFlat.as<RawMatch>(), nPool, nItems, maxMatchErr, PE_HOME);
        HIP_CHECK(hipEventRecord(ctx->ev[6], st));
        res->readLens = reads->lens;
        u32 flags[COUNTER_BYTES / 4];
        ctx->hostCounters.reserve(COUNTER_BYTES);
        HIP_CHECK(hipMemcpyAsync(ctx->hostCounters.p, ctx->counters.p, COUNTER_BYTES, hipMemcpyDeviceToHost, st));
        fetchSync(ctx);
        memcpy(flags, ctx->hostCounters.p, COUNTER_BYTES);
        collectSweepTime(ctx, reads);
        collectDeferredSweepTime(ctx);
        if (flags[7]) throw DyError(DY_EINVAL, "dy_shard_finalize: a raw match or pairing record names a read / bin outside the batch's selection");
        if (flags[COUNTER_BAD_GROUP]) throw DyError(DY_EHIP, "internal error: a raw match record is inconsistent with the per-read counts");
        float ms;
        HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev[5], ctx->ev[6])); ctx->stats.ms_finalize = ms;
        *out = res.release();
    });
}

} // extern "C"

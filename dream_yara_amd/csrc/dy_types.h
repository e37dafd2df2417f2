// Shared host/device plain-data types of libdyhip (MI355X / gfx950).
// Names follow the reference's domain: bins, contigs, reads, seeds, hits, matches.
#pragma once
#include <stdint.h>

namespace dy {

typedef uint8_t  u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t  i32;

// Geometry of an Interleaved Bloom filter (reference src/d_bloom_filter.h:245-254).
struct IbfGeom
{
    u64 nBits;
    u64 nHashPos;        // (bits - 256) / (64 W)
    u64 hashPosMask;     // nHashPos - 1 when nHashPos is a power of two, else 0
    u64 hashPosMagic;    // floor(2^64 / nHashPos) for the division-free modulo (unused when hashPosMask != 0)
    u32 nBins;
    u32 nHashes;
    u32 kmerSize;
    u32 W;               // 64-bit words per hash position
    u32 direct;          // 1 = k-mer direct addressing (reference src/d_kdx_filter.h): position = the k-mer hash itself, one 'hash function'
    u32 pad_;
    u64 preCalc[8];      // i ^ (k * 0x90b45d39fb6da1fa)
};

// One bin's slice of the device-resident index arena (on-disk layout: SURVEY.md Appendix A).
struct BinDesc
{
    u64 rankOff;         // first 32-byte rank block of this bin (lf.drv), in blocks
    u64 indOff;          // first 8-byte sampled-row entry (sa.ind)
    u64 valOff;          // first SA sample (sa.val)
    u64 contigOff;       // first base (Dna5 code, 1 B/base) of this bin's contigs
    u32 limOff;          // first contig limit (prefix sums relative to contigOff), nContigs+1 entries
    u32 nContigs;
    u32 n;               // BWT length incl. sentinels (sa.len)
    u32 pst[4];          // C[A..T] (lf.pst)
    u32 drs;             // sentinel substitute symbol (lf.drs)
    u32 drpOff;          // first sentinel row (lf.drp), nContigs entries, ascending
    u32 contigIdBase;    // global id of contig 0 of this bin (reference src/d_mapper.h:146)
    u64 saFullOff;       // first row of this bin in the expanded suffix array (device-only, see k_expand_sa)
    u64 prefixOff;       // first entry of this bin in the q-mer range table (device-only, see k_prefix_level)
};

// Raw extension result (reference src/bits_matches.h:116-127, unpacked; 16 B).
struct RawMatch
{
    u32 read;            // read id in the batch
    u32 begin;           // contig begin (0-based)
    u16 len;             // contig length of the match (the reference's contigEnd member)
    u8  errors;
    u8  strand;          // 1 = reverse complement
    u16 bin;
    u16 contig;          // contig id inside the bin
};

struct MapParams
{
    float errorRate;
    float strataRate;
    u32   sensitivity;
    u32   hitsThreshold;
};

enum { MAX_SEEDS = 32 };      // seeds per strand in round 0 = maxErrors + 1
enum { MAX_ERRORS = 127 };    // MemberBits<Match, Errors> = 7

} // namespace dy

/*
 * dyhip.h — C ABI of libdyhip.so, the MI355X (gfx950) implementation of DREAM-Yara's mapping hot path.
 *
 * The reference (temehi/dream_yara, /root/reference) has no FFI: its hot path sits behind C++ template
 * calls inside one process.  Each entry point below replaces one of those internal interfaces; the
 * reference file:line it stands in for is given per function.  Conventions: plain C types only, the caller
 * owns every buffer it passes, the library owns handles until the matching *_free; every function returns
 * DY_OK (0) or a negative DY_E* code and dy_last_error() gives the message; no exception crosses the ABI.
 * One host thread drives a context at a time.  There is NO CPU fallback: without a HIP device every
 * compute entry point fails with DY_ENODEVICE.
 */
#ifndef DYHIP_H_
#define DYHIP_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DY_ABI_VERSION 3

enum {
    DY_OK = 0,
    DY_EINVAL = -1,      /* bad argument */
    DY_EIO = -2,         /* file could not be read / written / parsed */
    DY_ENODEVICE = -3,   /* no usable HIP device */
    DY_EHIP = -4,        /* a HIP runtime call failed */
    DY_ENOMEM = -5,
    DY_EUNSUPPORTED = -6 /* valid for the reference, not implemented here (say which in dy_last_error) */
};

typedef struct dy_ctx   dy_ctx;    /* device, stream, workspaces */
typedef struct dy_ibf   dy_ibf;    /* Interleaved Bloom filter resident in HBM */
typedef struct dy_index dy_index;  /* all bins' FM indices + contigs resident in HBM */
typedef struct dy_reads dy_reads;  /* one batch of reads resident in HBM */
typedef struct dy_result dy_result;/* per-read results of one batch, resident in HBM + host mirror */

/* ---- library / context ------------------------------------------------------------------------------- */
int         dy_abi_version(void);
const char *dy_last_error(void);
/* Number of visible HIP devices (0 when none; never fails). */
int         dy_device_count(void);
/* Creates a context on `device` (ordinal among visible devices).  Replaces configureThreads()
 * (reference src/mapper.h:349-355): the "thread pool" of this implementation is one GPU + one stream. */
int         dy_ctx_create(int device, dy_ctx **out);
void        dy_ctx_free(dy_ctx *ctx);
/* Blocks until all work queued on the context's stream has finished. */
int         dy_ctx_sync(dy_ctx *ctx);
/* Returns the context's hipStream_t as an opaque pointer (for event timing by the caller). */
void       *dy_ctx_stream(dy_ctx *ctx);
/* Selects among result-identical code paths / device-only tables of a context (the library never reads the environment).
 * Keys (value range, default): sparse_sa (0-1, 0), prefix_q (0-12, 12), prefix_wide (0-1, 1: 16-byte q-mer table
 * entries that carry the bases following a unique q-mer, when they fit), no_isa (0-1, 0) — read by dy_index_load;
 * markers (0-1, 0: 1 = roctx ranges dy:sweep / dy:items / dy:search / dy:extend / dy:finalize around the stages' launches, for
 * rocprofv3 --marker-trace; the roctx library is bound at run time, only then),
 * extend_mode (0-3, 3: 1 or 2, chosen per batch from the share of items the last fast pass left over; 1 = K3 as a fast pass plus a second
 * launch over the items it could not finish; 0 = each block redoes its own unfinished items; 2 = one full pass);
 * finalize_block_min (>=1, 2048), finalize_no_keys (0-1, 0), pe_heavy_min (>=1, 256), pe_rescue_global (0-1, 0),
 * ibf_narrow (0-1, 0), ibf_full_tail (0-1, 0), ibf_lds_kb (4-60, 12), ibf_dense (0-1, 0), ibf_kernel (0-3, 0: sweep for more than 64 bins chosen from the filter fill; 1 lazy, 2 eager2, 3 wide),
 * ibf_slack (0-255, 0 = from the fill), ibf_own_stream (0-1, 0: 1 = dy_ibf_query runs on a second stream of the context, so the next
 * batch's sweep can be queued before the current batch is mapped), dp_small_blocks (64-65536, 4096) and dp_align_blocks (64-65536, 8192: grids of the
 * thread-level aggregation / alignment kernels), items_sorted (0-1, 1: work list as (bin, read) pairs + stable radix sort; 0 = bin-major
 * count matrix), sweep_chunk (0-2097151, 0: reads per chunk of a deferred sweep queued in front of every host wait; 0 = none, the sweep
 * runs whole when its masks are first needed), async_map (0-1, 0: 1 = single-end batches after a context's
 * first are queued whole — work list and pool sized from bounds, real sizes read on the device, one fetch at the end; measured slower than waiting for
 * the two sizes), pool_cap (initial raw-match
 * pool capacity, 0 = automatic; the pool grows and the extension is re-run on overflow).  Stands where the reference has
 * compile-time template switches (reference src/mapper.h:128-206 MapperConfig / Limits).  Unknown key or value: DY_EINVAL. */
int         dy_ctx_set_tuning(dy_ctx *ctx, const char *key, uint64_t value);

/* ---- Interleaved Bloom filter ------------------------------------------------------------------------ */
typedef struct {
    uint64_t n_bits;        /* including the 256-bit trailer */
    uint32_t n_bins;
    uint32_t n_hashes;
    uint32_t kmer_size;
    uint32_t words_per_pos; /* W = ceil(bins/64) */
    uint64_t n_hash_pos;    /* (bits-256)/(64 W) */
} dy_ibf_info;

/* SeqAnBloomFilter(const char*) load constructor — reference src/d_bloom_filter.h:79-89 (+ _getMetadata :255-265). */
int  dy_ibf_load(dy_ctx *ctx, const char *path, dy_ibf **out);
/* SeqAnBloomFilter(n_bins, n_hash, k, vec_size) — reference src/d_bloom_filter.h:48-58. n_bits includes the trailer. */
int  dy_ibf_create(dy_ctx *ctx, uint32_t n_bins, uint32_t n_hashes, uint32_t kmer_size, uint64_t n_bits, dy_ibf **out);
/* addKmers(text, binNo) — reference src/d_bloom_filter.h:91-94,293-315. `seq` = Dna5 codes (A0 C1 G2 T3 N4), host memory. */
int  dy_ibf_add(dy_ibf *f, const uint8_t *seq, uint64_t len, uint32_t bin);
/* save() — reference src/d_bloom_filter.h:150-154 (sdsl bit_vector file with the metadata trailer). */
int  dy_ibf_save(dy_ibf *f, const char *path);
int  dy_ibf_get_info(const dy_ibf *f, dy_ibf_info *info);
/* sdsl::bit_vector::data() / get_int — reference src/d_bloom_filter.h:190-193,262-264: copies the filter's 64-bit words
 * [first_word, first_word + n_words) to host memory, exactly as dy_ibf_save() would write them after the 8-byte bit count
 * (metadata trailer included).  n_bits/64 words in total. */
int  dy_ibf_read_words(dy_ibf *f, uint64_t first_word, uint64_t n_words, uint64_t *words_host);
void dy_ibf_free(dy_ibf *f);
/* DisOptions::getThreshold — reference src/d_mapper.h:83-92.  error_rate is the float the CLI stores (e/100.0). */
uint16_t dy_ibf_threshold(float error_rate, uint32_t kmer_size, uint16_t read_len);

/* k-mer direct-addressing filter, `-ft kmer_direct` — SeqAnKDXFilter, reference src/d_kdx_filter.h:45-268 (built by
 * src/d_build_filter.cpp:305-312, used by src/d_mapper.h:924-936).  One table position per possible k-mer (4^k positions of
 * W words), the k-mer's own hash is the position, no hash functions.  The handle is a dy_ibf: dy_ibf_add / dy_ibf_query /
 * dy_ibf_save / dy_ibf_get_info (n_hashes = 1) / dy_ibf_free apply.  The reference only bounds the hash by 64x the table
 * size and reads past the vector for k-mers poisoned by a leading N; here such k-mers select no bin. */
/* clearBins — reference src/d_bloom_filter.h:99-123 (and src/d_kdx_filter.h:74-99), the first half of dream_yara_update_filter
 * (src/d_update_filter.cpp:187-204): clears the given bins' bit in every hash position; dy_ibf_add then re-fills them. */
int  dy_ibf_clear_bins(dy_ibf *f, const uint32_t *bins, uint32_t n_bins);
int  dy_kdx_create(dy_ctx *ctx, uint32_t n_bins, uint32_t kmer_size, dy_ibf **out);
int  dy_kdx_load(dy_ctx *ctx, const char *path, dy_ibf **out);

/* ---- reads ------------------------------------------------------------------------------------------- */
/* Uploads a batch: `seqs` = concatenated Dna5 codes of the FORWARD reads, `offs` = n_reads+1 offsets.
 * The reverse complements (reference src/store_seqs.h:271-290, appendReverseComplement) are derived on the device.
 * Limits: n_reads <= 2097151 (21-bit read id, reference src/bits_matches.h:179-183), read length <= 8191 (:211-215). */
int  dy_reads_upload(dy_ctx *ctx, const uint8_t *seqs, const uint64_t *offs, uint32_t n_reads, dy_reads **out);
uint32_t dy_reads_count(const dy_reads *r);
void dy_reads_free(dy_reads *r);

/* clasifyLoadedReads — reference src/d_mapper.h:309-387 with whichBins — src/d_bloom_filter.h:161-224 (kernel K1).
 * For every read ORs whichBins(fwd) and whichBins(revcomp) into W 64-bit words (bin b = bit b%64 of word b/64).
 * bin_masks_host (n_reads*W words, may be NULL) receives a copy; the masks also stay on the device for dy_map_reads.
 * threshold 0 reproduces the reference's "every read to every bin" (src/d_mapper.h:320-333) without the duplicate pass. */
int  dy_ibf_query(dy_ibf *f, dy_reads *reads, uint16_t threshold, uint64_t *bin_masks_host);

/* The same sweep, DEFERRED: for the batch a caller maps NEXT.  A sweep holds every wave slot of the device, so a kernel queued beside it on
 * another stream only runs in its tail; what a sweep CAN fill are the host's waits inside the mapping of the CURRENT batch (for a count the
 * host must read, for a peer rank): with the tuning key sweep_chunk set, a chunk of that many reads is queued in front of each wait.
 * Whatever is left runs when the masks are first needed (dy_map_reads, dy_shard_c1_pack, ...).  One pending sweep per context; the caller
 * must not need the previous masks of `reads` any more.  Stands where the reference lets a prefetch thread load
 * the next reads while the current batch is mapped (src/file_prefetched.h:140; src/d_mapper.h:868-887 loop). */
int  dy_ibf_query_deferred(dy_ibf *f, dy_reads *reads, uint16_t threshold);
/* Queues one chunk of the pending deferred sweep (max_reads reads, 0 = the default chunk) on the context's stream: for a caller that
 * is about to wait for an event it recorded on dy_ctx_stream() (the fetch of an exchange's counts).  No-op without a pending sweep. */
int  dy_ctx_background_chunk(dy_ctx *ctx, uint32_t max_reads);

/* `-ft none`: every read goes to every bin (reference src/d_mapper.h:399-405) — sets the masks dy_ibf_query would set.  dy_map_reads does
 * this by itself when use_filter = 0; the bins-sharded sequence (dy_shard_c1_pack ...) needs the masks on the reads first. */
int  dy_reads_select_all_bins(dy_ctx *ctx, dy_reads *reads, uint32_t n_bins);

/* ---- index ------------------------------------------------------------------------------------------- */
/* loadContigs + loadContigsIndex for every bin — reference src/mapper.h:362-403, src/index_fm.h:84-103,
 * src/store_seqs.h:128-141.  Reads <dir>/<b>.{txt.size,txt.limits,txt.concat,rid.limits,rid.concat,sa.len,sa.ind,
 * sa.val,lf.pst,lf.drv,lf.drs,lf.drp} for b in [0,n_bins) and uploads them. */
int  dy_index_load(dy_ctx *ctx, const char *dir, uint32_t n_bins, dy_index **out);
uint32_t dy_index_bins(const dy_index *idx);
uint32_t dy_index_contigs(const dy_index *idx);                       /* over all bins */
const char *dy_index_contig_name(const dy_index *idx, uint32_t c);    /* global contig id, bin order */
uint64_t dy_index_contig_length(const dy_index *idx, uint32_t c);
/* What is resident for the index (observability; the device-only tables are sized from the free HBM at load). */
typedef struct {
    uint32_t n_bins, n_bins_resident;   /* bins in the directory / bins whose FM structures are on this device */
    uint32_t n_contigs, prefix_q;       /* q of the q-mer range table (0 = none) */
    uint64_t total_rows;                /* BWT rows over the resident bins */
    uint64_t bytes_rank, bytes_sparse_sa, bytes_contigs;                    /* the reference's own structures (lf.drv, sa.*, txt.concat unpacked) */
    uint64_t bytes_expanded_sa, bytes_prefix_table, bytes_isa_and_text;     /* device-only accelerators, 0 when not built */
} dy_index_info;
int  dy_index_get_info(const dy_index *idx, dy_index_info *info);
void dy_index_free(dy_index *idx);

/* ---- index construction (dream_yara_indexer) ----------------------------------------------------------- */
/* Suffix array of text[0, n) over symbols [0, sigma), every sequence terminated by its own sentinel symbol (so no two suffixes
 * compare equal), built on the device.  Stands where the reference lets SeqAn construct the FM index of a bin's reversed contigs
 * (reference src/d_indexer.cpp:278-333); BWT, rank dictionary and the sampled suffix array follow from it on the host. */
int  dy_suffix_array(dy_ctx *ctx, const uint32_t *text, uint64_t n, uint32_t sigma, uint32_t *sa_out);
/* The FM index arrays of one bin, built on the device from the same kind of text: contig c reversed with its Ns randomised, as symbols
 * n_contigs + base, followed by its sentinel symbol c; starts[c] = first position of contig c (n_contigs + 1 entries, the last = n).
 * Suffix array (as dy_suffix_array), then BWT, symbol counts, the rank dictionary blocks, the sentinel rows and the sampled suffix array
 * in the layouts of the reference's files (reference src/d_indexer.cpp:278-333 builds them inside SeqAn, the
 * sampling rate 10 is YaraFMConfig's, src/misc_types.h:151-177).  Host outputs, caller-allocated: drv_out ((n+63)/64 x 32 B = lf.drv),
 * ind_out ((n+31)/32 x 8 B = sa.ind), val_out (room for n/10 + n_contigs + 64 records of 5 B, or 6 B above 255 contigs = sa.val; *n_val
 * records are written), drp_out (n_contigs rows, ascending: lf.drp up to 255 contigs, the set bits of its dictionary above),
 * pst_out[5] (lf.pst), *drs_out (lf.drs). */
int  dy_fm_build(dy_ctx *ctx, const uint32_t *text, uint64_t n, uint32_t n_contigs, const uint32_t *starts, uint8_t *drv_out, uint8_t *ind_out, uint8_t *val_out,
                 uint64_t *n_val, uint32_t *drp_out, uint32_t *pst_out, uint8_t *drs_out);

/* ---- mapping ----------------------------------------------------------------------------------------- */
typedef struct {
    float    error_rate;      /* Options::errorRate  (reference src/mapper.h:68,99)  */
    float    strata_rate;     /* Options::strataRate (:70,101) */
    uint32_t sensitivity;     /* 0 low, 1 high, 2 full (edit-distance seeds, Options::sensitivity FULL) (:72,103) */
    uint32_t hits_threshold;  /* Options::hitsThreshold (:84,111), default 300 */
    uint32_t use_filter;      /* 0 = -ft none: every read goes to every bin (reference src/d_mapper.h:399-405) */
    uint32_t paired;          /* 1 = paired-end: reads [0,n/2) are the first mates, [n/2,n) the second mates (reference src/bits_reads.h) */
    uint32_t library_length;  /* Options::libraryLength, 0 = estimate per bin (reference src/mapper.h:76,929-985) */
    uint32_t library_dev;     /* Options::libraryDev,    0 = estimate per bin */
    uint32_t verify_matches;  /* Options::verifyMatches: mate rescue unless --no-indels (reference src/d_mapper.cpp:359) */
    float    indel_rate;      /* Options::indelRate (reference src/mapper.h:69,100), default 0.25 */
} dy_params;
void dy_params_default(dy_params *p);

/* One match, unpacked from the reference's bit-field (src/bits_matches.h:116-127). */
typedef struct {
    uint32_t contig;        /* global contig id (bin offset applied, reference src/d_mapper.h:146) */
    uint32_t begin;         /* 0-based */
    uint16_t length;        /* the reference's 13-bit contigEnd member */
    uint8_t  errors;
    uint8_t  is_rev;
} dy_match;

/* Per-read record: what MatchesWriter consumes (reference src/mapper_writer.h:296-345). */
typedef struct {
    dy_match primary;       /* valid iff mapped */
    uint32_t mapped;
    uint32_t best_count;    /* X0 */
    uint32_t sub_count;     /* X1 */
    uint32_t mapq;
    uint32_t set_begin;     /* [set_begin, set_begin+best_count+sub_count) in dy_result_matches(): matchesSet[read] */
    uint32_t cigar_begin;   /* [cigar_begin, cigar_begin+cigar_len) in dy_result_cigars() */
    uint32_t cigar_len;
    uint32_t paired;        /* paired-end: ReadsContext::paired of the main mapper (reference src/d_mapper.h:202-206) */
    uint32_t reserved;
    double   pair_prob;     /* paired-end: primaryMatchesProbs[read]; mapq already uses it when paired */
} dy_read_result;

/* runDisMapper's per-batch body — reference src/d_mapper.h:874-886: for every bin selected by the masks left on
 * `reads` by dy_ibf_query run _mapReadsImpl (:252-303: seeds, FM search K2, classify, rank, locate+extend K3,
 * aggregate, rank, align primaries), then finalizeMainMapper (:828-838: aggregate, rankMatches2, cigar transfer).
 * Everything runs on the device; the call returns after the results are also mirrored on the host. */
int  dy_map_reads(dy_ctx *ctx, const dy_index *idx, dy_reads *reads, const dy_params *p, dy_result **out);
/* Same, but asynchronous: results stay on the device until dy_result_fetch(). Used for timing the hot path. */
int  dy_map_reads_async(dy_ctx *ctx, const dy_index *idx, dy_reads *reads, const dy_params *p, dy_result **out);
int  dy_result_fetch(dy_result *r);
const dy_read_result *dy_result_reads(const dy_result *r);     /* n_reads entries */
const dy_match       *dy_result_matches(const dy_result *r);   /* matchesSet of all reads back to back */
uint64_t              dy_result_n_matches(const dy_result *r);
/* CIGAR ops packed as (count << 4) | op with op 0=M 1=I 2=D (BAM numbering). */
const uint32_t       *dy_result_cigars(const dy_result *r);
void dy_result_free(dy_result *r);

/* Stage counters of the last dy_map_reads call (observability; mirrors Stats, reference src/mapper.h:238-279). */
typedef struct {
    uint64_t items;          /* (read,bin) pairs = filteredReads (reference src/d_mapper.h:442) */
    uint64_t raw_matches;    /* extension results before compaction */
    uint64_t hard_items;     /* items re-seeded by classifyReads */
    uint64_t pool_retries;   /* times the match pool had to grow */
    double   ms_ibf;         /* K1 of the last dy_ibf_query on this context (hipEvent on the context's stream) */
    double   ms_items, ms_search, ms_extend, ms_finalize;   /* device time per stage of the last dy_map_reads */
    uint32_t ibf_kernel;     /* sweep the last dy_ibf_query ran: 1 k_ibf_query_two, 2 k_ibf_query<7>, 3 k_ibf_query<14>, 4 k_ibf_query_lazy,
                                5 k_ibf_query_eager2, 6 k_ibf_query_wide (0 = none yet) */
    uint32_t ibf_slack;      /* first check of that sweep at nK - threshold + slack k-mers (derived from the sampled filter fill) */
    uint32_t ibf_tail_pre;   /* k_ibf_query_two: tail k-mers requested with the first round (4 where the filter's fill makes weak bins the rule, else 0) */
    uint32_t second_pass_items;   /* items K3's fast pass could not finish (a side with more than one difference, a third match): redone in full */
} dy_map_stats;
int  dy_map_get_stats(const dy_ctx *ctx, dy_map_stats *s);

/* ---- text in, text out: FASTQ -> reads and results -> SAM on the device ------------------------------------------
 * The host share of a GPU box is a few cores: parsing FASTQ (215 B per 100-bp read) and formatting SAM (275 B) cost them more time
 * than the mapping costs the device.  With these entry points the host only moves bytes — file -> pinned buffer -> device ->
 * pinned buffer -> file — and the records are byte-identical to the host path's. */
/* Pinned host memory owned by the context (slots 0..7, grow-only): read a file straight into it / write a file straight out of it. */
int  dy_host_buffer(dy_ctx *ctx, uint32_t slot, uint64_t bytes, void **out);
/* readRecords of a four-line FASTQ file — reference src/mapper.h:445-469 (loadReads; SeqAn SeqFileIn): `text` holds the next bytes of the
 * file (host memory, best a dy_host_buffer), at_eof != 0 if they reach its end.  The batch takes the first max_reads records (all
 * complete ones if the window holds fewer: the caller retries with a larger window unless at_eof); *consumed = the bytes they occupy.
 * Dna5 codes, qualities (clamped to 6 bits, '!' under N) and names (cut at the first blank, tab or '/') as the reference stores them.
 * DY_EUNSUPPORTED (with the reason) for anything but plain four-line records — blank lines, CR line ends, wrapped sequences, FASTA:
 * parse such input on the host and use dy_reads_upload. */
int  dy_reads_from_fastq(dy_ctx *ctx, const void *text, uint64_t n_bytes, uint32_t max_reads, uint32_t at_eof, dy_reads **out, uint64_t *consumed);
uint64_t dy_reads_total_bases(const dy_reads *r);
/* MatchesWriter for single-end reads — reference src/mapper_writer.h:176-225 (tags NM X0 X1 RG XA, MAPQ), :296-345 (record), :593-634 (XA):
 * the SAM records of a batch of dy_reads_from_fastq, in read order, formatted on the device from the results left by
 * dy_map_reads_async (no dy_result_fetch needed) and copied into the context's pinned slot `out_slot`; *text stays valid until the
 * slot is used again.  DY_EUNSUPPORTED if a mapping quality lies on a rounding boundary of the device's arithmetic (then format the
 * batch on the host from dy_result_fetch).  Paired-end and BAM records are formatted by the caller. */
int  dy_result_sam(dy_result *r, const dy_reads *reads, const dy_index *idx, const char *read_group, uint32_t out_slot, const char **text, uint64_t *n_bytes,
                   uint64_t *n_mapped);

/* ---- bins sharded over ranks (one process or host thread per GPU) -----------------------------------------
 * The reference's unit of distribution is the bin (src/d_mapper.h:876-884: one child Mapper per bin; copyMatches
 * :127-155, copyCigars :186-215 and finalizeMainMapper :828-838 merge per read).  With G ranks every rank holds the filter,
 * the contig text of ALL bins (the home rank of a read aligns its primary) and the FM structures of the bins it owns.
 * The library never communicates: it packs each of the two exchanges into ONE device buffer (segments destination-major) plus
 * one small row of counts per destination, and unpacks what the caller received.  The CALLER exchanges the rows, fetches them
 * to the host (its one stream synchronisation per exchange: the sizes of an all-to-all(v) are host arguments) and moves the
 * segments (RCCL ncclSend/ncclRecv group, MPI, peer copies) — stream-ordered on dy_ctx_stream(), nothing else waits:
 *   dy_ibf_query(home reads)                                       K1 on the rank's own reads
 *   dy_shard_c1_pack            [C1] reads selected for each owner rank: headers (G x 32 B) + packed segments
 *       -> all-to-all of the headers, fetch, all-to-all(v) of the segments
 *   dy_shard_c1_unpack + dy_map_bins                               K2+K3 on the owner, for its bins only (paired: + the child mapper's
 *                                                                  pairing, library statistics and mate rescue, which are bin-local)
 *   dy_shard_c2_pack            [C2] raw matches (+ pairing records) for each read's home rank: rows (G x dy_shard_c2_row_bytes) + segments;
 *                               every row also carries this rank's items per bin, which the receivers add up (sortedBins(), :843-854)
 *       -> all-to-all of the rows, fetch, all-to-all(v) of the segments
 *   dy_shard_finalize(home reads, received segments, received rows)   K4 = aggregate + rankMatches2 on the home rank
 * Results equal dy_map_reads on the concatenation of all ranks' reads; for a paired job the concatenation is
 * [first mates of rank 0, first mates of rank 1, ... | second mates of rank 0, ...] (mates stay together: src/d_mapper.h:350-358).
 * With one rank the packed send buffer IS the receive buffer (no copy, no communication). */

/* Like dy_index_load, but FM structures only for bins with owned[b] != 0 (contigs and names for all). */
int  dy_index_load_sharded(dy_ctx *ctx, const char *dir, uint32_t n_bins, const uint8_t *owned, dy_index **out);

/* One row per peer rank.  C1: reads / bases in the segment, aux = longest | shortest << 32 read length of the sender's batch (bounds).
 * C2: n_reads = raw matches, n_bases = pairing records in the segment.  `bytes` = size of the segment (a multiple of 16). */
typedef struct { uint64_t n_reads, n_bases, bytes, aux; } dy_seg_header;

/* [C1] loadFilteredReads across ranks — reference src/d_mapper.h:394-443.  owner_masks: n_ranks x W words (host), rank d's bit set of
 * owned bins.  paired != 0: `reads` holds first mates then second mates and the masks of the two mates are OR-ed first, so both
 * travel wherever either is selected.  Queues the packing on the context's stream and returns at once:
 *   headers_dev (DEVICE, caller-owned, n_ranks x dy_seg_header): row d describes the segment for rank d;
 *   *send_dev: the context's packed buffer, segments for ranks 0..n_ranks-1 back to back — per segment the read ids on this rank
 *   (u32), lengths (u32), masks restricted to the destination's bins (W words per read), forward Dna5 codes at 4 bits each (two per byte, the first in the
 *   low half; every read starts a byte of its own); valid until the next call.  The layout is the library's own on both ends. */
int  dy_shard_c1_pack(dy_ctx *ctx, dy_reads *reads, uint32_t n_ranks, const uint64_t *owner_masks, uint32_t paired, void *headers_dev, void **send_dev);
/* The owner-side batch from the segments received from ranks 0..n_ranks-1 (recv_dev: DEVICE, back to back; recv_headers: HOST, the rows
 * the sources sent for this rank).  A paired batch is re-ordered to [all first mates | all second mates], pairs in source order,
 * which is what the child mapper sees in the reference (src/d_mapper.h:394-443).  Queued on the stream, no host synchronisation;
 * recv_dev may be reused once work queued after this call has started. */
int  dy_shard_c1_unpack(dy_ctx *ctx, const void *recv_dev, uint32_t n_ranks, const dy_seg_header *recv_headers, uint32_t n_bins, uint32_t paired, dy_reads **out);
/* K2 + K3 over the (read,bin) items of the masks; with p->paired also the bin-local pairing stages of the child mapper
 * (reference src/mapper.h:927-1172).  bin_loads (host, n_bins; may be NULL) receives the items per bin, n_raw (may be NULL) the
 * number of raw matches; they (and the pairing records) stay in the context until dy_shard_c2_pack. */
int  dy_map_bins(dy_ctx *ctx, const dy_index *idx, dy_reads *reads, const dy_params *p, uint32_t *bin_loads, uint64_t *n_raw);
/* [C2] copyMatches / copyCigars across ranks — reference src/d_mapper.h:127-215.  Bytes of one row: a dy_seg_header + n_bins x u32 (padded to 8). */
size_t dy_shard_c2_row_bytes(uint32_t n_bins);
/* After dy_map_bins on a batch of dy_shard_c1_unpack: rewrites the read id of every raw match to the id on its home rank and packs
 * the matches (16 B each; paired jobs: then one DY_PE_ITEM_BYTES record per (read,bin) item that carries information — the child's
 * primary as chosen by the pairing, `paired`, the pair probability, a rescued mate) by home rank.  rows_dev (DEVICE, caller-owned,
 * n_ranks rows); *send_dev: the context's packed buffer.  Queued on the stream, no host synchronisation. */
int  dy_shard_c2_pack(dy_ctx *ctx, void *rows_dev, void **send_dev);
/* K4 on the home rank over the segments gathered from all owners (recv_dev: DEVICE; recv_rows: HOST, the rows the owners sent for this
 * rank).  Returns like dy_map_reads_async. */
int  dy_shard_finalize(dy_ctx *ctx, const dy_index *idx, dy_reads *reads, const dy_params *p, const void *recv_dev, uint32_t n_ranks,
                       const void *recv_rows, dy_result **out);
#define DY_RAW_MATCH_BYTES 16
#define DY_PE_ITEM_BYTES 40

#ifdef __cplusplus
}
#endif
#endif /* DYHIP_H_ */

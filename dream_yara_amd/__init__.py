"""dream_yara_amd — Python binding of libdyhip.so (C ABI in include/dyhip.h).

This is the thin host-side mirror used by the tests and by bench.py: every call goes straight through ctypes into the
gfx950 library.  There is no Python or CPU implementation of the mapping path here; if the shared library has not been
built (python -m dream_yara_amd.build) or no HIP device is visible, the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DY_LIB") or os.path.join(_HERE, "csrc", "libdyhip.so")

DY_OK = 0
DY_ENODEVICE = -3


class DyError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libdyhip error %d: %s" % (code, msg))
        self.code = code


class IbfInfo(C.Structure):
    _fields_ = [("n_bits", C.c_uint64), ("n_bins", C.c_uint32), ("n_hashes", C.c_uint32), ("kmer_size", C.c_uint32),
                ("words_per_pos", C.c_uint32), ("n_hash_pos", C.c_uint64)]


class Params(C.Structure):
    _fields_ = [("error_rate", C.c_float), ("strata_rate", C.c_float), ("sensitivity", C.c_uint32),
                ("hits_threshold", C.c_uint32), ("use_filter", C.c_uint32), ("paired", C.c_uint32), ("library_length", C.c_uint32),
                ("library_dev", C.c_uint32), ("verify_matches", C.c_uint32), ("indel_rate", C.c_float)]


class Match(C.Structure):
    _fields_ = [("contig", C.c_uint32), ("begin", C.c_uint32), ("length", C.c_uint16), ("errors", C.c_uint8),
                ("is_rev", C.c_uint8)]


class ReadResult(C.Structure):
    _fields_ = [("primary", Match), ("mapped", C.c_uint32), ("best_count", C.c_uint32), ("sub_count", C.c_uint32),
                ("mapq", C.c_uint32), ("set_begin", C.c_uint32), ("cigar_begin", C.c_uint32), ("cigar_len", C.c_uint32),
                ("paired", C.c_uint32), ("reserved", C.c_uint32), ("pair_prob", C.c_double)]


class MapStats(C.Structure):
    _fields_ = [("items", C.c_uint64), ("raw_matches", C.c_uint64), ("hard_items", C.c_uint64), ("pool_retries", C.c_uint64),
                ("ms_ibf", C.c_double), ("ms_items", C.c_double), ("ms_search", C.c_double), ("ms_extend", C.c_double),
                ("ms_finalize", C.c_double), ("ibf_kernel", C.c_uint32), ("ibf_slack", C.c_uint32), ("ibf_tail_pre", C.c_uint32), ("second_pass_items", C.c_uint32)]


class IndexInfo(C.Structure):
    _fields_ = [("n_bins", C.c_uint32), ("n_bins_resident", C.c_uint32), ("n_contigs", C.c_uint32), ("prefix_q", C.c_uint32), ("total_rows", C.c_uint64),
                ("bytes_rank", C.c_uint64), ("bytes_sparse_sa", C.c_uint64), ("bytes_contigs", C.c_uint64), ("bytes_expanded_sa", C.c_uint64),
                ("bytes_prefix_table", C.c_uint64), ("bytes_isa_and_text", C.c_uint64)]


IBF_KERNEL_NAMES = {0: None, 1: "k_ibf_query_two", 2: "k_ibf_query<7>", 3: "k_ibf_query<14>", 4: "k_ibf_query_lazy", 5: "k_ibf_query_eager2", 6: "k_ibf_query_wide"}


MATCH_DTYPE = np.dtype([("contig", "<u4"), ("begin", "<u4"), ("length", "<u2"), ("errors", "u1"), ("is_rev", "u1")])
READ_RESULT_DTYPE = np.dtype([("primary", MATCH_DTYPE), ("mapped", "<u4"), ("best_count", "<u4"), ("sub_count", "<u4"),
                              ("mapq", "<u4"), ("set_begin", "<u4"), ("cigar_begin", "<u4"), ("cigar_len", "<u4"),
                              ("paired", "<u4"), ("reserved", "<u4"), ("pair_prob", "<f8")])

# every symbol include/dyhip.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("dy_abi_version", C.c_int, []),
    ("dy_last_error", C.c_char_p, []),
    ("dy_device_count", C.c_int, []),
    ("dy_ctx_create", C.c_int, [C.c_int, C.POINTER(_P)]),
    ("dy_ctx_free", None, [_P]),
    ("dy_ctx_sync", C.c_int, [_P]),
    ("dy_ctx_stream", _P, [_P]),
    ("dy_ctx_set_tuning", C.c_int, [_P, C.c_char_p, C.c_uint64]),
    ("dy_ibf_load", C.c_int, [_P, C.c_char_p, C.POINTER(_P)]),
    ("dy_ibf_create", C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(_P)]),
    ("dy_ibf_add", C.c_int, [_P, _P, C.c_uint64, C.c_uint32]),
    ("dy_ibf_save", C.c_int, [_P, C.c_char_p]),
    ("dy_ibf_get_info", C.c_int, [_P, C.POINTER(IbfInfo)]),
    ("dy_ibf_read_words", C.c_int, [_P, C.c_uint64, C.c_uint64, _P]),
    ("dy_ibf_free", None, [_P]),
    ("dy_ibf_threshold", C.c_uint16, [C.c_float, C.c_uint32, C.c_uint16]),
    ("dy_reads_upload", C.c_int, [_P, _P, _P, C.c_uint32, C.POINTER(_P)]),
    ("dy_reads_count", C.c_uint32, [_P]),
    ("dy_reads_free", None, [_P]),
    ("dy_ibf_query", C.c_int, [_P, _P, C.c_uint16, _P]),
    ("dy_ibf_query_deferred", C.c_int, [_P, _P, C.c_uint16]),
    ("dy_ctx_background_chunk", C.c_int, [_P, C.c_uint32]),
    ("dy_host_buffer", C.c_int, [_P, C.c_uint32, C.c_uint64, C.POINTER(_P)]),
    ("dy_reads_from_fastq", C.c_int, [_P, _P, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    ("dy_reads_total_bases", C.c_uint64, [_P]),
    ("dy_result_sam", C.c_int, [_P, _P, _P, C.c_char_p, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("dy_reads_select_all_bins", C.c_int, [_P, _P, C.c_uint32]),
    ("dy_suffix_array", C.c_int, [_P, _P, C.c_uint64, C.c_uint32, _P]),
    ("dy_fm_build", C.c_int, [_P, _P, C.c_uint64, C.c_uint32, _P, _P, _P, _P, C.POINTER(C.c_uint64), _P, _P, _P]),
    ("dy_index_load", C.c_int, [_P, C.c_char_p, C.c_uint32, C.POINTER(_P)]),
    ("dy_index_bins", C.c_uint32, [_P]),
    ("dy_index_contigs", C.c_uint32, [_P]),
    ("dy_index_contig_name", C.c_char_p, [_P, C.c_uint32]),
    ("dy_index_contig_length", C.c_uint64, [_P, C.c_uint32]),
    ("dy_index_get_info", C.c_int, [_P, C.POINTER(IndexInfo)]),
    ("dy_index_free", None, [_P]),
    ("dy_params_default", None, [C.POINTER(Params)]),
    ("dy_map_reads", C.c_int, [_P, _P, _P, C.POINTER(Params), C.POINTER(_P)]),
    ("dy_map_reads_async", C.c_int, [_P, _P, _P, C.POINTER(Params), C.POINTER(_P)]),
    ("dy_result_fetch", C.c_int, [_P]),
    ("dy_result_reads", _P, [_P]),
    ("dy_result_matches", _P, [_P]),
    ("dy_result_n_matches", C.c_uint64, [_P]),
    ("dy_result_cigars", _P, [_P]),
    ("dy_result_free", None, [_P]),
    ("dy_map_get_stats", C.c_int, [_P, C.POINTER(MapStats)]),
    ("dy_ibf_clear_bins", C.c_int, [_P, _P, C.c_uint32]),
    ("dy_kdx_create", C.c_int, [_P, C.c_uint32, C.c_uint32, C.POINTER(_P)]),
    ("dy_kdx_load", C.c_int, [_P, C.c_char_p, C.POINTER(_P)]),
    ("dy_index_load_sharded", C.c_int, [_P, C.c_char_p, C.c_uint32, _P, C.POINTER(_P)]),
    ("dy_shard_c1_pack", C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint32, _P, C.POINTER(_P)]),
    ("dy_shard_c1_unpack", C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint32, C.c_uint32, C.POINTER(_P)]),
    ("dy_map_bins", C.c_int, [_P, _P, _P, C.POINTER(Params), _P, C.POINTER(C.c_uint64)]),
    ("dy_shard_c2_row_bytes", C.c_size_t, [C.c_uint32]),
    ("dy_shard_c2_pack", C.c_int, [_P, _P, C.POINTER(_P)]),
    ("dy_shard_finalize", C.c_int, [_P, _P, _P, C.POINTER(Params), _P, C.c_uint32, _P, C.POINTER(_P)]),
]
RAW_MATCH_BYTES = 16
PE_ITEM_BYTES = 40
SEG_HEADER_BYTES = 32
SEG_HEADER_DTYPE = np.dtype([("n_reads", "<u8"), ("n_bases", "<u8"), ("bytes", "<u8"), ("aux", "<u8")])

_lib = None


def load_library():
    """Loads libdyhip.so and binds every declared symbol. Raises if the library is missing: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libdyhip.so is not built (%s); run: python -m dream_yara_amd.build" % LIB_PATH)
    # One HIP runtime per process: PyTorch ships its own libamdhip64.so.7 (+ HSA runtime) and the harness uses torch.distributed next to this
    # library.  Whichever copy is loaded first serves both (same SONAME) — but if ROCm's copy came first, torch's other libraries bring a second
    # HSA runtime along and torch finds no GPU.  So torch's copy, when there is one, is loaded before libdyhip (no import of torch needed).
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is not None and spec.origin:
            hip = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
            if os.path.exists(hip):
                C.CDLL(hip, mode=C.RTLD_GLOBAL)
    except (ImportError, OSError, ValueError):
        pass
    lib = C.CDLL(LIB_PATH)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the ABI and the header drifted apart
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def _check(rc):
    if rc != DY_OK:
        raise DyError(rc, load_library().dy_last_error().decode(errors="replace"))


def device_count():
    return load_library().dy_device_count()


class Context:
    def __init__(self, device=0):
        self.lib = load_library()
        h = _P()
        _check(self.lib.dy_ctx_create(device, C.byref(h)))
        self.h = h

    def sync(self):
        _check(self.lib.dy_ctx_sync(self.h))

    def host_buffer(self, slot, nbytes):
        """Address of the context's pinned host slot (device-accessible), at least nbytes large."""
        out = _P()
        _check(self.lib.dy_host_buffer(self.h, slot, nbytes, C.byref(out)))
        return out.value

    def background_chunk(self, max_reads=0):
        """One chunk of the pending deferred sweep, to be called right before waiting for an event recorded on the context's stream."""
        _check(self.lib.dy_ctx_background_chunk(self.h, max_reads))

    def stream(self):
        """The context's hipStream_t as an integer (e.g. for torch.cuda.ExternalStream)."""
        return self.lib.dy_ctx_stream(self.h) or 0

    def set_tuning(self, key, value):
        _check(self.lib.dy_ctx_set_tuning(self.h, key.encode(), int(value)))

    def stats(self):
        s = MapStats()
        _check(self.lib.dy_map_get_stats(self.h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in MapStats._fields_}

    def close(self):
        if self.h:
            self.lib.dy_ctx_free(self.h)
            self.h = None


class Ibf:
    """Interleaved Bloom filter in HBM (reference src/d_bloom_filter.h, class SeqAnBloomFilter)."""

    def __init__(self, ctx, handle):
        self.ctx, self.lib, self.h = ctx, ctx.lib, handle

    @classmethod
    def load(cls, ctx, path):
        h = _P()
        _check(ctx.lib.dy_ibf_load(ctx.h, os.fsencode(path), C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def create(cls, ctx, n_bins, n_hashes, kmer_size, n_bits):
        h = _P()
        _check(ctx.lib.dy_ibf_create(ctx.h, n_bins, n_hashes, kmer_size, n_bits, C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def load_direct(cls, ctx, path):
        """-ft kmer_direct: the k-mer direct-addressing filter (reference src/d_kdx_filter.h)."""
        h = _P()
        _check(ctx.lib.dy_kdx_load(ctx.h, os.fsencode(path), C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def create_direct(cls, ctx, n_bins, kmer_size):
        h = _P()
        _check(ctx.lib.dy_kdx_create(ctx.h, n_bins, kmer_size, C.byref(h)))
        return cls(ctx, h)

    def add(self, codes, bin_no):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        _check(self.lib.dy_ibf_add(self.h, codes.ctypes.data, codes.size, bin_no))

    def clear_bins(self, bins):
        bins = np.ascontiguousarray(bins, dtype=np.uint32)
        _check(self.lib.dy_ibf_clear_bins(self.h, bins.ctypes.data, bins.size))

    def save(self, path):
        _check(self.lib.dy_ibf_save(self.h, os.fsencode(path)))

    def info(self):
        i = IbfInfo()
        _check(self.lib.dy_ibf_get_info(self.h, C.byref(i)))
        return {k: getattr(i, k) for k, _ in IbfInfo._fields_}

    def words(self):
        """The filter's 64-bit words as dy_ibf_save() writes them (trailer included): (uint64 array, n_bits)."""
        n_bits = self.info()["n_bits"]
        out = np.empty(n_bits // 64, dtype=np.uint64)
        _check(self.lib.dy_ibf_read_words(self.h, 0, out.size, out.ctypes.data))
        return out, n_bits

    def query(self, reads, threshold, want_masks=True):
        """whichBins(fwd) | whichBins(revcomp) per read -> (n_reads, W) uint64 (or None)."""
        W = self.info()["words_per_pos"]
        masks = np.empty((reads.n, W), dtype=np.uint64) if want_masks else None
        _check(self.lib.dy_ibf_query(self.h, reads.h, threshold, masks.ctypes.data if want_masks else None))
        return masks

    def query_deferred(self, reads, threshold):
        """The sweep of the batch mapped NEXT: queued in chunks in front of the host's waits while the current batch is mapped."""
        _check(self.lib.dy_ibf_query_deferred(self.h, reads.h, threshold))

    def close(self):
        if self.h:
            self.lib.dy_ibf_free(self.h)
            self.h = None


def threshold(error_rate, kmer_size, read_len):
    return load_library().dy_ibf_threshold(np.float32(error_rate), kmer_size, read_len)


class Reads:
    @classmethod
    def from_fastq(cls, ctx, text, max_reads, at_eof=True):
        """The first max_reads four-line FASTQ records of `text` (bytes), parsed on the device.  Returns (Reads, bytes consumed)."""
        buf = np.frombuffer(text, dtype=np.uint8)
        h, used = _P(), C.c_uint64(0)
        _check(ctx.lib.dy_reads_from_fastq(ctx.h, buf.ctypes.data if buf.size else None, buf.size, max_reads, 1 if at_eof else 0, C.byref(h), C.byref(used)))
        return cls._from_handle(ctx, h), int(used.value)

    def total_bases(self):
        return int(self.lib.dy_reads_total_bases(self.h))

    @classmethod
    def _from_handle(cls, ctx, handle):
        self = cls.__new__(cls)
        self.ctx, self.lib, self.h = ctx, ctx.lib, handle
        self.n = ctx.lib.dy_reads_count(handle)
        return self

    def __init__(self, ctx, codes, offs):
        self.ctx, self.lib = ctx, ctx.lib
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        self.n = offs.size - 1
        h = _P()
        _check(self.lib.dy_reads_upload(ctx.h, codes.ctypes.data, offs.ctypes.data, self.n, C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            self.lib.dy_reads_free(self.h)
            self.h = None


class Index:
    def __init__(self, ctx, directory, n_bins, owned=None):
        """owned = None: every bin resident; else n_bins flags, FM structures of the flagged bins only (dy_index_load_sharded)."""
        self.ctx, self.lib = ctx, ctx.lib
        h = _P()
        if owned is None:
            _check(self.lib.dy_index_load(ctx.h, os.fsencode(directory), n_bins, C.byref(h)))
        else:
            owned = np.ascontiguousarray(owned, dtype=np.uint8)
            assert owned.size == n_bins
            _check(self.lib.dy_index_load_sharded(ctx.h, os.fsencode(directory), n_bins, owned.ctypes.data, C.byref(h)))
        self.h = h
        self.n_bins = n_bins

    def info(self):
        i = IndexInfo()
        _check(self.lib.dy_index_get_info(self.h, C.byref(i)))
        return {k: getattr(i, k) for k, _ in IndexInfo._fields_}

    def contig_names(self):
        return [self.lib.dy_index_contig_name(self.h, c).decode() for c in range(self.lib.dy_index_contigs(self.h))]

    def contig_lengths(self):
        return [self.lib.dy_index_contig_length(self.h, c) for c in range(self.lib.dy_index_contigs(self.h))]

    def close(self):
        if self.h:
            self.lib.dy_index_free(self.h)
            self.h = None


def suffix_array(ctx, text, sigma):
    """Suffix array of a uint32 text whose sequences end in distinct sentinel symbols (device sort, dy_suffix_array)."""
    text = np.ascontiguousarray(text, dtype=np.uint32)
    sa = np.empty(text.size, dtype=np.uint32)
    _check(ctx.lib.dy_suffix_array(ctx.h, text.ctypes.data, text.size, sigma, sa.ctypes.data))
    return sa


def default_params(**kw):
    p = Params()
    load_library().dy_params_default(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class Result:
    def __init__(self, ctx, handle, n_reads):
        self.ctx, self.lib, self.h, self.n = ctx, ctx.lib, handle, n_reads

    def fetch(self):
        _check(self.lib.dy_result_fetch(self.h))
        return self

    def reads(self):
        ptr = self.lib.dy_result_reads(self.h)
        if not ptr or not self.n:
            return np.zeros(0, dtype=READ_RESULT_DTYPE)
        buf = (C.c_char * (self.n * READ_RESULT_DTYPE.itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=READ_RESULT_DTYPE).copy()

    def matches(self):
        n = self.lib.dy_result_n_matches(self.h)
        ptr = self.lib.dy_result_matches(self.h)
        if not ptr or not n:
            return np.zeros(0, dtype=MATCH_DTYPE)
        buf = (C.c_char * (n * MATCH_DTYPE.itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=MATCH_DTYPE).copy()

    def cigars(self):
        rr = self.reads()
        n = int((rr["cigar_begin"] + rr["cigar_len"]).max()) if rr.size else 0
        ptr = self.lib.dy_result_cigars(self.h)
        if not ptr or not n:
            return np.zeros(0, dtype=np.uint32)
        buf = (C.c_char * (n * 4)).from_address(ptr)
        return np.frombuffer(buf, dtype=np.uint32).copy()

    def sam(self, reads, index, read_group="none", slot=0):
        """The batch's SAM records (single-end, reads from Reads.from_fastq), formatted on the device.  Returns (bytes, mapped reads)."""
        text, n, m = C.c_char_p(), C.c_uint64(0), C.c_uint64(0)
        ptr = C.cast(C.pointer(text), C.POINTER(C.c_char_p))
        _check(self.lib.dy_result_sam(self.h, reads.h, index.h, read_group.encode(), slot, ptr, C.byref(n), C.byref(m)))
        addr = C.cast(text, C.c_void_p).value
        return (C.string_at(addr, n.value) if n.value else b""), int(m.value)

    def close(self):
        if self.h:
            self.lib.dy_result_free(self.h)
            self.h = None


def map_reads(ctx, index, reads, params, asynchronous=False):
    """runDisMapper's per-batch body on the device (reference src/d_mapper.h:874-886)."""
    h = _P()
    fn = ctx.lib.dy_map_reads_async if asynchronous else ctx.lib.dy_map_reads
    _check(fn(ctx.h, index.h, reads.h, C.byref(params), C.byref(h)))
    return Result(ctx, h, reads.n)


def shard_c1_pack(ctx, reads, owner_masks, paired, headers_ptr):
    """[C1] Queues the packing of the reads selected for every owner rank.  owner_masks: (n_ranks, W) uint64; headers_ptr: device address of
    n_ranks x SEG_HEADER_BYTES (caller-owned).  Returns the device address of the context's packed send buffer."""
    owner_masks = np.ascontiguousarray(owner_masks, dtype=np.uint64)
    send = _P()
    _check(ctx.lib.dy_shard_c1_pack(ctx.h, reads.h, owner_masks.shape[0], owner_masks.ctypes.data, 1 if paired else 0, headers_ptr, C.byref(send)))
    return send.value or 0


def shard_c1_unpack(ctx, recv_ptr, recv_headers, n_bins, paired):
    """The owner-side batch from the received segments.  recv_headers: SEG_HEADER_DTYPE array (host), one row per source rank."""
    recv_headers = np.ascontiguousarray(recv_headers, dtype=SEG_HEADER_DTYPE)
    h = _P()
    _check(ctx.lib.dy_shard_c1_unpack(ctx.h, recv_ptr, recv_headers.size, recv_headers.ctypes.data, n_bins, 1 if paired else 0, C.byref(h)))
    return Reads._from_handle(ctx, h)


def map_bins(ctx, index, reads, params, want_counts=True):
    """K2 + K3 for the resident bins.  Returns (items per bin, number of raw matches left in the context) or None."""
    if not want_counts:
        _check(ctx.lib.dy_map_bins(ctx.h, index.h, reads.h, C.byref(params), None, None))
        return None
    loads = np.zeros(index.n_bins, np.uint32)
    n_raw = C.c_uint64(0)
    _check(ctx.lib.dy_map_bins(ctx.h, index.h, reads.h, C.byref(params), loads.ctypes.data, C.byref(n_raw)))
    return loads, int(n_raw.value)


def shard_c2_row_bytes(n_bins):
    return int(load_library().dy_shard_c2_row_bytes(n_bins))


def shard_c2_pack(ctx, rows_ptr):
    """[C2] Queues the packing of the raw matches (+ pairing records) of the last map_bins by home rank; rows_ptr: device address of
    n_ranks x shard_c2_row_bytes(n_bins).  Returns the device address of the context's packed send buffer."""
    send = _P()
    _check(ctx.lib.dy_shard_c2_pack(ctx.h, rows_ptr, C.byref(send)))
    return send.value or 0


def shard_finalize(ctx, index, reads, params, recv_ptr, recv_rows, n_ranks):
    """K4 on the home rank.  recv_rows: uint8 array (host) of n_ranks rows as received."""
    recv_rows = np.ascontiguousarray(recv_rows, dtype=np.uint8)
    h = _P()
    _check(ctx.lib.dy_shard_finalize(ctx.h, index.h, reads.h, C.byref(params), recv_ptr, n_ranks, recv_rows.ctypes.data, C.byref(h)))
    return Result(ctx, h, reads.n)


def cigar_string(ops):
    return "".join("%d%s" % (int(o) >> 4, "MID"[int(o) & 15]) for o in ops)

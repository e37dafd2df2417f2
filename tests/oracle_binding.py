"""ctypes binding of oracle/_build/liboracle.so — the CPU restatement used as the checker (test infrastructure only)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_build", "liboracle.so")

MATCH_DTYPE = np.dtype([("contig", "<u4"), ("begin", "<u4"), ("length", "<u2"), ("errors", "u1"), ("is_rev", "u1")])
READ_RESULT_DTYPE = np.dtype([("primary", MATCH_DTYPE), ("mapped", "<u4"), ("best_count", "<u4"), ("sub_count", "<u4"),
                              ("mapq", "<u4"), ("set_begin", "<u4"), ("cigar_begin", "<u4"), ("cigar_len", "<u4"),
                              ("paired", "<u4"), ("reserved", "<u4"), ("pair_prob", "<f8")])

_lib = None


def lib():
    global _lib
    if _lib is None:
        l = C.CDLL(LIB)
        P = C.c_void_p
        l.dyo_last_error.restype = C.c_char_p
        l.dyo_ibf_load.restype = P; l.dyo_ibf_load.argtypes = [C.c_char_p]
        l.dyo_kdx_load.restype = P; l.dyo_kdx_load.argtypes = [C.c_char_p]
        l.dyo_kdx_build.restype = C.c_int; l.dyo_kdx_build.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, P, P, C.c_uint32]
        l.dyo_align_match.restype = C.c_int
        l.dyo_align_match.argtypes = [P, C.c_int, P, C.c_int, C.c_int, P, C.c_int, P, P]
        l.dyo_ibf_free.argtypes = [P]
        l.dyo_ibf_info.argtypes = [P, P]
        l.dyo_ibf_query.restype = C.c_int; l.dyo_ibf_query.argtypes = [P, P, P, C.c_uint32, C.c_uint16, P]
        l.dyo_ibf_query_mt.restype = C.c_int; l.dyo_ibf_query_mt.argtypes = [P, P, P, C.c_uint32, C.c_uint16, P, C.c_uint]
        l.dyo_ibf_attach.restype = P; l.dyo_ibf_attach.argtypes = [P, C.c_uint64, C.c_int]
        l.dyo_ibf_build.restype = C.c_int
        l.dyo_ibf_build.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, P, P, C.c_uint32]
        l.dyo_index_load.restype = P; l.dyo_index_load.argtypes = [C.c_char_p, C.c_uint32]
        l.dyo_index_free.argtypes = [P]
        l.dyo_seed_occurrences.restype = C.c_uint64
        l.dyo_seed_occurrences.argtypes = [P, C.c_uint32, P, C.c_uint32, C.c_uint32, C.c_int, P, C.c_uint64]
        l.dyo_index_build.restype = C.c_int; l.dyo_index_build.argtypes = [C.c_char_p, P, P, C.c_uint32]
        l.dyo_map_batch.restype = P
        l.dyo_map_batch.argtypes = [P, P, P, P, C.c_uint32, C.c_float, C.c_float, C.c_int, C.c_uint, C.c_uint]
        l.dyo_map_batch_paired.restype = P
        l.dyo_map_batch_paired.argtypes = [P, P, P, P, C.c_uint32, C.c_float, C.c_float, C.c_int, C.c_uint, C.c_uint, C.c_uint, C.c_int]
        l.dyo_map_batch_paired_mt.restype = P
        l.dyo_map_batch_paired_mt.argtypes = [P, P, P, P, C.c_uint32, C.c_float, C.c_float, C.c_int, C.c_uint, C.c_uint, C.c_uint, C.c_int, C.c_uint, C.c_int]
        l.dyo_result_free.argtypes = [P]
        l.dyo_result_n_matches.restype = C.c_uint64; l.dyo_result_n_matches.argtypes = [P]
        l.dyo_result_n_cigars.restype = C.c_uint64; l.dyo_result_n_cigars.argtypes = [P]
        l.dyo_result_copy.argtypes = [P, P, P, P]
        l.dyo_result_stats.argtypes = [P, P]
        _lib = l
    return _lib


def _err():
    return RuntimeError(lib().dyo_last_error().decode())


class OracleIbf:
    def __init__(self, path, direct=False, words=None, n_bits=0):
        """direct=True: the k-mer direct-addressing filter of -ft kmer_direct (reference src/d_kdx_filter.h).
        words (uint64 array, trailer included) + n_bits: a filter already in memory instead of a file; the array is kept alive here."""
        if words is not None:
            assert words.dtype == np.uint64 and words.flags["C_CONTIGUOUS"] and words.size * 64 >= n_bits
            self._words = words
            self.h = lib().dyo_ibf_attach(words.ctypes.data, n_bits, 1 if direct else 0)
        else:
            self.h = (lib().dyo_kdx_load if direct else lib().dyo_ibf_load)(os.fsencode(path))
        if not self.h:
            raise _err()
        info = np.zeros(6, dtype=np.uint64)
        lib().dyo_ibf_info(self.h, info.ctypes.data)
        self.n_bits, self.n_bins, self.n_hashes, self.kmer_size, self.W, self.n_hash_pos = (int(x) for x in info)

    def query(self, codes, offs, threshold, threads=1):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n = offs.size - 1
        masks = np.zeros((n, self.W), dtype=np.uint64)
        if lib().dyo_ibf_query_mt(self.h, codes.ctypes.data, offs.ctypes.data, n, threshold, masks.ctypes.data, threads):
            raise _err()
        return masks

    def close(self):
        if self.h:
            lib().dyo_ibf_free(self.h)
            self.h = None


def build_filter(path, bins, hashes, k, bits, codes, offs):
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    offs = np.ascontiguousarray(offs, dtype=np.uint64)
    if lib().dyo_ibf_build(os.fsencode(path), bins, hashes, k, bits, codes.ctypes.data, offs.ctypes.data, offs.size - 1):
        raise _err()


def build_direct_filter(path, bins, k, codes, offs):
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    offs = np.ascontiguousarray(offs, dtype=np.uint64)
    if lib().dyo_kdx_build(os.fsencode(path), bins, k, codes.ctypes.data, offs.ctypes.data, offs.size - 1):
        raise _err()


def build_bin_index(prefix, codes, offs):
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    offs = np.ascontiguousarray(offs, dtype=np.uint64)
    if lib().dyo_index_build(os.fsencode(prefix), codes.ctypes.data, offs.ctypes.data, offs.size - 1):
        raise _err()


class OracleIndex:
    def __init__(self, directory, n_bins):
        self.h = lib().dyo_index_load(os.fsencode(directory), n_bins)
        if not self.h:
            raise _err()

    def close(self):
        if self.h:
            lib().dyo_index_free(self.h)
            self.h = None


def seed_occurrences(index, bin_no, seed, max_err, edit):
    """(contig, forward END position, errors) of every occurrence the oracle's seed search reports (Hamming or edit distance)."""
    seed = np.ascontiguousarray(seed, dtype=np.uint8)
    cap = 1 << 16
    out = np.zeros((cap, 3), dtype=np.uint32)
    n = int(lib().dyo_seed_occurrences(index.h, bin_no, seed.ctypes.data, seed.size, max_err, 1 if edit else 0, out.ctypes.data, cap))
    assert n <= cap
    return out[:n]


def map_batch(index, ibf, codes, offs, error_rate, strata_rate=0.0, sensitivity=1, hits_threshold=300, threads=4,
              paired=False, library_length=0, library_dev=0, verify=True, det_math=False):
    """Returns (reads, matches, cigars, stats) as numpy arrays laid out like the device results.
    paired=True: reads [0,n/2) are the first mates, [n/2,n) the second mates.  det_math=True evaluates the pairing weights with the
    device's fixed IEEE sequences instead of libm (the reference uses libm: src/bits_matches.h:898-966) — bit-exact pair_prob checks only."""
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    offs = np.ascontiguousarray(offs, dtype=np.uint64)
    n = offs.size - 1
    if paired:
        h = lib().dyo_map_batch_paired_mt(index.h, ibf.h if ibf is not None else None, codes.ctypes.data, offs.ctypes.data, n,
                                          np.float32(error_rate), np.float32(strata_rate), sensitivity, hits_threshold,
                                          library_length, library_dev, 1 if verify else 0, threads, 1 if det_math else 0)
    else:
        h = lib().dyo_map_batch(index.h, ibf.h if ibf is not None else None, codes.ctypes.data, offs.ctypes.data, n,
                                np.float32(error_rate), np.float32(strata_rate), sensitivity, hits_threshold, threads)
    if not h:
        raise _err()
    try:
        reads = np.zeros(n, dtype=READ_RESULT_DTYPE)
        matches = np.zeros(int(lib().dyo_result_n_matches(h)), dtype=MATCH_DTYPE)
        cigars = np.zeros(int(lib().dyo_result_n_cigars(h)), dtype=np.uint32)
        lib().dyo_result_copy(h, reads.ctypes.data, matches.ctypes.data if matches.size else None,
                              cigars.ctypes.data if cigars.size else None)
        stats = np.zeros(5, dtype=np.uint64)
        lib().dyo_result_stats(h, stats.ctypes.data)
    finally:
        lib().dyo_result_free(h)
    return reads, matches, cigars, {"items": int(stats[0]), "raw_matches": int(stats[1])}


def align_match(contig, read, errors):
    """oracle alignMatch (reference src/mapper_aligner.h:143-236): (cigar ops as (len << 4) | op, clip begin, clip end)."""
    contig = np.ascontiguousarray(contig, dtype=np.uint8); read = np.ascontiguousarray(read, dtype=np.uint8)
    out = np.zeros(64, np.uint32); cb = C.c_int(0); ce = C.c_int(0)
    k = lib().dyo_align_match(contig.ctypes.data, contig.size, read.ctypes.data, read.size, errors, out.ctypes.data, 64, C.byref(cb), C.byref(ce))
    return out[:k].tolist(), cb.value, ce.value

"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same inputs, against the committed
gold fixtures, and through size-independent properties at larger sizes.  Bit-exact everywhere (integer/byte work)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import bam_to_sam_lines, GOLD_INDEX, GENOMES_FA, READS_FA, GOLDEN, read_fasta_codes, sam_records

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dy(product_build):
    import dream_yara_amd as m
    if m.device_count() == 0:
        pytest.fail("no HIP device: the gpu-marked tests must run on the GPU box")
    return m


@pytest.fixture(scope="module")
def ctx(dy):
    c = dy.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ob(oracle_build):
    import oracle_binding
    return oracle_binding


def assert_results_equal(dev, ora):
    d_reads, d_matches, d_cigars = dev
    o_reads, o_matches, o_cigars = ora
    assert d_reads.size == o_reads.size
    for field in ("mapped", "best_count", "sub_count", "mapq", "cigar_len"):
        bad = np.nonzero(d_reads[field] != o_reads[field])[0]
        assert bad.size == 0, (field, bad[:10], d_reads[bad[:3]], o_reads[bad[:3]])
    m = o_reads["mapped"] == 1
    for field in ("contig", "begin", "length", "errors", "is_rev"):
        bad = np.nonzero(d_reads["primary"][field][m] != o_reads["primary"][field][m])[0]
        assert bad.size == 0, (field, bad[:10])
    # matchesSet and CIGARs, read by read (set_begin/cigar_begin are running offsets in both)
    assert np.array_equal(d_reads["set_begin"], o_reads["set_begin"])
    assert d_matches.size == o_matches.size
    assert d_matches.tobytes() == o_matches.tobytes()
    assert np.array_equal(d_reads["cigar_begin"], o_reads["cigar_begin"])
    assert np.array_equal(d_cigars, o_cigars)


def assert_pairing_equal(dev_reads, o_libm, o_det):
    """Paired-end fields: the oracle evaluates the pairing weights with libm like the reference (src/bits_matches.h:898-966); the
    device uses fixed IEEE sequences (dy_detmath.h).  Decisions must agree with the libm oracle, the probability to 1e-12 relative;
    against the oracle run with the same sequences (det_math) the doubles are bit-identical."""
    assert np.array_equal(dev_reads["paired"], o_libm["paired"])
    np.testing.assert_allclose(dev_reads["pair_prob"], o_libm["pair_prob"], rtol=1e-12, atol=0)
    assert np.array_equal(dev_reads["pair_prob"], o_det["pair_prob"])


def device_map(dy, ctx, index_dir, n_bins, filter_path, codes, offs, error_rate, strata_rate=0.0, sensitivity=1, read_len_for_thr=None):
    idx = dy.Index(ctx, index_dir, n_bins)
    reads = dy.Reads(ctx, codes, offs)
    params = dy.default_params(error_rate=error_rate, strata_rate=strata_rate, sensitivity=sensitivity)
    ibf = None
    if filter_path:
        ibf = dy.Ibf.load(ctx, filter_path)
        n = offs.size - 1
        avg = int(int(offs[-1]) * 2 // (n * 2))
        thr = dy.threshold(error_rate, ibf.info()["kmer_size"], avg)
        ibf.query(reads, thr, want_masks=False)
    else:
        params.use_filter = 0
    res = dy.map_reads(ctx, idx, reads, params)
    out = (res.reads(), res.matches(), res.cigars())
    stats = ctx.stats()
    res.close(); reads.close(); idx.close()
    if ibf:
        ibf.close()
    return out, stats


# ---------------------------------------------------------------------------------------------- K1: IBF sweep
def test_ibf_masks_match_oracle_on_gold_reads(dy, ctx, ob, small_filter):
    _, codes, offs = read_fasta_codes(READS_FA)
    oibf = ob.OracleIbf(small_filter)
    ibf = dy.Ibf.load(ctx, small_filter)
    info = ibf.info()
    assert (info["n_bins"], info["n_hashes"], info["kmer_size"], info["words_per_pos"]) == (64, 2, 19, 1)
    reads = dy.Reads(ctx, codes, offs)
    for thr in (25, 1, 0, 60):
        got = ibf.query(reads, thr)
        want = oibf.query(codes, offs, thr)
        assert np.array_equal(got, want), thr
    reads.close(); ibf.close(); oibf.close()


@pytest.mark.parametrize("bins,hashes,k,bits", [(200, 3, 18, 1 << 22), (64, 2, 14, (1 << 20) + 64 * 37), (130, 5, 32, 1 << 23), (1024, 3, 20, 1 << 24)])
def test_ibf_edge_cases(dy, ctx, ob, tmp_path, bins, hashes, k, bits):
    rng = np.random.default_rng(bins * 131 + k)
    genomes = [rng.integers(0, 4, 3000, dtype=np.uint8) for _ in range(bins)]
    g_codes = np.concatenate(genomes)
    g_offs = np.arange(bins + 1, dtype=np.uint64) * np.uint64(3000)
    path = str(tmp_path / "f.filter")
    ob.build_filter(path, bins, hashes, k, bits, g_codes, g_offs)
    # ragged reads: empty, shorter than k, exactly k, N first / last / inside, long
    lens = [0, 1, k - 1, k, k + 1, 50, 64, 65, 100, 127, 128, 129, 150, 250, 301, 1000]
    reads = []
    for i, L in enumerate(lens * 6):
        b = int(rng.integers(0, bins))
        s = int(rng.integers(0, 3000 - min(L, 2999)))
        r = genomes[b][s:s + L].copy() if L <= 3000 - s else rng.integers(0, 4, L, dtype=np.uint8)
        if r.size and i % 4 == 1: r[0] = 4
        if r.size and i % 4 == 2: r[-1] = 4
        if r.size > 10 and i % 4 == 3: r[rng.integers(0, r.size, 3)] = 4
        if i % 5 == 0 and r.size: r = r[::-1].copy(); r[r < 4] = 3 - r[r < 4]
        reads.append(r)
    codes = np.concatenate(reads)
    offs = np.zeros(len(reads) + 1, np.uint64); offs[1:] = np.cumsum([r.size for r in reads])
    oibf = ob.OracleIbf(path)
    ibf = dy.Ibf.load(ctx, path)
    dreads = dy.Reads(ctx, codes, offs)
    for thr in (0, 1, 5, 40):
        assert np.array_equal(ibf.query(dreads, thr), oibf.query(codes, offs, thr)), thr
    dreads.close(); ibf.close(); oibf.close()


@pytest.mark.parametrize("bins,bits,k", [(64, 1 << 24, 14), (37, 1 << 20, 12), (64, 1 << 18, 10)])
def test_ibf_adaptive_tail_matches_oracle(dy, ctx, ob, tmp_path, bins, bits, k):
    """64 bins or fewer and at most 127 k-mers per read: k_ibf_query_two fetches the second chunk of k-mers only as far as the
    decision needs.  Dense filters (false candidates that survive the first 64 k-mers, some until the very end), mixed read
    lengths around the 64-k-mer boundary, every threshold from 1 to beyond the k-mer count."""
    rng = np.random.default_rng(bins * 7 + k)
    genomes = [rng.integers(0, 4, 30000, dtype=np.uint8) for _ in range(bins)]
    goffs = np.arange(bins + 1, dtype=np.uint64) * np.uint64(30000)
    path = str(tmp_path / "f.filter")
    ob.build_filter(path, bins, 3, k, bits, np.concatenate(genomes), goffs)
    reads = []
    for i in range(800):
        b = int(rng.integers(0, bins)); L = int(rng.integers(k, k + 126)); s = int(rng.integers(0, 30000 - L))
        r = genomes[b][s:s + L].copy()
        for _ in range(int(rng.integers(0, 8))): r[int(rng.integers(0, L))] = rng.integers(0, 4)
        if i % 4 == 0: r[L // 2:] = rng.integers(0, 4, L - L // 2)          # only the first half matches: weak bins that die late or survive
        if i % 11 == 0: r[int(rng.integers(0, L))] = 4
        if i % 2: r = np.where(r[::-1] < 4, 3 - r[::-1], 4).astype(np.uint8)
        reads.append(r.astype(np.uint8))
    codes = np.concatenate(reads); offs = np.zeros(len(reads) + 1, np.uint64); offs[1:] = np.cumsum([r.size for r in reads])
    ibf = dy.Ibf.load(ctx, path); oibf = ob.OracleIbf(path); rd = dy.Reads(ctx, codes, offs)
    for thr in (1, 2, 5, 10, 20, 29, 40, 64, 65, 90, 127, 200):
        assert np.array_equal(ibf.query(rd, thr), oibf.query(codes, offs, thr)), thr
    rd.close(); ibf.close(); oibf.close()


@pytest.mark.parametrize("bins,bits,k", [(1024, 1 << 27, 12), (192, 1 << 24, 11), (256, 1 << 21, 10)])
def test_ibf_low_thresholds_many_false_candidates(dy, ctx, ob, tmp_path, bins, bits, k):
    """More than 64 bins, reads under 256 k-mers and thresholds far below the k-mer count: the regime of the sparse verification
    in k_ibf_query_wide (dozens of false candidate bins per read; the smallest filter overflows the pair list and takes the
    dense sweep).  Masks equal the oracle's for every threshold."""
    rng = np.random.default_rng(bins + k)
    genomes = [rng.integers(0, 4, 20000, dtype=np.uint8) for _ in range(bins)]
    goffs = np.arange(bins + 1, dtype=np.uint64) * np.uint64(20000)
    path = str(tmp_path / "f.filter")
    ob.build_filter(path, bins, 3, k, bits, np.concatenate(genomes), goffs)
    reads = []
    for i in range(600):
        b = int(rng.integers(0, bins)); L = int(rng.integers(60, 200)); s = int(rng.integers(0, 20000 - L))
        r = genomes[b][s:s + L].copy()
        for _ in range(int(rng.integers(0, 6))): r[int(rng.integers(0, L))] = rng.integers(0, 4)
        if i % 9 == 0: r[int(rng.integers(0, L))] = 4
        if i % 2: r = np.where(r[::-1] < 4, 3 - r[::-1], 4).astype(np.uint8)
        reads.append(r.astype(np.uint8))
    codes = np.concatenate(reads); offs = np.zeros(len(reads) + 1, np.uint64); offs[1:] = np.cumsum([r.size for r in reads])
    ibf = dy.Ibf.load(ctx, path); oibf = ob.OracleIbf(path); rd = dy.Reads(ctx, codes, offs)
    sel = []
    for thr in (1, 3, 7, 15, 40):
        m = ibf.query(rd, thr)
        assert np.array_equal(m, oibf.query(codes, offs, thr)), thr
        sel.append(float(np.unpackbits(m.view(np.uint8)).sum()) / len(reads))
    assert sel[0] >= sel[-1] and sel[0] > 2      # low thresholds select (many) more bins than the reads' own; the smallest filter is saturated
    rd.close(); ibf.close(); oibf.close()


@pytest.mark.parametrize("kernel", [1, 2, 3])
@pytest.mark.parametrize("bins,bits,k,hashes", [(1024, 1 << 27, 12, 3), (192, 1 << 24, 11, 3), (256, 1 << 21, 10, 3), (130, 1 << 23, 16, 5), (200, 1 << 25, 14, 2), (256, 1 << 24, 13, 1)])
def test_ibf_every_wide_sweep_matches_oracle(dy, ob, tmp_path, kernel, bins, bits, k, hashes):
    """More than 64 bins: each of the three sweeps forced in turn (dy_ctx_set_tuning ibf_kernel: 1 = k_ibf_query_lazy, 2 = k_ibf_query_eager2,
    3 = k_ibf_query_wide; 0 = automatic choice from the filter fill is what every other test runs) over sparse, dense and saturated filters
    (the smallest overflows eager2's list and takes its plain sweep), 1 / 2 / 3 / 5 hash functions, reads from 0 to 254 k-mers with Ns,
    every threshold from 1 to beyond the k-mer count, and a forced first-check position (ibf_slack) that makes phase A resume."""
    rng = np.random.default_rng(bins + k + hashes)
    genomes = [rng.integers(0, 4, 20000, dtype=np.uint8) for _ in range(bins)]
    goffs = np.arange(bins + 1, dtype=np.uint64) * np.uint64(20000)
    path = str(tmp_path / "f.filter")
    ob.build_filter(path, bins, hashes, k, bits, np.concatenate(genomes), goffs)
    reads = []
    for i in range(500):
        b = int(rng.integers(0, bins)); L = int(rng.integers(0, k + 254)); s = int(rng.integers(0, 20000 - L))
        r = genomes[b][s:s + L].copy()
        for _ in range(int(rng.integers(0, 6))):
            if L: r[int(rng.integers(0, L))] = rng.integers(0, 4)
        if i % 4 == 0 and L > 40: r[L // 2:] = rng.integers(0, 4, L - L // 2)      # half of the read matches: bins that die late or survive narrowly
        if i % 9 == 0 and L: r[int(rng.integers(0, L))] = 4
        if i % 13 == 0 and L: r[0] = 4
        if i % 2: r = np.where(r[::-1] < 4, 3 - r[::-1], 4).astype(np.uint8)
        reads.append(r.astype(np.uint8))
    codes = np.concatenate(reads); offs = np.zeros(len(reads) + 1, np.uint64); offs[1:] = np.cumsum([r.size for r in reads])
    oibf = ob.OracleIbf(path)
    want = {thr: oibf.query(codes, offs, thr, threads=8) for thr in (0, 1, 3, 7, 15, 29, 40, 90, 200, 254, 300)}
    oibf.close()
    for slack in (0, 1):
        tctx = dy.Context(0)
        try:
            tctx.set_tuning("ibf_kernel", kernel)
            tctx.set_tuning("ibf_slack", slack)
            ibf = dy.Ibf.load(tctx, path); rd = dy.Reads(tctx, codes, offs)
            for thr, m in want.items():
                assert np.array_equal(ibf.query(rd, thr), m), (thr, slack)
            rd.close(); ibf.close()
        finally:
            tctx.close()


def test_large_batch_upload_equals_small_batches(dy, ctx, ob, tmp_path):
    """dy_reads_upload stages batches above 32 MB through pinned memory on several threads (8 MB chunks); the bin masks of one
    40 MB batch must equal those of the same reads uploaded in small batches (plain copy path), and a sample must equal the oracle."""
    rng = np.random.default_rng(2024)
    bins, k = 8, 14
    genomes = [rng.integers(0, 4, 30000, dtype=np.uint8) for _ in range(bins)]
    goffs = np.arange(bins + 1, dtype=np.uint64) * np.uint64(30000)
    path = str(tmp_path / "f.filter")
    ob.build_filter(path, bins, 3, k, 1 << 22, np.concatenate(genomes), goffs)
    n, L = 400000, 100
    b = rng.integers(0, bins, n); s = rng.integers(0, 30000 - L, n)
    big = np.stack(genomes)                                  # (bins, 30000)
    codes2d = big[b[:, None], s[:, None] + np.arange(L)[None, :]].astype(np.uint8)
    flip = rng.random(n) < 0.5
    codes2d[flip] = 3 - codes2d[flip][:, ::-1]
    codes2d[rng.random((n, L)) < 0.01] = 2
    codes = np.ascontiguousarray(codes2d).reshape(-1)
    offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    ibf = dy.Ibf.load(ctx, path)
    thr = dy.threshold(np.float32(0.03), k, L)
    rd = dy.Reads(ctx, codes, offs)
    whole = ibf.query(rd, thr)
    rd.close()
    parts = []
    for lo in range(0, n, 100000):
        hi = lo + 100000
        rp = dy.Reads(ctx, codes[lo * L:hi * L], offs[lo:hi + 1] - offs[lo])
        parts.append(ibf.query(rp, thr))
        rp.close()
    assert np.array_equal(whole, np.concatenate(parts))
    oibf = ob.OracleIbf(path)
    sample = slice(390000, 400000)                           # the last chunks of the staged copy
    assert np.array_equal(whole[sample], oibf.query(codes[390000 * L:], offs[390000:] - offs[390000], thr))
    assert float((whole != 0).mean()) > 0.9
    ibf.close(); oibf.close()


def test_ibf_beyond_4gib_matches_oracle(dy, ctx, ob, tmp_path):
    """256 bins (W = 4) over a 5 GiB filter: word indices no longer fit 32 bits.  Filter built on the device, saved in
    the reference's file format, re-read by the oracle; masks must agree for reads sampled from the bins."""
    rng = np.random.default_rng(41)
    bins, k, h = 256, 18, 3
    bits = 5 * (1 << 33)
    genomes = [rng.integers(0, 4, 20000, dtype=np.uint8) for _ in range(bins)]
    f = dy.Ibf.create(ctx, bins, h, k, bits + 256)
    for b, g in enumerate(genomes):
        f.add(g, b)
    path = str(tmp_path / "big.filter")
    f.save(path)
    reads = []
    for i in range(3000):
        b = int(rng.integers(0, bins)); s = int(rng.integers(0, 20000 - 100))
        r = genomes[b][s:s + 100].copy()
        for _ in range(int(rng.integers(0, 4))):
            q = int(rng.integers(0, 100)); r[q] = (r[q] + 1) & 3
        if i % 2: r = (3 - r[::-1]).astype(np.uint8)
        reads.append(r)
    codes = np.concatenate(reads); offs = np.arange(len(reads) + 1, dtype=np.uint64) * np.uint64(100)
    dreads = dy.Reads(ctx, codes, offs)
    thr = dy.threshold(np.float32(0.03), k, 100)
    got = f.query(dreads, thr)
    oibf = ob.OracleIbf(path)
    assert (oibf.n_bins, oibf.W, oibf.n_hash_pos) == (bins, 4, bits // 256)
    want = oibf.query(codes, offs, thr)
    assert np.array_equal(got, want)
    assert got.any(axis=1).all()          # every read selects at least its own bin
    dreads.close(); f.close(); oibf.close()
    os.remove(path)


def test_ibf_built_on_gpu_equals_oracle_file(dy, ctx, ob, tmp_path):
    _, codes, offs = read_fasta_codes(GENOMES_FA)
    a, b = str(tmp_path / "gpu.filter"), str(tmp_path / "cpu.filter")
    ob.build_filter(b, 64, 2, 19, 1 << 26, codes, offs)
    f = dy.Ibf.create(ctx, 64, 2, 19, (1 << 26) + 256)
    for i in range(64):
        f.add(codes[int(offs[i]):int(offs[i + 1])], i)
    f.save(a)
    f.close()
    assert open(a, "rb").read() == open(b, "rb").read()


# ---------------------------------------------------------------------------------------------- K2/K3 + aggregation
@pytest.mark.parametrize("strata", [0.0, 0.10])
def test_map_matches_oracle_on_gold_data(dy, ctx, ob, small_filter, strata):
    _, codes, offs = read_fasta_codes(READS_FA)
    er = np.float32(3 / 100.0)
    dev, stats = device_map(dy, ctx, GOLD_INDEX, 64, small_filter, codes, offs, er, np.float32(strata))
    oidx = ob.OracleIndex(GOLD_INDEX, 64)
    oibf = ob.OracleIbf(small_filter)
    o_reads, o_matches, o_cigars, ostats = ob.map_batch(oidx, oibf, codes, offs, er, np.float32(strata))
    assert stats["items"] == ostats["items"]
    assert stats["raw_matches"] == ostats["raw_matches"]
    assert_results_equal(dev, (o_reads, o_matches, o_cigars))
    assert int(dev[0]["mapped"].sum()) == 6394
    oidx.close(); oibf.close()


def _gold_reads_as_fastq(rng, n=None):
    """The gold reads as four-line FASTQ with what a parser has to normalise: names with comments and /1 suffixes, lower-case bases, IUPAC
    codes and Ns, qualities up to '~' (clamped to 6 bits by Dna5Q) and under Ns."""
    names, codes, offs = read_fasta_codes(READS_FA)
    n = n or len(names)
    recs = []
    for i in range(n):
        seq = bytearray(np.frombuffer(b"ACGTN", np.uint8)[codes[int(offs[i]):int(offs[i + 1])]].tobytes())
        qual = bytearray(rng.integers(33, 127, len(seq), dtype=np.uint8).tobytes())
        if i % 7 == 0:
            seq = bytearray(bytes(seq).lower())
        if i % 11 == 0:
            seq[int(rng.integers(0, len(seq)))] = ord("R")            # IUPAC code: stored as N
        name = names[i] + ("/1" if i % 3 == 0 else "") + (" a comment" if i % 5 == 0 else "") + ("\twith a tab" if i % 13 == 0 else "")
        recs.append(b"@" + name.encode() + b"\n" + bytes(seq) + b"\n+" + (name.encode() if i % 17 == 0 else b"") + b"\n" + bytes(qual) + b"\n")
    return recs


def test_device_fastq_parse_and_sam_format_equal_the_host_path(product_build, dy, ob, small_filter, tmp_path):
    """dy_reads_from_fastq + dy_result_sam (FASTQ text in, SAM text out, both on the device) against the tool's host path (ReadsFile parser +
    samWriteRecord) on the same file: byte-identical records; the batch itself (codes, lengths) equals the host parser's."""
    rng = np.random.default_rng(11)
    recs = _gold_reads_as_fastq(rng)
    text = b"".join(recs)
    fq = tmp_path / "reads.fq"
    fq.write_bytes(text)
    exe = os.path.join(product_build, "dream_yara_mapper")
    host_sam = str(tmp_path / "host.sam")
    subprocess.check_call([exe, "-fi", small_filter, GOLD_INDEX + "/", str(fq), "-o", host_sam, "-e", "3", "-s", "10", "-rb", "1000", "--host-io"])
    want = [l for l in open(host_sam, "rb").read().split(b"\n") if l and not l.startswith(b"@")]
    c = dy.Context(0)
    try:
        idx = dy.Index(c, GOLD_INDEX, 64)
        ibf = dy.Ibf.load(c, small_filter)
        er = np.float32(0.03)
        params = dy.default_params(error_rate=er, strata_rate=np.float32(0.10))
        got, at, sizes = [], 0, []
        while at < len(text):
            window = text[at: at + 300_000]                              # about 1 300 records; the batch takes 1 000 of them
            reads, used = dy.Reads.from_fastq(c, window, 1000, at_eof=at + len(window) >= len(text))
            assert used > 0 and reads.n == min(1000, len(recs) - len(sizes) * 1000 if False else reads.n)
            avg = reads.total_bases() * 2 // (reads.n * 2)
            ibf.query(reads, dy.threshold(er, 19, avg), want_masks=False)
            res = dy.map_reads(c, idx, reads, params, asynchronous=True)
            sam, n_mapped = res.sam(reads, idx, "none")
            got += [l for l in sam.split(b"\n") if l]
            sizes.append(reads.n)
            res.close(); reads.close()
            at += used
        assert sizes == [1000] * 6 + [400]
        assert len(got) == len(want) == 6400
        bad = [i for i in range(6400) if got[i] != want[i]]
        assert not bad, (bad[:5], got[bad[0]], want[bad[0]])
        # the tool takes the device path by itself for such input: the same file, and also with a first window too small for a batch (long
        # comments: the window is read again, larger), and with CR line ends (the device parser refuses, the host parser takes over)
        dev_sam = str(tmp_path / "dev.sam")
        subprocess.check_call([exe, "-fi", small_filter, GOLD_INDEX + "/", str(fq), "-o", dev_sam, "-e", "3", "-s", "10", "-rb", "1000"])
        assert [l for l in open(dev_sam, "rb").read().split(b"\n") if l and not l.startswith(b"@")] == want
        for tag, variant in (("long", b"".join(r.replace(b"\n", b" " + b"x" * 420 + b"\n", 1) for r in recs)), ("crlf", text.replace(b"\n", b"\r\n"))):
            f2 = tmp_path / (tag + ".fq")
            f2.write_bytes(variant)
            outs = []
            for extra in ([], ["--host-io"]):
                o = str(tmp_path / (tag + "".join(extra) + ".sam"))
                subprocess.check_call([exe, "-fi", small_filter, GOLD_INDEX + "/", str(f2), "-o", o, "-e", "3", "-s", "10", "-rb", "1000", "-sk"] + extra)
                outs.append(open(o, "rb").read())
            assert outs[0] == outs[1] and outs[0].count(b"\n") == 6400, tag                 # -sk: no header lines
        # the last line without its newline, and a window that ends inside a record
        reads, used = dy.Reads.from_fastq(c, text[:-1], 7000, at_eof=True)
        assert reads.n == 6400 and used == len(text) - 1
        reads.close()
        cut = len(b"".join(recs[:10])) + 30
        reads, used = dy.Reads.from_fastq(c, text[:cut], 1000, at_eof=False)
        assert reads.n == 10 and used == cut - 30
        reads.close()
        # what the device parser leaves to the host: CR line ends, blank lines, FASTA, wrapped sequences, a short quality line
        for badtext in (text[:5000].replace(b"\n", b"\r\n"), recs[0] + b"\n" + recs[1], open(READS_FA, "rb").read()[:4000],
                        recs[0][:60] + b"\n" + recs[0][60:], recs[0][:-5] + b"\n"):
            with pytest.raises(dy.DyError) as e:
                dy.Reads.from_fastq(c, badtext, 1000, at_eof=True)
            assert e.value.code == -6 and "four-line FASTQ" in str(e.value)
        ibf.close(); idx.close()
    finally:
        c.close()


def test_owner_side_batch_ignores_mask_bits_beyond_the_bins(dy, ob, small_filter, tmp_path):
    """A C1 segment as a peer rank would send it, built by hand, with stray bits above the last bin in every read's mask word (6 bins: bits 6..63)
    and the bit of a bin this rank does not own: the owner-side work list counts and emits the same (read, bin) pairs (the count pass masks the
    last word like the emit pass does, and dy_map_bins ANDs the masks with the resident bins), so loads, raw matches and final results equal
    those of clean masks.  One rank, no transport: the packed send buffer is handed back as the receive buffer."""
    import ctypes as C
    rng = np.random.default_rng(5)
    n_bins, k = 6, 16
    genomes, fam = _repeat_genomes(rng, n_bins, 30000)
    idx_dir = str(tmp_path / "idx"); os.makedirs(idx_dir)
    for b, g in enumerate(genomes):
        ob.build_bin_index(os.path.join(idx_dir, str(b)), g, np.array([0, g.size], np.uint64))
    goffs = np.zeros(n_bins + 1, np.uint64); goffs[1:] = np.cumsum([g.size for g in genomes])
    fpath = str(tmp_path / "f.filter")
    ob.build_filter(fpath, n_bins, 3, k, 1 << 24, np.concatenate(genomes), goffs)
    codes, offs = _sample_reads(rng, genomes, fam, 500, 100, 3)
    n = offs.size - 1
    c = dy.Context(0)
    try:
        ibf = dy.Ibf.load(c, fpath)
        owned = np.array([1, 1, 1, 1, 1, 0], np.uint8)                           # bin 5 lives on another rank
        idx = dy.Index(c, idx_dir, n_bins, owned=owned)
        reads = dy.Reads(c, codes, offs)
        er = np.float32(0.03)
        masks = ibf.query(reads, dy.threshold(er, k, 100))[:, 0]
        params = dy.default_params(error_rate=er)
        row_bytes = dy.shard_c2_row_bytes(n_bins)

        def run(m):
            lens = np.diff(offs).astype(np.uint32)
            body = np.arange(n, dtype=np.uint32).tobytes() + lens.tobytes()
            packed = (codes[0::2] | (codes[1::2] << 4)).astype(np.uint8)       # 100-base reads: two codes per byte, the first in the low half
            body += b"\0" * (-len(body) % 16) + m.astype("<u8").tobytes() + packed.tobytes()
            body += b"\0" * (-len(body) % 16)
            hdr = np.zeros(1, dy.SEG_HEADER_DTYPE)
            hdr["n_reads"], hdr["n_bases"], hdr["bytes"], hdr["aux"] = n, codes.size, len(body), 100 | (100 << 32)
            seg = c.host_buffer(0, len(body))                                    # pinned: the kernels read it where it lies
            C.memmove(seg, body, len(body))
            sub = dy.shard_c1_unpack(c, seg, hdr, n_bins, False)
            loads, n_raw = dy.map_bins(c, idx, sub, params)
            rows = c.host_buffer(1, row_bytes)
            send = dy.shard_c2_pack(c, rows)
            c.sync(); sub.close()
            rows_host = np.frombuffer(C.string_at(rows, row_bytes), dtype=np.uint8).copy()
            assert int(rows_host[:8].view("<u8")[0]) == n_raw
            res = dy.shard_finalize(c, idx, reads, params, send, rows_host, 1)
            res.fetch()
            out = (res.reads(), res.matches(), res.cigars())
            res.close()
            return loads, n_raw, out
        clean = masks & np.uint64(0b011111)
        dirty = masks | np.uint64(0xFFFFFFFFFFFFFFC0) | np.uint64(0b100000)
        l0, r0, o0 = run(clean)
        l1, r1, o1 = run(dirty)
        assert l0[5] == 0 and np.array_equal(l0, l1) and int(l0.sum()) == int(np.unpackbits(clean.view(np.uint8)).sum())
        assert r0 == r1 and r0 > 300
        assert_results_equal(o0, o1)
        assert int(o0[0]["mapped"].sum()) > 300
        reads.close(); idx.close(); ibf.close()
    finally:
        c.close()


def test_batches_queued_whole_equal_the_waiting_way(dy, ob, small_filter):
    """Tuning key async_map: after a context's first batch, single-end batches are queued without the host waiting for the work list's size or
    the pool counter (bounds from the previous batch, real sizes on the device, flags fetched at the end).  Same results as the oracle — also
    when a bound does not hold (a batch four times the previous one: the work list does not fit, the batch is mapped again the waiting way)."""
    c = dy.Context(0)
    try:
        c.set_tuning("async_map", 1)
        _, codes, offs = read_fasta_codes(READS_FA)
        er = np.float32(3 / 100.0)
        idx = dy.Index(c, GOLD_INDEX, 64)
        ibf = dy.Ibf.load(c, small_filter)
        thr = dy.threshold(er, 19, 100)
        params = dy.default_params(error_rate=er, strata_rate=np.float32(0.10))
        oidx = ob.OracleIndex(GOLD_INDEX, 64); oibf = ob.OracleIbf(small_filter)
        retries = []
        # first: waiting; second: queued whole; third: unfiltered (64 items per read instead of one: the work list does not fit); fourth: queued whole
        for lo, hi, filtered in ((0, 1500, True), (1500, 3000, True), (3000, 4000, False), (4000, 6400, True)):
            cc, oo = codes[int(offs[lo]): int(offs[hi])], (offs[lo: hi + 1] - offs[lo]).astype(np.uint64)
            reads = dy.Reads(c, cc, oo)
            params.use_filter = 1 if filtered else 0
            if filtered:
                ibf.query(reads, thr, want_masks=False)
            res = dy.map_reads(c, idx, reads, params)
            got = (res.reads(), res.matches(), res.cigars()); st = c.stats()
            res.close(); reads.close()
            o = ob.map_batch(oidx, oibf if filtered else None, cc, oo, er, np.float32(0.10))
            assert_results_equal(got, o[:3])
            assert st["items"] == o[3]["items"] and st["raw_matches"] == o[3]["raw_matches"]
            retries.append(st["pool_retries"])
        assert retries[1:] == [0, 1, 0]                       # (the first batch may grow the pool the waiting way)
        oidx.close(); oibf.close(); ibf.close(); idx.close()
    finally:
        c.close()


@pytest.mark.parametrize("chunk", [0, 1024, 16384])
def test_deferred_sweep_gives_the_same_masks_and_results(dy, ob, small_filter, chunk):
    """dy_ibf_query_deferred: the sweep of the batch mapped next is queued chunk by chunk in front of the host's waits of the current
    mapping, the rest when its masks are first needed.  Two batches alternate three times; every result equals the plain sequence
    (sweep, then map) and the oracle.  1024 reads per chunk: more chunks than the timing slots hold, and the sweep finishes early."""
    c = dy.Context(0)
    try:
        c.set_tuning("sweep_chunk", chunk)
        _, codes, offs = read_fasta_codes(READS_FA)
        er = np.float32(3 / 100.0)
        cutA, cutB = 3500, 6400
        a = (codes[: int(offs[cutA])], offs[: cutA + 1].copy())
        b = (codes[int(offs[cutA]): int(offs[cutB])], (offs[cutA: cutB + 1] - offs[cutA]).astype(np.uint64))
        idx = dy.Index(c, GOLD_INDEX, 64)
        ibf = dy.Ibf.load(c, small_filter)
        thr = dy.threshold(er, 19, 100)
        params = dy.default_params(error_rate=er)
        ra, rb = dy.Reads(c, *a), dy.Reads(c, *b)
        want = {}
        for name, r in (("a", ra), ("b", rb)):
            masks = ibf.query(r, thr)
            res = dy.map_reads(c, idx, r, params)
            want[name] = (masks, res.reads(), res.matches(), res.cigars()); res.close()
        ibf.query(ra, thr, want_masks=False)
        for it in range(3):
            for cur, nxt, name in ((ra, rb, "a"), (rb, ra, "b")):
                ibf.query_deferred(nxt, thr)                     # registered before the current batch is mapped: its chunks ride on this mapping's waits
                res = dy.map_reads(c, idx, cur, params)
                got = (res.reads(), res.matches(), res.cigars()); res.close()
                assert_results_equal(got, want[name][1:])
        assert c.stats()["ms_ibf"] > 0
        # a deferred sweep whose masks are read back directly, and one that is superseded by a plain sweep
        ibf.query_deferred(rb, thr)
        assert np.array_equal(ibf.query(rb, thr), want["b"][0])
        ibf.query_deferred(ra, thr)
        res = dy.map_reads(c, idx, ra, params)                   # needs the masks at once: the whole sweep is flushed in front of it
        assert_results_equal((res.reads(), res.matches(), res.cigars()), want["a"][1:]); res.close()
        oidx = ob.OracleIndex(GOLD_INDEX, 64); oibf = ob.OracleIbf(small_filter)
        o = ob.map_batch(oidx, oibf, a[0], a[1], er)
        assert_results_equal(want["a"][1:], o[:3])
        oidx.close(); oibf.close()
        ra.close(); rb.close(); ibf.close(); idx.close()
    finally:
        c.close()


@pytest.mark.parametrize("tuning", [{"sparse_sa": 1}, {"prefix_q": 0}, {"no_isa": 1}, {"prefix_q": 6}, {"finalize_block_min": 3, "finalize_no_keys": 1},
                                    {"ibf_full_tail": 1}, {"items_sorted": 0}, {"async_map": 1}, {"prefix_wide": 0}, {"prefix_q": 6, "prefix_wide": 0}, {"extend_mode": 0}, {"extend_mode": 1}, {"extend_mode": 2}, {"markers": 1}])
def test_map_without_the_device_only_tables(dy, ob, small_filter, tuning):
    """The HBM-sized tables (expanded SA, q-mer ranges, inverse SA + indexed text) only accelerate: with each of them switched
    off at load time (dy_ctx_set_tuning) the results stay identical to the oracle (sparse-SA walk, plain LF search, row-range
    hits); likewise the alternative aggregation / sweep kernels."""
    tctx = dy.Context(0)
    try:
        for k, v in tuning.items():
            tctx.set_tuning(k, v)
        _, codes, offs = read_fasta_codes(READS_FA)
        codes, offs = codes[: int(offs[2500])], offs[:2501]
        er = np.float32(3 / 100.0)
        dev, stats = device_map(dy, tctx, GOLD_INDEX, 64, small_filter, codes, offs, er, np.float32(0.05))
    finally:
        tctx.close()
    oidx = ob.OracleIndex(GOLD_INDEX, 64); oibf = ob.OracleIbf(small_filter)
    o_reads, o_matches, o_cigars, ostats = ob.map_batch(oidx, oibf, codes, offs, er, np.float32(0.05))
    assert stats["items"] == ostats["items"] and stats["raw_matches"] == ostats["raw_matches"]
    assert_results_equal(dev, (o_reads, o_matches, o_cigars))
    oidx.close(); oibf.close()


def test_tuning_keys_are_validated(dy):
    tctx = dy.Context(0)
    try:
        with pytest.raises(dy.DyError) as e:
            tctx.set_tuning("no_such_key", 1)
        assert e.value.code == -1 and "unknown key" in str(e.value)
        with pytest.raises(dy.DyError):
            tctx.set_tuning("prefix_q", 13)
    finally:
        tctx.close()


@pytest.mark.parametrize("gold,extra", [("t1", []), ("tag.t1", ["-sm", "tag", "-s", "10"]), ("rec.t1", ["-sm", "record", "-s", "10"]),
                                        ("t1", ["-rb", "1000", "-t", "3"])])       # 7 batches through the 3-slot reader / device / writer pipeline
def test_cli_reproduces_gold_sam(product_build, dy, small_filter, tmp_path, gold, extra):
    out = str(tmp_path / "out.sam")
    subprocess.check_call([os.path.join(product_build, "dream_yara_mapper"), "-fi", small_filter, GOLD_INDEX + "/", READS_FA,
                           "-o", out, "-e", "3", "--threads", "1"] + extra)
    mine = sam_records(out)
    ref = sam_records(os.path.join(GOLDEN, "gold", "64-viral-reads.%s.sam" % gold))
    assert len(mine) == len(ref)
    assert [(a, b) for a, b in zip(mine, ref) if a != b] == []


@pytest.mark.parametrize("extra", [["--devices", "0,0"], ["--devices", "0,0,0", "--shard", "bins"], ["--devices", "0", "--shard", "bins", "--transport", "rccl"]])
def test_cli_multi_gpu_modes_reproduce_gold_sam(product_build, dy, small_filter, tmp_path, extra):
    """dream_yara_mapper --gpus / --devices: one context per listed device in one process (the same device may be listed more than once,
    which is how a one-GPU box rehearses it).  Reads sharded (whole batches round-robin) and bins sharded (FM structures split bin % N,
    reads -> owners and raw matches -> home through device-to-device copies; with one device and --transport rccl the same sequence runs
    as self send/receive through RCCL) all give the gold SAM, over 7 batches."""
    out = str(tmp_path / "out.sam")
    subprocess.check_call([os.path.join(product_build, "dream_yara_mapper"), "-fi", small_filter, GOLD_INDEX + "/", READS_FA,
                           "-o", out, "-e", "3", "-rb", "1000", "-t", "3"] + extra)
    mine = sam_records(out)
    ref = sam_records(os.path.join(GOLDEN, "gold", "64-viral-reads.t1.sam"))
    assert len(mine) == len(ref)
    assert [(a, b) for a, b in zip(mine, ref) if a != b] == []


def _repeat_genomes(rng, n_bins, length):
    """Random genomes with planted repeat families (to trigger re-seeding of hard reads) and a few Ns."""
    fam = [rng.integers(0, 4, 220, dtype=np.uint8) for _ in range(3)]
    genomes = []
    for b in range(n_bins):
        glen = length * 5 if b == 0 else length
        g = rng.integers(0, 4, glen, dtype=np.uint8)
        copies = 1200 if b == 0 else (40 if b < 3 else 2)
        for c in range(copies):          # > 300 exact seed hits in bin 0 -> re-seeding with one seed error
            f = fam[c % 3].copy()
            mut = rng.random(f.size) < 0.01
            f[mut] = (f[mut] + 1) & 3
            p = (c * 240) % (glen - f.size) if b == 0 else int(rng.integers(0, glen - f.size))
            g[p:p + f.size] = f
        if b == 1:                       # >= 120000 exact seed hits -> re-seeding with two seed errors
            g = np.concatenate([g[:1000], np.zeros(32000, np.uint8), g[1000:]])
        g[rng.integers(0, g.size, 6)] = 4
        genomes.append(g)
    return genomes, fam


def _sample_reads(rng, genomes, fam, n, L, err):
    reads = []
    for i in range(n):
        b = int(rng.integers(0, len(genomes)))
        g = genomes[b]
        if i % 3 == 0:      # from a repeat family
            f = fam[(i // 3) % 3]
            s = int(rng.integers(0, f.size - L)) if f.size > L else 0
            r = f[s:s + L].copy()
            if r.size < L:
                r = np.concatenate([r, rng.integers(0, 4, L - r.size, dtype=np.uint8)])
        else:
            s = int(rng.integers(0, g.size - L - 4))
            r = g[s:s + L + 4].copy()
        ne = int(rng.integers(0, err + 2))
        for _ in range(ne):
            kind = rng.random()
            p = int(rng.integers(0, min(L, r.size)))
            if kind < 0.7: r[p] = (r[p] + 1 + rng.integers(0, 3)) & 3 if r[p] < 4 else 0
            elif kind < 0.85: r = np.concatenate([r[:p], rng.integers(0, 4, 1, dtype=np.uint8), r[p:]])
            else: r = np.concatenate([r[:p], r[p + 1:]])
        r = r[:L]
        if r.size < L:
            r = np.concatenate([r, rng.integers(0, 4, L - r.size, dtype=np.uint8)])
        if i % 2: r = r[::-1].copy(); r[r < 4] = 3 - r[r < 4]
        if i % 17 == 0: r[int(rng.integers(0, L))] = 4
        reads.append(r.astype(np.uint8))
    for t in range(2):                   # poly-A reads with a couple of substitutions
        r = np.zeros(L, np.uint8); r[int(rng.integers(0, L))] = 1 + t
        reads.append(r)
    n = len(reads)
    codes = np.concatenate(reads)
    offs = np.zeros(n + 1, np.uint64); offs[1:] = np.cumsum([r.size for r in reads])
    return codes, offs


@pytest.mark.parametrize("L,e,strata,sens,use_filter", [(100, 3, 0, 1, True), (100, 3, 3, 1, True), (100, 5, 0, 0, True), (150, 5, 0, 1, True),
                                                        (60, 4, 2, 1, False), (36, 0, 0, 1, True),
                                                        # -y full (sens 2): edit-distance seeds for the re-seeded reads and the edit-mode extension anchor for
                                                        # every read.  PARITY UNPINNED: SeqAn's Backtracking<EditDistance> enumeration is not in the reference
                                                        # tree and no fixture uses -y full; device == oracle (oracle/dyo_mapper.hpp fmSearchEdit defines it)
                                                        (100, 3, 0, 2, True), (100, 5, 2, 2, True), (150, 5, 0, 2, False), (60, 4, 0, 2, True)])
def test_map_matches_oracle_on_repeats_and_indels(dy, ctx, ob, tmp_path, L, e, strata, sens, use_filter):
    rng = np.random.default_rng(L * 1000 + e * 10 + strata)
    n_bins = 6
    genomes, fam = _repeat_genomes(rng, n_bins, 60000)
    idx_dir = str(tmp_path / "idx")
    os.makedirs(idx_dir)
    for b, g in enumerate(genomes):
        ob.build_bin_index(os.path.join(idx_dir, str(b)), g, np.array([0, g.size], np.uint64))
    fpath = None
    if use_filter:
        fpath = str(tmp_path / "f.filter")
        goffs = np.zeros(n_bins + 1, np.uint64); goffs[1:] = np.cumsum([g.size for g in genomes])
        ob.build_filter(fpath, n_bins, 3, 14, 1 << 24, np.concatenate(genomes), goffs)
    codes, offs = _sample_reads(rng, genomes, fam, 1500, L, max(1, int(L * e / 100)))
    er, sr = np.float32(e / 100.0), np.float32(strata / 100.0)
    dev, stats = device_map(dy, ctx, idx_dir, n_bins, fpath, codes, offs, er, sr, sens)
    oidx = ob.OracleIndex(idx_dir, n_bins)
    oibf = ob.OracleIbf(fpath) if fpath else None
    o_reads, o_matches, o_cigars, ostats = ob.map_batch(oidx, oibf, codes, offs, er, sr, sens)
    assert stats["items"] == ostats["items"]
    assert stats["raw_matches"] == ostats["raw_matches"]
    assert_results_equal(dev, (o_reads, o_matches, o_cigars))
    if L == 100 and e == 3:
        assert stats["hard_items"] > 100        # the planted repeats must exercise the re-seeding rounds
        assert (o_reads["cigar_len"] > 1).sum() > 0
    oidx.close()
    if oibf: oibf.close()


def _low_complexity_genome(rng, length):
    """Homopolymer runs, di- and tri-nucleotide repeats between short random stretches: where one substitution and one indel cost the same
    and an extension's end position ties (the shifted end of extend_sides.h's closed forms)."""
    parts, total = [], 0
    while total < length:
        kind = int(rng.integers(0, 4))
        if kind == 0: p = np.full(int(rng.integers(3, 30)), rng.integers(0, 4), np.uint8)
        elif kind == 1: p = np.tile(rng.integers(0, 4, 2, dtype=np.uint8), int(rng.integers(2, 15)))
        elif kind == 2: p = np.tile(rng.integers(0, 4, 3, dtype=np.uint8), int(rng.integers(2, 10)))
        else: p = rng.integers(0, 4, int(rng.integers(5, 40)), dtype=np.uint8)
        parts.append(p); total += p.size
    return np.concatenate(parts)[:length]


@pytest.mark.parametrize("L,e,strata,mode", [(100, 3, 0, 1), (100, 3, 2, 1), (64, 5, 0, 1), (150, 4, 1, 1), (100, 3, 0, 0)])
def test_extension_closed_forms_on_low_complexity_genomes(dy, ob, tmp_path, L, e, strata, mode):
    """K3's fast pass closes single-substitution sides without the edit distance search (k_extend<NW, 1>, extend_sides.h); on genomes of
    homopolymers and short tandem repeats — where the search's end position ties between the diagonal and the shifted end, and substitutions
    tie with indels — every field still equals the oracle's, and both the fast pass and the second pass have had work."""
    rng = np.random.default_rng(77 + L + e)
    n_bins = 4
    genomes = [_low_complexity_genome(rng, 50000) for _ in range(n_bins)]
    idx_dir = str(tmp_path / "idx")
    os.makedirs(idx_dir)
    for b, g in enumerate(genomes):
        ob.build_bin_index(os.path.join(idx_dir, str(b)), g, np.array([0, g.size], np.uint64))
    reads = []
    for i in range(3000):
        g = genomes[int(rng.integers(0, n_bins))]
        s = int(rng.integers(0, g.size - L - 3))
        r = g[s:s + L + 3].copy()
        for _ in range(int(rng.integers(0, 4))):
            p = int(rng.integers(0, L))
            kind = rng.random()
            if kind < 0.8: r[p] = (r[p] + 1 + rng.integers(0, 3)) & 3
            elif kind < 0.9: r = np.concatenate([r[:p], rng.integers(0, 4, 1, dtype=np.uint8), r[p:]])
            else: r = np.concatenate([r[:p], r[p + 1:]])
        r = r[:L]
        if i % 2: r = (3 - r[::-1]).astype(np.uint8)
        reads.append(r.astype(np.uint8))
    codes = np.concatenate(reads)
    offs = np.zeros(len(reads) + 1, np.uint64); offs[1:] = np.cumsum([r.size for r in reads])
    er, sr = np.float32(e / 100.0), np.float32(strata / 100.0)
    tctx = dy.Context(0)
    try:
        tctx.set_tuning("extend_mode", mode)
        dev, stats = device_map(dy, tctx, idx_dir, n_bins, None, codes, offs, er, sr, 1)
    finally:
        tctx.close()
    oidx = ob.OracleIndex(idx_dir, n_bins)
    o_reads, o_matches, o_cigars, ostats = ob.map_batch(oidx, None, codes, offs, er, sr, 1)
    assert stats["raw_matches"] == ostats["raw_matches"]
    assert_results_equal(dev, (o_reads, o_matches, o_cigars))
    assert 0 < stats["second_pass_items"] < stats["items"]
    oidx.close()


def test_pool_overflow_retry_with_reseeded_reads(dy, ob, tmp_path):
    """The raw-match pool starts at 64 records (tuning key pool_cap) while the batch yields ~2 million, most of them from re-seeded
    reads appended 64 at a time by k_extend_hard: every append past the capacity must be dropped and counted, the pool regrown to
    the counted size and the extension re-run ('pool overflow').  Results and counts equal the oracle's."""
    L, e = 100, 3
    rng = np.random.default_rng(L * 1000 + e * 10)
    n_bins = 6
    genomes, fam = _repeat_genomes(rng, n_bins, 60000)
    idx_dir = str(tmp_path / "idx")
    os.makedirs(idx_dir)
    for b, g in enumerate(genomes):
        ob.build_bin_index(os.path.join(idx_dir, str(b)), g, np.array([0, g.size], np.uint64))
    fpath = str(tmp_path / "f.filter")
    goffs = np.zeros(n_bins + 1, np.uint64); goffs[1:] = np.cumsum([g.size for g in genomes])
    ob.build_filter(fpath, n_bins, 3, 14, 1 << 24, np.concatenate(genomes), goffs)
    codes, offs = _sample_reads(rng, genomes, fam, 1500, L, 3)
    er = np.float32(e / 100.0)
    tctx = dy.Context(0)
    try:
        tctx.set_tuning("pool_cap", 64)
        dev, stats = device_map(dy, tctx, idx_dir, n_bins, fpath, codes, offs, er)
        assert stats["pool_retries"] == 1 and stats["hard_items"] > 0 and stats["raw_matches"] > 100000
        # the grown pool is kept: the same batch again needs no retry and gives the same bytes
        dev2, stats2 = device_map(dy, tctx, idx_dir, n_bins, fpath, codes, offs, er)
        assert stats2["pool_retries"] == 0 and stats2["raw_matches"] == stats["raw_matches"]
        assert dev2[0].tobytes() == dev[0].tobytes() and dev2[1].tobytes() == dev[1].tobytes()
    finally:
        tctx.close()
    oidx = ob.OracleIndex(idx_dir, n_bins); oibf = ob.OracleIbf(fpath)
    o_reads, o_matches, o_cigars, ostats = ob.map_batch(oidx, oibf, codes, offs, er)
    oidx.close(); oibf.close()
    assert stats["items"] == ostats["items"] and stats["raw_matches"] == ostats["raw_matches"]
    assert_results_equal(dev, (o_reads, o_matches, o_cigars))


@pytest.mark.parametrize("sens", [1, 2])
def test_variable_length_batch_matches_oracle(dy, ctx, ob, tmp_path, sens):
    """Reads of 30..260 bases in one batch, a quarter of them from a planted repeat family (re-seeded: > 300 seed hits), so that seeds of
    every length the batch allows go through the backtracking — with -y full (sens 2) through the edit-distance trie walk, whose frame
    stack is sized by the longest read of the batch (a short read with few errors is cut into a single seed)."""
    rng = np.random.default_rng(99)
    genomes = [rng.integers(0, 4, 30000 if b else 150000, dtype=np.uint8) for b in range(4)]
    fam = rng.integers(0, 4, 270, dtype=np.uint8)
    for c in range(420):
        f = fam.copy(); m = rng.random(f.size) < 0.01; f[m] = (f[m] + 1) & 3
        genomes[0][c * 300:c * 300 + f.size] = f
    idx_dir = str(tmp_path / "idx"); os.makedirs(idx_dir)
    for b, g in enumerate(genomes):
        ob.build_bin_index(os.path.join(idx_dir, str(b)), g, np.array([0, g.size], np.uint64))
    reads = []
    for i in range(800):
        L = int(rng.integers(30, 260))
        if i % 4 == 0:
            s = int(rng.integers(0, fam.size - L)) if L < fam.size else 0
            r = fam[s:s + L].copy()
        else:
            b = int(rng.integers(0, 4)); s = int(rng.integers(0, genomes[b].size - L))
            r = genomes[b][s:s + L].copy()
        for _ in range(int(rng.integers(0, 3))):
            p = int(rng.integers(0, L)); r[p] = (r[p] + 1) & 3
        if i % 2: r = r[::-1].copy(); r = 3 - r
        reads.append(r)
    codes = np.concatenate(reads); offs = np.zeros(801, np.uint64); offs[1:] = np.cumsum([r.size for r in reads])
    er = np.float32(0.04)
    dev, stats = device_map(dy, ctx, idx_dir, 4, None, codes, offs, er, sensitivity=sens)
    oidx = ob.OracleIndex(idx_dir, 4)
    o = ob.map_batch(oidx, None, codes, offs, er, sensitivity=sens, threads=8)
    assert stats["hard_items"] > 100 and stats["raw_matches"] == o[3]["raw_matches"]
    assert_results_equal(dev, o[:3])
    oidx.close()


def test_long_reads_match_oracle(dy, ctx, ob, tmp_path):
    """Reads of 1 000 .. 6 000 bases (the reference allows up to 8 191): the widest Myers instantiation (k_extend<128>), bands of up to
    61 diagonals in the wave-parallel alignment DP, the 14-plane IBF counters."""
    rng = np.random.default_rng(4096)
    n_bins = 3
    genomes = [rng.integers(0, 4, 60000, dtype=np.uint8) for _ in range(n_bins)]
    idx_dir = str(tmp_path / "idx"); os.makedirs(idx_dir)
    for b, g in enumerate(genomes):
        ob.build_bin_index(os.path.join(idx_dir, str(b)), g, np.array([0, g.size], np.uint64))
    goffs = np.zeros(n_bins + 1, np.uint64); goffs[1:] = np.cumsum([g.size for g in genomes])
    fpath = str(tmp_path / "f.filter")
    ob.build_filter(fpath, n_bins, 3, 16, 1 << 24, np.concatenate(genomes), goffs)
    reads = []
    for i in range(90):
        L = int(rng.integers(1000, 6000)); b = int(rng.integers(0, n_bins)); s = int(rng.integers(0, 60000 - L - 40))
        r = genomes[b][s:s + L + 30].copy()
        for _ in range(int(rng.integers(0, 14))):
            q = int(rng.integers(0, L)); kind = rng.random()
            if kind < 0.6: r[q] = (r[q] + 1) & 3
            elif kind < 0.8: r = np.concatenate([r[:q], r[q + 1:]])
            else: r = np.concatenate([r[:q], rng.integers(0, 4, 1, dtype=np.uint8), r[q:]])
        r = r[:L]
        if i % 2: r = (3 - r[::-1]).astype(np.uint8)
        if i % 7 == 0: r[int(rng.integers(0, L))] = 4
        reads.append(r.astype(np.uint8))
    codes = np.concatenate(reads); offs = np.zeros(len(reads) + 1, np.uint64); offs[1:] = np.cumsum([r.size for r in reads])
    er = np.float32(0.005)
    dev, stats = device_map(dy, ctx, idx_dir, n_bins, fpath, codes, offs, er, strata_rate=np.float32(0.001), read_len_for_thr=int(offs[-1]) // len(reads))
    oidx = ob.OracleIndex(idx_dir, n_bins); oibf = ob.OracleIbf(fpath)
    o = ob.map_batch(oidx, oibf, codes, offs, er, np.float32(0.001))
    assert stats["items"] == o[3]["items"] and stats["raw_matches"] == o[3]["raw_matches"]
    assert_results_equal(dev, o[:3])
    assert int(dev[0]["mapped"].sum()) > 50 and int((dev[0]["cigar_len"] > 1).sum()) > 20
    oidx.close(); oibf.close()


@pytest.mark.parametrize("paired,items_sorted", [(False, 1), (True, 1), (False, 0), (True, 0)])
def test_more_than_64_bins_end_to_end(dy, ob, tmp_path, paired, items_sorted):
    """130 bins (W = 3 mask words per read): work list, ordering, finalize and the paired-end stages with multi-word bin masks; the work
    list built both ways (tuning key items_sorted: (bin, read) pairs + stable radix sort — the default — or the bin-major count matrix)."""
    ctx = dy.Context(0)
    ctx.set_tuning("items_sorted", items_sorted)
    rng = np.random.default_rng(130 + paired)
    n_bins = 130
    genomes = [rng.integers(0, 4, 6000, dtype=np.uint8) for _ in range(n_bins)]
    genomes[7][1000:1400] = genomes[99][2000:2400]          # the same 400 bases in two bins of different mask words
    idx_dir = str(tmp_path / "idx"); os.makedirs(idx_dir)
    for b, g in enumerate(genomes):
        ob.build_bin_index(os.path.join(idx_dir, str(b)), g, np.array([0, g.size], np.uint64))
    goffs = np.zeros(n_bins + 1, np.uint64); goffs[1:] = np.cumsum([g.size for g in genomes])
    fpath = str(tmp_path / "f.filter")
    ob.build_filter(fpath, n_bins, 3, 14, 1 << 24, np.concatenate(genomes), goffs)
    if paired:
        codes, offs = _make_pairs(rng, genomes, 900, 100)
    else:
        reads = []
        for i in range(2000):
            b = 7 if i % 9 == 0 else int(rng.integers(0, n_bins)); s = 1050 + int(rng.integers(0, 200)) if i % 9 == 0 else int(rng.integers(0, 5890))
            r = genomes[b][s:s + 100].copy()
            for _ in range(int(rng.integers(0, 4))): r[int(rng.integers(0, 100))] = rng.integers(0, 4)
            if i % 2: r = (3 - r[::-1]).astype(np.uint8)
            reads.append(r.astype(np.uint8))
        codes = np.concatenate(reads); offs = np.arange(2001, dtype=np.uint64) * np.uint64(100)
    er, sr = np.float32(0.03), np.float32(0.01)
    idx = dy.Index(ctx, idx_dir, n_bins); rd = dy.Reads(ctx, codes, offs); ibf = dy.Ibf.load(ctx, fpath)
    ibf.query(rd, dy.threshold(er, 14, 100), want_masks=False)
    res = dy.map_reads(ctx, idx, rd, dy.default_params(error_rate=er, strata_rate=sr, paired=1 if paired else 0))
    dev = (res.reads(), res.matches(), res.cigars()); stats = ctx.stats()
    res.close(); rd.close(); idx.close(); ibf.close(); ctx.close()
    oidx = ob.OracleIndex(idx_dir, n_bins); oibf = ob.OracleIbf(fpath)
    o = ob.map_batch(oidx, oibf, codes, offs, er, sr, paired=paired)
    assert stats["items"] == o[3]["items"] and stats["raw_matches"] == o[3]["raw_matches"]
    if paired:
        assert np.array_equal(dev[0]["paired"], o[0]["paired"]) and np.array_equal(dev[0]["pair_prob"], o[0]["pair_prob"])
    assert_results_equal(dev, o[:3])
    if not paired:
        assert stats["items"] > offs.size - 1                # some reads go to more than one bin
        assert int((dev[0]["best_count"] > 1).sum()) > 100   # the shared segment maps to bins 7 and 99
    oidx.close(); oibf.close()


@pytest.mark.parametrize("hits_threshold", [20, 100000])
def test_hits_threshold_parameter(dy, ctx, ob, tmp_path, hits_threshold):
    """classifyReads' threshold (reference src/mapper_classifier.h:111-155, option default 300): a low value re-seeds most repeat
    reads, a huge one none — both equal the oracle."""
    rng = np.random.default_rng(300 + hits_threshold % 7)
    n_bins = 4
    genomes, fam = _repeat_genomes(rng, n_bins, 40000)
    idx_dir = str(tmp_path / "idx"); os.makedirs(idx_dir)
    for b, g in enumerate(genomes):
        ob.build_bin_index(os.path.join(idx_dir, str(b)), g, np.array([0, g.size], np.uint64))
    codes, offs = _sample_reads(rng, genomes, fam, 900, 100, 3)
    er = np.float32(0.03)
    idx = dy.Index(ctx, idx_dir, n_bins); rd = dy.Reads(ctx, codes, offs)
    params = dy.default_params(error_rate=er, hits_threshold=hits_threshold); params.use_filter = 0
    res = dy.map_reads(ctx, idx, rd, params)
    dev = (res.reads(), res.matches(), res.cigars()); stats = ctx.stats()
    res.close(); rd.close(); idx.close()
    oidx = ob.OracleIndex(idx_dir, n_bins)
    o = ob.map_batch(oidx, None, codes, offs, er, hits_threshold=hits_threshold)
    assert stats["items"] == o[3]["items"] and stats["raw_matches"] == o[3]["raw_matches"]
    assert_results_equal(dev, o[:3])
    assert (stats["hard_items"] > 300) if hits_threshold == 20 else (stats["hard_items"] < 50)      # the poly-A reads exceed even 100 000 hits
    oidx.close()


def test_empty_batch_and_unmappable_reads(dy, ctx, ob, tmp_path):
    rng = np.random.default_rng(5)
    g = rng.integers(0, 4, 5000, dtype=np.uint8)
    idx_dir = str(tmp_path / "idx"); os.makedirs(idx_dir)
    ob.build_bin_index(os.path.join(idx_dir, "0"), g, np.array([0, g.size], np.uint64))
    idx = dy.Index(ctx, idx_dir, 1)
    assert idx.contig_names() == ["contig0"] and idx.contig_lengths() == [5000]
    p = dy.default_params(error_rate=np.float32(0.03), use_filter=0)
    empty = dy.Reads(ctx, np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    res = dy.map_reads(ctx, idx, empty, p)
    assert res.reads().size == 0
    res.close(); empty.close()
    junk = np.full(100, 4, np.uint8)      # all N: nothing can match
    r = dy.Reads(ctx, np.concatenate([junk, rng.integers(0, 4, 100, dtype=np.uint8)]), np.array([0, 100, 200], np.uint64))
    res = dy.map_reads(ctx, idx, r, p)
    rr = res.reads()
    assert list(rr["mapped"]) == [0, 0] and list(rr["best_count"]) == [0, 0]
    res.close(); r.close(); idx.close()


# ---------------------------------------------------------------------------------------------- paired-end (unpinned: oracle only)
def _make_pairs(rng, genomes, n_pairs, L, mean=300, sd=25):
    r1, r2 = [], []
    for i in range(n_pairs):
        b = int(rng.integers(0, len(genomes))); g = genomes[b]
        ins = max(L + 10, int(rng.normal(mean, sd))); s = int(rng.integers(0, g.size - ins - 12))
        a = g[s:s + L].copy()
        tmpl = g[s + ins - L - 8:s + ins].copy()
        def mutate(r, k):
            for _ in range(k):
                q = int(rng.integers(0, r.size)); r[q] = (r[q] + 1 + int(rng.integers(0, 3))) & 3
        kind = i % 10
        if kind == 0:      # long deletion in the mate: seeds + extension fail, the rescue DP finds it
            cut = int(rng.integers(3, 6)); q = int(rng.integers(20, L - 20))
            tmpl = np.concatenate([tmpl[:q], tmpl[q + cut:]])
        elif kind == 1:    # long insertion
            add = int(rng.integers(3, 6)); q = int(rng.integers(20, L - 20))
            tmpl = np.concatenate([tmpl[:q], rng.integers(0, 4, add, dtype=np.uint8), tmpl[q:]])
        elif kind == 2:    # hopeless mate
            tmpl = rng.integers(0, 4, L + 8, dtype=np.uint8)
        m = tmpl[-L:] if kind != 2 else tmpl[:L]
        m = (3 - m[::-1]).astype(np.uint8)
        mutate(a, int(rng.integers(0, 3))); mutate(m, int(rng.integers(0, 2)))
        if i % 2: a, m = m, a
        r1.append(a.astype(np.uint8)); r2.append(m.astype(np.uint8))
    reads = r1 + r2
    codes = np.concatenate(reads)
    offs = np.zeros(len(reads) + 1, np.uint64); offs[1:] = np.cumsum([r.size for r in reads])
    return codes, offs


@pytest.mark.parametrize("use_filter,ll,ld,verify,strata", [(False, 0, 0, True, 0), (True, 0, 0, True, 2), (True, 320, 30, True, 0), (False, 0, 0, False, 0)])
def test_paired_end_matches_oracle(dy, ctx, ob, tmp_path, use_filter, ll, ld, verify, strata):
    rng = np.random.default_rng(77 + ll + strata)
    n_bins = 5
    genomes = [rng.integers(0, 4, 50000, dtype=np.uint8) for _ in range(n_bins)]
    genomes[1][3000:3400] = genomes[1][9000:9400]          # a repeat: several co-optimal pairs
    idx_dir = str(tmp_path / "idx"); os.makedirs(idx_dir)
    for b, g in enumerate(genomes):
        ob.build_bin_index(os.path.join(idx_dir, str(b)), g, np.array([0, g.size], np.uint64))
    fpath = None
    if use_filter:
        fpath = str(tmp_path / "f.filter")
        goffs = np.zeros(n_bins + 1, np.uint64); goffs[1:] = np.cumsum([g.size for g in genomes])
        ob.build_filter(fpath, n_bins, 3, 16, 1 << 24, np.concatenate(genomes), goffs)
    codes, offs = _make_pairs(rng, genomes, 1200, 100)
    er, sr = np.float32(0.03), np.float32(strata / 100.0)
    idx = dy.Index(ctx, idx_dir, n_bins)
    reads = dy.Reads(ctx, codes, offs)
    params = dy.default_params(error_rate=er, strata_rate=sr, paired=1, library_length=ll, library_dev=ld, verify_matches=1 if verify else 0)
    ibf = None
    if fpath:
        ibf = dy.Ibf.load(ctx, fpath)
        ibf.query(reads, dy.threshold(er, 16, 100), want_masks=False)
    else:
        params.use_filter = 0
    res = dy.map_reads(ctx, idx, reads, params)
    dev = (res.reads(), res.matches(), res.cigars())
    stats = ctx.stats()
    res.close(); reads.close(); idx.close()
    if ibf: ibf.close()
    oidx = ob.OracleIndex(idx_dir, n_bins)
    oibf = ob.OracleIbf(fpath) if fpath else None
    o_reads, o_matches, o_cigars, ostats = ob.map_batch(oidx, oibf, codes, offs, er, sr, paired=True, library_length=ll, library_dev=ld, verify=verify)
    o_det = ob.map_batch(oidx, oibf, codes, offs, er, sr, paired=True, library_length=ll, library_dev=ld, verify=verify, det_math=True)[0]
    assert stats["items"] == ostats["items"]
    assert stats["raw_matches"] == ostats["raw_matches"]
    assert_pairing_equal(dev[0], o_reads, o_det)
    assert_results_equal(dev, (o_reads, o_matches, o_cigars))
    se_reads = ob.map_batch(oidx, oibf, codes, offs, er, sr)[0]
    rescued = int(o_reads["mapped"].sum()) - int(se_reads["mapped"].sum())
    assert o_reads["paired"].sum() > 1500
    if verify:
        assert rescued > 10, rescued           # the long-indel mates are only reachable through the rescue DP
    else:
        assert rescued == 0
    oidx.close()
    if oibf: oibf.close()


def _write_fastq(path, codes, offs, lo, hi, suffix):
    with open(path, "w") as f:
        for i in range(lo, hi):
            seq = "".join("ACGT"[c] for c in codes[int(offs[i]):int(offs[i + 1])])
            f.write("@pair%d%s\n%s\n+\n%s\n" % (i - lo, suffix, seq, "I" * len(seq)))


def test_cli_bam_output_and_gzip_input(product_build, small_filter, tmp_path):
    """-o x.bam gives a valid BGZF/BAM whose decoded records equal the SAM run; gzip- and bzip2-compressed reads give the same output."""
    import bz2, gzip, shutil
    exe = os.path.join(product_build, "dream_yara_mapper")
    sam, bam, sam_gz = str(tmp_path / "o.sam"), str(tmp_path / "o.bam"), str(tmp_path / "ogz.sam")
    gz_reads = str(tmp_path / "reads.fa.gz")
    with open(READS_FA, "rb") as fi, gzip.open(gz_reads, "wb") as fo:
        shutil.copyfileobj(fi, fo)
    base = [exe, "-fi", small_filter, GOLD_INDEX + "/", "-e", "3", "-s", "10"]
    subprocess.check_call(base + [READS_FA, "-o", sam])
    subprocess.check_call(base + [READS_FA, "-o", bam])
    subprocess.check_call(base + [gz_reads, "-o", sam_gz])
    bz_reads, sam_bz = str(tmp_path / "reads.fa.bz2"), str(tmp_path / "obz.sam")
    with open(READS_FA, "rb") as fi, open(bz_reads, "wb") as fo:
        fo.write(bz2.compress(fi.read()))
    subprocess.check_call(base + [bz_reads, "-o", sam_bz])
    assert sam_records(sam_bz) == sam_records(sam)
    sam_lines = [l.rstrip("\n") for l in open(sam)]
    text, refs, recs, n_blocks = bam_to_sam_lines(bam)
    hdr = [l for l in sam_lines if l.startswith("@")]
    assert [l for l in text.splitlines() if not l.startswith("@PG")] == [l for l in hdr if not l.startswith("@PG")]
    assert [("@SQ\tSN:%s\tLN:%d" % r) for r in refs] == [l for l in hdr if l.startswith("@SQ")]
    assert recs == [l for l in sam_lines if not l.startswith("@")]
    assert len(recs) == 6400 and n_blocks > 10
    # --bam-level: the same records whatever the deflate level, a bigger file at level 1
    bam1 = str(tmp_path / "o1.bam")
    subprocess.check_call(base + [READS_FA, "-o", bam1, "--bam-level", "1", "-t", "4"])
    assert bam_to_sam_lines(bam1)[2] == recs and os.path.getsize(bam1) > os.path.getsize(bam)
    assert sam_records(sam_gz) == sam_records(sam)
    assert sam_records(sam) == sam_records(os.path.join(GOLDEN, "gold", "64-viral-reads.tag.t1.sam"))
    # -f: the format of the standard output (reference src/d_mapper.cpp:144-148,306-312); -u: that BAM uncompressed (:150-153, src/d_mapper.h:771-776)
    out_bam, out_ubam, out_sam = str(tmp_path / "stdout.bam"), str(tmp_path / "stdout_u.bam"), str(tmp_path / "stdout.sam")
    for path, extra in ((out_bam, ["-f", "bam"]), (out_ubam, ["-f", "bam", "-u"]), (out_sam, ["--output-format", "sam", "-sm", "record"])):
        with open(path, "wb") as fo:
            subprocess.check_call(base + [READS_FA] + extra, stdout=fo)
    assert bam_to_sam_lines(out_bam)[2] == recs and bam_to_sam_lines(out_ubam)[2] == recs
    assert os.path.getsize(out_ubam) > 2 * os.path.getsize(out_bam)                      # stored, not deflated
    assert sam_records(out_sam) == sam_records(sam)


@pytest.mark.gpu
def test_paired_end_on_repeat_families_matches_oracle(dy, ctx, ob, tmp_path):
    """Mates from planted repeat families: hundreds of unique matches per mate and bin -> the wavefront pairing path
    (k_pe_pair_heavy), the re-seeded extension and the large per-read aggregation, all against the oracle."""
    from dream_yara_amd import synth
    rng = np.random.default_rng(4242)
    n_bins = 4
    genomes, _ = _repeat_genomes(rng, n_bins, 60000)
    genomes[1] = np.concatenate([genomes[1][:1000], genomes[1][33000:]])        # without the 32 kbp poly-A run (pairing there is quadratic on any machine)
    idx_dir = str(tmp_path / "idx"); os.makedirs(idx_dir)
    for b, g in enumerate(genomes):
        ob.build_bin_index(os.path.join(idx_dir, str(b)), g, np.array([0, g.size], np.uint64))
    codes, offs, _ = synth.simulate_pairs(genomes, 1500, 100, seed=11, insert_mean=330.0, insert_sd=25.0)
    er, sr = np.float32(0.03), np.float32(0.0)
    idx = dy.Index(ctx, idx_dir, n_bins)
    reads = dy.Reads(ctx, codes, offs)
    params = dy.default_params(error_rate=er, strata_rate=sr, paired=1)
    params.use_filter = 0
    res = dy.map_reads(ctx, idx, reads, params)
    dev = (res.reads(), res.matches(), res.cigars())
    stats = ctx.stats()
    res.close(); reads.close(); idx.close()
    oidx = ob.OracleIndex(idx_dir, n_bins)
    o_reads, o_matches, o_cigars, ostats = ob.map_batch(oidx, None, codes, offs, er, sr, paired=True, threads=8)
    o_det = ob.map_batch(oidx, None, codes, offs, er, sr, paired=True, threads=8, det_math=True)[0]
    oidx.close()
    assert stats["raw_matches"] == ostats["raw_matches"]
    assert stats["hard_items"] > 50
    assert_pairing_equal(dev[0], o_reads, o_det)
    assert_results_equal(dev, (o_reads, o_matches, o_cigars))
    assert (o_reads["best_count"] > 100).sum() > 100           # the repeat reads really carry hundreds of co-optimal locations


@pytest.mark.parametrize("extra", [[], ["-ll", "320", "-ld", "30", "-s", "2"], ["-ni"], ["--devices", "0,0", "--shard", "bins", "-rb", "1000"],
                                   ["--devices", "0,0", "-rb", "1000"]])
def test_cli_paired_end_sam_equals_oracle_cli(product_build, oracle_build, ob, tmp_path, extra):
    """Two reads files -> paired SAM (flags, RNEXT/PNEXT/TLEN, pair MAPQ): the product CLI's text equals the oracle CLI's.
    Parity unpinned by reference fixtures (the reference ships no paired gold output)."""
    rng = np.random.default_rng(4242)
    n_bins = 4
    genomes = [rng.integers(0, 4, 40000, dtype=np.uint8) for _ in range(n_bins)]
    idx_dir = str(tmp_path / "idx"); os.makedirs(idx_dir)
    for b, g in enumerate(genomes):
        ob.build_bin_index(os.path.join(idx_dir, str(b)), g, np.array([0, g.size], np.uint64))
    fpath = str(tmp_path / "f.filter")
    goffs = np.zeros(n_bins + 1, np.uint64); goffs[1:] = np.cumsum([g.size for g in genomes])
    ob.build_filter(fpath, n_bins, 3, 16, 1 << 24, np.concatenate(genomes), goffs)
    n_pairs = 700
    codes, offs = _make_pairs(rng, genomes, n_pairs, 100)
    f1, f2 = str(tmp_path / "r_1.fq"), str(tmp_path / "r_2.fq")
    _write_fastq(f1, codes, offs, 0, n_pairs, "/1"); _write_fastq(f2, codes, offs, n_pairs, 2 * n_pairs, "/2")
    mine, ref = str(tmp_path / "mine.sam"), str(tmp_path / "ref.sam")
    subprocess.check_call([os.path.join(product_build, "dream_yara_mapper"), "-fi", fpath, idx_dir + "/", f1, f2, "-o", mine, "-e", "3"] + extra)
    oracle_extra, skip = [], False
    for tok in extra:                       # the oracle CLI has no devices to choose from
        if skip: skip = False; continue
        if tok in ("--devices", "--shard", "--transport", "--gpus"): skip = True; continue
        oracle_extra.append(tok)
    subprocess.check_call([os.path.join(oracle_build, "oracle_mapper"), "-fi", fpath, idx_dir + "/", f1, f2, "-o", ref, "-e", "3"] + oracle_extra)
    a, b = sam_records(mine), sam_records(ref)
    assert a[:3 + n_bins - 1] == b[:3 + n_bins - 1]          # header (minus @PG)
    a, b = [r for r in a if r[0] != "@"], [r for r in b if r[0] != "@"]
    assert len(a) == len(b) == 2 * n_pairs
    assert [(x, y) for x, y in zip(a, b) if x != y] == []
    flags = [int(r.split("\t")[1]) for r in a]
    assert sum(1 for f in flags if f & 2) > n_pairs          # most pairs are proper


@pytest.mark.parametrize("bins,k", [(64, 8), (64, 11), (192, 9), (64, 4)])
def test_kmer_direct_filter_matches_oracle(dy, ctx, ob, tmp_path, bins, k):
    """-ft kmer_direct (reference src/d_kdx_filter.h): file written by the device == file written by the oracle, and the
    bin masks agree, incl. reads with a leading N (whose poisoned hashes fall outside the table)."""
    rng = np.random.default_rng(bins * 100 + k)
    genomes = [rng.integers(0, 4, 3000, dtype=np.uint8) for _ in range(bins)]
    genomes[3][100] = 4; genomes[5][0] = 4
    goffs = np.zeros(bins + 1, np.uint64); goffs[1:] = np.cumsum([g.size for g in genomes])
    fo, fd = str(tmp_path / "o.kdx"), str(tmp_path / "d.kdx")
    ob.build_direct_filter(fo, bins, k, np.concatenate(genomes), goffs)
    f = dy.Ibf.create_direct(ctx, bins, k)
    for b, g in enumerate(genomes):
        f.add(g, b)
    f.save(fd); f.close()
    assert open(fo, "rb").read() == open(fd, "rb").read()
    reads = []
    for i in range(400):
        b = int(rng.integers(0, bins)); L = int(rng.integers(max(k, 20), 120)); s = int(rng.integers(0, 3000 - L))
        r = genomes[b][s:s + L].copy()
        for _ in range(int(rng.integers(0, 3))): r[int(rng.integers(0, L))] = rng.integers(0, 4)
        if i % 7 == 0: r[0] = 4
        if i % 11 == 0: r[-1] = 4
        if i % 2: r = np.where(r[::-1] < 4, 3 - r[::-1], 4).astype(np.uint8)
        reads.append(r.astype(np.uint8))
    reads.append(np.zeros(k - 1, np.uint8))
    codes = np.concatenate(reads); offs = np.zeros(len(reads) + 1, np.uint64); offs[1:] = np.cumsum([r.size for r in reads])
    f = dy.Ibf.load_direct(ctx, fo)
    assert f.info()["n_hashes"] == 1 and f.info()["kmer_size"] == k and f.info()["n_bins"] == bins
    oibf = ob.OracleIbf(fo, direct=True)
    r = dy.Reads(ctx, codes, offs)
    for thr in (0, 1, 5, 30):
        assert np.array_equal(f.query(r, thr), oibf.query(codes, offs, thr)), thr
    r.close(); f.close(); oibf.close()


def test_map_with_kmer_direct_filter_matches_oracle(dy, ctx, ob, tmp_path):
    rng = np.random.default_rng(515)
    n_bins, k = 6, 10
    genomes, fam = _repeat_genomes(rng, n_bins, 30000)
    idx_dir = str(tmp_path / "idx"); os.makedirs(idx_dir)
    for b, g in enumerate(genomes):
        ob.build_bin_index(os.path.join(idx_dir, str(b)), g, np.array([0, g.size], np.uint64))
    goffs = np.zeros(n_bins + 1, np.uint64); goffs[1:] = np.cumsum([g.size for g in genomes])
    fpath = str(tmp_path / "f.kdx")
    ob.build_direct_filter(fpath, n_bins, k, np.concatenate(genomes), goffs)
    codes, offs = _sample_reads(rng, genomes, fam, 1500, 100, 3)
    er = np.float32(0.03)
    idx = dy.Index(ctx, idx_dir, n_bins); reads = dy.Reads(ctx, codes, offs); f = dy.Ibf.load_direct(ctx, fpath)
    f.query(reads, dy.threshold(er, k, 100), want_masks=False)
    res = dy.map_reads(ctx, idx, reads, dy.default_params(error_rate=er))
    dev = (res.reads(), res.matches(), res.cigars()); stats = ctx.stats()
    res.close(); reads.close(); idx.close(); f.close()
    oidx = ob.OracleIndex(idx_dir, n_bins); oibf = ob.OracleIbf(fpath, direct=True)
    o_reads, o_matches, o_cigars, ostats = ob.map_batch(oidx, oibf, codes, offs, er, np.float32(0))
    assert stats["items"] == ostats["items"] and stats["items"] < 1502 * n_bins          # the filter does filter
    assert_results_equal(dev, (o_reads, o_matches, o_cigars))
    oidx.close(); oibf.close()


def test_cli_kmer_direct_equals_oracle_cli(product_build, oracle_build, tmp_path):
    """dream_yara_build_filter -ft kmer_direct -k 14 (2 GiB table) + dream_yara_mapper -ft kmer_direct on the reference's
    test data; the filter file and the SAM equal the oracle tools' output."""
    fa_dir = tmp_path / "bins"; fa_dir.mkdir()
    recs = open(os.path.join(GOLDEN, "input", "64-viral-genomes.fa")).read().split(">")[1:]
    assert len(recs) == 64
    for b, rec in enumerate(recs):
        (fa_dir / ("%d.fna" % b)).write_text(">" + rec)
    mine, ref = str(tmp_path / "mine.kdx"), str(tmp_path / "ref.kdx")
    subprocess.check_call([os.path.join(product_build, "dream_yara_build_filter"), "-b", "64", "-k", "14", "-ft", "kmer_direct", "-o", mine, str(fa_dir)])
    subprocess.check_call([os.path.join(oracle_build, "oracle_build_filter"), "-b", "64", "-k", "14", "-ft", "kmer_direct", "-o", ref,
                           "--split", os.path.join(GOLDEN, "input", "64-viral-genomes.fa")])
    import hashlib
    def md5(p):
        h = hashlib.md5()
        with open(p, "rb") as f:
            for chunk in iter(lambda: f.read(1 << 24), b""):
                h.update(chunk)
        return h.hexdigest()
    assert os.path.getsize(mine) == 8 + (4 ** 14) * 8 + 32 and md5(mine) == md5(ref)
    os.remove(ref)
    a, b = str(tmp_path / "a.sam"), str(tmp_path / "b.sam")
    subprocess.check_call([os.path.join(product_build, "dream_yara_mapper"), "-ft", "kmer_direct", "-fi", mine, GOLD_INDEX + "/", READS_FA, "-o", a, "-e", "3"])
    subprocess.check_call([os.path.join(oracle_build, "oracle_mapper"), "-ft", "kmer_direct", "-fi", mine, GOLD_INDEX + "/", READS_FA, "-o", b, "-e", "3"])
    os.remove(mine)
    assert sam_records(a) == sam_records(b)
    assert sam_records(a) == sam_records(os.path.join(GOLDEN, "gold", "64-viral-reads.t1.sam"))     # a filter without false negatives changes nothing


@pytest.mark.parametrize("ftype", ["bloom", "kmer_direct"])
def test_update_filter_equals_fresh_build(product_build, dy, ctx, ob, tmp_path, ftype):
    """dream_yara_update_filter (clearBins + re-add, reference src/d_update_filter.cpp:187-240): a bit belongs to one
    (position, bin), so updating bins 2 and 5 must give exactly the filter built from the new genome set."""
    rng = np.random.default_rng(31)
    bins, k = 70, 9
    old = [rng.integers(0, 4, 4000, dtype=np.uint8) for _ in range(bins)]
    new = [g.copy() for g in old]
    new[2] = rng.integers(0, 4, 5000, dtype=np.uint8); new[5] = rng.integers(0, 4, 300, dtype=np.uint8); new[69] = rng.integers(0, 4, 1000, dtype=np.uint8)
    def build(path, genomes):
        offs = np.zeros(bins + 1, np.uint64); offs[1:] = np.cumsum([g.size for g in genomes])
        if ftype == "bloom": ob.build_filter(path, bins, 3, k, 1 << 22, np.concatenate(genomes), offs)
        else: ob.build_direct_filter(path, bins, k, np.concatenate(genomes), offs)
    mine, want = str(tmp_path / "mine.filter"), str(tmp_path / "want.filter")
    build(mine, old); build(want, new)
    assert open(mine, "rb").read() != open(want, "rb").read()
    files = []
    for b in (2, 5, 69):
        p = tmp_path / ("%d.fna" % b)
        p.write_text(">c\n" + "".join("ACGT"[c] for c in new[b]) + "\n")
        files.append(str(p))
    subprocess.check_call([os.path.join(product_build, "dream_yara_update_filter"), "-ft", ftype, mine] + files)
    assert open(mine, "rb").read() == open(want, "rb").read()
    # out-of-range bin: refused like the reference (exit 1), file untouched
    bad = tmp_path / "70.fna"; bad.write_text(">c\nACGTACGTACGT\n")
    assert subprocess.call([os.path.join(product_build, "dream_yara_update_filter"), "-ft", ftype, mine, str(bad)], stderr=subprocess.DEVNULL) == 1
    assert open(mine, "rb").read() == open(want, "rb").read()


@pytest.mark.parametrize("use_filter", [True, False])
def test_multi_contig_bins_match_oracle(dy, ctx, ob, tmp_path, use_filter):
    """Bins with several contigs (incl. contigs shorter than a read, a repeat shared between contigs and Ns): exercises the
    contig ids of the expanded SA, the inverse SA / indexed text and the position-carrying hits.  Unpinned by reference
    fixtures (every gold bin has one contig); device == oracle."""
    rng = np.random.default_rng(808)
    n_bins = 4
    all_codes, bin_offs = [], []
    contigs_of = []
    for b in range(n_bins):
        lens = [20000, 60, 7000, 150, 12000][: 3 + b % 3]
        cs = [rng.integers(0, 4, L, dtype=np.uint8) for L in lens]
        cs[0][500:900] = cs[2][100:500]                       # a repeat across two contigs of the bin
        cs[0][3000] = 4; cs[2][50:53] = 4
        contigs_of.append(cs)
    idx_dir = str(tmp_path / "idx"); os.makedirs(idx_dir)
    for b, cs in enumerate(contigs_of):
        offs = np.zeros(len(cs) + 1, np.uint64); offs[1:] = np.cumsum([c.size for c in cs])
        ob.build_bin_index(os.path.join(idx_dir, str(b)), np.concatenate(cs), offs)
    fpath = None
    if use_filter:
        fpath = str(tmp_path / "f.filter")
        whole = [np.concatenate(cs) for cs in contigs_of]
        goffs = np.zeros(n_bins + 1, np.uint64); goffs[1:] = np.cumsum([g.size for g in whole])
        ob.build_filter(fpath, n_bins, 3, 14, 1 << 22, np.concatenate(whole), goffs)
    reads = []
    for i in range(2500):
        cs = contigs_of[int(rng.integers(0, n_bins))]
        c = cs[int(rng.integers(0, len(cs)))]
        L = 100
        if c.size < L + 4:
            r = np.concatenate([c, rng.integers(0, 4, L - c.size, dtype=np.uint8)])[:L]      # hangs over a short contig
        else:
            s = int(rng.integers(0, c.size - L - 2)); r = c[s:s + L + 2].copy()
            for _ in range(int(rng.integers(0, 4))):
                q = int(rng.integers(0, L)); kind = rng.random()
                if kind < 0.75: r[q] = (r[q] + 1) & 3
                elif kind < 0.9: r = np.concatenate([r[:q], r[q + 1:]])
                else: r = np.concatenate([r[:q], rng.integers(0, 4, 1, dtype=np.uint8), r[q:]])
            r = r[:L]
            if r.size < L: r = np.concatenate([r, rng.integers(0, 4, L - r.size, dtype=np.uint8)])
        if i % 2: r = np.where(r[::-1] < 4, 3 - r[::-1], 4).astype(np.uint8)
        reads.append(r.astype(np.uint8))
    codes = np.concatenate(reads); offs = np.zeros(len(reads) + 1, np.uint64); offs[1:] = np.cumsum([r.size for r in reads])
    er = np.float32(0.03)
    idx = dy.Index(ctx, idx_dir, n_bins); rd = dy.Reads(ctx, codes, offs)
    params = dy.default_params(error_rate=er, strata_rate=np.float32(0.02))
    ibf = None
    if fpath:
        ibf = dy.Ibf.load(ctx, fpath); ibf.query(rd, dy.threshold(er, 14, 100), want_masks=False)
    else:
        params.use_filter = 0
    res = dy.map_reads(ctx, idx, rd, params)
    dev = (res.reads(), res.matches(), res.cigars()); stats = ctx.stats()
    names = idx.contig_names()
    res.close(); rd.close(); idx.close()
    if ibf: ibf.close()
    oidx = ob.OracleIndex(idx_dir, n_bins); oibf = ob.OracleIbf(fpath) if fpath else None
    o_reads, o_matches, o_cigars, ostats = ob.map_batch(oidx, oibf, codes, offs, er, np.float32(0.02))
    assert len(names) == sum(len(cs) for cs in contigs_of)
    assert stats["items"] == ostats["items"] and stats["raw_matches"] == ostats["raw_matches"]
    assert_results_equal(dev, (o_reads, o_matches, o_cigars))
    assert int(dev[0]["mapped"].sum()) > 1500          # reads sampled from the 60- and 150-base contigs cannot map
    assert len(set(dev[0]["primary"]["contig"][dev[0]["mapped"] == 1].tolist())) > 8         # many different contigs are hit
    oidx.close()
    if oibf: oibf.close()


@pytest.mark.parametrize("n_contigs,contig_len", [(300, 900), (40000, 130)])
def test_bins_with_more_than_255_contigs_match_oracle(dy, ctx, ob, tmp_path, n_contigs, contig_len):
    """Draft-assembly-like bins: hundreds to tens of thousands of contigs per bin -> u16 contig ids in sa.val and the Levels form of
    lf.drp (reference src/misc_types.h:165-188, src/d_indexer.cpp:366-383); a second, single-contig bin shares a repeat with the first.
    Device == oracle on every field, global contig ids beyond 65535 included.  PARITY UNPINNED (no multi-contig gold bin exists)."""
    rng = np.random.default_rng(n_contigs)
    big = [rng.integers(0, 4, contig_len + int(rng.integers(0, 40)), dtype=np.uint8) for _ in range(n_contigs)]
    big[7][:100] = big[3][:100]; big[11][20:23] = 4
    other = [rng.integers(0, 4, 30000, dtype=np.uint8)]
    other[0][1000:1120] = big[5][:120]
    bins = [other, big, [rng.integers(0, 4, 500, dtype=np.uint8) for _ in range(30000)]] if n_contigs > 30000 else [other, big]
    n_bins = len(bins)
    idx_dir = str(tmp_path / "idx"); os.makedirs(idx_dir)
    for b, cs in enumerate(bins):
        offs = np.zeros(len(cs) + 1, np.uint64); offs[1:] = np.cumsum([c.size for c in cs])
        ob.build_bin_index(os.path.join(idx_dir, str(b)), np.concatenate(cs), offs)
    whole = [np.concatenate(cs) for cs in bins]
    fpath = str(tmp_path / "f.filter")
    goffs = np.zeros(n_bins + 1, np.uint64); goffs[1:] = np.cumsum([g.size for g in whole])
    ob.build_filter(fpath, n_bins, 3, 16, 1 << 27, np.concatenate(whole), goffs)
    reads = []
    for i in range(3000):
        cs = bins[int(rng.integers(0, n_bins))]
        c = cs[int(rng.integers(0, len(cs)))]
        L = 100
        s = int(rng.integers(0, c.size - L)); r = c[s:s + L].copy()
        for _ in range(int(rng.integers(0, 4))): r[int(rng.integers(0, L))] = rng.integers(0, 4)
        if i % 2: r = np.where(r[::-1] < 4, 3 - r[::-1], 4).astype(np.uint8)
        reads.append(r.astype(np.uint8))
    codes = np.concatenate(reads); offs = np.zeros(len(reads) + 1, np.uint64); offs[1:] = np.cumsum([r.size for r in reads])
    er = np.float32(0.03)
    dev, stats = device_map(dy, ctx, idx_dir, n_bins, fpath, codes, offs, er, np.float32(0.02))
    oidx = ob.OracleIndex(idx_dir, n_bins); oibf = ob.OracleIbf(fpath)
    o = ob.map_batch(oidx, oibf, codes, offs, er, np.float32(0.02), threads=8)
    oidx.close(); oibf.close()
    assert stats["items"] == o[3]["items"] and stats["raw_matches"] == o[3]["raw_matches"]
    assert_results_equal(dev, o[:3])
    mapped = dev[0]["mapped"] == 1
    assert mapped.sum() > 2500
    assert dev[0]["primary"]["contig"][mapped].max() > (65535 if n_contigs > 30000 else 255)        # ids beyond the narrower type are reported


def test_gpu_indexer_reproduces_gold_bytes_and_the_host_sorter(product_build, dy, ctx, tmp_path):
    """dream_yara_indexer builds the FM index of every bin on the device (dy_fm_build: suffix array by prefix doubling over hipcub
    radix sorts, then BWT, rank dictionary blocks, sentinel rows and the sampled suffix array).  The 768 files it writes for the 64 gold
    genomes are the gold files byte for byte — also with only the suffix sort on the device (`--sorter sa`); on texts that need many doubling
    rounds (a 30 kbp homopolymer run, planted repeats) and on a bin of 301 contigs (u16 contig ids, sentinel rows as a Levels dictionary)
    device, suffix-sort-only and host builds write the same bytes."""
    import filecmp
    names, codes, offs = read_fasta_codes(GENOMES_FA)
    fa_dir = tmp_path / "fa"; fa_dir.mkdir()
    for b in range(64):
        with open(fa_dir / ("%d.fna" % b), "wb") as f:
            f.write(b">" + names[b].encode() + b"\n" + np.frombuffer(b"ACGTN", np.uint8)[codes[int(offs[b]):int(offs[b + 1])]].tobytes() + b"\n")
    out = tmp_path / "idx"; out.mkdir()
    log = subprocess.run([os.path.join(product_build, "dream_yara_indexer"), "-v", "--sorter", "gpu", "-t", "8", "-o", str(out) + "/"] +
                         [str(fa_dir / ("%d.fna" % b)) for b in range(64)], check=True, capture_output=True, text=True).stderr
    assert "index construction: device 0" in log
    files = sorted(os.listdir(GOLD_INDEX))
    assert len(files) == 768
    for n in files:
        assert filecmp.cmp(os.path.join(GOLD_INDEX, n), str(out / n), shallow=False), n
    out2 = tmp_path / "idx_sa"; out2.mkdir()
    log = subprocess.run([os.path.join(product_build, "dream_yara_indexer"), "-v", "--sorter", "sa", "-t", "8", "-o", str(out2) + "/"] +
                         [str(fa_dir / ("%d.fna" % b)) for b in range(0, 64, 7)], check=True, capture_output=True, text=True).stderr
    assert "suffix sort on device 0" in log
    for n in sorted(os.listdir(out2)):
        assert filecmp.cmp(os.path.join(GOLD_INDEX, n), str(out2 / n), shallow=False), n
    # many doubling rounds: homopolymer run + repeats; many sentinels
    rng = np.random.default_rng(5)
    g = rng.integers(0, 4, 200000, dtype=np.uint8)
    g[1000:31000] = 0
    g[60000:61000] = g[90000:91000]
    a = tmp_path / "a"; a.mkdir(); b = tmp_path / "b"; b.mkdir()
    with open(tmp_path / "0.fna", "wb") as f:
        f.write(b">hard\n" + np.frombuffer(b"ACGTN", np.uint8)[g].tobytes() + b"\n")
        for i in range(300):
            c = rng.integers(0, 4, int(rng.integers(30, 120)), dtype=np.uint8)
            f.write(b">c%d\n" % i + np.frombuffer(b"ACGTN", np.uint8)[c].tobytes() + b"\n")
    c3 = tmp_path / "c3"; c3.mkdir()
    subprocess.check_call([os.path.join(product_build, "dream_yara_indexer"), "--sorter", "gpu", "-o", str(a) + "/", str(tmp_path / "0.fna")])
    subprocess.check_call([os.path.join(product_build, "dream_yara_indexer"), "--sorter", "host", "-o", str(b) + "/", str(tmp_path / "0.fna")])
    subprocess.check_call([os.path.join(product_build, "dream_yara_indexer"), "--sorter", "sa", "-o", str(c3) + "/", str(tmp_path / "0.fna")])
    assert len(os.listdir(a)) == 12
    for n in sorted(os.listdir(a)):
        assert filecmp.cmp(str(a / n), str(b / n), shallow=False), n
        assert filecmp.cmp(str(a / n), str(c3 / n), shallow=False), n
    # the ABI entry on its own: two sequences with their sentinels
    text = np.array([2 + 1, 2 + 0, 2 + 1, 2 + 0, 0, 2 + 0, 2 + 1, 1], dtype=np.uint32)       # "CACA$0 AC$1" over {$0,$1,A,C,G,T}
    sa = dy.suffix_array(ctx, text, 6)
    want = sorted(range(text.size), key=lambda i: text[i:].tolist())
    assert sa.tolist() == want


# ---------------------------------------------------------------------------------------------- error behaviour of the boundary
def test_abi_error_codes_and_messages(dy, ctx, small_filter, tmp_path):
    """Errors come back as DY_E* codes with the reference's wording where it has one (reference src/mapper.h:453-454,
    src/d_mapper.cpp:574-582); nothing crashes, nothing falls back."""
    EINVAL, EIO, EUNSUPPORTED = -1, -2, -6
    codes = np.zeros(300, np.uint8); offs = np.array([0, 100, 200, 300], np.uint64)
    idx = dy.Index(ctx, GOLD_INDEX, 64)
    reads = dy.Reads(ctx, codes, offs)
    with pytest.raises(dy.DyError) as e:                      # masks missing: the filter has to run first
        dy.map_reads(ctx, idx, reads, dy.default_params(error_rate=np.float32(0.03)))
    assert e.value.code == EINVAL and "dy_ibf_query" in str(e.value)
    ibf = dy.Ibf.load(ctx, small_filter)
    ibf.query(reads, 10, want_masks=False)
    with pytest.raises(dy.DyError) as e:                      # odd number of reads in a paired batch
        dy.map_reads(ctx, idx, reads, dy.default_params(error_rate=np.float32(0.03), paired=1))
    assert e.value.code == EINVAL
    with pytest.raises(dy.DyError) as e:                      # there is no sensitivity beyond full
        dy.map_reads(ctx, idx, reads, dy.default_params(error_rate=np.float32(0.03), sensitivity=3))
    assert e.value.code == EINVAL and "sensitivity" in str(e.value)
    with pytest.raises(dy.DyError) as e:
        dy.map_reads(ctx, idx, reads, dy.default_params(error_rate=np.float32(0.5)))
    assert e.value.code == EINVAL
    with pytest.raises(dy.DyError) as e:                      # reference: "Maximum read length exceeded."
        dy.Reads(ctx, np.zeros(9000, np.uint8), np.array([0, 9000], np.uint64))
    assert e.value.code == EINVAL and "Maximum read length exceeded" in str(e.value)
    with pytest.raises(dy.DyError) as e:
        dy.Index(ctx, str(tmp_path / "nowhere"), 4)
    assert e.value.code == EIO
    bad = tmp_path / "bad.filter"; bad.write_bytes(b"\x00" * 20)
    with pytest.raises(dy.DyError) as e:
        dy.Ibf.load(ctx, str(bad))
    assert e.value.code in (EIO, EINVAL)
    with pytest.raises(dy.DyError) as e:                      # a filter of another bin count than the index
        f2 = dy.Ibf.create(ctx, 32, 2, 12, (1 << 20) + 256)
        f2.query(reads, 5, want_masks=False)
        dy.map_reads(ctx, idx, reads, dy.default_params(error_rate=np.float32(0.03)))
    assert e.value.code == EINVAL
    # after all those refusals the handles still work
    ibf.query(reads, 10, want_masks=False)
    res = dy.map_reads(ctx, idx, reads, dy.default_params(error_rate=np.float32(0.03)))
    assert res.reads().size == 3
    res.close(); f2.close(); ibf.close(); reads.close(); idx.close()


def test_cli_sensitivity_full_equals_oracle_cli(product_build, oracle_build, small_filter, tmp_path):
    """-y full through the command line on the gold data: the product's SAM equals the oracle CLI's (PARITY UNPINNED: the reference's tests
    never run -y full).  Every extension uses the edit-distance anchor in this mode, so a few records differ from the -y high gold file."""
    mine, ref = str(tmp_path / "mine.sam"), str(tmp_path / "ref.sam")
    subprocess.check_call([os.path.join(product_build, "dream_yara_mapper"), "-fi", small_filter, GOLD_INDEX + "/", READS_FA, "-o", mine, "-e", "3", "-y", "full"])
    subprocess.check_call([os.path.join(oracle_build, "oracle_mapper"), "-fi", small_filter, GOLD_INDEX + "/", READS_FA, "-o", ref, "-e", "3", "-y", "full"])
    a, b = sam_records(mine), sam_records(ref)
    assert len(a) == len(b) and [(x, y) for x, y in zip(a, b) if x != y] == []
    gold = sam_records(os.path.join(GOLDEN, "gold", "64-viral-reads.t1.sam"))
    assert sum(1 for x, y in zip(a, gold) if x != y) < len(gold) // 20          # the same reads map, a handful of alignments differ


def test_cli_errors_exit_1_with_the_tool_name(product_build, small_filter, tmp_path):
    exe = os.path.join(product_build, "dream_yara_mapper")
    def run(args):
        p = subprocess.run([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        return p.returncode, p.stderr.decode()
    rc, err = run(["-fi", small_filter, GOLD_INDEX + "/", str(tmp_path / "missing.fa")])
    assert rc == 1 and err.startswith("dream_yara_mapper: ")
    rc, err = run(["-fi", small_filter, GOLD_INDEX + "/", READS_FA, "--no-such-option"])
    assert rc == 1 and "unknown option" in err
    rc, err = run(["-fi", small_filter, GOLD_INDEX + "/", READS_FA, "-e", "11"])
    assert rc == 1 and "out of range" in err
    rc, err = run(["-fi", small_filter, GOLD_INDEX + "/", READS_FA, "-y", "fullest"])
    assert rc == 1 and "sensitivity" in err
    rc, err = run(["-fi", str(tmp_path / "none.filter"), GOLD_INDEX + "/", READS_FA])
    assert rc == 1 and err.startswith("dream_yara_mapper: ")
    rc, err = run([GOLD_INDEX + "/"])
    assert rc == 1
    # the legacy option the reference keeps only to say so (src/d_mapper.cpp:165-168,323-327)
    rc, err = run(["-fi", small_filter, GOLD_INDEX + "/", READS_FA, "-sa", "tag"])
    assert rc == 1 and err.strip() == "dream_yara_mapper: The 'secondary-alignments' option has been renamed to 'secondary-matches'!"
    rc, err = run(["-fi", small_filter, GOLD_INDEX + "/", READS_FA, "-f", "cram"])
    assert rc == 1 and "output-format" in err
    rc, err = run(["-fi", small_filter, GOLD_INDEX + "/", READS_FA, "-sm", "all"])
    assert rc == 1 and "secondary-matches" in err
    rc, err = run(["-fi", small_filter, GOLD_INDEX + "/", READS_FA, "-ra"])
    assert rc == 1 and "not implemented" in err                                       # parsed, and refused in so many words
    rc, err = run(["-fi", small_filter, GOLD_INDEX + "/", READS_FA, "-b", "64"])
    assert rc == 1 and "unknown option" in err                                        # commented out in the reference's parser too (src/d_mapper.cpp:247-251)


# ---------------------------------------------------------------------------------------------- bins sharded over ranks
def _run_sharded_workers(tmp_path, n_bins, k, strata, port, extra=()):
    import sys, json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    rc = subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(os.path.dirname(__file__), "sharded_worker.py"), str(tmp_path), str(n_bins), str(k),
                          str(strata)] + list(extra), env=env, timeout=600)
    outs = [json.load(open(tmp_path / ("out%d.json" % r))) for r in range(2)]
    assert rc == 0, outs
    assert all(o["bad"] == [] for o in outs), outs
    return outs


@pytest.mark.parametrize("strata,n_bins,ragged", [(0.0, 6, False), (0.02, 6, False), (0.0, 70, False), (0.0, 6, True)])
def test_bin_sharded_two_ranks_equal_single_gpu(product_build, ob, tmp_path, strata, n_bins, ragged):
    """two ranks (both on cuda:0, exchanges over gloo) each hold the FM structures of half the bins; the
    merged per-read results equal the single-GPU run on the concatenated batch (which the other tests tie to the oracle).
    ragged: reads of 61..200 bases, odd lengths among them (the bases travel two per byte, every read from a byte of its own)."""
    rng = np.random.default_rng(99)
    k = 16
    genomes, fam = _repeat_genomes(rng, n_bins, 60000 if n_bins < 10 else 8000)      # 70 bins: two mask words per read
    idx_dir = str(tmp_path / "idx"); os.makedirs(idx_dir)
    for b, g in enumerate(genomes):
        ob.build_bin_index(os.path.join(idx_dir, str(b)), g, np.array([0, g.size], np.uint64))
    goffs = np.zeros(n_bins + 1, np.uint64); goffs[1:] = np.cumsum([g.size for g in genomes])
    ob.build_filter(str(tmp_path / "f.filter"), n_bins, 3, k, 1 << 24, np.concatenate(genomes), goffs)
    codes, offs = _sample_reads(rng, genomes, fam, 2998, 200 if ragged else 100, 3)
    if ragged:
        keep = rng.integers(61, 201, offs.size - 1)
        codes = np.concatenate([codes[int(offs[i]):int(offs[i]) + int(keep[i])] for i in range(offs.size - 1)])
        offs = np.zeros(keep.size + 1, np.uint64); offs[1:] = np.cumsum(keep)
        assert (keep & 1).sum() > 500
    np.savez(str(tmp_path / "reads.npz"), codes=codes, offs=offs)
    outs = _run_sharded_workers(tmp_path, n_bins, k, strata, 29617)
    assert sum(o["n"] for o in outs) == 3000 and sum(o["mapped"] for o in outs) > 2000
    assert all(o["recv_reads"] > 0 and o["raw_in"] > 0 for o in outs)                 # both exchanges carried data
    assert sum(o["sent_reads"] for o in outs) == sum(o["recv_reads"] for o in outs)
    assert sum(o["raw_out"] for o in outs) == sum(o["raw_in"] for o in outs)


@pytest.mark.parametrize("strata,n_bins", [(0.0, 5), (0.02, 70)])
def test_bin_sharded_paired_end_two_ranks_equal_single_gpu(product_build, ob, tmp_path, strata, n_bins):
    """BASELINE configs[3]'s partitioning: paired-end over bins sharded across ranks.  Mates travel together to the owners of the bins
    either selects; the owner runs the child mapper's pairing, library statistics and mate rescue for its bins (reference
    src/mapper.h:927-1172) and ships raw matches + per-(read,bin) pairing records home, where the per-read merge runs
    (copyMatches / copyCigars, src/d_mapper.h:127-215).  Every field of every read equals the single-GPU paired run on the whole job."""
    rng = np.random.default_rng(4100 + n_bins)
    k = 16
    genomes = [rng.integers(0, 4, 50000 if n_bins < 10 else 9000, dtype=np.uint8) for _ in range(n_bins)]
    genomes[1][3000:3400] = genomes[1][7000:7400]                                  # a repeat: several co-optimal pairs
    idx_dir = str(tmp_path / "idx"); os.makedirs(idx_dir)
    for b, g in enumerate(genomes):
        ob.build_bin_index(os.path.join(idx_dir, str(b)), g, np.array([0, g.size], np.uint64))
    goffs = np.zeros(n_bins + 1, np.uint64); goffs[1:] = np.cumsum([g.size for g in genomes])
    ob.build_filter(str(tmp_path / "f.filter"), n_bins, 3, k, 1 << 24, np.concatenate(genomes), goffs)
    codes, offs = _make_pairs(rng, genomes, 1500, 100)                             # long-indel and hopeless mates: work for the rescue
    np.savez(str(tmp_path / "reads.npz"), codes=codes, offs=offs)
    outs = _run_sharded_workers(tmp_path, n_bins, k, strata, 29619, ["paired"])
    assert sum(o["n"] for o in outs) == 3000 and sum(o["mapped"] for o in outs) > 2500
    assert sum(o["paired_reads"] for o in outs) > 2000 and sum(o["gapped"] for o in outs) > 50
    assert all(o["recv_reads"] > 0 and o["raw_in"] > 0 and o["pe_items_in"] > 0 for o in outs)


# ---------------------------------------------------------------------------------------------- larger sizes: properties
def test_error_free_reads_map_to_their_origin_at_scale(dy, ctx, product_build, tmp_path):
    from dream_yara_amd import synth
    n_bins, glen, n_reads, L = 16, 300000, 200000, 100
    genomes = [synth.make_genome(b, glen) for b in range(n_bins)]
    idx_dir = synth.build_index_dir(genomes, str(tmp_path / "idx"), threads=8)
    f = dy.Ibf.create(ctx, n_bins, 3, 18, (1 << 28) + 256)
    for b, g in enumerate(genomes):
        f.add(g, b)
    codes, offs, truth = synth.simulate_reads(genomes, n_reads, L, seed=1234, sub_lo=0.0, sub_hi=0.0, indel_rate=0.0)
    reads = dy.Reads(ctx, codes, offs)
    er = np.float32(0.03)
    masks = f.query(reads, dy.threshold(er, 18, L))
    # every read's own bin must be selected (no false negatives in a Bloom filter)
    assert np.all((masks[np.arange(n_reads), 0] >> truth["bin"].astype(np.uint64)) & np.uint64(1))
    idx = dy.Index(ctx, idx_dir, n_bins)
    res = dy.map_reads(ctx, idx, reads, dy.default_params(error_rate=er))
    rr = res.reads()
    assert rr["mapped"].all()
    assert (rr["primary"]["errors"] == 0).all()
    assert np.array_equal(rr["primary"]["contig"], truth["bin"])
    want_begin = np.where(truth["strand"] == 0, truth["start"], truth["start"] + 8)
    assert np.array_equal(rr["primary"]["begin"].astype(np.int64), want_begin)
    assert np.array_equal(rr["primary"]["is_rev"], truth["strand"])
    assert (rr["best_count"] == 1).all() and (rr["mapq"] == 70).all()      # unique NM:0 hits get MAPQ 70 like the gold SAMs
    res.close(); idx.close(); reads.close(); f.close()

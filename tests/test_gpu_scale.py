"""GPU parity at the geometries BASELINE.json quotes (configs[1]..[4]): the HIP path through the C ABI against the CPU oracle
on the same seeded inputs, every field of every read.  The inputs are the bench's own (bench.WORKLOADS, dream_yara_amd/synth.py):

  config2   64 bins x 4 Mbp,  1 GiB IBF,  1 M x 100 bp SE, e=3     full size: masks, results, item / raw-match counts
  config3  256 bins x 4 Mbp, 16 GiB IBF,  250 k x 100 bp SE, e=3   real filter size (W = 4), all 256 FM indices resident
  config4  256 bins, 16 GiB IBF, 100 k pairs 2 x 100 bp, e=3       pairing + rescue (the oracle's PE half is parity-unpinned: no PE gold exists)
  config5 1024 bins x 4 Mbp, 64 GiB IBF, 150 bp, e=5 (threshold 7) all 1024 FM indices resident: K1 masks and every result field of 200 k reads
                                                                   against the oracle running on a host copy of the device-resident filter,
                                                                   the own-bin property on the whole 1 M batch
  northstar 64 bins, 16 GiB IBF, one 2 M batch of 100 bp SE, e=3   the target sentence's geometry (the 10 M reads are 5 such batches)
  256 bins sharded over two ranks (SE and PE)                      the partitioning of configs[2] / [3] at their bin count

A batch of 10 M / 50 M reads is that many independent repetitions of these steps (reads never interact outside a pair), so the
per-batch comparisons are the parity statement for the full configurations; the multi-GPU partitioning of the same work is covered
by test_bins_sharded_two_ranks_at_256_bins here and test_bin_sharded_*_two_ranks_equal_single_gpu in test_gpu_parity.py.  Index directories are cached under the bench's work dir and reused by bench.py.
"""
import os
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
THREADS = min(os.cpu_count() or 1, 64)


def _log(msg):
    print("[scale] " + msg, file=sys.stderr, flush=True)


@pytest.fixture(scope="module")
def dy(product_build):
    import dream_yara_amd as m
    if m.device_count() == 0:
        pytest.fail("no HIP device: the gpu-marked tests must run on the GPU box")
    return m


@pytest.fixture(scope="module")
def ob(oracle_build):
    import oracle_binding
    return oracle_binding


def _inputs(name):
    """Genomes + index directory of a bench workload (built once per box, shared with bench.py)."""
    import bench
    wl = bench.WORKLOADS[name]
    work = os.path.join(os.environ.get("DY_BENCH_DIR", "/tmp/dy_bench_%d" % os.getuid()), name)
    os.makedirs(work, exist_ok=True)
    t0 = time.time()
    genomes, idx_dir = bench.build_inputs(wl, work, 0, 1, lambda: None)
    _log("%s: genomes + index in %.1f s" % (name, time.time() - t0))
    return wl, genomes, idx_dir


def _device_filter(dy, ctx, wl, genomes):
    t0 = time.time()
    ibf = dy.Ibf.create(ctx, wl["bins"], wl["h"], wl["k"], (wl["ibf_gib"] << 33) + 256)
    for b, g in enumerate(genomes):
        ibf.add(g, b)
    _log("filter %d GiB built on the device in %.1f s" % (wl["ibf_gib"], time.time() - t0))
    return ibf


def _oracle_filter(ob, ibf):
    """The oracle reads the very words the device holds (copied back through the ABI)."""
    t0 = time.time()
    words, n_bits = ibf.words()
    _log("filter copied to the host in %.1f s" % (time.time() - t0))
    return ob.OracleIbf(None, words=words, n_bits=n_bits)


def assert_all_fields_equal(dev, ora, paired=False):
    d_reads, d_matches, d_cigars = dev
    o_reads, o_matches, o_cigars = ora
    assert d_reads.size == o_reads.size
    fields = ("mapped", "best_count", "sub_count", "mapq", "cigar_len", "set_begin", "cigar_begin") + (("paired",) if paired else ())
    if paired:      # the oracle evaluates the pairing weights with libm like the reference, the device with fixed IEEE sequences (dy_detmath.h)
        np.testing.assert_allclose(d_reads["pair_prob"], o_reads["pair_prob"], rtol=1e-12, atol=0)
    for f in fields:
        bad = np.nonzero(d_reads[f] != o_reads[f])[0]
        assert bad.size == 0, (f, bad.size, bad[:10], d_reads[bad[:3]], o_reads[bad[:3]])
    m = o_reads["mapped"] == 1
    assert d_reads["primary"][m].tobytes() == o_reads["primary"][m].tobytes(), "primary matches (contig, begin, length, errors, strand) differ"
    assert d_matches.tobytes() == o_matches.tobytes(), "matchesSet differs"
    assert np.array_equal(d_cigars, o_cigars), "CIGARs differ"


# ------------------------------------------------------------------------------------------------- configs[1], full size
def test_config2_full_size_every_field(dy, ob):
    from dream_yara_amd import synth
    wl, genomes, idx_dir = _inputs("config2")
    ctx = dy.Context(0)
    try:
        ibf = _device_filter(dy, ctx, wl, genomes)
        index = dy.Index(ctx, idx_dir, wl["bins"])
        codes, offs, truth = synth.simulate_reads(genomes, wl["reads"], wl["L"], seed=0x5EED0002)
        er = np.float32(wl["e"] / 100.0)
        thr = dy.threshold(er, wl["k"], wl["L"])
        assert thr == 29
        reads = dy.Reads(ctx, codes, offs)
        d_masks = ibf.query(reads, thr)
        res = dy.map_reads(ctx, index, reads, dy.default_params(error_rate=er))
        dev = (res.reads(), res.matches(), res.cigars()); st = ctx.stats()
        res.close(); reads.close(); index.close()
        oibf = _oracle_filter(ob, ibf)
        ibf.close()
        t0 = time.time()
        o_masks = oibf.query(codes, offs, thr, threads=THREADS)
        assert np.array_equal(d_masks, o_masks), "K1 masks differ at %d reads" % int((d_masks != o_masks).any(axis=1).sum())
        oidx = ob.OracleIndex(idx_dir, wl["bins"])
        o = ob.map_batch(oidx, oibf, codes, offs, er, threads=THREADS)
        _log("config2: oracle on 1 M reads in %.1f s" % (time.time() - t0))
        oidx.close(); oibf.close()
        assert_all_fields_equal(dev, o[:3])
        assert st["items"] == o[3]["items"] and st["raw_matches"] == o[3]["raw_matches"]
        m = dev[0]["mapped"] == 1
        assert m.mean() > 0.99 and (dev[0]["primary"]["contig"][m] == truth["bin"][m]).mean() > 0.9999
    finally:
        ctx.close()


# ------------------------------------------------------------------------------------------------- -y low / -y full at the config2 geometry
@pytest.mark.parametrize("sens,e,strata", [(0, 5, 0.0), (2, 3, 0.02)])
def test_config2_geometry_other_sensitivities(dy, ob, sens, e, strata):
    """The config2 index with harsher reads (4 % substitutions at the far end, 20x the indels, Ns): -y low at -e 5 (filter threshold 0:
    every read goes to all 64 bins) and -y full (edit-mode extension for every hit; PARITY UNPINNED) — every field
    of 300 k reads equals the oracle's."""
    from dream_yara_amd import synth
    wl, genomes, idx_dir = _inputs("config2")
    ctx = dy.Context(0)
    try:
        ibf = _device_filter(dy, ctx, wl, genomes)
        index = dy.Index(ctx, idx_dir, wl["bins"])
        N = 300_000
        codes, offs, _ = synth.simulate_reads(genomes, N, 100, seed=777, sub_lo=0.005, sub_hi=0.04, indel_rate=0.002)
        rng = np.random.default_rng(3)
        codes[rng.integers(0, codes.size, N // 20)] = 4
        er, sr = np.float32(e / 100.0), np.float32(strata)
        thr = dy.threshold(er, wl["k"], 100)
        reads = dy.Reads(ctx, codes, offs)
        ibf.query(reads, thr, want_masks=False)
        res = dy.map_reads(ctx, index, reads, dy.default_params(error_rate=er, strata_rate=sr, sensitivity=sens))
        dev = (res.reads(), res.matches(), res.cigars()); st = ctx.stats()
        res.close(); reads.close(); index.close()
        oibf = _oracle_filter(ob, ibf)
        ibf.close()
        oidx = ob.OracleIndex(idx_dir, wl["bins"])
        o = ob.map_batch(oidx, oibf, codes, offs, er, sr, sensitivity=sens, threads=THREADS)
        oidx.close(); oibf.close()
        assert_all_fields_equal(dev, o[:3])
        # threshold 0: the reference pushes every read to every bin once per strand (src/d_mapper.h:320-333), so its counters are twice
        # the device's, which skips the duplicate pass; the de-duplicated results are the same
        mult = 2 if thr == 0 else 1
        assert mult * st["items"] == o[3]["items"] and mult * st["raw_matches"] == o[3]["raw_matches"]
        assert (dev[0]["cigar_len"] > 1).sum() > 10000
    finally:
        ctx.close()


# ------------------------------------------------------------------------------------------------- L = 150, e = 5 on the 64-bin index
def test_150bp_e5_seeds_with_tail_match_oracle(dy, ob):
    """configs[4]'s read shape: 150 bp at -e 5 -> 7 errors, 8 seeds of 18 with a 6-base uncovered tail, IBF threshold 7."""
    from dream_yara_amd import synth
    wl, genomes, idx_dir = _inputs("config2")
    ctx = dy.Context(0)
    try:
        ibf = _device_filter(dy, ctx, wl, genomes)
        index = dy.Index(ctx, idx_dir, wl["bins"])
        N = 200_000
        codes, offs, _ = synth.simulate_reads(genomes, N, 150, seed=0x5EED0005, sub_lo=0.002, sub_hi=0.04, indel_rate=0.001)
        er = np.float32(0.05)
        thr = dy.threshold(er, wl["k"], 150)
        assert thr == 7
        reads = dy.Reads(ctx, codes, offs)
        d_masks = ibf.query(reads, thr)
        res = dy.map_reads(ctx, index, reads, dy.default_params(error_rate=er))
        dev = (res.reads(), res.matches(), res.cigars()); st = ctx.stats()
        res.close(); reads.close(); index.close()
        oibf = _oracle_filter(ob, ibf)
        ibf.close()
        assert np.array_equal(d_masks, oibf.query(codes, offs, thr, threads=THREADS))
        oidx = ob.OracleIndex(idx_dir, wl["bins"])
        o = ob.map_batch(oidx, oibf, codes, offs, er, threads=THREADS)
        oidx.close(); oibf.close()
        assert_all_fields_equal(dev, o[:3])
        assert st["items"] == o[3]["items"] and st["raw_matches"] == o[3]["raw_matches"]
        assert (dev[0]["cigar_len"] > 1).sum() > 1000          # the gapped paths are exercised
    finally:
        ctx.close()


# ------------------------------------------------------------------------------------------------- configs[2] and [3] geometry
@pytest.fixture(scope="module")
def geometry256(dy, ob):
    """256 bins x 4 Mbp: all FM indices resident, the real 16 GiB filter on the device and its host copy for the oracle."""
    wl, genomes, idx_dir = _inputs("config3")
    ctx = dy.Context(0)
    ibf = _device_filter(dy, ctx, wl, genomes)
    t0 = time.time()
    index = dy.Index(ctx, idx_dir, wl["bins"])
    _log("256 FM indices uploaded in %.1f s" % (time.time() - t0))
    oibf = _oracle_filter(ob, ibf)
    oidx = ob.OracleIndex(idx_dir, wl["bins"])
    yield wl, genomes, ctx, ibf, index, oibf, oidx
    oidx.close(); oibf.close(); index.close(); ibf.close(); ctx.close()


def test_config3_geometry_single_end(dy, ob, geometry256):
    from dream_yara_amd import synth
    wl, genomes, ctx, ibf, index, oibf, oidx = geometry256
    assert ibf.info()["words_per_pos"] == 4 and ibf.info()["n_bits"] == (16 << 33) + 256
    N = 250_000
    codes, offs, truth = synth.simulate_reads(genomes, N, 100, seed=0x5EED0003)
    er = np.float32(0.03)
    thr = dy.threshold(er, wl["k"], 100)
    reads = dy.Reads(ctx, codes, offs)
    d_masks = ibf.query(reads, thr)
    res = dy.map_reads(ctx, index, reads, dy.default_params(error_rate=er))
    dev = (res.reads(), res.matches(), res.cigars()); st = ctx.stats()
    res.close(); reads.close()
    t0 = time.time()
    o_masks = oibf.query(codes, offs, thr, threads=THREADS)
    assert np.array_equal(d_masks, o_masks), "K1 masks differ at %d reads" % int((d_masks != o_masks).any(axis=1).sum())
    o = ob.map_batch(oidx, oibf, codes, offs, er, threads=THREADS)
    _log("config3: oracle on %d reads in %.1f s" % (N, time.time() - t0))
    assert_all_fields_equal(dev, o[:3])
    assert st["items"] == o[3]["items"] and st["raw_matches"] == o[3]["raw_matches"]
    m = dev[0]["mapped"] == 1
    assert m.mean() > 0.99 and (dev[0]["primary"]["contig"][m] == truth["bin"][m]).mean() > 0.9999


def test_config4_geometry_paired_end(dy, ob, geometry256):
    """Parity-unpinned half: the reference holds no paired-end fixture; the oracle restates src/mapper.h:925-1172."""
    from dream_yara_amd import synth
    wl, genomes, ctx, ibf, index, oibf, oidx = geometry256
    P = 100_000
    codes, offs, truth = synth.simulate_pairs(genomes, P, 100, seed=0x5EED0004, sub_hi=0.02)
    rng = np.random.default_rng(4)
    for i in rng.integers(P, 2 * P, P // 50):                      # hopeless mates: work for the rescue
        codes[int(offs[i]):int(offs[i + 1])] = rng.integers(0, 4, 100)
    er = np.float32(0.03)
    reads = dy.Reads(ctx, codes, offs)
    ibf.query(reads, dy.threshold(er, wl["k"], 100), want_masks=False)
    res = dy.map_reads(ctx, index, reads, dy.default_params(error_rate=er, paired=1))
    dev = (res.reads(), res.matches(), res.cigars())
    res.close(); reads.close()
    t0 = time.time()
    o = ob.map_batch(oidx, oibf, codes, offs, er, paired=True, threads=THREADS)
    _log("config4: oracle on %d pairs in %.1f s" % (P, time.time() - t0))
    assert_all_fields_equal(dev, o[:3], paired=True)
    o_det = ob.map_batch(oidx, oibf, codes, offs, er, paired=True, threads=THREADS, det_math=True)[0]
    assert np.array_equal(dev[0]["pair_prob"], o_det["pair_prob"])      # bit-identical when the oracle runs the device's sequences
    assert dev[0]["paired"].mean() > 0.9 and (dev[0]["cigar_len"] > 1).sum() > 100


# ------------------------------------------------------------------------------------------------- configs[4] geometry: 1024 bins / 64 GiB, the whole path
def test_config5_geometry_full_path(dy, ob):
    """BASELINE configs[4] on one GPU: 1024 bins x 4 Mbp with all FM indices resident (q = 11 q-mer table, u16 work items, radix-sorted
    work list), the real 64 GiB filter (W = 16, k_ibf_query_eager2), 150 bp reads at -e 5 (threshold 7; 8 seeds of 18 + 6-base tail, the
    16-lane alignment of wide bands).  K1 masks, every result field, items and raw matches of 200 k reads equal the oracle's; the
    own-bin property holds on the whole 1 M batch."""
    from dream_yara_amd import synth
    wl, genomes, idx_dir = _inputs("config5")
    ctx = dy.Context(0)
    try:
        ibf = _device_filter(dy, ctx, wl, genomes)
        info = ibf.info()
        assert info["words_per_pos"] == 16 and info["n_bits"] == (64 << 33) + 256
        t0 = time.time()
        index = dy.Index(ctx, idx_dir, wl["bins"])
        _log("config5: 1024 FM indices uploaded in %.1f s" % (time.time() - t0))
        N = 1_000_000
        codes, offs, truth = synth.simulate_reads(genomes, N, 150, seed=0x5EED0005)
        del genomes
        er = np.float32(0.05)
        thr = dy.threshold(er, wl["k"], 150)
        assert thr == 7
        reads = dy.Reads(ctx, codes, offs)
        d_masks = ibf.query(reads, thr)
        assert ctx.stats()["ibf_kernel"] == 5                        # k_ibf_query_eager2
        reads.close()
        # property over the whole batch: a read with at most 7 errors keeps >= 7 error-free 18-mers, so its own bin is selected
        own = (d_masks[np.arange(N), truth["bin"] // 64] >> (truth["bin"] % 64).astype(np.uint64)) & np.uint64(1)
        assert own.mean() > 0.9999, own.mean()
        # the whole path on the first 200 k reads (a smaller batch gives the same mask words: no dependence on batch geometry)
        S = 200_000
        c, o_ = codes[: int(offs[S])], offs[: S + 1]
        reads = dy.Reads(ctx, c, o_)
        assert np.array_equal(ibf.query(reads, thr), d_masks[:S])
        res = dy.map_reads(ctx, index, reads, dy.default_params(error_rate=er))
        dev = (res.reads(), res.matches(), res.cigars()); st = ctx.stats()
        res.close(); reads.close(); index.close()
        oibf = _oracle_filter(ob, ibf)
        ibf.close()
        t0 = time.time()
        o_masks = oibf.query(c, o_, thr, threads=THREADS)
        assert np.array_equal(d_masks[:S], o_masks), "K1 masks differ at %d reads" % int((d_masks[:S] != o_masks).any(axis=1).sum())
        sel = np.unpackbits(o_masks.view(np.uint8), axis=1).sum(axis=1)
        assert sel.mean() < 3                                         # the filter discriminates (≈1 bin per read + a few false candidates)
        oidx = ob.OracleIndex(idx_dir, wl["bins"])
        o = ob.map_batch(oidx, oibf, c, o_, er, threads=THREADS)
        _log("config5: oracle (K1 + mapping) on %d reads in %.1f s" % (S, time.time() - t0))
        oidx.close(); oibf.close()
        assert_all_fields_equal(dev, o[:3])
        assert st["items"] == o[3]["items"] and st["raw_matches"] == o[3]["raw_matches"]
        m = dev[0]["mapped"] == 1
        assert m.mean() > 0.99 and (dev[0]["primary"]["contig"][m] == truth["bin"][:S][m]).mean() > 0.9999
        assert (dev[0]["primary"]["errors"][m] > 3).sum() > 100      # wide bands met
    finally:
        ctx.close()


# ------------------------------------------------------------------------------------------------- the north-star sentence: 64 bins / 16 GiB
def test_northstar_geometry(dy, ob):
    """BASELINE.json's target workload on one GPU: 64 bins, a 16 GiB filter, one resident batch of 2 M x 100 bp reads (the 10 M reads are 5
    such batches; a batch holds at most 2 097 151 reads).  At this fill (0.6 %) k_ibf_query_two takes its first check early (slack 4) and
    requests no tail k-mers with the first round (the other branch of the kernel from config 2's).  Masks and every field == oracle."""
    from dream_yara_amd import synth
    import bench
    wl = dict(bench.WORKLOADS["northstar"])
    _, genomes, idx_dir = _inputs("config2")                        # the same 64 genomes and indices as configs[1]
    ctx = dy.Context(0)
    try:
        ibf = _device_filter(dy, ctx, wl, genomes)
        assert ibf.info()["words_per_pos"] == 1 and ibf.info()["n_bits"] == (16 << 33) + 256
        index = dy.Index(ctx, idx_dir, wl["bins"])
        N = wl["batch"]
        codes, offs, truth = synth.simulate_reads(genomes, N, wl["L"], seed=0x5EED0002 + 7919)
        er = np.float32(wl["e"] / 100.0)
        thr = dy.threshold(er, wl["k"], wl["L"])
        reads = dy.Reads(ctx, codes, offs)
        d_masks = ibf.query(reads, thr)
        st = ctx.stats()
        assert st["ibf_kernel"] == 1 and st["ibf_tail_pre"] == 0 and 1 <= st["ibf_slack"] < 14, st      # config 2 (1 GiB): tail_pre 4, slack >= 14
        res = dy.map_reads(ctx, index, reads, dy.default_params(error_rate=er))
        dev = (res.reads(), res.matches(), res.cigars()); st = ctx.stats()
        res.close(); reads.close(); index.close()
        oibf = _oracle_filter(ob, ibf)
        ibf.close()
        t0 = time.time()
        o_masks = oibf.query(codes, offs, thr, threads=THREADS)
        assert np.array_equal(d_masks, o_masks), "K1 masks differ at %d reads" % int((d_masks != o_masks).any(axis=1).sum())
        oidx = ob.OracleIndex(idx_dir, wl["bins"])
        o = ob.map_batch(oidx, oibf, codes, offs, er, threads=THREADS)
        _log("northstar: oracle on %d reads in %.1f s" % (N, time.time() - t0))
        oidx.close(); oibf.close()
        assert_all_fields_equal(dev, o[:3])
        assert st["items"] == o[3]["items"] and st["raw_matches"] == o[3]["raw_matches"]
        m = dev[0]["mapped"] == 1
        assert m.mean() > 0.99 and (dev[0]["primary"]["contig"][m] == truth["bin"][m]).mean() > 0.9999
    finally:
        ctx.close()


# ------------------------------------------------------------------------------------------------- configs[2] / [3] partitioning at their bin count
@pytest.mark.parametrize("paired", [False, True])
def test_bins_sharded_two_ranks_at_256_bins(dy, ob, tmp_path, paired):
    """BASELINE configs[2] / [3] shard 256 bins over the ranks.  Two ranks on this GPU (exchanges over gloo) each hold the FM structures of
    128 of the 256 x 4 Mbp bins (W = 4 mask words per read); every field of every read equals the single-GPU run on the whole job (which
    test_config3_geometry_single_end / test_config4_geometry_paired_end tie to the oracle).  The filter is a 1 GiB one here (each worker
    process loads it from a file); the partitioning does not depend on the filter's size."""
    from dream_yara_amd import synth
    from test_gpu_parity import _run_sharded_workers
    wl, genomes, idx_dir = _inputs("config3")
    ctx = dy.Context(0)
    ibf = dy.Ibf.create(ctx, wl["bins"], wl["h"], wl["k"], (1 << 33) + 256)
    for b, g in enumerate(genomes):
        ibf.add(g, b)
    ibf.save(str(tmp_path / "f.filter")); ibf.close(); ctx.close()
    os.symlink(idx_dir.rstrip("/"), str(tmp_path / "idx"))
    if paired:
        codes, offs, _ = synth.simulate_pairs(genomes, 20_000, 100, seed=0x5EED0004 + 1, sub_hi=0.02)
    else:
        codes, offs, _ = synth.simulate_reads(genomes, 60_000, 100, seed=0x5EED0003 + 1)
    del genomes
    np.savez(str(tmp_path / "reads.npz"), codes=codes, offs=offs)
    outs = _run_sharded_workers(tmp_path, wl["bins"], wl["k"], 0.0, 29631 + int(paired), ["paired"] if paired else [])
    n = offs.size - 1
    assert sum(o["n"] for o in outs) == n and sum(o["mapped"] for o in outs) > 0.95 * n
    assert all(o["recv_reads"] > 0 and o["raw_in"] > 0 for o in outs)
    assert sum(o["sent_reads"] for o in outs) == sum(o["recv_reads"] for o in outs) >= 0.99 * n      # a read may select no bin at all
    if paired:
        assert all(o["pe_items_in"] > 0 for o in outs) and sum(o["paired_reads"] for o in outs) > 0.9 * n

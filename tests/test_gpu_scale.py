"""GPU parity at the geometries BASELINE.json quotes (configs[1]..[4]): the HIP path through the C ABI against the CPU oracle
on the same seeded inputs, every field of every read.  The inputs are the bench's own (bench.WORKLOADS, dream_yara_amd/synth.py):

  config2   64 bins x 4 Mbp,  1 GiB IBF,  1 M x 100 bp SE, e=3     full size: masks, results, item / raw-match counts
  config3  256 bins x 4 Mbp, 16 GiB IBF,  250 k x 100 bp SE, e=3   real filter size (W = 4), all 256 FM indices resident
  config4  256 bins, 16 GiB IBF, 100 k pairs 2 x 100 bp, e=3       pairing + rescue (the oracle's PE half is parity-unpinned: no PE gold exists)
  config5 1024 bins x 4 Mbp, 64 GiB IBF, 150 bp, e=5 (threshold 7) K1 masks on a read sample against the oracle running on a host copy of
                                                                   the device-resident filter + the own-bin property on the whole batch;
                                                                   full results at L=150 / e=5 (8 seeds of 18 + 6-base tail) on the 64-bin index

A batch of 10 M / 50 M reads is that many independent repetitions of these steps (reads never interact outside a pair), so the
per-batch comparisons are the parity statement for the full configurations; the multi-GPU partitioning of the same work is covered
by test_bin_sharded_two_ranks_equal_single_gpu.  Index directories are cached under the bench's work dir and reused by bench.py.
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


# ------------------------------------------------------------------------------------------------- configs[4] geometry: K1 at 1024 bins / 64 GiB
def test_config5_geometry_ibf_sweep(dy, ob):
    from dream_yara_amd import synth
    import bench
    wl = bench.WORKLOADS["config5"]
    t0 = time.time()
    genomes = [synth.make_genome(b, wl["glen"]) for b in range(wl["bins"])]
    _log("config5: 1024 genomes in %.1f s" % (time.time() - t0))
    ctx = dy.Context(0)
    try:
        ibf = _device_filter(dy, ctx, wl, genomes)
        info = ibf.info()
        assert info["words_per_pos"] == 16 and info["n_bits"] == (64 << 33) + 256
        N = 1_000_000
        codes, offs, truth = synth.simulate_reads(genomes, N, 150, seed=0x5EED0005)
        er = np.float32(0.05)
        thr = dy.threshold(er, wl["k"], 150)
        assert thr == 7
        reads = dy.Reads(ctx, codes, offs)
        d_masks = ibf.query(reads, thr)
        reads.close()
        # property over the whole batch: a read with at most 7 errors keeps >= 7 error-free 18-mers, so its own bin is selected
        own = (d_masks[np.arange(N), truth["bin"] // 64] >> (truth["bin"] % 64).astype(np.uint64)) & np.uint64(1)
        assert own.mean() > 0.9999, own.mean()
        # a second, smaller sweep gives the same words for the same reads (no dependence on batch geometry)
        S = 30_000
        reads = dy.Reads(ctx, codes[: int(offs[S])], offs[: S + 1])
        assert np.array_equal(ibf.query(reads, thr), d_masks[:S])
        reads.close()
        # exact: the oracle on a host copy of the device-resident 64 GiB filter, on a sample of the batch
        oibf = _oracle_filter(ob, ibf)
        ibf.close()
        t0 = time.time()
        o_masks = oibf.query(codes[: int(offs[S])], offs[: S + 1], thr, threads=THREADS)
        _log("config5: oracle K1 on %d reads in %.1f s" % (S, time.time() - t0))
        oibf.close()
        assert np.array_equal(d_masks[:S], o_masks), "K1 masks differ at %d reads" % int((d_masks[:S] != o_masks).any(axis=1).sum())
        sel = np.unpackbits(o_masks.view(np.uint8), axis=1).sum(axis=1)
        assert sel.mean() < 3                                         # the filter discriminates (≈1 bin per read + a few false candidates)
    finally:
        ctx.close()

"""CPU: the oracle's paired-end half evaluates the pairing weights with libm (std::pow, std::erfc) exactly as the reference does
(src/bits_matches.h:898-966).  The device cannot call libm and uses fixed sequences of IEEE + - * / (dream_yara_amd/csrc/dy_detmath.h,
mirrored for the checker in oracle/dyo_detmath.hpp, Options::detMath).  This test shows on every paired-end test set that the two
evaluations lead to the same pairing: identical paired flags, primaries, X0/X1, MAPQ, matchesSet and CIGARs; the pair probability itself
may differ in its last bits only.  Paired-end output stays PARITY-UNPINNED: the reference ships no paired fixture."""
import os

import numpy as np
import pytest

from test_gpu_parity import _make_pairs, _repeat_genomes


def _index(ob, tmp_path, genomes, with_filter, k=16):
    idx_dir = str(tmp_path / "idx"); os.makedirs(idx_dir)
    for b, g in enumerate(genomes):
        ob.build_bin_index(os.path.join(idx_dir, str(b)), g, np.array([0, g.size], np.uint64))
    fpath = None
    if with_filter:
        fpath = str(tmp_path / "f.filter")
        goffs = np.zeros(len(genomes) + 1, np.uint64); goffs[1:] = np.cumsum([g.size for g in genomes])
        ob.build_filter(fpath, len(genomes), 3, k, 1 << 24, np.concatenate(genomes), goffs)
    return idx_dir, fpath


def _both(ob, idx_dir, n_bins, fpath, codes, offs, er, sr, **kw):
    oidx = ob.OracleIndex(idx_dir, n_bins)
    oibf = ob.OracleIbf(fpath) if fpath else None
    a = ob.map_batch(oidx, oibf, codes, offs, er, sr, paired=True, threads=8, **kw)
    b = ob.map_batch(oidx, oibf, codes, offs, er, sr, paired=True, threads=8, det_math=True, **kw)
    oidx.close()
    if oibf:
        oibf.close()
    return a, b


def _assert_same_pairing(a, b):
    ra, rb = a[0], b[0]
    for f in ("mapped", "best_count", "sub_count", "mapq", "cigar_len", "set_begin", "cigar_begin", "paired"):
        assert np.array_equal(ra[f], rb[f]), f
    assert ra["primary"].tobytes() == rb["primary"].tobytes()
    assert a[1].tobytes() == b[1].tobytes() and np.array_equal(a[2], b[2])
    np.testing.assert_allclose(ra["pair_prob"], rb["pair_prob"], rtol=1e-12, atol=0)
    return float(np.max(np.abs(ra["pair_prob"] - rb["pair_prob"]) / np.maximum(ra["pair_prob"], 1e-300)))


@pytest.mark.parametrize("use_filter,ll,ld,verify,strata", [(False, 0, 0, True, 0), (True, 0, 0, True, 2), (True, 320, 30, True, 0), (False, 0, 0, False, 0)])
def test_libm_and_det_pairing_agree(oracle_build, tmp_path, use_filter, ll, ld, verify, strata):
    import oracle_binding as ob
    rng = np.random.default_rng(77 + ll + strata)               # the inputs of test_paired_end_matches_oracle
    genomes = [rng.integers(0, 4, 50000, dtype=np.uint8) for _ in range(5)]
    genomes[1][3000:3400] = genomes[1][9000:9400]
    idx_dir, fpath = _index(ob, tmp_path, genomes, use_filter)
    codes, offs = _make_pairs(rng, genomes, 1200, 100)
    a, b = _both(ob, idx_dir, 5, fpath, codes, offs, np.float32(0.03), np.float32(strata / 100.0), library_length=ll, library_dev=ld, verify=verify)
    assert a[0]["paired"].sum() > 1500
    _assert_same_pairing(a, b)


def test_libm_and_det_pairing_agree_on_repeat_families(oracle_build, tmp_path):
    import oracle_binding as ob
    from dream_yara_amd import synth
    rng = np.random.default_rng(4242)                           # the inputs of test_paired_end_on_repeat_families_matches_oracle
    genomes, _ = _repeat_genomes(rng, 4, 60000)
    genomes[1] = np.concatenate([genomes[1][:1000], genomes[1][33000:]])
    idx_dir, _ = _index(ob, tmp_path, genomes, False)
    codes, offs, _ = synth.simulate_pairs(genomes, 1500, 100, seed=11, insert_mean=330.0, insert_sd=25.0)
    a, b = _both(ob, idx_dir, 4, None, codes, offs, np.float32(0.03), np.float32(0.0))
    assert (a[0]["best_count"] > 100).sum() > 100               # hundreds of co-optimal locations: many near-equal pair weights
    _assert_same_pairing(a, b)

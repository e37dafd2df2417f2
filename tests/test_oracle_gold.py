"""Pins the CPU oracle against every fixture the reference's own tests hold:
the 768 gold index files (byte equality) and the three gold SAMs (record equality, @PG normalised)."""
import filecmp
import os
import subprocess

import pytest

from conftest import GOLD_INDEX, GENOMES_FA, READS_FA, GOLDEN, sam_records


def test_oracle_indexer_reproduces_gold_bytes(oracle_build, tmp_path):
    subprocess.check_call([os.path.join(oracle_build, "oracle_indexer"), "--split", GENOMES_FA, str(tmp_path)])
    names = sorted(os.listdir(GOLD_INDEX))
    assert len(names) == 768
    bad = [n for n in names if not filecmp.cmp(os.path.join(GOLD_INDEX, n), str(tmp_path / n), shallow=False)]
    assert bad == []


@pytest.mark.parametrize("gold,extra", [("t1", []), ("tag.t1", ["-sm", "tag", "-s", "10"]), ("rec.t1", ["-sm", "record", "-s", "10"])])
def test_oracle_mapper_reproduces_gold_sam(oracle_build, small_filter, tmp_path, gold, extra):
    out = str(tmp_path / "out.sam")
    subprocess.check_call([os.path.join(oracle_build, "oracle_mapper"), "-fi", small_filter, GOLD_INDEX + "/", READS_FA,
                           "-o", out, "-e", "3", "--threads", "1"] + extra)
    mine = sam_records(out)
    ref = sam_records(os.path.join(GOLDEN, "gold", "64-viral-reads.%s.sam" % gold))
    assert len(mine) == len(ref) == 6400 + 64 + 2
    diff = [(a, b) for a, b in zip(mine, ref) if a != b]
    assert diff == []


def test_oracle_without_filter_gives_same_records(oracle_build, tmp_path):
    # -ft none sends every read to every bin; the IBF only prunes bins without a match within the error budget
    out = str(tmp_path / "out.sam")
    subprocess.check_call([os.path.join(oracle_build, "oracle_mapper"), "-ft", "none", GOLD_INDEX + "/", READS_FA,
                           "-o", out, "-e", "3", "--threads", "4"])
    mine = sam_records(out)
    ref = sam_records(os.path.join(GOLDEN, "gold", "64-viral-reads.t1.sam"))
    diff = [(a, b) for a, b in zip(mine, ref) if a != b]
    assert len(diff) <= 2, diff[:3]      # CIGAR provenance across bins may differ; coordinates may not
    for a, b in diff:
        fa, fb = a.split("\t"), b.split("\t")
        assert fa[:5] == fb[:5] and fa[11:] == fb[11:]


def test_oracle_seed_searches_are_complete(oracle_build, tmp_path):
    """The oracle's two approximate seed searches against brute force on a small text: the Hamming backtracking reports exactly the
    positions whose window is within E substitutions of the seed (N never matches); the edit-distance backtracking of -y full reports
    exactly the set of END positions of text substrings within E edits of the seed — the anchor set the argument in DESIGN.md section 4
    rests on (each anchor with the smallest error count among its reports equal to the true minimum)."""
    import numpy as np
    import oracle_binding as ob
    rng = np.random.default_rng(12)
    text = rng.integers(0, 4, 1500, dtype=np.uint8)
    text[300:340] = text[100:140]                       # a repeat
    text[700:720] = 0                                   # a homopolymer run
    ob.build_bin_index(str(tmp_path / "0"), text, np.array([0, text.size], np.uint64))
    idx = ob.OracleIndex(str(tmp_path) + "/", 1)
    n = text.size
    for trial in range(12):
        L = int(rng.integers(12, 22))
        s = int(rng.integers(0, n - L))
        seed = text[s:s + L].copy()
        for _ in range(int(rng.integers(0, 3))):
            seed[int(rng.integers(0, L))] = rng.integers(0, 4)
        if trial == 5:
            seed[3] = 4                                 # an N in the seed: always an error
        for E in (1, 2):
            # Hamming
            got = ob.seed_occurrences(idx, 0, seed, E, edit=False)
            want = {}
            for p in range(L, n + 1):
                d = int(np.sum((text[p - L:p] != seed) | (seed > 3)))
                if d <= E:
                    want[p] = d
            assert {int(r[1]): int(r[2]) for r in got} == want and len(got) == len(want)
            # edit distance: D[i][j] = ed(seed[0..i), text[b..j)) minimised over b (free start): column recurrence over the text
            got = ob.seed_occurrences(idx, 0, seed, E, edit=True)
            best = {}
            for r in got:
                best[int(r[1])] = min(best.get(int(r[1]), 99), int(r[2]))
            prev = np.arange(L + 1)
            want = {}
            for j in range(1, n + 1):
                cur = np.empty(L + 1, dtype=np.int64); cur[0] = 0
                c = text[j - 1]
                for i in range(1, L + 1):
                    cur[i] = min(prev[i - 1] + (0 if (seed[i - 1] == c and c < 4) else 1), prev[i] + 1, cur[i - 1] + 1)
                if cur[L] <= E:
                    want[j] = int(cur[L])
                prev = cur
            assert best == want, (trial, E)
    idx.close()

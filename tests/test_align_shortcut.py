"""The device skips the banded DP for a 2-substitution match of full read length when alignIsUngapped2
(dream_yara_amd/csrc/finalize_kernels.h) holds.  This test restates that rule and checks it against the oracle's
alignMatch on random and adversarial (low-complexity, shifted) inputs: whenever the rule fires, the DP's answer is
'<m>M' without clipping; and the rule is not vacuous."""
import numpy as np

import oracle_binding as ob


def _eq(a, b):
    return (a == b) & (a < 4)


def lcp_f(c, r):
    n = min(c.size, r.size)
    ok = _eq(c[:n], r[:n])
    bad = np.nonzero(~ok)[0]
    return int(bad[0]) if bad.size else n


def lcp_b(c, r):
    return lcp_f(c[::-1], r[::-1])


def rule_ungapped2(ctg, seq):
    m = seq.size
    if int((~_eq(ctg[:m], seq)).sum()) != 2:
        return False
    F, B = lcp_f(ctg, seq), lcp_b(ctg[:m], seq)
    LB1, LB2 = lcp_b(ctg[:m - 1], seq), lcp_b(ctg[:m - 2], seq)
    LF1, LF2 = lcp_f(ctg[1:m], seq), lcp_f(ctg[2:m], seq)
    if F + LB1 >= m - 1 or F + LB2 >= m - 2:
        return False
    if LF1 + B >= m - 1 or LF2 + B >= m - 2:
        return False
    if LF1 + LB1 >= m - 2:
        return False
    return True


def test_ungapped_rule_is_sound_against_the_dp(oracle_build):
    rng = np.random.default_rng(2024)
    fired = checked = gapped_seen = 0
    for it in range(6000):
        m = int(rng.integers(12, 40))
        kind = it % 6
        alpha = 4 if kind < 2 else (2 if kind < 4 else 1 + it % 3)
        ctg = rng.integers(0, alpha, m, dtype=np.uint8)
        if kind == 5:        # homopolymer runs with a boundary: the classic "one insertion, drop the last base" case
            ctg[:] = 0; p = int(rng.integers(2, m - 2)); ctg[p:] = rng.integers(0, 2); ctg[-1] = 3
        read = ctg.copy()
        if it % 3 == 0:      # a read that really is the window with one inserted base (shifted tail)
            p = int(rng.integers(0, m - 1))
            read = np.concatenate([ctg[:p], rng.integers(0, 4, 1, dtype=np.uint8), ctg[p:m - 1]])
        elif it % 3 == 1:    # two inserted bases
            p = int(rng.integers(0, m - 2))
            read = np.concatenate([ctg[:p], rng.integers(0, 4, 2, dtype=np.uint8), ctg[p:m - 2]])
        while int((~_eq(ctg, read)).sum()) < 2:
            q = int(rng.integers(0, m)); read[q] = (read[q] + 1 + rng.integers(0, 3)) & 3
        if it % 50 == 0:
            read[int(rng.integers(0, m))] = 4          # N never matches
        if int((~_eq(ctg, read)).sum()) != 2:
            continue
        checked += 1
        cigar, cb, ce = ob.align_match(ctg, read, 2)
        plain = cigar == [(m << 4) | 0] and cb == 0 and ce == m
        gapped_seen += not plain
        if rule_ungapped2(ctg, read):
            fired += 1
            assert plain, (ctg.tolist(), read.tolist(), cigar, cb, ce)
    assert checked > 1500 and fired > 500 and gapped_seen > 50, (checked, fired, gapped_seen)

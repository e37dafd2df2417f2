import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")
GOLD_INDEX = os.path.join(GOLDEN, "gold", "64-viral-binned-indices")
GENOMES_FA = os.path.join(GOLDEN, "input", "64-viral-genomes.fa")
READS_FA = os.path.join(GOLDEN, "input", "64-viral-reads.fa")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_build():
    """Builds the CPU oracle (checker only). Returns the _build directory."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return os.path.join(ROOT, "oracle", "_build")


@pytest.fixture(scope="session")
def product_build():
    """libdyhip.so + host tools must exist in-tree (built by __graft_entry__.build / python -m dream_yara_amd.build)."""
    lib = os.path.join(ROOT, "dream_yara_amd", "csrc", "libdyhip.so")
    if not os.path.exists(lib):
        from dream_yara_amd import build as b
        b.build()
    return os.path.join(ROOT, "dream_yara_amd", "csrc")


@pytest.fixture(scope="session")
def small_filter(oracle_build, tmp_path_factory):
    """8 MiB IBF over the 64 viral bins (k=19, 2 hashes, like the reference test but not 1 GiB), built by the oracle."""
    out = str(tmp_path_factory.mktemp("filter") / "small.filter")
    subprocess.check_call([os.path.join(oracle_build, "oracle_build_filter"), "-b", "64", "-k", "19", "-nh", "2",
                           "--bits", str(1 << 26), "-o", out, "--split", GENOMES_FA])
    return out


def read_fasta_codes(path):
    """FASTA -> (names, codes uint8 array, offs uint64 array) with Dna5 codes A0 C1 G2 T3 N4."""
    import numpy as np
    names, seqs = [], []
    cur = []
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith(">"):
            if names:
                seqs.append("".join(cur))
            names.append(line[1:])
            cur = []
        else:
            cur.append(line)
    seqs.append("".join(cur))
    lut = np.full(256, 4, dtype=np.uint8)
    for ch, v in (("A", 0), ("C", 1), ("G", 2), ("T", 3), ("U", 3)):
        lut[ord(ch)] = v
        lut[ord(ch.lower())] = v
    codes = np.concatenate([lut[np.frombuffer(s.encode(), dtype=np.uint8)] for s in seqs]) if seqs else np.zeros(0, np.uint8)
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(s) for s in seqs])
    return names, codes, offs


def sam_records(path):
    return [l.rstrip("\n") for l in open(path) if not l.startswith("@PG")]

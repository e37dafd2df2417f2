"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/dyhip.h declares,
fails loudly without a device, and the host tools keep the reference's file formats."""
import ctypes
import filecmp
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, GOLD_INDEX, GENOMES_FA


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "dyhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dy_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported(product_build):
    lib = ctypes.CDLL(os.path.join(product_build, "libdyhip.so"))
    declared = _declared_symbols()
    assert len(declared) >= 30
    missing = [s for s in declared if not hasattr(lib, s)]
    assert missing == []


def test_python_binding_covers_header(product_build):
    import dream_yara_amd as dy
    bound = sorted(name for name, _, _ in dy.SYMBOLS)
    assert bound == _declared_symbols()
    assert dy.load_library().dy_abi_version() == 2


def test_no_device_fails_loudly(product_build):
    import dream_yara_amd as dy
    if dy.device_count() > 0:
        pytest.skip("a HIP device is visible")
    with pytest.raises(dy.DyError) as e:
        dy.Context(0)
    assert e.value.code == dy.DY_ENODEVICE
    assert "no CPU fallback" in str(e.value)


def test_threshold_matches_reference_formula(product_build):
    import dream_yara_amd as dy
    # C1 -> 25, C2-4 -> 29, C5 -> 7; and the k(1+e) > L case -> 0
    assert dy.threshold(3 / 100.0, 19, 100) == 25
    assert dy.threshold(3 / 100.0, 18, 100) == 29
    assert dy.threshold(5 / 100.0, 18, 150) == 7
    assert dy.threshold(5 / 100.0, 30, 100) == 0


def test_product_indexer_reproduces_gold_bytes(product_build, tmp_path):
    fasta = tmp_path / "fa"
    fasta.mkdir()
    recs = open(GENOMES_FA).read().split(">")[1:]
    files = []
    for i, r in enumerate(recs):
        p = fasta / ("%d.fna" % i)
        p.write_text(">" + r)
        files.append(str(p))
    out = tmp_path / "idx"
    out.mkdir()
    subprocess.check_call([os.path.join(product_build, "dream_yara_indexer"), "-t", "4", "-o", str(out) + "/"] + files)
    names = sorted(os.listdir(GOLD_INDEX))
    bad = [n for n in names if not filecmp.cmp(os.path.join(GOLD_INDEX, n), str(out / n), shallow=False)]
    assert bad == []


def test_product_indexer_matches_oracle_on_multi_contig_bin(product_build, oracle_build, tmp_path):
    # three contigs with Ns and a repeat in one bin: both suffix sorters must agree on every file
    import oracle_binding as ob
    rng = np.random.default_rng(7)
    contigs = [rng.integers(0, 4, n, dtype=np.uint8) for n in (700, 64, 1299)]
    contigs[0][100:110] = 4
    contigs[2][:300] = contigs[0][:300]
    fa = tmp_path / "0.fna"
    with open(fa, "wb") as f:
        for i, c in enumerate(contigs):
            f.write(b">contig%d extra words\n" % i + np.frombuffer(b"ACGTN", np.uint8)[c].tobytes() + b"\n")
    a = tmp_path / "a"; a.mkdir()
    b = tmp_path / "b"; b.mkdir()
    subprocess.check_call([os.path.join(product_build, "dream_yara_indexer"), "-o", str(a) + "/", str(fa)])
    offs = np.zeros(4, np.uint64); offs[1:] = np.cumsum([c.size for c in contigs])
    ob.build_bin_index(str(b / "0"), np.concatenate(contigs), offs)
    for n in sorted(os.listdir(a)):
        assert filecmp.cmp(str(a / n), str(b / n), shallow=False), n

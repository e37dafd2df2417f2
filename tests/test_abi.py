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
    assert dy.load_library().dy_abi_version() == 3


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


@pytest.mark.parametrize("n_contigs", [255, 256, 300, 3000])
def test_indexers_agree_across_the_contig_id_width_boundary(product_build, oracle_build, tmp_path, n_contigs):
    """Up to 255 contigs a bin is indexed with u8 contig ids (5-byte sa.val records, lf.drp = list of sentinel rows); from 256 on with
    u16 ids (6-byte records) and lf.drp as a Levels bool dictionary over the rows (reference src/d_indexer.cpp:366-383,
    src/misc_types.h:165-188).  Product indexer and oracle agree on every file, and the file sizes follow the layout.
    PARITY UNPINNED: every gold bin holds one contig, and SeqAn's serialisation of the Levels dictionary is inferred from sa.ind."""
    import oracle_binding as ob
    rng = np.random.default_rng(n_contigs)
    contigs = [rng.integers(0, 4, int(rng.integers(40, 200)), dtype=np.uint8) for _ in range(n_contigs)]
    contigs[1][:30] = contigs[0][:30]; contigs[2][5] = 4
    fa = tmp_path / "0.fna"
    with open(fa, "wb") as f:
        for i, c in enumerate(contigs):
            f.write(b">c%d\n" % i + np.frombuffer(b"ACGTN", np.uint8)[c].tobytes() + b"\n")
    a = tmp_path / "a"; a.mkdir()
    b = tmp_path / "b"; b.mkdir()
    subprocess.check_call([os.path.join(product_build, "dream_yara_indexer"), "-o", str(a) + "/", str(fa)])
    offs = np.zeros(n_contigs + 1, np.uint64); offs[1:] = np.cumsum([c.size for c in contigs])
    ob.build_bin_index(str(b / "0"), np.concatenate(contigs), offs)
    for n in sorted(os.listdir(a)):
        if n.startswith("0.rid"):
            continue                                    # the oracle binding names contigs contig<i>
        assert filecmp.cmp(str(a / n), str(b / n), shallow=False), n
    rows = int(offs[-1]) + n_contigs
    samples = os.path.getsize(a / "0.sa.val") // (5 if n_contigs <= 255 else 6)
    assert os.path.getsize(a / "0.sa.val") == samples * (5 if n_contigs <= 255 else 6)
    assert os.path.getsize(a / "0.lf.drp") == (n_contigs * 4 if n_contigs <= 255 else (rows + 31) // 32 * 8)
    # and the oracle loads what was written
    oidx = ob.OracleIndex(str(a) + "/", 1)
    oidx.close()

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


def bam_to_sam_lines(path):
    """Minimal BAM decoder (spec v1) for the test: returns (header text, reference names, SAM-formatted record lines)."""
    import gzip, struct
    raw = open(path, "rb").read()
    # BGZF structure: every member carries the BC extra field and the file ends with the 28-byte EOF block
    assert raw[-28:] == bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    off, n_blocks = 0, 0
    while off < len(raw):
        assert raw[off:off + 4] == b"\x1f\x8b\x08\x04" and raw[off + 12:off + 14] == b"BC"
        off += struct.unpack_from("<H", raw, off + 16)[0] + 1
        n_blocks += 1
    assert off == len(raw)
    data = gzip.decompress(raw)
    assert data[:4] == b"BAM\x01"
    l_text, = struct.unpack_from("<i", data, 4)
    text = data[8:8 + l_text].decode()
    p = 8 + l_text
    n_ref, = struct.unpack_from("<i", data, p); p += 4
    refs = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", data, p); p += 4
        name = data[p:p + l_name - 1].decode(); p += l_name
        l_ref, = struct.unpack_from("<i", data, p); p += 4
        refs.append((name, l_ref))
    lines = []
    while p < len(data):
        bs, = struct.unpack_from("<i", data, p); p += 4
        rid, pos, l_rn, mapq, bin_, n_cig, flag, l_seq, nrid, npos, tlen = struct.unpack_from("<iiBBHHHiiii", data, p)
        q = p + 32
        qname = data[q:q + l_rn - 1].decode(); q += l_rn
        cig = struct.unpack_from("<%dI" % n_cig, data, q); q += 4 * n_cig
        seqb = data[q:q + (l_seq + 1) // 2]; q += (l_seq + 1) // 2
        seq = "".join("=ACMGRSVTWYHKDBN"[(seqb[i // 2] >> (4 if i % 2 == 0 else 0)) & 15] for i in range(l_seq))
        qual = "".join(chr(c + 33) for c in data[q:q + l_seq]); q += l_seq
        tags = []
        end = p + bs
        while q < end:
            tag, typ = data[q:q + 2].decode(), chr(data[q + 2]); q += 3
            if typ == "i":
                tags.append("%s:i:%d" % (tag, struct.unpack_from("<i", data, q)[0])); q += 4
            else:
                assert typ == "Z"
                z = data.index(b"\0", q); tags.append("%s:Z:%s" % (tag, data[q:z].decode())); q = z + 1
        assert q == end
        ref_len = sum(c >> 4 for c in cig if (c & 15) != 1)
        beg, e = pos, (pos + ref_len if ref_len else pos + 1) if pos >= 0 else 0
        def reg2bin(b, e):
            e -= 1
            for sh, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
                if b >> sh == e >> sh: return base + (b >> sh)
            return 0
        assert bin_ == reg2bin(beg, e) & 0xffff
        cigar = "".join("%d%s" % (c >> 4, "MID"[c & 15]) for c in cig) or "*"
        rname = refs[rid][0] if rid >= 0 else "*"
        rnext = "*" if nrid < 0 else ("=" if nrid == rid else refs[nrid][0])
        lines.append("\t".join([qname, str(flag), rname, str(pos + 1), str(mapq), cigar, rnext, str(npos + 1), str(tlen), seq, qual] + tags))
        p = end
    return text, refs, lines, n_blocks

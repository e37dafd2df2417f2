"""Deterministic synthetic inputs for the benchmark and the large-size tests.

Genomes : one contig per bin, i.i.d. uniform ACGT, numpy MT19937 seeded with 0xD4EA0000 + bin.
Reads   : numpy MT19937 seeded with 0x5EED0000 + config; uniform bin, uniform start, strand 50/50; substitutions with a
          per-base rate ramping linearly 0.1 % -> 1.0 % along the read; indels 0.01 % per base (half insertions).
Nothing here is on the mapping path: it only produces Dna5 code arrays (A0 C1 G2 T3 N4), FASTA files and the index
directory by calling the repo's own host tools.
"""
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
ALPHABET = np.frombuffer(b"ACGTN", dtype=np.uint8)


def make_genome(bin_no, length, seed_base=0xD4EA0000):
    rng = np.random.Generator(np.random.MT19937(seed_base + bin_no))
    return rng.integers(0, 4, size=length, dtype=np.uint8)


def revcomp(codes):
    out = codes[::-1].copy()
    acgt = out < 4
    out[acgt] = 3 - out[acgt]
    return out


def simulate_reads(genomes, n_reads, read_len, seed, sub_lo=0.001, sub_hi=0.010, indel_rate=0.0001):
    """Returns (codes, offs, truth) with truth = structured array (bin, start, strand)."""
    rng = np.random.Generator(np.random.MT19937(seed))
    n_bins = len(genomes)
    bins = rng.integers(0, n_bins, size=n_reads, dtype=np.int64)
    lens = np.array([g.size for g in genomes], dtype=np.int64)
    # room for deletions: sample read_len + 8 template bases
    span = read_len + 8
    starts = (rng.random(n_reads) * (lens[bins] - span)).astype(np.int64)
    strands = rng.integers(0, 2, size=n_reads, dtype=np.uint8)
    reads = np.empty((n_reads, span), dtype=np.uint8)
    for b in range(n_bins):
        sel = np.nonzero(bins == b)[0]
        if sel.size:
            idx = starts[sel][:, None] + np.arange(span)[None, :]
            reads[sel] = genomes[b][idx]
    # reverse strand: reverse complement of the template window
    rev = strands == 1
    tmp = reads[rev][:, ::-1]
    reads[rev] = np.where(tmp < 4, 3 - tmp, tmp)
    # indels on a small subset (applied before substitutions, on the read orientation)
    p_indel = indel_rate * read_len
    with_indel = np.nonzero(rng.random(n_reads) < p_indel)[0]
    for i in with_indel:
        pos = int(rng.integers(5, read_len - 5))
        if rng.random() < 0.5:      # insertion of a random base
            reads[i, pos + 1:] = reads[i, pos:-1]
            reads[i, pos] = rng.integers(0, 4)
        else:                        # deletion of one template base
            reads[i, pos:-1] = reads[i, pos + 1:]
    reads = reads[:, :read_len]
    # substitutions: rate ramps along the read
    rate = np.linspace(sub_lo, sub_hi, read_len)[None, :]
    hit = rng.random((n_reads, read_len)) < rate
    shift = rng.integers(1, 4, size=(n_reads, read_len), dtype=np.uint8)
    reads = np.where(hit, (reads + shift) & 3, reads).astype(np.uint8)
    codes = np.ascontiguousarray(reads).reshape(-1)
    offs = (np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len))
    truth = np.zeros(n_reads, dtype=[("bin", "<u4"), ("start", "<i8"), ("strand", "u1")])
    truth["bin"], truth["start"], truth["strand"] = bins, starts, strands
    return codes, offs, truth


def write_fasta(path, name, codes, line=0):
    with open(path, "wb") as f:
        f.write(b">" + name.encode() + b"\n")
        f.write(ALPHABET[codes].tobytes())
        f.write(b"\n")


def build_index_dir(genomes, out_dir, threads=None, names=None):
    """Writes <out_dir>/fasta/<b>.fna and runs the repo's dream_yara_indexer into <out_dir>/ (12 files per bin)."""
    fasta_dir = os.path.join(out_dir, "fasta")
    os.makedirs(fasta_dir, exist_ok=True)
    files = []
    for b, g in enumerate(genomes):
        p = os.path.join(fasta_dir, "%d.fna" % b)
        write_fasta(p, names[b] if names else "synth_bin%d" % b, g)
        files.append(p)
    exe = os.path.join(CSRC, "dream_yara_indexer")
    if not os.path.exists(exe):
        raise RuntimeError("dream_yara_indexer is not built; run python -m dream_yara_amd.build")
    threads = threads or min(32, os.cpu_count() or 1)
    prefix = out_dir if out_dir.endswith("/") else out_dir + "/"
    subprocess.check_call([exe, "-t", str(threads), "-o", prefix] + files)
    return prefix


def reads_fasta(path, codes, offs, prefix="read"):
    with open(path, "wb") as f:
        n = offs.size - 1
        for i in range(n):
            f.write(b">%s.%d\n" % (prefix.encode(), i))
            f.write(ALPHABET[codes[int(offs[i]):int(offs[i + 1])]].tobytes())
            f.write(b"\n")


def simulate_pairs(genomes, n_pairs, read_len, seed, insert_mean=400.0, insert_sd=40.0, sub_lo=0.001, sub_hi=0.010):
    """FR pairs: fragment ~ N(mean, sd) from a uniform bin/position, first mate = fragment start on
    the forward strand, second mate = reverse complement of the fragment end; half of the pairs are flipped.  Returns
    (codes, offs, truth) in the batch layout of a paired run: first mates [0,P), second mates [P,2P)."""
    rng = np.random.Generator(np.random.MT19937(seed))
    n_bins = len(genomes)
    bins = rng.integers(0, n_bins, size=n_pairs, dtype=np.int64)
    lens = np.array([g.size for g in genomes], dtype=np.int64)
    frag = np.clip(np.rint(rng.normal(insert_mean, insert_sd, n_pairs)).astype(np.int64), read_len + 1, None)
    starts = (rng.random(n_pairs) * (lens[bins] - frag - 1)).astype(np.int64)
    m1 = np.empty((n_pairs, read_len), dtype=np.uint8)
    m2 = np.empty((n_pairs, read_len), dtype=np.uint8)
    ar = np.arange(read_len)[None, :]
    for b in range(n_bins):
        sel = np.nonzero(bins == b)[0]
        if sel.size:
            g = genomes[b]
            m1[sel] = g[starts[sel][:, None] + ar]
            tail = g[(starts[sel] + frag[sel] - read_len)[:, None] + ar]
            m2[sel] = np.where(tail[:, ::-1] < 4, 3 - tail[:, ::-1], tail[:, ::-1])
    flip = rng.random(n_pairs) < 0.5          # the fragment came from the reverse strand: the mates swap roles
    a = np.where(flip[:, None], m2, m1); b2 = np.where(flip[:, None], m1, m2)
    rate = np.linspace(sub_lo, sub_hi, read_len)[None, :]
    out = []
    for m in (a, b2):
        hit = rng.random((n_pairs, read_len)) < rate
        shift = rng.integers(1, 4, size=(n_pairs, read_len), dtype=np.uint8)
        out.append(np.where(hit, (m + shift) & 3, m).astype(np.uint8))
    codes = np.ascontiguousarray(np.concatenate(out, axis=0)).reshape(-1)
    offs = (np.arange(2 * n_pairs + 1, dtype=np.uint64) * np.uint64(read_len))
    truth = np.zeros(n_pairs, dtype=[("bin", "<u4"), ("start", "<i8"), ("frag", "<i8"), ("flip", "u1")])
    truth["bin"], truth["start"], truth["frag"], truth["flip"] = bins, starts, frag, flip
    return codes, offs, truth

"""Host-side I/O of dream_yara_mapper without a GPU: the FASTA/FASTQ/gzip/bzip2 parser and the SAM / BGZF-BAM writers
(dream_yara_amd/csrc/host_formats.hpp, bam_writer.hpp) through a small harness compiled here with g++."""
import gzip
import os
import subprocess

import pytest

from conftest import ROOT, bam_to_sam_lines


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("hostio") / "host_io_harness")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-DDY_HAVE_ZLIB", "-o", exe,
                           os.path.join(ROOT, "tests", "host", "host_io_harness.cpp"), "-lz", "-ldl"])
    return exe


def test_reads_parser_handles_fasta_fastq_gzip_and_odd_lines(harness, tmp_path):
    fq = tmp_path / "a.fq"
    fq.write_bytes(b"@r1 first read/1\nACGTNacgtn\n+\nIIII!IIII~\n"
                   b"@r2\r\nTTTT\r\n+r2\r\n@@@@\r\n"          # CRLF, quality line that starts with '@'
                   b"\n@r3\nA\n+\nI\n")                        # blank line between records, 1-base read
    out = subprocess.check_output([harness, "parse", str(fq), "2"], stderr=subprocess.PIPE).decode().splitlines()
    assert out == ["r1 first read/1\tACGTNACGTN\tIIII!IIII!",      # N carries no quality ('!'); '~' is clamped to 6 bits but N wins here
                   "r2\tTTTT\t@@@@",
                   "r3\tA\tI"]
    fa = tmp_path / "b.fa"
    fa.write_bytes(b">s1 desc\nACGT\nNNAC\n\n>s2\nGGGGGGGGGG\n>s3\nRYKM\n")       # wrapped FASTA, IUPAC codes -> N
    out = subprocess.check_output([harness, "parse", str(fa), "100"]).decode().splitlines()
    assert out == ["s1 desc\tACGTNNAC\tIIII!!II", "s2\tGGGGGGGGGG\tIIIIIIIIII", "s3\tNNNN\t!!!!"]
    gz = tmp_path / "a.fq.gz"
    with gzip.open(gz, "wb") as f:
        f.write(fq.read_bytes())
    assert subprocess.check_output([harness, "parse", str(gz), "1"]) == subprocess.check_output([harness, "parse", str(fq), "1000"])
    import bz2
    bz = tmp_path / "a.fq.bz2"
    raw = fq.read_bytes()
    bz.write_bytes(bz2.compress(raw[:40]) + bz2.compress(raw[40:]))          # two concatenated streams, as pbzip2 writes them
    assert subprocess.check_output([harness, "parse", str(bz), "2"]) == subprocess.check_output([harness, "parse", str(fq), "1000"])
    cut = tmp_path / "cut.fq.bz2"
    cut.write_bytes(bz2.compress(raw * 50)[:-20])
    assert subprocess.call([harness, "parse", str(cut), "10"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) == 1
    bad = tmp_path / "bad.txt"
    bad.write_text("hello\n")
    assert subprocess.call([harness, "parse", str(bad), "10"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) == 1


def test_threaded_parse_equals_sequential_parse(harness, tmp_path):
    import random
    rnd = random.Random(5)
    fq, fa = tmp_path / "big.fq", tmp_path / "big.fa"
    with open(fq, "w") as f, open(fa, "w") as g:
        for i in range(40000):
            L = rnd.randint(1, 180)
            seq = "".join(rnd.choice("ACGTN" if rnd.random() < 0.02 else "ACGT") for _ in range(L))
            f.write("@read_%d some description\n%s\n+\n%s\n" % (i, seq, "".join(chr(rnd.randint(33, 126)) for _ in range(L))))
            g.write(">read_%d\n%s\n" % (i, "\n".join(seq[k:k + 60] for k in range(0, L, 60))))
    for path in (fq, fa):
        one = subprocess.check_output([harness, "parse", str(path), "1000000", "1"])
        assert one.count(b"\n") == 40000
        assert subprocess.check_output([harness, "parse", str(path), "1000000", "6"]) == one
        assert subprocess.check_output([harness, "parse", str(path), "17000", "3"]) == one      # several batches


def test_threaded_fastq_path_falls_back_on_irregular_records(harness, tmp_path):
    """The threaded four-line FASTQ path (line index of a window sized from the previous batch) must give the sequential parser's
    batches whatever it meets: blank lines between records, CRLF, '+name' separator lines, reads much longer than the previous
    batch's (window too small), a quality line starting with '@', a last record without its newline."""
    import random
    rnd = random.Random(11)
    path = tmp_path / "odd.fq"
    with open(path, "wb") as f:
        def rec(i, L, eol=b"\n", plus=b"+", blank=False):
            seq = bytes(rnd.choice(b"ACGTN" if rnd.random() < 0.02 else b"ACGT") for _ in range(L))
            qual = bytes(rnd.randint(33, 126) for _ in range(L))
            if i % 97 == 0: qual = b"@" + qual[1:]
            out = b"@odd_%d extra words/1" % i + eol + seq + eol + plus + eol + qual + eol
            return (b"\n" if blank else b"") + out
        for i in range(12000): f.write(rec(i, rnd.randint(30, 60)))
        for i in range(12000, 15000): f.write(rec(i, rnd.randint(30, 60), eol=b"\r\n"))
        for i in range(15000, 18000): f.write(rec(i, rnd.randint(30, 60), plus=b"+odd_%d" % i))
        for i in range(18000, 21000): f.write(rec(i, rnd.randint(30, 60), blank=(i % 500 == 0)))
        for i in range(21000, 26000): f.write(rec(i, rnd.randint(200, 400)))                  # the window of the previous batch is too small
        for i in range(26000, 30000): f.write(rec(i, rnd.randint(30, 60)))
        f.write(rec(30000, 40)[:-1])                                                             # no newline at the end of the file
    one = subprocess.check_output([harness, "parse", str(path), "1000000", "1"])
    assert one.count(b"\n") == 30001
    for per, threads in ((2500, 4), (999, 3), (7000, 8), (1000000, 5)):
        assert subprocess.check_output([harness, "parse", str(path), str(per), str(threads)]) == one


@pytest.mark.parametrize("n,threads", [(0, 1), (50, 1), (4000, 4)])
def test_bam_stream_decodes_to_the_sam_text(harness, tmp_path, n, threads):
    sam, bam = str(tmp_path / "o.sam"), str(tmp_path / "o.bam")
    subprocess.check_call([harness, "write", sam, bam, str(n), str(threads)])
    lines = [l.rstrip("\n") for l in open(sam)]
    text, refs, recs, n_blocks = bam_to_sam_lines(bam)        # also checks the BGZF framing, the EOF block and the bin field
    assert text.splitlines() == [l for l in lines if l.startswith("@")]
    assert [r[0] for r in refs] == ["chrA", "chrB", "a_rather_long_contig_name.1"]
    assert recs == [l for l in lines if not l.startswith("@")]
    assert len(recs) == n
    if n == 4000:
        assert n_blocks > 5                                    # several blocks, deflated by the worker threads

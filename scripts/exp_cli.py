"""End-to-end CLI throughput on the bench workload: FASTQ in, SAM / BAM out."""
import os, sys, time, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dream_yara_amd as dy
from dream_yara_amd import synth
import bench
wl = bench.WORKLOADS["config2"]
work = "/tmp/dy_bench_%d/config2" % os.getuid(); os.makedirs(work, exist_ok=True)
genomes, idx_dir = bench.build_inputs(wl, work, 0, 1, lambda: None)
fpath = os.path.join(work, "cli.filter")
if not os.path.exists(fpath):
    ctx = dy.Context(0)
    ibf = dy.Ibf.create(ctx, wl["bins"], wl["h"], wl["k"], (1 << 33) + 256)
    for b, g in enumerate(genomes): ibf.add(g, b)
    ibf.save(fpath); ibf.close(); ctx.close()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
fq = os.path.join(work, "cli_%d.fq" % n)
if not os.path.exists(fq):
    codes, offs, _ = synth.simulate_reads(genomes, n, 100, seed=99)
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    seqs = lut[codes].reshape(n, 100)
    with open(fq, "wb") as f:
        for i in range(0, n, 100000):
            blk = seqs[i:i + 100000]
            lines = []
            for j in range(blk.shape[0]):
                s = blk[j].tobytes()
                lines.append(b"@r%d\n" % (i + j) + s + b"\n+\n" + b"I" * 100 + b"\n")
            f.write(b"".join(lines))
exe = os.path.join(ROOT, "dream_yara_amd", "csrc", "dream_yara_mapper")
for out in ("out.sam", "out.bam"):
    t0 = time.time()
    subprocess.check_call([exe, "-fi", fpath, idx_dir, fq, "-o", os.path.join(work, out), "-e", "3", "-rb", "1000000", "-v"], stderr=open(os.path.join(work, "cli.err"), "w"))
    dt = time.time() - t0
    print(out, "%.1f s total" % dt, "%.2f M reads/s" % (n / dt / 1e6), "| stderr:", " / ".join(l.strip() for l in open(os.path.join(work, "cli.err")) if "time" in l or "Parse" in l))

"""Device I/O path piece by piece on 1 M reads of the bench workload: FASTQ bytes -> batch, sweep + map, results -> SAM bytes."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import bench
from dream_yara_amd.dist import Group
import dream_yara_amd as dy
class A: pass
args = A(); args.work_dir = "/tmp/dy_bench_%d" % os.getuid(); args.tune = []; args.prefetch = "off"; args.pipeline = "off"
s = bench.Setup("config2", args, Group(None))
codes, offs = s.host_batch0
n, L = 1_000_000, 100
lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
rec = np.empty((n, 14 + 2 * L), dtype=np.uint8)
names = np.char.zfill(np.arange(n).astype(str), 7)
rec[:, 0] = ord("@"); rec[:, 1] = ord("r"); rec[:, 2:9] = np.frombuffer("".join(names).encode(), dtype=np.uint8).reshape(n, 7); rec[:, 9] = 10
rec[:, 10:10 + L] = lut[codes[: n * L]].reshape(n, L); rec[:, 10 + L] = 10; rec[:, 11 + L] = ord("+"); rec[:, 12 + L] = 10
rec[:, 13 + L:13 + 2 * L] = ord("I"); rec[:, 13 + 2 * L] = 10
text = rec.tobytes()
ctx, ibf = s.ctx, s.ibf
index = dy.Index(ctx, s.idx_dir, s.wl["bins"])
T = {}
for it in range(4):
    t0 = time.perf_counter(); reads, used = dy.Reads.from_fastq(ctx, text, n, True); ctx.sync(); t1 = time.perf_counter()
    ibf.query(reads, s.thr, want_masks=False); res = dy.map_reads(ctx, index, reads, s.params, asynchronous=True); ctx.sync(); t2 = time.perf_counter()
    sam, mapped = res.sam(reads, index); t3 = time.perf_counter()
    res.close(); reads.close()
    if it: T.setdefault("from_fastq", []).append(t1 - t0); T.setdefault("map", []).append(t2 - t1); T.setdefault("sam", []).append(t3 - t2)
print({k: round(1e3 * float(np.median(v)), 2) for k, v in T.items()}, "ms;", len(text), "bytes in,", len(sam), "bytes out,", mapped, "mapped")

"""End-to-end CLI throughput against host thread count (-t) on the bench workload: where the tool's time goes."""
import os, sys, time, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (torch's HIP runtime first, see bench.py)
import bench
from dream_yara_amd.dist import Group

class A: pass
args = A(); args.work_dir = "/tmp/dy_bench_%d" % os.getuid(); args.tune = []; args.prefetch = "off"; args.pipeline = "off"
s = bench.Setup("config2", args, Group(None))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
wl = s.wl; L = wl["L"]
codes, offs = s.host_batch0
fpath = os.path.join(s.work_dir, "cli.filter"); s.ibf.save(fpath)
block = 1_000_000
lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
rec = np.empty((block, 14 + 2 * L), dtype=np.uint8)
names = np.char.zfill(np.arange(block).astype(str), 7)
rec[:, 0] = ord("@"); rec[:, 1] = ord("r"); rec[:, 2:9] = np.frombuffer("".join(names).encode(), dtype=np.uint8).reshape(block, 7); rec[:, 9] = 10
rec[:, 10:10 + L] = lut[codes[: block * L]].reshape(block, L); rec[:, 10 + L] = 10; rec[:, 11 + L] = ord("+"); rec[:, 12 + L] = 10
rec[:, 13 + L:13 + 2 * L] = ord("I"); rec[:, 13 + 2 * L] = 10
fq = os.path.join(s.work_dir, "cli_%d.fq" % n)
with open(fq, "wb") as f:
    for _ in range(n // block): f.write(rec.tobytes())
idx_dir = s.idx_dir
s.close()
exe = os.path.join(ROOT, "dream_yara_amd", "csrc", "dream_yara_mapper")
for extra in sys.argv[2:] or ["-t 16", "-t 64"]:
    out = os.path.join(os.path.dirname(fq), "out.sam")
    t0 = time.time()
    subprocess.check_call([exe, "-fi", fpath, idx_dir, fq, "-o", out, "-e", "3", "-rb", "1000000", "-v"] + extra.split(), stderr=open("/tmp/cli.err", "w"))
    dt = time.time() - t0
    print(extra, "| %.2f s total, %.1f M reads/s |" % (dt, n / dt / 1e6), " / ".join(l.strip() for l in open("/tmp/cli.err") if "time" in l or "Parse" in l or "waited" in l or "Since" in l or "Device I" in l), flush=True)
    os.remove(out)
os.remove(fq)

"""Load times of the bench filter and index (what the CLI pays before the first batch)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dream_yara_amd as dy
import bench
wl = bench.WORKLOADS["config2"]
work = "/tmp/dy_bench_%d/config2" % os.getuid(); os.makedirs(work, exist_ok=True)
genomes, idx_dir = bench.build_inputs(wl, work, 0, 1, lambda: None)
fpath = os.path.join(work, "cli.filter")
ctx = dy.Context(0)
if not os.path.exists(fpath):
    ibf = dy.Ibf.create(ctx, wl["bins"], wl["h"], wl["k"], (1 << 33) + 256)
    for b, g in enumerate(genomes): ibf.add(g, b)
    ibf.save(fpath); ibf.close()
for it in range(3):
    t0 = time.perf_counter(); ibf = dy.Ibf.load(ctx, fpath); ctx.sync(); t1 = time.perf_counter()
    idx = dy.Index(ctx, idx_dir, wl["bins"]); ctx.sync(); t2 = time.perf_counter()
    print("filter load %.3f s, index load %.3f s" % (t1 - t0, t2 - t1), flush=True)
    ibf.close(); idx.close()

"""PCIe-inclusive step: upload the batch from host memory, map, copy the results back (not bench.py's `value`)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dream_yara_amd as dy
from dream_yara_amd import synth
import bench
wl = bench.WORKLOADS["config2"]
work = "/tmp/dy_bench_%d/config2" % os.getuid(); os.makedirs(work, exist_ok=True)
genomes, idx_dir = bench.build_inputs(wl, work, 0, 1, lambda: None)
ctx = dy.Context(0)
ibf = dy.Ibf.create(ctx, wl["bins"], wl["h"], wl["k"], (1 << 33) + 256)
for b, g in enumerate(genomes): ibf.add(g, b)
index = dy.Index(ctx, idx_dir, wl["bins"])
codes, offs, truth = synth.simulate_reads(genomes, wl["reads"], wl["L"], seed=0x5EED0002)
er = np.float32(0.03); params = dy.default_params(error_rate=er); thr = dy.threshold(er, wl["k"], wl["L"])
T = {"upload": [], "device": [], "fetch": []}
for it in range(5):
    t0 = time.perf_counter()
    reads = dy.Reads(ctx, codes, offs); t1 = time.perf_counter()
    ibf.query(reads, thr, want_masks=False)
    res = dy.map_reads(ctx, index, reads, params, asynchronous=True); ctx.sync(); t2 = time.perf_counter()
    res.fetch(); t3 = time.perf_counter()
    res.close(); reads.close()
    if it: T["upload"].append(t1 - t0); T["device"].append(t2 - t1); T["fetch"].append(t3 - t2)
m = {k: 1e3 * float(np.median(v)) for k, v in T.items()}
print({k: round(v, 2) for k, v in m.items()}, "total ms", round(sum(m.values()), 2), "reads/s", round(wl["reads"] / sum(m.values()) * 1e3 / 1e6, 1), "M")

"""Experiment: stage times of the config-2 step as a function of the reads' error content (error-free, the default ramp, heavier)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dream_yara_amd as dy
from dream_yara_amd import synth
import bench

wl = bench.WORKLOADS["config2"]
work = "/tmp/dy_bench_%d/config2" % os.getuid()
os.makedirs(work, exist_ok=True)
genomes, idx_dir = bench.build_inputs(wl, work, 0, 1, lambda: None)
er = np.float32(0.03)
ctx = dy.Context(0)
ibf = dy.Ibf.create(ctx, wl["bins"], wl["h"], wl["k"], (1 << 33) + 256)
for b, g in enumerate(genomes):
    ibf.add(g, b)
index = dy.Index(ctx, idx_dir, wl["bins"])
params = dy.default_params(error_rate=er)
thr = dy.threshold(er, wl["k"], wl["L"])
for name, kw in (("error-free", dict(sub_lo=0.0, sub_hi=0.0, indel_rate=0.0)), ("default 0.1%..1.0%", dict()), ("1%..3%", dict(sub_lo=0.01, sub_hi=0.03))):
    codes, offs, truth = synth.simulate_reads(genomes, wl["reads"], wl["L"], seed=0x5EED0002, **kw)
    reads = dy.Reads(ctx, codes, offs)
    acc = {}
    for it in range(6):
        ibf.query(reads, thr, want_masks=False)
        res = dy.map_reads(ctx, index, reads, params, asynchronous=True); ctx.sync()
        st = ctx.stats()
        if it:
            for k in ("ms_ibf", "ms_items", "ms_search", "ms_extend", "ms_finalize"):
                acc[k] = acc.get(k, 0) + st[k] / 5
        res.close()
    print("%-20s %s raw/read %.3f second-pass items/read %.3f (last batch)" % (name, {k: round(v, 3) for k, v in acc.items()}, st["raw_matches"] / wl["reads"],
                                                                                   st["second_pass_items"] / wl["reads"]), flush=True)
    reads.close()

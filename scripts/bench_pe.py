"""Paired-end throughput on the config-2 index (BASELINE config 4 is the same path on 256 bins): P pairs of 2 x 100 bp,
insert N(400, 40), -e 3.  Prints ms per batch, the SE figure on the same 2P reads for comparison, and pairing statistics."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dream_yara_amd as dy
from dream_yara_amd import synth
import bench

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
wl = bench.WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else "config2"]
work = "/tmp/dy_bench_%d/%s" % (os.getuid(), sys.argv[2] if len(sys.argv) > 2 else "config2")
os.makedirs(work, exist_ok=True)
genomes, idx_dir = bench.build_inputs(wl, work, 0, 1, lambda: None)
ctx = dy.Context(0)
ibf = dy.Ibf.create(ctx, wl["bins"], wl["h"], wl["k"], ((wl["ibf_gib"] << 33) if wl["ibf_gib"] else (1 << 28)) + 256)
for b, g in enumerate(genomes):
    ibf.add(g, b)
index = dy.Index(ctx, idx_dir, wl["bins"])
codes, offs, truth = synth.simulate_pairs(genomes, n_pairs, wl["L"], seed=0x5EED0004)
reads = dy.Reads(ctx, codes, offs)
er = np.float32(0.03)
thr = dy.threshold(er, wl["k"], wl["L"])
out = {}
for name, params in (("se", dy.default_params(error_rate=er)), ("pe", dy.default_params(error_rate=er, paired=1))):
    ms = []
    for it in range(4):
        ctx.sync(); t0 = time.perf_counter()
        ibf.query(reads, thr, want_masks=False)
        res = dy.map_reads(ctx, index, reads, params, asynchronous=True)
        ctx.sync(); ms.append(1e3 * (time.perf_counter() - t0))
        st = ctx.stats()
        if it < 3: res.close()
    rr = res.fetch().reads(); res.close()
    out[name] = {"ms_per_batch": float(np.median(ms[1:])), "reads_per_s": 2 * n_pairs / float(np.median(ms[1:])) * 1e3,
                 "stages_ms": {k: st[k] for k in ("ms_ibf", "ms_items", "ms_search", "ms_extend", "ms_finalize")}, "items": st["items"],
                 "mapped": float(rr["mapped"].mean()), "paired": float(rr["paired"].mean())}
    if name == "pe":
        p1, p2 = rr[:n_pairs], rr[n_pairs:]
        both = (p1["paired"] == 1) & (p2["paired"] == 1)
        left = np.minimum(p1["primary"]["begin"].astype(np.int64), p2["primary"]["begin"].astype(np.int64))
        out[name]["pairs_at_true_fragment_start"] = float((both & (p1["primary"]["contig"] == truth["bin"]) & (left == truth["start"])).mean())
    print(name, json.dumps(out[name]), flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "bench_pe.json"), "w"), indent=1)

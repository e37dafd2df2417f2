"""Experiment: do two independent pipelines (two contexts = two streams, two host threads) on ONE GPU overlap usefully?
K1 is gather-bound, K2..K4 are L2 / issue bound."""
import os, sys, time, threading
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
NP = int(sys.argv[1]) if len(sys.argv) > 1 else 2
pipes = []
host_batches = []
for i in range(NP):
    ctx = dy.Context(0)
    ibf = dy.Ibf.create(ctx, wl["bins"], wl["h"], wl["k"], (1 << 33) + 256)
    for b, g in enumerate(genomes):
        ibf.add(g, b)
    index = dy.Index(ctx, idx_dir, wl["bins"])
    codes, offs, truth = synth.simulate_reads(genomes, wl["reads"], wl["L"], seed=0x5EED0002 + 1000 * i)
    reads = dy.Reads(ctx, codes, offs)
    host_batches.append((codes, offs))
    pipes.append((ctx, ibf, index, reads))
params = dy.default_params(error_rate=er)
thr = dy.threshold(er, wl["k"], wl["L"])

PCIE = len(sys.argv) > 2 and sys.argv[2] == "pcie"       # every step uploads its batch from host memory and copies the results back
def run(pipe, steps):
    ctx, ibf, index, reads = pipe
    for _ in range(steps):
        if PCIE:
            codes, offs = host_batches[pipes.index(pipe)]
            reads = dy.Reads(ctx, codes, offs)
        ibf.query(reads, thr, want_masks=False)
        r = dy.map_reads(ctx, index, reads, params, asynchronous=True)
        ctx.sync()
        if PCIE:
            r.fetch(); reads.close()
        r.close()

for p in pipes: run(p, 2)
for n in range(1, NP + 1):
    steps = 10
    ths = [threading.Thread(target=run, args=(pipes[i], steps)) for i in range(n)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    print("pipelines=%d%s: %.2f ms per batch, %.1f M reads/s" % (n, " (upload + map + fetch per step)" if PCIE else "", 1e3 * dt / (steps * n), wl["reads"] * steps * n / dt / 1e6), flush=True)

"""Paired-end on repeat-rich genomes with real mates (planted families, no poly-A run): stage times of a 30 k-pair batch."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dream_yara_amd as dy
from dream_yara_amd import synth
import oracle_binding as ob
from test_gpu_parity import _repeat_genomes
rng = np.random.default_rng(5)
n_bins = 6
genomes, fam = _repeat_genomes(rng, n_bins, 60000)
genomes[1] = np.concatenate([genomes[1][:1000], genomes[1][33000:]])
idx_dir = "/tmp/dy_rep_pe_idx"; os.makedirs(idx_dir, exist_ok=True)
for b, g in enumerate(genomes):
    ob.build_bin_index(os.path.join(idx_dir, str(b)), g, np.array([0, g.size], np.uint64))
P = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
codes, offs, _ = synth.simulate_pairs(genomes, P, 100, seed=3, insert_mean=330.0, insert_sd=25.0)
ctx = dy.Context(0); idx = dy.Index(ctx, idx_dir, n_bins); rd = dy.Reads(ctx, codes, offs)
params = dy.default_params(error_rate=np.float32(0.03), paired=1); params.use_filter = 0
for it in range(3):
    t0 = time.perf_counter()
    res = dy.map_reads(ctx, idx, rd, params, asynchronous=True); ctx.sync()
    dt = time.perf_counter() - t0
    st = ctx.stats(); res.close()
print("pairs", P, "items", st["items"], "hard", st["hard_items"], "raw matches", st["raw_matches"], "ms total %.1f" % (1e3 * dt),
      {k: round(st[k], 2) for k in ("ms_items", "ms_search", "ms_extend", "ms_finalize")})

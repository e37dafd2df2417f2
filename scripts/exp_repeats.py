"""Device time on a repeat-rich batch (planted families with 1 200 copies, a 32 kbp poly-A run): how expensive is the re-seeding path?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dream_yara_amd as dy
import oracle_binding as ob
from test_gpu_parity import _repeat_genomes, _sample_reads
rng = np.random.default_rng(5)
n_bins = 6
genomes, fam = _repeat_genomes(rng, n_bins, 60000)
idx_dir = "/tmp/dy_rep_idx"; os.makedirs(idx_dir, exist_ok=True)
for b, g in enumerate(genomes):
    ob.build_bin_index(os.path.join(idx_dir, str(b)), g, np.array([0, g.size], np.uint64))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
PAIRED = len(sys.argv) > 2 and sys.argv[2] == "paired"      # first half / second half of the batch as mates (unrelated: exercises the path only)
codes, offs = _sample_reads(rng, genomes, fam, N, 100, 3)
if "nopolya" in sys.argv:      # drop the two poly-A reads that _sample_reads appends (about 10^6 matches each)
    offs = offs[:-2]; codes = codes[:int(offs[-1])]; N -= 2
ctx = dy.Context(0); idx = dy.Index(ctx, idx_dir, n_bins); rd = dy.Reads(ctx, codes, offs)
params = dy.default_params(error_rate=np.float32(0.03)); params.use_filter = 0
if PAIRED: params.paired = 1
for it in range(3):
    t0 = time.perf_counter()
    res = dy.map_reads(ctx, idx, rd, params, asynchronous=True); ctx.sync()
    dt = time.perf_counter() - t0
    st = ctx.stats()
    if it == 2:
        import hashlib
        res.fetch()
        digest = hashlib.sha1(res.reads().tobytes() + res.matches().tobytes() + res.cigars().tobytes()).hexdigest()[:16]
    res.close()
print("reads", N + 2, "items", st["items"], "hard", st["hard_items"], "raw matches", st["raw_matches"], "ms total %.1f" % (1e3 * dt),
      {k: round(st[k], 2) for k in ("ms_items", "ms_search", "ms_extend", "ms_finalize")}, "pool retries", st["pool_retries"], "paired" if PAIRED else "single-end", "result sha1", digest)

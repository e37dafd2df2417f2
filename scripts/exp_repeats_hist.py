"""Per-read raw-match counts of the repeat-rich batch of exp_repeats.py (which reads make the aggregation expensive)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import dream_yara_amd as dy
import oracle_binding as ob
from test_gpu_parity import _repeat_genomes, _sample_reads
rng = np.random.default_rng(5)
n_bins = 6
genomes, fam = _repeat_genomes(rng, n_bins, 60000)
idx_dir = "/tmp/dy_rep_idx"; os.makedirs(idx_dir, exist_ok=True)
for b, g in enumerate(genomes):
    ob.build_bin_index(os.path.join(idx_dir, str(b)), g, np.array([0, g.size], np.uint64))
N = 60000
codes, offs = _sample_reads(rng, genomes, fam, N, 100, 3)
n = offs.size - 1
ctx = dy.Context(0); idx = dy.Index(ctx, idx_dir, n_bins); rd = dy.Reads(ctx, codes, offs)
params = dy.default_params(error_rate=np.float32(0.03)); params.use_filter = 0
res = dy.map_reads(ctx, idx, rd, params); res.close()          # fills the all-bins masks
from dream_yara_amd import sharded
a, loads = sharded.raw_matches(ctx, idx, rd, params, n_bins)
n_raw = a.size
cnt = np.bincount(a["read"], minlength=n)
print("raw", n_raw, "reads", n, "max", cnt.max(), "percentiles 50/90/99/99.9:", [int(np.percentile(cnt, p)) for p in (50, 90, 99, 99.9)])
for lo, hi in ((0, 1), (2, 64), (65, 2048), (2049, 4096), (4097, 16384), (16385, 1 << 30)):
    m = (cnt >= lo) & (cnt <= hi)
    print("  reads with %d..%d matches: %d (sum %d)" % (lo, hi, int(m.sum()), int(cnt[m].sum())))

"""Dumps the sorted raw matches of K3 (dy_map_bins through the one-rank packed exchange) on the repeat-rich test data, with the library named by DY_LIB:
    DY_LIB=.../libdyhip_a.so python scripts/dump_raw_matches.py a /tmp/x; DY_LIB=.../libdyhip_b.so python scripts/dump_raw_matches.py b /tmp/x
then compare /tmp/x/raw_a_0.npy with raw_b_0.npy (A/B check of an extension-kernel change, independent of the finalize stage)."""
import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
import torch
import dream_yara_amd as dy
import oracle_binding as ob
from test_gpu_parity import _repeat_genomes, _sample_reads

tag, tmp = sys.argv[1], sys.argv[2]
L, e, strata, sens = 100, 3, 0, 1
rng = np.random.default_rng(L * 1000 + e * 10 + strata)
n_bins = 6
genomes, fam = _repeat_genomes(rng, n_bins, 60000)
idx_dir = os.path.join(tmp, "idx")
if not os.path.exists(idx_dir):
    os.makedirs(idx_dir)
    for b, g in enumerate(genomes):
        ob.build_bin_index(os.path.join(idx_dir, str(b)), g, np.array([0, g.size], np.uint64))
    goffs = np.zeros(n_bins + 1, np.uint64); goffs[1:] = np.cumsum([g.size for g in genomes])
    ob.build_filter(os.path.join(tmp, "f.filter"), n_bins, 3, 14, 1 << 24, np.concatenate(genomes), goffs)
codes, offs = _sample_reads(rng, genomes, fam, 1500, L, max(1, int(L * e / 100)))
ctx = dy.Context(0)
idx = dy.Index(ctx, idx_dir, n_bins)
reads = dy.Reads(ctx, codes, offs)
params = dy.default_params(error_rate=np.float32(e / 100.0), strata_rate=np.float32(strata / 100.0), sensitivity=sens)
ibf = dy.Ibf.load(ctx, os.path.join(tmp, "f.filter"))
n = offs.size - 1
thr = dy.threshold(np.float32(e / 100.0), ibf.info()["kmer_size"], int(int(offs[-1]) // n))
ibf.query(reads, thr, want_masks=False)
from dream_yara_amd import sharded
for rep in range(3):
    a, loads = sharded.raw_matches(ctx, idx, reads, params, n_bins)
    n_raw = a.size
    a = np.sort(a, order=["read", "bin", "contig", "strand", "begin", "len", "errors"])
    np.save(os.path.join(tmp, "raw_%s_%d.npy" % (tag, rep)), a)
    print(tag, rep, "items", int(loads.sum()), "raw", n_raw, flush=True)

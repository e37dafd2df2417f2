"""Paired-end parity at scale: device vs oracle on P pairs of the bench workload (every field of every read)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dream_yara_amd as dy
from dream_yara_amd import synth
import oracle_binding as ob
import bench
P = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000
wl = bench.WORKLOADS["config2"]
work = "/tmp/dy_bench_%d/config2" % os.getuid(); os.makedirs(work, exist_ok=True)
genomes, idx_dir = bench.build_inputs(wl, work, 0, 1, lambda: None)
ctx = dy.Context(0)
ibf = dy.Ibf.create(ctx, wl["bins"], wl["h"], wl["k"], (1 << 33) + 256)
for b, g in enumerate(genomes): ibf.add(g, b)
fpath = os.path.join(work, "pe_check.filter"); ibf.save(fpath)
index = dy.Index(ctx, idx_dir, wl["bins"])
codes, offs, truth = synth.simulate_pairs(genomes, P, 100, seed=4242, sub_hi=0.03)      # more errors than the bench: exercises rescue / DP paths
# a few hopeless mates so that the rescue has work
rng = np.random.default_rng(1)
for i in rng.integers(P, 2 * P, P // 50): codes[int(offs[i]):int(offs[i + 1])] = rng.integers(0, 4, 100)
reads = dy.Reads(ctx, codes, offs)
er = np.float32(0.03)
ibf.query(reads, dy.threshold(er, wl["k"], 100), want_masks=False)
res = dy.map_reads(ctx, index, reads, dy.default_params(error_rate=er, paired=1))
d_reads, d_matches, d_cigars = res.reads(), res.matches(), res.cigars()
oidx = ob.OracleIndex(idx_dir, wl["bins"]); oibf = ob.OracleIbf(fpath)
t0 = time.time()
o_reads, o_matches, o_cigars, ostats = ob.map_batch(oidx, oibf, codes, offs, er, np.float32(0), paired=True, threads=64)
print("oracle PE %.1f s" % (time.time() - t0))
os.remove(fpath)
bad = []
for f in ("mapped", "best_count", "sub_count", "mapq", "cigar_len", "paired", "pair_prob"):
    if not np.array_equal(d_reads[f], o_reads[f]): bad.append(f)
if d_reads["primary"].tobytes() != o_reads["primary"].tobytes(): bad.append("primary")
if d_matches.tobytes() != o_matches.tobytes(): bad.append("matches")
if not np.array_equal(d_cigars, o_cigars): bad.append("cigars")
print("pairs", P, "mapped", float(d_reads["mapped"].mean()), "paired", float(d_reads["paired"].mean()), "rescued-or-gapped cigars", int((d_reads["cigar_len"] > 1).sum()), "MISMATCH" if bad else "IDENTICAL", bad)
sys.exit(1 if bad else 0)

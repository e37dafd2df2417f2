"""K3's two-launch form (extend_mode 1: closed extension sides + second pass) against the one-pass form (extend_mode 2) on many reads of
low-complexity and random genomes at several error budgets: every result field must be equal.  usage (GPU box): python scripts/exp_extend_modes.py [reads]"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dream_yara_amd as dy
from test_gpu_parity import _low_complexity_genome

def build_index(dirname, genomes):
    import subprocess
    fa_dir = os.path.join(dirname, "fa"); os.makedirs(fa_dir)
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    files = []
    for b, g in enumerate(genomes):
        files.append(os.path.join(fa_dir, "%d.fna" % b))
        with open(files[-1], "wb") as f:
            f.write(b">c%d\n" % b + lut[g].tobytes() + b"\n")
    idx = os.path.join(dirname, "idx"); os.makedirs(idx)
    subprocess.check_call([os.path.join(ROOT, "dream_yara_amd", "csrc", "dream_yara_indexer"), "-o", idx + "/"] + files, stdout=subprocess.DEVNULL)
    return idx

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    rng = np.random.default_rng(4242)
    n_bins = 8
    genomes = [_low_complexity_genome(rng, 200000) if b % 2 == 0 else rng.integers(0, 4, 200000, dtype=np.uint8) for b in range(n_bins)]
    with tempfile.TemporaryDirectory() as d:
        idx_dir = build_index(d, genomes)
        for L, e, strata in [(100, 3, 0), (100, 3, 3), (150, 5, 0), (64, 6, 1), (250, 4, 0)]:
            reads = []
            for i in range(n):
                g = genomes[int(rng.integers(0, n_bins))]
                s = int(rng.integers(0, g.size - L - 4))
                r = g[s:s + L + 4].copy()
                for _ in range(int(rng.integers(0, e + 2))):
                    p = int(rng.integers(0, L)); kind = rng.random()
                    if kind < 0.8: r[p] = (r[p] + 1 + rng.integers(0, 3)) & 3
                    elif kind < 0.9: r = np.concatenate([r[:p], rng.integers(0, 4, 1, dtype=np.uint8), r[p:]])
                    else: r = np.concatenate([r[:p], r[p + 1:]])
                r = r[:L]
                if i % 2: r = (3 - r[::-1]).astype(np.uint8)
                reads.append(r.astype(np.uint8))
            codes = np.concatenate(reads); offs = np.zeros(n + 1, np.uint64); offs[1:] = np.cumsum([r.size for r in reads])
            outs = []
            for mode in (1, 2, 0):
                ctx = dy.Context(0)
                ctx.set_tuning("extend_mode", mode)
                idx = dy.Index(ctx, idx_dir, n_bins)
                rd = dy.Reads(ctx, codes, offs)
                params = dy.default_params(error_rate=np.float32(e / 100.0), strata_rate=np.float32(strata / 100.0))
                params.use_filter = 0
                res = dy.map_reads(ctx, idx, rd, params)
                outs.append((res.reads().copy(), res.matches().copy(), res.cigars().copy(), dict(ctx.stats())))
                res.close(); rd.close(); idx.close(); ctx.close()
            ref = outs[1]
            for name, o in (("two launches", outs[0]), ("block-local", outs[2])):
                same = all(np.array_equal(a, b) for a, b in zip(o[:3], ref[:3])) and o[3]["raw_matches"] == ref[3]["raw_matches"]
                print("L=%d e=%d strata=%d %s: %s (mapped %d of %d, raw matches %d, second-pass items %d of %d)" % (
                    L, e, strata, name, "EQUAL" if same else "DIFFERENT", int((ref[0]["mapped"] == 1).sum()), n, ref[3]["raw_matches"], o[3]["second_pass_items"], o[3]["items"]), flush=True)
                if not same:
                    sys.exit(1)

main()

import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, tempfile
import dream_yara_amd as dy
import oracle_binding as ob
import importlib.util
spec = importlib.util.spec_from_file_location("tg", "tests/test_gpu_parity.py"); tg = importlib.util.module_from_spec(spec); spec.loader.exec_module(tg)
rng = np.random.default_rng(77)
n_bins = 5
genomes = [rng.integers(0, 4, 50000, dtype=np.uint8) for _ in range(n_bins)]
genomes[1][3000:3400] = genomes[1][9000:9400]
d = tempfile.mkdtemp()
for b, g in enumerate(genomes): ob.build_bin_index(os.path.join(d, str(b)), g, np.array([0, g.size], np.uint64))
codes, offs = tg._make_pairs(rng, genomes, 1200, 100)
er = np.float32(0.03)
ctx = dy.Context(0); idx = dy.Index(ctx, d, n_bins); reads = dy.Reads(ctx, codes, offs)
params = dy.default_params(error_rate=er, paired=1); params.use_filter = 0
res = dy.map_reads(ctx, idx, reads, params)
dr = res.reads()
oidx = ob.OracleIndex(d, n_bins)
orr = ob.map_batch(oidx, None, codes, offs, er, paired=True)[0]
bad = np.nonzero(dr["pair_prob"] != orr["pair_prob"])[0]
print("n diff", bad.size, "of", dr.size)
for i in bad[:10]:
    print(i, repr(dr["pair_prob"][i]), repr(orr["pair_prob"][i]), dr["paired"][i], dr["mapped"][i], orr["mapped"][i], dr["primary"][i], orr["primary"][i], dr["best_count"][i])
for f in ("mapped", "best_count", "sub_count", "mapq", "cigar_len"):
    print(f, (dr[f] != orr[f]).sum())
print("primary diff", (dr["primary"].tobytes() != orr["primary"].tobytes()))

for f in ("best_count","cigar_len"):
    for i in np.nonzero(dr[f] != orr[f])[0][:6]:
        print(f, i, "dev", dr[i], "ora", orr[i])
dm, om = res.matches(), ob.map_batch(oidx, None, codes, offs, er, paired=True)[1]
i = int(np.nonzero(dr["best_count"] != orr["best_count"])[0][0]) if (dr["best_count"] != orr["best_count"]).any() else None
if i is not None:
    print("dev set", dm[dr["set_begin"][i]:dr["set_begin"][i]+dr["best_count"][i]+dr["sub_count"][i]])
    print("ora set", om[orr["set_begin"][i]:orr["set_begin"][i]+orr["best_count"][i]+orr["sub_count"][i]])
dc, oc = res.cigars(), ob.map_batch(oidx, None, codes, offs, er, paired=True)[2]
for i in np.nonzero(dr["cigar_len"] != orr["cigar_len"])[0][:5]:
    print("cigar", i, dy.cigar_string(dc[dr["cigar_begin"][i]:dr["cigar_begin"][i]+dr["cigar_len"][i]]), dy.cigar_string(oc[orr["cigar_begin"][i]:orr["cigar_begin"][i]+orr["cigar_len"][i]]))

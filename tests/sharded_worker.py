"""Worker of test_gpu_parity.test_bin_sharded_two_ranks_equal_single_gpu: started once per rank by torch.distributed.run.
Every rank maps its own reads through the bin-sharded path (FM structures of `bin mod world == rank` only, exchanges
over gloo staged through the host, both ranks on cuda:0) and compares them with its slice of the single-GPU result on
the concatenated batch."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dream_yara_amd as dy
from dream_yara_amd import sharded
from dream_yara_amd.dist import Group


def main():
    work, n_bins, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    paired = len(sys.argv) > 5 and sys.argv[5] == "paired"
    g = Group("gloo")
    torch.cuda.set_device(0)
    data = np.load(os.path.join(work, "reads.npz"))
    codes, offs = data["codes"], data["offs"]
    n = offs.size - 1
    units = n // 2 if paired else n                      # what is split across ranks: pairs or reads
    cut = [0, units * 2 // 5, units] if g.world == 2 else [round(i * units / g.world) for i in range(g.world + 1)]     # uneven shards
    lo, hi = cut[g.rank], cut[g.rank + 1]

    def take(a, b):
        return codes[int(offs[a]):int(offs[b])], (offs[a:b + 1] - offs[a]).astype(np.uint64)
    if paired:
        # the whole job is [first mates of rank 0, of rank 1, ... | second mates of rank 0, ...]; this rank's batch: its first mates, then its second mates
        P = units
        c1, o1 = take(lo, hi); c2, o2 = take(P + lo, P + hi)
        my_codes = np.concatenate([c1, c2]); my_offs = np.concatenate([o1, o2[1:] + o1[-1]]).astype(np.uint64)
        mine_idx = np.concatenate([np.arange(lo, hi), np.arange(P + lo, P + hi)])
    else:
        my_codes, my_offs = take(lo, hi)
        mine_idx = np.arange(lo, hi)
    er, sr = np.float32(0.03), np.float32(float(sys.argv[4]))
    ctx = dy.Context(0)
    ibf = dy.Ibf.load(ctx, os.path.join(work, "f.filter"))
    params = dy.default_params(error_rate=er, strata_rate=sr, paired=1 if paired else 0)

    sm = sharded.ShardedMapper(ctx, dist, g.rank, g.world, os.path.join(work, "idx"), n_bins, ibf, "cuda")
    reads = dy.Reads(ctx, my_codes, my_offs)
    thr = sm.threshold(er, k, int(my_offs[-1]), my_offs.size - 1)
    res = sm.map(reads, params, thr)
    res.fetch()
    got = (res.reads(), res.matches(), res.cigars())
    info = dict(sm.last)
    res.close(); reads.close(); sm.close()

    # single-GPU reference on the whole batch (every bin resident)
    full = dy.Index(ctx, os.path.join(work, "idx"), n_bins)
    allr = dy.Reads(ctx, codes, offs)
    thr1 = dy.threshold(er, k, int(offs[-1]) // n)
    assert thr1 == thr, (thr1, thr)
    ibf.query(allr, thr1, want_masks=False)
    ref = dy.map_reads(ctx, full, allr, params)
    rr, rm, rc = ref.reads(), ref.matches(), ref.cigars()
    items = ctx.stats()["items"]
    ref.close(); allr.close(); full.close(); ibf.close()

    bad = []
    mine = rr[mine_idx]
    for f in ("primary", "mapped", "best_count", "sub_count", "mapq", "cigar_len", "paired", "pair_prob"):
        if not np.array_equal(got[0][f], mine[f]):
            bad.append(f)
    for i in range(mine_idx.size):
        a, b = got[0][i], mine[i]
        na, nb = int(a["best_count"] + a["sub_count"]), int(b["best_count"] + b["sub_count"])
        if na != nb or not np.array_equal(got[1][int(a["set_begin"]):int(a["set_begin"]) + na], rm[int(b["set_begin"]):int(b["set_begin"]) + nb]):
            bad.append("set%d" % i); break
        if not np.array_equal(got[2][int(a["cigar_begin"]):int(a["cigar_begin"]) + int(a["cigar_len"])],
                              rc[int(b["cigar_begin"]):int(b["cigar_begin"]) + int(b["cigar_len"])]):
            bad.append("cigar%d" % i); break
    if info["items"] != items:
        bad.append("items %d != %d" % (info["items"], items))
    info.update(rank=g.rank, bad=bad, mapped=int(got[0]["mapped"].sum()), n=int(mine_idx.size), paired_reads=int(got[0]["paired"].sum()),
                gapped=int((got[0]["cigar_len"] > 1).sum()))
    open(os.path.join(work, "out%d.json" % g.rank), "w").write(json.dumps(info))
    g.barrier(); g.close(); ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

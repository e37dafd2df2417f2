"""K1-only benchmark over the filter geometries of BASELINE.json's configs (the FM indices are not needed for the IBF
sweep): (bins, GiB, read length, error %) = config 2 (64, 1), north_star target (64, 16), config 3 (256, 16), config 5
(1024, 64, 150 bp, e=5).  Reports the algorithmic GB/s of SURVEY.md §8d (strands*(L-k+1)*h*W*8 B per read) from the
HIP-event time of k_ibf_query, and checks that every read's own bin is selected."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dream_yara_amd as dy
from dream_yara_amd import synth

GEOMS = {"c2": (64, 1, 100, 3), "ns": (64, 16, 100, 3), "c3": (256, 16, 100, 3), "c5": (1024, 64, 150, 5)}
which = sys.argv[1].split(",") if len(sys.argv) > 1 else ["c2", "ns", "c3", "c5"]
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
glen = int(sys.argv[3]) if len(sys.argv) > 3 else 4_000_000
k, h = 18, 3
ctx = dy.Context(0)
out = {}
for name in which:
    bins, gib, L, e = GEOMS[name]
    t0 = time.time()
    ibf = dy.Ibf.create(ctx, bins, h, k, (gib << 33) + 256)
    rng = np.random.default_rng(7)
    src_bins = rng.integers(0, bins, n_reads)
    reads = np.empty((n_reads, L), np.uint8)
    for b in range(bins):
        g = synth.make_genome(b, glen)
        ibf.add(g, b)
        sel = np.nonzero(src_bins == b)[0]
        if sel.size:
            st = rng.integers(0, glen - L, sel.size)
            reads[sel] = g[st[:, None] + np.arange(L)[None, :]]
    rev = rng.random(n_reads) < 0.5
    reads[rev] = 3 - reads[rev][:, ::-1]
    sub = rng.random((n_reads, L)) < 0.005
    reads = np.where(sub, (reads + 1) & 3, reads).astype(np.uint8)
    offs = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(L)
    r = dy.Reads(ctx, reads.reshape(-1), offs)
    er = np.float32(e / 100.0)
    thr = dy.threshold(er, k, L)
    print("[%s] %d bins, %d GiB, L=%d, thr=%d: filter built in %.1f s" % (name, bins, gib, L, thr, time.time() - t0), flush=True)
    W = (bins + 63) // 64
    masks = ibf.query(r, thr)
    own = (masks[np.arange(n_reads), src_bins // 64] >> (src_bins % 64).astype(np.uint64)) & np.uint64(1)
    sel_per_read = float(np.unpackbits(masks.view(np.uint8)).sum()) / n_reads
    ms = []
    for _ in range(5):
        ibf.query(r, thr, want_masks=False); ctx.sync()
        masks2 = ibf.query(r, thr)          # the host copy resolves the event timing
        ms.append(ctx.stats()["ms_ibf"])
    t = float(np.median(ms))
    bytes_per_read = 2 * (L - k + 1) * h * W * 8
    out[name] = {"bins": bins, "ibf_gib": gib, "read_len": L, "threshold": int(thr), "n_reads": n_reads, "ms": t,
                 "algorithmic_bytes_per_read": bytes_per_read, "achieved_GBps": bytes_per_read * n_reads / t / 1e6,
                 "frac_of_8TBps": bytes_per_read * n_reads / t / 1e6 / 8000.0, "reads_per_s": n_reads / t * 1e3,
                 "own_bin_selected": float(own.mean()), "bins_selected_per_read": sel_per_read}
    print(json.dumps({name: out[name]}), flush=True)
    r.close(); ibf.close()
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "bench_ibf.json"), "w"), indent=1)

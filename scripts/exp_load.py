"""Where the tool's loading time goes: context creation, filter file -> device, index files -> device (host parse + device-side tables)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import bench
from dream_yara_amd.dist import Group
import dream_yara_amd as dy
class A: pass
args = A(); args.work_dir = "/tmp/dy_bench_%d" % os.getuid(); args.tune = []; args.prefetch = "off"; args.pipeline = "off"
s = bench.Setup(sys.argv[1] if len(sys.argv) > 1 else "config2", args, Group(None))
fpath = os.path.join(s.work_dir, "cli.filter"); s.ibf.save(fpath)
idx_dir, bins = s.idx_dir, s.wl["bins"]
s.close()
for rep in range(2):
    t0 = time.time(); ctx = dy.Context(0); t1 = time.time()
    ibf = dy.Ibf.load(ctx, fpath); t2 = time.time()
    idx = dy.Index(ctx, idx_dir, bins); t3 = time.time()
    for k, v in (("sparse_sa", 1), ("prefix_q", 0)):
        pass
    print("ctx %.3f s, filter %.3f s, index %.3f s" % (t1 - t0, t2 - t1, t3 - t2), flush=True)
    idx.close(); ibf.close(); ctx.close()
ctx = dy.Context(0); ctx.set_tuning("sparse_sa", 1); ctx.set_tuning("prefix_q", 0); ctx.set_tuning("no_isa", 1)
t0 = time.time(); idx = dy.Index(ctx, idx_dir, bins); print("index without device-only tables %.3f s" % (time.time() - t0))
os.remove(fpath)

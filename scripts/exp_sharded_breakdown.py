"""Where does the bins-sharded step spend its time at N=1 (self all-to-all over RCCL)?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
import dream_yara_amd as dy
from dream_yara_amd import synth, sharded
from dream_yara_amd.dist import Group
import bench
wl = bench.WORKLOADS["config2"]
work = "/tmp/dy_bench_%d/config2" % os.getuid(); os.makedirs(work, exist_ok=True)
grp = Group("nccl", force=True)
genomes, idx_dir = bench.build_inputs(wl, work, 0, 1, lambda: None)
ctx = dy.Context(0)
ibf = dy.Ibf.create(ctx, wl["bins"], wl["h"], wl["k"], (1 << 33) + 256)
for b, g in enumerate(genomes): ibf.add(g, b)
sm = sharded.ShardedMapper(ctx, grp.dist, 0, 1, idx_dir, wl["bins"], ibf, "cuda")
codes, offs, truth = synth.simulate_reads(genomes, wl["reads"], wl["L"], seed=0x5EED0002)
reads = dy.Reads(ctx, codes, offs)
er = np.float32(0.03); params = dy.default_params(error_rate=er); thr = dy.threshold(er, wl["k"], wl["L"])
import types
def timed_map(self, reads, params, threshold):
    T = {}; t = [time.perf_counter()]
    def lap(name):
        torch.cuda.synchronize(); self.ctx.sync(); now = time.perf_counter(); T[name] = T.get(name, 0) + 1e3 * (now - t[0]); t[0] = now
    ctx, dev, W = self.ctx, self.device, self.W
    u8 = lambda n: torch.empty(int(n), dtype=torch.uint8, device=dev)
    self.ibf.query(reads, threshold, want_masks=False); lap("k1")
    n_to, b_to = dy.shard_plan(ctx, reads, self.owner_masks); lap("plan")
    n_tot, b_tot = int(n_to.sum()), int(b_to.sum())
    ids, lens, masks, seqs = u8(n_tot * 4), u8(n_tot * 4), u8(n_tot * W * 8), u8(b_tot); lap("alloc")
    dy.shard_pack(ctx, reads, ids.data_ptr(), lens.data_ptr(), masks.data_ptr(), seqs.data_ptr()); lap("pack")
    r_ids, rs = sharded.exchange_bytes(self.dist, ids, n_to.astype(np.int64) * 4, dev)
    r_lens, _ = sharded.exchange_bytes(self.dist, lens, n_to.astype(np.int64) * 4, dev)
    r_masks, _ = sharded.exchange_bytes(self.dist, masks, n_to.astype(np.int64) * W * 8, dev)
    r_seqs, _ = sharded.exchange_bytes(self.dist, seqs, b_to.astype(np.int64), dev); lap("exchange C1")
    n_recv = r_ids.numel() // 4
    sub = dy.Reads.from_device(ctx, r_seqs.data_ptr(), r_lens.data_ptr(), r_masks.data_ptr(), n_recv, r_seqs.numel(), self.n_bins); lap("reads_from_device")
    loads, n_raw = dy.map_bins(ctx, self.index, sub, params); lap("map_bins (K2+K3)")
    seg = np.zeros(self.world + 1, np.uint32); seg[1:] = np.cumsum([x // 4 for x in rs])
    raw = u8(n_raw * dy.RAW_MATCH_BYTES)
    raw_to = dy.raw_pack(ctx, seg, r_ids.data_ptr(), raw.data_ptr()); lap("raw_pack")
    sub.close(); lap("sub.close")
    r_raw, _ = sharded.exchange_bytes(self.dist, raw, raw_to.astype(np.int64) * dy.RAW_MATCH_BYTES, dev)
    g_loads = sharded.all_reduce_sum(self.dist, loads, dev); lap("exchange C2 + allreduce")
    res = dy.finalize_raw(ctx, self.index, reads, params, r_raw.data_ptr(), r_raw.numel() // dy.RAW_MATCH_BYTES, g_loads.astype(np.uint32)); lap("finalize")
    self.T = T
    return res
sm.map = types.MethodType(timed_map, sm)
for it in range(4):
    r = sm.map(reads, params, thr); r.close()
print({k: round(v, 3) for k, v in sm.T.items()}, "total", round(sum(sm.T.values()), 3))
grp.close()

"""N>1 path on CPU: two gloo ranks run the bench's distributed protocol (disjoint read shards, barrier, max-over-ranks
timing, whole-job rate).  The mapping itself needs no collective (reads are the sharded unit)."""
import os
import subprocess
import sys
import textwrap

from conftest import ROOT


def test_two_rank_protocol_over_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent("""
        import os, sys, time, json
        sys.path.insert(0, %r)
        import numpy as np
        from dream_yara_amd.dist import Group, shard_seed, whole_job_rate
        from dream_yara_amd import synth
        g = Group("gloo")
        assert g.world == 2
        genomes = [synth.make_genome(b, 5000) for b in range(4)]          # same on every rank (seeded by bin)
        codes, offs, truth = synth.simulate_reads(genomes, 200, 100, seed=shard_seed(2, g.rank))
        g.barrier()
        elapsed = 0.5 + 0.25 * g.rank                                      # rank 1 is the slow one
        mx = g.max_over_ranks(elapsed)
        total = g.sum_over_ranks(float(codes.astype(np.int64).sum()))
        out = {"rank": g.rank, "max": mx, "sum": total, "mine": float(codes.astype(np.int64).sum()),
               "genome0": int(genomes[0][:32].astype(np.int64).sum()), "rate": whole_job_rate(200, 1, g.world, mx)}
        open(os.path.join(%r, "r%%d.json" %% g.rank), "w").write(json.dumps(out))
        g.barrier(); g.close()
    """ % (ROOT, str(tmp_path))))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", "29611", str(script)], env=env, timeout=300)
    import json
    r0 = json.load(open(tmp_path / "r0.json")); r1 = json.load(open(tmp_path / "r1.json"))
    assert r0["max"] == r1["max"] == 0.75
    assert r0["genome0"] == r1["genome0"]                  # replicated reference
    assert r0["mine"] != r1["mine"]                        # disjoint read shards
    assert abs(r0["sum"] - (r0["mine"] + r1["mine"])) < 1e-6
    assert abs(r0["rate"] - 400 / 0.75) < 1e-9             # whole-job aggregate over the slowest rank


def test_sharded_exchange_over_gloo(tmp_path):
    """The all-to-all(v) plumbing of the bin-sharded mode (dream_yara_amd/sharded.py) with ragged and empty splits."""
    script = tmp_path / "worker2.py"
    script.write_text(textwrap.dedent("""
        import os, sys, json
        sys.path.insert(0, %r)
        import numpy as np, torch
        import torch.distributed as dist
        from dream_yara_amd.dist import Group
        from dream_yara_amd import sharded
        g = Group("gloo")
        # rank r sends (r+1)*(d+2) bytes of value 10*r+d to rank d, except rank 1 sends nothing to rank 0
        splits = [(g.rank + 1) * (d + 2) for d in range(g.world)]
        if g.rank == 1: splits[0] = 0
        send = torch.cat([torch.full((n,), 10 * g.rank + d, dtype=torch.uint8) for d, n in enumerate(splits)])
        recv, rs = sharded.exchange_bytes(dist, send, splits, "cpu")
        loads = sharded.all_reduce_sum(dist, [g.rank + 1, 5, 0], "cpu")
        masks = sharded.owner_masks(130, g.world)
        out = {"recv": recv.tolist(), "splits": rs, "loads": loads.tolist(),
               "masks_or": [int(x) for x in np.bitwise_or.reduce(masks, axis=0)], "masks_and": [int(x) for x in np.bitwise_and.reduce(masks, axis=0)]}
        open(os.path.join(%r, "x%%d.json" %% g.rank), "w").write(json.dumps(out))
        g.barrier(); g.close()
    """ % (ROOT, str(tmp_path))))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", "29613", str(script)], env=env, timeout=300)
    import json
    x0 = json.load(open(tmp_path / "x0.json")); x1 = json.load(open(tmp_path / "x1.json"))
    assert x0["splits"] == [2, 0] and x0["recv"] == [0, 0]
    assert x1["splits"] == [3, 6] and x1["recv"] == [1] * 3 + [11] * 6
    assert x0["loads"] == x1["loads"] == [3, 10, 0]
    assert x0["masks_or"] == [2**64 - 1, 2**64 - 1, 3] and x0["masks_and"] == [0, 0, 0]      # every bin has exactly one owner

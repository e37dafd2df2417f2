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
    """The plumbing of the bin-sharded mode (dream_yara_amd/sharded.py): rows of counts, then ONE packed all-to-all(v) per exchange, with
    ragged and empty segments; and the static bin -> rank assignment."""
    script = tmp_path / "worker2.py"
    script.write_text(textwrap.dedent("""
        import os, sys, json
        sys.path.insert(0, %r)
        import numpy as np, torch
        import torch.distributed as dist
        from dream_yara_amd.dist import Group
        from dream_yara_amd import sharded
        g = Group("gloo")
        W = g.world
        # rank r has (r+1)*(d+2) bytes of value 10*r+d for rank d, except: rank 1 has nothing for rank 0, and nobody has anything for itself in round 2
        def run(round_no):
            splits = [(g.rank + 1) * (d + 2) for d in range(W)]
            if g.rank == 1: splits[0] = 0
            if round_no == 2: splits[g.rank] = 0
            rows = torch.zeros(W * 32, dtype=torch.uint8)
            rows.view(torch.int64).view(W, 4)[:, 2] = torch.tensor(splits)          # dy_seg_header.bytes
            rows.view(torch.int64).view(W, 4)[:, 0] = g.rank                        # n_reads: who sent the row
            mine, theirs = sharded.exchange_rows(dist, rows, 32, W)
            h_out = np.ascontiguousarray(mine).view("<u8").reshape(W, 4); h_in = np.ascontiguousarray(theirs).view("<u8").reshape(W, 4)
            send = torch.cat([torch.full((n,), 10 * g.rank + d, dtype=torch.uint8) for d, n in enumerate(splits)] + [torch.zeros(7, dtype=torch.uint8)])   # slack behind the segments
            recv = sharded.exchange_segments(dist, send, h_out[:, 2], h_in[:, 2], "cpu")
            return {"recv": recv.tolist(), "from": h_in[:, 0].tolist(), "bytes_in": h_in[:, 2].tolist(), "bytes_out": h_out[:, 2].tolist()}
        out = {"r1": run(1), "r2": run(2)}
        # all segments empty
        rows = torch.zeros(W * 32, dtype=torch.uint8)
        mine, theirs = sharded.exchange_rows(dist, rows, 32, W)
        recv = sharded.exchange_segments(dist, torch.zeros(0, dtype=torch.uint8), [0] * W, [0] * W, "cpu")
        out["empty"] = [int(recv.numel()), int(theirs.sum())]
        out["loads"] = sharded.all_reduce_sum(dist, [g.rank + 1, 5, 0], "cpu").tolist()
        masks = sharded.owner_masks(130, W)
        out["masks_or"] = [int(x) for x in np.bitwise_or.reduce(masks, axis=0)]; out["masks_and"] = [int(x) for x in np.bitwise_and.reduce(masks, axis=0)]
        open(os.path.join(%r, "x%%d.json" %% g.rank), "w").write(json.dumps(out))
        g.barrier(); g.close()
    """ % (ROOT, str(tmp_path))))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", "29613", str(script)], env=env, timeout=300)
    import json
    x0 = json.load(open(tmp_path / "x0.json")); x1 = json.load(open(tmp_path / "x1.json"))
    assert x0["r1"]["bytes_out"] == [2, 3] and x1["r1"]["bytes_out"] == [0, 6]
    assert x0["r1"]["bytes_in"] == [2, 0] and x0["r1"]["recv"] == [0, 0] and x0["r1"]["from"] == [0, 1]
    assert x1["r1"]["bytes_in"] == [3, 6] and x1["r1"]["recv"] == [1] * 3 + [11] * 6
    assert x0["r2"]["bytes_in"] == [0, 0] and x0["r2"]["recv"] == []                 # rank 1 sends rank 0 nothing, rank 0 nothing to itself
    assert x1["r2"]["bytes_in"] == [3, 0] and x1["r2"]["recv"] == [1] * 3
    assert x0["empty"] == [0, 0] and x1["empty"] == [0, 0]
    assert x0["loads"] == x1["loads"] == [3, 10, 0]
    assert x0["masks_or"] == [2**64 - 1, 2**64 - 1, 3] and x0["masks_and"] == [0, 0, 0]      # every bin has exactly one owner


def test_bin_assignment_lpt():
    """Static bin -> rank assignment: longest-processing-time-first on the bins' sizes; equal bins = bin mod G."""
    import numpy as np
    from dream_yara_amd import sharded
    assert sharded.bin_owner(10, 4, [7] * 10).tolist() == [b % 4 for b in range(10)]
    assert sharded.bin_owner(10, 4).tolist() == [b % 4 for b in range(10)]
    sizes = [100, 1, 1, 1, 50, 50, 1, 1]
    own = sharded.bin_owner(8, 2, sizes)
    load = [sum(s for s, o in zip(sizes, own) if o == r) for r in range(2)]
    assert own[0] != own[4] and own[4] == own[5] and abs(load[0] - load[1]) <= 1     # the big bin alone against the two halves
    m = sharded.owner_masks(8, 2, own)
    assert int(m[0, 0] | m[1, 0]) == 255 and int(m[0, 0] & m[1, 0]) == 0


def test_exchange_failure_is_collective(product_build, tmp_path):
    """csrc/multi_gpu.hpp Exchange (the C++ host's all-to-all between rank threads): rows travel with empty and ragged sizes, a rank that
    cannot prepare its side makes EVERY rank throw before any transfer is queued, and an aborted barrier wakes its waiters."""
    exe = tmp_path / "exchange_fail"
    src = os.path.join(ROOT, "tests", "native", "exchange_fail.cpp")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", str(exe), src,
                           "-L/opt/rocm/lib", "-lamdhip64", "-ldl", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rows ok" in out.stdout and "collective failure ok" in out.stdout and "abort ok" in out.stdout

#!/usr/bin/env python3
"""bench.py — DREAM-Yara mapping hot path on MI355X: mapped reads/s, 100 bp SE, 64-bin IBF (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = one pass of the whole device path over one resident batch: K1 IBF sweep -> read->bin work list -> K2 seed search
-> K3 classify/rank/locate/extend -> per-read aggregation (dedupe, strata, primary, CIGAR of primaries).  Inputs (reads,
filter, indices) are in HBM before the timed region; results stay in HBM (the PCIe-inclusive rate is in DESIGN.md).
Multi-GPU (default, --shard reads): reads are sharded across ranks, every rank holds the filter and all bins (1 GiB +
~11 GB of 288 GB), so there is no data-path collective; scaling is weak (1 M reads per GPU per step).
--shard bins: the reference's own unit of distribution — every rank holds the FM structures of
`bin mod N == rank` only; reads travel to the owners of their selected bins and raw matches travel back to the read's
home rank (two all-to-alls over RCCL, dream_yara_amd/sharded.py); same results, same weak-scaling accounting.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import shutil
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: bins, genome length per bin, IBF GiB, k, hashes, reads per GPU, read length, error %
    "config2": dict(bins=64, glen=4_000_000, ibf_gib=1, k=18, h=3, reads=1_000_000, L=100, e=3,
                    desc="64 bins x 4 Mbp, 1 GiB IBF (k=18, 3 hashes), 1M synthetic 100 bp SE reads, e=3"),
    # BASELINE.json configs[2] geometry on ONE GPU (256 bins, 16 GiB filter, W = 4): not the default; `--workload config3`
    "config3": dict(bins=256, glen=4_000_000, ibf_gib=16, k=18, h=3, reads=1_000_000, L=100, e=3,
                    desc="256 bins x 4 Mbp, 16 GiB IBF (k=18, 3 hashes), 1M synthetic 100 bp SE reads per step, e=3"),
    # BASELINE.json configs[4] geometry on ONE GPU (1024 bins, 64 GiB filter resident, W = 16, 150 bp, e = 5): `--workload config5`
    "config5": dict(bins=1024, glen=4_000_000, ibf_gib=64, k=18, h=3, reads=1_000_000, L=150, e=5,
                    desc="1024 bins x 4 Mbp, 64 GiB IBF (k=18, 3 hashes), 1M synthetic 150 bp SE reads per step, e=5"),
    # small variant for smoke runs / CI (NOT the reported configuration)
    "tiny": dict(bins=64, glen=100_000, ibf_gib=0, k=18, h=3, reads=20_000, L=100, e=3,
                 desc="64 bins x 100 kbp, 32 MiB IBF, 20k reads (smoke only)"),
}


def log(msg):
    print("[bench] " + msg, file=sys.stderr, flush=True)


def build_inputs(wl, work_dir, rank, world, barrier):
    """Setup (untimed): genomes, index files (rank 0 builds, all load), reads. Returns (genomes, index_dir)."""
    from dream_yara_amd import synth
    t0 = time.time()
    genomes = [synth.make_genome(b, wl["glen"]) for b in range(wl["bins"])]
    idx_dir = os.path.join(work_dir, "index")
    done_flag = os.path.join(idx_dir, "DONE")
    if rank == 0 and not os.path.exists(done_flag):
        if os.path.exists(idx_dir):
            shutil.rmtree(idx_dir)
        os.makedirs(idx_dir)
        synth.build_index_dir(genomes, idx_dir, threads=min(32, max(1, (os.cpu_count() or 1))))
        shutil.rmtree(os.path.join(idx_dir, "fasta"), ignore_errors=True)
        open(done_flag, "w").write("ok\n")
    barrier()
    log("rank %d: genomes + index ready in %.1f s" % (rank, time.time() - t0))
    return genomes, idx_dir + "/"


def cpu_baseline(wl, idx_dir, filter_path, codes, offs, error_rate, sample_reads):
    """Oracle (CPU restatement of the reference path, kind 'port') on a bounded sample of the same workload."""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    n = min(sample_reads, offs.size - 1)
    sub_codes = codes[: int(offs[n])]
    sub_offs = offs[: n + 1]
    oibf = ob.OracleIbf(filter_path)
    oidx = ob.OracleIndex(idx_dir, wl["bins"])
    t0 = time.time()
    reads, _, _, _ = ob.map_batch(oidx, oibf, sub_codes, sub_offs, error_rate, threads=threads)
    dt = time.time() - t0
    # the same restatement on ONE thread, on a smaller slice of the same batch (SURVEY.md §8d asks for both figures)
    n1 = min(n, 200000)
    t1 = time.time()
    ob.map_batch(oidx, oibf, codes[: int(offs[n1])], offs[: n1 + 1], error_rate, threads=1)
    dt1 = time.time() - t1
    oidx.close(); oibf.close()
    return {"value": n / dt, "unit": "reads/s", "cores": threads, "kind": "port",
            "sample": "%d reads of the same batch, oracle (oracle/dyo_*.hpp) with %d threads, %.1f s; filter/index load excluded" % (n, threads, dt),
            "mapped_fraction": float(reads["mapped"].mean()),
            "single_thread": {"value": n1 / dt1, "unit": "reads/s", "cores": 1, "sample": "%d reads, %.1f s" % (n1, dt1)}}, reads


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-sample", type=int, default=1000000, help="reads in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--paired", action="store_true", help="paired-end batch: reads/2 pairs of 2 x L bp, insert N(400, 40) (BASELINE configs[3]); not the default metric")
    ap.add_argument("--shard", default="reads", choices=["reads", "bins"], help="what is partitioned across ranks (see module docstring)")
    ap.add_argument("--work-dir", default=os.environ.get("DY_BENCH_DIR", "/tmp/dy_bench_%d" % os.getuid()))
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]

    from dream_yara_amd.dist import Group, shard_seed, whole_job_rate
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus))
        raise SystemExit("--gpus (%d) does not match WORLD_SIZE (%d)" % (args.gpus, world))

    import torch
    grp = Group("nccl" if (world > 1 or args.shard == "bins") else None, force=args.shard == "bins")
    rank, local_rank = grp.rank, grp.local_rank
    barrier = grp.barrier

    import dream_yara_amd as dy
    from dream_yara_amd import synth, build as dybuild
    if not os.path.exists(dy.LIB_PATH):
        dybuild.build()
    if dy.device_count() == 0:
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")

    work_dir = os.path.join(args.work_dir, args.workload)
    os.makedirs(work_dir, exist_ok=True)
    genomes, idx_dir = build_inputs(wl, work_dir, rank, world, barrier)

    ctx = dy.Context(local_rank)
    t0 = time.time()
    n_bits = (wl["ibf_gib"] << 33) if wl["ibf_gib"] else (1 << 28)
    ibf = dy.Ibf.create(ctx, wl["bins"], wl["h"], wl["k"], n_bits + 256)
    for b, g in enumerate(genomes):
        ibf.add(g, b)
    sm = None
    if args.shard == "bins":
        from dream_yara_amd import sharded
        sm = sharded.ShardedMapper(ctx, grp.dist, rank, world, idx_dir, wl["bins"], ibf, "cuda")
        index = sm.index
    else:
        index = dy.Index(ctx, idx_dir, wl["bins"])
    log("rank %d: filter built on device + index uploaded in %.1f s" % (rank, time.time() - t0))

    t0 = time.time()
    if args.paired:
        codes, offs, ptruth = synth.simulate_pairs(genomes, wl["reads"] // 2, wl["L"], seed=shard_seed(4, rank))
        truth = np.zeros(wl["reads"], dtype=[("bin", "<u4")]); truth["bin"] = np.concatenate([ptruth["bin"], ptruth["bin"]])
    else:
        codes, offs, truth = synth.simulate_reads(genomes, wl["reads"], wl["L"], seed=shard_seed(2, rank))
    reads = dy.Reads(ctx, codes, offs)
    del genomes
    log("rank %d: %d reads simulated + uploaded in %.1f s" % (rank, wl["reads"], time.time() - t0))

    error_rate = np.float32(wl["e"] / 100.0)
    params = dy.default_params(error_rate=error_rate, paired=1 if args.paired else 0)
    thr = dy.threshold(error_rate, wl["k"], wl["L"])

    def step():
        if sm is not None:
            res = sm.map(reads, params, thr)
        else:
            ibf.query(reads, thr, want_masks=False)
            res = dy.map_reads(ctx, index, reads, params, asynchronous=True)
        ctx.sync()
        return res

    last = None
    for _ in range(args.warmup):
        r = step()
        r.close()
    stage_ms = {k: 0.0 for k in ("ms_ibf", "ms_items", "ms_search", "ms_extend", "ms_finalize")}
    barrier(); torch.cuda.synchronize() if torch.cuda.is_available() else None; ctx.sync()
    t_start = time.perf_counter()
    for i in range(args.steps):
        r = step()
        st = ctx.stats()
        for k in stage_ms:
            stage_ms[k] += st[k]
        if i + 1 < args.steps:
            r.close()
        else:
            last = r
    ctx.sync(); torch.cuda.synchronize() if torch.cuda.is_available() else None; barrier()
    elapsed = grp.max_over_ranks(time.perf_counter() - t_start)

    # correctness spot-check of the last step against the simulation truth (not timed)
    rr = last.fetch().reads()
    stats = ctx.stats()
    mapped_fraction = float(rr["mapped"].mean())
    ok_bin = float((rr["primary"]["contig"][rr["mapped"] == 1] == truth["bin"][rr["mapped"] == 1]).mean())
    last.close()

    if rank == 0:
        steps = max(1, args.steps)
        ms_per_step = 1e3 * elapsed / steps
        value = whole_job_rate(wl["reads"], steps, world, elapsed)
        # roofline of the dominant HBM kernel K1: algorithmic bytes = strands * (L-k+1) * h * W * 8 B per read
        W = (wl["bins"] + 63) // 64
        bytes_per_read = 2 * (wl["L"] - wl["k"] + 1) * wl["h"] * W * 8
        k1_s = stage_ms["ms_ibf"] / steps / 1e3
        achieved = bytes_per_read * wl["reads"] / k1_s / 1e9 if k1_s > 0 else 0.0
        # HBM traffic of K1 per launch from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes;
        # narrow gathers are accounted at 64 B per request, calibrated on scripts/ubench_gather.hip: profiles/r01_fetch_size_calibration.json)
        traffic = None
        pmc_file = os.path.join(ROOT, "profiles", "r01n_pmc_config2.json")
        if args.workload == "config2" and os.path.exists(pmc_file):
            allk = json.load(open(pmc_file))
            pmc = allk.get("dy::k_ibf_query_two") or allk.get("void dy::k_ibf_query<7>") or {}
            if "FETCH_SIZE" in pmc:
                traffic = (pmc["FETCH_SIZE"] + pmc.get("WRITE_SIZE", 0.0)) * 1024.0
        out = {
            "metric": "mapped reads/sec (whole node), 100 bp SE, 64-bin IBF",
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": wl["desc"] + (" — PAIRED: %d pairs per step, insert N(400,40)" % (wl["reads"] // 2) if args.paired else ""), "reads_per_gpu_per_step": wl["reads"], "parallelism": ("reads sharded x%d, filter+index replicated" % world) if sm is None else
                                      ("bins sharded x%d (bin mod N), filter+contigs replicated, reads->owners and raw matches->home by all-to-all" % world),
                       "mapped_fraction": mapped_fraction, "primary_in_true_bin": ok_bin,
                       "items_per_read": (stats["items"] if sm is None else sm.last["items"] / world) / wl["reads"], "raw_matches_per_read": stats["raw_matches"] / wl["reads"],
                       "stage_ms_per_step": {k: v / steps for k, v in stage_ms.items()}},
            "roofline": {"kernel": "k_ibf_query_two" if W == 1 else "k_ibf_query_wide", "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic,
                         "note": ("algorithmic bytes %d B/read x %d reads per launch; " % (bytes_per_read, wl["reads"])) +
                                 ("every 8 B word costs one 64 B HBM request at W=1; the kernel runs at the measured random-gather ceiling of the chip (55.6 G gathers/s = 3.56 TB/s of 64 B lines, scripts/ubench_gather.hip) and prunes gathers that cannot change the result; its vector ALUs are 78% busy at the same time (profiles/r01m_pmc_sq_k1.json)"
                                  if W == 1 else "W = %d words per hash position are fetched by adjacent lanes (k_ibf_query_wide); bound by vector-ALU issue, see DESIGN.md" % W)},
        }
        if args.paired:
            out["config"]["paired_fraction"] = float(rr["paired"].mean())
        if args.cpu_sample > 0 and world == 1 and not args.paired:
            fpath = os.path.join(work_dir, "bench.filter")
            ibf.save(fpath)
            base, oreads = cpu_baseline(wl, idx_dir, fpath, codes, offs, error_rate, args.cpu_sample)
            # the sample doubles as a parity check of the timed configuration
            n = oreads.size
            same = (np.array_equal(oreads["mapped"], rr["mapped"][:n]) and
                    np.array_equal(oreads["primary"]["begin"][oreads["mapped"] == 1], rr["primary"]["begin"][:n][oreads["mapped"] == 1]) and
                    np.array_equal(oreads["best_count"], rr["best_count"][:n]) and np.array_equal(oreads["mapq"], rr["mapq"][:n]))
            base["parity_with_gpu_on_sample"] = bool(same)
            out["cpu_baseline"] = base
            try:
                os.remove(fpath)
            except OSError:
                pass
        print(json.dumps(out), flush=True)
    barrier()
    reads.close(); index.close(); ibf.close(); ctx.close()
    grp.close()


if __name__ == "__main__":
    main()

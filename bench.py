#!/usr/bin/env python3
"""bench.py — DREAM-Yara mapping hot path on MI355X: mapped reads/s, 100 bp SE, 64-bin IBF (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = one pass of the whole device path over the resident batch(es): K1 IBF sweep -> read->bin work list -> K2 seed search
-> K3 classify/rank/locate/extend -> per-read aggregation (dedupe, strata, primary, CIGAR of primaries).  Inputs (reads,
filter, indices) are in HBM before the timed region; results stay in HBM (the PCIe-inclusive rate is in DESIGN.md).

Multi-GPU (N > 1), --shard both (default): BOTH partitionings are timed back to back, K steps each, and reported in the one line:
  reads   reads are the sharded unit — every rank holds the filter and all bins and maps its own batch, no data-path collective
          (the faster layout while the index fits one GPU, which 288 GB makes true for every BASELINE configuration);
  bins    the reference's unit of distribution and BASELINE.json's north-star partitioning: every rank holds the FM
          structures of `bin mod N == rank` only; reads travel to the owners of their selected bins and raw matches (paired: plus
          pairing records) travel back to the read's home rank — all-to-alls over RCCL (dream_yara_amd/sharded.py).
`value` is the whole-job rate of the reads-sharded leg; `bins_sharded` carries the same fields for the other leg.  Both legs are weak
scaling (the same reads per GPU per step).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import shutil
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: bins, genome length per bin, IBF GiB, k, hashes, reads per GPU per step (batch = reads per resident batch), read length, error %
    "config2": dict(bins=64, glen=4_000_000, ibf_gib=1, k=18, h=3, reads=1_000_000, L=100, e=3,
                    desc="64 bins x 4 Mbp, 1 GiB IBF (k=18, 3 hashes), 1M synthetic 100 bp SE reads, e=3"),
    # BASELINE.json's north_star target sentence: 10 M reads against a 64-bin 16 GiB IBF + FM indices, as 5 resident batches of 2 M
    # (a batch holds at most 2 097 151 reads: 21-bit read id, reference src/bits_matches.h:179-183)
    "northstar": dict(bins=64, glen=4_000_000, ibf_gib=16, k=18, h=3, reads=10_000_000, batch=2_000_000, L=100, e=3,
                      desc="64 bins x 4 Mbp, 16 GiB IBF (k=18, 3 hashes), 10M synthetic 100 bp SE reads per step as 5 resident batches of 2M, e=3"),
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


# stdout carries the ONE JSON line and nothing else: RCCL prints a version banner to stdout when its first communicator comes up, so
# file descriptor 1 is pointed at stderr for the run and the line is written to the saved descriptor
_REAL_STDOUT = None


def capture_stdout():
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(line):
    data = (line + "\n").encode()
    os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, data)


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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def compare_all_fields(dev, ora):
    """Every field of every read: per-read records, primary match bytes, matchesSet bytes, CIGAR ops.  Returns the differing fields."""
    d_reads, d_matches, d_cigars = dev
    o_reads, o_matches, o_cigars = ora
    bad = []
    if d_reads.size != o_reads.size:
        return ["read count"]
    for f in ("mapped", "best_count", "sub_count", "mapq", "cigar_len", "set_begin", "cigar_begin"):
        if not np.array_equal(d_reads[f], o_reads[f]):
            bad.append(f)
    m = o_reads["mapped"] == 1
    if d_reads["primary"][m].tobytes() != o_reads["primary"][m].tobytes():
        bad.append("primary")
    if d_matches.tobytes() != o_matches.tobytes():
        bad.append("matchesSet")
    if not np.array_equal(d_cigars, o_cigars):
        bad.append("cigars")
    return bad


def first_pairs(codes, offs, k):
    """The first k pairs of a paired batch ([first mates | second mates]) as a paired batch of 2k reads."""
    P = (offs.size - 1) // 2
    lens = np.diff(offs).astype(np.uint64)
    sel = np.concatenate([np.arange(0, k), np.arange(P, P + k)])
    parts = [codes[int(offs[i]):int(offs[i + 1])] for i in sel]
    o = np.zeros(2 * k + 1, np.uint64); o[1:] = np.cumsum(lens[sel])
    return np.concatenate(parts), o


def cpu_baseline(wl, idx_dir, oibf, codes, offs, error_rate, sample_reads, paired=False):
    """Oracle (CPU restatement of the reference path, kind 'port') on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    n = min(sample_reads, offs.size - 1)
    if paired:
        n = offs.size - 1               # a paired batch is [first mates | second mates]: the whole batch, so that it doubles as the parity check
    oidx = ob.OracleIndex(idx_dir, wl["bins"])
    t0 = time.time()
    res = ob.map_batch(oidx, oibf, codes[: int(offs[n])], offs[: n + 1], error_rate, threads=threads, paired=paired)
    dt = time.time() - t0
    # the same restatement on ONE thread, on a smaller slice of the same batch (SURVEY.md §8d asks for both figures)
    n1 = min(n, 200000)
    c1, o1 = (codes[: int(offs[n1])], offs[: n1 + 1]) if not paired else first_pairs(codes, offs, n1 // 2)
    t1 = time.time()
    ob.map_batch(oidx, oibf, c1, o1, error_rate, threads=1, paired=paired)
    dt1 = time.time() - t1
    oidx.close()
    return {"value": n / dt, "unit": "reads/s", "cores": threads, "host_cores": cores, "host_cpu": cpu_model(), "kind": "port",
            "sample": "%d reads of the same batch, oracle (oracle/dyo_*.hpp: a restatement of the reference, not the SeqAn binary) with %d threads on a host with %d logical cores, %.1f s; filter/index load excluded" % (n, threads, cores, dt),
            "mapped_fraction": float(res[0]["mapped"].mean()),
            "single_thread": {"value": n1 / dt1, "unit": "reads/s", "cores": 1, "sample": "%d reads, %.1f s" % (n1, dt1)}}, res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-sample", type=int, default=1000000, help="reads in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--paired", action="store_true", help="paired-end batch: reads/2 pairs of 2 x L bp, insert N(400, 40) (BASELINE configs[3]); not the default metric")
    ap.add_argument("--shard", default="both", choices=["both", "reads", "bins"], help="what is partitioned across ranks (see module docstring); at N = 1 'both' means reads")
    ap.add_argument("--bins-leg-timeout", type=float, default=300.0, help="seconds after which the bins-sharded leg of --shard both is given up (the line is printed without it)")
    ap.add_argument("--prefetch", default="auto", choices=["auto", "on", "off"],
                    help="bins-sharded leg: queue the next batch's IBF sweep in chunks in front of the host's waits (count fetches, peers).  auto = on with more than one rank "
                         "(on one GPU the waits are short: 5.96 ms per step without, 6.00 ms with, profiles/r03_*)")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE", help="dy_ctx_set_tuning knob for experiments (result-identical code paths; see include/dyhip.h)")
    ap.add_argument("--work-dir", default=os.environ.get("DY_BENCH_DIR", "/tmp/dy_bench_%d" % os.getuid()))
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]
    capture_stdout()

    from dream_yara_amd.dist import Group, shard_seed, whole_job_rate
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus))
        raise SystemExit("--gpus (%d) does not match WORLD_SIZE (%d)" % (args.gpus, world))
    legs = ["reads", "bins"] if (args.shard == "both" and world > 1) else (["reads"] if args.shard in ("both", "reads") else ["bins"])

    import torch
    need_dist = world > 1 or "bins" in legs
    grp = Group("nccl" if need_dist else None, force="bins" in legs)
    rank, local_rank = grp.rank, grp.local_rank
    barrier = grp.barrier

    import dream_yara_amd as dy
    from dream_yara_amd import synth, build as dybuild
    if not os.path.exists(dy.LIB_PATH):
        dybuild.build()
    if dy.device_count() == 0:
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")

    work_dir = os.path.join(args.work_dir, "config2" if args.workload == "northstar" else args.workload)      # northstar shares config2's genomes + index
    os.makedirs(work_dir, exist_ok=True)
    genomes, idx_dir = build_inputs(wl, work_dir, rank, world, barrier)

    ctx = dy.Context(local_rank)
    for kv in args.tune:
        key, val = kv.split("=", 1)
        ctx.set_tuning(key, int(val))
    t0 = time.time()
    n_bits = (wl["ibf_gib"] << 33) if wl["ibf_gib"] else (1 << 28)
    ibf = dy.Ibf.create(ctx, wl["bins"], wl["h"], wl["k"], n_bits + 256)
    for b, g in enumerate(genomes):
        ibf.add(g, b)
    log("rank %d: filter built on device in %.1f s" % (rank, time.time() - t0))

    t0 = time.time()
    per_batch = wl.get("batch", wl["reads"])
    n_batches = wl["reads"] // per_batch
    batches, truths, host_batch0 = [], [], None
    for bi in range(n_batches):
        seed = shard_seed(4 if args.paired else 2, rank) + 7919 * bi
        if args.paired:
            codes, offs, ptruth = synth.simulate_pairs(genomes, per_batch // 2, wl["L"], seed=seed)
            truth_bin = np.concatenate([ptruth["bin"], ptruth["bin"]])
        else:
            codes, offs, t = synth.simulate_reads(genomes, per_batch, wl["L"], seed=seed)
            truth_bin = t["bin"]
        batches.append(dy.Reads(ctx, codes, offs))
        truths.append(truth_bin)
        if bi == 0:
            host_batch0 = (codes, offs)
    del genomes
    log("rank %d: %d reads simulated + uploaded in %.1f s" % (rank, wl["reads"], time.time() - t0))

    error_rate = np.float32(wl["e"] / 100.0)
    params = dy.default_params(error_rate=error_rate, paired=1 if args.paired else 0)
    thr = dy.threshold(error_rate, wl["k"], wl["L"])
    STAGES = ("ms_ibf", "ms_items", "ms_search", "ms_extend", "ms_finalize")

    def run_leg(mode):
        """W warm-up steps, then K timed steps between barriers; returns the leg's numbers and the last step's first-batch result."""
        sm = None
        t_load = time.time()
        if mode == "bins":
            from dream_yara_amd import sharded
            sm = sharded.ShardedMapper(ctx, grp.dist, rank, world, idx_dir, wl["bins"], ibf, "cuda", prefetch=args.prefetch == "on" or (args.prefetch == "auto" and world > 1))
            index = sm.index
        else:
            index = dy.Index(ctx, idx_dir, wl["bins"])
        log("rank %d [%s]: index uploaded in %.1f s" % (rank, mode, time.time() - t_load))
        info = index.info()

        def step(keep_first, more_steps=False):
            kept, st_sum, items, raws, st = None, {k: 0.0 for k in STAGES}, 0, 0, None
            for bi, reads in enumerate(batches):
                if sm is not None:
                    # the sweep of the batch mapped next is queued behind this batch's C1 packing (same work inside the timed region:
                    # the first step sweeps its own batch, the last one queues nothing)
                    nxt = batches[bi + 1] if bi + 1 < len(batches) else (batches[0] if more_steps else None)
                    res = sm.map(reads, params, thr, next_reads=nxt)
                else:
                    # (queuing the NEXT batch's sweep here, on the context's second stream, was measured: +1.5 % — K1 fills every CU and
                    # K2..K4 only run in its tail)
                    ibf.query(reads, thr, want_masks=False)
                    res = dy.map_reads(ctx, index, reads, params, asynchronous=True)
                ctx.sync()
                st = ctx.stats()
                for k in STAGES:
                    st_sum[k] += st[k]
                items += (st["items"] if sm is None else sm.last["items"] / world)
                raws += st["raw_matches"]
                if keep_first and bi == 0:
                    kept = res
                else:
                    res.close()
            return kept, st_sum, items, raws, st

        for _ in range(args.warmup):
            step(False)
        ctx.sync()
        stage_ms = {k: 0.0 for k in STAGES}
        barrier(); torch.cuda.synchronize() if torch.cuda.is_available() else None; ctx.sync()
        t_start = time.perf_counter()
        last, items, raws, st = None, 0, 0, None
        for i in range(args.steps):
            kept, st_sum, items, raws, st = step(i + 1 == args.steps, more_steps=i + 1 < args.steps)
            for k in STAGES:
                stage_ms[k] += st_sum[k]
            if kept is not None:
                last = kept
        ctx.sync(); torch.cuda.synchronize() if torch.cuda.is_available() else None; barrier()
        elapsed = grp.max_over_ranks(time.perf_counter() - t_start)
        steps = max(1, args.steps)
        out = {"value": whole_job_rate(wl["reads"], steps, world, elapsed), "ms_per_step": 1e3 * elapsed / steps,
               "parallelism": ("reads sharded x%d, filter+index replicated, no data-path collective" % world) if sm is None else
                              ("bins sharded x%d (bin mod N), filter+contigs replicated, reads->owners and raw matches->home by all-to-all over RCCL" % world),
               "items_per_read": items / wl["reads"], "raw_matches_per_read": raws / wl["reads"],
               "stage_ms_per_step": {k: v / steps for k, v in stage_ms.items()},
               "ibf_kernel": dy.IBF_KERNEL_NAMES.get(st["ibf_kernel"]), "ibf_first_check_slack": st["ibf_slack"],
               "device_tables": {"resident_bins": info["n_bins_resident"], "bwt_rows": info["total_rows"],
                                 "reference_structures_bytes": {"rank_dictionary": info["bytes_rank"], "sparse_sa": info["bytes_sparse_sa"], "contigs_unpacked": info["bytes_contigs"]},
                                 "device_only_bytes": {"expanded_sa": info["bytes_expanded_sa"], "qmer_range_table_q%d" % info["prefix_q"]: info["bytes_prefix_table"],
                                                       "inverse_sa_and_indexed_text": info["bytes_isa_and_text"]},
                                 "filter_bytes": n_bits // 8}}
        rr = last.fetch()
        result = (rr.reads(), rr.matches(), rr.cigars())
        last.close()
        if sm is not None:
            sm.close()
        else:
            index.close()
        return out, result

    first = {}

    def make_line(leg, result):
        rr = result[0]
        truth = truths[0]
        mapped = rr["mapped"] == 1
        # roofline of the dominant HBM kernel K1: algorithmic bytes = strands * (L-k+1) * h * W * 8 B per read
        W = (wl["bins"] + 63) // 64
        bytes_per_read = 2 * (wl["L"] - wl["k"] + 1) * wl["h"] * W * 8
        k1_s = leg["stage_ms_per_step"]["ms_ibf"] / 1e3 / n_batches       # average duration of ONE K1 launch (HIP events on the context's stream)
        achieved = bytes_per_read * per_batch / k1_s / 1e9 if k1_s > 0 else 0.0
        # HBM traffic of that kernel per launch comes from a committed PMC pass of the same build and workload (rocprofv3 --pmc, separate
        # passes; narrow gathers are accounted at 64 B per request, calibrated on scripts/ubench_gather.hip) — it is NOT measured in this run
        traffic, traffic_src, secondary, line_slots = None, None, None, None
        pmc_file = os.path.join(ROOT, "profiles", "r02_pmc_%s.json" % args.workload)
        if os.path.exists(pmc_file):
            allk = json.load(open(pmc_file))
            kname = (leg["ibf_kernel"] or "").split("<")[0]
            pmc = next((v for kk, v in allk.items() if kname and kname in kk and "FETCH_SIZE" in v), None)
            if pmc:
                traffic = (pmc["FETCH_SIZE"] + pmc.get("WRITE_SIZE", 0.0)) * 1024.0
                traffic_src = "profiles/" + os.path.basename(pmc_file)
                if "TCC_EA0_RDREQ_sum" in pmc and k1_s > 0:
                    # every L2 miss, narrow or wide, occupies one 128-B line slot of the memory side (profiles/r02_gather_mix.jsonl: random 8-B
                    # gathers and a coalesced stream running together add up in 128-B lines; FETCH_SIZE tallies a request at 64 B)
                    req = pmc["TCC_EA0_RDREQ_sum"] * (per_batch / pmc["SQ_WAVES"] if pmc.get("SQ_WAVES") else 1.0)      # one wavefront per read
                    line_slots = {"l2_read_requests_per_launch": req, "bytes_at_128B_per_request": req * 128.0,
                                  "achieved_GBps": req * 128.0 / k1_s / 1e9, "frac_of_hbm_peak": req * 128.0 / k1_s / 1e9 / 8000.0,
                                  "source": traffic_src + " (requests per read) and this run's launch duration"}
            k2 = next((v for kk, v in allk.items() if "k_seed_search" in kk and "TCC_HIT_sum" in v), None)
            if k2:
                hit, miss = k2["TCC_HIT_sum"], k2["TCC_MISS_sum"]
                secondary = {"kernel": "k_seed_search (K2: FM backward search / LF-mapping)", "bound": "l2", "l2_hit_rate": hit / max(1.0, hit + miss),
                             "l2_requests_per_launch": hit + miss, "fetched_bytes_per_launch": k2.get("FETCH_SIZE", 0.0) * 1024.0,
                             "ms_per_launch": leg["stage_ms_per_step"]["ms_search"] / n_batches, "source": "profiles/" + os.path.basename(pmc_file),
                             "note": "K2 resolves a seed with one gather into the q-mer range table and one compare against the indexed text (both HBM-sized: every one an L2 miss, a 128-B line slot each) and walks the rank dictionary (L2-resident per XCD: items are bin-major, an XCD gets a contiguous slice) only for repeats; what hits the L2 are the read bases and the contig limits"}
        line = {
            "metric": "mapped reads/sec (whole node), 100 bp SE, 64-bin IBF",
            "value": leg["value"], "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": leg["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": wl["desc"] + (" — PAIRED: %d pairs per step, insert N(400,40)" % (wl["reads"] // 2) if args.paired else ""),
                       "reads_per_gpu_per_step": wl["reads"], "resident_batches": n_batches, "parallelism": leg["parallelism"],
                       "mapped_fraction": float(mapped.mean()), "primary_in_true_bin": float((rr["primary"]["contig"][mapped] == truth[mapped]).mean()),
                       "items_per_read": leg["items_per_read"], "raw_matches_per_read": leg["raw_matches_per_read"],
                       "stage_ms_per_step": leg["stage_ms_per_step"], "device_tables": leg["device_tables"]},
            "roofline": {"kernel": leg["ibf_kernel"], "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": traffic, "traffic_source": traffic_src, "line_slots": line_slots, "launch_ms": 1e3 * k1_s,
                         "note": ("algorithmic bytes %d B/read x %d reads per launch; " % (bytes_per_read, per_batch)) +
                                 "the chip serves 48-60 G random requests/s whatever their width from 8 to 128 B (profiles/r02_gather_ceiling.json: each L2 miss takes one 128-B line slot, 56 G/s x 128 B = 7.2 TB/s) and a hash position of W <= 16 words is one request, "
                                 "so the sweep is bound by the number of (k-mer, hash) positions it touches; it prunes those that cannot change a result (first check at nK - threshold + %d k-mers, from the sampled filter fill); "
                                 "at W = 1 only 8 of the 128 B of a line slot are algorithmic bytes" % leg["ibf_first_check_slack"],
                         "secondary": secondary},
        }
        if args.paired:
            line["config"]["paired_fraction"] = float(rr["paired"].mean())
        return line

    for mode in legs:
        if mode == legs[0]:
            leg, result = run_leg(mode)
            first = {"line": make_line(leg, result), "result": result}
            continue
        # second leg (bins-sharded at N > 1).  The first leg's line must not be lost to a hang here: RCCL all-to-alls at this world size
        # cannot be rehearsed on the one-GPU development boxes, so a watchdog prints what is there and ends the process.
        done = threading.Event()

        def watchdog():
            if not done.wait(args.bins_leg_timeout):
                if rank == 0:
                    line = dict(first["line"]); line["bins_sharded"] = {"error": "no result within %.0f s" % args.bins_leg_timeout}
                    emit(json.dumps(line))
                os._exit(0)
        threading.Thread(target=watchdog, daemon=True).start()
        try:
            leg2, _ = run_leg(mode)
        except Exception as e:          # noqa: BLE001 — the reads leg's numbers are still valid
            leg2 = {"error": "%s: %s" % (type(e).__name__, e)}
        done.set()
        first["line"]["bins_sharded"] = leg2

    if rank == 0:
        out = first["line"]
        if args.cpu_sample > 0 and world == 1 and legs[0] == "reads":
            import subprocess
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_binding as ob
            words, nb = ibf.words()                      # the oracle reads the very filter the device holds
            oibf = ob.OracleIbf(None, words=words, n_bits=nb)
            codes, offs = host_batch0
            base, ores = cpu_baseline(wl, idx_dir, oibf, codes, offs, error_rate, args.cpu_sample, paired=args.paired)
            oibf.close(); del words
            # the sample doubles as a parity check of the timed configuration: every field of every read of the sample
            n = ores[0].size
            d_reads, d_matches, d_cigars = first["result"]
            nm = int(d_reads["set_begin"][n - 1] + d_reads["best_count"][n - 1] + d_reads["sub_count"][n - 1]) if n else 0
            nc = int(d_reads["cigar_begin"][n - 1] + d_reads["cigar_len"][n - 1]) if n else 0
            diff = compare_all_fields((d_reads[:n], d_matches[:nm], d_cigars[:nc]), ores[:3])
            if args.paired:
                # the pairing weights: the oracle evaluates them with libm like the reference, the device with its fixed IEEE sequences
                if not np.array_equal(d_reads["paired"][:n], ores[0]["paired"]):
                    diff.append("paired")
                if not np.allclose(d_reads["pair_prob"][:n], ores[0]["pair_prob"], rtol=1e-9, atol=0.0):
                    diff.append("pair_prob")
            base["parity_with_gpu_on_sample"] = not diff
            base["parity_fields"] = ("mapped, X0, X1, MAPQ, primary (contig, begin, length, errors, strand), matchesSet, CIGARs" +
                                     (", paired, pair probability (relative 1e-9)" if args.paired else "") + " of the first %d reads" % n)
            if diff:
                base["parity_differs_in"] = diff
            out["cpu_baseline"] = base
        emit(json.dumps(out))
    barrier()
    for r in batches:
        r.close()
    ibf.close(); ctx.close()
    grp.close()


if __name__ == "__main__":
    main()

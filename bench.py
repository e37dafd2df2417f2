#!/usr/bin/env python3
"""bench.py — DREAM-Yara mapping hot path on MI355X: mapped reads/s, 100 bp SE, 64-bin IBF (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = one pass of the whole device path over the resident batch(es): K1 IBF sweep -> read->bin work list -> K2 seed search
-> K3 classify/rank/locate/extend -> per-read aggregation (dedupe, strata, primary, CIGAR of primaries).  Inputs (reads,
filter, indices) are in HBM before the timed region; results stay in HBM.

N = 1: `value` is configs[1] (64 bins, 1 GiB filter, 1 M reads) with every FM index resident.  The same line carries, each measured
in this run (shorter step counts, see `steps`):
  workloads        the other geometries BASELINE.json names on one GPU — `northstar` (64 bins / 16 GiB filter / 10 M reads per step),
                   `config3` (256 bins / 16 GiB), `config5` (1024 bins / 64 GiB, 150 bp, e=5): reads/s, K1 launch time, the algorithmic
                   and the line-slot roofline fractions, and a parity flag of every field of >= 200 k reads against the oracle;
  bins_sharded     the north-star PARTITIONING at N = 1 through RCCL (self all-to-all): what the exchanges cost with nothing to gain;
  pcie_inclusive   upload + map + fetch per step from host memory, one and three contexts on the device (never `value`);
  end_to_end       the command line tool, FASTQ in -> SAM out, process start and index load included (BASELINE.md section 3's region is
                   reported beside it), with the oracle's command line timed the same way on a sample.
N > 1: BOTH partitionings are timed, K steps each, weak scaling (the same reads per GPU per step):
  bins  (`value`)          BASELINE.json's north-star partitioning and the reference's own unit of distribution: every
                           rank holds the FM structures of the bins it owns (LPT on the bins' sizes = bin mod N for equal bins); reads
                           travel to the owners of their selected bins and raw matches (paired: plus pairing records) travel back to
                           the read's home rank — two packed all-to-alls over RCCL per batch (dream_yara_amd/sharded.py);
  reads (`reads_sharded`)  every rank holds the filter and all bins and maps its own batch, no data-path collective (the faster layout
                           while the index fits one GPU, which 288 GB makes true for every BASELINE configuration).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import shutil
import subprocess
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
    # BASELINE.json configs[2] geometry on ONE GPU (256 bins, 16 GiB filter, W = 4)
    "config3": dict(bins=256, glen=4_000_000, ibf_gib=16, k=18, h=3, reads=1_000_000, L=100, e=3,
                    desc="256 bins x 4 Mbp, 16 GiB IBF (k=18, 3 hashes), 1M synthetic 100 bp SE reads per step, e=3"),
    # BASELINE.json configs[4] geometry on ONE GPU (1024 bins, 64 GiB filter resident, W = 16, 150 bp, e = 5)
    "config5": dict(bins=1024, glen=4_000_000, ibf_gib=64, k=18, h=3, reads=1_000_000, L=150, e=5,
                    desc="1024 bins x 4 Mbp, 64 GiB IBF (k=18, 3 hashes), 1M synthetic 150 bp SE reads per step, e=5"),
    # small variant for smoke runs / CI (NOT the reported configuration)
    "tiny": dict(bins=64, glen=100_000, ibf_gib=0, k=18, h=3, reads=20_000, L=100, e=3,
                 desc="64 bins x 100 kbp, 32 MiB IBF, 20k reads (smoke only)"),
}
STAGES = ("ms_ibf", "ms_items", "ms_search", "ms_extend", "ms_finalize")
T_START = time.time()


def log(msg):
    print("[bench %5.0f s] %s" % (time.time() - T_START, msg), file=sys.stderr, flush=True)


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
    log("rank %d: genomes + index of %d bins ready in %.1f s" % (rank, wl["bins"], time.time() - t0))
    return genomes, idx_dir + "/"


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def effective_cpus():
    """CPUs this process may actually use: the cgroup's quota (a GPU box gives a 16-CPU share of its 256 logical cores) or the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


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


def oracle_binding():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    return ob


class Setup:
    """One workload resident on one device: filter built on the device, reads simulated and uploaded (untimed)."""

    def __init__(self, name, args, grp, paired=False):
        import dream_yara_amd as dy
        from dream_yara_amd import synth
        from dream_yara_amd.dist import shard_seed
        self.dy, self.name, self.args, self.grp, self.paired = dy, name, args, grp, paired
        self.wl = wl = dict(WORKLOADS[name])
        rank, world = grp.rank, grp.world
        work_dir = os.path.join(args.work_dir, "config2" if name == "northstar" else name)      # northstar shares config2's genomes + index
        os.makedirs(work_dir, exist_ok=True)
        self.work_dir = work_dir
        genomes, self.idx_dir = build_inputs(wl, work_dir, rank, world, grp.barrier)
        self.ctx = dy.Context(grp.local_rank)
        for kv in args.tune:
            key, val = kv.split("=", 1)
            self.ctx.set_tuning(key, int(val))
        t0 = time.time()
        self.n_bits = (wl["ibf_gib"] << 33) if wl["ibf_gib"] else (1 << 28)
        self.ibf = dy.Ibf.create(self.ctx, wl["bins"], wl["h"], wl["k"], self.n_bits + 256)
        for b, g in enumerate(genomes):
            self.ibf.add(g, b)
        log("rank %d: %s: filter built on device in %.1f s" % (rank, name, time.time() - t0))
        t0 = time.time()
        self.per_batch = wl.get("batch", wl["reads"])
        self.n_batches = wl["reads"] // self.per_batch
        # a step maps n_batches batches; with one batch per step a second resident batch (other reads of the same simulation) lets consecutive
        # steps alternate, so that the sweep of the batch mapped next can be queued while the current one is mapped (dy_ibf_query_deferred)
        self.n_resident = max(self.n_batches, 2) if args.pipeline != "off" and not paired else self.n_batches
        self.batches, self.truths, self.host_batch0 = [], [], None
        for bi in range(self.n_resident):
            seed = shard_seed(4 if paired else 2, rank) + 7919 * bi
            if paired:
                codes, offs, ptruth = synth.simulate_pairs(genomes, self.per_batch // 2, wl["L"], seed=seed)
                truth_bin = np.concatenate([ptruth["bin"], ptruth["bin"]])
            else:
                codes, offs, t = synth.simulate_reads(genomes, self.per_batch, wl["L"], seed=seed)
                truth_bin = t["bin"]
            self.batches.append(dy.Reads(self.ctx, codes, offs))
            if bi == 0:
                self.host_batch0 = (codes, offs)
                self.truths.append(truth_bin)
        del genomes
        log("rank %d: %s: %d reads simulated + uploaded in %.1f s" % (rank, name, self.n_resident * self.per_batch, time.time() - t0))
        self.error_rate = np.float32(wl["e"] / 100.0)
        self.params = dy.default_params(error_rate=self.error_rate, paired=1 if paired else 0)
        self.thr = dy.threshold(self.error_rate, wl["k"], wl["L"])

    def close(self):
        for r in self.batches:
            r.close()
        self.ibf.close(); self.ctx.close()

    # ---- W warm-up steps, then K timed steps between barriers; returns the leg's numbers and the last step's first-batch result
    def run_leg(self, mode, steps, warmup):
        import torch
        dy, ctx, grp, wl, args = self.dy, self.ctx, self.grp, self.wl, self.args
        rank, world = grp.rank, grp.world
        sm = None
        t_load = time.time()
        if mode == "bins":
            from dream_yara_amd import sharded
            sm = sharded.ShardedMapper(ctx, grp.dist, rank, world, self.idx_dir, wl["bins"], self.ibf, "cuda",
                                       prefetch=args.prefetch == "on" or (args.prefetch == "auto" and world > 1),
                                       overlap_transfers=None if args.overlap_transfers == "auto" else args.overlap_transfers == "on")
            index = sm.index
        else:
            index = dy.Index(ctx, self.idx_dir, wl["bins"])
        info = index.info()
        log("rank %d: %s [%s]: index uploaded in %.1f s (q-mer table: q = %d, %.1f GB)" % (rank, self.name, mode, time.time() - t_load, info["prefix_q"], info["bytes_prefix_table"] / 1e9))
        info = index.info()
        batches, params, thr = self.batches, self.params, self.thr

        nb, per_step = len(batches), self.n_batches
        pipeline = sm is None and nb > 1 and args.pipeline != "off"
        state = {"at": 0, "swept": None, "kept": None}

        def step(keep, more_steps=False):
            """Maps the next per_step batches (round robin over the resident ones).  The sweep of the batch mapped NEXT is registered before the
            current one is mapped (same work inside the timed region: the first step sweeps its own batch, the last one registers nothing)."""
            st_sum, items, raws, st = {k: 0.0 for k in STAGES}, 0, 0, None
            for j in range(per_step):
                reads = batches[state["at"] % nb]
                nxt = batches[(state["at"] + 1) % nb] if (j + 1 < per_step or more_steps) else None
                if sm is not None:
                    res = sm.map(reads, params, thr, next_reads=nxt)
                else:
                    if state["swept"] is not reads:
                        self.ibf.query(reads, thr, want_masks=False)
                    state["swept"] = None
                    if pipeline and nxt is not None and nxt is not reads:
                        self.ibf.query_deferred(nxt, thr)
                        state["swept"] = nxt
                    res = dy.map_reads(ctx, index, reads, params, asynchronous=True)
                ctx.sync()
                st = ctx.stats()
                for k in STAGES:
                    st_sum[k] += st[k]
                items += (st["items"] if sm is None else sm.last["items"] / world)
                raws += st["raw_matches"]
                if keep and state["at"] % nb == 0:                  # the result of the first resident batch: the one the parity sample belongs to
                    if state["kept"] is not None:
                        state["kept"].close()
                    state["kept"] = res
                else:
                    res.close()
                state["at"] += 1
            return st_sum, items, raws, st

        for i in range(warmup):
            step(False, more_steps=i + 1 < warmup)
        ctx.sync()
        state["at"] = 0
        stage_ms = {k: 0.0 for k in STAGES}
        grp.barrier(); torch.cuda.synchronize(); ctx.sync()
        t_start = time.perf_counter()
        items, raws, st = 0, 0, None
        for i in range(steps):
            st_sum, items, raws, st = step(True, more_steps=i + 1 < steps)
            for k in STAGES:
                stage_ms[k] += st_sum[k]
        ctx.sync(); torch.cuda.synchronize(); grp.barrier()
        last = state["kept"]
        elapsed = grp.max_over_ranks(time.perf_counter() - t_start)
        from dream_yara_amd.dist import whole_job_rate
        steps = max(1, steps)
        out = {"value": whole_job_rate(wl["reads"], steps, world, elapsed), "ms_per_step": 1e3 * elapsed / steps, "steps": steps, "warmup": warmup,
               "parallelism": ("reads sharded x%d, filter+index replicated, no data-path collective%s" % (world, "; %d resident batches mapped in turn, the next batch's sweep queued beside the current batch's last aggregation kernels" % nb if pipeline else "")) if sm is None else
                              ("bins sharded x%d (LPT on the bins' sizes: bin mod N for equal bins), filter+contigs replicated, reads->owners and raw matches->home "
                               "by two packed all-to-alls over RCCL%s" % (world, ", next sweep queued in chunks in front of the host's waits" if sm.prefetch else "")),
               "items_per_read": items / wl["reads"], "raw_matches_per_read": raws / wl["reads"],
               "stage_ms_per_step": {k: v / steps for k, v in stage_ms.items()},
               "ibf_kernel": dy.IBF_KERNEL_NAMES.get(st["ibf_kernel"]), "ibf_first_check_slack": st["ibf_slack"], "ibf_tail_pre": st["ibf_tail_pre"], "extend_second_pass_items": st["second_pass_items"],
               "device_tables": {"resident_bins": info["n_bins_resident"], "bwt_rows": info["total_rows"],
                                 "reference_structures_bytes": {"rank_dictionary": info["bytes_rank"], "sparse_sa": info["bytes_sparse_sa"], "contigs_unpacked": info["bytes_contigs"]},
                                 "device_only_bytes": {"expanded_sa": info["bytes_expanded_sa"], "qmer_range_table_q%d" % info["prefix_q"]: info["bytes_prefix_table"],
                                                       "inverse_sa_and_indexed_text": info["bytes_isa_and_text"]},
                                 "filter_bytes": self.n_bits // 8}}
        if sm is not None:
            out["exchange_bytes_per_rank_per_batch"] = {"c1_reads_to_owners": sm.last.get("c1_bytes_out"), "c2_matches_to_home": sm.last.get("c2_bytes_out")}
        rr = last.fetch()
        result = (rr.reads(), rr.matches(), rr.cigars())
        last.close()
        if sm is not None:
            sm.close()
        else:
            index.close()
        return out, result

    # ---- roofline of the dominant HBM kernel K1 from this run's launch time (HIP events on the stream the sweep runs on)
    def roofline(self, leg):
        wl = self.wl
        W = (wl["bins"] + 63) // 64
        bytes_per_read = 2 * (wl["L"] - wl["k"] + 1) * wl["h"] * W * 8
        k1_s = leg["stage_ms_per_step"]["ms_ibf"] / 1e3 / self.n_batches       # average duration of ONE K1 launch
        achieved = bytes_per_read * self.per_batch / k1_s / 1e9 if k1_s > 0 else 0.0
        # HBM traffic of that kernel per launch comes from a committed PMC pass of the same build and workload (rocprofv3 --pmc, separate
        # passes; narrow gathers are accounted at 64 B per request, calibrated on scripts/ubench_gather.hip) — it is NOT measured in this run
        traffic, traffic_src, secondary, line_slots = None, None, None, None
        cands = [os.path.join(ROOT, "profiles", "%s_pmc_%s.json" % (r, self.name)) for r in ("r03", "r02")]
        pmc_file = next((p for p in cands if os.path.exists(p)), None)
        if pmc_file:
            allk = json.load(open(pmc_file))
            kname = (leg["ibf_kernel"] or "").split("<")[0]
            pmc = next((v for kk, v in allk.items() if kname and kname in kk and "FETCH_SIZE" in v), None)
            if pmc:
                traffic_src = "profiles/" + os.path.basename(pmc_file)
                scale = self.per_batch / pmc["SQ_WAVES"] if pmc.get("SQ_WAVES") else 1.0         # one wavefront per read: per-read figures, this run's batch
                traffic = (pmc["FETCH_SIZE"] + pmc.get("WRITE_SIZE", 0.0)) * 1024.0 * scale
                if "TCC_EA0_RDREQ_sum" in pmc and k1_s > 0:
                    # every L2 miss, narrow or wide, occupies one 128-B line slot of the memory side (profiles/r02_gather_mix.jsonl: random 8-B
                    # gathers and a coalesced stream running together add up in 128-B lines; FETCH_SIZE tallies a request at 64 B)
                    req = pmc["TCC_EA0_RDREQ_sum"] * scale
                    line_slots = {"l2_read_requests_per_launch": req, "bytes_at_128B_per_request": req * 128.0,
                                  "achieved_GBps": req * 128.0 / k1_s / 1e9, "frac_of_hbm_peak": req * 128.0 / k1_s / 1e9 / 8000.0,
                                  "kind": "memory-system utilisation: requests per read from " + traffic_src + " x this run's launch duration"}
            k2 = next((v for kk, v in allk.items() if "k_seed_search" in kk and "TCC_HIT_sum" in v), None)
            if k2:
                hit, miss = k2["TCC_HIT_sum"], k2["TCC_MISS_sum"]
                secondary = {"kernel": "k_seed_search (K2: FM backward search / LF-mapping)", "bound": "l2", "l2_hit_rate": hit / max(1.0, hit + miss),
                             "l2_requests_per_launch": hit + miss, "fetched_bytes_per_launch": k2.get("FETCH_SIZE", 0.0) * 1024.0,
                             "ms_per_launch": leg["stage_ms_per_step"]["ms_search"] / self.n_batches, "source": "profiles/" + os.path.basename(pmc_file)}
        return {"kernel": leg["ibf_kernel"], "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                "kind": "effective rate of the UNPRUNED sweep's algorithmic bytes (%d B/read x %d reads per launch / launch time): the kernel skips look-ups that cannot change a result, "
                        "so this is not memory-system utilisation and can exceed 1 — that is line_slots.frac_of_hbm_peak" % (bytes_per_read, self.per_batch),
                "traffic": traffic, "traffic_source": traffic_src, "line_slots": line_slots, "launch_ms": 1e3 * k1_s,
                "note": "the chip serves 48-60 G random requests/s whatever their width from 8 to 128 B (profiles/r02_gather_ceiling.json: each L2 miss takes one 128-B line slot, "
                        "56 G/s x 128 B = 7.2 TB/s) and a hash position of W <= 16 words is one request, so the sweep is bound by the number of (k-mer, hash) positions it touches; "
                        "first check at nK - threshold + %d k-mers (from the sampled filter fill); at W = 1 only 8 of the 128 B of a line slot are algorithmic bytes" % leg["ibf_first_check_slack"],
                "secondary": secondary}

    def oracle_filter(self, ob):
        words, nb = self.ibf.words()                      # the oracle reads the very filter the device holds
        return ob.OracleIbf(None, words=words, n_bits=nb)

    # ---- every field of the first n reads of the first batch against the oracle (and the oracle's rate on them: the CPU baseline)
    def parity(self, result, n, threads, ob, time_single=0):
        codes, offs = self.host_batch0
        oibf = self.oracle_filter(ob)
        n = min(n, offs.size - 1)
        if self.paired:
            n = offs.size - 1               # a paired batch is [first mates | second mates]: the whole batch
        oidx = ob.OracleIndex(self.idx_dir, self.wl["bins"])
        t0 = time.time()
        ores = ob.map_batch(oidx, oibf, codes[: int(offs[n])], offs[: n + 1], self.error_rate, threads=threads, paired=self.paired)
        dt = time.time() - t0
        single = None
        if time_single:
            n1 = min(n, time_single)
            c1, o1 = (codes[: int(offs[n1])], offs[: n1 + 1]) if not self.paired else first_pairs(codes, offs, n1 // 2)
            t1 = time.time()
            ob.map_batch(oidx, oibf, c1, o1, self.error_rate, threads=1, paired=self.paired)
            single = {"value": n1 / (time.time() - t1), "unit": "reads/s", "cores": 1, "sample": "%d reads, %.1f s" % (n1, time.time() - t1)}
        oidx.close(); oibf.close()
        d_reads, d_matches, d_cigars = result
        nm = int(d_reads["set_begin"][n - 1] + d_reads["best_count"][n - 1] + d_reads["sub_count"][n - 1]) if n else 0
        nc = int(d_reads["cigar_begin"][n - 1] + d_reads["cigar_len"][n - 1]) if n else 0
        diff = compare_all_fields((d_reads[:n], d_matches[:nm], d_cigars[:nc]), ores[:3])
        if self.paired:
            # the pairing weights: the oracle evaluates them with libm like the reference, the device with its fixed IEEE sequences
            if not np.array_equal(d_reads["paired"][:n], ores[0]["paired"]):
                diff.append("paired")
            if not np.allclose(d_reads["pair_prob"][:n], ores[0]["pair_prob"], rtol=1e-9, atol=0.0):
                diff.append("pair_prob")
        return {"n": n, "seconds": dt, "threads": threads, "differs_in": diff, "mapped_fraction": float(ores[0]["mapped"].mean()), "single_thread": single}

    def line_config(self, leg, result):
        rr = result[0]
        mapped = rr["mapped"] == 1
        truth = self.truths[0]
        cfg = {"workload": self.wl["desc"] + (" — PAIRED: %d pairs per step, insert N(400,40)" % (self.wl["reads"] // 2) if self.paired else ""),
               "reads_per_gpu_per_step": self.wl["reads"], "batches_per_step": self.n_batches, "resident_batches": self.n_resident, "parallelism": leg["parallelism"],
               "mapped_fraction": float(mapped.mean()), "primary_in_true_bin": float((rr["primary"]["contig"][mapped] == truth[mapped]).mean()),
               "items_per_read": leg["items_per_read"], "raw_matches_per_read": leg["raw_matches_per_read"],
               "extend_second_pass_items_per_read": leg["extend_second_pass_items"] / max(1, self.per_batch),
               "stage_ms_per_step": leg["stage_ms_per_step"], "device_tables": leg["device_tables"]}
        if self.paired:
            cfg["paired_fraction"] = float(rr["paired"].mean())
        return cfg


def measure_pcie(s, contexts, steps):
    """Upload + sweep + map + fetch per step from (pageable) host memory, with 1..contexts contexts on the device, each driven by its own
    host thread (a context is one stream: the calls are synchronous, a second context overlaps its PCIe hops with the first's kernels)."""
    dy, wl = s.dy, s.wl
    codes, offs = s.host_batch0
    pipes = [(s.ctx, s.ibf, dy.Index(s.ctx, s.idx_dir, wl["bins"]))]
    from dream_yara_amd import synth
    genomes = None
    for i in range(1, contexts):
        ctx = dy.Context(s.grp.local_rank)
        ibf = dy.Ibf.create(ctx, wl["bins"], wl["h"], wl["k"], s.n_bits + 256)
        if genomes is None:
            genomes = [synth.make_genome(b, wl["glen"]) for b in range(wl["bins"])]
        for b, g in enumerate(genomes):
            ibf.add(g, b)
        pipes.append((ctx, ibf, dy.Index(ctx, s.idx_dir, wl["bins"])))
    del genomes
    stage = {"upload": [], "device": [], "fetch": []}

    def run(pipe, n, record):
        ctx, ibf, index = pipe
        for _ in range(n):
            t0 = time.perf_counter()
            reads = dy.Reads(ctx, codes, offs); t1 = time.perf_counter()
            ibf.query(reads, s.thr, want_masks=False)
            r = dy.map_reads(ctx, index, reads, s.params, asynchronous=True); ctx.sync(); t2 = time.perf_counter()
            r.fetch(); t3 = time.perf_counter()
            r.close(); reads.close()
            if record:
                stage["upload"].append(t1 - t0); stage["device"].append(t2 - t1); stage["fetch"].append(t3 - t2)
    out = {}
    for p in pipes:
        run(p, 2, False)
    for n in sorted({1, contexts}):
        ths = [threading.Thread(target=run, args=(pipes[i], steps, n == 1)) for i in range(n)]
        t0 = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        dt = time.perf_counter() - t0
        out["contexts_%d" % n] = {"value": wl["reads"] * steps * n / dt, "unit": "reads/s", "ms_per_batch": 1e3 * dt / (steps * n)}
    out["stage_ms_one_context"] = {k: 1e3 * float(np.median(v)) for k, v in stage.items()}
    out["note"] = "per step: dy_reads_upload of the batch from pageable host memory (+ reverse complements on the device), sweep, map, dy_result_fetch into the ABI's host arrays (+ MAPQ); " \
                  "contexts are independent streams on the one device, each driven by a host thread"
    for ctx, ibf, index in pipes[1:]:
        index.close(); ibf.close(); ctx.close()
    pipes[0][2].close()
    return out


def measure_end_to_end(s, n_reads, ob_sample):
    """The command line tool on a FASTQ file: whole process (start, filter + index load, map, SAM written) and the region BASELINE.md section 3
    names (first read batch loaded -> last SAM byte written = whole process minus the tool's own loading time).  The oracle's command line
    (oracle/_build/oracle_mapper, the CPU restatement) is timed the same way on a sample of the same reads."""
    wl = s.wl
    codes, offs = s.host_batch0
    fpath = os.path.join(s.work_dir, "cli.filter")
    s.ibf.save(fpath)
    block = min(offs.size - 1, 1_000_000)
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    L = wl["L"]
    seqs = lut[codes[: block * L]].reshape(block, L)
    t0 = time.time()
    rec = np.empty((block, 14 + 2 * L), dtype=np.uint8)            # "@r0000000\n" + seq + "\n+\n" + quals + "\n"
    names = np.char.zfill(np.arange(block).astype(str), 7)
    rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
    rec[:, 2:9] = np.frombuffer("".join(names).encode(), dtype=np.uint8).reshape(block, 7)
    rec[:, 9] = 10
    rec[:, 10:10 + L] = seqs
    rec[:, 10 + L] = 10; rec[:, 11 + L] = ord("+"); rec[:, 12 + L] = 10
    rec[:, 13 + L:13 + 2 * L] = ord("I")
    rec[:, 13 + 2 * L] = 10
    blob = rec.tobytes()
    fq = os.path.join(s.work_dir, "cli_%d.fq" % n_reads)
    with open(fq, "wb") as f:
        for _ in range(max(1, n_reads // block)):
            f.write(blob)
    n_total = max(1, n_reads // block) * block
    log("end_to_end: FASTQ of %d reads written in %.1f s" % (n_total, time.time() - t0))
    exe = os.path.join(ROOT, "dream_yara_amd", "csrc", "dream_yara_mapper")
    out_sam = os.path.join(s.work_dir, "cli_out.sam")
    err = os.path.join(s.work_dir, "cli.err")

    def timed(cmd):
        t0 = time.time()
        subprocess.check_call(cmd, stderr=open(err, "w"))
        dt = time.time() - t0
        load = 0.0
        for line in open(err):
            if line.startswith("Loading time:"):                    # dream_yara_mapper -v (like the reference's printStats, src/d_mapper.h:947-959)
                load = float(line.split(":")[1].split()[0])
            elif line.startswith("load_s"):                         # oracle_mapper -v
                load = float(line.split()[1])
        return dt, load
    dt, load = timed([exe, "-fi", fpath, s.idx_dir, fq, "-o", out_sam, "-e", str(wl["e"]), "-rb", "1000000", "-v"])
    res = {"value": n_total / dt, "unit": "reads/s", "reads": n_total, "seconds": dt, "load_seconds": load,
           "after_load": {"value": n_total / max(dt - load, 1e-9), "unit": "reads/s", "note": "BASELINE.md section 3's region: first read batch loaded -> last SAM byte written"},
           "what": "dream_yara_mapper FASTQ (%.1f GB) -> SAM (%.1f GB) on one MI355X, process start + filter/index load + parse + map + format + write" %
                   (os.path.getsize(fq) / 1e9, os.path.getsize(out_sam) / 1e9)}
    os.remove(out_sam)
    if ob_sample:
        fq_s = os.path.join(s.work_dir, "cli_sample.fq")
        with open(fq_s, "wb") as f:
            f.write(blob[: ob_sample * (14 + 2 * L)])
        threads = min(effective_cpus(), 64)
        oexe = os.path.join(ROOT, "oracle", "_build", "oracle_mapper")
        dt_o, load_o = timed([oexe, "-fi", fpath, s.idx_dir, fq_s, "-o", out_sam, "-e", str(wl["e"]), "-t", str(threads), "-v"])
        res["cpu"] = {"value": ob_sample / dt_o, "unit": "reads/s", "reads": ob_sample, "seconds": dt_o, "load_seconds": load_o, "cores": threads, "kind": "port",
                      "after_load": {"value": ob_sample / max(dt_o - load_o, 1e-9), "unit": "reads/s"},
                      "what": "oracle/_build/oracle_mapper (CPU restatement of the reference, not the SeqAn binary), same command line, %d threads" % threads}
        os.remove(out_sam); os.remove(fq_s)
    os.remove(fq); os.remove(fpath)
    return res


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
                         "(on one GPU the waits are short: 5.96 ms per step without, 6.00 ms with)")
    ap.add_argument("--pipeline", default="off", choices=["on", "off"],
                    help="unsharded leg: keep (at least) two resident batches and register the next batch's IBF sweep before the current one is mapped (dy_ibf_query_deferred); "
                         "measured neutral on one GPU (5.10 vs 5.09 ms per step): nothing runs beside a sweep")
    ap.add_argument("--overlap-transfers", default="auto", choices=["auto", "on", "off"],
                    help="bins-sharded leg with --prefetch: a chunk of the next sweep sized to the transfer runs beside each all-to-all(v) (auto = with more than one rank)")
    ap.add_argument("--extras", default="auto", choices=["auto", "on", "off"],
                    help="N = 1: also measure the other geometries, the bins-sharded partitioning through RCCL, the PCIe-inclusive and the end-to-end rates (auto = with the default workload)")
    ap.add_argument("--extras-budget", type=float, default=420.0, help="seconds of wall clock after which no further extra measurement is started")
    ap.add_argument("--e2e-reads", type=int, default=32_000_000)
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE", help="dy_ctx_set_tuning knob for experiments (result-identical code paths; see include/dyhip.h)")
    ap.add_argument("--rehearse", action="store_true", help="N > 1 on a box with fewer GPUs than ranks: ranks share the devices (rank mod devices) and "
                    "talk over gloo — checks the multi-rank code path, its numbers mean nothing")
    ap.add_argument("--work-dir", default=os.environ.get("DY_BENCH_DIR", "/tmp/dy_bench_%d" % os.getuid()))
    args = ap.parse_args()
    capture_stdout()

    from dream_yara_amd.dist import Group
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus))
        raise SystemExit("--gpus (%d) does not match WORLD_SIZE (%d)" % (args.gpus, world))
    extras = world == 1 and args.shard != "bins" and not args.paired and (args.extras == "on" or (args.extras == "auto" and args.workload == "config2"))
    # N > 1: the reads-sharded leg runs first (it cannot hang: no collective on its data path), the bins-sharded leg second, and the line's `value`
    # is the bins-sharded one (north_star's partitioning) whenever that leg delivered
    legs = ["reads", "bins"] if (args.shard == "both" and world > 1) else (["reads"] if args.shard in ("both", "reads") else ["bins"])

    # torch (which brings its own HIP runtime) comes up before libdyhip touches the device: the other order leaves torch without a GPU
    import torch
    need_dist = world > 1 or "bins" in legs or extras
    grp = Group(("gloo" if args.rehearse else "nccl") if need_dist else None, force=need_dist)
    if args.rehearse:
        grp.local_rank %= max(1, torch.cuda.device_count())
    rank = grp.rank
    import dream_yara_amd as dy
    from dream_yara_amd import build as dybuild
    if not os.path.exists(dy.LIB_PATH):
        dybuild.build()
    if dy.device_count() == 0:
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")

    s = Setup(args.workload, args, grp, paired=args.paired)
    line_holder = {}

    def make_line(leg, result):
        return {"metric": "mapped reads/sec (whole node), 100 bp SE, 64-bin IBF",
                "value": leg["value"], "unit": "reads/s", "n_gpus": world, "steps": leg["steps"], "warmup": leg["warmup"],
                "ms_per_step": leg["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "u64", "data": "synthetic", "config": s.line_config(leg, result), "roofline": s.roofline(leg)}

    leg, result = s.run_leg(legs[0], args.steps, args.warmup)
    line = make_line(leg, result)
    line_holder["line"] = line
    log("%s [%s]: %.1f M reads/s, %.3f ms per step" % (args.workload, legs[0], leg["value"] / 1e6, leg["ms_per_step"]))
    if len(legs) > 1:
        # The first leg's line must not be lost to a hang in the second: RCCL all-to-alls at this world size cannot be rehearsed on the
        # one-GPU development boxes, so a watchdog prints what is there and ends the process.
        done = threading.Event()

        def watchdog():
            if not done.wait(args.bins_leg_timeout):
                if rank == 0:
                    out = dict(line_holder["line"]); out["bins_sharded"] = {"error": "no result within %.0f s" % args.bins_leg_timeout}
                    out["value_is"] = "reads_sharded (the bins-sharded leg did not finish)"
                    emit(json.dumps(out))
                os._exit(0)
        threading.Thread(target=watchdog, daemon=True).start()
        try:
            leg2, result2 = s.run_leg(legs[1], args.steps, args.warmup)
            same = not compare_all_fields(result2, result)          # the two partitionings must agree on this rank's reads
            bins_line = make_line(leg2, result2)
            bins_line["roofline"]["k1_runs_alone_in"] = "the reads_sharded leg (launch_ms %.3f); with the next sweep queued in chunks this leg's launch_ms is the sum of its chunks" % line["roofline"]["launch_ms"]
            bins_line["reads_sharded"] = {"value": line["value"], "ms_per_step": line["ms_per_step"], "parallelism": line["config"]["parallelism"],
                                          "stage_ms_per_step": line["config"]["stage_ms_per_step"], "roofline_launch_ms": line["roofline"]["launch_ms"], "roofline_frac": line["roofline"]["frac"]}
            bins_line["config"]["exchange_bytes_per_rank_per_batch"] = leg2.get("exchange_bytes_per_rank_per_batch")
            bins_line["config"]["both_partitionings_give_the_same_results_on_rank0"] = same
            bins_line["value_is"] = "bins_sharded"
            line = bins_line
            log("%s [bins]: %.1f M reads/s, %.3f ms per step" % (args.workload, leg2["value"] / 1e6, leg2["ms_per_step"]))
        except Exception as e:          # noqa: BLE001 — the reads leg's numbers are still valid
            line["bins_sharded"] = {"error": "%s: %s" % (type(e).__name__, e)}
            line["value_is"] = "reads_sharded (the bins-sharded leg failed)"
        done.set()
        line_holder["line"] = line

    ob = None
    if rank == 0 and (args.cpu_sample > 0 or extras):
        ob = oracle_binding()
    if rank == 0 and args.cpu_sample > 0 and legs[0] == "reads":
        # CPU baseline: the oracle (kind "port") on a bounded sample of the same batch; the sample doubles as the parity check of the timed configuration
        sample = args.cpu_sample if world == 1 else min(args.cpu_sample, 200000)
        threads = min(effective_cpus(), 64)
        p = s.parity(result, sample, threads, ob, time_single=200000 if world == 1 else 0)
        base = {"value": p["n"] / p["seconds"], "unit": "reads/s", "cores": threads, "host_cores": os.cpu_count() or 1, "host_cpu_quota": effective_cpus(), "host_cpu": cpu_model(), "kind": "port",
                "sample": "%d reads of the same batch, oracle (oracle/dyo_*.hpp: a restatement of the reference, not the SeqAn binary) with %d threads, %.1f s" % (p["n"], threads, p["seconds"]),
                "region": "map_batch on arrays resident in host memory: filter/index load, FASTQ parsing and SAM writing excluded (end_to_end.cpu times the command line)",
                "mapped_fraction": p["mapped_fraction"], "single_thread": p["single_thread"], "parity_with_gpu_on_sample": not p["differs_in"],
                "parity_fields": "mapped, X0, X1, MAPQ, primary (contig, begin, length, errors, strand), matchesSet, CIGARs" +
                                 (", paired, pair probability (relative 1e-9)" if args.paired else "") + " of the first %d reads" % p["n"]}
        if p["differs_in"]:
            base["parity_differs_in"] = p["differs_in"]
        line["cpu_baseline"] = base
        log("cpu baseline: %.0f reads/s on %d threads, parity %s" % (base["value"], threads, not p["differs_in"]))

    def within_budget(what):
        if time.time() - T_START > args.extras_budget:
            log("extras budget spent: skipping " + what)
            return False
        return True

    if extras:
        # ---- the north-star partitioning at N = 1 through RCCL
        try:
            leg_b, res_b = s.run_leg("bins", args.steps, args.warmup)
            line["bins_sharded"] = {"value": leg_b["value"], "ms_per_step": leg_b["ms_per_step"], "steps": leg_b["steps"], "parallelism": leg_b["parallelism"],
                                    "stage_ms_per_step": leg_b["stage_ms_per_step"], "exchange_bytes_per_rank_per_batch": leg_b.get("exchange_bytes_per_rank_per_batch"),
                                    "overhead_ms_per_step_vs_unsharded": leg_b["ms_per_step"] - leg["ms_per_step"],
                                    "same_results_as_unsharded": not compare_all_fields(res_b, result)}
            log("bins-sharded at N = 1 through RCCL: %.3f ms per step (unsharded %.3f)" % (leg_b["ms_per_step"], leg["ms_per_step"]))
        except Exception as e:          # noqa: BLE001
            line["bins_sharded"] = {"error": "%s: %s" % (type(e).__name__, e)}
        # ---- PCIe-inclusive and end-to-end rates of the headline workload
        if within_budget("pcie_inclusive"):
            try:
                line["pcie_inclusive"] = measure_pcie(s, 3, 8)
                log("pcie-inclusive: %s M reads/s" % {k: round(v["value"] / 1e6, 1) for k, v in line["pcie_inclusive"].items() if k.startswith("contexts")})
            except Exception as e:      # noqa: BLE001
                line["pcie_inclusive"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if within_budget("end_to_end"):
            try:
                line["end_to_end"] = measure_end_to_end(s, args.e2e_reads, 200000)
                log("end to end: %.1f M reads/s (after load %.1f M)" % (line["end_to_end"]["value"] / 1e6, line["end_to_end"]["after_load"]["value"] / 1e6))
            except Exception as e:      # noqa: BLE001
                line["end_to_end"] = {"error": "%s: %s" % (type(e).__name__, e)}
        s.close()
        s = None
        # ---- the other geometries on one GPU
        line["workloads"] = {}
        for name, steps_w in (("northstar", 4), ("config3", 8), ("config5", 5)):
            if not within_budget(name):
                line["workloads"][name] = {"skipped": "extras budget of %.0f s spent" % args.extras_budget}
                continue
            try:
                w = Setup(name, args, grp)
                leg_w, res_w = w.run_leg("reads", steps_w, 1)
                roof = w.roofline(leg_w)
                p = w.parity(res_w, 200000, min(effective_cpus(), 64), ob)
                line["workloads"][name] = {
                    "workload": w.wl["desc"], "value": leg_w["value"], "unit": "reads/s", "ms_per_step": leg_w["ms_per_step"], "steps": steps_w, "warmup": 1,
                    "batches_per_step": w.n_batches, "resident_batches": w.n_resident, "stage_ms_per_step": leg_w["stage_ms_per_step"],
                    "k1": {"kernel": roof["kernel"], "launch_ms": roof["launch_ms"], "effective_algorithmic_frac": roof["frac"], "effective_algorithmic_GBps": roof["achieved"],
                           "line_slot_frac_of_hbm_peak": roof["line_slots"]["frac_of_hbm_peak"] if roof["line_slots"] else None,
                           "line_slot_source": roof["traffic_source"], "first_check_slack": leg_w["ibf_first_check_slack"], "tail_pre": leg_w["ibf_tail_pre"]},
                    "parity_with_oracle": not p["differs_in"], "parity_reads": p["n"], "cpu_oracle_reads_per_s": p["n"] / p["seconds"], "cpu_threads": p["threads"]}
                if p["differs_in"]:
                    line["workloads"][name]["parity_differs_in"] = p["differs_in"]
                log("%s: %.1f M reads/s, %.3f ms per step, K1 %.3f ms, parity %s" % (name, leg_w["value"] / 1e6, leg_w["ms_per_step"], roof["launch_ms"], not p["differs_in"]))
                w.close()
            except Exception as e:      # noqa: BLE001
                line["workloads"][name] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0:
        emit(json.dumps(line))
    grp.barrier()
    if s is not None:
        s.close()
    grp.close()


if __name__ == "__main__":
    main()

"""Bins sharded over ranks: the harness side of include/dyhip.h's dy_shard_c1_* / dy_map_bins / dy_shard_c2_* /
dy_shard_finalize.  One process per GPU; the exchanges are torch.distributed all-to-alls (backend nccl == RCCL over xGMI; with gloo —
the CPU tests and the two-ranks-on-one-GPU rehearsal — the same bytes are staged through the host).  The reference distributes the
same unit: one child Mapper per bin, merged per read by copyMatches / finalizeMainMapper (reference src/d_mapper.h:127-155, 828-838,
876-884).

  C0  sum(read lengths), count         all-reduce    -> the batch's one IBF threshold (src/d_mapper.h:314); once per job here
  C1  reads -> owners of selected bins  all-to-all of G x 32-B headers, then ONE all-to-all(v) of packed segments (ids, lengths, masks, bases)
  C2  raw matches -> home ranks         all-to-all of G rows (counts + this rank's items per bin, which the receivers add up: sortedBins(),
                                        :843-854), then ONE all-to-all(v) of packed segments (16-B raw matches + 40-B pairing records of a paired
                                        job: the child's primary, `paired`, pair probability, rescued mate — what copyCigars ships, :186-215)

Everything is queued on the context's HIP stream, which is torch's current stream while a batch is mapped (torch.cuda.ExternalStream):
RCCL's own stream is ordered against it by events, so the only host synchronisations of a batch are the two fetches of the counts (the
sizes of an all-to-all(v) are host arguments), those inside dy_map_bins and the one that ends dy_shard_finalize.  With `prefetch`, the
IBF sweep of the NEXT batch is registered as soon as C1 is packed (dy_ibf_query_deferred): a chunk of it is queued in front of every
one of those waits, so the device sweeps while the host reads a count or waits for a peer (a whole sweep on a second stream would not
do: it holds every wave slot and everything else runs in its tail).
"""
import contextlib
import os

import numpy as np
import torch

import dream_yara_amd as dy


def bin_sizes(index_dir, n_bins):
    """contigsSum of every bin from <dir>/<b>.txt.size (3 x u64: contigsMaxLength, contigs, contigsSum — reference src/misc_options.h:293-331)."""
    out = np.ones(n_bins, np.uint64)
    for b in range(n_bins):
        try:
            v = np.fromfile(os.path.join(index_dir, "%d.txt.size" % b), dtype="<u8", count=3)
            if v.size == 3:
                out[b] = v[2]
        except OSError:
            pass
    return out


def bin_owner(n_bins, world, sizes=None):
    """Bin -> rank, static per job (the FM structures stay resident): longest-processing-time-first on the bins' text sizes — the static
    counterpart of sortedBins() (reference src/d_mapper.h:843-854) — which is `bin mod G` when the bins are equal (or no sizes are given).
    Same rule as assignBins() in csrc/multi_gpu.hpp."""
    if sizes is None:
        return np.arange(n_bins) % world
    sizes = np.maximum(np.asarray(sizes, dtype=np.uint64), 1)
    order = sorted(range(n_bins), key=lambda b: (-int(sizes[b]), b))
    load = [0] * world
    owner = np.zeros(n_bins, np.int64)
    for b in order:
        r = min(range(world), key=lambda x: (load[x], x))
        owner[b] = r
        load[r] += int(sizes[b])
    return owner


def owner_masks(n_bins, world, owner=None):
    W = (n_bins + 63) // 64
    m = np.zeros((world, W), np.uint64)
    own = bin_owner(n_bins, world) if owner is None else owner
    for b in range(n_bins):
        m[own[b], b // 64] |= np.uint64(1) << np.uint64(b % 64)
    return m


class _DeviceBytes:
    """A library-owned device buffer as seen through __cuda_array_interface__ (zero-copy torch view)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def device_view(ptr, nbytes, device):
    if nbytes == 0:
        return torch.empty(0, dtype=torch.uint8, device=device)
    return torch.as_tensor(_DeviceBytes(ptr, nbytes), device=device)


def exchange_rows(dist, rows, row_bytes, world, stream=None, pinned=None, before_wait=None):
    """rows: uint8 tensor of world x row_bytes, row d = what this rank has for rank d.  Returns (the rows as sent, the rows received) as
    host uint8 arrays of shape (world, row_bytes) — after ONE synchronisation of `stream` (nccl) / of the staging copies (gloo).
    pinned: a reusable pinned uint8 tensor of at least 2 x rows.numel() bytes (allocating one costs milliseconds).
    before_wait: called after the fetch is queued and before the host waits for it (nccl only)."""
    if dist is None:                                               # one rank, no process group: what is sent is what is received
        mine = rows.cpu().numpy().reshape(world, row_bytes)
        return mine, mine
    if dist.get_backend() == "nccl":
        got = torch.empty_like(rows)
        dist.all_to_all_single(got, rows)
        both = pinned[: 2 * rows.numel()] if pinned is not None else torch.empty(2 * rows.numel(), dtype=torch.uint8, pin_memory=True)
        both[: rows.numel()].copy_(rows, non_blocking=True)
        both[rows.numel():].copy_(got, non_blocking=True)
        if before_wait is not None:                                # something useful for the device to do while the host is away (a chunk of the next sweep)
            ev = torch.cuda.Event()
            ev.record(stream or torch.cuda.current_stream())
            before_wait()
            ev.synchronize()
        else:
            (stream or torch.cuda.current_stream()).synchronize()
        host = both.numpy()
        return host[: rows.numel()].reshape(world, row_bytes).copy(), host[rows.numel():].reshape(world, row_bytes).copy()
    mine = rows.cpu().contiguous()
    got = torch.empty_like(mine)
    dist.all_to_all_single(got, mine)
    return mine.numpy().reshape(world, row_bytes), got.numpy().reshape(world, row_bytes)


def exchange_segments(dist, send, send_bytes, recv_bytes, device, during=None):
    """ONE all-to-all(v) of a packed uint8 buffer (segments destination-major).  Returns the received buffer on `device`; with nccl nothing
    waits on the host (work queued on the current stream afterwards sees the data).
    during (nccl): called once the transfer is queued on RCCL's stream and before the current stream is made to wait for it — what it queues on
    the current stream (a chunk of the next batch's sweep) runs beside the transfer, which holds a few workgroups only and started first."""
    if dist is None:
        return send
    ss, rs = [int(x) for x in send_bytes], [int(x) for x in recv_bytes]
    if dist.get_backend() == "nccl":
        out = torch.empty(sum(rs), dtype=torch.uint8, device=device)
        if during is None:
            dist.all_to_all_single(out, send[: sum(ss)], output_split_sizes=rs, input_split_sizes=ss)
        else:
            work = dist.all_to_all_single(out, send[: sum(ss)], output_split_sizes=rs, input_split_sizes=ss, async_op=True)
            during()
            work.wait()                       # a stream-level wait: the host goes on
        return out
    src = send[: sum(ss)].cpu()
    out = torch.empty(sum(rs), dtype=torch.uint8)
    dist.all_to_all_single(out, src, output_split_sizes=rs, input_split_sizes=ss)
    return out.to(device)


def all_reduce_sum(dist, values, device):
    if dist is None:
        return np.array([int(v) for v in values], dtype=np.int64)
    comm_dev = device if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=comm_dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return np.array(t.tolist(), dtype=np.int64)


RAW_MATCH_DTYPE = np.dtype([("read", "<u4"), ("begin", "<u4"), ("len", "<u2"), ("errors", "u1"), ("strand", "u1"), ("bin", "<u2"), ("contig", "<u2")])


def raw_matches(ctx, index, reads, params, n_bins):
    """The raw matches of K2 + K3 (before any aggregation) for `reads`, whose bin masks are in place, against the resident bins of `index`,
    as a RAW_MATCH_DTYPE array in the order the extension appended them — through the one-rank form of the packed exchange (the send
    buffer is the receive buffer: no copy, no communication).  For A/B checks of the extension kernels and for the tests."""
    W = (n_bins + 63) // 64
    stream = torch.cuda.ExternalStream(ctx.stream())
    with torch.cuda.stream(stream):
        hdr = torch.zeros(dy.SEG_HEADER_BYTES, dtype=torch.uint8, device="cuda")
        rows = torch.zeros(dy.shard_c2_row_bytes(n_bins), dtype=torch.uint8, device="cuda")
        send_ptr = dy.shard_c1_pack(ctx, reads, owner_masks(n_bins, 1), bool(params.paired), hdr.data_ptr())
        h = hdr.cpu().numpy().view(dy.SEG_HEADER_DTYPE)
        sub = dy.shard_c1_unpack(ctx, send_ptr, h, n_bins, bool(params.paired))
        loads, n_raw = dy.map_bins(ctx, index, sub, params)
        send_ptr = dy.shard_c2_pack(ctx, rows.data_ptr())
        sub.close()
        r = rows.cpu().numpy()[: dy.SEG_HEADER_BYTES].view(dy.SEG_HEADER_DTYPE)
        assert int(r["n_reads"][0]) == n_raw
        raw = device_view(send_ptr, n_raw * dy.RAW_MATCH_BYTES, "cuda").cpu().numpy().view(RAW_MATCH_DTYPE)
    return raw, loads


class ShardedMapper:
    """One rank of a bin-sharded job: replicated filter + contigs, FM structures of the bins this rank owns.
    dist = None: a single rank without a process group (the packed send buffer is the receive buffer)."""

    def __init__(self, ctx, dist, rank, world, index_dir, n_bins, ibf, device, assignment="lpt", prefetch=False, overlap_transfers=None):
        self.ctx, self.dist, self.rank, self.world, self.ibf, self.device = ctx, dist, rank, world, ibf, device
        self.n_bins, self.W = n_bins, (n_bins + 63) // 64
        self.owner = bin_owner(n_bins, world, bin_sizes(index_dir, n_bins) if assignment == "lpt" else None)
        self.owner_masks = owner_masks(n_bins, world, self.owner)
        self.index = dy.Index(ctx, index_dir, n_bins, owned=(self.owner == rank).astype(np.uint8))
        self.row2 = dy.shard_c2_row_bytes(n_bins)
        self.stream = torch.cuda.ExternalStream(ctx.stream()) if device != "cpu" else None
        self.prefetch = bool(prefetch)
        if self.prefetch:
            ctx.set_tuning("sweep_chunk", 16384)         # chunks of the next sweep in front of the host's waits (count fetches, peers)
        # with a deferred sweep at hand, a chunk of it sized to the transfer runs beside each all-to-all(v) (default: when there are peers to wait for)
        self.overlap_transfers = (world > 1) if overlap_transfers is None else bool(overlap_transfers)
        with self._on_stream():
            self.hdr = torch.zeros(world * dy.SEG_HEADER_BYTES, dtype=torch.uint8, device=device)
            self.rows = torch.zeros(world * self.row2, dtype=torch.uint8, device=device)
        self.pinned = torch.empty(2 * world * max(self.row2, dy.SEG_HEADER_BYTES), dtype=torch.uint8, pin_memory=True) if device != "cpu" else None
        self._swept = None          # the reads object whose sweep is already queued (prefetch)
        self._keep = []
        self.last = {}

    def _on_stream(self):
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def threshold(self, error_rate, kmer_size, sum_len, n_reads):
        tot = all_reduce_sum(self.dist, [sum_len, n_reads], self.device)
        avg = int(tot[0] // max(int(tot[1]), 1))
        return dy.threshold(error_rate, kmer_size, avg)

    def map(self, reads, params, threshold, next_reads=None):
        """reads: this rank's dy.Reads.  Returns a dy.Result for them, equal to the single-GPU result on the whole job.
        next_reads (prefetch mode): the batch the caller maps next — its sweep is queued as soon as this batch's masks are packed (it may be
        the same object for a single-end job; a paired job needs its masks until the end, so pass a different object or None)."""
        ctx, dev, world = self.ctx, self.device, self.world
        paired = bool(params.paired)
        with self._on_stream():
            if self._swept is not reads:
                self.ibf.query(reads, threshold, want_masks=False)                       # K1
            self._swept = None
            # ---- C1: reads -> owners of their selected bins
            send_ptr = dy.shard_c1_pack(ctx, reads, self.owner_masks, paired, self.hdr.data_ptr())
            chunk = None
            if self.prefetch and next_reads is not None and not (paired and next_reads is reads):
                self.ibf.query_deferred(next_reads, threshold)                           # K1 of the next batch: chunks of it go in front of every wait of the host
                self._swept = next_reads
                chunk = ctx.background_chunk
            mine, theirs = exchange_rows(self.dist, self.hdr, dy.SEG_HEADER_BYTES, world, self.stream, self.pinned, chunk)
            h_out = np.ascontiguousarray(mine).view(dy.SEG_HEADER_DTYPE).reshape(world)
            h_in = np.ascontiguousarray(theirs).view(dy.SEG_HEADER_DTYPE).reshape(world)
            send = device_view(send_ptr, int(h_out["bytes"].sum()), dev)
            recv = exchange_segments(self.dist, send, h_out["bytes"], h_in["bytes"], dev, self._beside(chunk, h_out["bytes"]))
            sub = dy.shard_c1_unpack(ctx, recv.data_ptr() if recv.numel() else None, h_in, self.n_bins, paired)
            dy.map_bins(ctx, self.index, sub, params, want_counts=False)               # K2 + K3 (+ pairing, rescue) on the owner
            # ---- C2: raw matches (+ pairing records) -> the read's home rank
            send_ptr = dy.shard_c2_pack(ctx, self.rows.data_ptr())
            sub.close()
            mine, theirs = exchange_rows(self.dist, self.rows, self.row2, world, self.stream, self.pinned, chunk)
            r_out = np.ascontiguousarray(mine[:, : dy.SEG_HEADER_BYTES]).view(dy.SEG_HEADER_DTYPE).reshape(world)
            r_in = np.ascontiguousarray(theirs[:, : dy.SEG_HEADER_BYTES]).view(dy.SEG_HEADER_DTYPE).reshape(world)
            send2 = device_view(send_ptr, int(r_out["bytes"].sum()), dev)
            recv2 = exchange_segments(self.dist, send2, r_out["bytes"], r_in["bytes"], dev, self._beside(chunk, r_out["bytes"]))
            res = dy.shard_finalize(ctx, self.index, reads, params, recv2.data_ptr() if recv2.numel() else None, theirs, world)    # K4 on the home rank
            loads = np.ascontiguousarray(theirs[:, dy.SEG_HEADER_BYTES: dy.SEG_HEADER_BYTES + 4 * self.n_bins]).view("<u4").reshape(world, self.n_bins)
            self.last = {"sent_reads": int(h_out["n_reads"].sum()), "recv_reads": int(h_in["n_reads"].sum()), "raw_out": int(r_out["n_reads"].sum()),
                         "raw_in": int(r_in["n_reads"].sum()), "pe_items_in": int(r_in["n_bases"].sum()), "items": int(loads.astype(np.int64).sum()),
                         "c1_bytes_out": int(h_out["bytes"].sum()), "c2_bytes_out": int(r_out["bytes"].sum())}
            self._keep = [recv, recv2, send, send2]
        return res

    def _beside(self, chunk, bytes_out):
        """What runs beside an all-to-all(v): a chunk of the pending sweep about as long as the transfer (bytes to the peers over ~300 GB/s of xGMI
        links; the sweep passes ~285 M reads/s: one read per 1 000 bytes on the links), at least the default chunk."""
        if chunk is None or not self.overlap_transfers or self.dist is None or self.dist.get_backend() != "nccl":
            return None
        remote = int(bytes_out.sum()) - int(bytes_out[self.rank])
        n = max(16384, remote // 1000)
        return lambda: chunk(n)

    def close(self):
        # torch's caching allocator ties the blocks it handed out under the context's stream to that stream (and records events on it): they
        # go back to the driver here, while the stream still exists — the caller destroys it with the context
        self._keep = []
        self.hdr = self.rows = self.pinned = None
        if self.stream is not None:
            self.stream.synchronize()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
        self.index.close()

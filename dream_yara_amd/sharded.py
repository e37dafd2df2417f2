"""Bins sharded over ranks: the harness side of include/dyhip.h's dy_shard_* / dy_map_bins /
dy_finalize_raw.  One process per GPU; the three exchanges are torch.distributed all-to-alls on caller-owned device
buffers (backend nccl == RCCL over xGMI; with gloo — the CPU tests and the one-GPU rehearsal — the same bytes are
staged through the host).  The reference distributes the same unit: one child Mapper per bin, merged per read by
copyMatches / finalizeMainMapper (reference src/d_mapper.h:127-155, 828-838, 876-884).

  C0  sum(read lengths), count        all-reduce   -> the batch's one IBF threshold (src/d_mapper.h:314)
  C1  reads -> owners of selected bins all-to-all(v) ids, lengths, masks, bases
  C2  raw matches -> home ranks        all-to-all(v) 16 B records (+ 40 B pairing records per (read,bin) for paired jobs: the child's
                                                     primary, `paired`, pair probability, rescued mate — what copyCigars ships, :186-215)
  C3  items per bin                    all-reduce   -> sortedBins() order for the CIGAR provenance rule (:843-854)
"""
import numpy as np
import torch

import dream_yara_amd as dy


def bin_owner(n_bins, world):
    """Static `bin mod G` assignment (the FM structures stay resident, so ownership cannot follow the per-batch load)."""
    return np.arange(n_bins) % world


def owner_masks(n_bins, world):
    W = (n_bins + 63) // 64
    m = np.zeros((world, W), np.uint64)
    own = bin_owner(n_bins, world)
    for b in range(n_bins):
        m[own[b], b // 64] |= np.uint64(1) << np.uint64(b % 64)
    return m


def exchange_counts(dist, rows, device):
    """all-to-all of a small int64 matrix: rows[d] = the numbers this rank has for rank d.  Returns what every rank has for this one."""
    backend = dist.get_backend()
    comm_dev = device if backend == "nccl" else "cpu"
    s = torch.tensor(np.asarray(rows, dtype=np.int64), dtype=torch.int64, device=comm_dev).contiguous()
    r = torch.empty_like(s)
    dist.all_to_all_single(r, s)
    return r.cpu().numpy()


def exchange_data(dist, sends, send_splits, recv_splits, device):
    """Several all-to-all(v)s of uint8 tensors with known receive sizes, queued together and awaited once (the per-collective latency
    is what a small batch pays; the counts travel in ONE exchange_counts beforehand).  Returns the received tensors on `device`."""
    backend = dist.get_backend()
    comm_dev = device if backend == "nccl" else "cpu"
    outs, works, keep = [], [], []
    for send, ss, rs in zip(sends, send_splits, recv_splits):
        out = torch.empty(int(sum(rs)), dtype=torch.uint8, device=comm_dev)
        src = send if backend == "nccl" else send.cpu()
        keep.append(src)                                    # a staged copy must outlive its queued collective
        works.append(dist.all_to_all_single(out, src, output_split_sizes=[int(x) for x in rs], input_split_sizes=[int(x) for x in ss], async_op=True))
        outs.append(out)
    for w in works:
        w.wait()
    return [o.to(device) for o in outs]


def exchange_bytes(dist, send, send_splits, device):
    """all-to-all(v) of a uint8 tensor split per destination rank.  Returns (recv tensor on `device`, recv splits)."""
    counts = exchange_counts(dist, [[int(x)] for x in send_splits], device)
    recv_splits = [int(x) for x in counts[:, 0]]
    out, = exchange_data(dist, [send], [send_splits], [recv_splits], device)
    return out, recv_splits


def all_reduce_sum(dist, values, device):
    comm_dev = device if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=comm_dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return np.array(t.tolist(), dtype=np.int64)


class ShardedMapper:
    """One rank of a bin-sharded job: replicated filter + contigs, FM structures of `bin mod world == rank`."""

    def __init__(self, ctx, dist, rank, world, index_dir, n_bins, ibf, device):
        self.ctx, self.dist, self.rank, self.world, self.ibf, self.device = ctx, dist, rank, world, ibf, device
        self.n_bins, self.W = n_bins, (n_bins + 63) // 64
        self.owner_masks = owner_masks(n_bins, world)
        self.index = dy.Index(ctx, index_dir, n_bins, owned=(bin_owner(n_bins, world) == rank).astype(np.uint8))

    def threshold(self, error_rate, kmer_size, sum_len, n_reads):
        tot = all_reduce_sum(self.dist, [sum_len, n_reads], self.device)
        avg = int(tot[0] // max(int(tot[1]), 1))
        return dy.threshold(error_rate, kmer_size, avg)

    def map(self, reads, params, threshold):
        """reads: this rank's dy.Reads.  Returns a dy.Result for them, equal to the single-GPU result on the whole job."""
        ctx, dev, W = self.ctx, self.device, self.W
        u8 = lambda n: torch.empty(int(n), dtype=torch.uint8, device=dev)
        paired = bool(params.paired)
        self.ibf.query(reads, threshold, want_masks=False)                          # K1
        n_to, b_to = dy.shard_plan(ctx, reads, self.owner_masks, paired)            # C1 pack (paired: mates vote and travel together)
        n_tot, b_tot = int(n_to.sum()), int(b_to.sum())
        ids, lens, masks, seqs = u8(n_tot * 4), u8(n_tot * 4), u8(n_tot * W * 8), u8(b_tot)
        torch.cuda.synchronize()
        dy.shard_pack(ctx, reads, ids.data_ptr(), lens.data_ptr(), masks.data_ptr(), seqs.data_ptr())
        # C1: one exchange of (reads, bases) per pair of ranks, then the four buffers together
        got = exchange_counts(self.dist, np.stack([n_to.astype(np.int64), b_to.astype(np.int64)], axis=1), dev)
        n_from, b_from = got[:, 0], got[:, 1]
        r_ids, r_lens, r_masks, r_seqs = exchange_data(
            self.dist, [ids, lens, masks, seqs],
            [n_to.astype(np.int64) * 4, n_to.astype(np.int64) * 4, n_to.astype(np.int64) * W * 8, b_to.astype(np.int64)],
            [n_from * 4, n_from * 4, n_from * W * 8, b_from], dev)
        rs = [int(x) * 4 for x in n_from]
        torch.cuda.synchronize()
        n_recv = r_ids.numel() // 4
        seg = np.zeros(self.world + 1, np.uint32); seg[1:] = np.cumsum([x // 4 for x in rs])
        sub = dy.Reads.from_device(ctx, r_seqs.data_ptr(), r_lens.data_ptr(), r_masks.data_ptr(), n_recv, r_seqs.numel(), self.n_bins,
                                   paired_seg_starts=seg if paired else None)
        loads, n_raw = dy.map_bins(ctx, self.index, sub, params)                     # K2 + K3 (+ pairing, rescue) on the owner
        raw = u8(n_raw * dy.RAW_MATCH_BYTES)
        torch.cuda.synchronize()
        raw_to = dy.raw_pack(ctx, seg, r_ids.data_ptr(), raw.data_ptr())             # C2 pack
        r_items, n_items_in = None, 0
        items, items_to = None, np.zeros(self.world, np.uint64)
        if paired:
            items = u8(int(loads.sum()) * dy.PE_ITEM_BYTES)
            items_to = dy.pe_items_pack(ctx, seg, r_ids.data_ptr(), items.data_ptr())
            items = items[: int(items_to.sum()) * dy.PE_ITEM_BYTES]
        sub.close()
        # C2: one exchange of (raw matches, pairing records) per pair of ranks, then the buffers together
        got = exchange_counts(self.dist, np.stack([raw_to.astype(np.int64), items_to.astype(np.int64)], axis=1), dev)
        raw_from, items_from = got[:, 0], got[:, 1]
        sends, ssp, rsp = [raw], [raw_to.astype(np.int64) * dy.RAW_MATCH_BYTES], [raw_from * dy.RAW_MATCH_BYTES]
        if paired:
            sends.append(items); ssp.append(items_to.astype(np.int64) * dy.PE_ITEM_BYTES); rsp.append(items_from * dy.PE_ITEM_BYTES)
        recv = exchange_data(self.dist, sends, ssp, rsp, dev)
        r_raw = recv[0]
        if paired:
            r_items = recv[1]
            n_items_in = r_items.numel() // dy.PE_ITEM_BYTES
        g_loads = all_reduce_sum(self.dist, loads, dev)                              # C3
        torch.cuda.synchronize()
        self.last = {"sent_reads": n_tot, "recv_reads": n_recv, "raw_out": n_raw, "raw_in": r_raw.numel() // dy.RAW_MATCH_BYTES,
                     "items": int(g_loads.sum())}
        self.last["pe_items_in"] = n_items_in
        res = dy.finalize_raw(ctx, self.index, reads, params, r_raw.data_ptr(), r_raw.numel() // dy.RAW_MATCH_BYTES,
                              g_loads.astype(np.uint32), r_items.data_ptr() if n_items_in else None, n_items_in)    # K4 on the home rank
        return res

    def close(self):
        self.index.close()

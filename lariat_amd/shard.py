"""Barcode-range sharding across the GPUs of a node.

Barcodes are independent work units (lariat.go:461-547 touches only its own WorkUnit; the reference's own parallelism is a
worker pool over barcodes, lariat.go:348-350), so rank r of W takes a contiguous range of the barcode-sorted input,
balanced by pair count, with the index replicated in its HBM; there is no collective on the data path.  The only
communication is a barrier and a MAX over ranks of the elapsed time (bench.py) and, at the end, the host-side
concatenation of the per-GPU BAM shards in range order (lh_bam_concat).
"""
import numpy as np


def barcode_ranges(bc_pair_off, world, weights=None):
    """split barcodes [0,nb) into `world` contiguous ranges; returns [(b0,b1)] * world.  Without `weights`: ~equal pair counts.  With `weights` (one number per
    barcode: what the barcode costs — its pairs where all pairs cost alike; a pair on a repeat family costs two orders of magnitude more than one on unique
    sequence (bench.py's `repeats` and `mixed` legs), so a host that knows, say from an earlier pass, which barcodes are repeat-rich passes the cells or the
    candidates it counted): ~equal weight."""
    bc_pair_off = np.asarray(bc_pair_off, dtype=np.int64)
    nb = len(bc_pair_off) - 1
    if weights is None:
        cum = bc_pair_off.astype(np.float64)
    else:
        w = np.asarray(weights, dtype=np.float64)
        assert len(w) == nb and (w >= 0).all()
        cum = np.concatenate([[0.0], np.cumsum(w)])
    total = float(cum[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        b = int(np.searchsorted(cum, target, side="left"))
        b = min(max(b, cuts[-1]), nb)
        cuts.append(b)
    cuts.append(nb)
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


def slice_batch(seq, seq_off, bc_pair_off, name_seed, b0, b1):
    """the arrays of barcodes [b0,b1) of a barcode-sorted batch, re-based (what rank r uploads)"""
    p0, p1 = int(bc_pair_off[b0]), int(bc_pair_off[b1])
    s0, s1 = int(seq_off[2 * p0]), int(seq_off[2 * p1])
    return (seq[s0:s1], seq_off[2 * p0:2 * p1 + 1] - s0, (np.asarray(bc_pair_off[b0:b1 + 1]) - p0).astype(np.int32), name_seed[p0:p1])


def align_rank_shard(lib, idx, seq, seq_off, bc_pair_off, name_seed, rank, world, opts=None, max_pairs_per_batch=1 << 20, weights=None):
    """what ONE rank of a `world`-GPU job does: its barcode range of the sorted input through lh_align_barcodes, batch by batch
    (whole barcodes per batch).  `weights`: a cost per barcode for barcode_ranges (default: its pairs) — every rank must pass the same.
    Returns ((b0, b1), [Result ...]); nothing is exchanged with other ranks."""
    from . import capi
    b0, b1 = barcode_ranges(bc_pair_off, world, weights=weights)[rank]
    out = []
    if b1 <= b0:
        return (b0, b1), out
    sizes = np.diff(np.asarray(bc_pair_off[b0:b1 + 1], dtype=np.int64))
    ctx = idx.context(int(max(min(max_pairs_per_batch, sizes.sum()), sizes.max())))
    k = b0
    while k < b1:
        e, n = k, 0
        while e < b1 and (e == k or n + sizes[e - b0] <= max_pairs_per_batch):
            n += sizes[e - b0]
            e += 1
        s, so, bo, ns = slice_batch(seq, seq_off, bc_pair_off, name_seed, k, e)
        out.append(ctx.align_barcodes(capi.Batch.from_arrays(s, so, bo, ns), opts))
        k = e
    return (b0, b1), out


def reduce_max(dist, value, device="cpu"):
    """max over ranks of a python float (used for the elapsed time)"""
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

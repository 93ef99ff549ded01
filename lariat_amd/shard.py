"""Barcode-range sharding across the GPUs of a node.

Barcodes are independent work units (lariat.go:461-547 touches only its own WorkUnit), so rank r of W takes a
contiguous range of the barcode-sorted input, balanced by pair count; there is no collective on the data path.  The
only communication is a barrier and a MAX over ranks of the elapsed time (bench.py) — and, in a full pipeline, the
host-side concatenation of per-GPU BAM shards in range order.
"""
import numpy as np


def barcode_ranges(bc_pair_off, world):
    """split barcodes [0,nb) into `world` contiguous ranges with ~equal pair counts; returns [(b0,b1)] * world"""
    bc_pair_off = np.asarray(bc_pair_off, dtype=np.int64)
    nb = len(bc_pair_off) - 1
    total = int(bc_pair_off[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        b = int(np.searchsorted(bc_pair_off, target, side="left"))
        b = min(max(b, cuts[-1]), nb)
        cuts.append(b)
    cuts.append(nb)
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


def reduce_max(dist, value, device="cpu"):
    """max over ranks of a python float (used for the elapsed time)"""
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

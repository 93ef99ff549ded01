"""N>1 control path on CPU: two gloo ranks shard a barcode-sorted batch by barcode range, align their shards
independently (with the oracle standing in for the device, which this container lacks), and the concatenation in range
order equals the single-process result.  No data-path collective is used — only gather-for-checking and a MAX reduce."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers


def _worker(rank, world, port, out):
    sys.path.insert(0, helpers.ROOT)
    sys.path.insert(0, os.path.join(helpers.ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_py
    from lariat_amd import capi, shard
    o = oracle_py.load()
    names, contigs = helpers.small_genome()
    idx = o.index_build_naive(names, contigs)
    rs = helpers.small_reads(names, contigs, n_barcodes=7, pairs=20, seed=3)
    b0, b1 = shard.barcode_ranges(rs.bc_pair_off, world)[rank]
    sub = rs.slice_barcodes(b0, b1)
    res = idx.align_barcodes(helpers.batch_of(sub))
    act = res.active_idx
    mine = dict(range=(b0, b1), pos=res.pos[act].tolist(), mapq=res.mapq[act].tolist(), rid=res.rid[act].tolist())
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    tmax = shard.reduce_max(dist, 1.0 + rank)
    if rank == 0:
        full = idx.align_barcodes(helpers.batch_of(rs))
        fa = full.active_idx
        cat = {k: sum((g[k] for g in gathered), []) for k in ("pos", "mapq", "rid")}
        ok = (cat["pos"] == full.pos[fa].tolist() and cat["mapq"] == full.mapq[fa].tolist() and cat["rid"] == full.rid[fa].tolist())
        ranges = [g["range"] for g in gathered]
        with open(out, "w") as f:
            f.write("%d %s %s\n" % (int(ok), tmax, ranges))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_barcode_sharding(tmp_path):
    out = str(tmp_path / "r.txt")
    port = 29500 + (os.getpid() % 1000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    ok, tmax, ranges = open(out).read().split(" ", 2)
    assert ok == "1"
    assert float(tmax) == 2.0
    assert "(0, " in ranges


def test_barcode_ranges_balanced():
    from lariat_amd import shard
    off = np.concatenate([[0], np.cumsum([100] * 10 + [5] * 40 + [300] * 3)])
    for w in (1, 2, 4, 8):
        rg = shard.barcode_ranges(off, w)
        assert rg[0][0] == 0 and rg[-1][1] == len(off) - 1
        assert all(rg[i][1] == rg[i + 1][0] for i in range(w - 1))
        loads = [off[b1] - off[b0] for b0, b1 in rg]
        assert max(loads) <= off[-1] / w + 300

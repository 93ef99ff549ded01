"""differential fuzzing of the HIP path against the oracle over random genomes / read sets / options (development aid; the
pytest suite holds the fixed cases).  Stops at the first difference and prints the seed that reproduces it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers, oracle_py
from lariat_amd import capi, synth


def run_case(lib, oracle, seed):
    """one random genome / index path / read set / option set, product against oracle; raises on a difference"""
    rng = np.random.default_rng(seed)
    ncont = int(rng.integers(1, 5))
    lens = [int(rng.integers(60000, 400000)) for _ in range(ncont)]
    names = ["c%d" % i for i in range(ncont)]
    contigs = synth.make_genome(lens, seed=seed, n_dup=int(rng.integers(0, 25)), dup_len=int(rng.integers(500, 6000)), dup_identity=float(rng.uniform(0.97, 1.0)),
                                n_rep_family=int(rng.integers(0, 6)), rep_len=int(rng.integers(100, 400)), rep_copies=int(rng.integers(5, 60)))
    # repeat FAMILIES (r04): n copies that all resemble each other — a read on one of them has n candidates, its pair up to 50 + 50 rescue attempts,
    # thousands of (alignment, mate) combinations; reads are then drawn on and around the copies
    windows = None
    if rng.random() < 0.4:
        windows = []
        for _ in range(int(rng.integers(1, 4))):
            unit = int(rng.integers(150, 2500))
            cons = rng.integers(0, 4, size=unit).astype(np.uint8)
            for _ in range(int(rng.integers(4, 28))):
                seg = cons.copy()
                m = rng.random(unit) < float(rng.uniform(0.0, 0.03))
                seg[m] = (seg[m] + rng.integers(1, 4, size=int(m.sum()))) & 3
                if rng.random() < 0.3 and unit > 60:   # an indel between the copies
                    at = int(rng.integers(20, unit - 20))
                    seg = np.concatenate([seg[:at], seg[at + int(rng.integers(1, 6)):], rng.integers(0, 4, size=8).astype(np.uint8)])[:unit]
                if rng.random() < 0.5:
                    seg = (3 - seg[::-1]).astype(np.uint8)
                k = int(rng.integers(0, ncont))
                if lens[k] < unit + 2000:
                    continue
                at = int(rng.integers(0, lens[k] - unit))
                contigs[k][at:at + unit] = seg
                windows.append((k, max(0, at - 1200), min(lens[k], at + unit + 1200)))
    oidx = oracle.index_build_naive(names, contigs)
    iopts = {}
    if rng.random() < 0.5:
        iopts["ktree_levels"] = int(rng.choice([-1, 3, 8, 11, 12]))
    if rng.random() < 0.2:
        iopts["sb_shift"] = int(rng.choice([10, 14, 20]))
    if rng.random() < 0.5:   # the index built on the device from the .pac image instead of uploaded
        pac, l_pac, _, _ = lib.reference_pack(contigs)
        offs = np.concatenate([[0], np.cumsum(lens)])
        idx = lib.index_build_device(pac, l_pac, [(names[i], lens[i], int(offs[i])) for i in range(ncont)], build_chunk_log2=int(rng.choice([0, 12, 16])), **iopts)
    else:
        idx = lib.index_from_arrays(oidx.arrays(), **iopts)
    if rng.random() < 0.3:
        idx.resample_sa(int(rng.choice([2, 8, 32])))
    alt = None
    if ncont > 1 and rng.random() < 0.3:   # some contigs are ALT contigs (mem_chain_flt's is_alt rule, the regions' is_alt)
        alt = (rng.random(ncont) < 0.5).astype(np.uint8)
        idx.set_alt(alt); oidx.set_alt(alt)
    l1, l2 = int(rng.integers(50, 240)), int(rng.integers(50, 240))
    src_contigs, src_names = contigs, names
    if windows and rng.random() < 0.7:   # the reads' molecules lie on the copies (the reads are aligned against the whole genome all the same)
        src_contigs = [contigs[k][a:b_] for k, a, b_ in windows if b_ - a >= 2400]
        src_names = ["w%d" % i for i in range(len(src_contigs))]
        if not src_contigs:
            src_contigs, src_names = contigs, names
    ins = dict(ins_mean=float(rng.choice([350, 470])), ins_sd=float(rng.choice([50, 120])), ins_max=int(rng.choice([700, 900]))) if rng.random() < 0.4 else {}
    rs = synth.make_reads(src_contigs, src_names, n_barcodes=int(rng.integers(1, 12)), pairs_per_barcode=int(rng.integers(1, 120)), seed=seed + 7, len1=l1, len2=l2,
                          sub_lo=0.0, sub_hi=float(rng.uniform(0.0, 0.06)), indel_rate=float(rng.choice([0.0, 0.001, 0.01])), junk_frac=float(rng.choice([0.0, 0.05, 0.3])),
                          mol_min=1 if windows else 4, mol_max=4 if windows else 10, **ins)
    if rng.random() < 0.5:   # sprinkle ambiguous bases
        k = rng.integers(0, len(rs.seq), size=max(1, len(rs.seq) // int(rng.integers(20, 400))))
        rs.seq[k] = 4
    rfa = (rng.random(len(rs.bc_pair_off) - 1) < 0.8).astype(np.uint8)
    b = capi.Batch.from_arrays(rs.seq, rs.seq_off, rs.bc_pair_off, rs.name_seed, bc_do_rfa=rfa)
    kw = {}
    if rng.random() < 0.2:
        kw["flags"] = int(rng.choice([capi.LH_F_NO_SWEEP_FILTER, capi.LH_F_EXT_SERIAL, capi.LH_F_EXT_WAVE, capi.LH_F_CHAIN_WAVE, capi.LH_F_P2_TASKS, capi.LH_F_P2_TASKS | capi.LH_F_NO_SWEEP_FILTER, capi.LH_F_RESCUE_FULL, 0, 0]))
    if rng.random() < 0.3:
        kw = dict(b=int(rng.integers(2, 7)), o_del=int(rng.integers(3, 9)), o_ins=int(rng.integers(3, 9)), e_del=int(rng.integers(1, 3)), e_ins=int(rng.integers(1, 3)),
                  w=int(rng.choice([20, 100])), zdrop=int(rng.choice([50, 100])), min_seed_len=int(rng.choice([15, 19, 25])))
    copts = {"big_slots": int(rng.integers(1, 4))} if rng.random() < 0.1 else {}
    what = "contigs %s, alt %s, families %s, reads %dx%d/%d %s, opts %s, index %s, context %s" % (lens, None if alt is None else alt.tolist(), len(windows) if windows else 0, l1, l2, rs.n_pairs, ins, kw, iopts, copts)
    try:
        ctx = idx.context(rs.n_pairs, **copts)
        okw = {k: v for k, v in kw.items() if k != "flags"}
        helpers.assert_same_dump(ctx.stage_dump(b, lib.opts(**kw)), oidx.stage_dump(b, oracle.opts(**okw)), helpers.DUMP_FRONT + helpers.DUMP_REGS)
        want = oidx.align_barcodes(b, oracle.opts(**okw), threads=16)
        if rng.random() < 0.5:
            helpers.assert_same_result(ctx.align_barcodes(b, lib.opts(**kw)), want, inference=True)
        else:   # the split boundary: resident slot, align, download in two steps
            ctx.upload_slot(1, b); ctx.select(1); ctx.align_resident(lib.opts(**kw)); ctx.download_begin()
            helpers.assert_same_result(ctx.download_end(), want, inference=True)
    except Exception as e:
        raise AssertionError("seed %d (%s): %s" % (seed, what, str(e)[:600]))


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    lib = capi.load_library()
    oracle = oracle_py.load()
    t_end = time.time() + budget
    it = 0
    while time.time() < t_end:
        try:
            run_case(lib, oracle, seed0 + it)
        except AssertionError as e:
            print("DIFF at " + str(e), flush=True)
            sys.exit(1)
        it += 1
    print("fuzz ok: %d cases from seed %d in %.0f s" % (it, seed0, budget))

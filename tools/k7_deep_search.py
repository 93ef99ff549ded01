"""development aid for K7's second look (k_aln.h, aln_deep_check): reads with equal spans and five to seven mismatches on low-complexity sequence, where paths
with gaps come close to the diagonal or beat it.  Each batch goes through the emulator build of the product, the two `weak` builds (one check of the proof left
out each) and the oracle; the product must agree with the oracle on every read; the reads on which a weak build does NOT are the adversarial cases (the test
suite runs two batches of the same generator, helpers.k7_deep_batch: tests/test_emu_front.py).

    make -C tests/hipemu weak1 weak2 && python tools/k7_deep_search.py [--batches 20]
"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers, oracle_py
from lariat_amd import capi


def differing_reads(res, ref):
    """read indices whose candidates differ in position / CIGAR / mismatch counts"""
    bad = []
    for r in range(ref.n_reads):
        ca, cb = list(res.cands_of_read(r)), list(ref.cands_of_read(r))
        same = len(ca) == len(cb)
        for x, y in zip(ca, cb):
            if not same: break
            for f in ("pos", "aend", "mismatches", "indels", "nm", "score"):
                if int(getattr(res, f)[x]) != int(getattr(ref, f)[y]): same = False
            if same and not np.array_equal(res.cigar_of(x), ref.cigar_of(y)): same = False
        if not same: bad.append(r)
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=10)
    ap.add_argument("--reads", type=int, default=192)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--gpu", action="store_true", help="the HIP library against the oracle only (no weak builds), scoring drawn per batch")
    a = ap.parse_args()
    oracle = oracle_py.load()
    build = os.path.join(ROOT, "tests", "_build")
    if a.gpu:
        libs = {"product": capi.load_library()}
    else:
        libs = {"product": capi.Library(os.path.join(build, "liblariat_emu.so")), "weak1": capi.Library(os.path.join(build, "liblariat_emu_weak1.so")),
                "weak2": capi.Library(os.path.join(build, "liblariat_emu_weak2.so"))}
    kept = {"weak1": [], "weak2": []}
    n_deep = n_noindel = 0
    for bi in range(a.batches):
        names, contigs, reads = helpers.k7_deep_batch(a.seed + bi, a.reads)
        oidx = oracle.index_build_naive(names, contigs)
        b = capi.Batch(reads, [0, len(reads) // 2])
        kw = {}
        if a.gpu and bi % 2:
            rng = np.random.default_rng(a.seed + bi)
            kw = dict(b=int(rng.integers(2, 7)), o_del=int(rng.integers(3, 9)), o_ins=int(rng.integers(3, 9)), e_del=int(rng.integers(1, 3)), e_ins=int(rng.integers(1, 3)))
        ref = oidx.align_barcodes(b, oracle.opts(run_inference=0, **kw), threads=8)
        ok = ref.rid >= 0
        listed = exec_ = 0
        n_noindel += int(((ref.indels[ok] == 0) & (ref.mismatches[ok] >= 5) & (ref.mismatches[ok] <= 7)).sum())
        for name, lib in libs.items():
            idx = lib.index_from_arrays(oidx.arrays())
            res = idx.context(len(reads) // 2).align_barcodes(b, lib.opts(run_inference=0, **kw))
            bad = differing_reads(res, ref)
            if name == "product":
                listed, exec_ = res.counters["n_glob_listed"], res.counters["n_glob_exec"]
                assert not bad, "the product differs from the oracle: seed %d opts %s reads %s" % (a.seed + bi, kw, bad[:8])
            else:
                for r in bad:
                    if r % 2 == 0:
                        kept[name].append((a.seed + bi, r, reads[r], reads[r + 1], contigs[0]))
        if a.gpu and bi % 20 != 19:
            continue
        print("batch %d: %d reads, the reference's DPs %d, listed by the first look %d, by the second %d; ungapped results with 5-7 mismatches so far %d; weak1 wrong on %d, weak2 on %d"
              % (bi, len(reads), ref.counters["n_glob_exec"], listed, exec_, n_noindel, len(kept["weak1"]), len(kept["weak2"])), flush=True)


if __name__ == "__main__":
    main()

"""The synthetic hg38-scale workload of BASELINE.json's configs: no human reference exists offline, so the
genome is generated (seed 20261002, iid ACGT with GC 0.41) with hg38's contig structure — 24 contigs in the proportions of
chr1..22, X, Y — and the reads by the linked-read model of lh_synth_reads."""

# hg38 primary assembly chromosome lengths, bases
HG38_LENGTHS = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622, 133275309,
                114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415]
HG38_NAMES = ["chr%d" % i for i in range(1, 23)] + ["chrX", "chrY"]
GENOME_SEED = 20261002
READS_SEED = 20261003


def hg38_like_contigs(total_bases):
    """[(name, len, offset)]: hg38's 24 primary contigs scaled so that they sum to about `total_bases` (each a multiple of 4)"""
    tot = float(sum(HG38_LENGTHS))
    out, off = [], 0
    for name, ln in zip(HG38_NAMES, HG38_LENGTHS):
        n = max(4000, int(ln / tot * total_bases) // 4 * 4)
        out.append((name, n, off))
        off += n
    return out

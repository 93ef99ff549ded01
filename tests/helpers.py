"""shared helpers for the test-suite"""
import gzip
import os

import numpy as np

from lariat_amd import capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
PHIX = os.path.join(GOLDEN, "phix", "PhiX.fa")

# go/src/test/gobwa_test.go:18,23
PHIX_READ_A = "TCAAAAACTGACGCGTTGGATGAGGAGAAGTGGCTTAATATGCTTGGCACGTTCGTCAAGGACTGGTTTA"
PHIX_READ_B = "TATGACCAGTGTTTCCAGTCCGTTCAGTTGTTGCAGTGGAATAGTCAGGTTAAATTTAATGTGACCGCTT"


def read_fasta(path):
    names, seqs = [], []
    for line in open(path):
        if line.startswith(">"):
            names.append(line[1:].split()[0])
            seqs.append([])
        else:
            seqs[-1].append(line.strip())
    return names, ["".join(s) for s in seqs]


def read_fastq9(path, trim):
    """minimal 9-line reader (reader.go:91-174 semantics that matter here: trim the first `trim` bases of read 1)"""
    op = gzip.open if open(path, "rb").read(2) == b"\x1f\x8b" else open
    lines = [l.rstrip(b"\n") for l in op(path, "rb").read().split(b"\n")]
    recs = []
    for i in range(0, len(lines) - 8, 9):
        name, r1, q1, r2, q2, bc = lines[i:i + 6]
        recs.append(dict(name=name[1:].decode(), r1=r1[trim:].decode(), r2=r2.decode(), bc=bc.decode()))
    return recs


def small_genome(seed=1):
    names = ["chrA", "chrB", "chrC"]
    contigs = synth.make_genome([300000, 200000, 100000], seed=seed, n_dup=6, dup_len=5000, dup_identity=0.99, n_rep_family=2, rep_copies=20)
    return names, contigs


def small_reads(names, contigs, n_barcodes=8, pairs=40, seed=5, junk=0.02):
    return synth.make_reads(contigs, names, n_barcodes=n_barcodes, pairs_per_barcode=pairs, seed=seed, junk_frac=junk)


def batch_of(rs):
    return capi.Batch.from_arrays(rs.seq, rs.seq_off, rs.bc_pair_off, rs.name_seed)

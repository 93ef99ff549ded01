"""ctypes mirror of include/lariat_hip.h.

Plumbing only: struct layouts, numpy views over result arenas, and the loader of
``liblariat_hip.so``.  The product is the shared library; there is NO CPU fallback here —
if the HIP library is missing or no device is present every entry point raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LARIAT_HIP_LIB") or os.path.join(_HERE, "_build", "liblariat_hip.so")

LH_OK = 0
LH_ABI_VERSION = 5
LH_E_ARG, LH_E_IO, LH_E_HIP, LH_E_CAPACITY, LH_E_NODEVICE, LH_E_LIMIT = 1, 2, 3, 4, 5, 6
# lh_opts.flags
LH_REC_DEBUG_TAGS = 1
LH_F_NO_SWEEP_FILTER, LH_F_EXT_WAVE, LH_F_EXT_SERIAL, LH_F_CHAIN_WAVE, LH_F_P2_TASKS, LH_F_RESCUE_FULL = 1, 16, 32, 64, 128, 256
LH_MAX_READ_LEN = 250

c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_u8p = C.POINTER(C.c_uint8)
c_u32p = C.POINTER(C.c_uint32)
c_u64p = C.POINTER(C.c_uint64)
c_f64p = C.POINTER(C.c_double)


class LhOpts(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("a", C.c_int32), ("b", C.c_int32), ("o_del", C.c_int32), ("e_del", C.c_int32), ("o_ins", C.c_int32), ("e_ins", C.c_int32),
        ("pen_unpaired", C.c_int32), ("pen_clip5", C.c_int32), ("pen_clip3", C.c_int32),
        ("w", C.c_int32), ("zdrop", C.c_int32), ("T", C.c_int32),
        ("min_seed_len", C.c_int32), ("min_chain_weight", C.c_int32), ("max_chain_extend", C.c_int32),
        ("split_factor", C.c_float),
        ("split_width", C.c_int32), ("max_occ", C.c_int32), ("max_chain_gap", C.c_int32), ("max_ins", C.c_int32),
        ("mask_level", C.c_float), ("drop_ratio", C.c_float), ("XA_drop_ratio", C.c_float), ("mask_level_redun", C.c_float),
        ("mapQ_coef_len", C.c_float),
        ("max_mem_intv", C.c_int32), ("max_matesw", C.c_int32),
        ("pes_low", C.c_int32), ("pes_high", C.c_int32),
        ("rescue_score_delta", C.c_int32), ("rescue_max_hits", C.c_int32), ("aln_score_delta", C.c_int32),
        ("improper_pair_penalty", C.c_double), ("genome_length", C.c_double),
        ("run_inference", C.c_int32), ("flags", C.c_uint32),
    ]


class LhIndexOpts(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("sa_intv", C.c_int32), ("sb_shift", C.c_int32), ("no_kmer_table", C.c_int32), ("no_unique_runs", C.c_int32),
                ("no_sweep_filter", C.c_int32), ("build_chunk_log2", C.c_int32), ("ktree_levels", C.c_int32)]


class LhContextOpts(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("smem_grid", C.c_int32), ("aln_grid", C.c_int32), ("rfa_grid", C.c_int32), ("rfa_slab_kb", C.c_int32),
                ("lanes", C.c_int32), ("big_slots", C.c_int32), ("rfa_tier_kb", C.c_int32 * 2), ("rfa_tier_grid", C.c_int32 * 2), ("reserved", C.c_int32)]


class LhBatch(C.Structure):
    _fields_ = [
        ("n_barcodes", C.c_int32), ("n_pairs", C.c_int32),
        ("bc_pair_off", c_i32p), ("bc_do_rfa", c_u8p),
        ("seq_off", c_i64p), ("seq", c_u8p), ("name_seed", c_u64p),
        ("cen_start", c_i64p), ("cen_end", c_i64p),
    ]


_RESULT_CAND_FIELDS = [  # (name, ctype pointer, numpy dtype) — one entry per candidate
    ("rid", c_i32p, np.int32), ("pos", c_i64p, np.int64), ("aend", c_i64p, np.int64), ("rb", c_i64p, np.int64), ("re", c_i64p, np.int64),
    ("reversed", c_u8p, np.uint8), ("score", c_i32p, np.int32), ("qb", c_i32p, np.int32), ("qe", c_i32p, np.int32), ("nm", c_i32p, np.int32),
    ("matches", c_i32p, np.int32), ("mismatches", c_i32p, np.int32), ("indels", c_i32p, np.int32),
    ("soft_clipped", c_i32p, np.int32), ("soft_clipped_length", c_i32p, np.int32), ("in_filtered", c_u8p, np.uint8),
]
_RESULT_INF_FIELDS = [
    ("active", c_u8p, np.uint8), ("is_proper", c_u8p, np.uint8), ("bwa_pick", c_u8p, np.uint8), ("active_molecule", c_u8p, np.uint8),
    ("duplicate", c_u8p, np.uint8), ("molecule_id", c_i32p, np.int32), ("mapq", c_i32p, np.int32),
    ("molecule_difference", c_f64p, np.float64), ("molecule_confidence", c_f64p, np.float64),
    ("sum_move_probability_change", c_f64p, np.float64), ("mate_idx", c_i64p, np.int64),
]
_RESULT_READ_FIELDS = [
    ("active_idx", c_i64p, np.int64), ("second_best_idx", c_i64p, np.int64), ("second_best_score", c_f64p, np.float64),
    ("as_score", c_f64p, np.float64), ("split_idx", c_i64p, np.int64), ("split_mapq", c_i32p, np.int32),
    ("split_second_best", c_f64p, np.float64), ("split_score", c_f64p, np.float64),
]
_COUNTERS = ["n_ext", "n_lf", "n_sa", "win_bases", "n_chain_ext", "ext_cells", "glob_cells", "n_rescue", "rescue_cells", "n_ext_exec_p1", "n_ext_exec_p2", "n_ext_exec_p3", "n_ktree_p1", "n_ktree_p2", "n_ktree_p3", "n_calls_by_text", "rescue_cells_exec", "n_glob_listed", "n_glob_exec"]


class LhResult(C.Structure):
    _fields_ = (
        [("abi_version", C.c_int32), ("n_reads", C.c_int32), ("n_cand", C.c_int64), ("cand_off", c_i64p)]
        + [(n, t) for n, t, _ in _RESULT_CAND_FIELDS]
        + [("cigar_off", c_i64p), ("cigar", c_u32p), ("mm_off", c_i64p), ("mm_ref_loc", c_i32p), ("mm_read_loc", c_i32p),
           ("log_alignment_probability", c_f64p)]
        + [(n, t) for n, t, _ in _RESULT_INF_FIELDS]
        + [(n, t) for n, t, _ in _RESULT_READ_FIELDS]
        + [(n, C.c_uint64) for n in _COUNTERS]
        + [("arena_", C.c_void_p)]
    )


_DUMP_FIELDS = [
    ("intv_off", c_i64p, np.int64, "reads1"), ("intv", c_u64p, np.uint64, "intv4"),
    ("seed_off", c_i64p, np.int64, "reads1"), ("seed_rbeg", c_i64p, np.int64, "seed"), ("seed_qbeg", c_i32p, np.int32, "seed"),
    ("seed_len", c_i32p, np.int32, "seed"), ("seed_rid", c_i32p, np.int32, "seed"),
    ("chain_off", c_i64p, np.int64, "reads1"), ("chain_nseeds", c_i32p, np.int32, "chain"), ("chain_rid", c_i32p, np.int32, "chain"),
    ("chain_w", c_i32p, np.int32, "chain"), ("chain_kept", c_i32p, np.int32, "chain"), ("chain_pos", c_i64p, np.int64, "chain"),
    ("reg_off", c_i64p, np.int64, "reads1"), ("reg_rb", c_i64p, np.int64, "reg"), ("reg_re", c_i64p, np.int64, "reg"),
    ("reg_qb", c_i32p, np.int32, "reg"), ("reg_qe", c_i32p, np.int32, "reg"), ("reg_rid", c_i32p, np.int32, "reg"),
    ("reg_score", c_i32p, np.int32, "reg"), ("reg_truesc", c_i32p, np.int32, "reg"), ("reg_w", c_i32p, np.int32, "reg"),
    ("reg_seedcov", c_i32p, np.int32, "reg"), ("reg_seedlen0", c_i32p, np.int32, "reg"), ("reg_csub", c_i32p, np.int32, "reg"),
    ("reg_secondary", c_i32p, np.int32, "reg"),
]


class LhStageDump(C.Structure):
    _fields_ = [("n_reads", C.c_int32)] + [(n, t) for n, t, _, _ in _DUMP_FIELDS] + [("arena_", C.c_void_p)]


def _view(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(int(n),)).astype(dtype, copy=True)


class Result:
    """Owned numpy copy of an lh_result (so the arena can be freed immediately)."""

    def __init__(self, r: LhResult):
        self.n_reads = int(r.n_reads)
        self.n_cand = int(r.n_cand)
        self.cand_off = _view(r.cand_off, self.n_reads + 1, np.int64)
        for n, _, dt in _RESULT_CAND_FIELDS + _RESULT_INF_FIELDS:
            setattr(self, n, _view(getattr(r, n), self.n_cand, dt))
        self.log_alignment_probability = _view(r.log_alignment_probability, self.n_cand, np.float64)
        self.cigar_off = _view(r.cigar_off, self.n_cand + 1, np.int64)
        self.cigar = _view(r.cigar, self.cigar_off[-1] if self.n_cand else 0, np.uint32)
        self.mm_off = _view(r.mm_off, self.n_cand + 1, np.int64)
        nmm = self.mm_off[-1] if self.n_cand else 0
        self.mm_ref_loc = _view(r.mm_ref_loc, nmm, np.int32)
        self.mm_read_loc = _view(r.mm_read_loc, nmm, np.int32)
        for n, _, dt in _RESULT_READ_FIELDS:
            setattr(self, n, _view(getattr(r, n), self.n_reads, dt))
        self.counters = {n: int(getattr(r, n)) for n in _COUNTERS}

    def as_struct(self):
        """an LhResult whose pointers reference this object's arrays (kept alive on the struct)"""
        r = LhResult()
        r.n_reads, r.n_cand = self.n_reads, self.n_cand
        keep = []
        for name, ctype in LhResult._fields_:
            if name in ("abi_version", "n_reads", "n_cand", "arena_") or name in _COUNTERS:
                continue
            arr = np.ascontiguousarray(getattr(self, name))
            keep.append(arr)
            setattr(r, name, arr.ctypes.data_as(ctype))
        r._keep = keep
        return r

    def cigar_of(self, i):
        return self.cigar[self.cigar_off[i]:self.cigar_off[i + 1]]

    def cigar_str(self, i):
        return "".join("%d%s" % (c >> 4, "MIDSH"[c & 0xF]) for c in self.cigar_of(i))

    def cands_of_read(self, r):
        return range(int(self.cand_off[r]), int(self.cand_off[r + 1]))


class StageDump:
    def __init__(self, d: LhStageDump):
        self.n_reads = int(d.n_reads)
        offs = {}
        for n, _, dt, kind in _DUMP_FIELDS:
            if kind == "reads1":
                offs[n] = _view(getattr(d, n), self.n_reads + 1, np.int64)
                setattr(self, n, offs[n])
        sizes = {"intv4": int(offs["intv_off"][-1]) * 4, "seed": int(offs["seed_off"][-1]), "chain": int(offs["chain_off"][-1]),
                 "reg": int(offs["reg_off"][-1])}
        for n, _, dt, kind in _DUMP_FIELDS:
            if kind != "reads1":
                setattr(self, n, _view(getattr(d, n), sizes[kind], dt))
        self.intv = self.intv.reshape(-1, 4)


class Batch:
    """Host-side packing of barcode-grouped read pairs into an lh_batch (keeps the numpy buffers alive)."""

    def __init__(self, reads_nt4, bc_pair_off, name_seed=None, bc_do_rfa=None, cen_start=None, cen_end=None):
        # reads_nt4: list of 2*n_pairs uint8 arrays (R1,R2,R1,R2...) already nt4-coded and trimmed
        n_reads = len(reads_nt4)
        assert n_reads % 2 == 0
        self.n_pairs = n_reads // 2
        lens = np.fromiter((len(x) for x in reads_nt4), dtype=np.int64, count=n_reads)
        self.seq_off = np.zeros(n_reads + 1, dtype=np.int64)
        np.cumsum(lens, out=self.seq_off[1:])
        self.seq = np.concatenate([np.asarray(x, dtype=np.uint8) for x in reads_nt4]) if n_reads else np.zeros(0, np.uint8)
        if self.seq.size == 0:
            self.seq = np.zeros(1, np.uint8)
        self._init_rest(bc_pair_off, name_seed, bc_do_rfa, cen_start, cen_end)

    @classmethod
    def from_arrays(cls, seq, seq_off, bc_pair_off, name_seed=None, bc_do_rfa=None, cen_start=None, cen_end=None):
        self = cls.__new__(cls)
        self.seq = np.ascontiguousarray(seq, dtype=np.uint8)
        self.seq_off = np.ascontiguousarray(seq_off, dtype=np.int64)
        self.n_pairs = (len(self.seq_off) - 1) // 2
        self._init_rest(bc_pair_off, name_seed, bc_do_rfa, cen_start, cen_end)
        return self

    def _init_rest(self, bc_pair_off, name_seed, bc_do_rfa, cen_start, cen_end):
        self.bc_pair_off = np.ascontiguousarray(bc_pair_off, dtype=np.int32)
        self.n_barcodes = len(self.bc_pair_off) - 1
        assert self.bc_pair_off[0] == 0 and self.bc_pair_off[-1] == self.n_pairs
        self.name_seed = np.ascontiguousarray(name_seed if name_seed is not None else np.arange(1, self.n_pairs + 1), dtype=np.uint64)
        self.bc_do_rfa = np.ascontiguousarray(bc_do_rfa if bc_do_rfa is not None else np.ones(self.n_barcodes), dtype=np.uint8)
        self.cen_start = None if cen_start is None else np.ascontiguousarray(cen_start, dtype=np.int64)
        self.cen_end = None if cen_end is None else np.ascontiguousarray(cen_end, dtype=np.int64)
        b = LhBatch()
        b.n_barcodes = self.n_barcodes
        b.n_pairs = self.n_pairs
        b.bc_pair_off = self.bc_pair_off.ctypes.data_as(c_i32p)
        b.bc_do_rfa = self.bc_do_rfa.ctypes.data_as(c_u8p)
        b.seq_off = self.seq_off.ctypes.data_as(c_i64p)
        b.seq = self.seq.ctypes.data_as(c_u8p)
        b.name_seed = self.name_seed.ctypes.data_as(c_u64p)
        b.cen_start = self.cen_start.ctypes.data_as(c_i64p) if self.cen_start is not None else None
        b.cen_end = self.cen_end.ctypes.data_as(c_i64p) if self.cen_end is not None else None
        self.c = b


_NT4 = np.full(256, 4, dtype=np.uint8)
for _i, _ch in enumerate("ACGT"):
    _NT4[ord(_ch)] = _i
    _NT4[ord(_ch.lower())] = _i


def sequence_convert(seq):
    """SequenceConvert (gobwa.go:159-167): ASCII -> nt4."""
    if isinstance(seq, str):
        seq = seq.encode()
    return _NT4[np.frombuffer(seq, dtype=np.uint8)]


def _declare(L):
    L.lh_last_error.restype = C.c_char_p
    L.lh_device_count.restype = C.c_int
    L.lh_opts_init.argtypes = [C.POINTER(LhOpts)]
    L.lh_index_opts_init.argtypes = [C.POINTER(LhIndexOpts)]
    L.lh_index_opts_init.restype = None
    L.lh_context_opts_init.argtypes = [C.POINTER(LhContextOpts)]
    L.lh_context_opts_init.restype = None
    L.lh_index_load.argtypes = [C.c_char_p, C.c_int, C.POINTER(LhIndexOpts), C.POINTER(C.c_void_p)]
    L.lh_index_from_arrays.argtypes = [C.c_int, C.c_uint64, c_u64p, c_u32p, C.c_uint64, C.c_int32, c_u64p, C.c_uint64, c_u8p, C.c_int64,
                                       C.c_int32, c_i64p, c_i32p, C.POINTER(C.c_char_p), C.POINTER(LhIndexOpts), C.POINTER(C.c_void_p)]
    L.lh_index_build_device.argtypes = [C.c_int, c_u8p, C.c_int64, C.c_int32, c_i64p, c_i32p, C.POINTER(C.c_char_p), C.POINTER(LhIndexOpts), C.POINTER(C.c_void_p)]
    L.lh_index_export.argtypes = [C.c_void_p, c_u64p, c_u64p, c_u32p, c_u64p, C.c_int32, c_u64p, c_u64p]
    L.lh_index_save.argtypes = [C.c_void_p, C.c_char_p]
    L.lh_synth_genome.argtypes = [C.c_uint64, C.c_double, C.c_int64, c_u8p, C.c_int32]
    L.lh_synth_reads.argtypes = [c_u8p, C.c_int64, C.c_int32, c_i64p, c_i32p, C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double,
                                 C.c_double, C.c_double, C.c_int32, C.c_int32, C.c_int32, c_u8p, c_i64p, c_i32p, c_u64p, c_i32p, c_i64p, c_i64p]
    L.lh_index_contigs.argtypes = [C.c_void_p, c_i32p, C.POINTER(C.POINTER(C.c_char_p)), C.POINTER(c_i64p), C.POINTER(c_i64p)]
    L.lh_index_l_pac.argtypes = [C.c_void_p]
    L.lh_index_l_pac.restype = C.c_int64
    L.lh_ingest_open.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    L.lh_ingest_open.restype = C.c_int
    L.lh_ingest_next.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.POINTER(LhIngestBatch))]
    L.lh_ingest_next.restype = C.c_int
    L.lh_ingest_batch_free.argtypes = [C.POINTER(LhIngestBatch)]
    L.lh_ingest_batch_free.restype = None
    L.lh_ingest_close.argtypes = [C.c_void_p]
    L.lh_ingest_close.restype = None
    L.lh_records_text.argtypes = [C.POINTER(LhResult), C.POINTER(LhIngestBatch), C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    L.lh_records_text.restype = C.c_int
    L.lh_records_text_ex.argtypes = [C.POINTER(LhResult), C.POINTER(LhIngestBatch), C.c_int32, C.POINTER(C.c_char_p), C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    L.lh_records_text_ex.restype = C.c_int
    L.lh_bam_set_flags.argtypes = [C.c_void_p, C.c_int32]
    L.lh_bam_set_flags.restype = C.c_int
    L.lh_bam_open.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_char_p), c_i64p, C.c_char_p, C.c_int32, C.c_int32, C.c_char_p, C.c_int32, C.POINTER(C.c_void_p)]
    L.lh_bam_open.restype = C.c_int
    L.lh_bam_append.argtypes = [C.c_void_p, C.POINTER(LhResult), C.POINTER(LhIngestBatch)]
    L.lh_bam_append.restype = C.c_int
    L.lh_bam_close.argtypes = [C.c_void_p]
    L.lh_bam_close.restype = C.c_int
    L.lh_records_free.argtypes = [C.c_void_p]
    L.lh_records_free.restype = None
    L.lh_name_seed.argtypes = [C.c_char_p, C.c_int64]
    L.lh_name_seed.restype = C.c_uint64
    L.lh_index_resample_sa.argtypes = [C.c_void_p, C.c_int32]
    L.lh_index_resample_sa.restype = C.c_int
    L.lh_index_sa_interval.argtypes = [C.c_void_p]
    L.lh_index_sa_interval.restype = C.c_int32
    L.lh_index_free.argtypes = [C.c_void_p]
    L.lh_index_build.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(c_u8p), c_i64p, C.c_int32]
    L.lh_context_create.argtypes = [C.c_void_p, C.c_int64, C.POINTER(LhContextOpts), C.POINTER(C.c_void_p)]
    L.lh_context_free.argtypes = [C.c_void_p]
    L.lh_align_barcodes.argtypes = [C.c_void_p, C.POINTER(LhOpts), C.POINTER(LhBatch), C.POINTER(C.POINTER(LhResult))]
    L.lh_batch_upload.argtypes = [C.c_void_p, C.POINTER(LhBatch)]
    L.lh_align_resident.argtypes = [C.c_void_p, C.POINTER(LhOpts)]
    L.lh_batch_upload_slot.argtypes = [C.c_void_p, C.c_int32, C.POINTER(LhBatch)]
    L.lh_batch_select.argtypes = [C.c_void_p, C.c_int32]
    L.lh_result_download.argtypes = [C.c_void_p, C.POINTER(C.POINTER(LhResult))]
    L.lh_result_free.argtypes = [C.POINTER(LhResult)]
    L.lh_last_timings.argtypes = [C.c_void_p, c_i32p, C.POINTER(C.POINTER(C.c_char_p)), C.POINTER(C.POINTER(C.c_float))]
    L.lh_stage_dump_resident.argtypes = [C.c_void_p, C.POINTER(LhOpts), C.POINTER(C.POINTER(LhStageDump))]
    L.lh_stage_dump_free.argtypes = [C.POINTER(LhStageDump)]
    L.lh_get_seq.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_char_p]
    L.lh_device_memory.argtypes = [C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.lh_diag_random_read.argtypes = [C.c_int, C.c_int64, C.c_int32, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.lh_diag_rescue_sw.argtypes = [C.c_int, C.c_int32, c_i32p, c_u8p, c_i32p, c_u8p, C.c_int32, C.c_int32, c_i32p]
    L.lh_diag_valu_rate.argtypes = [C.c_int, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.c_int32]
    return L


class LhError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("liblariat_hip error %d: %s" % (code, msg))
        self.code = code


class Library:
    """A loaded C-ABI library (the product liblariat_hip.so; tests may point it at the emulator build)."""

    def __init__(self, path=None):
        path = path or LIB_PATH
        if not os.path.exists(path):
            raise RuntimeError("%s not built: run __graft_entry__.build(); there is no CPU fallback" % path)
        self.path = path
        self.L = _declare(C.CDLL(path))

    def check(self, rc):
        if rc != LH_OK:
            raise LhError(rc, self.L.lh_last_error().decode(errors="replace"))

    def pinned_copy(self, arr):
        """a copy of a numpy array in page-locked host memory (lh_host_alloc); freed with the returned array's `_lh_pin` holder"""
        a = np.ascontiguousarray(arr)
        self.L.lh_host_alloc.argtypes = [C.c_size_t]
        self.L.lh_host_alloc.restype = C.c_void_p
        self.L.lh_host_free.argtypes = [C.c_void_p]
        p = self.L.lh_host_alloc(max(1, a.nbytes))
        if not p:
            raise LhError(5, self.L.lh_last_error().decode(errors="replace"))
        lib = self

        class _Pin:
            def __init__(self, ptr):
                self.ptr = ptr

            def __del__(self):
                if self.ptr:
                    lib.L.lh_host_free(self.ptr)
                    self.ptr = None

        hold = _Pin(p)
        buf = (C.c_uint8 * max(1, a.nbytes)).from_address(p)
        out = np.frombuffer(buf, dtype=a.dtype, count=a.size).reshape(a.shape)
        out[...] = a
        self._pins = getattr(self, "_pins", [])
        self._pins.append((out, hold))   # (kept until the library object goes away: the array must not outlive its memory)
        return out

    def device_count(self):
        return self.L.lh_device_count()

    def diag_gosort(self, first, keys, device=0):
        """(perm_serial, perm_wave): Go's sort.Sort over the index spaces [first[k], first[k+1]) of keys, by K8's two restatements"""
        first = np.ascontiguousarray(first, dtype=np.int32)
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        n = int(first[-1])
        ps, pw = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
        self.L.lh_diag_gosort.argtypes = [C.c_int, C.c_int32, c_i32p, c_i64p, c_i32p, c_i32p]
        self.check(self.L.lh_diag_gosort(device, len(first) - 1, first.ctypes.data_as(c_i32p), keys.ctypes.data_as(c_i64p), ps.ctypes.data_as(c_i32p), pw.ctypes.data_as(c_i32p)))
        return ps, pw

    def diag_gosort_split(self, keys, limit, device=0):
        """Go's sort.Sort of keys as K8 sorts a long contig list: ranges longer than `limit` by the whole wave, the others with the depth left; returns the permutation"""
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        perm = np.zeros(len(keys), dtype=np.int32)
        self.L.lh_diag_gosort_split.argtypes = [C.c_int, C.c_int32, c_i64p, c_i32p, C.c_int32]
        self.check(self.L.lh_diag_gosort_split(device, len(keys), keys.ctypes.data_as(c_i64p), perm.ctypes.data_as(c_i32p), limit))
        return perm

    def diag_bitonic(self, keys, block, device=0):
        """keys (distinct u64) ascending by K8's sorting network, `block` places in LDS at a time (64 / 1024; 0: all steps in memory)"""
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        out = np.zeros(len(keys), dtype=np.uint64)
        self.L.lh_diag_bitonic.argtypes = [C.c_int, C.c_int32, c_u64p, c_u64p, C.c_int32]
        self.check(self.L.lh_diag_bitonic(device, len(keys), keys.ctypes.data_as(c_u64p), out.ctypes.data_as(c_u64p), block))
        return out

    def diag_introsort(self, first, keys, device=0):
        """(perm_serial, perm_wave): klib's ks_introsort over the index spaces [first[k], first[k+1]) of keys, by the one-lane and the wave-wide restatement"""
        first = np.ascontiguousarray(first, dtype=np.int32)
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        n = int(first[-1])
        ps, pw = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
        self.L.lh_diag_introsort.argtypes = [C.c_int, C.c_int32, c_i32p, c_i64p, c_i32p, c_i32p]
        self.check(self.L.lh_diag_introsort(device, len(first) - 1, first.ctypes.data_as(c_i32p), keys.ctypes.data_as(c_i64p), ps.ctypes.data_as(c_i32p), pw.ctypes.data_as(c_i32p)))
        return ps, pw

    def device_memory(self, device=0):
        """(free, total) bytes of device memory"""
        f, t = C.c_int64(0), C.c_int64(0)
        self.check(self.L.lh_device_memory(device, C.byref(f), C.byref(t)))
        return f.value, t.value

    def opts(self, **kw):
        o = LhOpts()
        self.L.lh_opts_init(C.byref(o))
        for k, v in kw.items():
            if not hasattr(o, k):
                raise AttributeError(k)
            setattr(o, k, v)
        return o

    def diag_random_read(self, table_bytes, granule_bytes, n_access, device=0):
        g, ms = C.c_double(), C.c_double()
        self.check(self.L.lh_diag_random_read(device, int(table_bytes), int(granule_bytes), int(n_access), C.byref(g), C.byref(ms)))
        return g.value, ms.value

    def diag_rescue_sw(self, queries, windows, full=False, weaken=0, device=0):
        """lh_diag_rescue_sw: per case (score, te, qe, tb, qb, rlo, rn, rows2) as an int32 array [n, 8]"""
        n = len(queries)
        q_off = np.zeros(n + 1, dtype=np.int32); t_off = np.zeros(n + 1, dtype=np.int32)
        q_off[1:] = np.cumsum([len(x) for x in queries]); t_off[1:] = np.cumsum([len(x) for x in windows])
        q = np.ascontiguousarray(np.concatenate(queries).astype(np.uint8)); t = np.ascontiguousarray(np.concatenate(windows).astype(np.uint8))
        out = np.zeros((n, 8), dtype=np.int32)
        self.check(self.L.lh_diag_rescue_sw(device, n, q_off.ctypes.data_as(c_i32p), q.ctypes.data_as(c_u8p), t_off.ctypes.data_as(c_i32p), t.ctypes.data_as(c_u8p), int(bool(full)), int(weaken),
                                            out.ctypes.data_as(c_i32p)))
        return out

    def diag_valu_rate(self, op, waves_per_simd, iters=2048, device=0):
        """lh_diag_valu_rate -> dict (k_valu_rate.h)"""
        out = (C.c_double * 10)()
        self.check(self.L.lh_diag_valu_rate(device, int(op), int(waves_per_simd), int(iters), out, 10))
        keys = ("ms", "wave_instr", "ginstr_per_s", "mhz", "cycles_per_instr_wave", "cycles_per_instr_simd", "simds", "min_waves_simd", "max_waves_simd", "min_mhz")
        return dict(zip(keys, list(out)))

    def diag_rescue_dedup(self, first, regions, added, max_chain_gap=10000, device=0):
        """(verdict, n_out) per case — lh_diag_rescue_dedup"""
        first = np.ascontiguousarray(first, dtype=np.int32)
        regions = np.ascontiguousarray(regions, dtype=np.int64).reshape(-1, 6)
        added = np.ascontiguousarray(added, dtype=np.int64).reshape(-1, 6)
        nc = len(first) - 1
        assert len(added) == nc and len(regions) == first[-1]
        v, n = np.zeros(nc, dtype=np.int32), np.zeros(nc, dtype=np.int32)
        self.L.lh_diag_rescue_dedup.argtypes = [C.c_int, C.c_int32, c_i32p, c_i64p, c_i64p, C.c_int32, c_i32p, c_i32p]
        self.check(self.L.lh_diag_rescue_dedup(device, nc, first.ctypes.data_as(c_i32p), regions.ctypes.data_as(c_i64p), added.ctypes.data_as(c_i64p), int(max_chain_gap),
                                               v.ctypes.data_as(c_i32p), n.ctypes.data_as(c_i32p)))
        return v, n

    def diag_go_rand(self, seed, n, device=0):
        """(fast-path u64 draws, ring-path u64 draws, ring-path Float64 draws) of the device's Go math/rand source"""
        a, b, f = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.float64)
        self.L.lh_diag_go_rand.argtypes = [C.c_int, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        self.check(self.L.lh_diag_go_rand(device, int(seed), int(n), a.ctypes.data, b.ctypes.data, f.ctypes.data))
        return a, b, f

    def reference_pack(self, contigs):
        """bns_fasta2bntseq: contigs (nt4 codes or raw FASTA bytes) -> (pac, l_pac, n_ambs, holes=[(offset, len, letter)])"""
        n = len(contigs)
        seqs = [np.ascontiguousarray(np.frombuffer(s, dtype=np.uint8) if isinstance(s, (bytes, bytearray)) else s, dtype=np.uint8) for s in contigs]
        ptrs = (c_u8p * n)(*[s.ctypes.data_as(c_u8p) for s in seqs])
        lens = np.array([len(s) for s in seqs], dtype=np.int64)
        l_pac = int(lens.sum())
        pac = np.zeros(l_pac // 4 + 1, dtype=np.uint8)
        n_ambs = np.zeros(n, dtype=np.int32)
        nh = C.c_int32()
        self.L.lh_reference_pack.argtypes = [C.c_int32, C.POINTER(c_u8p), c_i64p, c_u8p, c_i32p, C.c_int32, c_i64p, c_i32p, C.c_char_p, c_i32p]
        self.L.lh_reference_pack(n, ptrs, lens.ctypes.data_as(c_i64p), pac.ctypes.data_as(c_u8p), n_ambs.ctypes.data_as(c_i32p), 0, None, None, None, C.byref(nh))
        ho = np.zeros(nh.value + 1, dtype=np.int64)
        hl = np.zeros(nh.value + 1, dtype=np.int32)
        hc = C.create_string_buffer(nh.value + 1)
        self.check(self.L.lh_reference_pack(n, ptrs, lens.ctypes.data_as(c_i64p), pac.ctypes.data_as(c_u8p), n_ambs.ctypes.data_as(c_i32p), nh.value,
                                            ho.ctypes.data_as(c_i64p), hl.ctypes.data_as(c_i32p), hc, C.byref(nh)))
        holes = [(int(ho[i]), int(hl[i]), hc.raw[i:i + 1].decode()) for i in range(nh.value)]
        return pac, l_pac, n_ambs, holes

    def index_opts(self, **kw):
        io = LhIndexOpts()
        self.L.lh_index_opts_init(C.byref(io))
        for k, v in kw.items():
            if not hasattr(io, k):
                raise AttributeError(k)
            setattr(io, k, v)
        return io

    def context_opts(self, **kw):
        co = LhContextOpts()
        self.L.lh_context_opts_init(C.byref(co))
        for k, v in kw.items():
            if not hasattr(co, k):
                raise AttributeError(k)
            if isinstance(v, (tuple, list)):
                arr = getattr(co, k)
                for i, x in enumerate(v):
                    arr[i] = int(x)
            else:
                setattr(co, k, v)
        return co

    def index_load(self, prefix, device=0, **index_opts):
        h = C.c_void_p()
        io = self.index_opts(**index_opts)
        self.check(self.L.lh_index_load(prefix.encode(), device, C.byref(io), C.byref(h)))
        return Index(self, h)

    def index_build_device(self, pac, l_pac, contigs, device=0, **index_opts):
        """contigs = [(name, len, off)]; pac = the .pac image (uint8, l_pac // 4 + 1 bytes)"""
        n = len(contigs)
        names = (C.c_char_p * n)(*[c[0].encode() for c in contigs])
        lens = np.array([c[1] for c in contigs], dtype=np.int32)
        offs = np.array([c[2] for c in contigs], dtype=np.int64)
        pac = np.ascontiguousarray(pac, dtype=np.uint8)
        assert len(pac) >= l_pac // 4 + 1
        h = C.c_void_p()
        io = self.index_opts(**index_opts)
        self.check(self.L.lh_index_build_device(device, pac.ctypes.data_as(c_u8p), int(l_pac), n, offs.ctypes.data_as(c_i64p), lens.ctypes.data_as(c_i32p), names,
                                                C.byref(io), C.byref(h)))
        return Index(self, h)

    def synth_genome(self, l_pac, seed=20261002, gc=0.41, threads=0):
        """.pac image of an iid genome (host, threaded; reproducible from the seed)"""
        pac = np.zeros(l_pac // 4 + 1, dtype=np.uint8)
        self.check(self.L.lh_synth_genome(int(seed), float(gc), int(l_pac), pac.ctypes.data_as(c_u8p), int(threads)))
        return pac

    def synth_reads(self, pac, l_pac, contigs, seed, n_barcodes, pairs_per_barcode=100, len1=143, len2=150, sub_lo=0.001, sub_hi=0.01, indel_rate=0.0001,
                    junk_frac=0.0, mol_min=4, mol_max=10, threads=0):
        """barcode-sorted synthetic pairs (linked-read model) as a dict of the arrays an lh_batch wants + truth"""
        n = len(contigs)
        lens = np.array([c[1] for c in contigs], dtype=np.int32)
        offs = np.array([c[2] for c in contigs], dtype=np.int64)
        n_pairs = n_barcodes * pairs_per_barcode
        seq = np.zeros(2 * n_pairs * (max(len1, len2) + 3), dtype=np.uint8)
        seq_off = np.zeros(2 * n_pairs + 1, dtype=np.int64)
        bco = np.zeros(n_barcodes + 1, dtype=np.int32)
        ns = np.zeros(n_pairs, dtype=np.uint64)
        trid = np.zeros(n_pairs, dtype=np.int32)
        tp1 = np.zeros(n_pairs, dtype=np.int64)
        tp2 = np.zeros(n_pairs, dtype=np.int64)
        self.check(self.L.lh_synth_reads(pac.ctypes.data_as(c_u8p), int(l_pac), n, offs.ctypes.data_as(c_i64p), lens.ctypes.data_as(c_i32p), int(seed), int(n_barcodes),
                                         int(pairs_per_barcode), int(len1), int(len2), float(sub_lo), float(sub_hi), float(indel_rate), float(junk_frac), int(mol_min),
                                         int(mol_max), int(threads), seq.ctypes.data_as(c_u8p), seq_off.ctypes.data_as(c_i64p), bco.ctypes.data_as(c_i32p),
                                         ns.ctypes.data_as(c_u64p), trid.ctypes.data_as(c_i32p), tp1.ctypes.data_as(c_i64p), tp2.ctypes.data_as(c_i64p)))
        return dict(seq=seq[:seq_off[-1]], seq_off=seq_off, bc_pair_off=bco, name_seed=ns, truth_rid=trid, truth_pos1=tp1, truth_pos2=tp2, n_pairs=n_pairs)

    def write_fastq9(self, path, reads, first_barcode=0, trim=7, gz_level=1, seed=7):
        """`reads` (a dict from synth_reads) as a 9-line barcode-sorted FASTQ file — the reference's input format (gzip when gz_level > 0)"""
        self.L.lh_synth_write_fastq9.argtypes = [C.c_char_p, c_u8p, c_i64p, c_i32p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_uint64, c_i32p, c_i64p, c_i64p]
        seq, so, bco = (np.ascontiguousarray(reads["seq"], dtype=np.uint8), np.ascontiguousarray(reads["seq_off"], dtype=np.int64),
                        np.ascontiguousarray(reads["bc_pair_off"], dtype=np.int32))
        tr, t1, t2 = reads.get("truth_rid"), reads.get("truth_pos1"), reads.get("truth_pos2")
        self.check(self.L.lh_synth_write_fastq9(path.encode(), seq.ctypes.data_as(c_u8p), so.ctypes.data_as(c_i64p), bco.ctypes.data_as(c_i32p), len(bco) - 1, int(first_barcode),
                                                int(trim), int(gz_level), int(seed), tr.ctypes.data_as(c_i32p) if tr is not None else None,
                                                t1.ctypes.data_as(c_i64p) if t1 is not None else None, t2.ctypes.data_as(c_i64p) if t2 is not None else None))

    def index_from_arrays(self, arrs, device=0, **index_opts):
        """arrs: dict(primary, L2, bwt, sa, sa_intv, pac, l_pac, contigs=[(name,len,off)])"""
        n = len(arrs["contigs"])
        names = (C.c_char_p * n)(*[c[0].encode() for c in arrs["contigs"]])
        lens = np.array([c[1] for c in arrs["contigs"]], dtype=np.int32)
        offs = np.array([c[2] for c in arrs["contigs"]], dtype=np.int64)
        L2 = np.ascontiguousarray(arrs["L2"], dtype=np.uint64)
        bwt = np.ascontiguousarray(arrs["bwt"], dtype=np.uint32)
        sa = np.ascontiguousarray(arrs["sa"], dtype=np.uint64)
        pac = np.ascontiguousarray(arrs["pac"], dtype=np.uint8)
        h = C.c_void_p()
        io = self.index_opts(**index_opts)
        self.check(self.L.lh_index_from_arrays(device, int(arrs["primary"]), L2.ctypes.data_as(c_u64p), bwt.ctypes.data_as(c_u32p), len(bwt),
                                               int(arrs["sa_intv"]), sa.ctypes.data_as(c_u64p), len(sa), pac.ctypes.data_as(c_u8p), int(arrs["l_pac"]),
                                               n, offs.ctypes.data_as(c_i64p), lens.ctypes.data_as(c_i32p), names, C.byref(io), C.byref(h)))
        return Index(self, h)

    def ingest(self, path, trim=7, cap=0, chunk=0, max_pairs=1 << 20):
        """9-line FASTQ reader with the reference's work-unit rules (fastqreader/reader.go); host only"""
        return Ingest(self, path, trim, cap, chunk, max_pairs)

    def bam_writer(self, directory, contig_names, contig_lens, read_groups="", position_chunk_size=40000000, first_chunk=True, command_line="", threads=0):
        return BamWriter(self, directory, contig_names, contig_lens, read_groups, position_chunk_size, first_chunk, command_line, threads)

    def records_text(self, result, ingest_batch, contig_names, debug_tags=False):
        """BAM record content (bamwriter.go AppendBam) for one batch: `result` = Result of aligning `ingest_batch`; debug_tags = -debugBamTags"""
        rs = result.as_struct()
        names = (C.c_char_p * len(contig_names))(*[n.encode() for n in contig_names])
        txt, n = C.c_void_p(), C.c_int64()
        self.check(self.L.lh_records_text_ex(C.byref(rs), ingest_batch.ptr, len(contig_names), names, LH_REC_DEBUG_TAGS if debug_tags else 0, C.byref(txt), C.byref(n)))
        out = C.string_at(txt, n.value).decode()
        self.L.lh_records_free(txt)
        return out

    def bam_concat(self, shard_dirs, out_dir):
        n = len(shard_dirs)
        dirs = (C.c_char_p * n)(*[d.encode() for d in shard_dirs])
        self.L.lh_bam_concat.argtypes = [C.c_int32, C.POINTER(C.c_char_p), C.c_char_p]
        self.check(self.L.lh_bam_concat(n, dirs, out_dir.encode()))

    def name_seed(self, name):
        if isinstance(name, str):
            name = name.encode()
        return int(self.L.lh_name_seed(name, len(name)))

    def index_build(self, prefix, names, contigs_nt4, threads=0):
        n = len(names)
        seqs = [np.ascontiguousarray(s, dtype=np.uint8) for s in contigs_nt4]
        nm = (C.c_char_p * n)(*[x.encode() for x in names])
        ptrs = (c_u8p * n)(*[s.ctypes.data_as(c_u8p) for s in seqs])
        lens = np.array([len(s) for s in seqs], dtype=np.int64)
        rc = self.L.lh_index_build(prefix.encode(), n, nm, ptrs, lens.ctypes.data_as(c_i64p), threads)
        if rc:
            raise LhError(rc, "lh_index_build failed")


class Index:
    def __init__(self, lib, h):
        self.lib, self.h = lib, h

    def close(self):
        if self.h:
            self.lib.L.lh_index_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def l_pac(self):
        return self.lib.L.lh_index_l_pac(self.h)

    @property
    def sa_interval(self):
        """sampling interval of the suffix array resident in HBM (the .sa file's is 32; see lh_index_resample_sa)"""
        return int(self.lib.L.lh_index_sa_interval(self.h))

    def resample_sa(self, intv):
        self.lib.check(self.lib.L.lh_index_resample_sa(self.h, int(intv)))

    def contigs(self):
        n = C.c_int32()
        names = C.POINTER(C.c_char_p)()
        lens = c_i64p()
        offs = c_i64p()
        self.lib.check(self.lib.L.lh_index_contigs(self.h, C.byref(n), C.byref(names), C.byref(lens), C.byref(offs)))
        return [(names[i].decode(), int(lens[i]), int(offs[i])) for i in range(n.value)]

    def get_seq(self, rid, start, end, reversed_):
        buf = C.create_string_buffer(max(1, end - start))
        self.lib.check(self.lib.L.lh_get_seq(self.h, rid, start, end, int(reversed_), buf))
        return buf.raw[: end - start]

    def context(self, max_pairs, **context_opts):
        h = C.c_void_p()
        co = self.lib.context_opts(**context_opts)
        self.lib.check(self.lib.L.lh_context_create(self.h, int(max_pairs), C.byref(co), C.byref(h)))
        return Context(self, h)

    def export(self, sa_intv=32):
        """the index in the layout of bwa's files: dict(primary, L2, bwt, sa, sa_intv, pac-less) for oracle / file writers"""
        prim, nw, ns = C.c_uint64(), C.c_uint64(), C.c_uint64()
        L2 = np.zeros(5, dtype=np.uint64)
        L = self.lib.L
        self.lib.check(L.lh_index_export(self.h, C.byref(prim), L2.ctypes.data_as(c_u64p), None, C.byref(nw), sa_intv, None, C.byref(ns)))
        bwt = np.zeros(nw.value, dtype=np.uint32)
        sa = np.zeros(ns.value, dtype=np.uint64)
        self.lib.check(L.lh_index_export(self.h, C.byref(prim), L2.ctypes.data_as(c_u64p), bwt.ctypes.data_as(c_u32p), C.byref(nw), sa_intv, sa.ctypes.data_as(c_u64p),
                                         C.byref(ns)))
        return dict(primary=prim.value, L2=L2, bwt=bwt, sa=sa, sa_intv=sa_intv, l_pac=self.l_pac, contigs=self.contigs())

    def check(self, stride=64):
        """(rows checked, order violations, BWT/LF violations) — lh_diag_index_check"""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self.lib.L.lh_diag_index_check.argtypes = [C.c_void_p, C.c_uint64, c_u64p, c_u64p, c_u64p]
        self.lib.check(self.lib.L.lh_diag_index_check(self.h, int(stride), C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def digest(self):
        """(lcp checksum, k-mer tree checksum, tree levels) — lh_diag_index_digest"""
        a, b, l = C.c_uint64(), C.c_uint64(), C.c_int32()
        self.lib.L.lh_diag_index_digest.argtypes = [C.c_void_p, c_u64p, c_u64p, c_i32p]
        self.lib.check(self.lib.L.lh_diag_index_digest(self.h, C.byref(a), C.byref(b), C.byref(l)))
        return a.value, b.value, l.value

    def set_alt(self, flags):
        """bntann1_t.is_alt per contig (what <prefix>.alt sets when lh_index_load finds it)"""
        f = np.ascontiguousarray(flags, dtype=np.uint8)
        assert len(f) == len(self.contigs())
        self.lib.L.lh_index_set_alt.argtypes = [C.c_void_p, c_u8p]
        self.lib.check(self.lib.L.lh_index_set_alt(self.h, f.ctypes.data_as(c_u8p)))

    def alt(self):
        self.lib.L.lh_index_alt.argtypes = [C.c_void_p]
        self.lib.L.lh_index_alt.restype = c_u8p
        p = self.lib.L.lh_index_alt(self.h)
        n = len(self.contigs())
        return [0] * n if not p else [int(p[i]) for i in range(n)]

    def set_holes(self, holes):
        n = len(holes)
        ho = np.array([h[0] for h in holes], dtype=np.int64)
        hl = np.array([h[1] for h in holes], dtype=np.int32)
        hc = "".join(h[2] for h in holes).encode()
        self.lib.L.lh_index_set_holes.argtypes = [C.c_void_p, C.c_int32, c_i64p, c_i32p, C.c_char_p]
        self.lib.check(self.lib.L.lh_index_set_holes(self.h, n, ho.ctypes.data_as(c_i64p), hl.ctypes.data_as(c_i32p), hc))

    def save(self, prefix):
        self.lib.check(self.lib.L.lh_index_save(self.h, prefix.encode()))


class Context:
    def __init__(self, index, h):
        self.index, self.lib, self.h = index, index.lib, h

    def close(self):
        if self.h:
            self.lib.L.lh_context_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, batch):
        self._batch = batch
        self.lib.check(self.lib.L.lh_batch_upload(self.h, C.byref(batch.c)))

    def upload_slot(self, slot, batch):
        self.lib.check(self.lib.L.lh_batch_upload_slot(self.h, int(slot), C.byref(batch.c)))

    def select(self, slot):
        self.lib.check(self.lib.L.lh_batch_select(self.h, int(slot)))

    def align_resident(self, opts):
        self.lib.check(self.lib.L.lh_align_resident(self.h, C.byref(opts)))

    def download(self):
        res = C.POINTER(LhResult)()
        self.lib.check(self.lib.L.lh_result_download(self.h, C.byref(res)))
        out = Result(res.contents)
        self.lib.L.lh_result_free(res)
        return out

    def stage_slot(self, slot, batch):
        """lh_batch_stage_slot: upload into a slot that is not selected, on the upload stream (from a second host thread)"""
        self.lib.L.lh_batch_stage_slot.argtypes = [C.c_void_p, C.c_int32, C.POINTER(LhBatch)]
        self.lib.check(self.lib.L.lh_batch_stage_slot(self.h, int(slot), C.byref(batch.c)))

    def download_begin(self):
        self.lib.L.lh_result_download_begin.argtypes = [C.c_void_p]
        self.lib.check(self.lib.L.lh_result_download_begin(self.h))

    def download_end(self, raw=False):
        self.lib.L.lh_result_download_end.argtypes = [C.c_void_p, C.POINTER(C.POINTER(LhResult))]
        res = C.POINTER(LhResult)()
        self.lib.check(self.lib.L.lh_result_download_end(self.h, C.byref(res)))
        out = (int(res.contents.n_reads), int(res.contents.n_cand)) if raw else Result(res.contents)
        self.lib.L.lh_result_free(res)
        return out

    def download_raw(self):
        """lh_result_download + lh_result_free without the numpy copies of `Result`: what a cgo / C host pays, since it reads
        the library-owned SoA block in place.  Returns (n_reads, n_cand)."""
        res = C.POINTER(LhResult)()
        self.lib.check(self.lib.L.lh_result_download(self.h, C.byref(res)))
        n = (int(res.contents.n_reads), int(res.contents.n_cand))
        self.lib.L.lh_result_free(res)
        return n

    def align_barcodes(self, batch, opts=None, raw=False):
        opts = opts or self.lib.opts()
        self.upload(batch)
        self.align_resident(opts)
        return self.download_raw() if raw else self.download()

    def timings(self):
        n = C.c_int32()
        names = C.POINTER(C.c_char_p)()
        ms = C.POINTER(C.c_float)()
        self.lib.check(self.lib.L.lh_last_timings(self.h, C.byref(n), C.byref(names), C.byref(ms)))
        return [(names[i].decode(), float(ms[i])) for i in range(n.value)]

    def stage_dump(self, batch=None, opts=None):
        opts = opts or self.lib.opts()
        if batch is not None:
            self.upload(batch)
        d = C.POINTER(LhStageDump)()
        self.lib.check(self.lib.L.lh_stage_dump_resident(self.h, C.byref(opts), C.byref(d)))
        out = StageDump(d.contents)
        self.lib.L.lh_stage_dump_free(d)
        return out


_lib = None


def load_library(path=None):
    """The product library (singleton).  Raises if it is missing: no fallback."""
    global _lib
    if _lib is None or (path and _lib.path != path):
        _lib = Library(path)
    return _lib


class LhIngestBatch(C.Structure):
    _fields_ = ([("batch", LhBatch), ("n_sets", C.c_int32), ("set_complete", c_u8p)]
                + [(n + "_off", c_i64p) for n in ("name", "rgid", "qual1", "qual2", "trim", "bc", "rawbc", "bcqual", "si", "siqual")]
                + [(n, C.c_void_p) for n in ("name", "rgid", "qual1", "qual2", "trim_bases", "trim_quals", "bc", "rawbc", "bcqual", "si", "siqual")]
                + [("first_set_index", C.c_int64), ("at_eof", C.c_int32), ("arena_", C.c_void_p)])


class IngestBatch:
    """one lh_ingest_batch: `.c_batch` goes to Context.upload / align_barcodes; text columns are exposed as lists of bytes"""

    def __init__(self, lib, ptr, views_only=False):
        self.lib, self.ptr = lib, ptr
        b = ptr.contents
        self.c_batch = b.batch
        self.c = b.batch   # so that an IngestBatch can be handed to Context.upload / align_barcodes like a Batch
        self.n_pairs, self.n_sets = int(b.batch.n_pairs), int(b.n_sets)
        self.first_set_index, self.at_eof = int(b.first_set_index), bool(b.at_eof)
        if views_only:   # (a host that only passes the batch on: no numpy copies of its arrays)
            return
        self.bc_pair_off = _view(b.batch.bc_pair_off, self.n_sets + 1, np.int32).copy()
        self.bc_do_rfa = _view(b.batch.bc_do_rfa, self.n_sets, np.uint8).copy()
        self.set_complete = _view(b.set_complete, self.n_sets, np.uint8).copy()
        self.seq_off = _view(b.batch.seq_off, 2 * self.n_pairs + 1, np.int64).copy()
        self.seq = _view(b.batch.seq, int(self.seq_off[-1]), np.uint8).copy()
        self.name_seed = _view(b.batch.name_seed, self.n_pairs, np.uint64).copy()

    def column(self, name):
        b = self.ptr.contents
        offname = {"trim_bases": "trim_off", "trim_quals": "trim_off"}.get(name, name + "_off")
        off = _view(getattr(b, offname), self.n_pairs + 1, np.int64)
        base = getattr(b, name)
        raw = C.string_at(base, int(off[-1])) if self.n_pairs and off[-1] else b""
        return [raw[off[i]:off[i + 1]] for i in range(self.n_pairs)]

    def read(self, r):
        return self.seq[self.seq_off[r]:self.seq_off[r + 1]]

    def close(self):
        if self.ptr:
            self.lib.L.lh_ingest_batch_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BamWriter:
    """the reference's set of BAM files (bamwriter.go CreateBAMs): bc_sorted_bam.bam + position buckets + ZZZ_unmapped"""

    def __init__(self, lib, directory, contig_names, contig_lens, read_groups, position_chunk_size, first_chunk, command_line, threads):
        self.lib = lib
        self.h = C.c_void_p()
        names = (C.c_char_p * len(contig_names))(*[n.encode() for n in contig_names])
        lens = np.ascontiguousarray(contig_lens, dtype=np.int64)
        lib.check(lib.L.lh_bam_open(directory.encode(), len(contig_names), names, lens.ctypes.data_as(c_i64p), read_groups.encode(), int(position_chunk_size),
                                    int(bool(first_chunk)), command_line.encode(), int(threads), C.byref(self.h)))

    def set_debug_tags(self, on=True):
        self.lib.check(self.lib.L.lh_bam_set_flags(self.h, LH_REC_DEBUG_TAGS if on else 0))

    def append(self, result, ingest_batch):
        rs = result.as_struct()
        self.lib.check(self.lib.L.lh_bam_append(self.h, C.byref(rs), ingest_batch.ptr))

    def close(self):
        if self.h:
            h, self.h = self.h, None
            self.lib.check(self.lib.L.lh_bam_close(h))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Ingest:
    """9-line FASTQ reader (fastqreader/reader.go); iterate to get IngestBatch objects of whole barcode sets"""

    def __init__(self, lib, path, trim=7, cap=0, chunk=0, max_pairs=1 << 20):
        self.lib, self.max_pairs = lib, max_pairs
        self.h = C.c_void_p()
        lib.check(lib.L.lh_ingest_open(path.encode(), int(trim), int(cap), int(chunk), C.byref(self.h)))

    def next(self, max_pairs=None, views_only=False):
        p = C.POINTER(LhIngestBatch)()
        self.lib.check(self.lib.L.lh_ingest_next(self.h, int(max_pairs or self.max_pairs), C.byref(p)))
        return IngestBatch(self.lib, p, views_only)

    def __iter__(self):
        while True:
            b = self.next()
            if b.n_pairs == 0:
                return
            yield b
            if b.at_eof:
                return

    def close(self):
        if self.h:
            self.lib.L.lh_ingest_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


EXPORTED_SYMBOLS = [
    "lh_last_error", "lh_device_count", "lh_opts_init", "lh_index_load", "lh_index_from_arrays", "lh_index_contigs", "lh_index_l_pac",
    "lh_index_resample_sa", "lh_index_sa_interval",
    "lh_ingest_open", "lh_ingest_next", "lh_ingest_batch_free", "lh_ingest_close", "lh_name_seed",
    "lh_records_text", "lh_records_text_ex", "lh_records_free", "lh_bam_open", "lh_bam_append", "lh_bam_set_flags", "lh_bam_close",
    "lh_index_free", "lh_index_build", "lh_context_create", "lh_context_free", "lh_align_barcodes", "lh_batch_upload", "lh_align_resident",
    "lh_result_download", "lh_result_free", "lh_last_timings", "lh_stage_dump_resident", "lh_stage_dump_free", "lh_get_seq", "lh_device_memory", "lh_diag_gosort", "lh_diag_gosort_split", "lh_diag_bitonic", "lh_diag_introsort", "lh_diag_random_read", "lh_diag_valu_rate", "lh_diag_rescue_sw", "lh_diag_go_rand", "lh_diag_rescue_dedup",
    "lh_index_opts_init", "lh_context_opts_init", "lh_index_build_device", "lh_index_export", "lh_index_save", "lh_synth_genome", "lh_synth_reads", "lh_synth_write_fastq9", "lh_diag_index_check", "lh_batch_upload_slot", "lh_batch_select", "lh_bam_concat", "lh_reference_pack", "lh_index_set_holes", "lh_diag_index_digest", "lh_index_set_alt", "lh_index_alt", "lh_bam_set_level", "lh_bam_timings", "lh_result_download_begin", "lh_result_download_end", "lh_batch_stage_slot", "lh_host_alloc", "lh_host_free",
]

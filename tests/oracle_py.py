"""ctypes binding of oracle/_build/liboracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from lariat_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "_build", "liboracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


class Oracle:
    def __init__(self, lib):
        self.L = L = lib
        L.lo_last_error.restype = C.c_char_p
        L.lo_opts_init.argtypes = [C.POINTER(capi.LhOpts)]
        L.lo_index_load.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.lo_index_build_naive.argtypes = [C.c_int32, C.POINTER(C.c_char_p), C.POINTER(capi.c_u8p), capi.c_i64p, C.POINTER(C.c_void_p)]
        L.lo_index_from_arrays.argtypes = [C.c_uint64, capi.c_u64p, capi.c_u32p, C.c_uint64, C.c_int32, capi.c_u64p, C.c_uint64, capi.c_u8p, C.c_int64, C.c_int32,
                                           capi.c_i64p, capi.c_i32p, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p)]
        L.lo_index_free.argtypes = [C.c_void_p]
        L.lo_index_set_alt.argtypes = [C.c_void_p, capi.c_u8p]
        L.lo_index_contig_alt.argtypes = [C.c_void_p, C.c_int]
        L.lo_index_contig_alt.restype = C.c_int32
        L.lo_index_l_pac.argtypes = [C.c_void_p]
        L.lo_index_l_pac.restype = C.c_int64
        L.lo_index_n_contigs.argtypes = [C.c_void_p]
        L.lo_index_contig_name.argtypes = [C.c_void_p, C.c_int]
        L.lo_index_contig_name.restype = C.c_char_p
        L.lo_index_contig_len.argtypes = [C.c_void_p, C.c_int]
        L.lo_index_contig_len.restype = C.c_int64
        L.lo_index_contig_offset.argtypes = [C.c_void_p, C.c_int]
        L.lo_index_contig_offset.restype = C.c_int64
        L.lo_index_primary.argtypes = [C.c_void_p]
        L.lo_index_primary.restype = C.c_uint64
        L.lo_index_L2.argtypes = [C.c_void_p]
        L.lo_index_L2.restype = capi.c_u64p
        L.lo_index_bwt.argtypes = [C.c_void_p, capi.c_u64p]
        L.lo_index_bwt.restype = capi.c_u32p
        L.lo_index_sa.argtypes = [C.c_void_p, capi.c_u64p, capi.c_i32p]
        L.lo_index_sa.restype = capi.c_u64p
        L.lo_index_pac.argtypes = [C.c_void_p]
        L.lo_index_pac.restype = capi.c_u8p
        L.lo_index_image.argtypes = [C.c_void_p, C.c_int, capi.c_u8p]
        L.lo_index_image.restype = C.c_int64
        L.lo_get_seq.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_char_p]
        L.lo_align_barcodes.argtypes = [C.c_void_p, C.POINTER(capi.LhOpts), C.POINTER(capi.LhBatch), C.c_int32, C.POINTER(C.POINTER(capi.LhResult))]
        L.lo_result_free.argtypes = [C.POINTER(capi.LhResult)]
        L.lo_stage_dump.argtypes = [C.c_void_p, C.POINTER(capi.LhOpts), C.POINTER(capi.LhBatch), C.POINTER(C.POINTER(capi.LhStageDump))]
        L.lo_stage_dump_free.argtypes = [C.POINTER(capi.LhStageDump)]

    def opts(self, **kw):
        o = capi.LhOpts()
        self.L.lo_opts_init(C.byref(o))
        for k, v in kw.items():
            setattr(o, k, v)
        return o

    def index_load(self, prefix):
        h = C.c_void_p()
        rc = self.L.lo_index_load(prefix.encode(), C.byref(h))
        if rc:
            raise RuntimeError(self.L.lo_last_error().decode())
        return OracleIndex(self, h)

    def index_from_arrays(self, arrs, pac):
        """arrs: what capi.Index.export() returns; pac: the .pac image"""
        n = len(arrs["contigs"])
        names = (C.c_char_p * n)(*[c[0].encode() for c in arrs["contigs"]])
        lens = np.array([c[1] for c in arrs["contigs"]], dtype=np.int32)
        offs = np.array([c[2] for c in arrs["contigs"]], dtype=np.int64)
        L2 = np.ascontiguousarray(arrs["L2"], dtype=np.uint64)
        bwt = np.ascontiguousarray(arrs["bwt"], dtype=np.uint32)
        sa = np.ascontiguousarray(arrs["sa"], dtype=np.uint64)
        pac = np.ascontiguousarray(pac, dtype=np.uint8)
        h = C.c_void_p()
        self.L.lo_index_from_arrays(int(arrs["primary"]), L2.ctypes.data_as(capi.c_u64p), bwt.ctypes.data_as(capi.c_u32p), len(bwt), int(arrs["sa_intv"]),
                                    sa.ctypes.data_as(capi.c_u64p), len(sa), pac.ctypes.data_as(capi.c_u8p), int(arrs["l_pac"]), n,
                                    offs.ctypes.data_as(capi.c_i64p), lens.ctypes.data_as(capi.c_i32p), names, C.byref(h))
        return OracleIndex(self, h)

    def index_build_naive(self, names, seqs_nt4):
        n = len(names)
        seqs = [np.ascontiguousarray(s, dtype=np.uint8) for s in seqs_nt4]
        nm = (C.c_char_p * n)(*[x.encode() for x in names])
        ptrs = (capi.c_u8p * n)(*[s.ctypes.data_as(capi.c_u8p) for s in seqs])
        lens = np.array([len(s) for s in seqs], dtype=np.int64)
        h = C.c_void_p()
        self.L.lo_index_build_naive(n, nm, ptrs, lens.ctypes.data_as(capi.c_i64p), C.byref(h))
        return OracleIndex(self, h)


class OracleIndex:
    def __init__(self, o, h):
        self.o, self.h = o, h

    def __del__(self):
        try:
            self.o.L.lo_index_free(self.h)
        except Exception:
            pass

    @property
    def l_pac(self):
        return self.o.L.lo_index_l_pac(self.h)

    def set_alt(self, flags):
        f = np.ascontiguousarray(flags, dtype=np.uint8)
        assert len(f) == self.o.L.lo_index_n_contigs(self.h)
        self.o.L.lo_index_set_alt(self.h, f.ctypes.data_as(capi.c_u8p))

    def alt(self):
        return [int(self.o.L.lo_index_contig_alt(self.h, i)) for i in range(self.o.L.lo_index_n_contigs(self.h))]

    def contigs(self):
        L = self.o.L
        n = L.lo_index_n_contigs(self.h)
        return [(L.lo_index_contig_name(self.h, i).decode(), L.lo_index_contig_len(self.h, i), L.lo_index_contig_offset(self.h, i)) for i in range(n)]

    def image(self, which):
        n = self.o.L.lo_index_image(self.h, which, None)
        buf = np.zeros(n, dtype=np.uint8)
        self.o.L.lo_index_image(self.h, which, buf.ctypes.data_as(capi.c_u8p))
        return buf.tobytes()

    def arrays(self):
        """raw arrays for lh_index_from_arrays"""
        L = self.o.L
        nw = C.c_uint64()
        bwt = np.ctypeslib.as_array(L.lo_index_bwt(self.h, C.byref(nw)), shape=(nw.value,)).copy()
        nsa = C.c_uint64()
        intv = C.c_int32()
        sap = L.lo_index_sa(self.h, C.byref(nsa), C.byref(intv))
        sa = np.ctypeslib.as_array(sap, shape=(nsa.value,)).copy()
        l_pac = self.l_pac
        pac = np.ctypeslib.as_array(L.lo_index_pac(self.h), shape=(l_pac // 4 + 1,)).copy()
        L2 = np.ctypeslib.as_array(L.lo_index_L2(self.h), shape=(5,)).copy()
        return dict(primary=L.lo_index_primary(self.h), L2=L2, bwt=bwt, sa=sa, sa_intv=intv.value, pac=pac, l_pac=l_pac, contigs=self.contigs())

    def get_seq(self, rid, start, end, reversed_):
        buf = C.create_string_buffer(max(1, end - start))
        self.o.L.lo_get_seq(self.h, rid, start, end, int(reversed_), buf)
        return buf.raw[: end - start]

    def align_barcodes(self, batch, opts=None, threads=1):
        opts = opts or self.o.opts()
        res = C.POINTER(capi.LhResult)()
        rc = self.o.L.lo_align_barcodes(self.h, C.byref(opts), C.byref(batch.c), threads, C.byref(res))
        if rc:
            raise RuntimeError(self.o.L.lo_last_error().decode())
        out = capi.Result(res.contents)
        # MapQData of every candidate as the oracle's molecules left it (what -debugBamTags prints): [n_cand, 7] ints + a confidence
        mi, mc = C.POINTER(C.c_int32)(), C.POINTER(C.c_double)()
        self.o.L.lo_result_mapq_data.argtypes = [C.POINTER(capi.LhResult), C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.POINTER(C.c_double))]
        if self.o.L.lo_result_mapq_data(res, C.byref(mi), C.byref(mc)) == 0 and out.n_cand:
            out.md_int = np.ctypeslib.as_array(mi, shape=(out.n_cand, 7)).copy()
            out.md_sb_conf = np.ctypeslib.as_array(mc, shape=(out.n_cand,)).copy()
        self.o.L.lo_result_free(res)
        return out

    def time_align(self, batch, opts=None, threads=1):
        """run without materialising a result (cpu_baseline leg)"""
        opts = opts or self.o.opts()
        return self.o.L.lo_align_barcodes(self.h, C.byref(opts), C.byref(batch.c), threads, None)

    def stage_dump(self, batch, opts=None):
        opts = opts or self.o.opts()
        d = C.POINTER(capi.LhStageDump)()
        self.o.L.lo_stage_dump(self.h, C.byref(opts), C.byref(batch.c), C.byref(d))
        out = capi.StageDump(d.contents)
        self.o.L.lo_stage_dump_free(d)
        return out


_inst = None


def load():
    global _inst
    if _inst is None:
        build()
        _inst = Oracle(C.CDLL(LIB))
    return _inst

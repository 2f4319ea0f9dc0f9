"""ctypes mirror of include/d2g.h.  No computation happens here: every method forwards to the
C ABI of libd2g.so (HIP kernels + x86 host half).  Missing library => ImportError-like failure,
never a silent fallback."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("D2G_LIB") or os.path.join(_HERE, "libd2g.so")
_lib = None

SIMILARITY, CONTAINMENT, SYMMETRIC_CONTAINMENT, POISSON_LLR, INTERSECTION, UNION_SIZE = range(6)
CMP_AUTO, CMP_DIRECT, CMP_BITSLICE = 0, 1, 2
# bit-sliced pair kernel: VALU operations per pair and 32-register group = (id planes of the group) + this (one v_bcnt)
BITSLICE_OPS_PER_GROUP_EXTRA = 1


TIME_K1, TIME_K2, TIME_K2PREP, TIME_K3, TIME_K0 = 2, 4, 8, 16, 32          # include/d2g.h D2G_TIME_*


class D2GError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        name = lib().d2g_strerror(status).decode() if _lib is not None else str(status)
        super().__init__(f"d2g error {status} ({name}){': ' + detail if detail else ''}")


def build(jobs=8):
    """Compile libd2g.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-s", f"-j{jobs}", "-C", os.path.join(_HERE, "csrc")])
    return LIB_PATH


_u64, _sz, _dbl, _int, _f32 = C.c_uint64, C.c_size_t, C.c_double, C.c_int, C.c_float
_vp, _cp = C.c_void_p, C.c_char_p
_pu64, _pu32, _pdbl, _pf32, _pu8 = (C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_double),
                                    C.POINTER(C.c_float), C.POINTER(C.c_uint8))

# name: (restype, argtypes) -- must cover every d2g_* declared in include/d2g.h (tested)
SIGNATURES = {
    "d2g_version": (_int, []),
    "d2g_strerror": (_cp, [_int]),
    "d2g_device_count": (_int, []),
    "d2g_ctx_create": (_int, [_int, C.POINTER(_vp)]),
    "d2g_ctx_destroy": (None, [_vp]),
    "d2g_last_error": (_cp, [_vp]),
    "d2g_ctx_device": (_int, [_vp]),
    "d2g_ctx_reload_tuning": (_int, [_vp]),
    "d2g_ctx_tuning": (_int, [_vp, C.c_char_p, _sz]),
    "d2g_sync": (_int, [_vp, _vp]),
    "d2g_malloc": (_int, [_vp, _sz, C.POINTER(_vp)]),
    "d2g_free": (_int, [_vp, _vp]),
    "d2g_malloc_host": (_int, [_vp, _sz, C.POINTER(_vp)]),
    "d2g_host_register": (_int, [_vp, _vp, _sz]),
    "d2g_host_unregister": (_int, [_vp, _vp]),
    "d2g_free_host": (_int, [_vp, _vp]),
    "d2g_memcpy_h2d": (_int, [_vp, _vp, _vp, _sz, _vp]),
    "d2g_memcpy_d2h": (_int, [_vp, _vp, _vp, _sz, _vp]),
    "d2g_set_timing": (_int, [_vp, _int]),
    "d2g_kernel_ms": (_int, [_vp, _cp, _int, C.POINTER(_int), C.POINTER(_f32), C.POINTER(_f32)]),
    "d2g_wang_hash": (_u64, [_u64]),
    "d2g_seed_mask": (_u64, [_u64]),
    "d2g_oph_xor_const": (_u64, []),
    "d2g_oph_m": (_sz, [_sz]),
    "d2g_oph_card": (_dbl, [_pu64, _sz]),
    "d2g_oph_signatures": (_int, [_pu64, _sz, _pdbl]),
    "d2g_oph_finalize": (_int, [_pu64, _sz, _sz, _sz, _pdbl, _pdbl, _int]),
    "d2g_densify": (_int, [_pdbl, _sz, _sz, C.POINTER(_sz), _int]),
    "d2g_epilogue_gtlt": (_f32, [_u64, _u64, _sz, _dbl, _dbl, _int, _int]),
    "d2g_epilogue_neq": (_f32, [_u64, _sz, _dbl, _dbl, _int, _int]),
    "d2g_epilogue_ut": (_int, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _int, _int, _int, _int, _vp]),
    "d2g_epilogue_lut": (_int, [_sz, _int, _int, _int, _pf32]),
    "d2g_seqpack_create": (_int, [_int, C.POINTER(_vp)]),
    "d2g_seqpack_destroy": (None, [_vp]),
    "d2g_seqpack_clear": (None, [_vp]),
    "d2g_seqpack_add_path": (_int, [_vp, _cp]),
    "d2g_seqpack_add_fastx": (_int, [_vp, _cp, _sz]),
    "d2g_seqpack_add_sequence": (_int, [_vp, _cp, _sz]),
    "d2g_seqpack_ngenomes": (_sz, [_vp]),
    "d2g_seqpack_nruns": (_sz, [_vp]),
    "d2g_seqpack_packed_bytes": (_sz, [_vp]),
    "d2g_seqpack_packed": (_pu8, [_vp]),
    "d2g_seqpack_run_start": (_pu64, [_vp]),
    "d2g_seqpack_run_len": (_pu32, [_vp]),
    "d2g_seqpack_genome_run_off": (_pu64, [_vp]),
    "d2g_seqpack_nkmers": (_u64, [_vp, _sz]),
    "d2g_seqpack_nbases": (_u64, [_vp]),
    "d2g_oph_sketch": (_int, [_vp, _vp, _sz, _vp, _vp, _sz, _vp, _sz, _int, _int, _u64, _sz, _vp]),
    "d2g_oph_plan_create": (_int, [_vp, _vp, _vp, _sz, _vp, _sz, _int, C.POINTER(_vp)]),
    "d2g_oph_plan_destroy": (None, [_vp]),
    "d2g_oph_plan_nkmers": (_u64, [_vp]),
    "d2g_oph_plan_nbases": (_u64, [_vp]),
    "d2g_oph_sketch_dev": (_int, [_vp, _vp, _vp, _int, _u64, _sz, _vp, _vp]),
    "d2g_sketcher_create": (_int, [_vp, C.POINTER(_vp)]),
    "d2g_sketcher_destroy": (None, [_vp]),
    "d2g_sketcher_run": (_int, [_vp, _vp, _sz, _vp, _vp, _sz, _vp, _sz, _int, _int, _u64, _sz, _vp]),
    "d2g_sketcher_run_bmh": (_int, [_vp, _vp, _sz, _vp, _vp, _sz, _vp, _sz, _int, _int, _u64, _sz, C.c_double, _vp, _vp]),
    "d2g_bmh_sketch_dev": (_int, [_vp, _vp, _vp, _int, _u64, _sz, C.c_double, _vp, _vp, _vp]),
    "d2g_bmh_sketch": (_int, [_vp, _vp, _sz, _vp, _vp, _sz, _vp, _sz, _int, _int, _u64, _sz, C.c_double, _vp, _vp]),
    "d2g_kmer_count": (_int, [_vp, _vp, _sz, _vp, _vp, _sz, _vp, _sz, _int, _int, _u64, C.c_double, _vp, _vp, _sz, _vp]),
    "d2g_kmer_distinct": (_int, [_vp, _vp, _sz, _vp, _vp, _sz, _vp, _sz, _int, _int, _u64, _vp]),
    "d2g_sketcher_run_distinct": (_int, [_vp, _vp, _sz, _vp, _vp, _sz, _vp, _sz, _int, _int, _u64, _vp]),
    "d2g_sketcher_ingest_fasta": (_int, [_vp, _vp, _sz, _vp, _vp, _sz, _vp, _sz, _int]),
    "d2g_sketcher_ingested_runs": (_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_sz), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_u64)]),
    "d2g_seqpack_add_path_by_record": (_int, [_vp, C.c_char_p]),
    "d2g_seqpack_add_fastx_by_record": (_int, [_vp, C.c_char_p, _sz]),
    "d2g_seqpack_name": (C.c_char_p, [_vp, _sz]),
    "d2g_bmh_from_weighted": (_int, [_vp, _vp, _vp, _vp, _sz, _sz, _vp, _vp]),
    "d2g_bmh_from_weighted_ids": (_int, [_vp, _vp, _vp, _vp, _sz, _sz, _vp, _vp, _vp]),
    "d2g_ut_count": (_sz, [_sz, _sz, _sz]),
    "d2g_cmp_set_create_dev": (_int, [_vp, _vp, _sz, _sz, _int, _vp, C.POINTER(_vp)]),
    "d2g_cmp_set_create": (_int, [_vp, _vp, _sz, _sz, _int, C.POINTER(_vp)]),
    "d2g_cmp_set_update_dev": (_int, [_vp, _vp, _vp, _vp]),
    "d2g_cmp_set_planes": (_int, [_vp, _vp, _vp, C.POINTER(C.c_uint), C.POINTER(_int), C.POINTER(_f32)]),
    "d2g_cmp_set_status": (_int, [_vp, _vp, _vp]),
    "d2g_cmp_set_sparse_info": (_int, [_vp, _vp, _vp, _vp]),
    "d2g_cmp_set_debug_pairs": (_int, [_vp, _vp, _vp, _vp, _sz, C.POINTER(_sz), _vp]),
    "d2g_warmup": (_int, [_vp, _int]),
    "d2g_device_name": (_int, [_int, C.c_char_p, _sz]),
    "d2g_comm_unique_id": (_int, [_vp]),
    "d2g_comm_create": (_int, [_vp, _vp, _int, _int, C.POINTER(_vp)]),
    "d2g_comm_create_all": (_int, [C.POINTER(_vp), _int, C.POINTER(_vp)]),
    "d2g_comm_destroy": (None, [_vp]),
    "d2g_comm_rank": (_int, [_vp]),
    "d2g_comm_world": (_int, [_vp]),
    "d2g_comm_is_rccl": (_int, [_vp]),
    "d2g_bcast_sigs": (_int, [C.POINTER(_vp), C.POINTER(_vp), _int, _vp, _sz, _sz, C.POINTER(_vp)]),
    "d2g_allpairs_create": (_int, [_vp, _vp, _sz, _sz, C.POINTER(_vp)]),
    "d2g_allpairs_destroy": (None, [_vp]),
    "d2g_allpairs_rows_held": (_int, [_vp, C.POINTER(_sz), C.POINTER(_sz)]),
    "d2g_allpairs_rows_computed": (_int, [_vp, C.POINTER(_sz), C.POINTER(_sz)]),
    "d2g_allpairs_prepare_dev": (_int, [_vp, _vp, _vp]),
    "d2g_allpairs_prepare_all": (_int, [C.POINTER(_vp), _int, C.POINTER(_vp), C.POINTER(_vp)]),
    "d2g_allpairs_operand": (_vp, [_vp]),
    "d2g_allpairs_status": (_int, [_vp, _vp]),
    "d2g_allpairs_chunks": (_int, [_vp]),
    "d2g_allpairs_set_phase_timing": (_int, [_vp, _int]),
    "d2g_allpairs_phase_times": (_int, [_vp, _int, C.POINTER(_int), _vp, _vp, _vp, _vp]),
    "d2g_allpairs_sparse_info": (_int, [_vp, _vp]),
    "d2g_allpairs_step_lut_dev": (_int, [_vp, _vp, _vp, _vp, _vp]),
    "d2g_allpairs_step_eqcount_dev": (_int, [_vp, _vp, _vp, _vp]),
    "d2g_allpairs_step_all": (_int, [C.POINTER(_vp), _int, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "d2g_allpairs_enqueue_lut_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _int]),
    "d2g_operand_layout": (_int, [_sz, _sz, C.POINTER(_sz), C.POINTER(_sz)]),
    "d2g_cmp_set_export_operand_dev": (_int, [_vp, _vp, _vp, _vp, _vp]),
    "d2g_cmp_set_from_planes_dev": (_int, [_vp, _sz, _sz, _vp, _vp, C.POINTER(_vp)]),
    "d2g_pack_column_slices_dev": (_int, [_vp, _vp, _sz, _sz, _int, _vp, _vp]),
    "d2g_cmp_set_destroy": (None, [_vp]),
    "d2g_cmp_set_algo": (_int, [_vp]),
    "d2g_cmp_eqcount_ut_dev": (_int, [_vp, _vp, _sz, _sz, _vp, _vp]),
    "d2g_cmp_lut_ut_dev": (_int, [_vp, _vp, _sz, _sz, _vp, _vp, _vp]),
    "d2g_cmp_ut_prefill_dev": (_int, [_vp, _vp, _sz, _sz, _vp, _vp, _vp, _vp]),
    "d2g_cmp_ut_announce_dev": (_int, [_vp, _vp, _sz, _sz, _vp, _vp, _vp]),
    "d2g_cmp_set_forget": (_int, [_vp, _vp]),
    "d2g_cmp_gtlt_ut_dev": (_int, [_vp, _vp, _sz, _sz, _vp, _vp, _vp]),
    "d2g_cmp_eqcount_rect_dev": (_int, [_vp, _vp, _sz, _sz, _sz, _sz, _vp, _vp]),
    "d2g_cmp_gtlt_rect_dev": (_int, [_vp, _vp, _sz, _sz, _sz, _sz, _vp, _vp, _vp]),
    "d2g_cmp_eqcount_ut": (_int, [_vp, _vp, _sz, _sz, _sz, _sz, _int, _vp]),
    "d2g_cmp_dist_ut": (_int, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _int, _int, _int, _int, _int, _vp]),
    "d2g_ut_partition": (_int, [_sz, _int, C.POINTER(_sz)]),
}


def sparse_info_dict(a):
    """info4 of d2g_cmp_set_sparse_info / d2g_allpairs_sparse_info as a dict"""
    return {"sorted_operand": bool(a[0]), "tiles_listed": int(a[1]), "dense_decided_by_prepare": bool(a[2] & 1), "dense_kernel_ran": bool(a[2] & 2),
            "tiles_and_pair_list": bool(a[2] & 4), "callers_order_kept": bool(a[2] & 8), "ordering_skipped": bool(a[2] & 16), "pairs_listed": int(a[3])}


def lib():
    """Load libd2g.so (once).  Raises if the extension has not been built: no fallback."""
    global _lib
    if _lib is None:
        # torch wheels bundle their own HIP/HSA runtime; two runtimes in one process cannot both own
        # the GPU.  If torch is going to be used next to libd2g (bench, dist), load it FIRST so that
        # libd2g's libamdhip64.so.7 dependency resolves to the copy that is already mapped.
        if os.environ.get("D2G_NO_TORCH_PRELOAD") != "1":
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        if not os.path.exists(LIB_PATH):
            raise OSError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(dashing2_amd has no CPU fallback)")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(l, name)
            f.restype = res
            f.argtypes = args
        _lib = l
    return _lib


def _np_ptr(a):
    return a.ctypes.data if a is not None else None


# ---------------------------------------------------------------- host-side primitives
def wang_hash(x):
    return int(lib().d2g_wang_hash(x & 0xFFFFFFFFFFFFFFFF))


def seed_mask(seed):
    return int(lib().d2g_seed_mask(seed))


def oph_xor_const():
    return int(lib().d2g_oph_xor_const())


def oph_m(S):
    return int(lib().d2g_oph_m(S))


def oph_finalize(regs, S, nthreads=1):
    """regs u64 [n][m] -> (sigs f64 [n][S], cards f64 [n])  (x87 host arithmetic in libd2g)"""
    regs = np.ascontiguousarray(regs, np.uint64)
    n, m = regs.shape
    sigs = np.empty((n, S), np.float64)
    cards = np.empty(n, np.float64)
    rc = lib().d2g_oph_finalize(regs.ctypes.data_as(_pu64), n, m, S, sigs.ctypes.data_as(_pdbl),
                                cards.ctypes.data_as(_pdbl), nthreads)
    if rc:
        raise D2GError(rc)
    return sigs, cards


def densify(sigs, nthreads=1):
    sigs = np.ascontiguousarray(sigs, np.float64).copy()
    n, S = sigs.shape
    nf = _sz()
    rc = lib().d2g_densify(sigs.ctypes.data_as(_pdbl), n, S, C.byref(nf), nthreads)
    if rc:
        raise D2GError(rc)
    return sigs, nf.value


def epilogue_lut(S, measure=SIMILARITY, k=31, multiset_space=False):
    lut = np.empty(S + 1, np.float32)
    rc = lib().d2g_epilogue_lut(S, measure, k, int(multiset_space), lut.ctypes.data_as(_pf32))
    if rc:
        raise D2GError(rc)
    return lut


def host_epilogue_ut(ca, cb, cards, N, S, r0, r1, measure=SIMILARITY, k=31, multiset_space=False, nthreads=0):
    """integer counts of rows [r0,r1) -> float32 values (libd2g x87 host arithmetic, OpenMP)"""
    ca = np.ascontiguousarray(ca, np.uint32)
    cb = None if cb is None else np.ascontiguousarray(cb, np.uint32)
    cards = np.ascontiguousarray(cards, np.float64)
    out = np.empty(ca.size, np.float32)
    rc = lib().d2g_epilogue_ut(_np_ptr(ca), _np_ptr(cb), _np_ptr(cards), N, S, r0, r1, measure, k, int(multiset_space),
                               nthreads or (os.cpu_count() or 1), _np_ptr(out))
    if rc:
        raise D2GError(rc)
    return out


def epilogue_gtlt(gt, lt, S, lhc, rhc, measure=SIMILARITY, k=31):
    return float(lib().d2g_epilogue_gtlt(gt, lt, S, lhc, rhc, measure, k))


def epilogue_neq(neq, S, lhc, rhc, measure=SIMILARITY, k=31):
    return float(lib().d2g_epilogue_neq(neq, S, lhc, rhc, measure, k))


def operand_layout(N, S):
    """-> (u32 words per 32-register group, number of groups) of the bit-sliced operand"""
    gw, ng = _sz(), _sz()
    rc = lib().d2g_operand_layout(N, S, C.byref(gw), C.byref(ng))
    if rc:
        raise D2GError(rc)
    return int(gw.value), int(ng.value)


def ut_count(N, r0=0, r1=None):
    return int(lib().d2g_ut_count(N, r0, N if r1 is None else r1))


def ut_partition(N, nparts):
    b = (_sz * (nparts + 1))()
    rc = lib().d2g_ut_partition(N, nparts, b)
    if rc:
        raise D2GError(rc)
    return [int(x) for x in b]


# ---------------------------------------------------------------- ingest
class SeqPack:
    """FASTA/FASTQ -> packed run stream (d2g_seqpack_*)."""

    def __init__(self, k):
        self.k = k
        self._h = None
        h = _vp()
        rc = lib().d2g_seqpack_create(k, C.byref(h))
        if rc:
            raise D2GError(rc)
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().d2g_seqpack_destroy(self._h)
            self._h = None

    __del__ = close

    def clear(self):
        """forget the content, keep the allocations"""
        lib().d2g_seqpack_clear(self._h)

    def add_path(self, path):
        rc = lib().d2g_seqpack_add_path(self._h, os.fsencode(path))
        if rc:
            raise D2GError(rc, str(path))

    def add_path_by_record(self, path):
        rc = lib().d2g_seqpack_add_path_by_record(self._h, os.fsencode(path))
        if rc:
            raise D2GError(rc, str(path))

    def add_fastx_by_record(self, buf):
        rc = lib().d2g_seqpack_add_fastx_by_record(self._h, buf, len(buf))
        if rc:
            raise D2GError(rc, "add_fastx_by_record")

    def name(self, g):
        return lib().d2g_seqpack_name(self._h, g).decode()

    def add_fastx(self, data: bytes):
        rc = lib().d2g_seqpack_add_fastx(self._h, data, len(data))
        if rc:
            raise D2GError(rc)

    def add_sequence(self, seq: bytes):
        rc = lib().d2g_seqpack_add_sequence(self._h, seq, len(seq))
        if rc:
            raise D2GError(rc)

    @property
    def ngenomes(self):
        return int(lib().d2g_seqpack_ngenomes(self._h))

    @property
    def nruns(self):
        return int(lib().d2g_seqpack_nruns(self._h))

    @property
    def nbases(self):
        return int(lib().d2g_seqpack_nbases(self._h))

    def nkmers(self, g):
        return int(lib().d2g_seqpack_nkmers(self._h, g))

    def arrays(self):
        """-> (packed u8, run_start u64, run_len u32, genome_run_off u64) as numpy copies"""
        l = lib()
        nb = l.d2g_seqpack_packed_bytes(self._h)
        nr, ng = self.nruns, self.ngenomes
        packed = np.ctypeslib.as_array(l.d2g_seqpack_packed(self._h), shape=(nb,)).copy()
        rs = np.ctypeslib.as_array(l.d2g_seqpack_run_start(self._h), shape=(nr,)).copy() if nr else np.empty(0, np.uint64)
        rl = np.ctypeslib.as_array(l.d2g_seqpack_run_len(self._h), shape=(nr,)).copy() if nr else np.empty(0, np.uint32)
        go = np.ctypeslib.as_array(l.d2g_seqpack_genome_run_off(self._h), shape=(ng + 1,)).copy()
        return packed, rs, rl, go


# ---------------------------------------------------------------- device context
class Context:
    """One GPU (d2g_ctx).  Raises D2GError when no gfx950 device is usable."""

    def __init__(self, device=0):
        self._h = None
        h = _vp()
        rc = lib().d2g_ctx_create(device, C.byref(h))
        if rc:
            raise D2GError(rc, f"d2g_ctx_create(device={device})")
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            lib().d2g_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        if rc:
            raise D2GError(rc, lib().d2g_last_error(self._h).decode())

    def sync(self, stream=None):
        self._check(lib().d2g_sync(self._h, stream))

    def reload_tuning(self):
        """the D2G_* switches are read once, at context creation: read them again (tests that change one between calls)"""
        self._check(lib().d2g_ctx_reload_tuning(self._h))

    def tuning(self):
        """-> dict of the D2G_* switches this context resolved (only those that were set)"""
        import json
        n = lib().d2g_ctx_tuning(self._h, None, 0)
        if n < 0:
            self._check(n)
        buf = C.create_string_buffer(n + 1)
        lib().d2g_ctx_tuning(self._h, buf, n + 1)
        return json.loads(buf.value.decode())

    def set_timing(self, on=True):
        """True / False, or an OR of TIME_K1 / TIME_K2 / TIME_K2PREP / TIME_K3 (time only what is reported: an event pair in
        the stream costs a few microseconds of device time per launch)"""
        self._check(lib().d2g_set_timing(self._h, int(on)))

    def kernel_ms(self, which, reset=True):
        """-> (count, avg_ms, last_ms) of the launches logged since the last reset"""
        n, avg, last = _int(), _f32(), _f32()
        self._check(lib().d2g_kernel_ms(self._h, which.encode(), int(reset), C.byref(n), C.byref(avg), C.byref(last)))
        return int(n.value), float(avg.value), float(last.value)

    # -- K1 ------------------------------------------------------------------
    def oph_sketch(self, packed, run_start, run_len, genome_run_off, k, S, canon=True, xormask=0):
        """host arrays -> regs u64 [n][m]"""
        packed = np.ascontiguousarray(packed, np.uint8)
        run_start = np.ascontiguousarray(run_start, np.uint64)
        run_len = np.ascontiguousarray(run_len, np.uint32)
        genome_run_off = np.ascontiguousarray(genome_run_off, np.uint64)
        n = genome_run_off.size - 1
        m = oph_m(S)
        regs = np.empty((n, m), np.uint64)
        self._check(lib().d2g_oph_sketch(self._h, _np_ptr(packed), packed.size, _np_ptr(run_start), _np_ptr(run_len),
                                         run_start.size, _np_ptr(genome_run_off), n, k, int(canon), xormask, S,
                                         _np_ptr(regs)))
        return regs

    def oph_sketch_seqpack(self, sp: SeqPack, S, canon=True, xormask=0):
        packed, rs, rl, go = sp.arrays()
        return self.oph_sketch(packed, rs, rl, go, sp.k, S, canon, xormask)

    def sketcher(self):
        return Sketcher(self)

    # -- K3 (--multiset: exact k-mer counts -> BagMinHash) ---------------------
    def bmh_sketch_seqpack(self, sp: "SeqPack", S, canon=True, xormask=0, count_threshold=0.0):
        """-> (sig float64[n][S], total_weight float64[n])"""
        packed, rs, rl, go = sp.arrays()
        n = go.size - 1
        sig = np.empty((n, S), np.float64)
        tw = np.empty(n, np.float64)
        self._check(lib().d2g_bmh_sketch(self._h, _np_ptr(packed), packed.size, _np_ptr(rs), _np_ptr(rl), rs.size,
                                         _np_ptr(go), n, sp.k, int(canon), xormask, S, count_threshold,
                                         _np_ptr(sig), _np_ptr(tw)))
        return sig, tw

    def bmh_sketch_dev(self, plan, packed_dev_ptr, S, sig_dev_ptr, tw_dev_ptr, canon=True, xormask=0, count_threshold=0.0,
                       stream=None):
        self._check(lib().d2g_bmh_sketch_dev(self._h, plan._h, packed_dev_ptr, int(canon), xormask, S, count_threshold,
                                             sig_dev_ptr, tw_dev_ptr, stream))

    def kmer_count_seqpack(self, sp: "SeqPack", canon=True, xormask=0, count_threshold=0.0):
        """-> list over genomes of (keys uint64[nd], counts uint32[nd]), each sorted by key"""
        packed, rs, rl, go = sp.arrays()
        n = go.size - 1
        cap = int(sum(sp.nkmers(g) for g in range(n)))
        keys = np.empty(max(cap, 1), np.uint64)
        counts = np.empty(max(cap, 1), np.uint32)
        off = np.zeros(n + 1, np.uint64)
        self._check(lib().d2g_kmer_count(self._h, _np_ptr(packed), packed.size, _np_ptr(rs), _np_ptr(rl), rs.size,
                                         _np_ptr(go), n, sp.k, int(canon), xormask, count_threshold,
                                         _np_ptr(keys), _np_ptr(counts), cap, _np_ptr(off)))
        out = []
        for g in range(n):
            k_, c_ = keys[int(off[g]):int(off[g + 1])], counts[int(off[g]):int(off[g + 1])]
            o = np.argsort(k_, kind="stable")
            out.append((k_[o].copy(), c_[o].copy()))
        return out

    def kmer_distinct_seqpack(self, sp: "SeqPack", canon=True, xormask=0):
        """-> uint64[n]: exact number of distinct masked k-mers per genome"""
        packed, rs, rl, go = sp.arrays()
        n = go.size - 1
        out = np.zeros(n, np.uint64)
        self._check(lib().d2g_kmer_distinct(self._h, _np_ptr(packed), packed.size, _np_ptr(rs), _np_ptr(rl), rs.size,
                                            _np_ptr(go), n, sp.k, int(canon), xormask, _np_ptr(out)))
        return out

    def bmh_from_weighted(self, ids, weights, set_off, S):
        """-> (sig float64[nsets][S], total_weight float64[nsets])"""
        ids = np.ascontiguousarray(ids, np.uint64)
        w = None if weights is None else np.ascontiguousarray(weights, np.float64)
        set_off = np.ascontiguousarray(set_off, np.uint64)
        ns = set_off.size - 1
        sig = np.empty((ns, S), np.float64)
        tw = np.empty(ns, np.float64)
        self._check(lib().d2g_bmh_from_weighted(self._h, _np_ptr(ids), None if w is None else _np_ptr(w), _np_ptr(set_off),
                                                ns, S, _np_ptr(sig), _np_ptr(tw)))
        return sig, tw

    def bmh_from_weighted_ids(self, ids, weights, set_off, S):
        """-> (sig, total_weight, owner uint64[nsets][S]): owner = position within its set of the element that set the register"""
        ids = np.ascontiguousarray(ids, np.uint64)
        w = None if weights is None else np.ascontiguousarray(weights, np.float64)
        set_off = np.ascontiguousarray(set_off, np.uint64)
        ns = set_off.size - 1
        sig = np.empty((ns, S), np.float64)
        tw = np.empty(ns, np.float64)
        own = np.empty((ns, S), np.uint64)
        self._check(lib().d2g_bmh_from_weighted_ids(self._h, _np_ptr(ids), None if w is None else _np_ptr(w), _np_ptr(set_off),
                                                    ns, S, _np_ptr(sig), _np_ptr(tw), _np_ptr(own)))
        return sig, tw, own

    def oph_plan(self, run_start, run_len, genome_run_off, k):
        run_start = np.ascontiguousarray(run_start, np.uint64)
        run_len = np.ascontiguousarray(run_len, np.uint32)
        genome_run_off = np.ascontiguousarray(genome_run_off, np.uint64)
        h = _vp()
        self._check(lib().d2g_oph_plan_create(self._h, _np_ptr(run_start), _np_ptr(run_len), run_start.size,
                                              _np_ptr(genome_run_off), genome_run_off.size - 1, k, C.byref(h)))
        return OphPlan(self, h, genome_run_off.size - 1)

    def oph_sketch_dev(self, plan, packed_dev_ptr, S, regs_dev_ptr, canon=True, xormask=0, stream=None):
        self._check(lib().d2g_oph_sketch_dev(self._h, plan._h, packed_dev_ptr, int(canon), xormask, S, regs_dev_ptr, stream))

    # -- K2 ------------------------------------------------------------------
    def cmp_set(self, sig_bits_host, algo=CMP_AUTO):
        a = np.ascontiguousarray(sig_bits_host)
        assert a.dtype.itemsize == 8 and a.ndim == 2
        h = _vp()
        self._check(lib().d2g_cmp_set_create(self._h, _np_ptr(a), a.shape[0], a.shape[1], algo, C.byref(h)))
        return CmpSet(self, h, a.shape[0], a.shape[1])

    def cmp_set_dev(self, dev_ptr, N, S, algo=CMP_AUTO, stream=None):
        h = _vp()
        self._check(lib().d2g_cmp_set_create_dev(self._h, dev_ptr, N, S, algo, stream, C.byref(h)))
        return CmpSet(self, h, N, S)

    def cmp_set_from_planes(self, N, S, planes_dev_ptr, meta_dev_ptr):
        h = _vp()
        self._check(lib().d2g_cmp_set_from_planes_dev(self._h, N, S, planes_dev_ptr, meta_dev_ptr, C.byref(h)))
        return CmpSet(self, h, N, S)

    def pack_column_slices_dev(self, rows_dev_ptr, n, S, W, out_dev_ptr, stream=None):
        self._check(lib().d2g_pack_column_slices_dev(self._h, rows_dev_ptr, n, S, W, out_dev_ptr, stream))

    def cmp_eqcount_ut(self, sig_bits_host, r0=0, r1=None, algo=CMP_AUTO):
        a = np.ascontiguousarray(sig_bits_host)
        assert a.dtype.itemsize == 8 and a.ndim == 2
        N, S = a.shape
        r1 = N if r1 is None else r1
        out = np.empty(ut_count(N, r0, r1), np.uint32)
        self._check(lib().d2g_cmp_eqcount_ut(self._h, _np_ptr(a), N, S, r0, r1, algo, _np_ptr(out)))
        return out

    def cmp_dist_ut(self, sig_bits_host, cards, measure=SIMILARITY, k=31, multiset_space=False, r0=0, r1=None,
                    algo=CMP_AUTO, nthreads=1):
        a = np.ascontiguousarray(sig_bits_host)
        assert a.dtype.itemsize == 8 and a.ndim == 2
        cards = np.ascontiguousarray(cards, np.float64)
        N, S = a.shape
        r1 = N if r1 is None else r1
        out = np.empty(ut_count(N, r0, r1), np.float32)
        self._check(lib().d2g_cmp_dist_ut(self._h, _np_ptr(a), _np_ptr(cards), N, S, r0, r1, measure, k,
                                          int(multiset_space), algo, nthreads, _np_ptr(out)))
        return out

    # -- raw device memory (tests / bench without torch) ------------------------
    def malloc(self, nbytes):
        p = _vp()
        self._check(lib().d2g_malloc(self._h, nbytes, C.byref(p)))
        return p.value

    def free(self, ptr):
        self._check(lib().d2g_free(self._h, ptr))

    def h2d(self, dptr, arr, stream=None):
        arr = np.ascontiguousarray(arr)
        self._check(lib().d2g_memcpy_h2d(self._h, dptr, _np_ptr(arr), arr.nbytes, stream))

    def d2h(self, arr, dptr, stream=None):
        assert arr.flags["C_CONTIGUOUS"]
        self._check(lib().d2g_memcpy_d2h(self._h, _np_ptr(arr), dptr, arr.nbytes, stream))


class Sketcher:
    """persistent K1 front end (d2g_sketcher): reuses device buffers across calls"""

    def __init__(self, ctx):
        self.ctx = ctx
        self._h = None
        h = _vp()
        ctx._check(lib().d2g_sketcher_create(ctx._h, C.byref(h)))
        self._h = h

    def run(self, sp, S, canon=True, xormask=0):
        packed, rs, rl, go = sp.arrays()
        n = go.size - 1
        regs = np.empty((n, oph_m(S)), np.uint64)
        self.ctx._check(lib().d2g_sketcher_run(self._h, _np_ptr(packed), packed.size, _np_ptr(rs), _np_ptr(rl), rs.size,
                                               _np_ptr(go), n, sp.k, int(canon), xormask, S, _np_ptr(regs)))
        return regs

    def run_bmh(self, sp, S, canon=True, xormask=0, count_threshold=0.0):
        packed, rs, rl, go = sp.arrays()
        n = go.size - 1
        sig = np.empty((n, S), np.float64)
        tw = np.empty(n, np.float64)
        self.ctx._check(lib().d2g_sketcher_run_bmh(self._h, _np_ptr(packed), packed.size, _np_ptr(rs), _np_ptr(rl), rs.size,
                                                   _np_ptr(go), n, sp.k, int(canon), xormask, S, count_threshold,
                                                   _np_ptr(sig), _np_ptr(tw)))
        return sig, tw

    # -- K0: FASTA bytes -> packed run stream on the GPU ------------------------------------------
    def ingest_fasta(self, buffers, k, genome_nfiles=None, pinned=None):
        """buffers: list of bytes-like FASTA inputs; genome_nfiles: files per genome (default one each).  The bytes are laid out
        16-byte aligned in ONE host buffer (`pinned`: a PinnedArray to reuse, else a plain numpy array) and parsed on the
        device.  Raises D2GError(D2G_ERR_UNSUPPORTED) for inputs only the host parser handles.
        -> (run_start u64, run_len u32, genome_run_off u64, genome_nkmers u64, nbases)"""
        nf = len(buffers)
        lens = np.array([len(b) for b in buffers], np.uint64)
        offs = np.zeros(nf, np.uint64)
        pos = 0
        for i in range(nf):
            offs[i] = pos
            pos += (int(lens[i]) + 15) // 16 * 16
        total = max(pos, 16)
        raw = pinned.array[:total] if pinned is not None else np.empty(total, np.uint8)
        for i, b in enumerate(buffers):
            raw[int(offs[i]):int(offs[i]) + int(lens[i])] = np.frombuffer(b, np.uint8)
        gn = np.ones(nf, np.uint64) if genome_nfiles is None else np.asarray(genome_nfiles, np.uint64)
        gfo = np.concatenate([[0], np.cumsum(gn)]).astype(np.uint64)
        n = gfo.size - 1
        self.ingest_raw(raw, pos, offs, lens, gfo, k)
        return self.ingested_runs(n)

    def ingest_raw(self, raw, raw_bytes, file_off, file_len, genome_file_off, k):
        file_off = np.ascontiguousarray(file_off, np.uint64)
        file_len = np.ascontiguousarray(file_len, np.uint64)
        genome_file_off = np.ascontiguousarray(genome_file_off, np.uint64)
        self.k = k
        self.ctx._check(lib().d2g_sketcher_ingest_fasta(self._h, _np_ptr(raw), raw_bytes, _np_ptr(file_off), _np_ptr(file_len), file_off.size,
                                                        _np_ptr(genome_file_off), genome_file_off.size - 1, k))

    def ingested_runs(self, n):
        rs, rl, go, nk = _vp(), _vp(), _vp(), _vp()
        nrun, nb = _sz(), _u64()
        self.ctx._check(lib().d2g_sketcher_ingested_runs(self._h, C.byref(rs), C.byref(rl), C.byref(nrun), C.byref(go), C.byref(nk), C.byref(nb)))
        nr = nrun.value

        def arr(p, cnt, dt):
            if not cnt:
                return np.zeros(0, dt)
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(cnt * np.dtype(dt).itemsize,)).view(dt).copy()
        return arr(rs, nr, np.uint64), arr(rl, nr, np.uint32), arr(go, n + 1, np.uint64), arr(nk, n, np.uint64), int(nb.value)

    def run_ingested(self, runs, S, canon=True, xormask=0):
        """K1 over the stream ingested last (packed == NULL): -> regs u64 [n][m]"""
        rs, rl, go = runs[0], runs[1], runs[2]
        n = go.size - 1
        regs = np.empty((n, oph_m(S)), np.uint64)
        self.ctx._check(lib().d2g_sketcher_run(self._h, None, 0, _np_ptr(rs), _np_ptr(rl), rs.size, _np_ptr(go), n, self.k, int(canon),
                                               xormask, S, _np_ptr(regs)))
        return regs

    def run_bmh_ingested(self, runs, S, canon=True, xormask=0, count_threshold=0.0):
        rs, rl, go = runs[0], runs[1], runs[2]
        n = go.size - 1
        sig = np.empty((n, S), np.float64)
        tw = np.empty(n, np.float64)
        self.ctx._check(lib().d2g_sketcher_run_bmh(self._h, None, 0, _np_ptr(rs), _np_ptr(rl), rs.size, _np_ptr(go), n, self.k, int(canon),
                                                   xormask, S, count_threshold, _np_ptr(sig), _np_ptr(tw)))
        return sig, tw

    def close(self):
        if getattr(self, "_h", None):
            lib().d2g_sketcher_destroy(self._h)
            self._h = None

    __del__ = close


class PinnedArray:
    """page-locked host bytes (d2g_malloc_host) as a numpy uint8 array: uploads from it are one DMA"""

    def __init__(self, ctx, nbytes):
        self.ctx, self._p = ctx, _vp()
        ctx._check(lib().d2g_malloc_host(ctx._h, nbytes, C.byref(self._p)))
        self.array = np.ctypeslib.as_array(C.cast(self._p, C.POINTER(C.c_uint8)), shape=(nbytes,))

    def close(self):
        if getattr(self, "_p", None):
            self.array = None
            lib().d2g_free_host(self.ctx._h, self._p)
            self._p = None

    __del__ = close


class OphPlan:
    def __init__(self, ctx, h, n):
        self.ctx, self._h, self.n = ctx, h, n

    @property
    def nkmers(self):
        return int(lib().d2g_oph_plan_nkmers(self._h))

    @property
    def nbases(self):
        return int(lib().d2g_oph_plan_nbases(self._h))

    def close(self):
        if self._h:
            lib().d2g_oph_plan_destroy(self._h)
            self._h = None

    __del__ = close


class CmpSet:
    """Device-resident prepared signature matrix (d2g_cmp_set)."""

    def __init__(self, ctx, h, N, S, owned=True):
        self.ctx, self._h, self.N, self.S, self._owned = ctx, h, N, S, owned

    @property
    def algo(self):
        return int(lib().d2g_cmp_set_algo(self._h))

    def close(self):
        if getattr(self, "_h", None):
            if self._owned:
                lib().d2g_cmp_set_destroy(self._h)
            self._h = None

    __del__ = close

    def export_operand_dev(self, planes_out_ptr, meta_out_ptr, stream=None):
        self.ctx._check(lib().d2g_cmp_set_export_operand_dev(self.ctx._h, self._h, planes_out_ptr, meta_out_ptr, stream))

    def update_dev(self, dev_ptr, stream=None):
        self.ctx._check(lib().d2g_cmp_set_update_dev(self.ctx._h, self._h, dev_ptr, stream))

    def status(self, stream=None):
        """synchronises; raises D2GError(D2G_ERR_INTERNAL) if the asynchronous prepare overflowed"""
        self.ctx._check(lib().d2g_cmp_set_status(self.ctx._h, self._h, stream))

    def sparse_info(self, stream=None):
        """-> dict of the sparse path's last upper-triangle launch (synchronises)"""
        a = np.zeros(4, np.uint32)
        self.ctx._check(lib().d2g_cmp_set_sparse_info(self.ctx._h, self._h, stream, a.ctypes.data))
        return sparse_info_dict(a)

    def debug_pairs(self, cap=1 << 24, stream=None):
        """-> (pairs [n][2] uint32: the pair list of the last prepare (i < j), roots [N] uint32: the family root of every sketch)"""
        buf = np.empty(cap, np.uint64)
        roots = np.empty(self.N, np.uint32)
        n = _sz()
        self.ctx._check(lib().d2g_cmp_set_debug_pairs(self.ctx._h, self._h, stream, buf.ctypes.data, cap, C.byref(n), roots.ctypes.data))
        e = buf[:n.value]
        e = e[e != np.uint64(0xFFFFFFFFFFFFFFFF)]                     # empty slots (two outsiders of one value in one segment)
        return np.stack([(e & np.uint64(0xFFFFFFFF)).astype(np.uint32), (e >> np.uint64(32)).astype(np.uint32)], axis=1), roots

    def planes(self, stream=None):
        """-> (max shared values per column + 1, max id planes of a group, mean id planes); zeros for DIRECT"""
        md, nb, mean = C.c_uint(), _int(), _f32()
        self.ctx._check(lib().d2g_cmp_set_planes(self.ctx._h, self._h, stream, C.byref(md), C.byref(nb), C.byref(mean)))
        return int(md.value), int(nb.value), float(mean.value)

    def eqcount_ut_dev(self, out_dev_ptr, r0=0, r1=None, stream=None):
        r1 = self.N if r1 is None else r1
        self.ctx._check(lib().d2g_cmp_eqcount_ut_dev(self.ctx._h, self._h, r0, r1, out_dev_ptr, stream))

    def lut_ut_dev(self, lut_dev_ptr, out_dev_ptr, r0=0, r1=None, stream=None):
        r1 = self.N if r1 is None else r1
        self.ctx._check(lib().d2g_cmp_lut_ut_dev(self.ctx._h, self._h, r0, r1, lut_dev_ptr, out_dev_ptr, stream))

    def prefill_ut_dev(self, out_dev_ptr, r0=0, r1=None, lut_dev_ptr=None, stream=None):
        """the fill of the NEXT upper-triangle launch into `out`, enqueued now on `stream` (counts when no table is given)"""
        r1 = self.N if r1 is None else r1
        if lut_dev_ptr is None:
            self.ctx._check(lib().d2g_cmp_ut_prefill_dev(self.ctx._h, self._h, r0, r1, out_dev_ptr, None, None, stream))
        else:
            self.ctx._check(lib().d2g_cmp_ut_prefill_dev(self.ctx._h, self._h, r0, r1, None, lut_dev_ptr, out_dev_ptr, stream))

    def forget(self):
        """the set's next prepare decides as the first prepare of a new set does (measurements)"""
        self.ctx._check(lib().d2g_cmp_set_forget(self.ctx._h, self._h))

    def announce_ut_dev(self, out_dev_ptr, r0=0, r1=None, lut_dev_ptr=None):
        """the output of the NEXT upper-triangle launch, told ahead of the update_dev that precedes it: the prepare carries the fill"""
        r1 = self.N if r1 is None else r1
        if lut_dev_ptr is None:
            self.ctx._check(lib().d2g_cmp_ut_announce_dev(self.ctx._h, self._h, r0, r1, out_dev_ptr, None, None))
        else:
            self.ctx._check(lib().d2g_cmp_ut_announce_dev(self.ctx._h, self._h, r0, r1, None, lut_dev_ptr, out_dev_ptr))

    def gtlt_ut_dev(self, gt_dev_ptr, lt_dev_ptr, r0=0, r1=None, stream=None):
        r1 = self.N if r1 is None else r1
        self.ctx._check(lib().d2g_cmp_gtlt_ut_dev(self.ctx._h, self._h, r0, r1, gt_dev_ptr, lt_dev_ptr, stream))

    def eqcount_rect_dev(self, out_dev_ptr, a0, a1, b0, b1, stream=None):
        self.ctx._check(lib().d2g_cmp_eqcount_rect_dev(self.ctx._h, self._h, a0, a1, b0, b1, out_dev_ptr, stream))

    # host-returning helpers built on the raw device-memory calls
    def eqcount_ut(self, r0=0, r1=None):
        r1 = self.N if r1 is None else r1
        out = np.empty(ut_count(self.N, r0, r1) if r0 <= r1 <= self.N else 0, np.uint32)
        d = self.ctx.malloc(max(out.nbytes, 4))
        try:
            self.eqcount_ut_dev(d, r0, r1)
            if out.size:
                self.ctx.d2h(out, d)
        finally:
            self.ctx.free(d)
        return out

    def gtlt_ut(self, r0=0, r1=None):
        r1 = self.N if r1 is None else r1
        n = ut_count(self.N, r0, r1)
        gt, lt = np.empty(n, np.uint32), np.empty(n, np.uint32)
        if n == 0:
            return gt, lt
        dg, dl = self.ctx.malloc(gt.nbytes), self.ctx.malloc(lt.nbytes)
        try:
            self.gtlt_ut_dev(dg, dl, r0, r1)
            self.ctx.d2h(gt, dg)
            self.ctx.d2h(lt, dl)
        finally:
            self.ctx.free(dg)
            self.ctx.free(dl)
        return gt, lt

    def gtlt_rect(self, a0, a1, b0, b1):
        gt, lt = np.empty((a1 - a0, b1 - b0), np.uint32), np.empty((a1 - a0, b1 - b0), np.uint32)
        if gt.size == 0:
            return gt, lt
        dg, dl = self.ctx.malloc(gt.nbytes), self.ctx.malloc(lt.nbytes)
        try:
            self.ctx._check(lib().d2g_cmp_gtlt_rect_dev(self.ctx._h, self._h, a0, a1, b0, b1, dg, dl, None))
            self.ctx.d2h(gt, dg)
            self.ctx.d2h(lt, dl)
        finally:
            self.ctx.free(dg)
            self.ctx.free(dl)
        return gt, lt

    def eqcount_rect(self, a0, a1, b0, b1):
        out = np.empty((a1 - a0, b1 - b0), np.uint32)
        if out.size == 0:
            return out
        d = self.ctx.malloc(out.nbytes)
        try:
            self.eqcount_rect_dev(d, a0, a1, b0, b1)
            self.ctx.d2h(out, d)
        finally:
            self.ctx.free(d)
        return out

    def lut_ut(self, lut, r0=0, r1=None):
        r1 = self.N if r1 is None else r1
        lut = np.ascontiguousarray(lut, np.float32)
        out = np.empty(ut_count(self.N, r0, r1), np.float32)
        if out.size == 0:
            return out
        dl, d = self.ctx.malloc(lut.nbytes), self.ctx.malloc(out.nbytes)
        try:
            self.ctx.h2d(dl, lut)
            self.lut_ut_dev(dl, d, r0, r1)
            self.ctx.d2h(out, d)
        finally:
            self.ctx.free(dl)
            self.ctx.free(d)
        return out


# ---------------------------------------------------------------- multi-GPU (RCCL communicator + row-sharded engine)
COMM_ID_BYTES = 128


def comm_unique_id():
    """128 opaque bytes (ncclGetUniqueId) rank 0 hands to every rank"""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = lib().d2g_comm_unique_id(buf)
    if rc:
        raise D2GError(rc, "d2g_comm_unique_id (RCCL not loadable?)")
    return buf.raw


class Comm:
    def __init__(self, ctx, h):
        self.ctx, self._h = ctx, h

    @classmethod
    def create(cls, ctx, rank=0, world=1, unique_id=None):
        h = _vp()
        ctx._check(lib().d2g_comm_create(ctx._h, unique_id, rank, world, C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def create_all(cls, ctxs):
        """one process, one communicator per context: RCCL clique over distinct devices, loopback transport otherwise"""
        n = len(ctxs)
        arr = (_vp * n)(*[c._h for c in ctxs])
        out = (_vp * n)()
        ctxs[0]._check(lib().d2g_comm_create_all(arr, n, out))
        return [cls(c, _vp(out[i])) for i, c in enumerate(ctxs)]

    rank = property(lambda self: int(lib().d2g_comm_rank(self._h)))
    world = property(lambda self: int(lib().d2g_comm_world(self._h)))
    is_rccl = property(lambda self: bool(lib().d2g_comm_is_rccl(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            lib().d2g_comm_destroy(self._h)
            self._h = None

    __del__ = close


class AllPairs:
    """one rank of the row-sharded all-pairs engine (d2g_allpairs)"""

    def __init__(self, ctx, comm, N, S):
        self.ctx, self.comm, self.N, self.S = ctx, comm, N, S
        self._h = None
        h = _vp()
        ctx._check(lib().d2g_allpairs_create(ctx._h, comm._h, N, S, C.byref(h)))
        self._h = h
        a, b = _sz(), _sz()
        lib().d2g_allpairs_rows_held(h, C.byref(a), C.byref(b))
        self.rows_held = (a.value, b.value)
        lib().d2g_allpairs_rows_computed(h, C.byref(a), C.byref(b))
        self.rows_computed = (a.value, b.value)
        self.r0, self.r1 = self.rows_computed

    def prepare_dev(self, rows_ptr, stream=None):
        self.ctx._check(lib().d2g_allpairs_prepare_dev(self._h, rows_ptr, stream))

    def operand(self):
        """the gathered operand as a (non-owning) CmpSet"""
        return CmpSet(self.ctx, _vp(lib().d2g_allpairs_operand(self._h)), self.N, self.S, owned=False)

    def status(self, stream=None):
        """synchronises; raises if ANY rank's sharded prepare overflowed its rank table (results invalid everywhere)"""
        self.ctx._check(lib().d2g_allpairs_status(self._h, stream))

    @property
    def chunks(self):
        return lib().d2g_allpairs_chunks(self._h)

    PHASE_NAMES = ("pack", "x1", "prepare", "x2", "derive", "pair", "order", "fill")

    def set_phase_timing(self, on=True):
        """bracket every phase of the next prepare/step with timing events (switch off again for timed runs)"""
        self.ctx._check(lib().d2g_allpairs_set_phase_timing(self._h, int(bool(on))))

    def phase_times(self):
        """synchronises; -> [{"phase", "chunk", "start_ms", "ms"}] of the last step enqueued with phase timing on"""
        cap = 64
        kind, chunk = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        start, dur = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
        n = _int()
        self.ctx._check(lib().d2g_allpairs_phase_times(self._h, cap, C.byref(n), kind.ctypes.data, chunk.ctypes.data,
                                                       start.ctypes.data, dur.ctypes.data))
        return [{"phase": self.PHASE_NAMES[int(kind[i])], "chunk": int(chunk[i]), "start_ms": float(start[i]), "ms": float(dur[i])}
                for i in range(n.value)]

    def sparse_info(self):
        """-> dict of the sparse-tile path in this rank's last pair phase (synchronises the device); keys as CmpSet.sparse_info"""
        a = np.zeros(4, np.uint32)
        self.ctx._check(lib().d2g_allpairs_sparse_info(self._h, a.ctypes.data))
        return sparse_info_dict(a)

    def step_lut_dev(self, rows_ptr, lut_ptr, out_ptr, stream=None):
        self.ctx._check(lib().d2g_allpairs_step_lut_dev(self._h, rows_ptr, lut_ptr, out_ptr, stream))

    def step_eqcount_dev(self, rows_ptr, out_ptr, stream=None):
        self.ctx._check(lib().d2g_allpairs_step_eqcount_dev(self._h, rows_ptr, out_ptr, stream))

    def enqueue_lut_dev(self, rows_ptr, lut_ptr, out_ptr, stream=None, input_ready=False):
        self.ctx._check(lib().d2g_allpairs_enqueue_lut_dev(self._h, rows_ptr, lut_ptr, out_ptr, stream, int(input_ready)))

    def close(self):
        if getattr(self, "_h", None):
            lib().d2g_allpairs_destroy(self._h)
            self._h = None

    __del__ = close


def allpairs_step_all(engs, rows_ptrs, lut_ptrs, out_ptrs, streams=None):
    """single-process multi-rank step: every collective phase is issued for all ranks inside one group"""
    n = len(engs)
    E = (_vp * n)(*[e._h for e in engs])
    R = (_vp * n)(*rows_ptrs)
    L = (_vp * n)(*lut_ptrs) if lut_ptrs is not None else None
    O = (_vp * n)(*out_ptrs)
    St = (_vp * n)(*(streams or [None] * n))
    engs[0].ctx._check(lib().d2g_allpairs_step_all(E, n, R, L, O, St))


def allpairs_prepare_all(engs, rows_ptrs, streams=None):
    n = len(engs)
    E = (_vp * n)(*[e._h for e in engs])
    R = (_vp * n)(*rows_ptrs)
    St = (_vp * n)(*(streams or [None] * n))
    engs[0].ctx._check(lib().d2g_allpairs_prepare_all(E, n, R, St))


def bcast_sigs(ctxs, comms, sig_bits_host):
    """-> list of device pointers (one per context) holding the whole matrix (d2g_bcast_sigs)"""
    a = np.ascontiguousarray(sig_bits_host, np.uint64)
    N, S = a.shape
    n = len(ctxs)
    Cx = (_vp * n)(*[c._h for c in ctxs])
    Cm = (_vp * n)(*[c._h for c in comms])
    out = (_vp * n)()
    ctxs[0]._check(lib().d2g_bcast_sigs(Cx, Cm, n, _np_ptr(a), N, S, out))
    return [int(out[i]) for i in range(n)]

"""ctypes wrapper over oracle/libd2oracle.so  --  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product path (dashing2_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SIMILARITY, CONTAINMENT, SYMMETRIC_CONTAINMENT, POISSON_LLR, INTERSECTION, UNION_SIZE = range(6)


def build(march=None, out=None):
    """Compile the oracle (gcc). Returns the path of the shared object."""
    cmd = ["make", "-s", "-C", _HERE]
    if march:
        cmd.append(f"MARCH={march}")
    if out:
        cmd.append(f"OUT={out}")
    subprocess.check_call(cmd)
    return out or os.path.join(_HERE, "libd2oracle.so")


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def load(path=None):
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    so = path or os.path.join(_HERE, "libd2oracle.so")
    if not os.path.exists(so):
        build()
    lib = C.CDLL(so)
    u64, sz, dbl, i32 = C.c_uint64, C.c_size_t, C.c_double, C.c_int
    pu64, pdbl, pf32, pu32 = C.POINTER(u64), C.POINTER(dbl), C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    sig = {
        "d2o_wang_hash": (u64, [u64]),
        "d2o_wang_inverse": (u64, [u64]),
        "d2o_mt19937_64_first": (u64, [u64]),
        "d2o_seed_mask": (u64, [u64]),
        "d2o_default_xormask": (u64, []),
        "d2o_oph_xor_const": (u64, []),
        "d2o_maskfn": (u64, [u64, u64]),
        "d2o_oph_id": (u64, [u64]),
        "d2o_wyhash64_stateless": (u64, [pu64]),
        "d2o_oph_bucket": (C.c_uint32, [u64, sz]),
        "d2o_regs_getcard": (dbl, [pu64, sz]),
        "d2o_regs_data": (None, [pu64, sz, pdbl]),
        "d2o_sketch_buffer": (i32, [C.c_char_p, sz, i32, i32, u64, sz, pu64, pdbl, pdbl, pu64]),
        "d2o_sketch_file": (i32, [C.c_char_p, i32, i32, u64, sz, pu64, pdbl, pdbl, pu64]),
        "d2o_sketch_files": (i32, [C.POINTER(C.c_char_p), sz, i32, i32, u64, sz, pdbl, pdbl, i32]),
        "d2o_densify": (sz, [pdbl, sz]),
        "d2o_count_gtlt": (None, [pdbl, pdbl, sz, pu64, pu64]),
        "d2o_count_eq": (u64, [pdbl, pdbl, sz]),
        "d2o_compare_from_gtlt": (C.c_float, [u64, u64, sz, dbl, dbl, i32, i32]),
        "d2o_compare_from_neq": (C.c_float, [u64, sz, dbl, dbl, i32, i32]),
        "d2o_compare": (C.c_float, [pdbl, pdbl, sz, sz, sz, i32, i32]),
        "d2o_allpairs_ut": (None, [pdbl, pdbl, sz, sz, i32, i32, pf32, i32, sz]),
        "d2o_allpairs_ut_rows": (None, [pdbl, pdbl, sz, sz, i32, i32, sz, sz, pf32, i32, sz]),
        "d2o_eqcounts_ut": (None, [pdbl, sz, sz, pu32]),
        "d2o_eqcounts_ut_rows": (None, [pdbl, sz, sz, sz, sz, pu32]),
        "d2o_default_batchsize": (sz, [sz, sz, C.c_uint]),
        "d2o_dlog": (dbl, [dbl]),
        "d2o_bmh_create": (C.c_void_p, [sz]),
        "d2o_bmh_destroy": (None, [C.c_void_p]),
        "d2o_bmh_reset": (None, [C.c_void_p]),
        "d2o_bmh_update": (u64, [C.c_void_p, u64, dbl]),
        "d2o_bmh_total_weight": (dbl, [C.c_void_p]),
        "d2o_bmh_data": (None, [C.c_void_p, pdbl]),
        "d2o_bmh_from_weighted": (i32, [pu64, pdbl, sz, sz, pdbl, pdbl]),
        "d2o_bmh_from_weighted_ids": (i32, [pu64, pdbl, sz, sz, pdbl, pdbl, pu64]),
        "d2o_kmer_count_buffer": (i32, [C.c_char_p, sz, i32, i32, u64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                        C.POINTER(sz), pu64]),
        "d2o_free": (None, [C.c_void_p]),
        "d2o_sketch_buffer_byseq": (i32, [C.c_char_p, sz, i32, i32, u64, sz, i32, dbl, C.POINTER(sz), C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
        "d2o_bmh_sketch_buffer": (i32, [C.c_char_p, sz, i32, i32, u64, sz, dbl, pdbl, pdbl, pu64]),
        "d2o_bmh_sketch_file": (i32, [C.c_char_p, i32, i32, u64, sz, dbl, pdbl, pdbl, pu64]),
        "d2o_bmh_sketch_files": (i32, [C.POINTER(C.c_char_p), sz, i32, i32, u64, sz, dbl, pdbl, pdbl, pu64]),
    }
    for name, (res, args) in sig.items():
        f = getattr(lib, name)
        f.restype = res
        f.argtypes = args
    if path is None:
        _LIB = lib
    return lib


# ---- thin numpy-level helpers ------------------------------------------------

def wang_hash(x):
    return int(load().d2o_wang_hash(x & 0xFFFFFFFFFFFFFFFF))


def wang_inverse(x):
    return int(load().d2o_wang_inverse(x & 0xFFFFFFFFFFFFFFFF))


def oph_m(S):
    return S + (S & 1)


def sketch_buffer(data: bytes, k=31, canon=True, xormask=0, S=1024):
    """-> (regs u64[m], sigs f64[S], card, nkmers)"""
    lib = load()
    m = oph_m(S)
    regs = np.empty(m, np.uint64)
    sig = np.empty(S, np.float64)
    card = C.c_double()
    nk = C.c_uint64()
    rc = lib.d2o_sketch_buffer(data, len(data), k, int(canon), xormask, S, _p(regs, C.c_uint64),
                               _p(sig, C.c_double), C.byref(card), C.byref(nk))
    if rc:
        raise RuntimeError(f"oracle sketch failed rc={rc}")
    return regs, sig, card.value, nk.value


def sketch_file(path, k=31, canon=True, xormask=0, S=1024):
    lib = load()
    m = oph_m(S)
    regs = np.empty(m, np.uint64)
    sig = np.empty(S, np.float64)
    card = C.c_double()
    nk = C.c_uint64()
    rc = lib.d2o_sketch_file(os.fsencode(path), k, int(canon), xormask, S, _p(regs, C.c_uint64),
                             _p(sig, C.c_double), C.byref(card), C.byref(nk))
    if rc:
        raise RuntimeError(f"oracle sketch_file failed rc={rc}")
    return regs, sig, card.value, nk.value


def sketch_files(paths, k=31, canon=True, xormask=0, S=1024, nthreads=1):
    lib = load()
    n = len(paths)
    arr = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
    sigs = np.empty((n, S), np.float64)
    cards = np.empty(n, np.float64)
    rc = lib.d2o_sketch_files(arr, n, k, int(canon), xormask, S, _p(sigs, C.c_double), _p(cards, C.c_double), nthreads)
    if rc:
        raise RuntimeError(f"oracle sketch_files failed rc={rc}")
    return sigs, cards


def regs_finalize(regs):
    """regs u64[m] -> (sig f64[m], card)"""
    lib = load()
    regs = np.ascontiguousarray(regs, np.uint64)
    sig = np.empty(regs.size, np.float64)
    lib.d2o_regs_data(_p(regs, C.c_uint64), regs.size, _p(sig, C.c_double))
    return sig, float(lib.d2o_regs_getcard(_p(regs, C.c_uint64), regs.size))


def densify(sig):
    sig = np.ascontiguousarray(sig, np.float64).copy()
    n = load().d2o_densify(_p(sig, C.c_double), sig.size)
    return sig, int(n)


def count_gtlt(a, b):
    a = np.ascontiguousarray(a, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    gt, lt = C.c_uint64(), C.c_uint64()
    load().d2o_count_gtlt(_p(a, C.c_double), _p(b, C.c_double), a.size, C.byref(gt), C.byref(lt))
    return gt.value, lt.value


def compare_from_gtlt(gt, lt, S, lhc, rhc, measure=SIMILARITY, k=31):
    return float(load().d2o_compare_from_gtlt(gt, lt, S, lhc, rhc, measure, k))


def compare_from_neq(neq, S, lhc, rhc, measure=SIMILARITY, k=31):
    return float(load().d2o_compare_from_neq(neq, S, lhc, rhc, measure, k))


def allpairs_ut(sigs, cards, measure=SIMILARITY, k=31, nthreads=1, batch=0, rows=None):
    lib = load()
    sigs = np.ascontiguousarray(sigs, np.float64)
    cards = np.ascontiguousarray(cards, np.float64)
    N, S = sigs.shape
    bs = lib.d2o_default_batchsize(batch, S, nthreads)
    if rows is None:
        out = np.empty(N * (N - 1) // 2, np.float32)
        lib.d2o_allpairs_ut(_p(sigs, C.c_double), _p(cards, C.c_double), N, S, measure, k, _p(out, C.c_float), nthreads, bs)
    else:
        r0, r1 = rows
        n = sum(N - r - 1 for r in range(r0, r1))
        out = np.empty(n, np.float32)
        lib.d2o_allpairs_ut_rows(_p(sigs, C.c_double), _p(cards, C.c_double), N, S, measure, k, r0, r1,
                                 _p(out, C.c_float), nthreads, bs)
    return out


def eqcounts_ut(sigs):
    sigs = np.ascontiguousarray(sigs, np.float64)
    N, S = sigs.shape
    out = np.empty(N * (N - 1) // 2, np.uint32)
    load().d2o_eqcounts_ut(_p(sigs, C.c_double), N, S, _p(out, C.c_uint32))
    return out


def eqcounts_rows(sigs, r0, r1):
    """rows [r0, r1) of the condensed upper triangle of equality counts"""
    sigs = np.ascontiguousarray(sigs, np.float64)
    N, S = sigs.shape
    out = np.empty(sum(N - r - 1 for r in range(r0, min(r1, N))), np.uint32)
    load().d2o_eqcounts_ut_rows(_p(sigs, C.c_double), N, S, r0, r1, _p(out, C.c_uint32))
    return out


# ---- --multiset path (R11 exact k-mer counts, R12 BagMinHash; see d2_bmh_oracle.c) ----------

def dlog(u):
    return float(load().d2o_dlog(float(u)))


def bmh_from_weighted(ids, weights, S):
    """-> (sig float64[S], total_weight)"""
    ids = np.ascontiguousarray(ids, np.uint64)
    w = None if weights is None else np.ascontiguousarray(weights, np.float64)
    sig = np.empty(S, np.float64)
    tw = C.c_double()
    rc = load().d2o_bmh_from_weighted(_p(ids, C.c_uint64), None if w is None else _p(w, C.c_double), ids.size, S,
                                      _p(sig, C.c_double), C.byref(tw))
    assert rc == 0
    return sig, tw.value


def bmh_from_weighted_ids(ids, weights, S):
    """-> (sig float64[S], total_weight, owner uint64[S])"""
    ids = np.ascontiguousarray(ids, np.uint64)
    w = None if weights is None else np.ascontiguousarray(weights, np.float64)
    sig = np.empty(S, np.float64)
    own = np.empty(S, np.uint64)
    tw = C.c_double()
    rc = load().d2o_bmh_from_weighted_ids(_p(ids, C.c_uint64), None if w is None else _p(w, C.c_double), ids.size, S,
                                          _p(sig, C.c_double), C.byref(tw), _p(own, C.c_uint64))
    assert rc == 0
    return sig, tw.value, own


def kmer_count_buffer(buf, k, canon=True, xormask=0):
    """-> (keys uint64[nd] sorted, counts uint32[nd], nkmers)"""
    keys, counts, nd, nk = C.c_void_p(), C.c_void_p(), C.c_size_t(), C.c_uint64()
    rc = load().d2o_kmer_count_buffer(buf, len(buf), k, int(canon), xormask, C.byref(keys), C.byref(counts),
                                      C.byref(nd), C.byref(nk))
    assert rc == 0
    n = nd.value
    ko = np.ctypeslib.as_array(C.cast(keys, C.POINTER(C.c_uint64)), (max(n, 1),))[:n].copy()
    co = np.ctypeslib.as_array(C.cast(counts, C.POINTER(C.c_uint32)), (max(n, 1),))[:n].copy()
    load().d2o_free(keys)
    load().d2o_free(counts)
    return ko, co, nk.value


def bmh_sketch_buffer(buf, k, S, canon=True, xormask=0, count_threshold=0.0):
    """-> (sig float64[S], total_weight, nkmers)"""
    sig = np.empty(S, np.float64)
    tw, nk = C.c_double(), C.c_uint64()
    rc = load().d2o_bmh_sketch_buffer(buf, len(buf), k, int(canon), xormask, S, count_threshold,
                                      _p(sig, C.c_double), C.byref(tw), C.byref(nk))
    assert rc == 0
    return sig, tw.value, nk.value


def bmh_sketch_files(paths, k, S, canon=True, xormask=0, count_threshold=0.0):
    """-> (sigs float64[n][S], total_weights float64[n], nkmers uint64[n])"""
    n = len(paths)
    arr = (C.c_char_p * n)(*[p.encode() for p in paths])
    sigs = np.empty((n, S), np.float64)
    tw = np.empty(n, np.float64)
    nk = np.empty(n, np.uint64)
    rc = load().d2o_bmh_sketch_files(arr, n, k, int(canon), xormask, S, count_threshold, _p(sigs, C.c_double),
                                     _p(tw, C.c_double), _p(nk, C.c_uint64))
    assert rc == 0, rc
    return sigs, tw, nk


def sketch_buffer_byseq(buf, k, S, canon=True, xormask=0, multiset=False, count_threshold=0.0):
    """--parse-by-seq: -> (names list[str], sigs float64[n][S], cards float64[n])"""
    n, sigs, cards, names = C.c_size_t(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    rc = load().d2o_sketch_buffer_byseq(buf, len(buf), k, int(canon), xormask, S, int(multiset), count_threshold,
                                        C.byref(n), C.byref(sigs), C.byref(cards), C.byref(names))
    assert rc == 0
    nn = n.value
    so = np.ctypeslib.as_array(C.cast(sigs, C.POINTER(C.c_double)), (max(nn * S, 1),))[:nn * S].reshape(nn, S).copy()
    co = np.ctypeslib.as_array(C.cast(cards, C.POINTER(C.c_double)), (max(nn, 1),))[:nn].copy()
    nm = C.string_at(names).decode().split("\n")[:nn] if nn else []
    for p in (sigs, cards, names):
        load().d2o_free(p)
    return nm, so, co

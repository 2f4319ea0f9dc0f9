"""CPU tests: the oracle against the reference's own (few) pins and the frozen golden vectors."""
import os

import numpy as np
import pytest

from conftest import GOLDEN


def test_wang_roundtrip_and_kats(oracle):
    O = oracle
    # oph.h:61-65: assert(bh.inverse(bh(133348)) == 133348)
    for x in [133348, 0, 1, 2 ** 64 - 1, 0x724526e320f9967d, 0xdeadbeefcafebabe]:
        assert O.wang_inverse(O.wang_hash(x)) == x
    # frozen known answers of Thomas Wang's mix (regression pins)
    assert O.wang_hash(0) == 0x77cfa1eef01bca90
    assert O.wang_hash(2 ** 64 - 1) == 0x1f89206e3f8ec794
    lib = O.load()
    x = 133348 ^ lib.d2o_oph_xor_const()
    assert O.wang_inverse(lib.d2o_oph_id(133348)) == x          # DHasher round trip (oph.h:64)


def test_seed_constants(oracle):
    lib = oracle.load()
    # std::mt19937_64 known answer (ISO C++: 10000th value of default seed; first value below)
    assert lib.d2o_mt19937_64_first(5489) == 14514284786278117030
    # oph.h:59,142  seed_ = mt19937_64(0x321b919a61cb41f7)()
    assert lib.d2o_mt19937_64_first(0x321b919a61cb41f7) == 0x8f1896f3f85ef4a3
    assert lib.d2o_oph_xor_const() == 0x8f1896f3f85ef4a3 ^ 0x533f8c2151b20f97 == 0xdc271ad2a9ecfb34
    # enums.cpp:131-140
    assert lib.d2o_seed_mask(0) == 0
    assert lib.d2o_seed_mask(13) == oracle.wang_hash(13)
    assert lib.d2o_default_xormask() == 0x724526e320f9967d


@pytest.mark.parametrize("name", ["eqcount_n7_s16", "eqcount_n33_s64", "eqcount_n64_s128", "eqcount_n40_s100"])
def test_eqcounts_vs_reference_numpy(oracle, name):
    """python/parse.py:128-156 pairwise_equality_compare (NumPy fallback) is the independent pin."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    got = oracle.eqcounts_ut(z["sigs"])
    np.testing.assert_array_equal(got, z["expected"])
    # cmp_core.cpp:465 invariant: S - gt - lt == #equal
    sigs = z["sigs"]
    n, S = sigs.shape
    idx = 0
    for i in range(n):
        for j in range(i + 1, n):
            gt, lt = oracle.count_gtlt(sigs[i], sigs[j])
            assert S - gt - lt == got[idx]
            idx += 1


@pytest.mark.parametrize("m,n", [(128, 100000), (1024, 100000), (1024, 10000000), (8192, 10000000), (16384, 10000000)])
def test_oph_cardinality_statistical_kat(oracle, m, n):
    """test/oph.cpp:6-23: insert 0..n-1, expect |card - n|/n ~ 1/sqrt(m) for getcard() and m/sum(data())."""
    lib = oracle.load()
    import ctypes as C
    # drive d2o_oph_update through a tiny C-level loop: reuse sketch state via ctypes struct
    class Oph(C.Structure):
        _fields_ = [("m", C.c_size_t), ("regs", C.POINTER(C.c_uint64)), ("counts", C.POINTER(C.c_double)),
                    ("total_updates", C.c_uint64)]
    lib.d2o_oph_init.argtypes = [C.POINTER(Oph), C.c_size_t]
    lib.d2o_oph_update_range.argtypes = [C.POINTER(Oph), C.c_uint64, C.c_uint64]
    lib.d2o_oph_free.argtypes = [C.POINTER(Oph)]
    s = Oph()
    assert lib.d2o_oph_init(C.byref(s), m) == 0
    lib.d2o_oph_update_range(C.byref(s), 0, n)
    regs = np.ctypeslib.as_array(s.regs, shape=(s.m,)).copy()
    lib.d2o_oph_free(C.byref(s))
    sig, card = oracle.regs_finalize(regs)
    tol = 4.0 / np.sqrt(m)
    assert abs(card - n) / n < tol
    card2 = m / sig.sum()
    assert abs(card2 - n) / n < tol


def test_encoder_semantics(oracle):
    O = oracle
    # window resets on non-ACGT, lowercase accepted, records independent, len<k yields nothing
    fa = b">a\nACGTNACGT\n>b\nacgtacgt\n>c\nAC\n"
    regs, sig, card, nk = O.sketch_buffer(fa, k=4, canon=False, S=8)
    assert nk == 1 + 1 + 5 + 0
    # canonical: a sequence and its reverse complement give identical sketches
    seq = b"ACGGTCATTACGGATCGGATTTACGCGAT"
    rc = seq[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA"))
    r1 = O.sketch_buffer(b">x\n" + seq + b"\n", k=7, canon=True, S=16)[0]
    r2 = O.sketch_buffer(b">x\n" + rc + b"\n", k=7, canon=True, S=16)[0]
    np.testing.assert_array_equal(r1, r2)
    r3 = O.sketch_buffer(b">x\n" + rc + b"\n", k=7, canon=False, S=16)[0]
    assert not np.array_equal(r1, r3)
    # FASTQ: quality lines starting with '@' or '>' must not start records
    fq = b"@r1\nACGTACGTAC\n+\n@@@@>>>>II\n@r2\nTTTTACGTAC\n+r2\nIIIIIIIIII\n"
    assert O.sketch_buffer(fq, k=5, canon=True, S=8)[3] == 6 + 6
    # multi-line FASTA == single-line FASTA
    a = O.sketch_buffer(b">s\nACGTAC\nGTTGCA\n", k=5, S=8)[0]
    b = O.sketch_buffer(b">s\nACGTACGTTGCA\n", k=5, S=8)[0]
    np.testing.assert_array_equal(a, b)


def test_oph_odd_sketchsize(oracle):
    # oph.h:143-146: odd S is rounded up to even m; only the first S doubles are kept (fastxsketch.cpp:605,610)
    regs, sig, card, nk = oracle.sketch_buffer(b">s\n" + b"ACGTTGCAAGCTTAGCTAGGATCGATCGATTAGC" * 20 + b"\n", k=9, S=15)
    assert regs.size == 16 and sig.size == 15


def test_densify(oracle):
    S = 64
    rng = np.random.default_rng(5)
    sig = rng.random(S)
    sig[rng.random(S) < 0.5] = 0.0
    out, ne = oracle.densify(sig)
    assert ne == int((sig == 0).sum())
    assert (out != 0).all()                      # cmp_core.cpp:611 post-condition
    nz = sig != 0
    np.testing.assert_array_equal(out[nz], sig[nz])
    assert set(out.tolist()) <= set(sig[nz].tolist())
    # all-empty sketch is returned unchanged (cmp_core.cpp:585-587)
    z, ne = oracle.densify(np.zeros(S))
    assert ne == S and (z == 0).all()
    # deterministic
    out2, _ = oracle.densify(sig)
    np.testing.assert_array_equal(out, out2)


def test_compare_epilogue_table(oracle):
    O = oracle
    S = 1024
    # SIMILARITY on power-of-two S is exactly neq/S
    for neq in [0, 1, 5, 512, 1023, 1024]:
        v = O.compare_from_gtlt(S - neq, 0, S, 1000., 2000., O.SIMILARITY, 31)
        assert v == np.float32(neq) / np.float32(S)
        v2 = O.compare_from_gtlt((S - neq) // 2, (S - neq) - (S - neq) // 2, S, 1000., 2000., O.SIMILARITY, 31)
        assert v2 == v
    # eq <= 0 -> 0, or +inf for POISSON_LLR (cmp_core.cpp:473-475)
    assert O.compare_from_gtlt(S, 0, S, 1., 1., O.SIMILARITY, 31) == 0.0
    assert O.compare_from_gtlt(S, 0, S, 1., 1., O.POISSON_LLR, 31) == np.inf
    # mash distance of identical sketches is 0; formula cmp_core.cpp:361
    assert O.compare_from_gtlt(0, 0, S, 1., 1., O.POISSON_LLR, 31) == 0.0
    sim = np.float32(512 / 1024)
    exp = np.float32(np.log(2. * float(sim) / (1. + float(sim))) * (-1. / 31))
    assert O.compare_from_gtlt(256, 256, S, 1., 1., O.POISSON_LLR, 31) == exp
    # union/intersection consistency
    isz = O.compare_from_gtlt(256, 256, S, 1000., 3000., O.INTERSECTION, 31)
    usz = O.compare_from_gtlt(256, 256, S, 1000., 3000., O.UNION_SIZE, 31)
    assert abs((1000. + 3000. - isz) - usz) < 1e-3 * usz
    assert abs(isz - 0.5 * 4000. / 1.5) < 1e-3 * isz
    c = O.compare_from_gtlt(256, 256, S, 1000., 3000., O.CONTAINMENT, 31)
    assert abs(c - isz / 3000.) < 1e-6
    sc = O.compare_from_gtlt(256, 256, S, 1000., 3000., O.SYMMETRIC_CONTAINMENT, 31)
    assert abs(sc - isz / 1000.) < 1e-6


def test_allpairs_matches_pairwise(oracle):
    rng = np.random.default_rng(11)
    N, S = 23, 64
    vals = rng.random((4, S))
    sigs = vals[rng.integers(0, 4, (N, S)), np.arange(S)[None, :]]
    cards = rng.random(N) * 1e5 + 1
    for meas in range(6):
        full = oracle.allpairs_ut(sigs, cards, measure=meas, k=21, nthreads=3)
        idx = 0
        lib = oracle.load()
        for i in range(N):
            for j in range(i + 1, N):
                gt, lt = oracle.count_gtlt(sigs[i], sigs[j])
                assert full[idx] == np.float32(oracle.compare_from_gtlt(gt, lt, S, cards[i], cards[j], meas, 21))
                idx += 1
        part = oracle.allpairs_ut(sigs, cards, measure=meas, k=21, nthreads=2, rows=(5, 11))
        off = sum(N - r - 1 for r in range(5))
        np.testing.assert_array_equal(part, full[off:off + part.size])


def test_text_formatter_vs_fmt_golden():
    """tests/golden/fmt_float.tsv was produced by fmt 12.1.0 `fmt::format("{}", float)` (exp_upper = 7);
    fmt10_float.tsv is the fmt < 11 layout (exp_upper = 16, the oracle's and the CLI's default)."""
    from oracle import textfmt
    import struct
    for table, eu in (("fmt_float.tsv", 7), ("fmt10_float.tsv", None)):
        n = 0
        for line in open(os.path.join(GOLDEN, table)):
            bits, exp = line.rstrip("\n").split("\t")
            v = struct.unpack("<f", struct.pack("<I", int(bits, 16)))[0]
            assert textfmt.fmt_float(v, eu) == exp, (table, bits, exp)
            n += 1
        assert n > 12000

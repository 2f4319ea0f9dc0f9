"""CPU tests of the --multiset oracle (oracle/d2_bmh_oracle.c): R11 exact k-mer counts and the
BMH-D2G restatement of BagMinHash (R12, parity unpinned against a real dashing2 binary because
sketch/bmh.h is absent from the reference).  What can be pinned is pinned here:
the deterministic log against libm, the estimator's defining properties, and order independence."""
import math

import numpy as np

from dashing2_amd import synth


def test_dlog_within_one_ulp_of_libm(oracle):
    rng = np.random.default_rng(1)
    us = np.concatenate([rng.random(20000), 2.0 ** -rng.integers(0, 54, 500).astype(np.float64),
                         [1.0, 2.0 ** -53, 0.5, 0.7071067811865476, 0.7071067811865475, 1 - 2.0 ** -53]])
    worst = 0.0
    for u in us:
        if u <= 0:
            continue
        a, b = oracle.dlog(u), math.log(u)
        if b == 0.0:
            assert a == 0.0
        else:
            worst = max(worst, abs(a - b) / math.ulp(b))
    assert worst <= 1.0, worst


def test_kmer_counts_match_python_dict(oracle):
    g = np.concatenate([synth.random_genome(3, 3000), np.tile(synth.random_genome(4, 50), 20)])
    buf = synth.fasta_bytes("x", g) + b">y\nACGTNNACGTACGTAC\n"
    for k, canon in ((5, True), (11, False), (21, True)):
        keys, counts, nk = oracle.kmer_count_buffer(buf, k, canon=canon)
        # independent restatement: python loop over the records
        comp = {65: 84, 67: 71, 71: 67, 84: 65}
        code = {65: 0, 67: 1, 71: 2, 84: 3}
        d = {}
        n = 0
        for rec in (bytes(g), b"ACGTNNACGTACGTAC"):
            for run in rec.split(b"N"):
                for i in range(len(run) - k + 1):
                    s = run[i:i + k]
                    f = 0
                    for c in s:
                        f = (f << 2) | code[c]
                    x = f
                    if canon:
                        r = 0
                        for c in reversed(s):
                            r = (r << 2) | code[comp[c]]
                        x = min(f, r)
                    key = oracle.wang_hash(x)          # maskfn with XORMASK = 0
                    d[key] = d.get(key, 0) + 1
                    n += 1
        assert nk == n
        ek = np.array(sorted(d), np.uint64)
        np.testing.assert_array_equal(keys, ek)
        np.testing.assert_array_equal(counts, np.array([d[int(x)] for x in ek], np.uint32))


def test_bmh_estimates_weighted_jaccard_and_is_order_free(oracle):
    rng = np.random.default_rng(2)
    S, n = 2048, 2500
    for ids in (rng.integers(0, 2 ** 63, n).astype(np.uint64), np.arange(n, dtype=np.uint64)):   # random and wsketch-style ids
        wa = rng.integers(1, 6, n).astype(np.float64)
        wb = wa.copy()
        sel = rng.random(n) < 0.5
        wb[sel] = rng.integers(1, 6, int(sel.sum()))
        wb[rng.random(n) < 0.1] = 0.0
        sa, ta = oracle.bmh_from_weighted(ids, wa, S)
        sb, tb = oracle.bmh_from_weighted(ids, wb, S)
        assert ta == wa.sum() and tb == wb.sum()
        J = np.minimum(wa, wb).sum() / np.maximum(wa, wb).sum()
        est = (sa == sb).mean()
        assert abs(est - J) < 4.5 * math.sqrt(J * (1 - J) / S), (est, J)
        perm = rng.permutation(n)
        sa2, _ = oracle.bmh_from_weighted(ids[perm], wa[perm], S)
        np.testing.assert_array_equal(sa2.view(np.uint64), sa.view(np.uint64))
        # registers are exponential with rate W/m: mean m/W
        assert abs(sa.mean() * wa.sum() / S - 1.0) < 5 / math.sqrt(S)


def test_bmh_consistent_in_the_weight(oracle):
    """the points of (d, w1) are a subset of the points of (d, w2 >= w1): registers only go down,
    and non-integer / tiny / huge weights are all inside the level set"""
    rng = np.random.default_rng(3)
    ids = rng.integers(0, 2 ** 63, 300).astype(np.uint64)
    w1 = rng.random(300) * 10.0 ** rng.integers(-6, 9, 300).astype(np.float64)
    w2 = w1 * (1.0 + rng.random(300))
    s1, _ = oracle.bmh_from_weighted(ids, w1, 128)
    s2, _ = oracle.bmh_from_weighted(ids, w2, 128)
    assert (s2 <= s1).all()
    # zero / negative weights are ignored; a set without weight stays at +inf
    s0, t0 = oracle.bmh_from_weighted(ids[:5], np.array([0.0, -1.0, 0.0, 0.0, -3.5]), 16)
    assert np.isinf(s0).all() and t0 == 0.0


def test_bmh_sketch_buffer_is_counts_then_update(oracle):
    g = np.concatenate([synth.random_genome(8, 5000), np.tile(synth.random_genome(9, 64), 10)])
    buf = synth.fasta_bytes("x", g)
    for thr in (0.0, 1.0):
        keys, counts, nk = oracle.kmer_count_buffer(buf, 13)
        keep = counts.astype(np.float64) > thr
        esig, etw = oracle.bmh_from_weighted(keys[keep], counts[keep].astype(np.float64), 200)
        sig, tw, nk2 = oracle.bmh_sketch_buffer(buf, 13, 200, count_threshold=thr)
        assert nk2 == nk and tw == etw == counts[keep].sum()
        np.testing.assert_array_equal(sig.view(np.uint64), esig.view(np.uint64))


def test_bmh_spec_known_answers(oracle):
    """the frozen BMH-D2G known answers (tests/golden/make_bmh_golden.py): any change of the spec shows here"""
    import os, sys
    from conftest import GOLDEN
    sys.path.insert(0, GOLDEN)
    import make_bmh_golden as G
    kat = np.load(os.path.join(GOLDEN, "bmh_kat.npz"))
    ids, w, seq_ids, fasta = G.inputs()
    for S in (64, 1000):
        sig, tw = oracle.bmh_from_weighted(ids, w, S)
        np.testing.assert_array_equal(sig.view(np.uint64), kat[f"weighted_S{S}"].view(np.uint64))
        assert tw == float(kat[f"weighted_tw_S{S}"])
        np.testing.assert_array_equal(oracle.bmh_from_weighted(seq_ids, None, S)[0].view(np.uint64), kat[f"unit_S{S}"].view(np.uint64))
    sig, tw, nk = oracle.bmh_sketch_buffer(fasta, 21, 256)
    np.testing.assert_array_equal(sig.view(np.uint64), kat["fasta_k21_S256"].view(np.uint64))
    assert tw == float(kat["fasta_tw"]) and nk == int(kat["fasta_nk"])
    sig, tw, _ = oracle.bmh_sketch_buffer(fasta, 11, 128, canon=False, count_threshold=1.0)
    np.testing.assert_array_equal(sig.view(np.uint64), kat["fasta_k11_S128_thr1"].view(np.uint64))
    assert tw == float(kat["fasta_tw_thr1"])
    np.testing.assert_array_equal(np.array([oracle.dlog(u) for u in kat["dlog_u"]]).view(np.uint64), kat["dlog"].view(np.uint64))

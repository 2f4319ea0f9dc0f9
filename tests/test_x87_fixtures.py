"""The x87 `long double` host rows (R5 getcard, R6 data(), R9 both compare() branches) checked against a THIRD,
independent restatement: tests/golden/x87_kat.npz, frozen from NumPy np.longdouble arithmetic by
tests/golden/make_x87_golden.py.  The product's host half (libd2g.so, d2g_host.cpp) and the oracle
(oracle/d2_oracle.c) are two C restatements by the same hand; a shared misreading of the reference would pass
a twin-vs-twin test, so both must reproduce these values bit for bit."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

Z = np.load(os.path.join(GOLDEN, "x87_kat.npz"))


def _sets():
    off = Z["regs_off"]
    return [Z["regs"][off[i]:off[i + 1]] for i in range(len(off) - 1)], [Z["sigs"][off[i]:off[i + 1]] for i in range(len(off) - 1)]


def test_fixture_generator_is_reproducible(tmp_path):
    """the committed fixture is what the committed script produces on this machine's x87 unit"""
    import subprocess
    import sys
    if np.finfo(np.longdouble).nmant != 63:
        pytest.skip("no 80-bit long double on this platform")
    src = os.path.join(GOLDEN, "make_x87_golden.py")
    code = open(src).read().replace('os.path.dirname(os.path.abspath(__file__))', repr(str(tmp_path)))
    p = tmp_path / "gen.py"
    p.write_text(code)
    subprocess.check_call([sys.executable, str(p)], stdout=subprocess.DEVNULL)
    new = np.load(tmp_path / "x87_kat.npz")
    for k in Z.files:
        assert np.array_equal(new[k].view(np.uint8), Z[k].view(np.uint8)), k


def test_getcard_and_data_product_and_oracle(d2g, oracle):
    import ctypes as C
    lib, olib = d2g.lib(), oracle.load()
    regsets, sigsets = _sets()
    for r, esig, ecard in zip(regsets, sigsets, Z["cards"]):
        r = np.ascontiguousarray(r)
        m = r.size
        # product: the single-sketch entry points of the C ABI
        got_card = lib.d2g_oph_card(r.ctypes.data_as(C.POINTER(C.c_uint64)), m)
        got_sig = np.empty(m, np.float64)
        assert lib.d2g_oph_signatures(r.ctypes.data_as(C.POINTER(C.c_uint64)), m, got_sig.ctypes.data_as(C.POINTER(C.c_double))) == 0
        assert np.float64(got_card).view(np.uint64) == np.float64(ecard).view(np.uint64)
        np.testing.assert_array_equal(got_sig.view(np.uint64), esig.view(np.uint64))
        # oracle
        osig, ocard = oracle.regs_finalize(r)
        assert np.float64(ocard).view(np.uint64) == np.float64(ecard).view(np.uint64)
        np.testing.assert_array_equal(osig.view(np.uint64), esig.view(np.uint64))
        # product: the batched form (even m only: it takes the sketch size and derives m)
        if m % 2 == 0:
            s2, c2 = d2g.oph_finalize(r.reshape(1, m), m)
            np.testing.assert_array_equal(s2[0].view(np.uint64), esig.view(np.uint64))
            assert c2[0].view(np.uint64) == np.float64(ecard).view(np.uint64)
    assert olib is not None


def test_compare_set_branch_product_and_oracle(d2g, oracle):
    rows, cards, exp = Z["set_in"], Z["set_cards"], Z["set_out"]
    assert np.isinf(exp.view(np.float32)).any() and (exp == 0).any()
    for (gt, lt, S, meas, k), (lhc, rhc), e in zip(rows, cards, exp):
        a = np.float32(d2g.epilogue_gtlt(int(gt), int(lt), int(S), float(lhc), float(rhc), int(meas), int(k))).view(np.uint32)
        b = np.float32(oracle.compare_from_gtlt(int(gt), int(lt), int(S), float(lhc), float(rhc), int(meas), int(k))).view(np.uint32)
        assert a == e and b == e, (gt, lt, S, meas, k, lhc, rhc, hex(int(a)), hex(int(b)), hex(int(e)))


def test_compare_count_eq_branch_product_and_oracle(d2g, oracle):
    rows, cards, exp = Z["neq_in"], Z["neq_cards"], Z["neq_out"]
    for (neq, S, meas, k), (lhc, rhc), e in zip(rows, cards, exp):
        a = np.float32(d2g.epilogue_neq(int(neq), int(S), float(lhc), float(rhc), int(meas), int(k))).view(np.uint32)
        b = np.float32(oracle.compare_from_neq(int(neq), int(S), float(lhc), float(rhc), int(meas), int(k))).view(np.uint32)
        assert a == e and b == e, (neq, S, meas, k, lhc, rhc, hex(int(a)), hex(int(b)), hex(int(e)))


def test_epilogue_table_matches_fixture(d2g):
    """the fused device epilogue is a table lookup lut[neq]; the table itself must hold the fixture's values
    (SIMILARITY / POISSON_LLR, power-of-two S in set space where value = f(neq); any S in multiset space)"""
    rows, exp = Z["neq_in"], Z["neq_out"]
    for S in (64, 1024, 2048, 100):
        for meas in (0, 3):
            for k in (31, 1, 0):
                lut = d2g.epilogue_lut(S, meas, k, multiset_space=True)
                sel = (rows[:, 1] == S) & (rows[:, 2] == meas) & (rows[:, 3] == k)
                for neq, e in zip(rows[sel, 0], exp[sel]):
                    assert lut[int(neq)].view(np.uint32) == e
    rows, exp = Z["set_in"], Z["set_out"]
    for S in (64, 1024, 2048):
        for meas in (0, 3):
            for k in (31, 21, 1, 0):
                lut = d2g.epilogue_lut(S, meas, k, multiset_space=False)
                sel = (rows[:, 2] == S) & (rows[:, 3] == meas) & (rows[:, 4] == k)
                assert sel.any() or k != 31
                for gt, lt, e in zip(rows[sel, 0], rows[sel, 1], exp[sel]):
                    assert lut[int(S - gt - lt)].view(np.uint32) == e

#!/usr/bin/env python3
"""Third, independent restatement of the reference's x87 `long double` host arithmetic -- in NumPy
(np.longdouble is the 80-bit x87 extended type on x86-64 Linux), written from the reference source and from
nothing else in this repository -- and the fixture it freezes: tests/golden/x87_kat.npz.

    python tests/golden/make_x87_golden.py          (run in the build container; the .npz is committed)

Both the product's host half (dashing2_amd/csrc/d2g_host.cpp) and the oracle (oracle/d2_oracle.c) must
reproduce every value bit for bit (tests/test_x87_fixtures.py).  Rows covered:
  R5  LazyOnePermSetSketch::getcard()   /root/reference/src/oph.h:240-247   (omul: oph.h:213-218)
  R6  LazyOnePermSetSketch::data()      /root/reference/src/oph.h:248-257
  R9  compare(), SPACE_SET branch       /root/reference/src/cmp_core.cpp:355-356,461-489,573-575
      compare(), count_eq branch        /root/reference/src/cmp_core.cpp:355-356,506-517,573-575
"""
import os
import sys

import numpy as np

LD = np.longdouble
assert np.finfo(LD).nmant == 63, "needs the x87 80-bit long double"
F32, F64 = np.float32, np.float64
U64MAX = (1 << 64) - 1
SIMILARITY, CONTAINMENT, SYMMETRIC_CONTAINMENT, POISSON_LLR, INTERSECTION, UNION_SIZE = range(6)


def ld_from_u64(x):
    """exact uint64 -> long double (64-bit significand): built from two exactly representable halves"""
    x = int(x)
    return LD(x >> 32) * LD(4294967296.0) + LD(x & 0xFFFFFFFF)


OMUL = LD(2.0) ** LD(-64)                      # oph.h:215  0x1p-64L


def getcard(regs):
    """oph.h:240-247: std::accumulate(regs, 0.L, x + y * omul) in index order; inf if the sum is 0;
    m_ * (m_ / sum) with m_ a size_t (converted exactly) -- returned as double"""
    m = len(regs)
    s = LD(0)
    for y in regs:
        s = s + ld_from_u64(y) * OMUL
    if s == 0:
        return F64(np.inf)
    return F64(LD(m) * (LD(m) / s))


def data(regs):
    """oph.h:248-257: mul = -SigT(1) / (m_ - count(max)) is a DOUBLE division (SigT = double, the divisor a
    size_t converted to double), widened to long double afterwards; per register: 0 if it is max or 0, else
    (double)(mul * logl(omul * (max - x + 1))) where max - x + 1 is evaluated in uint64"""
    m = len(regs)
    nmax = sum(1 for x in regs if int(x) == U64MAX)
    with np.errstate(divide="ignore"):
        mul = LD(F64(-1.0) / F64(m - nmax))
    out = np.zeros(m, F64)
    for i, x in enumerate(regs):
        x = int(x)
        if x == U64MAX or x == 0:
            out[i] = 0.0
        else:
            arg = OMUL * ld_from_u64((U64MAX - x + 1) & U64MAX)
            out[i] = F64(mul * np.log(arg))
    return out


def finish(ret):
    """cmp_core.cpp:573-575: NaN/Inf -> LDBL_MAX, then the long double is returned as LSHDistType = float"""
    ret = LD(ret)
    if np.isnan(ret) or np.isinf(ret):
        ret = np.finfo(LD).max
    with np.errstate(over="ignore"):
        return F32(ret)


def sim2dist_float(x, k):
    """cmp_core.cpp:356 with x = sim, a float: 2. * x / (1. + x) and std::log in DOUBLE"""
    pm = F64(-1.0) / F64(max(1, k))
    if x != 0:
        xd = F64(x)
        with np.errstate(divide="ignore"):
            return F64(np.log(F64(2.0) * xd / (F64(1.0) + xd)) * pm)
    return F64(np.inf)


def sim2dist_ld(x, k):
    """cmp_core.cpp:356 with x = ret, a long double: the arithmetic and the log are long double, the lambda
    returns double"""
    pm = LD(F64(-1.0) / F64(max(1, k)))
    if x != 0:
        return F64(np.log(LD(2.0) * x / (LD(1.0) + x)) * pm)
    return F64(np.inf)


def compare_set(gt, lt, S, lhc, rhc, measure, k):
    """cmp_core.cpp:461-489 (SPACE_SET, no truncation)"""
    invdenom = LD(1) / LD(S)                    # 355: 1.L / opts.sketchsize_
    lhcard, rhcard = LD(F64(lhc)), LD(F64(rhc))
    alpha = LD(gt) * invdenom                   # 463-464 (counts are far below 2^53: LD(int) is exact)
    beta = LD(lt) * invdenom
    eq = LD(1.0) - alpha - beta                 # 467
    with np.errstate(divide="ignore", invalid="ignore"):
        a = (lhcard + rhcard) / (LD(2) - alpha - beta)
    ucard = LD(0) if a < 0 else a               # 468 std::max(a, 0.L): (a < 0.L) ? 0.L : a  (NaN stays NaN)
    if eq <= 0:                                 # 473-475
        return F32(0.0) if measure != POISSON_LLR else F32(np.inf)     # (float)DBL_MAX == +inf
    if eq <= LD(F64(1e-15)):                    # 476-479: EPS is the double literal 1e-15 widened
        eq = LD(0)
    with np.errstate(over="ignore", invalid="ignore"):
        isz = F32(ucard * eq)                   # 480: const LSHDistType isz = ucard * eq, sim = eq
        sim = F32(eq)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        if measure == SIMILARITY:
            ret = LD(sim)
        elif measure == INTERSECTION:
            ret = LD(isz)
        elif measure == CONTAINMENT:
            ret = LD(isz) / rhcard
        elif measure == SYMMETRIC_CONTAINMENT:
            ret = LD(isz) / (rhcard if rhcard < lhcard else lhcard)    # std::min(lhcard, rhcard)
        elif measure == POISSON_LLR:
            ret = LD(sim2dist_float(sim, k))
        else:
            ret = lhcard + rhcard - LD(isz)
    return finish(ret)


def compare_neq(neq, S, lhc, rhc, measure, k):
    """cmp_core.cpp:506-517 (count_eq branch: multiset / BagMinHash space)"""
    invdenom = LD(1) / LD(S)
    lhcard, rhcard = LD(F64(lhc)), LD(F64(rhc))
    ret = invdenom * LD(neq)

    def uc():
        with np.errstate(divide="ignore", invalid="ignore"):
            a = (lhcard + rhcard) / (LD(1) + ret)
        return LD(0) if a < 0 else a

    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        if measure == INTERSECTION:
            ret = ret * uc()
        elif measure == SYMMETRIC_CONTAINMENT:
            ret = ret * (uc() / (rhcard if rhcard < lhcard else lhcard))
        elif measure == CONTAINMENT:
            ret = ret * (uc() / lhcard)
        elif measure == POISSON_LLR:
            ret = LD(sim2dist_ld(ret, k))
        elif measure == UNION_SIZE:
            isz = ret * uc()
            ret = lhcard + rhcard - isz
    return finish(ret)


def main():
    rng = np.random.default_rng(20260928)
    # ---- R5 / R6
    reg_sets = []
    for m in (2, 64, 100, 1024):
        r = rng.integers(0, 1 << 63, m, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, m, dtype=np.uint64)
        reg_sets.append(r)
        r2 = (r >> np.uint64(int(rng.integers(1, 40)))).copy()          # small registers (big sketches)
        r2[rng.integers(0, m, max(1, m // 8))] = np.uint64(U64MAX)      # empty buckets
        r2[rng.integers(0, m, max(1, m // 16))] = np.uint64(0)          # zero registers
        reg_sets.append(r2)
    reg_sets.append(np.full(16, U64MAX, np.uint64))                      # no k-mer at all: mul = -1/0
    reg_sets.append(np.zeros(8, np.uint64))                              # sum == 0 -> inf
    reg_sets.append(np.array([1, 2, U64MAX - 1, U64MAX, 1 << 63, (1 << 63) + 1, 3, 0], np.uint64))
    regs_flat = np.concatenate(reg_sets)
    regs_off = np.concatenate([[0], np.cumsum([len(r) for r in reg_sets])]).astype(np.int64)
    cards = np.array([getcard(r) for r in reg_sets], F64)
    sigs = np.concatenate([data(r) for r in reg_sets])
    # ---- R9, both branches
    rows_set, rows_neq = [], []
    card_choices = [(1e6, 2.5e6), (123456.789, 99.5), (1.0, 1.0), (0.0, 5.0), (np.inf, 3.0), (3e9, 4e9), (7.25, 0.0)]
    for S in (3, 7, 64, 100, 1000, 1024, 2048):
        pairs = {(0, 0), (S, 0), (0, S), (S // 2, S - S // 2), (1, 0), (0, 1), (S - 1, 0), (S // 3, S // 3), (S // 3, S - S // 3 - 1)}
        for _ in range(6):
            g = int(rng.integers(0, S + 1))
            pairs.add((g, int(rng.integers(0, S - g + 1))))
        if S == 3:
            pairs.add((1, 2))                                            # 1 - 1/3 - 2/3 in long double: tiny or <= 0
        if S == 7:
            pairs |= {(3, 4), (1, 6), (2, 5)}
        for gt, lt in sorted(pairs):
            for lhc, rhc in card_choices[:3] if S not in (3, 1024) else card_choices:
                for meas in range(6):
                    for k in ((31,) if meas != POISSON_LLR else (31, 21, 1, 0)):
                        rows_set.append((gt, lt, S, lhc, rhc, meas, k, compare_set(gt, lt, S, lhc, rhc, meas, k)))
        for neq in sorted({0, 1, S // 2, S - 1, S} | {int(x) for x in rng.integers(0, S + 1, 5)}):
            for lhc, rhc in card_choices[:3] if S not in (3, 1024) else card_choices:
                for meas in range(6):
                    for k in ((31,) if meas != POISSON_LLR else (31, 1, 0)):
                        rows_neq.append((neq, S, lhc, rhc, meas, k, compare_neq(neq, S, lhc, rhc, meas, k)))
    rs = np.array([(r[0], r[1], r[2], r[5], r[6]) for r in rows_set], np.int64)
    rs_c = np.array([(r[3], r[4]) for r in rows_set], F64)
    rs_o = np.array([r[7] for r in rows_set], F32)
    rn = np.array([(r[0], r[1], r[4], r[5]) for r in rows_neq], np.int64)
    rn_c = np.array([(r[2], r[3]) for r in rows_neq], F64)
    rn_o = np.array([r[6] for r in rows_neq], F32)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "x87_kat.npz")
    np.savez_compressed(out, regs=regs_flat, regs_off=regs_off, cards=cards, sigs=sigs,
                        set_in=rs, set_cards=rs_c, set_out=rs_o.view(np.uint32),
                        neq_in=rn, neq_cards=rn_c, neq_out=rn_o.view(np.uint32))
    print(f"{out}: {len(reg_sets)} register sets, {len(rows_set)} SPACE_SET compare rows, {len(rows_neq)} count_eq compare rows")
    print("  special outputs: inf %d, zero %d (set branch)" % (int(np.isinf(rs_o).sum()), int((rs_o == 0).sum())))


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/.  RUNS ONLY IN THE BUILD CONTAINER
(it imports the reference's python/parse.py from /root/reference, which cannot travel).

Fixtures are DATA (inputs + expected outputs):
  eqcount_*.npz   input signature matrices + the condensed equality counts computed by the
                  reference's own NumPy fallback  python/parse.py:128-156 pairwise_equality_compare
  stacked_*.bin   a stacked sketch file in the reference layout (F-b) + the arrays
                  python/parse.py:61-74 parse_binary_signatures reads back from it
  sketch_*.opss   a single cached sketch (F-a) + what python/parse.py:78-82 reads back
  distmat_*.bin   binary distance matrix (F-d) + python/parse.py:173-177 view
"""
import importlib.util
import io
import contextlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/python/parse.py"


def load_ref():
    spec = importlib.util.spec_from_file_location("d2_parse_ref", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.sys = sys   # parse.py uses sys.stderr without importing sys (python/parse.py:147)
    return mod


def main():
    ref = load_ref()
    rng = np.random.default_rng(20260928)
    # ---- equality-count fixtures -------------------------------------------------
    cases = {
        "eqcount_n7_s16": (7, 16, 3),
        "eqcount_n33_s64": (33, 64, 5),
        "eqcount_n64_s128": (64, 128, 9),
        "eqcount_n40_s100": (40, 100, 4),     # non power-of-two S
    }
    for name, (n, s, nvals) in cases.items():
        # few distinct values per column so that equalities are frequent; stored as doubles
        vals = rng.random((nvals, s))
        pick = rng.integers(0, nvals, size=(n, s))
        m = vals[pick, np.arange(s)[None, :]]
        m[rng.random((n, s)) < 0.1] = 0.0          # some empty registers
        with contextlib.redirect_stderr(io.StringIO()):
            exp = ref.pairwise_equality_compare(m)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), sigs=m, expected=np.asarray(exp, np.uint32))
    # ---- file-format fixtures ----------------------------------------------------
    n, s = 5, 32
    cards = rng.random(n) * 1e6
    sigs = rng.random((n, s))
    path = os.path.join(HERE, "stacked_n5_s32.bin")
    with open(path, "wb") as f:
        np.array([n, s], np.uint64).tofile(f)
        cards.tofile(f)
        sigs.tofile(f)
    parsed = ref.parse_binary_signatures(path)
    assert parsed.nseqs == n
    np.savez_compressed(os.path.join(HERE, "stacked_n5_s32.expected.npz"), nseqs=parsed.nseqs,
                        cardinalities=np.array(parsed.cardinalities), signatures=np.array(parsed.signatures))
    p1 = os.path.join(HERE, "sketch_s32.opss")
    with open(p1, "wb") as f:
        np.array([cards[0]]).tofile(f)
        sigs[0].tofile(f)
    one = ref.parse_binary_sketch(p1)
    np.savez_compressed(os.path.join(HERE, "sketch_s32.expected.npz"), cardinality=one["cardinality"],
                        signatures=np.array(one["signatures"]))
    # convert_sketches_to_packed_sketch (python/parse.py:85-99): N single sketches -> stacked layout
    singles = []
    for i in range(3):
        p = os.path.join(HERE, f"_tmp_{i}.opss")
        with open(p, "wb") as f:
            np.array([cards[i]]).tofile(f)
            sigs[i].tofile(f)
        singles.append(p)
    packed = os.path.join(HERE, "packed_from_singles.bin")
    ref.convert_sketches_to_packed_sketch(singles, packed)
    for p in singles:
        os.remove(p)
    dm = rng.random(n * (n - 1) // 2).astype(np.float32)
    pd = os.path.join(HERE, "distmat_n5.bin")
    dm.tofile(pd)
    np.savez_compressed(os.path.join(HERE, "distmat_n5.expected.npz"), values=np.array(ref.parse_binary_distmat(pd)))
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()

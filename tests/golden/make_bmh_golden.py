#!/usr/bin/env python3
"""Freezes known answers of the BMH-D2G spec under tests/golden/bmh_kat.npz.

The reference supplies no vectors for BagMinHash (its sketch/bmh.h is absent: parity unpinned), so
these are OUR oracle's outputs at the commit that fixed the spec; the CPU test
tests/test_oracle_bmh.py::test_bmh_spec_known_answers and the GPU test
tests/test_gpu_k3.py::test_bmh_golden_known_answers compare against them, which turns any silent
change of the level set, the generator, the seeding, the log or the strip structure into a failure.
Inputs are regenerated from fixed seeds by the tests (dashing2_amd.synth is deterministic)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O              # noqa: E402
from dashing2_amd import synth             # noqa: E402


def inputs():
    rng = np.random.default_rng(20260928)
    ids = rng.integers(0, 2 ** 63, 4000).astype(np.uint64)
    w = np.round(rng.random(4000) * 10.0 ** rng.integers(-2, 4, 4000).astype(np.float64), 6)
    w[::11] = 0.0
    seq_ids = np.arange(3000, dtype=np.uint64)
    fasta = synth.fasta_bytes("kat", np.concatenate([synth.random_genome(424242, 40000), np.tile(synth.random_genome(7, 200), 25)]))
    return ids, w, seq_ids, fasta


def main():
    ids, w, seq_ids, fasta = inputs()
    out = {}
    for S in (64, 1000):
        out[f"weighted_S{S}"], out[f"weighted_tw_S{S}"] = O.bmh_from_weighted(ids, w, S)
        out[f"unit_S{S}"], _ = O.bmh_from_weighted(seq_ids, None, S)
    sig, tw, nk = O.bmh_sketch_buffer(fasta, 21, 256)
    out["fasta_k21_S256"], out["fasta_tw"], out["fasta_nk"] = sig, tw, nk
    sig, tw, _ = O.bmh_sketch_buffer(fasta, 11, 128, canon=False, count_threshold=1.0)
    out["fasta_k11_S128_thr1"], out["fasta_tw_thr1"] = sig, tw
    out["dlog_u"] = np.array([1.0, 0.5, 0.75, 2.0 ** -53, 0.1, 0.7071067811865476, 0.9999999999999999, 1e-9])
    out["dlog"] = np.array([O.dlog(u) for u in out["dlog_u"]])
    np.savez_compressed(os.path.join(HERE, "bmh_kat.npz"), **{k: np.asarray(v) for k, v in out.items()})
    print("wrote bmh_kat.npz:", {k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    main()

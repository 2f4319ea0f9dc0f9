#!/usr/bin/env python3
"""Randomised parity campaign on a GPU box: product (C ABI) vs oracle on random parameters.
usage: fuzz_parity.py [seconds=120] [seed=1]   -- prints one line per failure and a summary"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dashing2_amd as D                       # noqa: E402
from dashing2_amd import synth                # noqa: E402
from oracle import oracle as O                # noqa: E402


def rand_fasta(rng, maxlen):
    recs = []
    for r in range(int(rng.integers(1, 6))):
        L = int(rng.integers(0, maxlen))
        g = synth.random_genome(int(rng.integers(0, 1 << 30)), max(L, 1))[:L].copy()
        if L > 50 and rng.random() < 0.5:
            for _ in range(int(rng.integers(1, 5))):
                a = int(rng.integers(0, L - 1))
                g[a:a + int(rng.integers(1, 60))] = ord("N")
        if L > 200 and rng.random() < 0.4:                         # repeats: counts > 1
            seg = g[:int(rng.integers(20, 200))]
            g = np.concatenate([g, np.tile(seg, int(rng.integers(2, 30)))])
        if rng.random() < 0.3:
            g = np.frombuffer(bytes(g).lower(), np.uint8)
        hdr = ">" if rng.random() < 0.8 else "@"
        if hdr == ">":
            recs.append(synth.fasta_bytes(f"r{r} c", g, width=int(rng.integers(10, 100))))
        else:
            recs.append(b"@q%d\n" % r + bytes(g) + b"\n+\n" + b"I" * len(g) + b"\n")
    return b"".join(recs)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    ctx = D.Context(0)
    t0 = time.time()
    n = {"k0": 0, "k1": 0, "k2": 0, "k3": 0, "byseq": 0, "wsets": 0, "mgpu": 0}
    sk0 = ctx.sketcher()
    fails = 0
    while time.time() - t0 < budget:
        kinds = ["k0", "k1", "k2", "k3", "k3", "byseq", "wsets", "mgpu"]
        if os.environ.get("D2G_FUZZ_ONLY"):                        # e.g. D2G_FUZZ_ONLY=k2: a campaign over one path
            kinds = os.environ["D2G_FUZZ_ONLY"].split(",")
        which = rng.choice(kinds)
        try:
            if which == "k0":
                # device FASTA parser (K0) -> K1 / K3 on the ingested stream, vs the oracle on the same bytes
                k = int(rng.integers(1, 33)); S = int(rng.choice([8, 64, 100, 1024]))
                fa = []
                for _ in range(int(rng.integers(1, 6))):
                    f = rand_fasta(rng, int(rng.choice([300, 5000, 120000])))
                    # what the device parser accepts: starts with '>', no line starts with '+' (FASTQ records are rewritten as FASTA)
                    f = f.replace(b"\n+\n", b"\n>q\n").replace(b"\n@", b"\n>")
                    f = (b">" + f[1:]) if f[:1] == b"@" else f
                    if rng.random() < 0.3:
                        f = f.replace(b"\n", b"\r\n")
                    fa.append(f)
                runs = sk0.ingest_fasta(fa, k)
                regs = sk0.run_ingested(runs, S)
                for i, f in enumerate(fa):
                    er, _, _, enk = O.sketch_buffer(f, k=k, S=S)
                    assert int(runs[3][i]) == enk and np.array_equal(regs[i], er)
                if k >= 3 and rng.random() < 0.4:
                    sig, tw = sk0.run_bmh_ingested(runs, 64)
                    for i, f in enumerate(fa):
                        es, et, _ = O.bmh_sketch_buffer(f, k, 64)
                        assert tw[i] == et and np.array_equal(sig[i].view(np.uint64), es.view(np.uint64))
            elif which == "k1":
                k = int(rng.integers(1, 33)); S = int(rng.choice([8, 63, 64, 100, 1000, 1024, 4096])); canon = bool(rng.integers(0, 2))
                xm = int(rng.choice([0, 0x724526e320f9967d, int(rng.integers(1, 1 << 62))]))
                fa = [rand_fasta(rng, int(rng.choice([300, 5000, 120000]))) for _ in range(int(rng.integers(1, 5)))]
                sp = D.SeqPack(k)
                for f in fa:
                    sp.add_fastx(f)
                regs = ctx.oph_sketch_seqpack(sp, S, canon=canon, xormask=xm)
                sig, card = D.oph_finalize(regs, S)
                for i, f in enumerate(fa):
                    er, es, ec, _ = O.sketch_buffer(f, k=k, canon=canon, xormask=xm, S=S)
                    assert np.array_equal(regs[i], er) and np.array_equal(sig[i].view(np.uint64), es.view(np.uint64)) and card[i] == ec
            elif which == "k2":
                N = int(rng.integers(2, 700)) if rng.random() < 0.8 else int(rng.integers(700, 2600))
                S = int(rng.choice([32, 64, 100, 128, 1000, 1024])); meas = int(rng.integers(0, 6))
                os.environ["D2G_BS_SORT"] = str(int(rng.integers(0, 2)))                    # column plan on / off
                # sparse tiles + pair list (rounds 5-6): forced on for these small matrices half of the time -- families found or not
                # (D2G_SP_LINK=0: the pair list alone), any tile budget, a short list (overflow -> dense walk), the list applied entry by
                # entry or binned + composed, with or without the first look at the matrix; the other half takes the default (dense walk
                # below 8192 sketches)
                for var in ("D2G_SP_RIDE", "D2G_BS_SPARSE_MIN_N", "D2G_SP_LINK", "D2G_SP_TILE_FRAC", "D2G_SP_LIST_DIV", "D2G_SP_OLINK", "D2G_SP_REMEMBER", "D2G_SP_EMIT_BIG",
                            "D2G_SP_LIST_FORM", "D2G_SP_PREDICT"):
                    os.environ.pop(var, None)
                if rng.random() < 0.5:
                    os.environ["D2G_BS_SPARSE_MIN_N"] = "1"
                    if rng.random() < 0.2:
                        os.environ["D2G_SP_LINK"] = "0"
                    os.environ["D2G_SP_LIST_FORM"] = str(int(rng.choice([0, 1, 2])))              # the list's two forms (0: the last list's length decides)
                    if rng.random() < 0.5:
                        os.environ["D2G_SP_PREDICT"] = "0"                                    # no sample before a set's first ordering
                    os.environ["D2G_SP_TILE_FRAC"] = str(rng.choice([0.05, 0.35, 1.0]))
                    if rng.random() < 0.2:
                        os.environ["D2G_SP_LIST_DIV"] = str(int(rng.choice([1, 64, 4096])))
                    if rng.random() < 0.3:
                        os.environ["D2G_SP_OLINK"] = "0"                                      # the link passes in their table form
                    if rng.random() < 0.3:
                        os.environ["D2G_SP_EMIT_BIG"] = "1"                                   # the pair-list kernel's form for N >= 65 536
                    os.environ["D2G_SP_REMEMBER"] = "0"                                       # one-shot sets: every prepare decides afresh
                ctx.reload_tuning()
                r = rng.random()
                if r < 0.25:
                    regs = synth.skewed_registers(N, S, seed=int(rng.integers(0, 1 << 30)), max_shared=int(rng.choice([4, 64, 300])))
                elif r < 0.35:
                    regs = synth.paired_registers(N - (N & 1), S, seed=int(rng.integers(0, 1 << 30)))[:N]
                    N = regs.shape[0]
                elif r < 0.42:
                    regs = synth.unrelated_registers(N, S, seed=int(rng.integers(0, 1 << 30)))
                else:
                    regs = synth.synthetic_registers(N, S, nclusters=int(rng.integers(1, 12)), seed=int(rng.integers(0, 1 << 30)))
                    if rng.random() < 0.5:                                                   # families + chance collisions with strangers
                        regs = synth.add_chance_collisions(regs, int(rng.choice([1, 3, 10, 40])), seed=int(rng.integers(0, 1 << 30)))
                sig, card = D.oph_finalize(regs, S)
                multiset = bool(rng.integers(0, 2))
                got = ctx.cmp_dist_ut(sig.view(np.uint64), card, measure=meas, k=int(rng.integers(1, 33)) if False else 31,
                                      multiset_space=multiset, algo=int(rng.choice([D.CMP_AUTO, D.CMP_DIRECT, D.CMP_BITSLICE])))
                if N > 3 and rng.random() < 0.5:                                             # a row range of the triangle (a shard / a CLI batch)
                    a = int(rng.integers(0, N - 1)); b = int(rng.integers(a + 1, N + 1))
                    part = ctx.cmp_dist_ut(sig.view(np.uint64), card, measure=meas, k=31, multiset_space=multiset, r0=a, r1=b, algo=D.CMP_BITSLICE if (multiset or S & (S - 1) == 0) else D.CMP_AUTO)
                    o0 = D.ut_count(N, 0, a)
                    assert np.array_equal(part.view(np.uint32), got[o0:o0 + part.size].view(np.uint32))
                if multiset:
                    neq = O.eqcounts_ut(sig)
                    iu = np.triu_indices(N, 1)
                    exp = np.array([O.compare_from_neq(int(c), S, card[i], card[j], meas, 31) for c, i, j in zip(neq, iu[0], iu[1])], np.float32)
                else:
                    exp = O.allpairs_ut(sig, card, measure=meas, k=31, nthreads=4)
                assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
                if os.environ.get("D2G_BS_SPARSE_MIN_N") == "1" and N > 3 and rng.random() < 0.6:
                    # the announced output (d2g_cmp_ut_announce_dev): any subset of the prepare's small kernels carries the fill, the
                    # launch finishes it; a row range, a 4-byte aligned output, counts against the oracle; then a launch nobody announced
                    os.environ["D2G_SP_RIDE"] = str(int(rng.integers(0, 64)))
                    ctx.reload_tuning()
                    bits = np.ascontiguousarray(sig.view(np.uint64))
                    neq = O.eqcounts_ut(sig)
                    a = int(rng.integers(0, N - 1)); b = int(rng.integers(a + 1, N + 1))
                    o0, cnt = D.ut_count(N, 0, a), D.ut_count(N, a, b)
                    off = 4 * int(rng.integers(0, 4))
                    d_sig, d_out = ctx.malloc(bits.nbytes), ctx.malloc(4 * cnt + 32)
                    try:
                        ctx.h2d(d_sig, bits)
                        cs = ctx.cmp_set_dev(d_sig, N, S, algo=D.CMP_BITSLICE)
                        for announced in (True, False, True):
                            ctx.h2d(d_out, np.full(cnt + 8, 0xDEADBEEF, np.uint32))
                            if announced:
                                cs.announce_ut_dev(d_out + off, a, b)
                            cs.update_dev(d_sig)
                            cs.eqcount_ut_dev(d_out + off, a, b)
                            back = np.empty(cnt + 8, np.uint32)
                            ctx.d2h(back, d_out)
                            q = off // 4
                            assert np.array_equal(back[q:q + cnt], neq[o0:o0 + cnt]), "announced output"
                            assert np.all(back[:q] == 0xDEADBEEF) and np.all(back[q + cnt:] == 0xDEADBEEF), "announced output: wrote outside"
                        cs.close()
                    finally:
                        ctx.free(d_sig); ctx.free(d_out)
                    os.environ.pop("D2G_SP_RIDE", None)
            elif which == "k3":
                k = int(rng.integers(3, 33)); S = int(rng.choice([16, 100, 256, 2048])); canon = bool(rng.integers(0, 2))
                thr = float(rng.choice([0, 0, 1, 3]))
                # library test hooks: small table rounds / split thresholds push small inputs through the big-input paths;
                # D2G_K3_COMPACT selects the compact (4-byte, low-traffic) key path where it applies (k <= 21)
                for var, choices in (("D2G_K3_ROUND_KEYS", [None, None, "40", "300"]), ("D2G_K3_SPLIT_MIN", [None, None, "20", "400"]),
                                     ("D2G_K3_COMPACT", [None, "1", "1"]), ("D2G_K3_GUESS_SCALE", [None, None, None, "0.02"]),
                                     ("D2G_K3_L1BITS", [None, None, "0", "2", "5"]), ("D2G_K3_LIGHT", [None, None, None, "0"]),
                                     ("D2G_K3_GQ_SCALE", [None, None, None, "0.1"]), ("D2G_K3_SUBBATCH", [None, "2", "3", "5"])):
                    v = choices[int(rng.integers(0, len(choices)))]
                    if v is None:
                        os.environ.pop(var, None)
                    else:
                        os.environ[var] = v
                ctx.reload_tuning()
                fa = [rand_fasta(rng, int(rng.choice([300, 5000, 80000, 400000]))) for _ in range(int(rng.integers(1, 6)))]
                sp = D.SeqPack(k)
                for f in fa:
                    sp.add_fastx(f)
                sig, tw = ctx.bmh_sketch_seqpack(sp, S, canon=canon, count_threshold=thr)
                kc = ctx.kmer_count_seqpack(sp, canon=canon, count_threshold=thr)
                for i, f in enumerate(fa):
                    es, et, _ = O.bmh_sketch_buffer(f, k, S, canon=canon, count_threshold=thr)
                    ek, ec, _ = O.kmer_count_buffer(f, k, canon=canon)
                    keep = ec.astype(np.float64) > thr
                    assert tw[i] == et and np.array_equal(sig[i].view(np.uint64), es.view(np.uint64))
                    assert np.array_equal(kc[i][0], ek[keep]) and np.array_equal(kc[i][1], ec[keep])
            elif which == "wsets":
                S = int(rng.choice([8, 64, 200])); nsets = int(rng.integers(1, 6))
                sizes = [int(rng.choice([0, 1, 50, 3000])) for _ in range(nsets)]
                off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
                ids = rng.integers(0, 1 << 60, int(off[-1]), dtype=np.uint64)
                w = None if rng.random() < 0.3 else np.where(rng.random(ids.size) < 0.1, 0.0, np.round(rng.gamma(1.5, 6.0, ids.size) * rng.choice([1, 1000])) / 4)
                sig, tw, own = ctx.bmh_from_weighted_ids(ids, w, off, S)
                for i in range(nsets):
                    lo, hi = int(off[i]), int(off[i + 1])
                    es, et, eo = O.bmh_from_weighted_ids(ids[lo:hi], None if w is None else w[lo:hi], S)
                    assert tw[i] == et and np.array_equal(sig[i].view(np.uint64), es.view(np.uint64)) and np.array_equal(own[i], eo)
            elif which == "mgpu":
                W = int(rng.integers(1, 7)); N = int(rng.integers(2, 400)); S = int(rng.choice([32, 100, 256, 1000, 1024]))
                os.environ["D2G_MGPU_CHUNKS"] = str(int(rng.integers(1, 5)))                # chunked exchange inside the step
                regs = synth.synthetic_registers(N, S, nclusters=int(rng.integers(1, 9)), seed=int(rng.integers(0, 1 << 30)))
                exp = O.eqcounts_ut(regs.view(np.float64))
                ctxs = [D.Context(0) for _ in range(W)]
                comms = D.Comm.create_all(ctxs)
                engs = [D.AllPairs(ctxs[r], comms[r], N, S) for r in range(W)]
                rows, outs = [], []
                for r in range(W):
                    lo, hi = engs[r].rows_held
                    p = ctxs[r].malloc(max((hi - lo) * S * 8, 8))
                    if hi > lo:
                        ctxs[r].h2d(p, np.ascontiguousarray(regs[lo:hi]))
                    rows.append(p)
                    outs.append(ctxs[r].malloc(max(D.ut_count(N, *engs[r].rows_computed), 1) * 4))
                D.allpairs_step_all(engs, rows, None, outs)
                offp = np.concatenate([[0], np.cumsum(N - 1 - np.arange(N, dtype=np.int64))])
                ok = True
                for r in range(W):
                    r0, r1 = engs[r].rows_computed
                    got = np.empty(D.ut_count(N, r0, r1), np.uint32)
                    ctxs[r].sync()
                    if got.size:
                        ctxs[r].d2h(got, outs[r])
                    ok &= np.array_equal(got, exp[offp[r0]:offp[r1]])
                    ctxs[r].free(rows[r]); ctxs[r].free(outs[r])
                for e in engs:
                    e.close()
                for c in comms:
                    c.close()
                for c in ctxs:
                    c.close()
                assert ok
            else:
                k = int(rng.integers(3, 33)); S = int(rng.choice([16, 64, 256]))
                f = b"".join(rand_fasta(rng, 2000) for _ in range(int(rng.integers(1, 8))))
                sp = D.SeqPack(k)
                sp.add_fastx_by_record(f)
                names, es, ec = O.sketch_buffer_byseq(f, k, S)
                assert [sp.name(i) for i in range(sp.ngenomes)] == names
                regs = ctx.oph_sketch_seqpack(sp, S)
                sig, card = D.oph_finalize(regs, S)
                nd = ctx.kmer_distinct_seqpack(sp)
                card = np.where(np.isnan(card), 0.0, card)
                card = np.where(card < 10.0 * S, nd.astype(np.float64), card)
                assert np.array_equal(sig.view(np.uint64), es.view(np.uint64)) and np.array_equal(card, ec)
            n[which] += 1
        except AssertionError:
            fails += 1
            print(f"FAIL {which} (case #{sum(n.values()) + fails}, seed {seed})", flush=True)
    print(f"fuzz: {n} cases passed, {fails} failed in {time.time() - t0:.0f}s")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()

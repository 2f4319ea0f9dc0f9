"""GPU parity tests for K3, the --multiset sketch path: exact k-mer counting (R11) and BagMinHash
(R12) through the C ABI versus the oracle.  Counts are integers and must match exactly; the
BagMinHash registers are doubles produced by the same IEEE operation sequence on both sides
(own log, no FMA contraction) and must match BIT FOR BIT even though the GPU explores the Poisson
process tree depth-first against a stale bound while the oracle uses a time-ordered heap."""
import numpy as np
import pytest

from dashing2_amd import synth

pytestmark = pytest.mark.gpu


def _genomes(rng):
    base = synth.random_genome(11, 60000)
    rep = np.tile(synth.random_genome(12, 500), 40)                     # 40 tandem copies: counts up to 40
    low = np.frombuffer(b"ACGT" * 3000, np.uint8)                       # 4 distinct k-mers, huge counts
    polya = np.frombuffer(b"A" * 5000, np.uint8)                        # ONE distinct k-mer
    return [
        synth.fasta_bytes("g0", base),
        synth.fasta_bytes("g1", synth.mutate(base, 0.02, 5)) + synth.fasta_bytes("g1b", rep),
        synth.fasta_bytes("rep", rep),
        synth.fasta_bytes("low", low),
        synth.fasta_bytes("polyA", polya),
        b">tiny\nACGTTGCA\n",
        b"",
        synth.fasta_bytes("big", synth.random_genome(13, 400000)),       # several buckets, > 1 workgroup of chunks
        synth.fasta_bytes("n", synth.random_genome(14, 3000)) + b">x\nNNNNACGTNNNN\n",
    ]


@pytest.mark.parametrize("k,canon,xormask,thr", [(21, True, 0, 0.0), (31, True, 0, 0.0), (11, False, 0x1234, 0.0),
                                                  (7, True, 0, 2.0), (32, True, 0, 0.0)])
def test_kmer_counts_exact(gpu_ctx, d2g, oracle, k, canon, xormask, thr):
    rng = np.random.default_rng(k)
    genomes = _genomes(rng)
    sp = d2g.SeqPack(k)
    for g in genomes:
        sp.add_fastx(g)
    got = gpu_ctx.kmer_count_seqpack(sp, canon=canon, xormask=xormask, count_threshold=thr)
    assert len(got) == len(genomes)
    for gi, g in enumerate(genomes):
        ek, ec, enk = oracle.kmer_count_buffer(g, k, canon=canon, xormask=xormask)
        keep = ec.astype(np.float64) > thr
        assert sp.nkmers(gi) == enk
        np.testing.assert_array_equal(got[gi][0], ek[keep], err_msg=f"genome {gi} keys")
        np.testing.assert_array_equal(got[gi][1], ec[keep], err_msg=f"genome {gi} counts")


@pytest.mark.parametrize("k,S,canon,thr", [(21, 2048, True, 0.0),       # BASELINE config 5 shape
                                           (31, 1024, True, 0.0),
                                           (15, 100, False, 0.0),       # non power-of-two register count
                                           (9, 64, True, 1.0)])         # count threshold
def test_bmh_sketch_bit_exact(gpu_ctx, d2g, oracle, k, S, canon, thr):
    rng = np.random.default_rng(S)
    genomes = _genomes(rng)
    sp = d2g.SeqPack(k)
    for g in genomes:
        sp.add_fastx(g)
    sig, tw = gpu_ctx.bmh_sketch_seqpack(sp, S, canon=canon, count_threshold=thr)
    assert sig.shape == (len(genomes), S)
    for gi, g in enumerate(genomes):
        esig, etw, enk = oracle.bmh_sketch_buffer(g, k, S, canon=canon, count_threshold=thr)
        assert tw[gi] == etw, f"genome {gi} total weight"
        np.testing.assert_array_equal(sig[gi].view(np.uint64), esig.view(np.uint64), err_msg=f"genome {gi}")
    # the persistent sketcher gives the same answer and can be reused
    sk = gpu_ctx.sketcher()
    for _ in range(2):
        sig2, tw2 = sk.run_bmh(sp, S, canon=canon, count_threshold=thr)
        np.testing.assert_array_equal(sig2.view(np.uint64), sig.view(np.uint64))
        np.testing.assert_array_equal(tw2, tw)
    sk.close()


def test_bmh_weighted_jaccard_property(gpu_ctx, d2g):
    """size-independent property at BASELINE config-5 scale (one 5 Mbp genome, k=21, S=2048):
    P[register equal] = weighted Jaccard of the k-mer count vectors"""
    base = synth.random_genome(21, 5_000_000)
    mut = synth.mutate(base, 0.01, 22)
    sp = d2g.SeqPack(21)
    sp.add_fastx(synth.fasta_bytes("a", base))
    sp.add_fastx(synth.fasta_bytes("b", mut))
    sp.add_fastx(synth.fasta_bytes("a2", base) + synth.fasta_bytes("a3", base))     # every count doubled
    S = 2048
    sig, tw = gpu_ctx.bmh_sketch_seqpack(sp, S)
    assert np.isfinite(sig).all() and (sig > 0).all()
    nk = 5_000_000 - 20
    assert tw[0] == nk and tw[1] == nk and tw[2] == 2 * nk
    # expected Jaccard of k-mer sets under 1% substitutions: each k-mer survives with (0.99)^21
    p = 0.99 ** 21
    j_ab = p / (2 - p)
    est = (sig[0] == sig[1]).mean()
    assert abs(est - j_ab) < 5 * np.sqrt(j_ab * (1 - j_ab) / S), (est, j_ab)
    # doubling every weight: weighted Jaccard = 1/2
    est2 = (sig[0] == sig[2]).mean()
    assert abs(est2 - 0.5) < 5 * np.sqrt(0.25 / S), est2
    # consistency in the weight: more weight can only lower a register
    assert (sig[2] <= sig[0]).all()


def test_bmh_from_weighted_matches_oracle(gpu_ctx, d2g, oracle):
    rng = np.random.default_rng(5)
    sets = []
    for n in (0, 1, 7, 1024, 1025, 5000, 20000):
        ids = rng.integers(0, 2 ** 63, n).astype(np.uint64) if n != 5000 else np.arange(n, dtype=np.uint64)
        w = rng.random(n) * 10 ** rng.integers(-3, 6, n).astype(np.float64)
        if n > 10:
            w[::7] = 0.0
            w[3] = -1.0
        sets.append((ids, w))
    ids = np.concatenate([s[0] for s in sets])
    w = np.concatenate([s[1] for s in sets])
    off = np.cumsum([0] + [len(s[0]) for s in sets]).astype(np.uint64)
    for S in (64, 1000):
        sig, tw = gpu_ctx.bmh_from_weighted(ids, w, off, S)
        for i, (si, wi) in enumerate(sets):
            esig, etw = oracle.bmh_from_weighted(si, wi, S)
            np.testing.assert_array_equal(sig[i].view(np.uint64), esig.view(np.uint64), err_msg=f"set {i} S={S}")
            assert abs(tw[i] - etw) <= 1e-9 * max(1.0, etw)      # double sums in a different order
    # unit weights
    sig, tw = gpu_ctx.bmh_from_weighted(sets[5][0], None, np.array([0, 5000], np.uint64), 256)
    esig, etw = oracle.bmh_from_weighted(sets[5][0], None, 256)
    np.testing.assert_array_equal(sig[0].view(np.uint64), esig.view(np.uint64))
    assert tw[0] == etw == 5000.0
    with pytest.raises(d2g.D2GError):
        gpu_ctx.bmh_from_weighted(np.array([1], np.uint64), np.array([2.0 ** 60]), np.array([0, 1], np.uint64), 16)


def test_k3_multi_round_buckets_and_redo_paths(d2g, oracle, tmp_path):
    """fresh processes with the test hooks: D2G_K3_ROUND_KEYS=64 makes every bucket need several
    table rounds (the path genomes above ~5.7 Mbp take), D2G_K3_GUESS_SCALE=1e-3 makes the guessed
    pruning bound fail its verification so the main pass is repeated under a larger bound,
    D2G_K3_SPLIT_MIN lowers the bucket size above which buckets are split once more by their low key
    bits (the path inputs above ~23 Mbp take), D2G_K3_L1BITS the number of bucket bits the scatter resolves itself before
    k3_refine_kernel takes over (default 8: genomes above ~260 kbp).  Counts and BagMinHash registers must not change."""
    import os, subprocess, sys, json
    g = synth.fasta_bytes("a", synth.random_genome(31, 150000)) + synth.fasta_bytes("r", np.tile(synth.random_genome(32, 300), 30))
    fa = tmp_path / "x.fa"
    fa.write_bytes(g)
    esig, etw, _ = oracle.bmh_sketch_buffer(g, 17, 128)
    ek, ec, _ = oracle.kmer_count_buffer(g, 17)
    code = (
        "import sys, json, numpy as np; sys.path.insert(0, %r); import dashing2_amd as D\n"
        "ctx = D.Context(0); sp = D.SeqPack(17); sp.add_path(%r)\n"
        "sig, tw = ctx.bmh_sketch_seqpack(sp, 128); kc = ctx.kmer_count_seqpack(sp)\n"
        "print(json.dumps([sig.view(np.uint64).tolist(), tw.tolist(), kc[0][0].tolist(), kc[0][1].tolist()]))\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(fa))
    # D2G_K3_COMPACT=1 selects the low-traffic path (4-byte stored words, tile-sorted split; k <= 21) instead of the default
    # 64-bit-key path, so both are exercised with every hook
    for env in ({"D2G_K3_ROUND_KEYS": "64"}, {"D2G_K3_GUESS_SCALE": "0.001"}, {"D2G_K3_ROUND_KEYS": "100", "D2G_K3_GUESS_SCALE": "0.01"},
                {"D2G_K3_SPLIT_MIN": "100"},                                   # big-input path: buckets pre-split by low key bits
                {"D2G_K3_SPLIT_MIN": "40", "D2G_K3_ROUND_KEYS": "200"},        # ... and sub-ranges that still need rounds
                {"D2G_K3_L1BITS": "3"}, {"D2G_K3_L1BITS": "0"},                # two-level scatter (genomes above ~260 kbp): 8 / 1 write fronts, then k3_refine_kernel
                {"D2G_K3_L1BITS": "12"},                                       # ... and never (the single-level scatter)
                {"D2G_K3_SUBBATCH": "2"},                                      # sub-batch pipeline: bucketing of range 2 under the counting of range 1 (two streams)
                {"D2G_K3_SUBBATCH": "2", "D2G_K3_L1BITS": "3"}, {"D2G_K3_SUBBATCH": "2", "D2G_K3_GQ_SCALE": "0.05"},
                {"D2G_K3_SUBBATCH": "2", "D2G_K3_GUESS_SCALE": "0.001"}, {"D2G_K3_SUBBATCH": "1"},
                {"D2G_K3_LIGHT": "0"},                                         # first pass in the heavy form (survivors walked in the counting kernel)
                {"D2G_K3_GQ_SCALE": "0.05"},                                   # survivor regions far too small: overflow -> the pass is repeated in the heavy form
                {"D2G_K3_GQ_SCALE": "0.3", "D2G_K3_GUESS_SCALE": "0.01"},      # ... and together with a failed bound guess
                {"D2G_K3_L1BITS": "2", "D2G_K3_SPLIT_MIN": "40", "D2G_K3_ROUND_KEYS": "200"},
                {"D2G_K3_COMPACT": "1"}, {"D2G_K3_COMPACT": "1", "D2G_K3_ROUND_KEYS": "64"}, {"D2G_K3_COMPACT": "1", "D2G_K3_GUESS_SCALE": "0.001"},
                {"D2G_K3_COMPACT": "1", "D2G_K3_SPLIT_MIN": "40", "D2G_K3_ROUND_KEYS": "200"},
                {"D2G_K3_COMPACT": "1", "D2G_K3_SPLIT_MIN": "100000"}):        # compact path without the sub-range split: table rounds
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, env={**os.environ, **env}, timeout=300)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        sig, tw, keys, counts = json.loads(r.stdout.decode().strip().splitlines()[-1])
        assert tw[0] == etw, env
        np.testing.assert_array_equal(np.array(sig[0], np.uint64), esig.view(np.uint64), err_msg=str(env))
        np.testing.assert_array_equal(np.array(keys, np.uint64), ek, err_msg=str(env))
        np.testing.assert_array_equal(np.array(counts, np.uint32), ec, err_msg=str(env))


@pytest.mark.parametrize("subbatch", [None, "3", "8"])
def test_k3_many_small_and_one_large_input(gpu_ctx, d2g, oracle, monkeypatch, subbatch):
    """batch shapes: 300 read-sized inputs (one bucket each, many workgroup-less genomes) next to a
    12 Mbp genome (4096 buckets of ~2900 keys: 4 table rounds each); D2G_K3_SUBBATCH cuts the batch into that many
    genome ranges for the two-stream pipeline (ranges of very different weight, empty inputs at their borders)"""
    if subbatch:
        monkeypatch.setenv("D2G_K3_SUBBATCH", subbatch)
    rng = np.random.default_rng(9)
    sp = d2g.SeqPack(21)
    small = [synth.fasta_bytes(f"s{i}", synth.random_genome(1000 + i, int(rng.integers(30, 3000)))) for i in range(300)]
    for f in small:
        sp.add_fastx(f)
    big = synth.random_genome(77, 12_000_000)
    sp.add_fastx(synth.fasta_bytes("big", big))
    sp.add_fastx(synth.fasta_bytes("big2", big) + synth.fasta_bytes("big3", big[:6_000_000]))
    S = 512
    sig, tw = gpu_ctx.bmh_sketch_seqpack(sp, S)
    for i in (0, 17, 150, 299):
        esig, etw, _ = oracle.bmh_sketch_buffer(small[i], 21, S)
        assert tw[i] == etw
        np.testing.assert_array_equal(sig[i].view(np.uint64), esig.view(np.uint64), err_msg=f"small {i}")
    nk = 12_000_000 - 20
    assert tw[300] == nk and tw[301] == nk + 6_000_000 - 20
    assert np.isfinite(sig[300:]).all()
    # big2 = big + the first half again: weighted Jaccard = |big| / (|big| + |half|) = 2/3
    est = (sig[300] == sig[301]).mean()
    assert abs(est - 2 / 3) < 5 * np.sqrt((2 / 9) / S), est
    assert (sig[301] <= sig[300]).all()


def test_bmh_golden_known_answers(gpu_ctx, d2g):
    """GPU vs the frozen BMH-D2G known answers (no oracle in the loop): tests/golden/bmh_kat.npz"""
    import os, sys
    from conftest import GOLDEN
    sys.path.insert(0, GOLDEN)
    import make_bmh_golden as G
    kat = np.load(os.path.join(GOLDEN, "bmh_kat.npz"))
    ids, w, seq_ids, fasta = G.inputs()
    for S in (64, 1000):
        sig, tw = gpu_ctx.bmh_from_weighted(np.concatenate([ids, seq_ids]), np.concatenate([w, np.ones(seq_ids.size)]),
                                            np.array([0, ids.size, ids.size + seq_ids.size], np.uint64), S)
        np.testing.assert_array_equal(sig[0].view(np.uint64), kat[f"weighted_S{S}"].view(np.uint64))
        np.testing.assert_array_equal(sig[1].view(np.uint64), kat[f"unit_S{S}"].view(np.uint64))
        assert tw[0] == float(kat[f"weighted_tw_S{S}"]) and tw[1] == float(seq_ids.size)
    sp = d2g.SeqPack(21)
    sp.add_fastx(fasta)
    sig, tw = gpu_ctx.bmh_sketch_seqpack(sp, 256)
    np.testing.assert_array_equal(sig[0].view(np.uint64), kat["fasta_k21_S256"].view(np.uint64))
    assert tw[0] == float(kat["fasta_tw"]) and sp.nkmers(0) == int(kat["fasta_nk"])
    sp = d2g.SeqPack(11)
    sp.add_fastx(fasta)
    sig, tw = gpu_ctx.bmh_sketch_seqpack(sp, 128, canon=False, count_threshold=1.0)
    np.testing.assert_array_equal(sig[0].view(np.uint64), kat["fasta_k11_S128_thr1"].view(np.uint64))
    assert tw[0] == float(kat["fasta_tw_thr1"])


def test_bmh_from_weighted_owner_ids(gpu_ctx, oracle):
    """BagMinHash2::ids() for explicit weighted sets (wsketch.cpp:36-37,66-67): the position of the element that
    owns each register, several sets per call incl. an empty one and one with duplicate ids (exact ties go to
    the smaller position), must equal the oracle's; registers and weights are unchanged by the extra pass."""
    rng = np.random.default_rng(8)
    S = 128
    sizes = [3000, 0, 1, 777, 4100]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    ids = rng.integers(0, 1 << 50, int(off[-1]), dtype=np.uint64)
    ids[off[3] + 5:off[3] + 300] = ids[off[3] + 4]            # set 3: 296 copies of one id -> identical points, ties
    w = np.round(rng.gamma(1.5, 4.0, ids.size)) + 1.0
    w[off[3] + 5:off[3] + 300] = w[off[3] + 4]
    w[::53] = 0.0
    sig, tw, own = gpu_ctx.bmh_from_weighted_ids(ids, w, off, S)
    sig0, tw0 = gpu_ctx.bmh_from_weighted(ids, w, off, S)
    np.testing.assert_array_equal(sig.view(np.uint64), sig0.view(np.uint64))
    np.testing.assert_array_equal(tw, tw0)
    for i in range(len(sizes)):
        lo, hi = int(off[i]), int(off[i + 1])
        es, et, eo = oracle.bmh_from_weighted_ids(ids[lo:hi], w[lo:hi], S)
        assert tw[i] == et
        np.testing.assert_array_equal(sig[i].view(np.uint64), es.view(np.uint64))
        np.testing.assert_array_equal(own[i], eo)
    assert (own[1] == np.uint64(0xFFFFFFFFFFFFFFFF)).all()

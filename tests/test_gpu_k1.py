"""GPU parity tests for K1 (2-bit packed bases -> OPH registers): the HIP kernel, called through
the C ABI, must reproduce the oracle's uint64 registers BIT-EXACTLY."""
import numpy as np
import pytest

from dashing2_amd import synth

pytestmark = pytest.mark.gpu


def _messy_fasta(rng, nrec, maxlen, with_n=True):
    recs = []
    for r in range(nrec):
        L = int(rng.integers(0, maxlen))
        g = synth.random_genome(int(rng.integers(0, 1 << 30)), max(L, 1))[:L].copy()
        if with_n and L > 10:
            for _ in range(int(rng.integers(0, 4))):
                a = int(rng.integers(0, L - 1))
                g[a:a + int(rng.integers(1, 40))] = ord("N")
        if r % 3 == 1:
            g = np.frombuffer(bytes(g).lower(), np.uint8)
        recs.append(synth.fasta_bytes(f"rec{r}", g, width=int(rng.integers(20, 90))))
    return b"".join(recs)


@pytest.mark.parametrize("k,S,canon,xormask", [
    (31, 1024, True, 0),              # BASELINE config (CLI default: XORMASK = 0, canon on)
    (21, 2048, True, 0),
    (32, 1024, True, 0),              # k = 32: full 64-bit k-mers
    (31, 1024, False, 0),
    (31, 1000, True, 0),              # non power-of-two m: (uint32_t)id % m
    (15, 63, True, 0),                # odd S -> m = 64
    (5, 16, True, 0x724526e320f9967d),  # library-default XORMASK (enums.cpp:131)
    (1, 8, False, 0),
    (27, 4096, True, 12345),
])
def test_k1_registers_bit_exact(gpu_ctx, d2g, oracle, k, S, canon, xormask):
    rng = np.random.default_rng(k * 1000 + S)
    genomes = [
        _messy_fasta(rng, 5, 5000),
        synth.fasta_bytes("clean", synth.random_genome(7, 200000)),     # > 1 workgroup of chunks
        _messy_fasta(rng, 40, 300),                                     # many short runs (read-like)
        b">tiny\nACG\n",                                                # shorter than k (mostly)
        b"",                                                            # empty input
        synth.fasta_bytes("b", synth.random_genome(9, 70000)) + synth.fasta_bytes("c", synth.random_genome(10, 1234)),
    ]
    sp = d2g.SeqPack(k)
    for g in genomes:
        sp.add_fastx(g)
    regs = gpu_ctx.oph_sketch_seqpack(sp, S, canon=canon, xormask=xormask)
    m = d2g.oph_m(S)
    assert regs.shape == (len(genomes), m)
    for gi, g in enumerate(genomes):
        eregs, esig, ecard, enk = oracle.sketch_buffer(g, k=k, canon=canon, xormask=xormask, S=S)
        assert sp.nkmers(gi) == enk
        np.testing.assert_array_equal(regs[gi], eregs, err_msg=f"genome {gi}")
    # host finalisation of the GPU registers == oracle doubles / cardinalities, bit for bit
    sigs, cards = d2g.oph_finalize(regs, S)
    for gi, g in enumerate(genomes):
        _, esig, ecard, _ = oracle.sketch_buffer(g, k=k, canon=canon, xormask=xormask, S=S)
        np.testing.assert_array_equal(sigs[gi].view(np.uint64), esig.view(np.uint64))
        assert np.float64(cards[gi]).view(np.uint64) == np.float64(ecard).view(np.uint64)


def test_k1_split_invariance(gpu_ctx, d2g):
    """OPH min is associative/commutative: sketching a genome as one run or as overlapping pieces
    (k-1 overlap) or with its records in another order gives identical registers."""
    k, S = 31, 1024
    g = synth.random_genome(42, 300000)
    sp = d2g.SeqPack(k)
    sp.add_sequence(g.tobytes())
    # same bases presented as 3 records with k-1 overlap => same k-mer multiset
    cut1, cut2 = 100000, 200007
    fa = (synth.fasta_bytes("a", g[:cut1 + k - 1]) + synth.fasta_bytes("b", g[cut1:cut2 + k - 1]) +
          synth.fasta_bytes("c", g[cut2:]))
    sp.add_fastx(fa)
    regs = gpu_ctx.oph_sketch_seqpack(sp, S)
    np.testing.assert_array_equal(regs[0], regs[1])
    # canonical k-mers: reverse complement gives the same sketch
    rc = g[::-1].tobytes().translate(bytes.maketrans(b"ACGT", b"TGCA"))
    sp2 = d2g.SeqPack(k)
    sp2.add_sequence(rc)
    regs2 = gpu_ctx.oph_sketch_seqpack(sp2, S)
    np.testing.assert_array_equal(regs[0], regs2[0])


def test_k1_full_size_cardinality(gpu_ctx, d2g):
    """BASELINE-size genome (5 Mbp, k=31, S=1024): statistical known-answer of test/oph.cpp
    (|card - n|/n ~ 1/sqrt(m)) on the GPU registers, plus idempotence of a second run."""
    k, S, L = 31, 1024, 5_000_000
    g = synth.random_genome(3, L)
    sp = d2g.SeqPack(k)
    sp.add_sequence(g.tobytes())
    regs = gpu_ctx.oph_sketch_seqpack(sp, S)
    regs_again = gpu_ctx.oph_sketch_seqpack(sp, S)
    np.testing.assert_array_equal(regs, regs_again)
    sigs, cards = d2g.oph_finalize(regs, S)
    n = L - k + 1                      # random 31-mers: essentially all distinct
    assert abs(cards[0] - n) / n < 4 / np.sqrt(S)
    assert (regs[0] != np.uint64(2 ** 64 - 1)).all()
    assert ((regs[0] & np.uint64(S - 1)) == np.arange(S, dtype=np.uint64)).all()   # id mod m == bucket


def test_k1_config2_shaped_launch_vs_oracle(gpu_ctx, d2g, oracle):
    """BASELINE config 2's launch shape -- MANY multi-megabase genomes in ONE launch (config 2: 1000 x 5 Mbp; here 160 x
    1-5 Mbp = 0.45 Gbp, ~7000 workgroups, k = 31, S = 1024), ingested through d2g_seqpack: a sample of genomes (first,
    last, the 5 Mbp ones, some in between) is compared register for register with the ORACLE, every genome with its own
    single-genome launch result computed from the same pack order, and the x87 finalisation with the oracle's."""
    from concurrent.futures import ThreadPoolExecutor
    k, S, n = 31, 1024, 160
    lens = [5_000_000 if i % 40 == 0 else 1_000_000 + 25_000 * (i % 37) for i in range(n)]
    with ThreadPoolExecutor(16) as ex:
        fastas = list(ex.map(lambda i: synth.fasta_bytes_fast("g%03d" % i, synth.random_genome(4000 + i, lens[i])), range(n)))
    sp = d2g.SeqPack(k)
    for f in fastas:
        sp.add_fastx(f)
    assert sp.ngenomes == n and sp.nbases == sum(lens)
    regs = gpu_ctx.oph_sketch_seqpack(sp, S)
    sigs, cards = d2g.oph_finalize(regs, S, nthreads=8)
    sample = sorted({0, 1, 39, 40, 41, 79, 80, 120, 121, 158, 159})
    with ThreadPoolExecutor(len(sample)) as ex:
        exp = list(ex.map(lambda i: oracle.sketch_buffer(fastas[i], k=k, S=S), sample))
    for i, (eregs, esig, ecard, enk) in zip(sample, exp):
        assert enk == lens[i] - k + 1
        np.testing.assert_array_equal(regs[i], eregs, err_msg=f"genome {i}")
        np.testing.assert_array_equal(sigs[i].view(np.uint64), esig.view(np.uint64), err_msg=f"genome {i}")
        assert cards[i] == ecard
    # every genome: no empty bucket, id mod m == bucket, distinct sketches, cardinality near its k-mer count
    assert (regs != np.uint64(2 ** 64 - 1)).all()
    assert ((regs & np.uint64(S - 1)) == np.arange(S, dtype=np.uint64)[None, :]).all()
    assert len({r.tobytes() for r in regs}) == n
    nk = np.array(lens, np.float64) - k + 1
    assert (np.abs(cards - nk) / nk < 5 / np.sqrt(S)).all()


def test_k1_rejects_bad_input(gpu_ctx, d2g):
    sp = d2g.SeqPack(31)
    sp.add_sequence(synth.random_genome(1, 1000).tobytes())
    packed, rs, rl, go = sp.arrays()
    with pytest.raises(d2g.D2GError):
        gpu_ctx.oph_sketch(packed[:-64], rs, rl, go, 31, 1024)      # missing tail pad
    with pytest.raises(d2g.D2GError):
        gpu_ctx.oph_sketch(packed, rs, rl, go, 33, 1024)            # k > 32 unsupported
    with pytest.raises(d2g.D2GError):
        gpu_ctx.oph_sketch(packed, rs, (rl * 0 + 5).astype(np.uint32), go, 31, 1024)   # run shorter than k


def test_k1_huge_sketch_global_registers(gpu_ctx, d2g, oracle):
    """m * 8 bytes > 128 KiB does not fit the LDS register file: the kernel then min-updates HBM directly."""
    k, S = 21, 32768
    fa = synth.fasta_bytes("g", synth.random_genome(5, 400000))
    sp = d2g.SeqPack(k)
    sp.add_fastx(fa)
    regs = gpu_ctx.oph_sketch_seqpack(sp, S)
    np.testing.assert_array_equal(regs[0], oracle.sketch_buffer(fa, k=k, S=S)[0])
    # and the largest LDS-resident size
    S2 = 16384
    regs2 = gpu_ctx.oph_sketch_seqpack(sp, S2)
    np.testing.assert_array_equal(regs2[0], oracle.sketch_buffer(fa, k=k, S=S2)[0])


def test_k1_long_run_split_registers(oracle, tmp_path):
    """same split exercised end to end on the GPU (fresh process with D2G_MAX_RUN=4096)."""
    import subprocess, sys, os
    from conftest import ROOT
    code = """
import sys, numpy as np
sys.path.insert(0, %r)
import dashing2_amd as D
from dashing2_amd import synth
from oracle import oracle as O
g = synth.fasta_bytes("g", synth.random_genome(12, 300000))
sp = D.SeqPack(31); sp.add_fastx(g)
assert sp.nruns > 50
ctx = D.Context(0)
regs = ctx.oph_sketch_seqpack(sp, 1024)
assert np.array_equal(regs[0], O.sketch_buffer(g, k=31, S=1024)[0])
print("OK")
""" % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, D2G_MAX_RUN="4096"))
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-2000:]


def test_k1_sketcher_reuse(gpu_ctx, d2g, oracle):
    """the persistent front end (d2g_sketcher) over groups of different shapes: buffers grow and are
    reused, results stay bit-exact"""
    sk = gpu_ctx.sketcher()
    rng = np.random.default_rng(1)
    for rnd, (k, S, lens) in enumerate([(31, 1024, [50000, 1200]), (21, 256, [300000, 7, 90000, 4000]), (31, 1024, [100]), (15, 1000, [20000] * 5)]):
        fas = [synth.fasta_bytes(f"g{rnd}_{i}", synth.random_genome(100 * rnd + i, L)) for i, L in enumerate(lens)]
        sp = d2g.SeqPack(k)
        for f in fas:
            sp.add_fastx(f)
        regs = sk.run(sp, S)
        for i, f in enumerate(fas):
            np.testing.assert_array_equal(regs[i], oracle.sketch_buffer(f, k=k, S=S)[0], err_msg=f"round {rnd} genome {i}")
    sk.close()


def test_k1_registers_on_kseq_oddities(gpu_ctx, d2g, oracle):
    """the record-walk oddities of tests/test_host.py (headers inside lines, junk around FASTQ records, bad quality strings ending
    the input, CR LF) through the product path: host parser -> K1 registers, bit for bit the oracle's"""
    from test_host import KSEQ_ODDITIES
    k, S = 5, 64
    sp = d2g.SeqPack(k)
    for buf in KSEQ_ODDITIES:
        sp.add_fastx(buf)
    regs = gpu_ctx.oph_sketch_seqpack(sp, S)
    for i, buf in enumerate(KSEQ_ODDITIES):
        eregs, _, _, nk = oracle.sketch_buffer(buf, k=k, S=S)
        assert sp.nkmers(i) == nk
        np.testing.assert_array_equal(regs[i], eregs, err_msg=repr(buf))
    sp.close()

"""GPU parity tests for K0 (FASTA bytes -> packed run stream on the device, d2g_sketcher_ingest_fasta): sketches built
from the device-parsed stream must equal the oracle's -- and the host parser's (d2g_seqpack) -- bit for bit, for every
oddity of the format the host parser handles; what the device parser does not handle must be refused, not guessed."""
import numpy as np
import pytest

from dashing2_amd import synth

pytestmark = pytest.mark.gpu


def _messy(rng, nrec, maxlen, crlf=False, junk=True):
    out = []
    for r in range(nrec):
        L = int(rng.integers(0, maxlen))
        g = synth.random_genome(int(rng.integers(0, 1 << 30)), max(L, 1))[:L].copy()
        if junk and L > 10:
            for _ in range(int(rng.integers(0, 5))):
                a = int(rng.integers(0, L - 1))
                g[a:a + int(rng.integers(1, 60))] = ord("NnRY-*.x>@+"[int(rng.integers(0, 11))])
        if r % 3 == 1:
            g = np.frombuffer(bytes(g).lower(), np.uint8)
        width = int(rng.integers(1, 120))
        body = bytes(g)
        nl = b"\r\n" if crlf else b"\n"
        lines = [body[i:i + width] for i in range(0, len(body), width)]
        rec = (b">" if (r % 7 or r == 0) else b"@") + b"rec%d some description > with @ signs + plus" % r + nl + nl.join(lines) + (nl if r % 5 else b"")
        if r % 11 == 3:
            rec += nl + nl                                             # empty lines inside a record
        out.append(rec)
    s = b"".join(out)
    # a line may not START with '+' (that is FASTQ: refused) -- keep such junk off the line starts
    return s.replace(b"\n+", b"\nA") if junk else s


def _inputs(rng):
    return [
        _messy(rng, 6, 6000),
        synth.fasta_bytes("clean", synth.random_genome(7, 300_000)),           # many tiles, one run
        _messy(rng, 60, 200),                                                  # read-like: many short records
        b">tiny\nACG\n",                                                       # shorter than most k
        b">only header\n",
        b">x\n" + b"ACGT" * 3000,                                              # no trailing line feed
        _messy(rng, 8, 3000, crlf=True),                                       # CRLF line ends
        b">cr\nACGTACGTAC\rGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\r\r\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACG\r",   # CR inside / doubled / at EOF
        b">a\n" + bytes(synth.random_genome(3, 4096 * 3 - 3)) + b"\n>b\n" + bytes(synth.random_genome(4, 5000)) + b"\n",   # header exactly at a tile edge
        b">w\n" + b"\n".join(bytes(synth.random_genome(50 + i, 15)) for i in range(2000)) + b"\n",   # 15-base lines: line feeds in every chunk
        b">n\n" + b"N" * 9000 + bytes(synth.random_genome(5, 20000)) + b"N" * 5000 + b"\n",   # break runs spanning tiles
        b">" + b"h" * 10_000 + b" ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\n" + bytes(synth.random_genome(6, 300_000)) + b"\n>" + b"ACGT" * 3000,   # a header
        # line longer than two tiles (its ACGT text is not sequence), an unwrapped 300 kb sequence line, a last header without a line feed
    ]


@pytest.mark.parametrize("k,S,canon", [(31, 1024, True), (21, 256, True), (32, 64, False), (1, 8, True), (13, 1000, True)])
def test_k0_ingest_matches_oracle_and_host_parser(gpu_ctx, d2g, oracle, k, S, canon):
    rng = np.random.default_rng(k * 77 + S)
    genomes = _inputs(rng)
    sk = gpu_ctx.sketcher()
    runs = sk.ingest_fasta(genomes, k)
    rs, rl, go, nk, nbases = runs
    regs = sk.run_ingested(runs, S, canon=canon)
    sp = d2g.SeqPack(k)
    for g in genomes:
        sp.add_fastx(g)
    hregs = gpu_ctx.oph_sketch_seqpack(sp, S, canon=canon)
    _, hrs, hrl, hgo = sp.arrays()
    # the same runs in the same order (lengths; the device stream keeps the dead short runs, so starts differ)
    np.testing.assert_array_equal(go, hgo)
    np.testing.assert_array_equal(rl, hrl)
    assert (rl >= k).all() and nbases >= int(rl.sum()) == sp.nbases
    for gi, g in enumerate(genomes):
        eregs, _, _, enk = oracle.sketch_buffer(g, k=k, canon=canon, S=S)
        assert int(nk[gi]) == enk == sp.nkmers(gi), gi
        np.testing.assert_array_equal(regs[gi], eregs, err_msg=f"genome {gi} vs oracle")
        np.testing.assert_array_equal(regs[gi], hregs[gi], err_msg=f"genome {gi} vs host parser")
    sk.close()


def test_k0_multiset_and_multi_file_genomes(gpu_ctx, d2g, oracle):
    """several files feeding ONE sketch (a reference input line with spaces) and the --multiset chain over a device-parsed
    stream: equal to the oracle on the concatenation of the files' records"""
    rng = np.random.default_rng(5)
    k, S = 21, 128
    files = [_messy(rng, 3, 4000) for _ in range(5)] + [synth.fasta_bytes("z", synth.random_genome(99, 50_000))]
    per_genome = [2, 1, 3]
    sk = gpu_ctx.sketcher()
    runs = sk.ingest_fasta(files, k, genome_nfiles=per_genome)
    sig, tw = sk.run_bmh_ingested(runs, S)
    regs = sk.run_ingested(runs, 512)
    at = 0
    for gi, nfi in enumerate(per_genome):
        cat = b"".join(files[at:at + nfi])
        at += nfi
        esig, etw, _ = oracle.bmh_sketch_buffer(cat, k, S)
        assert tw[gi] == etw
        np.testing.assert_array_equal(sig[gi].view(np.uint64), esig.view(np.uint64), err_msg=f"genome {gi}")
        np.testing.assert_array_equal(regs[gi], oracle.sketch_buffer(cat, k=k, S=512)[0])
    sk.close()


def test_k0_refuses_what_only_the_host_parser_handles(gpu_ctx, d2g):
    import gzip
    sk = gpu_ctx.sketcher()
    fa = synth.fasta_bytes("g", synth.random_genome(1, 5000))
    fq = b"@r1\nACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIII\n"
    # "junk" + fa: kseq finds the header in the MIDDLE of the first line (byte-wise search) -- the host parser follows it, K0 refuses
    for bad in (fq, gzip.compress(fa), b"junk first\n" + fa, b"junk" + fa, fa + b"+\nIIII\n"):
        with pytest.raises(d2g.D2GError) as ei:
            sk.ingest_fasta([fa, bad], 21)
        assert ei.value.status == -5                                   # D2G_ERR_UNSUPPORTED
        # nothing is staged after a refusal: the "ingested stream" form must not run on stale data
        rs = np.zeros(1, np.uint64), np.full(1, 100, np.uint32), np.array([0, 1], np.uint64)
        with pytest.raises(d2g.D2GError):
            sk.run_ingested(rs, 64)
    # and the sketcher is still usable afterwards, both ways
    runs = sk.ingest_fasta([fa], 21)
    a = sk.run_ingested(runs, 64)
    sp = d2g.SeqPack(21)
    sp.add_fastx(fa)
    np.testing.assert_array_equal(a, sk.run(sp, 64))
    with pytest.raises(d2g.D2GError):                                  # the host-packed run overwrote the device stream
        sk.run_ingested(runs, 64)
    sk.close()


def test_k0_long_run_split_and_many_run_starts(gpu_ctx, d2g, oracle, monkeypatch):
    """D2G_MAX_RUN (test hook) splits long runs exactly as the host packer does; an input with more run starts than the
    device list's first allocation (2^20) makes the emit pass run twice"""
    monkeypatch.setenv("D2G_MAX_RUN", "1000")
    k, S = 31, 256
    g = synth.fasta_bytes("long", synth.random_genome(11, 50_000))
    sk = gpu_ctx.sketcher()
    runs = sk.ingest_fasta([g], k)
    assert runs[1].max() <= 1000 and runs[1].size > 40
    sp = d2g.SeqPack(k)
    sp.add_fastx(g)
    np.testing.assert_array_equal(runs[1], sp.arrays()[2])
    np.testing.assert_array_equal(sk.run_ingested(runs, S)[0], oracle.sketch_buffer(g, k=k, S=S)[0])
    monkeypatch.delenv("D2G_MAX_RUN")
    # 1.2 million runs of 3 bases: ACGNACGN...
    many = b">m\n" + b"ACGN" * 1_200_000 + b"\n"
    runs = sk.ingest_fasta([many], 3)
    assert runs[0].size == 1_200_000 and (runs[1] == 3).all()
    np.testing.assert_array_equal(sk.run_ingested(runs, 64)[0], oracle.sketch_buffer(many, k=3, S=64)[0])
    sk.close()

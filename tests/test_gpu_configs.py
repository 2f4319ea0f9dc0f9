"""BASELINE.json configs at their FULL launch sizes under -m gpu: the 1000-genome single launch of
config 2 with the all-pairs PHYLIP through the CLI, and a 250-genome K3 call of config 5.  The oracle finishes a sample
in seconds; size-independent properties cover every genome."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from conftest import ROOT
from dashing2_amd import synth

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "dashing2_amd", "bin", "dashing2")


def _cores():
    try:
        return max(1, min(32, len(os.sched_getaffinity(0))))
    except (AttributeError, OSError):
        return 8


def test_config2_thousand_genomes_one_launch_and_cli_phylip(gpu_ctx, d2g, oracle, tmp_path):
    """BASELINE config 2 as stated: 1 000 synthetic 5 Mbp genomes, k = 31, S = 1024, OPH, full all-pairs PHYLIP.
    (a) ONE K1 launch over all 1000 genomes (5 Gbp, ingested through d2g_seqpack): registers of a 16-genome sample equal
        the oracle's bit for bit; every genome: no empty bucket, id mod m == bucket, cardinality near its k-mer count.
    (b) `dashing2 sketch --cmpout --phylip` on the same genomes as FASTA files on disk: its stacked signatures equal the
        x87 finalisation of (a)'s registers for ALL 1000 genomes, and the PHYLIP rows of the sample equal the oracle's
        distances in the reference's float text."""
    from oracle import textfmt
    k, S, n, L = 31, 1024, 1000, 5_000_000
    nthr = _cores()
    d = tmp_path / "fa"
    d.mkdir()
    paths = [str(d / ("g%05d.fa" % i)) for i in range(n)]
    sample = sorted({0, 1, 2, 63, 64, 249, 250, 251, 499, 500, 623, 750, 876, 997, 998, 999})
    packs = []

    def make(chunk):
        sp = d2g.SeqPack(k)
        kept = {}
        for i in chunk:
            fa = synth.fasta_bytes_fast("g%05d" % i, synth.random_genome(i, L))
            with open(paths[i], "wb") as f:
                f.write(fa)
            sp.add_fastx(fa)
            if i in sample:
                kept[i] = fa
        arr = sp.arrays()
        nb = sp.nbases
        sp.close()
        return chunk, arr, nb, kept

    chunks = [list(range(c, min(n, c + 8))) for c in range(0, n, 8)]
    with ThreadPoolExecutor(nthr) as ex:
        parts = list(ex.map(make, chunks))
    # one packed run stream for all 1000 genomes (what bench.py's sketch leg builds)
    packed, rs, rl, go, fastas = [], [], [], [np.zeros(1, np.uint64)], {}
    byte_off = run_off = 0
    for chunk, (p, s, l, g), nb, kept in parts:
        p = p[:(nb + 3) // 4]
        packed.append(p)
        rs.append(s + np.uint64(byte_off * 4))
        rl.append(l)
        go.append(g[1:] + np.uint64(run_off))
        byte_off += p.size
        run_off += s.size
        fastas.update(kept)
    packed.append(np.zeros(64, np.uint8))
    packed = np.concatenate(packed)
    regs = gpu_ctx.oph_sketch(packed, np.concatenate(rs), np.concatenate(rl), np.concatenate(go), k, S)   # ONE launch
    del packed
    assert regs.shape == (n, S)
    # every genome
    assert (regs != np.uint64(2 ** 64 - 1)).all()
    assert ((regs & np.uint64(S - 1)) == np.arange(S, dtype=np.uint64)[None, :]).all()
    sigs, cards = d2g.oph_finalize(regs, S, nthreads=nthr)
    nk = float(L - k + 1)
    assert (np.abs(cards - nk) / nk < 5 / np.sqrt(S)).all()
    assert len({r.tobytes() for r in regs[::7]}) == len(regs[::7])
    # the sample against the oracle
    with ThreadPoolExecutor(min(nthr, len(sample))) as ex:
        exp = dict(zip(sample, ex.map(lambda i: oracle.sketch_buffer(fastas[i], k=k, S=S), sample)))
    for i in sample:
        eregs, esig, ecard, enk = exp[i]
        assert enk == L - k + 1
        np.testing.assert_array_equal(regs[i], eregs, err_msg=f"genome {i}")
        np.testing.assert_array_equal(sigs[i].view(np.uint64), esig.view(np.uint64), err_msg=f"genome {i}")
        assert cards[i] == ecard
    # (b) the CLI on the files
    lst = tmp_path / "files.txt"
    lst.write_text("".join(p + "\n" for p in paths))
    out, phy = tmp_path / "stack.bin", tmp_path / "dist.phylip"
    r = subprocess.run([EXE, "sketch", "-k", str(k), "-S", str(S), "-p", str(nthr), "-F", str(lst), "-o", str(out), "--cmpout", str(phy), "--phylip"],
                       capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    raw = np.fromfile(out, np.uint8)
    assert tuple(raw[:16].view(np.uint64)) == (n, S)
    np.testing.assert_array_equal(raw[16:16 + 8 * n].view(np.float64), cards)
    np.testing.assert_array_equal(raw[16 + 8 * n:].view(np.uint64).reshape(n, S), sigs.view(np.uint64))
    lines = phy.read_text().split("\n")
    assert lines[0] == str(n) and len(lines) == n + 2 and lines[-1] == ""
    ssig = np.stack([exp[i][1] for i in sample])
    scard = np.array([exp[i][2] for i in sample])
    want = oracle.allpairs_ut(ssig, scard, measure=oracle.SIMILARITY, k=k, nthreads=2)
    idx = 0
    for a, i in enumerate(sample):
        row = lines[1 + i].split("\t")
        assert row[0] == textfmt.padded(paths[i])
        vals = row[1:]
        assert len(vals) == n - 1 - i
        for j in sample[a + 1:]:
            assert vals[j - i - 1] == textfmt.fmt_float(want[idx]), (i, j)
            idx += 1
    assert idx == want.size


def test_config5_250_genome_k3_call(gpu_ctx, d2g, oracle):
    """BASELINE config 5's launch: ONE d2g_bmh_sketch call over 250 genomes x 5 Mbp (k = 21, S = 2048, --multiset: exact k-mer
    counts + BagMinHash; 1.25e9 k-mers, the sub-batch pipeline and the two-level split at their real sizes).
    Oracle on 4 genomes (registers bit-identical, total weight); every genome: total weight = its k-mer count exactly,
    finite positive registers, distinct sketches."""
    k, S, n, L = 21, 2048, 250, 5_000_000
    nthr = _cores()
    sample = [0, 83, 166, 249]

    def make(chunk):
        sp = d2g.SeqPack(k)
        kept = {}
        for i in chunk:
            fa = synth.fasta_bytes_fast("m%05d" % i, synth.random_genome(70_000 + i, L))
            sp.add_fastx(fa)
            if i in sample:
                kept[i] = fa
        arr, nb = sp.arrays(), sp.nbases
        sp.close()
        return arr, nb, kept

    chunks = [list(range(c, min(n, c + 5))) for c in range(0, n, 5)]
    with ThreadPoolExecutor(nthr) as ex:
        parts = list(ex.map(make, chunks))
    packed, rs, rl, go, fastas = [], [], [], [np.zeros(1, np.uint64)], {}
    byte_off = run_off = 0
    for (p, s, l, g), nb, kept in parts:
        p = p[:(nb + 3) // 4]
        packed.append(p)
        rs.append(s + np.uint64(byte_off * 4))
        rl.append(l)
        go.append(g[1:] + np.uint64(run_off))
        byte_off += p.size
        run_off += s.size
        fastas.update(kept)
    packed.append(np.zeros(64, np.uint8))
    packed, rs, rl, go = np.concatenate(packed), np.concatenate(rs), np.concatenate(rl), np.concatenate(go)
    import ctypes as C
    sig = np.empty((n, S), np.float64)
    tw = np.empty(n, np.float64)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    gpu_ctx._check(d2g.lib().d2g_bmh_sketch(gpu_ctx._h, P(packed), packed.size, P(rs), P(rl), rs.size, P(go), n, k, 1, 0, S, 0.0, P(sig), P(tw)))
    assert (tw == float(L - k + 1)).all()                               # total-weight identity on every genome
    assert np.isfinite(sig).all() and (sig > 0).all()
    assert len({s.tobytes() for s in sig}) == n
    with ThreadPoolExecutor(len(sample)) as ex:
        exp = list(ex.map(lambda i: oracle.bmh_sketch_buffer(fastas[i], k, S), sample))
    for i, (esig, etw, enk) in zip(sample, exp):
        assert etw == tw[i] and enk == L - k + 1
        np.testing.assert_array_equal(sig[i].view(np.uint64), esig.view(np.uint64), err_msg=f"genome {i}")

"""CPU tests of the product's host half (libd2g.so loads without a GPU) against the oracle,
and of the C-ABI surface: every symbol include/d2g.h declares is exported and bound."""
import os
import re
import subprocess
import time

import numpy as np
import pytest

from conftest import ROOT, GOLDEN


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "d2g.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(d2g_[a-z0-9_]+)\s*\(", txt)))


def test_capi_exports_every_declared_symbol(d2g):
    syms = _header_symbols()
    assert len(syms) > 50
    out = subprocess.check_output(["nm", "-D", "--defined-only", d2g.LIB_PATH], text=True)
    exported = set(l.split()[-1] for l in out.splitlines() if " T " in l)
    missing = [s for s in syms if s not in exported]
    assert not missing, f"declared in d2g.h but not exported: {missing}"
    from dashing2_amd import capi
    unbound = [s for s in syms if s not in capi.SIGNATURES]
    assert not unbound, f"declared in d2g.h but not bound in capi.py: {unbound}"
    L = d2g.lib()
    for s in syms:
        assert hasattr(L, s)
    assert L.d2g_version() == 1


def test_no_gpu_means_loud_failure(d2g):
    """Without a usable gfx950 device the product refuses to run: there is no CPU fallback."""
    if d2g.lib().d2g_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(d2g.D2GError):
        d2g.Context(0)


def test_host_primitives_match_oracle(d2g, oracle):
    lib = oracle.load()
    for x in [0, 1, 133348, 2 ** 63, 2 ** 64 - 1, 0x0123456789abcdef]:
        assert d2g.wang_hash(x) == oracle.wang_hash(x)
    assert d2g.oph_xor_const() == lib.d2o_oph_xor_const() == 0xdc271ad2a9ecfb34
    for s in [0, 1, 13, 2 ** 40 + 7]:
        assert d2g.seed_mask(s) == lib.d2o_seed_mask(s)
    for S in [1, 2, 15, 16, 1000, 1023, 1024]:
        assert d2g.oph_m(S) == S + (S & 1)


def test_oph_finalize_bit_exact(d2g, oracle):
    rng = np.random.default_rng(3)
    for S in [16, 15, 100, 1024]:
        m = d2g.oph_m(S)
        n = 7
        regs = rng.integers(0, 2 ** 63, size=(n, m), dtype=np.uint64) >> np.uint64(rng.integers(0, 20))
        regs[0, :3] = np.uint64(2 ** 64 - 1)           # empty buckets
        regs[1, 0] = 0                                 # zero register -> 0 signature
        regs[2, :] = np.uint64(2 ** 64 - 1)            # fully empty sketch -> card inf? (sum != 0) fine
        sigs, cards = d2g.oph_finalize(regs, S, nthreads=2)
        for g in range(n):
            esig, ecard = oracle.regs_finalize(regs[g])
            np.testing.assert_array_equal(sigs[g].view(np.uint64), esig[:S].view(np.uint64))
            assert np.float64(cards[g]).view(np.uint64) == np.float64(ecard).view(np.uint64)


def test_densify_matches_oracle(d2g, oracle):
    rng = np.random.default_rng(8)
    S, n = 128, 9
    sigs = rng.random((n, S))
    sigs[rng.random((n, S)) < 0.4] = 0.0
    sigs[3] = 0.0
    out, nf = d2g.densify(sigs, nthreads=2)
    tot = 0
    for g in range(n):
        e, ne = oracle.densify(sigs[g])
        np.testing.assert_array_equal(out[g], e)
        tot += ne
    assert nf == tot


@pytest.mark.parametrize("S", [64, 100, 1024, 1000])
def test_epilogues_match_oracle(d2g, oracle, S):
    rng = np.random.default_rng(S)
    for _ in range(300):
        gt = int(rng.integers(0, S + 1))
        lt = int(rng.integers(0, S - gt + 1))
        lh, rh = float(rng.random() * 1e7 + 1), float(rng.random() * 1e7 + 1)
        for meas in range(6):
            a = d2g.epilogue_gtlt(gt, lt, S, lh, rh, meas, 31)
            b = oracle.compare_from_gtlt(gt, lt, S, lh, rh, meas, 31)
            assert np.float32(a).view(np.uint32) == np.float32(b).view(np.uint32), (gt, lt, meas)
            a = d2g.epilogue_neq(S - gt - lt, S, lh, rh, meas, 21)
            b = oracle.compare_from_neq(S - gt - lt, S, lh, rh, meas, 21)
            assert np.float32(a).view(np.uint32) == np.float32(b).view(np.uint32), (gt, lt, meas, "neq")


def test_epilogue_lut(d2g, oracle):
    lut = d2g.epilogue_lut(1024, d2g.SIMILARITY, 31)
    np.testing.assert_array_equal(lut, (np.arange(1025, dtype=np.float32) / np.float32(1024)))
    lut = d2g.epilogue_lut(1024, d2g.POISSON_LLR, 31)
    assert lut[0] == np.inf and lut[1024] == 0.0
    for e in [1, 17, 512, 1000]:
        assert lut[e] == np.float32(oracle.compare_from_gtlt(1024 - e, 0, 1024, 1., 1., oracle.POISSON_LLR, 31))
    # set space, non power-of-two S: value depends on (gt, lt) separately -> no table
    with pytest.raises(d2g.D2GError):
        d2g.epilogue_lut(1000, d2g.SIMILARITY, 31)
    # card-dependent measure -> no table
    with pytest.raises(d2g.D2GError):
        d2g.epilogue_lut(1024, d2g.INTERSECTION, 31)
    lut = d2g.epilogue_lut(1000, d2g.SIMILARITY, 31, multiset_space=True)
    for e in [0, 3, 999, 1000]:
        assert lut[e] == np.float32(oracle.compare_from_neq(e, 1000, 1., 1., oracle.SIMILARITY, 31))


def test_ut_count_and_partition(d2g):
    for N in [0, 1, 2, 5, 100, 1001]:
        assert d2g.ut_count(N) == N * (N - 1) // 2
        for r0, r1 in [(0, 0), (0, 1), (1, N), (N // 3, N // 2), (N, N)]:
            if r0 <= r1 <= N:
                assert d2g.ut_count(N, r0, r1) == sum(N - 1 - r for r in range(r0, r1))
    for N, P in [(10, 3), (1000, 8), (10000, 8), (50000, 8), (5, 8), (1, 2)]:
        b = d2g.ut_partition(N, P)
        assert b[0] == 0 and b[-1] == N and all(x <= y for x, y in zip(b, b[1:]))
        cnt = [d2g.ut_count(N, b[i], b[i + 1]) for i in range(P)]
        assert sum(cnt) == N * (N - 1) // 2
        if N >= 1000:
            assert max(cnt) <= 1.02 * (sum(cnt) / P) + N     # balanced to within one row


def _decode_runs(packed, rs, rl):
    out = []
    for s, l in zip(rs.tolist(), rl.tolist()):
        idx = np.arange(s, s + l)
        codes = (packed[idx >> 2] >> ((idx & 3) * 2).astype(np.uint8)) & 3
        out.append(np.frombuffer(b"ACGT", np.uint8)[codes].tobytes())
    return out


def test_seqpack_runs_and_kmer_counts(d2g, oracle):
    fa = (b">r1 desc\nACGTNNACGTACGTTTGA\nCCAGT\n>r2\nacgtgatcgatgctagctagc\r\n>r3\nAC\n"
          b">r4\nGGGGGGGGGGGGNGGGGG\n")
    for k in [3, 5, 11]:
        sp = d2g.SeqPack(k)
        sp.add_fastx(fa)
        sp.add_sequence(b"ACGTTGCATTGACNNNNACGTAGCTAGCTAGCATCGATCGAT")
        sp.add_fastx(b"")                     # empty genome
        packed, rs, rl, go = sp.arrays()
        assert sp.ngenomes == 3 and go[-1] == rs.size
        assert (rl >= k).all()
        # k-mer count equals the oracle's (== total_updates of the reference sketch)
        assert sp.nkmers(0) == oracle.sketch_buffer(fa, k=k, S=8)[3]
        assert sp.nkmers(2) == 0
        runs = _decode_runs(packed, rs, rl)
        import re as _re
        exp = [r.upper() for rec in [b"ACGTNNACGTACGTTTGACCAGT", b"acgtgatcgatgctagctagc", b"AC", b"GGGGGGGGGGGGNGGGGG"]
               for r in _re.split(rb"[^ACGTacgt]+", rec) if len(r) >= k]
        assert runs[:int(go[1])] == exp
        assert packed.size >= (sp.nbases + 3) // 4 + 64      # tail pad for the kernel's window reads
        sp.close()
    with pytest.raises(d2g.D2GError):
        d2g.SeqPack(33)                        # k > 32 is the reference's rolling-hash path: unsupported


def _random_fastx(rng, nrec):
    """records with random line widths (1..200), lower case, sprinkled non-ACGT bytes, LF or CRLF, FASTA or FASTQ
    (quality lines starting with '>' '@' '+'); returns the bytes and, per record, the concatenated sequence"""
    out, seqs = [], []
    for r in range(nrec):
        n = int(rng.choice([0, 1, 5, 63, 64, 65, 127, 128, 129, 300, 1000, 5000]))
        seq = rng.choice(np.frombuffer(b"ACGTacgt", np.uint8), n)
        if n and rng.random() < 0.6:                                   # other bytes: N, IUPAC, '-', '*'
            pos = rng.integers(0, n, max(1, n // int(rng.choice([7, 50, 400]))))
            seq[pos] = rng.choice(np.frombuffer(b"NnRYK-*.", np.uint8), pos.size)
        seq = seq.tobytes()
        width = int(rng.choice([1, 7, 60, 63, 64, 65, 70, 80, 128, 200, 10 ** 6]))
        eol = b"\r\n" if rng.random() < 0.2 else b"\n"
        fastq = rng.random() < 0.3
        lines = [seq[i:i + width] for i in range(0, len(seq), width)] if not fastq else [seq]
        rec = (b"@" if fastq else b">") + b"rec%d some description >@+" % r + eol + b"".join(l + eol for l in lines)
        if fastq:
            q = rng.choice(np.frombuffer(b">@+I5#", np.uint8), n).tobytes()
            rec += b"+" + eol + q + eol
        out.append(rec)
        seqs.append(seq)
    buf = b"".join(out)
    if rng.random() < 0.3 and buf.endswith(b"\n"):
        buf = buf[:-1]                                                  # no final line feed
    return buf, seqs


def test_seqpack_random_fastx_block_and_line_paths(d2g):
    """The packer takes 64-byte blocks of bases + line feeds in one step (AVX-512 VBMI2, where the CPU has it) and everything
    else line by line; the two alternate inside a record wherever a block holds another byte.  Runs must equal the maximal
    ACGT runs (>= k) of every record's concatenated sequence, in order, whatever the line width and line ends."""
    import re as _re
    rng = np.random.default_rng(20260929)
    for trial in range(60):
        k = int(rng.choice([1, 3, 11, 31]))
        buf, seqs = _random_fastx(rng, int(rng.integers(1, 8)))
        sp = d2g.SeqPack(k)
        sp.add_fastx(buf)
        packed, rs, rl, go = sp.arrays()
        exp = [r.upper() for seq in seqs for r in _re.split(rb"[^ACGTacgt]+", seq) if len(r) >= k]
        assert _decode_runs(packed, rs, rl) == exp, (trial, k)
        assert sp.nkmers(0) == sum(len(r) - k + 1 for r in exp)
        sp.close()
        spr = d2g.SeqPack(k)
        spr.add_fastx_by_record(buf)                                   # one genome per record: the same runs, grouped
        packed, rs, rl, go = spr.arrays()
        assert _decode_runs(packed, rs, rl) == exp and spr.ngenomes == len(seqs)
        per = [sum(len(r) - k + 1 for r in _re.split(rb"[^ACGTacgt]+", seq) if len(r) >= k) for seq in seqs]
        assert [spr.nkmers(i) for i in range(spr.ngenomes)] == per
        spr.close()


def test_seqpack_gz_and_multipath(d2g, oracle, tmp_path):
    import gzip
    from dashing2_amd import synth
    g1, g2 = synth.random_genome(1, 3000), synth.random_genome(2, 2500)
    p1, p2 = tmp_path / "a.fa", tmp_path / "b.fa.gz"
    synth.write_fasta(p1, "a", g1)
    with gzip.open(p2, "wb") as f:
        f.write(synth.fasta_bytes("b", g2))
    sp = d2g.SeqPack(21)
    sp.add_path(f"{p1} {p2}")                 # one "line" with two sub-paths feeds ONE sketch (d2.h:52-71)
    sp.add_path(str(p2))
    assert sp.ngenomes == 2
    assert sp.nkmers(0) == (3000 - 20) + (2500 - 20)
    assert sp.nkmers(1) == 2500 - 20
    assert oracle.sketch_file(f"{p1} {p2}", k=21, S=64)[3] == sp.nkmers(0)
    with pytest.raises(d2g.D2GError):
        sp.add_path(str(tmp_path / "missing.fa"))


def test_seqpack_truncated_gz_and_fifo(d2g, tmp_path):
    """a truncated / corrupt .gz must fail (D2G_ERR_IO), never be sketched from its readable prefix; a plain
    FIFO (process substitution) must deliver all of its bytes."""
    import gzip
    import threading
    from dashing2_amd import synth
    fa = synth.fasta_bytes("x", synth.random_genome(3, 200_000))
    good = tmp_path / "g.fa.gz"
    with gzip.open(good, "wb") as f:
        f.write(fa)
    raw = good.read_bytes()
    sp = d2g.SeqPack(21)
    sp.add_path(str(good))
    assert sp.nkmers(0) == 200_000 - 20
    for name, data in (("trunc.fa.gz", raw[:len(raw) // 2]), ("tail.fa.gz", raw[:-6]),
                       ("corrupt.fa.gz", raw[:len(raw) // 2] + bytes(64) + raw[len(raw) // 2 + 64:])):
        p = tmp_path / name
        p.write_bytes(data)
        with pytest.raises(d2g.D2GError):
            sp.add_path(str(p))
    fifo = tmp_path / "pipe.fa"
    os.mkfifo(fifo)

    def feed():
        with open(fifo, "wb") as f:
            f.write(fa)
    t = threading.Thread(target=feed)
    t.start()
    sp2 = d2g.SeqPack(21)
    sp2.add_path(str(fifo))
    t.join()
    assert sp2.nkmers(0) == 200_000 - 20
    a, b = sp.arrays(), sp2.arrays()
    n = (200_000 + 3) // 4
    assert np.array_equal(a[0][:n], b[0][:n])


def test_format_fixture_roundtrip():
    """tests/golden/stacked_*.bin parsed by the reference's python/parse.py (frozen) has the layout
    [u64 N][u64 S][f64 card x N][f64 x N*S]  (sketch_core.cpp:130-140, cmp_main.cpp:61-94)."""
    raw = np.fromfile(os.path.join(GOLDEN, "stacked_n5_s32.bin"), np.uint8)
    exp = np.load(os.path.join(GOLDEN, "stacked_n5_s32.expected.npz"))
    n, s = raw[:16].view(np.uint64)
    assert n == exp["nseqs"] == 5 and s == 32
    np.testing.assert_array_equal(raw[16:16 + 8 * n].view(np.float64), exp["cardinalities"])
    np.testing.assert_array_equal(raw[16 + 8 * n:].view(np.float64).reshape(n, s), exp["signatures"])
    one = np.fromfile(os.path.join(GOLDEN, "sketch_s32.opss"), np.float64)
    e1 = np.load(os.path.join(GOLDEN, "sketch_s32.expected.npz"))
    assert one[0] == e1["cardinality"]
    np.testing.assert_array_equal(one[1:], e1["signatures"])
    # convert_sketches_to_packed_sketch output == the stacked layout of the first 3 sketches
    pk = np.fromfile(os.path.join(GOLDEN, "packed_from_singles.bin"), np.uint8)
    assert tuple(pk[:16].view(np.uint64)) == (3, 32)
    np.testing.assert_array_equal(pk[16:16 + 24].view(np.float64), exp["cardinalities"][:3])
    np.testing.assert_array_equal(pk[40:].view(np.float64).reshape(3, 32), exp["signatures"][:3])


def test_cli_float_formatter_vs_fmt_golden():
    """the CLI's text path (dashing2_amd/host/fmtfloat.cpp) in both float layouts: `--fmt-compat 11` against the table fmt 12.1.0
    itself produced, the default (fmt < 11: fixed notation below 1e16) against the table derived from it by rule
    (tests/golden/make_fmt10_golden.py).  The two tables differ exactly in the values >= 1e7."""
    exe = os.path.join(ROOT, "dashing2_amd", "bin", "fmtcheck")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "dashing2_amd", "host")])
    for table, compat in (("fmt_float.tsv", ["11"]), ("fmt10_float.tsv", ["10"]), ("fmt10_float.tsv", [])):
        out = subprocess.run([exe, os.path.join(GOLDEN, table)] + compat, capture_output=True, text=True)
        assert out.returncode == 0, (table, compat, out.stdout)
        assert "0 bad" in out.stdout
    out = subprocess.run([exe, os.path.join(GOLDEN, "fmt_float.tsv"), "10"], capture_output=True, text=True)
    assert out.returncode != 0 and "1.2345678e+07" in out.stdout            # the layouts really differ there
    assert subprocess.run([exe, os.path.join(GOLDEN, "fmt_float.tsv"), "12"]).returncode == 2   # only 10 and 11 exist


def test_host_code_under_sanitizers():
    """ASan + UBSan builds of the x86 host half (seqpack ingest, x87 finalisation, densify, epilogues, partition: csrc
    `make sanitize`) and of the CLI's float formatter (host `make sanitize`) run clean (reference Makefile:102-103)"""
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "dashing2_amd", "csrc"), "sanitize"], capture_output=True, text=True)
    assert r.returncode == 0 and "host selftest OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "dashing2_amd", "host"), "sanitize"], capture_output=True, text=True)
    assert r.returncode == 0 and "0 bad" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def _cli():
    exe = os.path.join(ROOT, "dashing2_amd", "bin", "dashing2")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "dashing2_amd", "host")])
    return exe


def test_cli_flag_validation_and_errors(tmp_path):
    exe = _cli()
    r = subprocess.run([exe, "sketch", "--no-such-flag", "x.fa"], capture_output=True, text=True)
    assert r.returncode == 1 and "flag no-such-flag not found in expected set. See usage." in r.stderr   # options.h:298
    assert r.stderr.startswith("#Calling Dashing2 version")                                              # d2.cpp:136
    r = subprocess.run([exe, "sketch"], capture_output=True, text=True)
    assert r.returncode == 1 and "No paths provided. See usage." in r.stderr                              # sketch_main.cpp:131-134
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "dashing2 has several subcommands" in r.stderr                           # d2.cpp:112
    r = subprocess.run([exe, "sketch", "-h"], capture_output=True, text=True)
    assert r.returncode == 1                                                                              # sketch_main.cpp:66
    r = subprocess.run([exe, "cmp", "--presketched", "--bogus"], capture_output=True, text=True)
    assert r.returncode == 1 and "flag bogus not found" in r.stderr
    r = subprocess.run([exe, "sketch", "--presketched", "x"], capture_output=True, text=True)
    assert r.returncode == 1 and "flag presketched not found" in r.stderr      # only valid for cmp (options.h:287-289)
    for flag in (["--prob"], ["--full"], ["-k", "40"], ["--edit-distance"], ["--topk", "3"]):
        r = subprocess.run([exe, "sketch"] + flag + ["x.fa"], capture_output=True, text=True)
        assert r.returncode == 1 and "outside" in r.stderr, flag


def test_seqpack_long_run_split(oracle, tmp_path):
    """runs longer than the u32-safe limit are split into overlapping pieces (k-1 bases shared in place):
    the k-mer multiset must be unchanged.  D2G_MAX_RUN lowers the limit for the test (fresh process)."""
    import subprocess, sys, json
    code = """
import sys, json, numpy as np
sys.path.insert(0, %r)
import dashing2_amd as D
from dashing2_amd import synth
g = synth.random_genome(11, 5000).tobytes()
sp = D.SeqPack(21)
sp.add_sequence(g)
packed, rs, rl, go = sp.arrays()
kmers = set()
tot = 0
for s0, l in zip(rs.tolist(), rl.tolist()):
    assert l <= 1000 and l >= 21
    tot += l - 20
    for p in range(s0, s0 + l - 20):
        kmers.add(p)
print(json.dumps({"nruns": int(rs.size), "sum_kmers": tot, "distinct_starts": len(kmers), "nkmers": sp.nkmers(0)}))
""" % ROOT
    env = dict(os.environ, D2G_MAX_RUN="1000")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["nruns"] > 4
    assert r["nkmers"] == 5000 - 20
    assert r["sum_kmers"] == 5000 - 20 == r["distinct_starts"]      # every k-mer start covered exactly once


def test_seqpack_by_record_matches_oracle_walk(d2g, oracle):
    """--parse-by-seq packing: one genome per FASTX record, kseq names, k-mer counts per record equal
    to the oracle's record walk (fastxsketchbyseq.cpp:233-252)."""
    buf = (b">r1 desc\nACGTACGTTTGACCA\nACGGT\n>r2\nNNNN\n>r3\tx\nACGTACGTTTGACCAACGGT\r\n@q\nACGTAGCATCGACTAGCTA\n+\n"
           b"IIIIIIIIIIIIIIIIIII\n>last")
    sp = d2g.SeqPack(5)
    sp.add_fastx_by_record(buf)
    names, sigs, cards = oracle.sketch_buffer_byseq(buf, 5, 8)
    assert [sp.name(i) for i in range(sp.ngenomes)] == names == ["r1", "r2", "r3", "q", "last"]
    assert [sp.nkmers(i) for i in range(sp.ngenomes)] == [16, 0, 16, 15, 0]
    assert cards.tolist() == [14.0, 0.0, 14.0, 14.0, 0.0]          # below 10 S: exact distinct canonical 5-mers
    _, _, mcards = oracle.sketch_buffer_byseq(buf, 5, 8, multiset=True)
    assert mcards.tolist() == [16.0, 0.0, 16.0, 15.0, 0.0]         # multiset: total weight = k-mer count


def test_public_header_is_plain_c_and_links(tmp_path):
    """include/d2g.h is the drop-in boundary: it must compile as C11 and as C++17, and a C program must
    link against libd2g.so through it (host-only entry points; no GPU needed)."""
    import subprocess
    from conftest import ROOT
    src = tmp_path / "t.c"
    src.write_text('#include "d2g.h"\n#include <stdio.h>\n'
                   'int main(void) { double sig[4]; double card; unsigned long long regs[4] = {1ull << 60, 3ull << 61, ~0ull, 0};\n'
                   '  if (d2g_oph_finalize((const uint64_t *)regs, 1, 4, 4, sig, &card, 1) != D2G_OK) return 2;\n'
                   '  printf("%llu %d %s\\n", (unsigned long long)d2g_wang_hash(133348), d2g_version(), d2g_strerror(D2G_ERR_INTERNAL));\n'
                   '  return sig[2] == 0.0 && sig[3] == 0.0 && sig[0] > 0.0 ? 0 : 3; }\n')
    inc, libdir = os.path.join(ROOT, "include"), os.path.join(ROOT, "dashing2_amd")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-fsyntax-only", str(src)])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-pedantic", "-I", inc, "-fsyntax-only", "-x", "c++", str(src)])
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-std=c11", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-ld2g", f"-Wl,-rpath,{libdir}"])
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stderr)
    assert r.stdout.split()[2:] == ["internal", "invariant", "failed"]


def test_missing_rccl_is_an_error_code_not_a_crash():
    """with no loadable RCCL the communicator entry points must return D2G_ERR_UNSUPPORTED (the loader
    once read dlerror() twice and dereferenced the NULL of the second call).  Fresh process: the lookup is cached."""
    code = ("import os, sys, ctypes as C\n"
            "os.environ['D2G_NO_TORCH_PRELOAD'] = '1'\n"
            "sys.path.insert(0, %r)\n"
            "import dashing2_amd as D\n"
            "buf = C.create_string_buffer(128)\n"
            "rc = D.lib().d2g_comm_unique_id(buf)\n"
            "print('rc', rc)\n" % ROOT)
    env = dict(os.environ, D2G_RCCL_LIB="/nonexistent/librccl-missing.so")
    r = subprocess.run([os.environ.get("PYTHON", "python3"), "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-400:]
    assert "rc -5" in r.stdout, r.stdout


def test_host_threading_under_thread_sanitizer():
    """the two host pipelines of the CLI -- parser pool -> bounded queue -> consumer over d2g_seqpack, and producer -> slot
    queue -> emitter thread over the float formatter -- built with -fsanitize=thread (csrc `make tsan`): no report, exit 0"""
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "dashing2_amd", "csrc"), "tsan"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "host threads selftest OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    assert "ThreadSanitizer" not in r.stdout + r.stderr


def test_bench_refuses_to_mislabel_a_smaller_job(d2g):
    """`python bench.py --gpus N` without a launcher used to run the 1-GPU job and report n_gpus: 1.
    It now launches N ranks itself -- or, with fewer than N devices visible, prints a JSON line with "error" and exits 2."""
    import json
    import sys
    have = d2g.lib().d2g_device_count()
    want = have + 2
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(want), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == want and line["value"] is None and "refusing" in line["error"]


def _bench(extra_env, *argv, timeout=300):
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(extra_env)
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True, env=env, timeout=timeout)
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None), time.monotonic() - t0


def test_bench_ladder_survives_hung_and_failing_rungs(d2g):
    """the first hardware N > 1 run must not be lost to a hang.  Every rung of the multi-GPU ladder runs as a child
    process group under a watchdog: rungs that hang (test hook) are killed at the timeout, rungs that exit non-zero are abandoned at
    once, and with every rung gone the supervisor still prints ONE JSON line with "error" and the story of the ladder, exit code 2."""
    hooks = {"D2G_BENCH_TEST_SKIP_DEVICE_CHECK": "1", "D2G_BENCH_TEST_HANG": "cabi,inproc", "D2G_BENCH_TEST_FAIL": "torch,broadcast",
             "D2G_BENCH_RUNG_TIMEOUT": "3"}
    r, line, secs = _bench(hooks, "--gpus", "2", "--sketches", "64", "--steps", "1", "--warmup", "0")
    assert r.returncode == 2 and secs < 120, (r.returncode, secs, r.stderr[-500:])
    assert line["value"] is None and line["n_gpus"] == 2 and "every rung" in line["error"]
    lad = line["launcher"]["ladder"]
    assert [l["engine"] for l in lad] == ["cabi", "inproc", "torch", "broadcast"]
    assert lad[0]["outcome"].startswith("timeout") and lad[1]["outcome"].startswith("timeout")
    assert "exited with code 3" in lad[2]["outcome"] and "exited with code 3" in lad[3]["outcome"]
    # no child of a killed rung is left behind
    left = [pid for l in lad for pid in l["pids"] if os.path.exists("/proc/%d" % pid)]
    assert not left, left


def test_bench_ladder_keeps_a_headline_that_was_already_out(d2g):
    """a rung whose headline line is out before something later hangs (a secondary leg) still delivers the measurement: the
    supervisor kills it at the timeout and prints that line, exit code 0; an earlier rung that failed is recorded beside it"""
    hooks = {"D2G_BENCH_TEST_SKIP_DEVICE_CHECK": "1", "D2G_BENCH_TEST_FAIL": "cabi", "D2G_BENCH_TEST_LINE_THEN_HANG": "inproc",
             "D2G_BENCH_RUNG_TIMEOUT": "3"}
    r, line, secs = _bench(hooks, "--gpus", "2", "--sketches", "64", "--steps", "1", "--warmup", "0")
    assert r.returncode == 0 and secs < 120, (r.returncode, r.stderr[-500:])
    assert line["value"] == 1.0 and [l["engine"] for l in line["launcher"]["ladder"]] == ["cabi", "inproc"]
    assert line["launcher"]["ladder"][1]["line"] is True and line["launcher"]["ladder"][1]["outcome"].startswith("timeout")


def test_bench_under_a_launcher_only_rank0_supervises(d2g):
    """started by `torch.distributed.run` (RANK / WORLD_SIZE set): ranks other than 0 exit 0 at once without output, rank 0 is the
    supervisor; a WORLD_SIZE that contradicts --gpus is an error line"""
    r, line, _ = _bench({"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1"}, "--gpus", "2")
    assert r.returncode == 0 and line is None and r.stdout.strip() == ""
    r, line, _ = _bench({"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"}, "--gpus", "2")
    assert r.returncode == 2 and "WORLD_SIZE=4" in line["error"]
    hooks = {"D2G_BENCH_TEST_SKIP_DEVICE_CHECK": "1", "D2G_BENCH_TEST_FAIL": "cabi,inproc,torch,broadcast", "WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0",
             "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"}
    r, line, _ = _bench(hooks, "--gpus", "2", "--sketches", "64")
    assert r.returncode == 2 and "torch.distributed.run" in line["launcher"]["launched_by"]


def _kseq_records(buf):
    """klib kseq_read() restated character by character from the published macro (a third restatement, next to the oracle's line-wise
    C and the product's block-wise C++): -> [(name, sequence)] of every record a `while (kseq_read(ks) >= 0)` loop sees."""
    n, pos, last, out = len(buf), 0, 0, []

    def getc():
        nonlocal pos
        if pos >= n:
            return -1
        pos += 1
        return buf[pos - 1]

    def until_line(acc):                                     # ks_getuntil2(KS_SEP_LINE, append = 1)
        nonlocal pos
        if pos >= n:
            return -1
        e = buf.find(b"\n", pos)
        acc += buf[pos:e if e >= 0 else n]
        pos = e + 1 if e >= 0 else n
        if len(acc) > 1 and acc[-1] == 13:
            del acc[-1]
        return len(acc)

    while True:
        if last == 0:
            c = getc()
            while c != -1 and c not in (62, 64):
                c = getc()
            if c == -1:
                return out
            last = c
        if pos >= n:
            return out                                       # ks_getuntil(name) < 0
        name, c = bytearray(), None
        while pos < n:
            ch = buf[pos]
            pos += 1
            if ch in b" \t\n\v\f\r":
                c = ch
                break
            name.append(ch)
        if c is not None and c != 10:
            until_line(bytearray())                          # the comment
        seq = bytearray()
        c = getc()
        while c != -1 and c not in (62, 43, 64):
            if c != 10:
                seq.append(c)
                until_line(seq)
            c = getc()
        if c in (62, 64):
            last = c
        if c != 43:
            out.append((bytes(name), bytes(seq)))
            continue
        c = getc()
        while c != -1 and c != 10:
            c = getc()
        if c == -1:
            return out                                       # -2: no quality string
        qual = bytearray()
        while until_line(qual) >= 0 and len(qual) < len(seq):
            pass
        last = 0
        if len(qual) != len(seq):
            return out                                       # -2: quality of a different length
        out.append((bytes(name), bytes(seq)))


KSEQ_ODDITIES = [
    b"junk>r1 c\nACGTACGTAC\nGGTTA\n",                                   # header in the middle of the first line
    b"ACGTACGT\nxx@r1\nACGTTGCAAC\n+\nIIIIIIIIII\ntrailing junk with @r2 inside\nACGTACGTAC\n",   # '@' inside junk after a quality block
    b"@r1\nACGTACGTAC\n+\nIIIII\n@r2\nTTTTTGGGGG\n+\nIIIIIIIIII\n",       # quality too short: swallows the next header, then too long -> error
    b"@r1\nACGTACGTAC\n+\nIIIIIIIIIIII\n@r2\nTTTTTGGGGG\n+\nIIIIIIIIII\n",  # quality too long -> error, r2 never read
    b"@r1\nACGTACGTAC\n+",                                                # the input ends inside the '+' line
    b"@r1\nACGTACGTAC\n+\n",                                              # no quality line at all
    b"@r1\n+\n@r2\nACGTACGTAC\n+\nIIIIIIIIII\n",                          # empty sequence: one quality line is read anyway
    b">r1\nACGTACGTAC\n>",                                                # a header character as the last byte
    b">r1\r\nACGTA\r\nCGTAC\r\n\r\n>r2\r\nA\r\nC\r\n",                    # CR LF, an empty CR LF line, one-character lines
    b">r1\n\rACGTACGTAC\n",                                               # a lone CR first: kept while the sequence has one character
    b"@r1\nACGTACGTAC\n+r1\nIIIII\nIIIII\n@r2\nGGGGGCCCCC\n+\n>>>>>>>>>>\n>r3\nACGTAACGTA\n",  # multi-line quality, '>' quality, FASTA after FASTQ
    b"\n\n  >r1\nACGTACGTAC\n",                                           # leading blank lines and spaces
    b"",
    b"no header at all\nACGT\n",
]


def test_seqpack_and_oracle_follow_kseq_read(d2g, oracle):
    """the record walk of the product's host parser and of the oracle against klib's kseq_read() restated
    character by character: byte-wise search for the first header (and after every FASTQ record), headers inside lines, quality read
    by length with at least one line, a record with a bad quality string ends the input (kseq returns -2) -- names and per-record
    k-mer counts (parse-by-seq), and the k-mers of the whole input, on hand-made oddities and 300 random messy inputs."""
    import re as _re
    rng = np.random.default_rng(4242)
    cases = list(KSEQ_ODDITIES)
    for t in range(300):
        buf, _ = _random_fastx(rng, int(rng.integers(1, 6)))
        r = rng.random()
        if r < 0.25:
            buf = rng.choice(np.frombuffer(b"ACGT xyz\n", np.uint8), int(rng.integers(1, 40))).tobytes() + buf      # leading junk
        elif r < 0.5 and len(buf) > 4:
            buf = buf[:int(rng.integers(1, len(buf)))]                                                            # truncated anywhere
        elif r < 0.65 and b"+" in buf:
            i = buf.rfind(b"\n", 0, len(buf) - 1)
            buf = buf[:i] + b"#" + buf[i:]                                                                        # one quality line too long
        cases.append(buf)
    k = 5
    for buf in cases:
        recs = _kseq_records(buf)
        per = [sum(len(r) - k + 1 for r in _re.split(rb"[^ACGTacgt]+", s) if len(r) >= k) for _, s in recs]
        names, _, mcards = oracle.sketch_buffer_byseq(buf, k, 8, multiset=True)
        assert names == [nm.decode("latin1") for nm, _ in recs], buf
        assert mcards.tolist() == [float(p) for p in per], buf
        sp = d2g.SeqPack(k)
        sp.add_fastx_by_record(buf)
        assert [sp.name(i) for i in range(sp.ngenomes)] == names, buf
        assert [sp.nkmers(i) for i in range(sp.ngenomes)] == per, buf
        sp.close()
        sp = d2g.SeqPack(k)
        sp.add_fastx(buf)
        packed, rs, rl, go = sp.arrays()
        exp = [r.upper() for _, s in recs for r in _re.split(rb"[^ACGTacgt]+", s) if len(r) >= k]
        assert _decode_runs(packed, rs, rl) == exp, buf
        assert sp.nkmers(0) == sum(per) == oracle.sketch_buffer(buf, k=k, S=8)[3], buf
        sp.close()

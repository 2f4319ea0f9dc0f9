"""GPU end-to-end tests of the drop-in CLI (dashing2_amd/bin/dashing2): on-disk formats and text
output must be byte-identical to what the oracle + the reference's layouts prescribe."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from dashing2_amd import synth

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "dashing2_amd", "bin", "dashing2")


def _run(args, **kw):
    r = subprocess.run([EXE] + args, capture_output=True, **kw)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return r


@pytest.fixture(scope="module")
def genomes(tmp_path_factory):
    d = tmp_path_factory.mktemp("fa")
    base = synth.random_genome(100, 120000)
    paths = []
    for i, rate in enumerate([0.0, 0.001, 0.01, 0.05, 0.2]):
        g = synth.mutate(base, rate, seed=i) if rate else base
        p = d / f"g{i}.fa"
        synth.write_fasta(p, f"g{i}", g)
        paths.append(str(p))
    p = d / "other.fa"
    synth.write_fasta(p, "other", synth.random_genome(7, 90000))
    paths.append(str(p))
    p = d / "tiny.fa"            # fewer k-mers than buckets: exercises densify (cmp_core.cpp:577-613)
    synth.write_fasta(p, "tiny", synth.random_genome(9, 400))
    paths.append(str(p))
    return paths


def _oracle_result(oracle, paths, k, S, canon=True, seed=0):
    xm = oracle.load().d2o_seed_mask(seed)
    sigs, cards = oracle.sketch_files(paths, k=k, canon=canon, xormask=xm, S=S, nthreads=2)
    return sigs, cards


def _densified(oracle, sigs):
    return np.stack([oracle.densify(s)[0] for s in sigs])


OPTSTR = "Dashing2Options;k:{k};parsebyfile;trimchr;sketchsize:{S};sketchtype:onepermsetsketch;Fastx;canon"


@pytest.mark.parametrize("k,S", [(31, 1024), (21, 256)])
def test_cli_sketch_stacked_and_phylip(oracle, genomes, tmp_path, k, S):
    from oracle import textfmt
    out = tmp_path / "stack.bin"
    phy = tmp_path / "dist.phylip"
    r = _run(["sketch", "-k", str(k), "-S", str(S), "-p", "4", "-o", str(out), "--cmpout", str(phy), "--phylip"] + genomes)
    assert r.stderr.decode().startswith("#Calling Dashing2 version")
    esigs, ecards = _oracle_result(oracle, genomes, k, S)
    N = len(genomes)
    # F-b stacked file: [u64 N][u64 S][f64 card x N][f64 x N*S], sketches stored UN-densified
    raw = np.fromfile(out, np.uint8)
    exp = np.concatenate([np.array([N, S], np.uint64).view(np.uint8), ecards.view(np.uint8), esigs.reshape(-1).view(np.uint8)])
    assert raw.tobytes() == exp.tobytes()
    # F-c names file
    lines = open(str(out) + ".names.txt").read().splitlines()
    assert lines[0] == "#Name\tCardinality"
    for i, l in enumerate(lines[1:]):
        name, card = l.split("\t")
        assert name == genomes[i] and card == "%0.24g" % ecards[i]
    # F-e PHYLIP: byte-identical text
    dens = _densified(oracle, esigs)
    dist = oracle.allpairs_ut(dens, ecards, measure=oracle.SIMILARITY, k=k, nthreads=2)
    assert open(phy).read() == textfmt.render_symmetric(genomes, dist, phylip=True)
    assert dist.max() > 0.5 and dist.min() == 0.0


def test_cli_cmp_presketched_all_outputs(oracle, genomes, tmp_path):
    from oracle import textfmt
    k, S = 31, 512
    out = tmp_path / "s.bin"
    _run(["sketch", "-k", str(k), "-S", str(S), "-o", str(out)] + genomes)
    esigs, ecards = _oracle_result(oracle, genomes, k, S)
    dens = _densified(oracle, esigs)
    N = len(genomes)
    iu = np.triu_indices(N, 1)
    for flags, meas in [([], oracle.SIMILARITY), (["--distance"], oracle.POISSON_LLR), (["--intersection"], oracle.INTERSECTION),
                        (["--containment"], oracle.CONTAINMENT), (["--symmetric-containment"], oracle.SYMMETRIC_CONTAINMENT),
                        (["--union-size"], oracle.UNION_SIZE)]:
        exp = oracle.allpairs_ut(dens, ecards, measure=meas, k=k, nthreads=2)
        # F-d binary matrix
        b = tmp_path / "d.bin"
        _run(["cmp", "--presketched", "-k", str(k), "--binary-output", "--cmpout", str(b)] + flags + [str(out)])
        got = np.fromfile(b, np.float32)
        np.testing.assert_array_equal(got.view(np.uint32), exp.view(np.uint32), err_msg=str(flags))
        # F-f default TSV on stdout
        r = _run(["cmp", "--presketched", "-k", str(k)] + flags + [str(out)])
        opt = OPTSTR.format(k=k, S=S)
        assert r.stdout.decode() == textfmt.render_symmetric(genomes, exp, phylip=False, options_string=opt), str(flags)
    # asymmetric all-pairs (square): compare(i, j) for every ordered pair incl. the diagonal
    full = np.empty((N, N), np.float32)
    lib = oracle.load()
    import ctypes as C
    for i in range(N):
        for j in range(N):
            full[i, j] = lib.d2o_compare(dens.ctypes.data_as(C.POINTER(C.c_double)), ecards.ctypes.data_as(C.POINTER(C.c_double)),
                                         S, i, j, oracle.CONTAINMENT, k)
    b = tmp_path / "sq.bin"
    _run(["cmp", "--presketched", "-k", str(k), "--binary-output", "--asymmetric-all-pairs", "--containment", "--cmpout", str(b), str(out)])
    np.testing.assert_array_equal(np.fromfile(b, np.float32).view(np.uint32), full.reshape(-1).view(np.uint32))
    r = _run(["cmp", "--presketched", "-k", str(k), "--square", "--containment", str(out)])
    assert r.stdout.decode() == textfmt.render_rect(genomes, genomes, full, "Asymmetric pairwise", OPTSTR.format(k=k, S=S))


def test_cli_nonpow2_sketchsize_and_nocanon(oracle, genomes, tmp_path):
    k, S = 25, 1000                    # (gt, lt) path: the value depends on both counts
    b = tmp_path / "d.bin"
    _run(["sketch", "-k", str(k), "-S", str(S), "--no-canon", "--seed", "13", "--binary-output", "--cmpout", str(b), "--distance"] + genomes)
    esigs, ecards = _oracle_result(oracle, genomes, k, S, canon=False, seed=13)
    dens = _densified(oracle, esigs)
    exp = oracle.allpairs_ut(dens, ecards, measure=oracle.POISSON_LLR, k=k, nthreads=2)
    np.testing.assert_array_equal(np.fromfile(b, np.float32).view(np.uint32), exp.view(np.uint32))


def test_cli_cache_and_panel(oracle, genomes, tmp_path):
    k, S = 31, 256
    pref = tmp_path / "cache"
    pref.mkdir()
    refs, qs = genomes[:4], genomes[4:]
    ff, qf = tmp_path / "refs.txt", tmp_path / "qs.txt"
    ff.write_text("\n".join(refs) + "\n")
    qf.write_text("\n".join(qs) + "\n")
    b1, b2 = tmp_path / "p1.bin", tmp_path / "p2.bin"
    args = ["sketch", "-k", str(k), "-S", str(S), "--cache", "--outprefix", str(pref), "-F", str(ff), "-Q", str(qf), "--binary-output"]
    _run(args + ["--cmpout", str(b1)])
    esigs, ecards = _oracle_result(oracle, genomes, k, S)
    # F-a cache files: [f64 card][f64 x S], reference naming (fastxmerge.cpp:70-120)
    for i, p in enumerate(genomes):
        name = f"{pref}/{os.path.basename(p)}.rc_canon.sketchsize{S}.k{k}.SetSpace.DNA.opss"
        raw = np.fromfile(name, np.float64)
        assert raw[0] == ecards[i]
        np.testing.assert_array_equal(raw[1:].view(np.uint64), esigs[i].view(np.uint64))
    # second run loads the caches (inputs may even be gone) and must give the same bytes
    os.rename(genomes[0], genomes[0] + ".moved")
    try:
        _run(args + ["--cmpout", str(b2)])
    finally:
        os.rename(genomes[0] + ".moved", genomes[0])
    assert open(b1, "rb").read() == open(b2, "rb").read()
    # panel: rows = references, columns = queries (emitrect.cpp:211-247)
    dens = _densified(oracle, esigs)
    lib = oracle.load()
    import ctypes as C
    nf, nq = len(refs), len(qs)
    exp = np.empty((nf, nq), np.float32)
    for i in range(nf):
        for j in range(nq):
            exp[i, j] = lib.d2o_compare(dens.ctypes.data_as(C.POINTER(C.c_double)), ecards.ctypes.data_as(C.POINTER(C.c_double)),
                                        S, i, nf + j, oracle.SIMILARITY, k)
    np.testing.assert_array_equal(np.fromfile(b1, np.float32).view(np.uint32), exp.reshape(-1).view(np.uint32))
    # cmp --presketched over the individual cache files (multi-file loader, cmp_main.cpp:95-197)
    files = [f"{pref}/{os.path.basename(p)}.rc_canon.sketchsize{S}.k{k}.SetSpace.DNA.opss" for p in genomes]
    b3 = tmp_path / "m.bin"
    _run(["cmp", "--presketched", "-k", str(k), "--binary-output", "--cmpout", str(b3)] + files)
    exp_ut = oracle.allpairs_ut(dens, ecards, measure=oracle.SIMILARITY, k=k, nthreads=2)
    np.testing.assert_array_equal(np.fromfile(b3, np.float32).view(np.uint32), exp_ut.view(np.uint32))


def test_dist_tool_single_rank(oracle, genomes, tmp_path):
    """python -m dashing2_amd.dist cmp (launches the C++ CLI over every visible GPU: one here) == dashing2 cmp"""
    import sys
    k, S = 31, 256
    st = tmp_path / "s.bin"
    _run(["sketch", "-k", str(k), "-S", str(S), "-o", str(st)] + genomes)
    b1, b2 = tmp_path / "cli.bin", tmp_path / "dist.bin"
    _run(["cmp", "--presketched", "-k", str(k), "--binary-output", "--distance", "--cmpout", str(b1), str(st)])
    r = subprocess.run([sys.executable, "-m", "dashing2_amd.dist", "cmp", "--presketched", "-k", str(k), "--binary-output", "--distance",
                        "--cmpout", str(b2), str(st)], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(b1, "rb").read() == open(b2, "rb").read()


def test_cli_multiset_sketch_and_cmp(oracle, genomes, tmp_path):
    """BASELINE config 5 shape at test size: `dashing2 sketch --multiset` (exact k-mer counts ->
    BagMinHash on the GPU) then all-pairs: stacked file, names, .d2gbmh cache files and the distance
    matrix must be byte-identical to the oracle pipeline (src/fastxsketch.cpp:425-445 ;
    src/cmp_core.cpp:495-517 for the multiset-space compare)."""
    from oracle import textfmt
    k, S = 21, 256
    out = tmp_path / "ms.bin"
    phy = tmp_path / "ms.phylip"
    r = _run(["sketch", "--multiset", "-k", str(k), "-S", str(S), "-p", "4", "-o", str(out), "--cmpout", str(phy), "--phylip",
              "--cache", "--outprefix", str(tmp_path)] + genomes)
    esigs, ecards, _ = oracle.bmh_sketch_files(genomes, k, S)
    N = len(genomes)
    raw = np.fromfile(out, np.uint8)
    exp = np.concatenate([np.array([N, S], np.uint64).view(np.uint8), ecards.view(np.uint8), esigs.reshape(-1).view(np.uint8)])
    assert raw.tobytes() == exp.tobytes()
    # F-a cache naming for multiset sketches (fastxmerge.cpp:70-120, enums.cpp:28-47)
    for i, g in enumerate(genomes):
        dest = tmp_path / (os.path.basename(g) + f".rc_canon.sketchsize{S}.k{k}.ExactCounting.MultisetSpace.DNA.d2gbmh")
        blob = np.fromfile(dest, np.float64)
        assert blob[0] == ecards[i]
        np.testing.assert_array_equal(blob[1:].view(np.uint64), esigs[i].view(np.uint64))
    # multiset-space compare: from the equality count and the total weights
    neq = oracle.eqcounts_ut(esigs)
    iu = np.triu_indices(N, 1)
    exp_d = np.array([oracle.compare_from_neq(int(c), S, ecards[i], ecards[j], oracle.SIMILARITY, k)
                      for c, i, j in zip(neq, iu[0], iu[1])], np.float32)
    assert open(phy).read() == textfmt.render_symmetric(genomes, exp_d, phylip=True)
    assert exp_d.max() > 0.5
    # cmp --presketched picks the space up from the cache suffix; count threshold changes names and sketches
    b = tmp_path / "d.bin"
    caches = [str(tmp_path / (os.path.basename(g) + f".rc_canon.sketchsize{S}.k{k}.ExactCounting.MultisetSpace.DNA.d2gbmh")) for g in genomes]
    _run(["cmp", "--presketched", "-k", str(k), "--binary-output", "--cmpout", str(b)] + caches)
    np.testing.assert_array_equal(np.fromfile(b, np.float32).view(np.uint32), exp_d.view(np.uint32))
    out2 = tmp_path / "ms2.bin"
    _run(["sketch", "--multiset", "-m", "1", "-k", "7", "-S", "64", "-o", str(out2)] + genomes[:3])
    esigs2, ecards2, _ = oracle.bmh_sketch_files(genomes[:3], 7, 64, count_threshold=1.0)
    raw2 = np.fromfile(out2, np.uint8)
    exp2 = np.concatenate([np.array([3, 64], np.uint64).view(np.uint8), ecards2.view(np.uint8), esigs2.reshape(-1).view(np.uint8)])
    assert raw2.tobytes() == exp2.tobytes()
    # OPH min-count filtering stays out of scope
    r = subprocess.run([EXE, "sketch", "-m", "2", "-k", "21"] + genomes[:2], capture_output=True)
    assert r.returncode != 0 and b"outside this build" in r.stderr
    # interop guard: a stock dashing2 `.bmh` cache has the same layout but registers drawn by a
    # different BagMinHash; one matrix over both kinds is refused, `.bmh` among themselves is fine, and --cache never
    # picks a `.bmh` file up as its own
    import shutil
    stock = [c[:-len(".d2gbmh")] + ".bmh" for c in caches]
    for c, sname in zip(caches, stock):
        shutil.copy(c, sname)
    r = subprocess.run([EXE, "cmp", "--presketched", "-k", str(k), "--binary-output", "--cmpout", str(b), caches[0], stock[1]], capture_output=True)
    assert r.returncode != 0 and b"cannot compare stock dashing2 BagMinHash" in r.stderr
    _run(["cmp", "--presketched", "-k", str(k), "--binary-output", "--cmpout", str(b)] + stock)
    np.testing.assert_array_equal(np.fromfile(b, np.float32).view(np.uint32), exp_d.view(np.uint32))
    for c in caches:
        os.remove(c)
    with open(stock[0], "r+b") as f:                  # a poisoned stock cache: must not be read by sketch --cache
        f.seek(8)
        f.write(np.full(S, 123.0).tobytes())
    out3 = tmp_path / "ms3.bin"
    _run(["sketch", "--multiset", "-k", str(k), "-S", str(S), "-o", str(out3), "--cache", "--outprefix", str(tmp_path)] + genomes)
    assert np.fromfile(out3, np.uint8).tobytes() == exp.tobytes()
    assert all(os.path.exists(c) for c in caches)


def test_cli_wsketch(oracle, tmp_path):
    """`dashing2 wsketch` (src/wsketch.cpp:264-377): binary id / weight / indptr inputs, the reference's output
    files.  Registers, total weights and the sampled ids must equal the oracle's (BMH-D2G spec); the flag
    combinations that select ProbMinHash / FullSetSketch in the reference are refused."""
    rng = np.random.default_rng(12)
    S = 64
    n = 5000
    ids = rng.choice(1 << 40, n, replace=False).astype(np.uint64)
    w = np.round(rng.gamma(2.0, 3.0, n)) + 1.0
    w[::97] = 0.0                                            # ignored by update()
    (tmp_path / "ids.u64").write_bytes(ids.tobytes())
    (tmp_path / "w.f64").write_bytes(w.tobytes())
    (tmp_path / "w.f32").write_bytes(w.astype(np.float32).tobytes())
    (tmp_path / "ids.u32").write_bytes((ids & np.uint64(0xFFFFFFFF)).astype(np.uint32).tobytes())
    pos = np.arange(n, dtype=np.uint64)                      # update(i, w[i]): the sketch sees positions (wsketch.cpp:57-61)
    esig, etw, eown = oracle.bmh_from_weighted_ids(pos, w, S)
    # ---- two inputs (BagMinHash is the code's default here: wsketch.cpp:80-84)
    pref = str(tmp_path / "o1")
    r = _run(["wsketch", "-S", str(S), "-o", pref, str(tmp_path / "ids.u64"), str(tmp_path / "w.f64")])
    blob = np.fromfile(pref + ".sampled.hashes.f64", np.float64)
    assert blob[0] == etw
    np.testing.assert_array_equal(blob[1:].view(np.uint64), esig.view(np.uint64))
    np.testing.assert_array_equal(np.fromfile(pref + ".sampled.ids.u64", np.uint64), ids[eown])
    np.testing.assert_array_equal(np.fromfile(pref + ".sampled.indices.u64", np.uint64), esig.view(np.uint64))
    tail = chr((ord(";") + ord("d") + ord(";") + ord("L")) & 0xFF)
    exp_msg = "Total weight: %f;%s;%s%s\n" % (etw, tmp_path / "ids.u64", tmp_path / "w.f64", tail)
    assert open(pref + ".sampled.tw.txt", encoding="latin-1").read() == exp_msg
    assert exp_msg.encode("latin-1") in r.stderr
    # float32 weights + 32-bit ids, default out prefix = the id path
    _run(["wsketch", "-S", str(S), "-f", "-u", str(tmp_path / "ids.u32"), str(tmp_path / "w.f32")])
    blob = np.fromfile(str(tmp_path / "ids.u32") + ".sampled.hashes.f64", np.float64)
    np.testing.assert_array_equal(blob[1:].view(np.uint64), esig.view(np.uint64))      # the integer weights survive float32
    np.testing.assert_array_equal(np.fromfile(str(tmp_path / "ids.u32") + ".sampled.ids.u64", np.uint64), (ids & np.uint64(0xFFFFFFFF))[eown])
    # one input: unit weights
    pref = str(tmp_path / "o2")
    _run(["wsketch", "-S", str(S), "-o", pref, str(tmp_path / "ids.u64")])
    esig1, etw1, _ = oracle.bmh_from_weighted_ids(pos, None, S)
    blob = np.fromfile(pref + ".sampled.hashes.f64", np.float64)
    assert blob[0] == etw1 == n
    np.testing.assert_array_equal(blob[1:].view(np.uint64), esig1.view(np.uint64))
    # ---- three inputs (CSR; BagMinHash needs -B here: wsketch.cpp:148-150)
    indptr = np.array([0, 700, 700, 2500, 5000], np.uint64)  # one empty row
    (tmp_path / "ip.u64").write_bytes(indptr.tobytes())
    (tmp_path / "ip.u32").write_bytes(indptr.astype(np.uint32).tobytes())
    for ipf, flags in (("ip.u64", []), ("ip.u32", ["-P"])):
        pref = str(tmp_path / ("csr" + ipf))
        _run(["wsketch", "-B", "-S", str(S), "-o", pref] + flags + [str(tmp_path / "ids.u64"), str(tmp_path / "w.f64"), str(tmp_path / ipf)])
        regs = np.fromfile(pref + f".sampled.regs.stacked.4.{S}.f64", np.uint8)
        assert regs[:16].view(np.uint64).tolist() == [4, S]
        tws = regs[16:16 + 32].view(np.float64)
        sig = regs[48:].view(np.float64).reshape(4, S)
        samp = np.fromfile(pref + f".sampled.indices.stacked.4.{S}.i64", np.uint64).reshape(4, S)
        info = open(pref + ".sampled.info.txt").read().split()
        for i in range(4):
            lo, hi = int(indptr[i]), int(indptr[i + 1])
            es, et, eo = oracle.bmh_from_weighted_ids(np.arange(hi - lo, dtype=np.uint64), w[lo:hi], S)
            assert tws[i] == et and float(info[i]) == et
            np.testing.assert_array_equal(sig[i].view(np.uint64), es.view(np.uint64))
            if hi > lo:
                np.testing.assert_array_equal(samp[i], ids[lo:hi][eo])
            else:
                assert np.isinf(sig[i]).all() and (samp[i] == np.uint64(0xFFFFFFFFFFFFFFFF)).all()
        # the stacked register file is a valid `cmp --presketched` input
        b = tmp_path / "csr.bin"
        _run(["cmp", "--presketched", "--multiset", "--binary-output", "--cmpout", str(b), pref + f".sampled.regs.stacked.4.{S}.f64"])
        assert np.fromfile(b, np.float32).size == 6
    # unit weights through '-'
    pref = str(tmp_path / "csr_unit")
    _run(["wsketch", "-B", "-S", str(S), "-o", pref, str(tmp_path / "ids.u64"), "-", str(tmp_path / "ip.u64")])
    regs = np.fromfile(pref + f".sampled.regs.stacked.4.{S}.f64", np.uint8)
    assert regs[16:48].view(np.float64).tolist() == [700.0, 0.0, 1800.0, 2500.0]
    # out-of-scope selections say so
    for args in (["-B", str(tmp_path / "ids.u64")], ["-q", str(tmp_path / "ids.u64")],
                 [str(tmp_path / "ids.u64"), str(tmp_path / "w.f64"), str(tmp_path / "ip.u64")]):
        r = subprocess.run([EXE, "wsketch", "-S", "32"] + args, capture_output=True)
        assert r.returncode == 1 and b"outside this" in r.stderr
    r = subprocess.run([EXE, "wsketch"], capture_output=True)
    assert r.returncode == 1 and b"Required: between one and three positional arguments" in r.stderr


def test_cli_parse_by_seq(oracle, tmp_path):
    """--parse-by-seq (src/fastxsketchbyseq.cpp): one sketch per record of one file, names = record
    names, cardinality = exact distinct k-mer count below 10 S; OPH and multiset; stacked file,
    names file and PHYLIP byte-identical to the oracle pipeline."""
    from oracle import textfmt
    rng = np.random.default_rng(3)
    base = synth.random_genome(50, 30000)
    recs = [synth.fasta_bytes("base some comment", base), synth.fasta_bytes("mut1\tx", synth.mutate(base, 0.01, 1)),
            b">empty\n\n", b">short\nACGTACG\n", synth.fasta_bytes("mut5", synth.mutate(base, 0.05, 2)),
            synth.fasta_bytes("big", synth.random_genome(51, 300000)),      # estimate above 10 S: keeps getcard()
            b"@fq1\nACGTTGCAAGCTAGCTAGCTAGGATCGATCGATTTAGC\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n"]
    for i in range(40):
        recs.append(synth.fasta_bytes(f"read{i}", synth.random_genome(600 + i, int(rng.integers(20, 400)))))
    fa = tmp_path / "multi.fa"
    fa.write_bytes(b"".join(recs))
    buf = fa.read_bytes()
    for extra, multiset, k, S in ([], False, 21, 256), (["--multiset"], True, 15, 128):
        out = tmp_path / ("bs%d.bin" % multiset)
        phy = tmp_path / ("bs%d.phy" % multiset)
        _run(["sketch", "--parse-by-seq", "-k", str(k), "-S", str(S), "-p", "4", "-o", str(out), "--cmpout", str(phy), "--phylip"]
             + extra + [str(fa)])
        names, esigs, ecards = oracle.sketch_buffer_byseq(buf, k, S, multiset=multiset)
        N = len(names)
        assert N == len(recs) and names[0] == "base" and names[1] == "mut1" and names[6] == "fq1"
        raw = np.fromfile(out, np.uint8)
        exp = np.concatenate([np.array([N, S], np.uint64).view(np.uint8), ecards.view(np.uint8), esigs.reshape(-1).view(np.uint8)])
        assert raw.tobytes() == exp.tobytes()
        lines = open(str(out) + ".names.txt").read().splitlines()
        assert [l.split("\t")[0] for l in lines[1:]] == names
        if not multiset:
            assert ecards[2] == 0 and ecards[3] == 0 and ecards[6] == 18.0          # exact distinct counts (38 - 21 + 1 = 18 k-mers)
            assert (ecards[7:] == np.floor(ecards[7:])).all() and ecards[5] != float(int(ecards[5]))   # estimate kept above 10 S
            dens = _densified(oracle, esigs)
            dist = oracle.allpairs_ut(dens, ecards, measure=oracle.SIMILARITY, k=k, nthreads=2)
        else:
            neq = oracle.eqcounts_ut(esigs)
            iu = np.triu_indices(N, 1)
            dist = np.array([oracle.compare_from_neq(int(c), S, ecards[i], ecards[j], oracle.SIMILARITY, k)
                             for c, i, j in zip(neq, iu[0], iu[1])], np.float32)
        assert open(phy).read() == textfmt.render_symmetric(names, dist, phylip=True)
    r = subprocess.run([EXE, "sketch", "--parse-by-seq", str(fa), str(fa)], capture_output=True)
    assert r.returncode != 0 and b"only handles one file at a time" in r.stderr


def test_kmer_distinct_matches_oracle(gpu_ctx, d2g, oracle):
    g = [synth.fasta_bytes("a", synth.random_genome(3, 50000)), b">x\nACGTACGTACGTACGT\n", b"",
         synth.fasta_bytes("r", np.tile(synth.random_genome(4, 100), 50))]
    sp = d2g.SeqPack(11)
    for f in g:
        sp.add_fastx(f)
    got = gpu_ctx.kmer_distinct_seqpack(sp)
    exp = [oracle.kmer_count_buffer(f, 11)[0].size for f in g]
    assert got.tolist() == exp


def test_dist_sketch_tool_single_rank(genomes, tmp_path):
    """python -m dashing2_amd.dist sketch (world 1): same stacked file and names as the CLI itself"""
    import sys
    lst = tmp_path / "files.txt"
    lst.write_text("\n".join(genomes) + "\n")
    a, b = tmp_path / "a.bin", tmp_path / "b.bin"
    _run(["sketch", "-k", "21", "-S", "128", "-F", str(lst), "-o", str(a)])
    r = subprocess.run([sys.executable, "-m", "dashing2_amd.dist", "sketch", "-F", str(lst), "-o", str(b), "-k", "21", "-S", "128"],
                       capture_output=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert a.read_bytes() == b.read_bytes()
    assert open(str(a) + ".names.txt").read() == open(str(b) + ".names.txt").read()
    # the rank-per-GPU form (what runs under a launcher), world 1
    c = tmp_path / "c.bin"
    r = subprocess.run([sys.executable, "-m", "dashing2_amd.dist", "sketch", "-F", str(lst), "-o", str(c), "-k", "21", "-S", "128"],
                       capture_output=True, cwd=ROOT, env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert a.read_bytes() == c.read_bytes()


def test_cli_multiset_shapes_and_measures(oracle, genomes, tmp_path):
    """multiset space through the other dense shapes and every measure: asymmetric all-pairs (square)
    and -Q panel, binary outputs, expected from the oracle's multiset-space compare
    (src/cmp_core.cpp:495-517) pair by pair."""
    k, S = 17, 128
    esigs, ecards, _ = oracle.bmh_sketch_files(genomes, k, S)
    N = len(genomes)

    def cmp_ms(i, j, meas):
        neq = int((esigs[i] == esigs[j]).sum())
        return oracle.compare_from_neq(neq, S, ecards[i], ecards[j], meas, k)

    st = tmp_path / "ms.bin"
    _run(["sketch", "--multiset", "-k", str(k), "-S", str(S), "-o", str(st)] + genomes)
    for flags, meas in [([], oracle.SIMILARITY), (["--distance"], oracle.POISSON_LLR), (["--intersection"], oracle.INTERSECTION),
                        (["--containment"], oracle.CONTAINMENT), (["--symmetric-containment"], oracle.SYMMETRIC_CONTAINMENT),
                        (["--union-size"], oracle.UNION_SIZE)]:
        b = tmp_path / "sq.bin"
        _run(["cmp", "--multiset", "--presketched", "-k", str(k), "--binary-output", "--square", "--cmpout", str(b)] + flags + [str(st)])
        exp = np.array([[cmp_ms(i, j, meas) for j in range(N)] for i in range(N)], np.float32)
        np.testing.assert_array_equal(np.fromfile(b, np.float32).view(np.uint32), exp.reshape(-1).view(np.uint32), err_msg=str(flags))
    # panel straight from FASTA: rows = references, columns = queries
    refs, qs = genomes[:3], genomes[3:]
    ff, qf = tmp_path / "r.txt", tmp_path / "q.txt"
    ff.write_text("\n".join(refs) + "\n")
    qf.write_text("\n".join(qs) + "\n")
    b = tmp_path / "panel.bin"
    _run(["sketch", "--multiset", "-k", str(k), "-S", str(S), "-F", str(ff), "-Q", str(qf), "--binary-output", "--containment", "--cmpout", str(b)])
    exp = np.array([[cmp_ms(i, len(refs) + j, oracle.CONTAINMENT) for j in range(len(qs))] for i in range(len(refs))], np.float32)
    np.testing.assert_array_equal(np.fromfile(b, np.float32).view(np.uint32), exp.reshape(-1).view(np.uint32))


def test_cli_config1_shape(oracle, tmp_path):
    """BASELINE config 1's shape through the drop-in CLI: `dashing2 sketch --cmpout` on 32 synthetic 1 Mbp
    FASTA files, k = 31, S = 1024 (OPH default).  Stacked sketches, names and the PHYLIP / binary matrices
    must equal what the oracle produces from the same files."""
    from oracle import textfmt
    paths = []
    base = synth.random_genome(500, 1_000_000)
    for i in range(32):
        # 24 independent genomes + a family of 8 mutated copies, so that the matrix is not all zeros
        g = synth.random_genome(1000 + i, 1_000_000) if i < 24 else synth.mutate(base, 0.002 * (i - 23), seed=i)
        p = tmp_path / ("g%05d.fa" % i)
        synth.write_fasta(p, "g%05d" % i, g)
        paths.append(str(p))
    lst = tmp_path / "files.txt"
    lst.write_text("".join(p + "\n" for p in paths))
    out, phy, binm = tmp_path / "s.bin", tmp_path / "d.phylip", tmp_path / "d.bin"
    _run(["sketch", "-k", "31", "-S", "1024", "-p", "8", "-F", str(lst), "-o", str(out), "--cmpout", str(phy), "--phylip"])
    esigs, ecards = oracle.sketch_files(paths, k=31, S=1024, nthreads=8)
    raw = np.fromfile(out, np.uint8)
    exp = np.concatenate([np.array([32, 1024], np.uint64).view(np.uint8), ecards.view(np.uint8), esigs.reshape(-1).view(np.uint8)])
    assert raw.tobytes() == exp.tobytes()
    dist = oracle.allpairs_ut(_densified(oracle, esigs), ecards, measure=oracle.SIMILARITY, k=31, nthreads=8)
    assert open(phy).read() == textfmt.render_symmetric(paths, dist, phylip=True)
    assert (dist > 0.3).sum() >= 20 and (dist == 0).sum() > 300
    _run(["cmp", "--presketched", "-k", "31", "--binary-output", "--distance", "--cmpout", str(binm), str(out)])
    mash = oracle.allpairs_ut(_densified(oracle, esigs), ecards, measure=oracle.POISSON_LLR, k=31, nthreads=8)
    assert np.fromfile(binm, np.float32).tobytes() == mash.tobytes()


def test_cli_cmp_multi_gpu_loopback(oracle, genomes, tmp_path):
    """`dashing2 cmp` spread over several GPUs from one process (D2G_DEVICES): on this 1-GPU box the
    device list repeats device 0, which selects the loopback transport under the same row-sharded exchange.
    Outputs must be byte-identical to the single-GPU run for table-epilogue and host-epilogue measures, text and
    binary, set and multiset space."""
    k, S = 31, 512
    out = tmp_path / "s.bin"
    _run(["sketch", "-k", str(k), "-S", str(S), "-o", str(out)] + genomes)
    for flags in ([], ["--distance"], ["--containment"], ["--union-size"]):
        ref = tmp_path / "ref.bin"
        _run(["cmp", "--presketched", "-k", str(k), "--binary-output", "--cmpout", str(ref)] + flags + [str(out)])
        for devs in ("0,0", "0,0,0,0,0"):
            got = tmp_path / "got.bin"
            r = subprocess.run([EXE, "cmp", "--presketched", "-k", str(k), "--binary-output", "--cmpout", str(got), "-v"] + flags + [str(out)],
                               capture_output=True, env=dict(os.environ, D2G_DEVICES=devs))
            assert r.returncode == 0, r.stderr.decode()[-1500:]
            assert b"GPUs (loopback)" in r.stderr
            assert got.read_bytes() == ref.read_bytes(), (flags, devs)
    reft = _run(["cmp", "--presketched", "-k", str(k), "--phylip", str(out)]).stdout
    r = subprocess.run([EXE, "cmp", "--presketched", "-k", str(k), "--phylip", str(out)], capture_output=True, env=dict(os.environ, D2G_DEVICES="0,0,0"))
    assert r.returncode == 0 and r.stdout == reft
    # multiset space (count_eq branch of compare())
    ms = tmp_path / "ms.bin"
    _run(["sketch", "--multiset", "-k", "21", "-S", "256", "-o", str(ms)] + genomes)
    ref = _run(["cmp", "--presketched", "--multiset", "-k", "21", "--intersection", str(ms)]).stdout
    r = subprocess.run([EXE, "cmp", "--presketched", "--multiset", "-k", "21", "--intersection", str(ms)], capture_output=True,
                       env=dict(os.environ, D2G_DEVICES="0,0"))
    assert r.returncode == 0 and r.stdout == ref


def _write_stacked(path, sigs, cards):
    """format F-b (src/fastxsketch.cpp:236-240, src/sketch_core.cpp:130-140): [u64 N][u64 S][f64 card x N][f64 x N*S]"""
    N, S = sigs.shape
    with open(path, "wb") as f:
        np.array([N, S], np.uint64).tofile(f)
        np.ascontiguousarray(cards, np.float64).tofile(f)
        np.ascontiguousarray(sigs, np.float64).tofile(f)


def test_cli_float_text_of_large_values_both_fmt_generations(oracle, tmp_path):
    """--union-size / --intersection on real genomes exceed 1e7 routinely, where fmt < 11 (fixed
    notation below 1e16 -- what a 2.1.x-era dashing2 linked, and this CLI's default) and fmt >= 11 (exponent form from 1e7
    for float) print different text.  Both layouts are checked value for value against the oracle's renderer, the default
    carries no extra header line, and an explicit --fmt-compat is recorded in the TSV header."""
    from oracle import textfmt
    rng = np.random.default_rng(4)
    N, S, k = 9, 256, 31
    regs = synth.synthetic_registers(N, S, nclusters=2, seed=9, share_lo=0.3, share_hi=0.9)
    import dashing2_amd as D
    sigs, _ = D.oph_finalize(regs, S)
    cards = rng.uniform(2.0e6, 9.0e7, N)                       # genome-sized cardinalities: unions well above 1e7
    st = tmp_path / "big.bin"
    _write_stacked(st, sigs, cards)
    names = [str(i) for i in range(N)]
    for measure_flag, meas in (("--union-size", oracle.UNION_SIZE), ("--intersection", oracle.INTERSECTION)):
        exp = oracle.allpairs_ut(sigs, cards, measure=meas, k=k, nthreads=2)
        assert meas != oracle.UNION_SIZE or (exp >= 1e7).sum() > 10
        for compat, eu in ((None, 16), ("10", 16), ("11", 7)):
            flags = [measure_flag] + (["--fmt-compat", compat] if compat else [])
            got = _run(["cmp", "--presketched", "-k", str(k), "--phylip"] + flags + [str(st)]).stdout.decode()
            old = textfmt.EXP_UPPER
            textfmt.EXP_UPPER = eu
            try:
                want = textfmt.render_symmetric(names, exp, phylip=True)
                opt = "Dashing2Options;k:%d;parsebyfile;trimchr;sketchsize:%d;sketchtype:onepermsetsketch;Fastx;canon" % (k, S)
                want_tsv = textfmt.render_symmetric(names, exp, phylip=False, options_string=opt)
            finally:
                textfmt.EXP_UPPER = old
            assert got == want, (measure_flag, compat)
            tsv = _run(["cmp", "--presketched", "-k", str(k)] + flags + [str(st)]).stdout.decode()
            if compat:
                lines = tsv.split("\n")
                assert lines[2] == "#Dashing2FloatText: fmt-compat=" + compat
                tsv = "\n".join(lines[:2] + lines[3:])
            assert tsv == want_tsv, (measure_flag, compat)
    a = _run(["cmp", "--presketched", "-k", str(k), "--phylip", "--union-size", str(st)]).stdout
    b = _run(["cmp", "--presketched", "-k", str(k), "--phylip", "--union-size", "--fmt-compat", "11", str(st)]).stdout
    assert a != b and b"e+07" in b and b"e+07" not in a
    r = subprocess.run([EXE, "cmp", "--presketched", "--fmt-compat", "9", str(st)], capture_output=True)
    assert r.returncode == 1 and b"--fmt-compat takes 10" in r.stderr
    # the round-2 environment switch still selects the layout (with a deprecation warning) when --fmt-compat is absent
    r = subprocess.run([EXE, "cmp", "--presketched", "-k", str(k), "--phylip", "--union-size", str(st)], capture_output=True, env=dict(os.environ, D2_FMT_EXP_UPPER="7"))
    assert r.returncode == 0 and r.stdout == b and b"D2_FMT_EXP_UPPER=7 is deprecated" in r.stderr
    r = subprocess.run([EXE, "cmp", "--presketched", "-k", str(k), "--phylip", "--union-size", "--fmt-compat", "10", str(st)], capture_output=True,
                       env=dict(os.environ, D2_FMT_EXP_UPPER="7"))
    assert r.returncode == 0 and r.stdout == a and b"deprecated" not in r.stderr


def test_cli_cmp_multi_gpu_overflow_falls_back_to_one_gpu(tmp_path):
    """with D2G_DEVICES the bit-sliced prepare's overflow status was never seen and the distances came out wrong
    silently.  Every rank's status word now travels with its groups; `cmp` checks it and takes the single-GPU path (whose AUTO
    algorithm falls back to the direct kernel).  D2G_BS_TAGBITS=0 (test hook) forces the overflow."""
    rng = np.random.default_rng(3)
    N, S = 2500, 128
    sigs = rng.random((N, S))
    sigs[:, ::2] = rng.random((40, S // 2))[rng.integers(0, 40, N)]      # some equalities, many distinct values per column
    st = tmp_path / "adv.bin"
    _write_stacked(st, sigs, np.ones(N))
    ref = tmp_path / "ref.bin"
    _run(["cmp", "--presketched", "-k", "31", "--binary-output", "--cmpout", str(ref), str(st)])
    got = tmp_path / "got.bin"
    r = subprocess.run([EXE, "cmp", "--presketched", "-k", "31", "--binary-output", "--cmpout", str(got), str(st)],
                       capture_output=True, env=dict(os.environ, D2G_DEVICES="0,0,0", D2G_BS_TAGBITS="0"))
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    assert b"falling back to one GPU" in r.stderr
    assert got.read_bytes() == ref.read_bytes()


def test_cli_cache_names_with_count_threshold(genomes, tmp_path):
    """cache file names follow src/fastxmerge.cpp:70-120; `-m c` adds `.ct_threshold<int>` -- the reference parses -m with atoi
    into a uint32_t (options.h:352, d2.h:103), so `-m 2.7` is 2 there and here and the fractional branch at fastxmerge.cpp:93
    is unreachable from its CLI"""
    import shutil
    d = tmp_path / "fa"
    d.mkdir()
    g = d / "a.fa"
    shutil.copy(genomes[0], g)
    for m, tag in (("3", ".ct_threshold3"), ("2.7", ".ct_threshold2")):
        _run(["sketch", "--multiset", "-k", "21", "-S", "64", "-m", m, "--cache", str(g)])
        want = str(g) + ".rc_canon.sketchsize64.k21" + tag + ".ExactCounting.MultisetSpace.DNA.d2gbmh"
        assert os.path.exists(want), sorted(os.listdir(d))
        os.remove(want)


def test_cli_cmp_batches_are_invisible(genomes, tmp_path):
    """`cmp` now runs device batch i+1 under the emit of batch i (three pinned slots, an emitter thread).
    The batching may not change a byte: tiny slots (many batches, D2G_CMP_SLOT_VALUES) against the default, for every shape,
    text and binary, table-epilogue and host-epilogue measures."""
    rng = np.random.default_rng(21)
    N, S = 157, 128
    sigs = rng.random((12, S))[rng.integers(0, 12, (N, S)), np.arange(S)[None, :]]
    cards = rng.uniform(1e3, 1e6, N)
    st = tmp_path / "m.bin"
    _write_stacked(st, sigs, cards)
    shapes = [[], ["--phylip"], ["--asymmetric-all-pairs"]]
    for shape in shapes:
        for meas in ([], ["--containment"], ["--distance"]):
            for binary in ([], ["--binary-output"]):
                if binary and shape == ["--phylip"]:
                    continue
                outs = []
                for env in ({}, {"D2G_CMP_SLOT_VALUES": "999"}, {"D2G_CMP_SLOT_VALUES": "1"}):
                    o = tmp_path / "o.out"
                    r = subprocess.run([EXE, "cmp", "--presketched", "-k", "31", "-p", "4", "--cmpout", str(o)] + shape + meas + binary + [str(st)],
                                       capture_output=True, env=dict(os.environ, **env))
                    assert r.returncode == 0, r.stderr.decode()[-800:]
                    outs.append(o.read_bytes())
                assert all(x == outs[0] for x in outs[1:]), (shape, meas, binary)
                if binary:
                    nv = N * N if shape else N * (N - 1) // 2
                    assert len(outs[0]) == 4 * nv


def test_cli_sketch_device_parser_equals_host_parser(genomes, tmp_path):
    """With D2G_DEVICE_PARSE=1 `dashing2 sketch` parses plain FASTA on the device (K0) and everything else -- gz members, FASTQ,
    leading junk -- with the host parser, group by group.  The stacked sketches of a mixed input list must be byte-identical
    to the default all-host run, for set and multiset sketches; the verbose line must show both kinds of groups at work."""
    import gzip
    d = tmp_path / "mix"
    d.mkdir()
    paths = list(genomes)
    gz = d / "g0.fa.gz"
    gz.write_bytes(gzip.compress(open(genomes[0], "rb").read()))
    fq = d / "reads.fq"
    rng = np.random.default_rng(2)
    with open(fq, "wb") as f:
        for i in range(300):
            s = bytes(synth.random_genome(1000 + i, 150))
            f.write(b"@r%d\n" % i + s + b"\n+\n" + b"I" * 150 + b"\n")
    plus = d / "plus.fa"                                              # starts like FASTA, turns FASTQ-like: refused on the device, host takes it
    plus.write_bytes(open(genomes[1], "rb").read() + b"+\nIIII\n")
    empty = d / "empty.fa"
    empty.write_bytes(b"")
    two = genomes[2] + " " + genomes[3]                               # one sketch from two files (an input line with a space)
    paths += [str(gz), str(fq), str(plus), str(empty)]
    lst = tmp_path / "l.txt"
    lst.write_text("".join(p + "\n" for p in paths) + two + "\n")
    for extra in ([], ["--multiset", "-k", "21", "-S", "256"]):
        outs = []
        for env in ({"D2G_GROUP_BYTES": "100000", "D2G_DEVICE_PARSE": "1"}, {}):    # device parser, one input per group / the default: host parser
            o = tmp_path / "s.bin"
            r = subprocess.run([EXE, "sketch", "-v", "-p", "3", "-F", str(lst), "-o", str(o)] + extra, capture_output=True, env=dict(os.environ, **env))
            assert r.returncode == 0, r.stderr.decode()[-1500:]
            outs.append((o.read_bytes(), open(str(o) + ".names.txt", "rb").read(), r.stderr.decode()))
        assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1]
        assert "parsed on the device" in outs[0][2] and "(0 parsed on the device" not in outs[0][2] and " 0 by the host" not in outs[0][2]
        assert "(0 parsed on the device" in outs[1][2]
    r = subprocess.run([EXE, "sketch", "-o", str(tmp_path / "x.bin"), str(d / "missing.fa")], capture_output=True)
    assert r.returncode != 0 and b"Failed to open" in r.stderr


def test_cli_sketch_multi_gpu_loopback(genomes, tmp_path):
    """`D2G_DEVICES=... dashing2 sketch` -- every listed GPU gets its pair of device threads (own context + sketcher),
    all of them take input groups from the one queue (file-sharded, no collectives); results land by input index.  On this box the
    list repeats device 0.  Stacked sketches, names and the --cmpout matrix must be byte-identical to the single-GPU run, for OPH and
    --multiset sketches, with one input per group so that every thread gets work."""
    lst = tmp_path / "l.txt"
    lst.write_text("".join(p + "\n" for p in genomes * 3))
    for extra in (["-k", "31", "-S", "512"], ["--multiset", "-k", "21", "-S", "256"]):
        outs = []
        for env in ({}, {"D2G_DEVICES": "0,0,0", "D2G_GROUP_BYTES": "1000"}, {"D2G_DEVICES": "0,0", "D2G_DEVICE_THREADS": "1", "D2G_GROUP_BYTES": "300000"}):
            o, c = tmp_path / "s.bin", tmp_path / "c.phy"
            r = subprocess.run([EXE, "sketch", "-v", "-p", "4", "-F", str(lst), "-o", str(o), "--cmpout", str(c), "--phylip"] + extra,
                               capture_output=True, env=dict(os.environ, **env))
            assert r.returncode == 0, r.stderr.decode()[-1500:]
            outs.append((o.read_bytes(), open(str(o) + ".names.txt", "rb").read(), c.read_bytes(), r.stderr.decode()))
        for x in outs[1:]:
            assert x[:3] == outs[0][:3], extra
        assert "6 device threads over 3 GPU(s)" in outs[1][3] and "2 device threads over 2 GPU(s)" in outs[2][3]
        assert "GPUs (loopback)" in outs[1][3]                               # the --cmpout half went over the same device list


def test_cli_cmp_multi_gpu_square_and_panel(genomes, tmp_path):
    """--square and -Q panel shapes spread over several GPUs (rows of the rectangle dealt round-robin over the gathered
    operand): byte-identical to the single-GPU output, text and binary, table-epilogue and card-dependent measures, many tiny batches."""
    k, S = 31, 256
    out = tmp_path / "s.bin"
    _run(["sketch", "-k", str(k), "-S", str(S), "-o", str(out)] + genomes)
    fl, ql = tmp_path / "f.txt", tmp_path / "q.txt"
    fl.write_text("".join(p + "\n" for p in genomes[:5]))
    ql.write_text("".join(p + "\n" for p in genomes[4:]))
    jobs = [["cmp", "--presketched", "-k", str(k), "--asymmetric-all-pairs", str(out)],
            ["cmp", "--presketched", "-k", str(k), "--square", "--binary-output", "--containment", str(out)],
            ["sketch", "-k", str(k), "-S", str(S), "-F", str(fl), "-Q", str(ql)],
            ["sketch", "-k", str(k), "-S", str(S), "-F", str(fl), "-Q", str(ql), "--distance", "--binary-output"],
            ["sketch", "-k", str(k), "-S", str(S), "-F", str(fl), "-Q", str(ql), "--symmetric-containment"]]
    for job in jobs:
        ref = tmp_path / "ref.out"
        _run(job[:1] + ["--cmpout", str(ref)] + job[1:])
        for env in ({"D2G_DEVICES": "0,0,0"}, {"D2G_DEVICES": "0,0", "D2G_CMP_SLOT_VALUES": "3"}):
            got = tmp_path / "got.out"
            r = subprocess.run([EXE] + job[:1] + ["-v", "--cmpout", str(got)] + job[1:], capture_output=True, env=dict(os.environ, **env))
            assert r.returncode == 0, r.stderr.decode()[-1500:]
            assert b"GPUs (loopback)" in r.stderr, r.stderr.decode()[-800:]
            assert got.read_bytes() == ref.read_bytes(), (job, env)
    # what still runs on one GPU says so WITHOUT -v: a sketch size that is not a power of two (gt / lt counts need the raw registers)
    o2 = tmp_path / "s100.bin"
    _run(["sketch", "-k", str(k), "-S", "100", "-o", str(o2)] + genomes)
    ref = _run(["cmp", "--presketched", "-k", str(k), str(o2)]).stdout
    r = subprocess.run([EXE, "cmp", "--presketched", "-k", str(k), str(o2)], capture_output=True, env=dict(os.environ, D2G_DEVICES="0,0"))
    assert r.returncode == 0 and r.stdout == ref and b"D2G_DEVICES ignored for this job" in r.stderr


def test_cli_gpu_stats_json(genomes, tmp_path):
    """VERDICT r3 #7 / SURVEY 5 "Metrics": --gpu-stats FILE writes one JSON object per run -- device, HIP-event milliseconds per kernel
    family, bit-plane counts, algorithmic bytes, wall phases -- for sketch (+ --cmpout), cmp, and cmp over several GPUs; the outputs of
    the run are the same bytes with and without it."""
    import json
    k, S = 31, 256
    st, o, c = tmp_path / "st.json", tmp_path / "s.bin", tmp_path / "c.bin"
    _run(["sketch", "-k", str(k), "-S", str(S), "-o", str(o), "--cmpout", str(c), "--binary-output", "--gpu-stats", str(st)] + genomes)
    j = json.loads(st.read_text())
    assert j["command"] == "sketch" and j["in_process_s"] > 0 and j["context"]["create_s"] > 0
    assert isinstance(j["context"]["switches"], dict)                 # the D2G_* switches the context resolved (d2g_ctx_tuning)
    sk = j["sketch"]
    assert sk["inputs"] == len(genomes) and sk["k"] == k and sk["sketchsize"] == S and sk["bases"] > 500000
    assert sk["devices"][0]["k1"]["launches"] >= 1 and sk["devices"][0]["k1"]["total_ms"] > 0 and "gfx950" in sk["devices"][0]["name"]
    assert sk["algorithmic_bytes"] == (sk["bases"] + 3) // 4 + len(genomes) * 8 * S
    cm = j["cmp"]
    n = len(genomes)
    assert cm["sketches"] == n and cm["values"] == n * (n - 1) // 2 and cm["algo"] == "bitslice" and cm["bit_planes"]["max"] >= 1
    assert cm["devices"][0]["k2"]["launches"] >= 1 and cm["devices"][0]["k2prep"]["launches"] >= 1
    # 7 sketches: far below the size from which tiles are listed -- unless the environment forces the sparse path on (the second GPU run of the suite)
    assert cm["sparse_tiles"]["sorted_operand"] is (os.environ.get("D2G_BS_SPARSE_MIN_N") == "1")
    assert cm["algorithmic_bytes"] == 8 * S * n + 4 * cm["values"]
    ref_s, ref_c = o.read_bytes(), c.read_bytes()
    _run(["sketch", "-k", str(k), "-S", str(S), "-o", str(o), "--cmpout", str(c), "--binary-output"] + genomes)
    assert (o.read_bytes(), c.read_bytes()) == (ref_s, ref_c)
    # multiset sketch: K3's time is reported; cmp over three (loopback) GPUs: one entry per device
    _run(["sketch", "--multiset", "-k", "21", "-S", "128", "-o", str(tmp_path / "m.bin"), "--gpu-stats", str(st)] + genomes)
    j = json.loads(st.read_text())
    assert j["sketch"]["devices"][0]["k3"]["launches"] >= 1 and "cmp" not in j
    r = subprocess.run([EXE, "cmp", "--presketched", "-k", str(k), "--cmpout", str(c), "--binary-output", "--gpu-stats", str(st), str(o)],
                       capture_output=True, env=dict(os.environ, D2G_DEVICES="0,0,0"))
    assert r.returncode == 0 and c.read_bytes() == ref_c
    j = json.loads(st.read_text())
    assert j["command"] == "cmp" and len(j["cmp"]["devices"]) == 3 and j["cmp"]["transport"] == "loopback"
    assert sum(d["k2"]["launches"] for d in j["cmp"]["devices"]) >= 1 and all(d["k2prep"]["launches"] >= 1 for d in j["cmp"]["devices"])
    # an unknown flag is still rejected like the reference does (options.h:290-304)
    r = subprocess.run([EXE, "cmp", "--gpu-statz", "x", str(o)], capture_output=True)
    assert r.returncode != 0 and b"not found in expected set" in r.stderr

"""Multi-GPU K2 through the C ABI (include/d2g.h: d2g_comm_*, d2g_allpairs_*, d2g_bcast_sigs) on ONE GPU:
W contexts on device 0 form a loopback communicator group that runs exactly the send/recv lists the RCCL
transport runs between devices, so the row-sharded exchange, the sharded prepare, the gathered operand and
every rank's slab are checked against the oracle for world sizes 1..5 and shapes no divisibility rule fits.
A real RCCL communicator is initialised too (world 1: the only clique a 1-GPU box can form)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _planted(rng, N, S, nvals=5, zero_frac=0.05):
    vals = rng.random((nvals, S))
    m = vals[rng.integers(0, nvals, (N, S)), np.arange(S)[None, :]]
    m[rng.random((N, S)) < zero_frac] = 0.0
    return m


def _upload(ctx, arr):
    p = ctx.malloc(max(arr.nbytes, 8))
    if arr.nbytes:
        ctx.h2d(p, np.ascontiguousarray(arr))
    return p


@pytest.mark.parametrize("W,N,S", [(1, 300, 256), (2, 263, 1000), (3, 517, 96), (4, 1024, 1024), (5, 129, 100), (8, 40, 64), (3, 2, 32)])
def test_allpairs_loopback_vs_oracle(d2g, oracle, W, N, S):
    rng = np.random.default_rng(W * 1000 + N + S)
    sigs = _planted(rng, N, S, nvals=int(rng.integers(2, 7)))
    bits = sigs.view(np.uint64)
    exp = oracle.eqcounts_ut(sigs)
    ctxs = [d2g.Context(0) for _ in range(W)]
    comms = d2g.Comm.create_all(ctxs)
    assert [c.rank for c in comms] == list(range(W)) and all(c.world == W for c in comms)
    assert not any(c.is_rccl for c in comms)                       # same device: loopback transport
    engs = [d2g.AllPairs(ctxs[r], comms[r], N, S) for r in range(W)]
    # rows held: contiguous cover of [0, N), sizes differ by at most one; rows computed: the pair-balanced partition
    held = [e.rows_held for e in engs]
    assert held[0][0] == 0 and held[-1][1] == N and all(held[i][1] == held[i + 1][0] for i in range(W - 1))
    sizes = [b - a for a, b in held]
    assert max(sizes) - min(sizes) <= 1
    b = d2g.ut_partition(N, W)
    assert [e.rows_computed for e in engs] == [(b[r], b[r + 1]) for r in range(W)]
    rows = [_upload(ctxs[r], bits[held[r][0]:held[r][1]]) for r in range(W)]
    outs = [ctxs[r].malloc(max(d2g.ut_count(N, *engs[r].rows_computed), 1) * 4) for r in range(W)]
    off = np.concatenate([[0], np.cumsum(N - 1 - np.arange(N, dtype=np.int64))])
    for rep in range(2):                                            # the second step reuses every buffer
        d2g.allpairs_step_all(engs, rows, None, outs)
        for r in range(W):
            r0, r1 = engs[r].rows_computed
            got = np.empty(d2g.ut_count(N, r0, r1), np.uint32)
            ctxs[r].sync()
            if got.size:
                ctxs[r].d2h(got, outs[r])
            np.testing.assert_array_equal(got, exp[off[r0]:off[r1]], err_msg=f"rank {r} rep {rep}")
    # every rank holds the WHOLE operand afterwards: any row range, any rectangle (what the CLI's round-robin uses)
    full = engs[W - 1].operand()
    np.testing.assert_array_equal(full.eqcount_ut(), exp)
    if N > 10:
        blk = full.eqcount_rect(1, N // 2, N // 3, N)
        ref = np.zeros((N, N), np.uint32)
        ref[np.triu_indices(N, 1)] = exp
        ref = ref + ref.T + np.diag(np.full(N, S, np.uint32))
        np.testing.assert_array_equal(blk, ref[1:N // 2, N // 3:N])
    # fused float epilogue on every rank's slab
    lut = d2g.epilogue_lut(S, d2g.POISSON_LLR, 21, multiset_space=True)
    luts = [_upload(ctxs[r], lut) for r in range(W)]
    d2g.allpairs_step_all(engs, rows, luts, outs)
    for r in range(W):
        r0, r1 = engs[r].rows_computed
        got = np.empty(d2g.ut_count(N, r0, r1), np.float32)
        ctxs[r].sync()
        if got.size:
            ctxs[r].d2h(got, outs[r])
        np.testing.assert_array_equal(got.view(np.uint32), lut[exp[off[r0]:off[r1]]].view(np.uint32))
    for r in range(W):
        for p in (rows[r], outs[r], luts[r]):
            ctxs[r].free(p)
    for e in engs:
        e.close()
    for c in comms:
        c.close()
    for c in ctxs:
        c.close()


def test_allpairs_pipelined_enqueue(d2g, oracle, gpu_ctx):
    """d2g_allpairs_enqueue_lut_dev over a stream of DIFFERENT matrices (world 1): step i+1's exchange + prepare run
    on the engine's stream over the other operand buffer while step i's pair kernel is busy; results, copied out
    on the caller's stream right after each call, must equal the plain per-step results.  The input of every
    step is produced on the caller's stream immediately before the call (input_ready = 0)."""
    import torch
    rng = np.random.default_rng(3)
    N, S = 640, 1024
    dev = torch.device("cuda", 0)
    comm = d2g.Comm.create(gpu_ctx)
    eng = d2g.AllPairs(gpu_ctx, comm, N, S)
    lut = torch.from_numpy(d2g.epilogue_lut(S, d2g.SIMILARITY, 31)).to(dev)
    out = torch.empty(N * (N - 1) // 2, dtype=torch.float32, device=dev)
    inputs = [_planted(rng, N, S, nvals=3 + i) for i in range(6)]
    staged = [torch.from_numpy(x.view(np.int64)).to(dev) for x in inputs]
    rows = torch.zeros((N, S), dtype=torch.int64, device=dev)
    big = torch.zeros(32 << 20, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    st = torch.cuda.current_stream().cuda_stream
    got = []
    for x in staged:
        big.add_(1)                                    # a slow producer on the caller's stream, then the real input
        rows.copy_(x)
        eng.enqueue_lut_dev(rows.data_ptr(), lut.data_ptr(), out.data_ptr(), st, input_ready=False)
        got.append(out.clone())
    torch.cuda.synchronize()
    for x, g in zip(inputs, got):
        exp = oracle.allpairs_ut(x, np.ones(N), measure=oracle.SIMILARITY, k=31, nthreads=4)
        np.testing.assert_array_equal(g.cpu().numpy().view(np.uint32), exp.view(np.uint32))
    eng.close()
    comm.close()


def test_rccl_communicator_world1_and_bcast(d2g, oracle, gpu_ctx):
    """RCCL itself, loaded by libd2g (dlopen) and initialised through the C ABI: unique id + ncclCommInitRank with one
    member -- the clique a 1-GPU box can form -- then the engine over that communicator; d2g_bcast_sigs in its
    one-context form."""
    uid = d2g.comm_unique_id()
    assert len(uid) == d2g.capi.COMM_ID_BYTES and any(uid)
    comm = d2g.Comm.create(gpu_ctx, rank=0, world=1, unique_id=uid)
    assert comm.is_rccl and comm.rank == 0 and comm.world == 1
    rng = np.random.default_rng(9)
    N, S = 200, 128
    sigs = _planted(rng, N, S)
    ptrs = d2g.bcast_sigs([gpu_ctx], [comm], sigs.view(np.uint64))
    back = np.empty((N, S), np.uint64)
    gpu_ctx.d2h(back, ptrs[0])
    np.testing.assert_array_equal(back, sigs.view(np.uint64))
    eng = d2g.AllPairs(gpu_ctx, comm, N, S)
    out = gpu_ctx.malloc(N * (N - 1) // 2 * 4)
    eng.step_eqcount_dev(ptrs[0], out)
    got = np.empty(N * (N - 1) // 2, np.uint32)
    gpu_ctx.sync()
    gpu_ctx.d2h(got, out)
    np.testing.assert_array_equal(got, oracle.eqcounts_ut(sigs))
    gpu_ctx.free(out)
    gpu_ctx.free(ptrs[0])
    eng.close()
    comm.close()


def _one_step_counts(d2g, ctxs, comms, bits, N, S):
    """one eqcount step of a W-rank loopback group; returns (engines, per-rank outputs as numpy)"""
    W = len(ctxs)
    engs = [d2g.AllPairs(ctxs[r], comms[r], N, S) for r in range(W)]
    held = [e.rows_held for e in engs]
    rows = [_upload(ctxs[r], bits[held[r][0]:held[r][1]]) for r in range(W)]
    outs = [ctxs[r].malloc(max(d2g.ut_count(N, *engs[r].rows_computed), 1) * 4) for r in range(W)]
    d2g.allpairs_step_all(engs, rows, None, outs)
    got = []
    for r in range(W):
        r0, r1 = engs[r].rows_computed
        g = np.empty(d2g.ut_count(N, r0, r1), np.uint32)
        ctxs[r].sync()
        if g.size:
            ctxs[r].d2h(g, outs[r])
        got.append(g)
    for r in range(W):
        ctxs[r].free(rows[r])
        ctxs[r].free(outs[r])
    return engs, got


@pytest.mark.parametrize("chunks", ["1", "2", "3", "4"])
@pytest.mark.parametrize("W,N,S", [(2, 301, 1000), (3, 200, 512), (8, 77, 1024), (4, 64, 96)])
def test_allpairs_chunked_exchange(d2g, oracle, monkeypatch, chunks, W, N, S):
    """Inside one step a rank's column slice travels and is prepared in C chunks (the exchange of chunk c+1 under the
    prepare of chunk c); the gathered operand keeps its groups in chunk-major order.  Every chunk count, including ones
    that leave some (rank, chunk) blocks without a register group, must count exactly like the oracle."""
    monkeypatch.setenv("D2G_MGPU_CHUNKS", chunks)
    rng = np.random.default_rng(int(chunks) * 7 + W + N)
    sigs = _planted(rng, N, S, nvals=int(rng.integers(2, 9)))
    exp = oracle.eqcounts_ut(sigs)
    off = np.concatenate([[0], np.cumsum(N - 1 - np.arange(N, dtype=np.int64))])
    ctxs = [d2g.Context(0) for _ in range(W)]
    comms = d2g.Comm.create_all(ctxs)
    engs, got = _one_step_counts(d2g, ctxs, comms, sigs.view(np.uint64), N, S)
    assert all(e.chunks == int(chunks) for e in engs)
    for r in range(W):
        r0, r1 = engs[r].rows_computed
        np.testing.assert_array_equal(got[r], exp[off[r0]:off[r1]], err_msg=f"rank {r}")
        engs[r].status()                                              # no overflow anywhere
    np.testing.assert_array_equal(engs[0].operand().eqcount_ut(), exp)
    for x in engs + comms + ctxs:
        x.close()


def test_allpairs_default_chunks_follow_the_shape(d2g):
    """chunks per rank = groups per rank / 2, clamped to [1, 4]; one rank never chunks"""
    want = {(1, 1024): 1, (2, 1024): 4, (4, 1024): 4, (8, 1024): 2, (8, 256): 1, (3, 2048): 4}
    for (W, S), c in want.items():
        ctxs = [d2g.Context(0) for _ in range(W)]
        comms = d2g.Comm.create_all(ctxs)
        engs = [d2g.AllPairs(ctxs[r], comms[r], 100, S) for r in range(W)]
        assert [e.chunks for e in engs] == [c] * W, (W, S)
        for x in engs + comms + ctxs:
            x.close()


def test_allpairs_status_reaches_every_rank(d2g, oracle, monkeypatch):
    """the sharded prepare's overflow status used to stay private to the rank that raised it.  Every block's
    status word now travels with its groups: a rank whose own slice is harmless still learns that the operand is invalid.
    D2G_BS_TAGBITS=0 (test hook) makes every occupied slot a candidate, so a column with many distinct values overflows
    the fix list of the rank kernel; constant columns never do."""
    monkeypatch.setenv("D2G_BS_TAGBITS", "0")
    rng = np.random.default_rng(12)
    N, S, W = 3000, 128, 2
    sigs = np.empty((N, S))
    sigs[:, :64] = rng.random((N, 64))                                # rank 0's slice: all distinct -> overflow
    sigs[:, 64:] = rng.random(64)[None, :]                            # rank 1's slice: constant columns -> clean
    ctxs = [d2g.Context(0) for _ in range(W)]
    comms = d2g.Comm.create_all(ctxs)
    engs, _ = _one_step_counts(d2g, ctxs, comms, sigs.view(np.uint64), N, S)
    for e in engs:
        with pytest.raises(d2g.D2GError):
            e.status()
        with pytest.raises(d2g.D2GError):
            e.operand().status()
    monkeypatch.delenv("D2G_BS_TAGBITS")
    for c in ctxs:
        c.reload_tuning()                                             # a context reads its switches when it is created; these were created under the hook
    # the same engines recover on the next (clean) step: the status words are rewritten by every prepare
    held = [e.rows_held for e in engs]
    rows = [_upload(ctxs[r], sigs.view(np.uint64)[held[r][0]:held[r][1]]) for r in range(W)]
    d2g.allpairs_prepare_all(engs, rows)
    for e in engs:
        e.status()
    exp = oracle.eqcounts_ut(sigs[:400])
    np.testing.assert_array_equal(engs[1].operand().eqcount_ut(0, 1)[:399], exp[:399])
    for r in range(W):
        ctxs[r].free(rows[r])
    for x in engs + comms + ctxs:
        x.close()


def test_allpairs_plain_and_pipelined_steps_mix(d2g, oracle, gpu_ctx):
    """plain and pipelined steps share the send/receive buffers and the exporter sets; each form now waits
    (events) for what the other still has in flight, so they can be interleaved without a device synchronisation."""
    rng = np.random.default_rng(31)
    N, S = 700, 256
    mats = [_planted(rng, N, S, nvals=3 + i) for i in range(4)]
    exps = [oracle.eqcounts_ut(m) for m in mats]
    lut = np.arange(S + 1, dtype=np.float32)                          # value = neq as a float
    comm = d2g.Comm.create(gpu_ctx, 0, 1)
    eng = d2g.AllPairs(gpu_ctx, comm, N, S)
    rows = [_upload(gpu_ctx, m.view(np.uint64)) for m in mats]
    outs = [gpu_ctx.malloc(N * (N - 1) // 2 * 4) for _ in mats]
    lut_d = _upload(gpu_ctx, lut)
    eng.enqueue_lut_dev(rows[0], lut_d, outs[0], None, input_ready=True)
    eng.step_lut_dev(rows[1], lut_d, outs[1], None)                   # plain right behind a pipelined one
    eng.enqueue_lut_dev(rows[2], lut_d, outs[2], None, input_ready=True)   # pipelined right behind a plain one
    eng.enqueue_lut_dev(rows[3], lut_d, outs[3], None, input_ready=True)
    gpu_ctx.sync()
    for i in range(4):
        got = np.empty(N * (N - 1) // 2, np.float32)
        gpu_ctx.d2h(got, outs[i])
        np.testing.assert_array_equal(got.astype(np.uint32), exps[i], err_msg=f"step {i}")
    for p in rows + outs + [lut_d]:
        gpu_ctx.free(p)
    eng.close()
    comm.close()


def test_allpairs_config4_shape_world8_vs_single_gpu(d2g, gpu_ctx):
    """BASELINE config 4's 8-rank form (N = 50 000, S = 1024: 2 chunks per rank, the split multi-partition rank kernel on
    64-column chunks, the column plan per chunk) through the loopback transport on one GPU: every rank's slab of the
    condensed triangle must equal the single-GPU result on the same rows -- first rows, rows around every partition seam,
    last rows -- for the fused float epilogue and the integer counts."""
    from dashing2_amd import synth
    N, S, W = 50_000, 1024, 8
    regs = synth.synthetic_registers(N, S, nclusters=N // 150, seed=20260929)
    bits = d2g.oph_finalize(regs, S, nthreads=16)[0].view(np.uint64)
    del regs
    lut = d2g.epilogue_lut(S, d2g.SIMILARITY, 31)
    ref = gpu_ctx.cmp_set(bits, algo=d2g.CMP_BITSLICE)
    ctxs = [d2g.Context(0) for _ in range(W)]
    comms = d2g.Comm.create_all(ctxs)
    engs = [d2g.AllPairs(ctxs[r], comms[r], N, S) for r in range(W)]
    assert all(e.chunks == 2 for e in engs)
    held = [e.rows_held for e in engs]
    rows = [_upload(ctxs[r], bits[held[r][0]:held[r][1]]) for r in range(W)]
    outs = [ctxs[r].malloc(max(d2g.ut_count(N, *engs[r].rows_computed), 1) * 4) for r in range(W)]
    luts = [_upload(ctxs[r], lut) for r in range(W)]
    d2g.allpairs_step_all(engs, rows, luts, outs)
    for r in range(W):
        ctxs[r].sync()
        engs[r].status()
    b = d2g.ut_partition(N, W)
    off = lambda r0, i: d2g.ut_count(N, r0, i)                       # offset of row i inside a slab that starts at row r0
    for r in range(W):
        r0, r1 = engs[r].rows_computed
        assert (r0, r1) == (b[r], b[r + 1])
        for a, z in ((r0, min(r0 + 3, r1)), (max(r0, r1 - 3), r1)):  # the first and the last rows of the slab (the seams)
            if z <= a:
                continue
            want = ref.lut_ut(lut, a, z) if hasattr(ref, "lut_ut") else None
            n = d2g.ut_count(N, a, z)
            got = np.empty(n, np.float32)
            ctxs[r].d2h(got, outs[r] + 4 * off(r0, a))
            if want is None:
                want = lut[ref.eqcount_ut(a, z)]
            np.testing.assert_array_equal(got.view(np.uint32), np.asarray(want, np.float32).view(np.uint32), err_msg=f"rank {r} rows [{a},{z})")
    # integer counts of one rank through the step form
    d2g.allpairs_step_all(engs, rows, None, outs)
    r = 5
    ctxs[r].sync()
    r0, r1 = engs[r].rows_computed
    got = np.empty(d2g.ut_count(N, r0, r0 + 2), np.uint32)
    ctxs[r].d2h(got, outs[r])
    np.testing.assert_array_equal(got, ref.eqcount_ut(r0, r0 + 2))
    ref.close()
    for r in range(W):
        for p in (rows[r], outs[r], luts[r]):
            ctxs[r].free(p)
    for x in engs + comms + ctxs:
        x.close()


@pytest.mark.parametrize("W,N,S,kind", [(2, 2600, 512, "families"), (3, 1900, 256, "families"), (4, 1500, 1024, "families"), (2, 900, 128, "planted"), (3, 1100, 128, "one_family")])
def test_allpairs_sparse_tiles_on_the_gathered_operand(d2g, oracle, monkeypatch, W, N, S, kind):
    """The engine's pair phase over the gathered operand takes the sparse-tile path (production: N >= 8192; forced here): every rank
    re-derives ids from the exchanged planes, finds the families, lists the pairs across families and runs the families' tiles (or
    the dense walk behind the gate) -- every rank's whole slab against the oracle, both epilogues, twice in a row (the second step
    re-orders a re-gathered operand), and the path reported by d2g_allpairs_sparse_info."""
    monkeypatch.setenv("D2G_BS_SPARSE_MIN_N", "1")
    monkeypatch.setenv("D2G_SP_TILE_FRAC", "1")                       # (matrices this small have few tiles: the families' share of them is large)
    rng = np.random.default_rng(N + W)
    if kind == "families":                                            # families of ~40 sketches sharing most registers, strangers otherwise
        fam = rng.integers(0, max(2, N // 40), N)
        base = rng.random((fam.max() + 1, S))
        sigs = np.where(rng.random((N, S)) < 0.7, base[fam], rng.random((N, S)))
    elif kind == "one_family":
        base = rng.random(S)
        sigs = np.where(rng.random((N, S)) < 0.5, base[None, :], rng.random((N, S)))
    else:
        sigs = _planted(rng, N, S)
    bits = np.ascontiguousarray(sigs).view(np.uint64)
    exp = oracle.eqcounts_ut(sigs)
    lut = d2g.epilogue_lut(S, d2g.SIMILARITY, 31)
    ctxs = [d2g.Context(0) for _ in range(W)]
    comms = d2g.Comm.create_all(ctxs)
    engs = [d2g.AllPairs(ctxs[r], comms[r], N, S) for r in range(W)]
    rows = [_upload(ctxs[r], bits[engs[r].rows_held[0]:engs[r].rows_held[1]]) for r in range(W)]
    outs = [ctxs[r].malloc(max(d2g.ut_count(N, *engs[r].rows_computed), 1) * 4) for r in range(W)]
    luts = [_upload(ctxs[r], lut) for r in range(W)]
    for e in engs:
        e.set_phase_timing(True)
    for rnd, use_lut in enumerate((False, True, False)):
        d2g.allpairs_step_all(engs, rows, luts if use_lut else None, outs)
        for r in range(W):
            ctxs[r].sync()
            engs[r].status()
            r0, r1 = engs[r].rows_computed
            n = d2g.ut_count(N, r0, r1)
            if not n:
                continue
            a = d2g.ut_count(N, 0, r0)
            if use_lut:
                got = np.empty(n, np.float32)
                ctxs[r].d2h(got, outs[r])
                np.testing.assert_array_equal(got.view(np.uint32), lut[exp[a:a + n]].view(np.uint32), err_msg=f"round {rnd} rank {r}")
            else:
                got = np.empty(n, np.uint32)
                ctxs[r].d2h(got, outs[r])
                np.testing.assert_array_equal(got, exp[a:a + n], err_msg=f"round {rnd} rank {r}")
            info = engs[r].sparse_info()
            assert info["sorted_operand"]
            if kind == "families":
                assert info["tiles_listed"] > 0 and info["tiles_and_pair_list"] and not info["dense_decided_by_prepare"]
            if kind == "one_family":
                assert info["dense_kernel_ran"]
    assert "order" in {p["phase"] for p in engs[0].phase_times()}
    for r in range(W):
        for p in (rows[r], outs[r], luts[r]):
            ctxs[r].free(p)
    for x in engs + comms + ctxs:
        x.close()


def test_allpairs_refuses_ranks_with_different_switches(d2g, monkeypatch):
    """a context reads the library's D2G_* switches once, when it is created; the engines of one job must have resolved
    the same D2G_BS_* / D2G_SP_* set (they select the kernels a rank runs).  Two loopback ranks whose contexts were created under
    different environments: the step is refused with a message that names the switches; d2g_ctx_tuning shows what each resolved;
    after a reload under one environment the same engines run."""
    N, S, W = 300, 64, 2
    rng = np.random.default_rng(3)
    bits = np.ascontiguousarray(_planted(rng, N, S)).view(np.uint64)
    ctx0 = d2g.Context(0)
    monkeypatch.setenv("D2G_SP_LINK", "0")
    ctx1 = d2g.Context(0)                                             # created under another switch
    monkeypatch.delenv("D2G_SP_LINK")
    assert ctx0.tuning().get("D2G_SP_LINK") is None and ctx1.tuning().get("D2G_SP_LINK") == "0"
    ctxs = [ctx0, ctx1]
    comms = d2g.Comm.create_all(ctxs)
    engs = [d2g.AllPairs(ctxs[r], comms[r], N, S) for r in range(W)]
    rows = [_upload(ctxs[r], bits[engs[r].rows_held[0]:engs[r].rows_held[1]]) for r in range(W)]
    outs = [ctxs[r].malloc(max(d2g.ut_count(N, *engs[r].rows_computed), 1) * 4) for r in range(W)]
    with pytest.raises(d2g.D2GError, match="D2G_BS_"):
        d2g.allpairs_step_all(engs, rows, None, outs)
    ctx1.reload_tuning()
    assert ctx1.tuning().get("D2G_SP_LINK") is None
    d2g.allpairs_step_all(engs, rows, None, outs)
    for r in range(W):
        ctxs[r].sync()
        engs[r].status()
        for p in (rows[r], outs[r]):
            ctxs[r].free(p)
    for x in engs + comms + ctxs:
        x.close()


def _run_bench(*argv, env=None, timeout=900):
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True, env=e, timeout=timeout)
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and lines, (r.returncode, r.stdout[-300:], r.stderr[-1500:])
    return json.loads(lines[-1])


@pytest.mark.parametrize("W", [2, 8])
def test_bench_inprocess_rung_over_loopback(d2g, W):
    """bench.py's second rung -- ONE process driving W ranks through d2g_comm_create_all + d2g_allpairs_step_all, here
    over the loopback transport on one GPU -- produces a complete N > 1 line: every rank's WHOLE slab equal to a single-GPU
    computation, the same-config 1-GPU base, the per-phase times of one step on every rank (each phase of each chunk present), and
    the supervisor's record of the ladder."""
    N, S = 3000 + 8 * W, 1024
    line = _run_bench("--gpus", str(W), "--loopback", "--sketches", str(N), "--steps", "3", "--warmup", "1")
    assert line["n_gpus"] == W and line["value"] > 0 and "valid" not in line
    cfg = line["config"]
    assert cfg["sketches"] == (N + W - 1) // W * W and "WHOLE slab equals" in cfg["slab_check"] and "ONE process" in cfg["exchange_engine"]
    assert [l["engine"] for l in line["launcher"]["ladder"]] == ["inproc"] and line["launcher"]["ladder"][0]["outcome"] == "ok"
    base = line["scaling_base"]
    assert base["best_1gpu_pairs_per_s"] > 0 and base["base_1gpu_same_config_pairs_per_s"] > 0 and abs(base["speedup"] - line["value"] / base["base_1gpu_same_config_pairs_per_s"]) < 1e-9
    ph = line["phases"]
    C = cfg["exchange_chunks"]
    assert len(ph["per_rank"]) == W
    for rec in ph["per_rank"]:
        kinds = [(p[0], p[1]) for p in rec if p[0] not in ("order", "fill")]   # "order" / "fill": only when the sparse path is on at this N
        want = [("pack", 0)] + [(k, c) for k in ("x1", "prepare", "x2", "derive") for c in range(C)] + [("pair", 0)]
        assert sorted(kinds) == sorted(want), kinds
        assert all(p[3] >= 0 and p[2] >= 0 for p in rec)
    assert set(ph["max_over_ranks_ms"]) - {"order", "fill"} == {"pack", "x1", "prepare", "x2", "derive", "pair"}
    assert len(line["per_rank"]) == W and sum(p["pairs"] for p in line["per_rank"]) == cfg["pairs"]


def test_allpairs_phase_times_cover_the_step(d2g, oracle):
    """d2g_allpairs_set_phase_timing / d2g_allpairs_phase_times: one record per phase and chunk, non-negative, the pair kernel last;
    switching it off again leaves the step's results untouched"""
    rng = np.random.default_rng(77)
    W, N, S = 3, 700, 256
    sigs = _planted(rng, N, S)
    bits = sigs.view(np.uint64)
    exp = oracle.eqcounts_ut(sigs)
    off = np.concatenate([[0], np.cumsum(N - 1 - np.arange(N, dtype=np.int64))])
    ctxs = [d2g.Context(0) for _ in range(W)]
    comms = d2g.Comm.create_all(ctxs)
    engs = [d2g.AllPairs(ctxs[r], comms[r], N, S) for r in range(W)]
    rows = [_upload(ctxs[r], bits[engs[r].rows_held[0]:engs[r].rows_held[1]]) for r in range(W)]
    outs = [ctxs[r].malloc(max(d2g.ut_count(N, *engs[r].rows_computed), 1) * 4) for r in range(W)]
    for e in engs:
        e.set_phase_timing(True)
    d2g.allpairs_step_all(engs, rows, None, outs)
    C = engs[0].chunks
    for r, e in enumerate(engs):
        recs = e.phase_times()
        assert [p["phase"] for p in recs][-1] == "pair" and recs[0]["phase"] == "pack"
        assert sorted((p["phase"], p["chunk"]) for p in recs if p["phase"] not in ("order", "fill")) == sorted([("pack", 0), ("pair", 0)] + [(k, c) for k in ("x1", "prepare", "x2", "derive") for c in range(C)])
        assert all(p["ms"] >= 0 and p["start_ms"] >= 0 for p in recs)
        e.set_phase_timing(False)
        assert e.phase_times() == []
    d2g.allpairs_step_all(engs, rows, None, outs)
    for r in range(W):
        r0, r1 = engs[r].rows_computed
        got = np.empty(d2g.ut_count(N, r0, r1), np.uint32)
        ctxs[r].sync()
        ctxs[r].d2h(got, outs[r])
        np.testing.assert_array_equal(got, exp[off[r0]:off[r1]])
    for e in engs:
        e.close()
    for c in comms:
        c.close()
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("engine", ["cabi", "torch", "broadcast"])
def test_bench_ranked_rungs_at_world_size_one(d2g, engine, tmp_path):
    """The ranked rungs of bench.py's N > 1 ladder cannot form a clique of two on a 1-GPU box (RCCL refuses one device twice), but their
    worker code can run at world size 1: process group (gloo for `cabi`, nccl for the others), libd2g's communicator from a unique id
    (a real one-member RCCL communicator), the engine, the whole-slab check against a single-GPU computation, both bases, the phase
    times, the timed region and the JSON line -- everything but a second rank."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    from dashing2_amd import synth
    N, S = 2600, 1024
    regs = synth.synthetic_registers(N, S, nclusters=17, seed=5)
    sig, cards = d2g.oph_finalize(regs, S, nthreads=4)
    np.save(tmp_path / "sig.npy", sig.view(np.uint64))
    np.save(tmp_path / "cards.npy", cards)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--worker", "--engine", engine, "--gpus", "1", "--sketches", str(N), "--steps", "3", "--warmup", "1",
                        "--sig-file", str(tmp_path / "sig.npy"), "--cards-file", str(tmp_path / "cards.npy")], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2500:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    line = json.loads(lines[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and "valid" not in line and "WHOLE slab equals" in line["config"]["slab_check"]
    assert line["scaling_base"]["base_1gpu_same_config_pairs_per_s"] > 0 and line["per_rank"][0]["pairs"] == N * (N - 1) // 2
    if engine == "cabi":
        assert {p[0] for p in line["phases"]["per_rank"][0]} - {"order", "fill"} == {"pack", "x1", "prepare", "x2", "derive", "pair"}
        assert line["stream_of_matrices"]["outputs_identical_to_the_one_job_step"] is True

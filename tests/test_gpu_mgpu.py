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

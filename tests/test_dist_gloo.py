"""CPU tests of the N>1 path (world_size 2 and 3, gloo): pair-balanced row sharding, the one
broadcast, disjoint slab writes and the rank-0 gather.  The device call is replaced by the checker
(oracle) through dist.sharded_allpairs(compute=...): these tests cover the HOST logic only;
the GPU slab computation itself is covered by tests/test_gpu_k2.py (row-range parity)."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp, N, S, measure):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch.distributed as dist
    from dashing2_amd import dist as D
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(123)
    vals = rng.random((4, S))
    sigs = vals[rng.integers(0, 4, (N, S)), np.arange(S)[None, :]]
    cards = rng.random(N) * 1e5 + 1

    def compute(t, c, n, s, r0, r1):
        m = t.numpy().view(np.float64).reshape(n, s)
        return O.allpairs_ut(m, c, measure=measure, k=31, nthreads=1, rows=(r0, r1))

    r0, r1, slab = D.sharded_allpairs(sigs.view(np.uint64) if rank == 0 else None, cards if rank == 0 else None, N, S, compute)
    b = D.row_bounds(N, world)
    assert (r0, r1) == (b[rank], b[rank + 1])
    out = os.path.join(tmp, "dist.bin")
    D.write_slab(out, N, r0, slab)
    full = D.gather_slabs(slab)
    if rank == 0:
        exp = O.allpairs_ut(sigs, cards, measure=measure, k=31, nthreads=2)
        np.testing.assert_array_equal(full.view(np.uint32), exp.view(np.uint32))
        np.testing.assert_array_equal(np.fromfile(out, np.float32).view(np.uint32), exp.view(np.uint32))
        np.save(os.path.join(tmp, "ok.npy"), np.array([1]))
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,N", [(2, 37), (3, 50), (2, 2)])
def test_sharded_allpairs_gloo(tmp_path, world, N):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), N, 64, 3), nprocs=world, join=True)
    assert os.path.exists(tmp_path / "ok.npy")


def test_partition_balance():
    from dashing2_amd import dist as D
    for N, w in [(10000, 8), (28284, 8), (50000, 8), (1000, 3)]:
        b = D.row_bounds(N, w)
        cnt = [D.slab_offset(N, b[i + 1]) - D.slab_offset(N, b[i]) for i in range(w)]
        assert sum(cnt) == N * (N - 1) // 2
        assert max(cnt) - min(cnt) <= 2 * N


def _worker_v2(rank, world, port, tmp, n_loc, S_loc):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch
    import torch.distributed as dist
    from dashing2_amd import dist as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, S = n_loc * world, S_loc * world
    full = torch.arange(N * S, dtype=torch.int64).reshape(N, S) * 7 + 3          # the whole matrix (every rank can rebuild it)
    mine = full[rank * n_loc:(rank + 1) * n_loc]                                 # rows this rank owns
    # the send layout libd2g's d2g_pack_column_slices_dev produces: [W][n_loc][S_loc]
    send = mine.reshape(n_loc, world, S_loc).permute(1, 0, 2).contiguous()
    recv = torch.empty((N, S_loc), dtype=torch.int64)
    D.exchange_rows_to_colslices(send, recv)
    assert torch.equal(recv, full[:, rank * S_loc:(rank + 1) * S_loc])           # all rows of my column slice, global row order
    # stand-in for the per-rank operand piece: any function of the column slice
    piece = (recv.sum(dim=0) + rank).to(torch.int32)                            # [S_loc]
    allp = torch.empty(world * S_loc, dtype=torch.int32)
    D.gather_groups(piece, allp)
    exp = torch.cat([(full[:, q * S_loc:(q + 1) * S_loc].sum(dim=0) + q).to(torch.int32) for q in range(world)])
    assert torch.equal(allp, exp)                                               # rank-major == register-group order
    if rank == 0:
        np.save(os.path.join(tmp, "ok2.npy"), np.array([1]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_row_sharded_exchange_gloo(tmp_path, world):
    import torch.multiprocessing as mp
    mp.spawn(_worker_v2, args=(world, _free_port(), str(tmp_path), 5, 64), nprocs=world, join=True)
    assert os.path.exists(tmp_path / "ok2.npy")


def _sketch_worker(rank, world, port, tmp, npaths):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch.distributed as dist
    from dashing2_amd import dist as D
    from dashing2_amd import synth
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    k, S = 15, 64
    paths = [os.path.join(tmp, f"g{i}.fa") for i in range(npaths)]
    if rank == 0:
        for i, p in enumerate(paths):
            synth.write_fasta(p, f"g{i}", synth.random_genome(100 + i, 2000 + 300 * i))
    dist.barrier()

    def fake_cli(args, device):              # stands in for the GPU CLI: the oracle writes the shard's stacked file
        assert args[0] == "sketch" and "-F" in args and "-o" in args
        lst, out = args[args.index("-F") + 1], args[args.index("-o") + 1]
        mine = [l.strip() for l in open(lst) if l.strip()]
        sigs, cards = O.sketch_files(mine, k=k, canon=True, xormask=0, S=S, nthreads=1)
        with open(out, "wb") as f:
            np.array([len(mine), S], np.uint64).tofile(f)
            cards.tofile(f)
            sigs.tofile(f)

    out = os.path.join(tmp, "stack.bin")
    res = D.sketch_sharded(paths, out, ["-k", str(k), "-S", str(S)], run=fake_cli, device=rank)
    if rank == 0:
        N, S2, cards, sigs = res
        esigs, ecards = O.sketch_files(paths, k=k, canon=True, xormask=0, S=S, nthreads=2)
        assert N == npaths and S2 == S
        np.testing.assert_array_equal(sigs.view(np.uint64), esigs.view(np.uint64))
        np.testing.assert_array_equal(cards, ecards)
        n2, s2, c2, g2 = D.load_stacked(out)
        assert (n2, s2) == (npaths, S) and np.array_equal(g2.view(np.uint64), esigs.view(np.uint64)) and np.array_equal(c2, ecards)
        lines = open(out + ".names.txt").read().splitlines()
        assert lines[0] == "#Name\tCardinality" and [l.split("\t")[0] for l in lines[1:]] == paths
        assert all(l.split("\t")[1] == "%0.24g" % c for l, c in zip(lines[1:], ecards))
        assert not any(f.startswith("stack.bin.shard") for f in os.listdir(tmp))
        np.save(os.path.join(tmp, "ok.npy"), np.array([1]))
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,npaths", [(2, 7), (3, 2)])
def test_sketch_sharded_gloo(tmp_path, world, npaths):
    """SKETCH across ranks: files r, r+world, ... per rank, one barrier, rank 0 interleaves the shards
    back into input order (world 3 with 2 inputs leaves one rank without work)."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_sketch_worker, args=(world, port, str(tmp_path), npaths), nprocs=world, join=True)
    assert os.path.exists(tmp_path / "ok.npy")


def _sketch_fail_worker(rank, world, port, tmp):
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch.distributed as dist
    from dashing2_amd import dist as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    paths = [os.path.join(tmp, f"g{i}.fa") for i in range(4)]

    def failing_cli(args, device):
        if device == 1:
            raise RuntimeError("simulated CLI failure")
        out = args[args.index("-o") + 1]
        mine = [l for l in open(args[args.index("-F") + 1]) if l.strip()]
        with open(out, "wb") as f:
            np.array([len(mine), 4], np.uint64).tofile(f)
            np.zeros(len(mine) * 5).tofile(f)

    t0 = time.time()
    try:
        D.sketch_sharded(paths, os.path.join(tmp, "stack.bin"), [], run=failing_cli, device=rank)
        raised = False
    except RuntimeError as e:
        raised = "simulated CLI failure" in str(e)
    assert raised and time.time() - t0 < 60                   # every rank learns of the failure, none waits for a timeout
    np.save(os.path.join(tmp, f"fail{rank}.npy"), np.array([1]))
    dist.destroy_process_group()


def test_sketch_sharded_failure_reaches_every_rank(tmp_path):
    """a rank whose CLI call raises used to skip the barrier and leave the others hanging"""
    import torch.multiprocessing as mp
    mp.spawn(_sketch_fail_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "fail0.npy") and os.path.exists(tmp_path / "fail1.npy")

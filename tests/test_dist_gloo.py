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

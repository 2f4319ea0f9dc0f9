"""Multi-GPU all-pairs comparison: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU tests).

The path shards: the upper triangle's rows are split into `world` contiguous ranges of
equal pair count (d2g_ut_partition); rank 0's signature matrix is broadcast ONCE (the path's only
exchange), every rank prepares the operand on its GPU and computes its slab; slabs are disjoint,
so there is no reduction.  Each rank writes its slab at its own offset of the binary matrix
(F-d layout, src/emitrect.cpp:373-397) or the slabs are gathered to rank 0 for text output.

Launch:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
             -m dashing2_amd.dist --presketched stack.bin --cmpout dist.bin [--distance] [-k 31]

PyTorch is plumbing only (process group, device tensors); the computation is libd2g's.
`compute=` lets the CPU tests substitute the checker for the device call; the product default is
the GPU path and raises without one.
"""
import argparse
import os
import sys

import numpy as np


def _dist():
    import torch.distributed as dist
    return dist


def rank_world():
    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def row_bounds(N, world):
    """pair-balanced contiguous row ranges (host arithmetic in libd2g)."""
    from . import capi
    return capi.ut_partition(N, world)


def slab_offset(N, r0):
    """index of pair (r0, r0+1) in the condensed upper triangle"""
    from . import capi
    return capi.ut_count(N, 0, r0)


def gpu_compute(ctx, measure, k, multiset_space=False, algo=0):
    """default slab computation: libd2g on this rank's GPU, device-resident operand."""
    from . import capi
    import torch

    def run(sig_t, cards, N, S, r0, r1):
        cnt = capi.ut_count(N, r0, r1)
        stream = torch.cuda.current_stream().cuda_stream
        need_gtlt = (not multiset_space) and (S & (S - 1)) != 0
        cs = ctx.cmp_set_dev(sig_t.data_ptr(), N, S, algo=capi.CMP_DIRECT if need_gtlt else algo, stream=stream)
        try:
            try:
                lut = capi.epilogue_lut(S, measure, k, multiset_space)
            except capi.D2GError:
                lut = None
            if lut is not None:
                lut_t = torch.from_numpy(lut).to(sig_t.device)
                out = torch.empty(max(cnt, 1), dtype=torch.float32, device=sig_t.device)
                cs.lut_ut_dev(lut_t.data_ptr(), out.data_ptr(), r0, r1, stream)
                torch.cuda.synchronize()
                return out[:cnt].cpu().numpy()
            # card-dependent measure / non power-of-two S: integer counts + x87 host epilogue (libd2g)
            a = torch.empty(max(cnt, 1), dtype=torch.int32, device=sig_t.device)
            cb = None
            if need_gtlt:
                b = torch.empty(max(cnt, 1), dtype=torch.int32, device=sig_t.device)
                cs.gtlt_ut_dev(a.data_ptr(), b.data_ptr(), r0, r1, stream)
                torch.cuda.synchronize()
                cb = b[:cnt].cpu().numpy().view(np.uint32)
            else:
                cs.eqcount_ut_dev(a.data_ptr(), r0, r1, stream)
                torch.cuda.synchronize()
            ca = a[:cnt].cpu().numpy().view(np.uint32)
            return capi.host_epilogue_ut(ca, cb, cards, N, S, r0, r1, measure, k, multiset_space)
        finally:
            cs.close()
    return run


def sharded_allpairs(sig_bits, cards, N, S, compute, device="cpu"):
    """sig_bits: uint64 [N][S] numpy on rank 0 (None elsewhere); cards likewise.
    Returns (r0, r1, slab float32) for this rank."""
    import torch
    dist = _dist()
    rank, world = rank_world()
    t = torch.empty((N, S), dtype=torch.int64, device=device)
    c = torch.empty(N, dtype=torch.float64, device=device)
    if rank == 0:
        t.copy_(torch.from_numpy(np.ascontiguousarray(sig_bits).view(np.int64).reshape(N, S)))
        c.copy_(torch.from_numpy(np.ascontiguousarray(cards, np.float64)))
    if world > 1:
        dist.broadcast(t, 0)           # the one exchange of the path (RCCL over xGMI on GPUs)
        dist.broadcast(c, 0)
    b = row_bounds(N, world)
    r0, r1 = b[rank], b[rank + 1]
    slab = compute(t, c.cpu().numpy(), N, S, r0, r1)
    return r0, r1, np.ascontiguousarray(slab, np.float32)


def write_slab(path, N, r0, slab):
    """every rank writes its disjoint byte range of the condensed float32 matrix (F-d)."""
    rank, world = rank_world()
    total = N * (N - 1) // 2
    if rank == 0:
        with open(path, "wb") as f:
            f.truncate(total * 4)
    if world > 1:
        _dist().barrier()
    fd = os.open(path, os.O_WRONLY)
    try:
        os.pwrite(fd, slab.tobytes(), slab_offset(N, r0) * 4)
    finally:
        os.close(fd)
    if world > 1:
        _dist().barrier()


def gather_slabs(slab):
    """concatenate the ranks' slabs on rank 0 (row order == rank order); None elsewhere."""
    rank, world = rank_world()
    if world == 1:
        return slab
    import torch
    dist = _dist()
    sizes = [None] * world
    dist.all_gather_object(sizes, int(slab.size))
    if rank == 0:
        parts = [slab]
        for r in range(1, world):
            buf = torch.empty(max(sizes[r], 1), dtype=torch.float32)
            dist.recv(buf, src=r)
            parts.append(buf[:sizes[r]].numpy())
        return np.concatenate(parts)
    t = torch.from_numpy(slab if slab.size else np.zeros(1, np.float32))
    dist.send(t, dst=0)
    return None


def load_stacked(path):
    """[u64 N][u64 S][f64 card x N][f64 x N*S]  (src/sketch_core.cpp:130-140, cmp_main.cpp:61-94)"""
    raw = np.fromfile(path, np.uint8)
    N, S = (int(x) for x in raw[:16].view(np.uint64))
    cards = raw[16:16 + 8 * N].view(np.float64).copy()
    sigs = raw[16 + 8 * N:].view(np.float64).reshape(N, S).copy()
    return N, S, cards, sigs


def main(argv=None):
    import torch
    from . import capi
    ap = argparse.ArgumentParser(prog="dashing2_amd.dist")
    ap.add_argument("--presketched", required=True, help="stacked sketch file written by `dashing2 sketch -o`")
    ap.add_argument("--cmpout", required=True, help="binary float32 condensed matrix (reference --binary-output layout)")
    ap.add_argument("-k", "--kmer-length", type=int, default=32)
    for flag, m in (("--distance", capi.POISSON_LLR), ("--mash-distance", capi.POISSON_LLR), ("--containment", capi.CONTAINMENT),
                    ("--symmetric-containment", capi.SYMMETRIC_CONTAINMENT), ("--intersection", capi.INTERSECTION),
                    ("--union-size", capi.UNION_SIZE)):
        ap.add_argument(flag, dest="measure", action="store_const", const=m)
    ap.set_defaults(measure=capi.SIMILARITY)
    args = ap.parse_args(argv)
    dist = _dist()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    ctx = capi.Context(local)
    hdr = torch.zeros(2, dtype=torch.int64, device="cuda")
    sigs = cards = None
    if rank == 0:
        N, S, cards, sigs = load_stacked(args.presketched)
        sigs, _ = capi.densify(sigs, nthreads=os.cpu_count() or 1)     # cmp_core.cpp:686-718
        hdr[0], hdr[1] = N, S
    if world > 1:
        dist.broadcast(hdr, 0)
    N, S = int(hdr[0]), int(hdr[1])
    r0, r1, slab = sharded_allpairs(sigs.view(np.uint64) if rank == 0 else None, cards, N, S,
                                    gpu_compute(ctx, args.measure, args.kmer_length), device="cuda")
    write_slab(args.cmpout, N, r0, slab)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())

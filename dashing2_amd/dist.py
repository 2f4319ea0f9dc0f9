"""Multi-GPU all-pairs comparison: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU tests).

The path shards: the upper triangle's rows are split into `world` contiguous ranges of
equal pair count (d2g_ut_partition); rank 0's signature matrix is broadcast ONCE (the path's only
exchange), every rank prepares the operand on its GPU and computes its slab; slabs are disjoint,
so there is no reduction.  Each rank writes its slab at its own offset of the binary matrix
(F-d layout, src/emitrect.cpp:373-397) or the slabs are gathered to rank 0 for text output.

Launch:  python -m dashing2_amd.dist cmp --presketched stack.bin --cmpout dist.bin [--distance] [-k 31]
             (one process: the C++ CLI drives every visible GPU through libd2g's RCCL communicator)
         python -m dashing2_amd.dist sketch -F files.txt -o stack.bin [...]
             (one process: `D2G_DEVICES=all dashing2 sketch`, a pair of device threads per GPU -- the product path since round 4)
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
             -m dashing2_amd.dist sketch -F files.txt -o stack.bin [-k 31 -S 1024 --multiset ...]
The torch.distributed classes below (sharded_allpairs, RowShardedAllPairs) are the same exchange written against
torch collectives: bench.py's fallback when the C-ABI engine cannot be used, and what the gloo CPU tests exercise.
SKETCH shards by input (rank r takes files r, r+world, ...; each rank runs the CLI on its own GPU;
no collective, one barrier) and rank 0 interleaves the shards back into input order.

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



# --------------------------------------------------------------------------------------------
# Scalable exchange for row-sharded sketches (what sharded sketching leaves in HBM):
#   all-to-all (row slices -> column slices), per-rank prepare of S/W columns, all-gather of the
#   compact bit-sliced operand (<= (ceil(log2 N)+1)/64 of the raw 8-byte registers).
# --------------------------------------------------------------------------------------------
def exchange_rows_to_colslices(send, recv):
    """send: [W][n_loc][S_loc] (block q = my rows, column slice q); recv: [W*n_loc][S_loc] = all rows of
    my column slice, in global row order (ranks own consecutive row blocks)."""
    rank, world = rank_world()
    if world == 1:
        recv.copy_(send.view(recv.shape))
    else:
        _dist().all_to_all_single(recv.view(-1), send.view(-1))
    return recv


def gather_groups(local, full):
    """all-gather of per-rank operand pieces, rank-major (= register-group order)."""
    rank, world = rank_world()
    if world == 1:
        full.copy_(local.view(full.shape))
    else:
        _dist().all_gather_into_tensor(full.view(-1), local.view(-1))
    return full


class RowShardedAllPairs:
    """All-pairs over an N x S sketch matrix whose rows [rank*n_loc, (rank+1)*n_loc) live on this
    rank's GPU.  step() runs one whole pass: exchange, sharded prepare, this rank's slab of pairs."""

    def __init__(self, ctx, N, S, device):
        import torch
        from . import capi
        self.ctx = ctx
        self.rank, self.world = rank_world()
        W = self.world
        if N % W or S % (32 * W):
            raise ValueError(f"row-sharded all-pairs needs N % world == 0 and S % (32*world) == 0 (N={N}, S={S}, world={W})")
        self.N, self.S, self.n_loc, self.S_loc = N, S, N // W, S // W
        gw, ng = capi.operand_layout(N, S)
        gw_l, ng_l = capi.operand_layout(N, self.S_loc)
        assert gw == gw_l and ng == ng_l * W
        i64, i32 = torch.int64, torch.int32
        self.send = torch.empty((W, self.n_loc, self.S_loc), dtype=i64, device=device)
        self.recv = torch.empty((N, self.S_loc), dtype=i64, device=device)
        self.planes_loc = torch.empty(ng_l * gw, dtype=i32, device=device)
        self.meta_loc = torch.empty(ng_l, dtype=i32, device=device)
        self.planes_all = torch.empty(ng * gw, dtype=i32, device=device)
        self.meta_all = torch.empty(ng, dtype=i32, device=device)
        self.local = None
        self.full = ctx.cmp_set_from_planes(N, S, self.planes_all.data_ptr(), self.meta_all.data_ptr())
        b = row_bounds(N, W)
        self.r0, self.r1 = b[self.rank], b[self.rank + 1]

    def prepare(self, rows_t, stream):
        from . import capi
        self.ctx.pack_column_slices_dev(rows_t.data_ptr(), self.n_loc, self.S, self.world, self.send.data_ptr(), stream)
        exchange_rows_to_colslices(self.send, self.recv)
        if self.local is None:
            self.local = self.ctx.cmp_set_dev(self.recv.data_ptr(), self.N, self.S_loc, algo=capi.CMP_BITSLICE, stream=stream)
        else:
            self.local.update_dev(self.recv.data_ptr(), stream)
        self.local.export_operand_dev(self.planes_loc.data_ptr(), self.meta_loc.data_ptr(), stream)
        gather_groups(self.planes_loc, self.planes_all)
        gather_groups(self.meta_loc, self.meta_all)

    def step_lut(self, rows_t, lut_t, out_t, stream):
        self.prepare(rows_t, stream)
        self.full.lut_ut_dev(lut_t.data_ptr(), out_t.data_ptr(), self.r0, self.r1, stream)

    def step_eqcount(self, rows_t, out_t, stream):
        self.prepare(rows_t, stream)
        self.full.eqcount_ut_dev(out_t.data_ptr(), self.r0, self.r1, stream)

    # ---- software-pipelined form: exchange + prepare of batch i+1 overlap the pair kernel of batch i
    def _pipe_init(self):
        import torch
        dev = self.planes_all.device
        self._xs = torch.cuda.Stream(device=dev)                   # exchange / prepare stream
        self._pl = [self.planes_all, torch.empty_like(self.planes_all)]
        self._mt = [self.meta_all, torch.empty_like(self.meta_all)]
        self._full = [self.full, self.ctx.cmp_set_from_planes(self.N, self.S, self._pl[1].data_ptr(), self._mt[1].data_ptr())]
        self._x_done = [torch.cuda.Event(), torch.cuda.Event()]
        self._p_done = [None, None]
        self._n = 0

    def enqueue_lut(self, rows_t, lut_t, out_t, ready=None):
        """One step, pipelined over two operand buffers: the all-to-all, the sharded prepare and the
        all-gather of step i run on their own stream while the pair kernel of step i-1 is still busy on
        the caller's current stream.  Results land in out_t in step order; torch.cuda.synchronize()
        (or any later work on the current stream) observes them.

        ready: when rows_t becomes valid.  None (default) = after everything queued so far on the current
        stream (safe for inputs produced there right before the call); a torch.cuda.Event = after that
        event; False = the input is already complete (e.g. synchronised earlier), no dependency."""
        import torch
        from . import capi
        if not hasattr(self, "_xs"):
            self._pipe_init()
        main = torch.cuda.current_stream()
        i = self._n & 1
        xs = self._xs
        # rows_t may have been produced on the caller's stream just before this call (in a
        # sketch-then-compare loop every step's input is), so by default the exchange stream waits for
        # everything the main stream has queued so far.  That includes the previous step's pair kernel:
        # the default is safe but serialises exchange(i) behind pair(i-1).  A caller that knows when its
        # input is complete passes that event (or False) and keeps the overlap.
        if ready is None:
            ev_in = torch.cuda.Event()
            ev_in.record(main)
            xs.wait_event(ev_in)
        elif ready is not False:
            xs.wait_event(ready)
        if self._p_done[i] is not None:
            xs.wait_event(self._p_done[i])                          # buffer i is free once pair(i-2) is done
        with torch.cuda.stream(xs):
            st = xs.cuda_stream
            self.ctx.pack_column_slices_dev(rows_t.data_ptr(), self.n_loc, self.S, self.world, self.send.data_ptr(), st)
            exchange_rows_to_colslices(self.send, self.recv)
            if self.local is None:
                self.local = self.ctx.cmp_set_dev(self.recv.data_ptr(), self.N, self.S_loc, algo=capi.CMP_BITSLICE, stream=st)
            else:
                self.local.update_dev(self.recv.data_ptr(), st)
            self.local.export_operand_dev(self.planes_loc.data_ptr(), self.meta_loc.data_ptr(), st)
            gather_groups(self.planes_loc, self._pl[i])
            gather_groups(self.meta_loc, self._mt[i])
            self._x_done[i].record(xs)
        main.wait_event(self._x_done[i])
        self._full[i].lut_ut_dev(lut_t.data_ptr(), out_t.data_ptr(), self.r0, self.r1, main.cuda_stream)
        ev = torch.cuda.Event()
        ev.record(main)
        self._p_done[i] = ev
        self._n += 1

    def close(self):
        if self.local is not None:
            self.local.close()
        self.full.close()
        if hasattr(self, "_full"):
            self._full[1].close()


def _default_cli_run(args, device):
    """run the drop-in CLI on one GPU (D2G_DEVICE selects it); raises on failure"""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "dashing2")
    env = dict(os.environ, D2G_DEVICE=str(device))
    r = subprocess.run([exe] + list(args), env=env, capture_output=True)
    if r.returncode:
        raise RuntimeError("dashing2 %s failed on device %s: %s" % (" ".join(args[:1]), device, r.stderr.decode()[-2000:]))


def sketch_sharded(paths, out, cli_args=(), run=None, device=None):
    """SKETCH across ranks (inputs sharded one-per-GPU, no collectives): rank r sketches
    paths[r::world] with the CLI on its GPU into `out`.shard<r>; after a barrier rank 0 interleaves the
    shards back into input order and writes the stacked file + names (F-b / F-c layouts,
    src/sketch_core.cpp:130-161).  `run(args, device)` is the per-rank CLI call (tests substitute it)."""
    rank, world = rank_world()
    run = run or _default_cli_run
    device = int(os.environ.get("LOCAL_RANK", "0")) if device is None else device
    mine = list(range(rank, len(paths), world))
    shard = f"{out}.shard{rank}"
    lst = shard + ".files.txt"
    with open(lst, "w") as f:
        f.write("".join(paths[i] + "\n" for i in mine))
    err = None
    if mine:
        try:
            run(["sketch", *cli_args, "-F", lst, "-o", shard], device)
        except Exception as e:                                   # noqa: BLE001 - re-raised on every rank below
            err = f"rank {rank}: {type(e).__name__}: {e}"
    if world > 1:
        # one collective carries the failure to every rank: a rank that raised before the barrier used to leave the
        # others waiting for the process-group timeout
        errs = [None] * world
        _dist().all_gather_object(errs, err)
        err = next((e for e in errs if e), None)
    if err:
        raise RuntimeError("sharded sketching failed: " + err)
    if rank != 0:
        return None
    N = len(paths)
    S = cards = sigs = None
    for r in range(world):
        idx = list(range(r, N, world))
        if not idx:
            continue
        n_r, S_r, c_r, s_r = load_stacked(f"{out}.shard{r}")
        if n_r != len(idx):
            raise RuntimeError(f"shard {r} holds {n_r} sketches, expected {len(idx)}")
        if S is None:
            S, cards, sigs = S_r, np.empty(N, np.float64), np.empty((N, S_r), np.float64)
        cards[idx] = c_r
        sigs[idx] = s_r
    with open(out, "wb") as f:
        np.array([N, S or 0], np.uint64).tofile(f)
        if N:
            cards.tofile(f)
            sigs.tofile(f)
    with open(out + ".names.txt", "w") as f:
        f.write("#Name\tCardinality\n")
        for i in range(N):
            f.write("%s\t%0.24g\n" % (paths[i], cards[i]))
    for r in range(world):
        for suffix in ("", ".names.txt", ".files.txt"):
            try:
                os.remove(f"{out}.shard{r}{suffix}")
            except OSError:
                pass
    return N, S, cards, sigs


def sketch_main(argv):
    """python -m dashing2_amd.dist sketch -F files.txt -o stacked.bin [any `dashing2 sketch` flags]"""
    ap = argparse.ArgumentParser(prog="dashing2_amd.dist sketch")
    ap.add_argument("-F", "--ffile", required=True)
    ap.add_argument("-o", "--outfile", required=True)
    args, passthrough = ap.parse_known_args(argv)
    if "WORLD_SIZE" not in os.environ:
        # no launcher: the product path -- ONE `dashing2 sketch` process whose C++ host deals the input groups to a pair of device
        # threads per GPU (D2G_DEVICES; dashing2_main.cpp: sketch_core).  The rank-per-GPU form below is what runs under a launcher.
        import subprocess
        exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "dashing2")
        env = dict(os.environ)
        env.setdefault("D2G_DEVICES", "all")
        return subprocess.call([exe, "sketch", "-F", args.ffile, "-o", args.outfile] + passthrough, env=env)
    dist = _dist()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)     # a barrier is all this path needs
    paths = [l.rstrip("\n") for l in open(args.ffile) if l.strip()]
    sketch_sharded(paths, args.outfile, passthrough)
    if world > 1:
        dist.destroy_process_group()
    return 0


def load_stacked(path):
    """[u64 N][u64 S][f64 card x N][f64 x N*S]  (src/sketch_core.cpp:130-140, cmp_main.cpp:61-94)"""
    raw = np.fromfile(path, np.uint8)
    N, S = (int(x) for x in raw[:16].view(np.uint64))
    cards = raw[16:16 + 8 * N].view(np.float64).copy()
    sigs = raw[16 + 8 * N:].view(np.float64).reshape(N, S).copy()
    return N, S, cards, sigs


def main(argv=None):
    """python -m dashing2_amd.dist sketch ...   -> file-sharded sketching, one rank per GPU (above)
    python -m dashing2_amd.dist [cmp] ...     -> `dashing2 cmp ...` spread over every visible GPU.

    The multi-GPU comparison itself lives behind the C ABI (d2g_comm_* / d2g_allpairs_*, RCCL linked by libd2g) and
    is driven from ONE process by the drop-in CLI (D2G_DEVICES); this entry point only launches it, with every flag
    of `dashing2 cmp` (measures, --multiset, text or binary output ...) passed through unchanged."""
    argv = sys.argv[1:] if argv is None else list(argv)
    if argv and argv[0] == "sketch":
        return sketch_main(argv[1:])
    if argv and argv[0] in ("cmp", "dist"):
        argv = argv[1:]
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "dashing2")
    env = dict(os.environ)
    env.setdefault("D2G_DEVICES", "all")
    import subprocess
    return subprocess.call([exe, "cmp"] + argv, env=env)


if __name__ == "__main__":
    sys.exit(main())

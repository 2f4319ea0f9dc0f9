#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X dashing2 hot paths.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

Primary metric (BASELINE.json): all-pairs sketch comparison throughput, **pairs/s**, on
BASELINE config 3 at N=1: 10 000 pre-built OPH sketches, S = 1024 (49 995 000 pairs), float32
Jaccard output.  One step = one whole pass of the path over sketches already resident in HBM:
prepare (transpose, per-column dense ids, bit planes) + the pair kernel with its fused epilogue.
Secondary metric in the same JSON line: sketch construction **bases/s** (K1) on BASELINE config 2's
shape (1 000 x 5 Mbp, k=31, S=1024) with the packed bases resident in HBM.

N > 1 (weak scaling, constant pairs per GPU): N_sketches = round(10000 * sqrt(N)); rank 0 owns the
sketches, each step broadcasts them over RCCL, every rank prepares the operand and computes its
pair-balanced row range of the upper triangle.  No other collective on the data path.

PyTorch is plumbing only (device memory, streams, torch.distributed); all computation goes through
the C ABI of libd2g.so.  The oracle is used ONLY for the cpu_baseline leg.
"""
import argparse
import json
import math
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK_LANEOPS = 256 * 4 * 32 * 2.4e9   # 256 CU x 4 SIMD-32 x 2.4 GHz = 7.86e13 lane-ops/s


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--algo", default="auto", choices=["auto", "direct", "bitslice"])
    ap.add_argument("--sketches", type=int, default=10000, help="sketches at 1 GPU (config 3: 10000)")
    ap.add_argument("--sketchsize", type=int, default=1024)
    ap.add_argument("--no-sketch", action="store_true", help="skip the secondary K1 measurement")
    ap.add_argument("--sketch-genomes", type=int, default=1000)
    ap.add_argument("--sketch-len", type=int, default=5_000_000)
    ap.add_argument("--no-multiset", action="store_true", help="skip the secondary K3 (--multiset / BagMinHash) measurement")
    ap.add_argument("--multiset-genomes", type=int, default=1000)
    ap.add_argument("--multiset-batch", type=int, default=250, help="genomes per d2g_bmh_sketch_dev call (8 B of key per k-mer live in HBM)")
    ap.add_argument("--exchange", default="alltoall", choices=["alltoall", "broadcast"],
                    help="N>1: row-sharded sketches + all-to-all/all-gather of the compact operand (default), "
                         "or rank-0 sketches broadcast whole")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="sharded path: run exchange+prepare and the pair kernel of a step back to back instead of "
                         "overlapping step i+1's exchange with step i's pair kernel")
    ap.add_argument("--force-sharded", action="store_true", help="debug: run the N>1 code path at N=1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


def pmc_traffic(kernel_substr, wide_loads):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/r01_c_pmc.json,
    produced by tools/pmc_summary.py from separate FETCH_SIZE / WRITE_SIZE runs of this same
    command).  MI355X_MICROARCH.md (HBM): FETCH_SIZE/WRITE_SIZE are KiB; on gfx950 FETCH_SIZE
    tallies wide (16 B/lane) coalesced reads at half their bytes -> doubled for such kernels;
    narrow-load kernels are reported raw (uncalibrated per the guide)."""
    path = os.path.join(ROOT, "profiles", "r01_c_pmc.json")
    if not os.path.exists(path):
        return None
    try:
        for k, e in json.load(open(path)).items():
            if kernel_substr in k and "hbm_write_bytes" in e:
                rd = e["hbm_read_bytes_x2_wide_load_correction"] if wide_loads else e["hbm_read_bytes_raw"]
                return rd + e["hbm_write_bytes"]
    except Exception:
        return None
    return None


def cpu_baseline(sig_np, cards_np, S, seconds):
    """The oracle's OpenMP all-pairs (reference loop structure) on a bounded row sample."""
    from oracle import oracle as O
    so = None
    try:   # native build for the box's host CPU; falls back to the portable x86-64-v3 build
        so = O.build(march="native", out="/tmp/libd2oracle_native.so")
    except Exception:
        so = None
    lib = O.load(so) if so else O.load()
    import ctypes as C
    ncores = os.cpu_count() or 1
    N = sig_np.shape[0]
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    bs = lib.d2o_default_batchsize(0, S, ncores)

    def run(r0, r1):
        n = sum(N - r - 1 for r in range(r0, r1))
        out = np.empty(n, np.float32)
        t0 = time.perf_counter()
        lib.d2o_allpairs_ut_rows(P(sig_np, C.c_double), P(cards_np, C.c_double), N, S, 0, 31, r0, r1,
                                 P(out, C.c_float), ncores, bs)
        return n, time.perf_counter() - t0

    n, dt = run(0, min(N, 2 * ncores))                    # probe
    rate = n / max(dt, 1e-9)
    rows = int(min(N, max(2 * ncores, seconds * rate / max(N - 1, 1))))
    rows = max(ncores, rows // ncores * ncores)
    n, dt = run(0, min(rows, N))
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": n / dt, "unit": "pairs/s", "cores": ncores, "kind": "port",
            "sample": f"rows [0,{min(rows, N)}) of the same {N}x{S} matrix = {n} pairs in {dt:.2f}s; "
                      f"oracle OpenMP restatement of emit_rectangular+compare, batch={bs}, "
                      f"{'-march=native' if so else '-march=x86-64-v3'}; cpu='{model}'"}


def k3_pmc_traffic():
    """HBM bytes per K3 call (250 genomes x 5 Mbp) from the committed PMC passes, or None"""
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_d_k3_pmc.json")
    try:
        d = json.load(open(p))
        return float(sum(v.get("fetch_bytes_raw", 0.0) + v.get("write_bytes", 0.0) for k, v in d.items() if k.startswith("k3_")))
    except (OSError, ValueError):
        return None


def cpu_baseline_multiset(L, k, S):
    """The oracle's --multiset restatement (sort + run-length Counter, time-ordered BagMinHash) on all
    host cores: one thread per input, like the reference's OpenMP loop over files (fastxsketch.cpp:302)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    from dashing2_amd import synth
    so = "/tmp/libd2oracle_native.so"
    lib = O.load(so) if os.path.exists(so) else O.load()
    import ctypes as C
    ncores = os.cpu_count() or 1
    Ls = min(L, 2_000_000)
    buf = synth.fasta_bytes("g", synth.random_genome(7, Ls))

    def one(_):
        sig = np.empty(S, np.float64)
        tw, nk = C.c_double(), C.c_uint64()
        lib.d2o_bmh_sketch_buffer(buf, len(buf), k, 1, 0, S, 0.0, sig.ctypes.data_as(C.POINTER(C.c_double)), C.byref(tw), C.byref(nk))
        return tw.value

    t0 = time.perf_counter()
    with ThreadPoolExecutor(ncores) as ex:
        tws = list(ex.map(one, range(ncores)))
    dt = time.perf_counter() - t0
    assert all(t == Ls - k + 1 for t in tws)
    return {"value": ncores * Ls / dt, "unit": "bases/s", "cores": ncores, "kind": "port",
            "sample": f"{ncores} inputs of {Ls} bp sketched concurrently (one thread each) in {dt:.2f}s; oracle restatement of "
                      f"Counter + BagMinHash (BMH-D2G spec), k={k}, S={S}"}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    import dashing2_amd as D
    from dashing2_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    ctx = D.Context(local_rank)
    algo = {"auto": D.CMP_AUTO, "direct": D.CMP_DIRECT, "bitslice": D.CMP_BITSLICE}[args.algo]
    S = args.sketchsize
    N = int(round(args.sketches * math.sqrt(world)))
    if world > 1:
        N = (N + world - 1) // world * world                 # equal row blocks per rank
    pairs_total = N * (N - 1) // 2
    bounds = D.ut_partition(N, world)
    r0, r1 = bounds[rank], bounds[rank + 1]
    my_pairs = D.ut_count(N, r0, r1)
    sharded = (world > 1 and args.exchange == "alltoall") or args.force_sharded

    # ---- synthetic pre-built sketches, resident in HBM before the timed region.
    # N == 1 or --exchange broadcast: the whole matrix lives on rank 0.
    # N > 1 (default): rank r holds rows [r N/W, (r+1) N/W) -- what sharded sketching leaves behind.
    sig_np = cards_np = None
    if rank == 0:
        regs = synth.synthetic_registers(N, S, nclusters=max(8, N // 150), seed=20260928)
        sig_np, cards_np = D.oph_finalize(regs, S, nthreads=os.cpu_count() or 1)
        sig_dev = torch.from_numpy(sig_np.view(np.int64)).to(dev)
    else:
        sig_dev = torch.empty((N, S), dtype=torch.int64, device=dev)
    lut = torch.from_numpy(D.epilogue_lut(S, D.SIMILARITY, 31)).to(dev)
    out = torch.empty(max(my_pairs, 1), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    if world > 1:
        dist.broadcast(sig_dev, 0)                           # untimed distribution of the synthetic input
    eng = cs = None
    pipelined = not args.no_pipeline
    exchange_fallback = None
    if sharded:
        # first (untimed) pass under a guard: if the row-sharded exchange fails on any rank (every rank
        # learns it through one flag all-reduce) the run falls back to the whole-matrix broadcast
        # instead of losing the measurement
        from dashing2_amd import dist as DD
        n_loc = N // world
        my_rows = sig_dev[rank * n_loc:(rank + 1) * n_loc].clone()
        err = None
        try:
            eng = DD.RowShardedAllPairs(ctx, N, S, dev)
            assert (eng.r0, eng.r1) == (r0, r1)
            eng.step_lut(my_rows, lut, out, stream)
            torch.cuda.synchronize()
        except Exception as e:                                   # noqa: BLE001 - reported in the JSON line
            err = f"{type(e).__name__}: {e}"
        if world > 1:
            flag = torch.tensor([1.0 if err else 0.0], dtype=torch.float64, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if flag.item() > 0 and err is None:
                err = "row-sharded exchange failed on another rank"
        if err:
            exchange_fallback = err
            sharded = False
            eng = None
    if sharded:
        del sig_dev

        def step():
            # all-to-all (rows -> column slices), prepare S/W columns, all-gather planes, pair kernel;
            # pipelined: the exchange + prepare of the next step overlap this step's pair kernel (two
            # operand buffers, own stream) -- every step still does all of its work inside the timed region
            if pipelined:
                eng.enqueue_lut(my_rows, lut, out)
            else:
                eng.step_lut(my_rows, lut, out, stream)
        cs = eng.full
    else:
        cs = ctx.cmp_set_dev(sig_dev.data_ptr(), N, S, algo=algo, stream=stream)

        def step():
            if world > 1:
                dist.broadcast(sig_dev, 0)                      # the path's one exchange (RCCL over xGMI)
            cs.update_dev(sig_dev.data_ptr(), stream)           # transpose + ids + planes (async)
            cs.lut_ut_dev(lut.data_ptr(), out.data_ptr(), r0, r1, stream)   # pair kernel + fused epilogue

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_run():
        for _ in range(args.warmup):
            step()
        barrier()
        ctx.set_timing(True)
        ctx.kernel_ms("k2"), ctx.kernel_ms("k2prep")
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        ctx.set_timing(False)
        return dt

    dt = timed_run()
    pipeline_check = None
    if sharded and pipelined:
        # the pipelined steps must have produced exactly what one plain step produces; if they did not,
        # the measurement is repeated unpipelined so that the reported number is never from a wrong run
        got = out.clone()
        eng.step_lut(my_rows, lut, out, stream)
        torch.cuda.synchronize()
        same = torch.tensor([1.0 if torch.equal(got, out) else 0.0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(same, op=dist.ReduceOp.MIN)
        pipeline_check = "ok"
        if same.item() != 1.0:
            pipeline_check = "MISMATCH: re-measured without the overlap"
            pipelined = False
            dt = timed_run()
    nk2, k2_ms, _ = ctx.kernel_ms("k2")
    _, prep_ms, _ = ctx.kernel_ms("k2prep")
    max_distinct, nbits, mean_nbits = cs.planes(stream)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = pairs_total / (dt / args.steps)

    # ---- roofline of the dominant kernel (the pair kernel), rank 0's launch
    alg_bytes = 8 * S * N + 4 * my_pairs          # each sketch read once + one float per pair
    achieved = alg_bytes / (k2_ms * 1e-3) / 1e9 if k2_ms > 0 else 0.0
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic("k2_bitslice_kernel", False) if (cs.algo == D.CMP_BITSLICE and world == 1 and N == 10000 and S == 1024) else None,
                "traffic_note": "rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE (separate passes) of the same workload, profiles/r01_c_pmc.json; dword loads: read side raw/uncalibrated",
                "kernel": "k2_bitslice_kernel" if cs.algo == D.CMP_BITSLICE else "k2_direct_kernel",
                "kernel_ms": k2_ms, "launches": nk2, "algorithmic_bytes": alg_bytes,
                "prep_ms": prep_ms,
                "note": "equality counting is VALU-bound, not HBM-bound (SURVEY 8d); see compute"}
    if cs.algo == D.CMP_BITSLICE:
        ops = my_pairs * ((S + 31) // 32) * (mean_nbits + 2)  # id planes + unique plane (v_bitop3) + v_bcnt
    else:
        ops = my_pairs * S * 2
    compute = {"bound": "valu", "unit": "lane-ops/s", "achieved": ops / (k2_ms * 1e-3) if k2_ms > 0 else 0.0,
               "peak": VALU_PEAK_LANEOPS, "frac": (ops / (k2_ms * 1e-3) / VALU_PEAK_LANEOPS) if k2_ms > 0 else 0.0,
               "bit_planes_max": nbits, "bit_planes_mean": mean_nbits, "max_shared_values_per_column_plus1": max_distinct}

    def guarded(fn):
        """A secondary leg does LOCAL work only and returns (seconds, build(seconds_max) -> dict).  One
        collective afterwards carries the failure flag and the max time, so an exception on any rank
        becomes {"error": ...} on all of them instead of hanging the others or losing the primary line."""
        err, secs, build = None, 0.0, None
        try:
            secs, build = fn()
        except Exception as e:                                   # noqa: BLE001 - reported, not swallowed
            err = f"{type(e).__name__}: {e}"
        if world > 1:
            t = torch.tensor([1.0 if err else 0.0, secs], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            if t[0].item() > 0 and err is None:
                err = "failed on another rank"
            secs = float(t[1].item())
        return {"error": err} if err else build(secs)

    # ---- secondary: K1 sketch construction, packed bases resident in HBM
    def sketch_leg():
        n_g, L, k = args.sketch_genomes, args.sketch_len, 31
        n_g = max(1, n_g // world * world) // world          # genomes are sharded one-per-rank, no collectives
        Lb = (L + 3) // 4
        packed = torch.randint(0, 256, (n_g * Lb + 64,), dtype=torch.uint8, device=dev)   # uniform random 2-bit bases
        run_start = (np.arange(n_g, dtype=np.uint64) * np.uint64(Lb * 4))
        run_len = np.full(n_g, L, np.uint32)
        goff = np.arange(n_g + 1, dtype=np.uint64)
        plan = ctx.oph_plan(run_start, run_len, goff, k)
        m = D.oph_m(S)
        regs_dev = torch.empty((n_g, m), dtype=torch.int64, device=dev)
        for _ in range(2):
            ctx.oph_sketch_dev(plan, packed.data_ptr(), S, regs_dev.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        ctx.set_timing(True)
        ctx.kernel_ms("k1")
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.oph_sketch_dev(plan, packed.data_ptr(), S, regs_dev.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        sdt = time.perf_counter() - t0
        ctx.set_timing(False)
        _, k1_ms, _ = ctx.kernel_ms("k1")
        # sanity: a sketch of random bases has no empty bucket and id % m == bucket
        chk = regs_dev[0].cpu().numpy().view(np.uint64)
        assert ((chk & np.uint64(m - 1)) == np.arange(m, dtype=np.uint64)).all()
        bases = n_g * L
        k1_bytes = n_g * (Lb + 8 * m)
        ach = k1_bytes / (k1_ms * 1e-3) / 1e9 if k1_ms > 0 else 0.0

        def build(sdt):
            return {"metric": "sketch input bases/s (K1, packed bases resident in HBM)", "value": bases * world / (sdt / reps),
                    "unit": "bases/s", "ms_per_step": sdt / reps * 1e3,
                    "config": {"workload": f"{n_g * world} synthetic random genomes x {L} bp, k=31, S={S}, OPH, canonical"},
                    "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": ach / HBM_PEAK_GBS,
                                 "traffic": pmc_traffic("k1_oph_kernel", True) if (world == 1 and n_g == 1000 and L == 5_000_000) else None,
                                 "kernel": "k1_oph_kernel",
                                 "kernel_ms": k1_ms, "algorithmic_bytes": k1_bytes}}
        return sdt, build

    sketch = None if args.no_sketch else guarded(sketch_leg)

    # ---- secondary: K3 --multiset sketch construction (BASELINE config 5: k=21, S=2048, exact k-mer
    # counts -> BagMinHash), packed bases resident in HBM
    def multiset_leg():
        n_g, L, k3, S3 = args.multiset_genomes, args.sketch_len, 21, 2048
        n_g = max(1, n_g // world * world) // world          # inputs are sharded one-per-rank, no collectives
        nb = max(1, min(args.multiset_batch, n_g))
        n_g = n_g // nb * nb if n_g >= nb else n_g
        Lb = ((L + 3) // 4 + 63) // 64 * 64
        packed = torch.randint(0, 256, (n_g * Lb + 64,), dtype=torch.uint8, device=dev)
        plan = ctx.oph_plan(np.arange(nb, dtype=np.uint64) * np.uint64(Lb * 4), np.full(nb, L, np.uint32),
                            np.arange(nb + 1, dtype=np.uint64), k3)
        sig3 = torch.empty((n_g, S3), dtype=torch.float64, device=dev)
        tw3 = torch.empty((n_g,), dtype=torch.float64, device=dev)

        def k3_pass():
            for b0 in range(0, n_g, nb):
                ctx.bmh_sketch_dev(plan, packed.data_ptr() + b0 * Lb, S3, sig3[b0:].data_ptr(), tw3[b0:].data_ptr(), stream=stream)

        k3_pass()
        torch.cuda.synchronize()
        ctx.set_timing(True)
        ctx.kernel_ms("k3")
        reps = 2
        t0 = time.perf_counter()
        for _ in range(reps):
            k3_pass()
        torch.cuda.synchronize()
        mdt = time.perf_counter() - t0
        ctx.set_timing(False)
        ncalls, k3_ms, _ = ctx.kernel_ms("k3")
        k3_bytes = nb * ((L + 3) // 4 + 8 * S3 + 8)
        ach = k3_bytes / (k3_ms * 1e-3) / 1e9 if k3_ms > 0 else 0.0
        assert bool(torch.isfinite(sig3).all()) and bool((tw3 == float(L - k3 + 1)).all())
        cpu_ms = cpu_baseline_multiset(L, k3, S3) if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None

        def build(mdt):
            out = {"metric": "multiset sketch input bases/s (K3: exact k-mer counts + BagMinHash, packed bases resident in HBM)",
                   "value": n_g * L * world / (mdt / reps), "unit": "bases/s", "ms_per_step": mdt / reps * 1e3,
                   "config": {"workload": f"BASELINE config 5: {n_g * world} synthetic random genomes x {L} bp, k={k3}, S={S3}, "
                                          f"--multiset (BagMinHash), canonical, {nb} genomes per call"},
                   "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                "traffic": k3_pmc_traffic() if (world == 1 and nb == 250 and L == 5_000_000) else None,
                                "traffic_note": "sum over the chain's kernels of rocprofv3 --pmc FETCH_SIZE (raw) + WRITE_SIZE, same shape, profiles/r01_d_k3_pmc.json",
                                "kernel": "k3 chain (hist, scan, scatter, bmh_main, verify)",
                                "kernel_ms": k3_ms, "launches": ncalls, "algorithmic_bytes": k3_bytes,
                                "note": "per call of %d genomes; the chain also writes and re-reads 8 B of key per k-mer "
                                        "(bucketed multi-split), which the compulsory-byte figure does not count" % nb},
                   "parity": "bit-exact vs oracle/d2_bmh_oracle.c (published BagMinHash under the BMH-D2G spec; the "
                             "reference's sketch/bmh.h is absent: parity unpinned against a real dashing2 binary)"}
            if cpu_ms is not None:
                out["cpu_baseline"] = cpu_ms
            return out
        return mdt, build

    multiset = None if args.no_multiset else guarded(multiset_leg)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(sig_np, cards_np, S, args.cpu_seconds)

    if rank == 0:
        line = {
            "metric": "all-pairs sketch comparison throughput (pairs/s)", "value": value, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": f"BASELINE config 3 (x sqrt(n_gpus) sketches): {N} pre-built OPH sketches, S={S}, "
                                   f"all-pairs cmp only, {pairs_total} pairs, float32 Jaccard",
                       "sketches": N, "sketchsize": S, "pairs": pairs_total, "algo": "bitslice" if cs.algo == D.CMP_BITSLICE else "direct",
                       "step": ("all-to-all rows->column slices + per-rank prepare of S/W columns + all-gather of bit planes + pair kernel w/ fused epilogue; row-sharded sketches resident in HBM"
                                if sharded else "RCCL broadcast (n_gpus>1) + prepare + pair kernel w/ fused epilogue; sketches resident in HBM"),
                       "parallelism": f"upper-triangle rows sharded over {world} GPU(s) by pair count",
                       **({"exchange_fallback": exchange_fallback} if exchange_fallback else {}),
                       **({"pipelined_exchange": pipeline_check} if pipeline_check else {})},
            "roofline": roofline, "compute": compute, "cpu_baseline": cpu, "sketch": sketch, "multiset_sketch": multiset,
        }
        print(json.dumps(line))
    if eng is not None:
        eng.close()
    else:
        cs.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

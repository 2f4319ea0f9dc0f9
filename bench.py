#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X dashing2 hot paths.

    python bench.py --gpus N --steps K --warmup W

N > 1 runs one process per GPU: either launched by the caller (`python -m torch.distributed.run --nproc-per-node N ...
bench.py --gpus N`: RANK / WORLD_SIZE in the environment) or -- when WORLD_SIZE is not set -- by bench.py itself, which
re-executes under torch.distributed.run with N ranks after checking that N devices are visible (fewer: a JSON line with
"error" and exit code 2, never a silent 1-GPU measurement).

Primary metric (BASELINE.json): all-pairs sketch comparison throughput, **pairs/s**.
N = 1: BASELINE config 3 -- 10 000 pre-built OPH sketches, S = 1024 (49 995 000 pairs), float32
Jaccard output.  One step = one whole pass of the path over sketches already resident in HBM:
prepare (transpose, per-column dense ids, bit planes) + the pair kernel with its fused epilogue.

N > 1 (default --scaling strong): BASELINE config 4 -- 50 000 sketches, S = 1024 (1 249 975 000 pairs),
the SAME total work at every N > 1; rank r holds rows [r N/W, (r+1) N/W) (what sharded sketching leaves
in HBM), one all-to-all + one all-gather of the compact bit-plane operand per step, every rank computes
its pair-balanced row range of the upper triangle.  `value` / `ms_per_step` are ONE JOB's step, the same
definition as at N = 1: exchange + prepare + pair kernel of one matrix, nothing carried over between steps
(inside the step the exchange of chunk c+1 overlaps the prepare of chunk c).  The software-pipelined rate
for a STREAM of matrices (exchange + prepare of step i+1 under the pair kernel of step i) is reported
beside it as `stream_of_matrices`, never as `value`.  `--scaling weak` keeps pairs per GPU constant
instead (N_sketches = 10000 * sqrt(N)).  The N = 1 line also carries `config4_1gpu`: config 4 on one GPU,
the base a strong-scaling curve should be read against.

Secondary objects in the same JSON line (N = 1 measures all of them; N > 1 only with --all-legs, sharded by input):
  compute.matrices  the pair kernel on three matrices: unrelated sketches (1 id plane), the stated one,
                    and an adversarial one where every value occurs exactly twice per column
  sketch            K1 bases/s on BASELINE config 2's shape (1 000 x 5 Mbp, k=31, S=1024): synthetic
                    genomes -> FASTA bytes -> d2g_seqpack (the product's ingest) -> HBM; kernel-only and
                    parse-inclusive rates, and the oracle timed on the host cores beside it
  multiset_sketch   K3 bases/s on BASELINE config 5's shape (k=21, S=2048, --multiset)

PyTorch is plumbing only (device memory, streams, torch.distributed); all computation goes through
the C ABI of libd2g.so.  The oracle is used ONLY for the cpu_baseline legs.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK_LANEOPS = 256 * 4 * 32 * 2.4e9   # 256 CU x 4 SIMD-32 x 2.4 GHz = 7.86e13 lane-ops/s
PMC_FILE = os.path.join(ROOT, "profiles", "r03_pmc.json")   # tools/pmc_round.sh -> tools/pmc_summary.py


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--algo", default="auto", choices=["auto", "direct", "bitslice"])
    ap.add_argument("--sketches", type=int, default=0, help="sketches (default: 10000 = config 3 at 1 GPU and for "
                    "--scaling weak; 50000 = config 4 for N > 1 --scaling strong)")
    ap.add_argument("--scaling", default=None, choices=["strong", "weak"],
                    help="N > 1: strong (default) = BASELINE config 4, fixed 50000 sketches; weak = 10000*sqrt(N) sketches")
    ap.add_argument("--sketchsize", type=int, default=1024)
    ap.add_argument("--no-config4", action="store_true", help="N = 1: skip the secondary config-4 (50000 sketches) run")
    ap.add_argument("--no-matrices", action="store_true", help="N = 1: skip the plane-count sensitivity runs")
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
                    help="sharded path: skip the secondary stream-of-matrices measurement (exchange + prepare of step i+1 "
                         "under the pair kernel of step i); the headline is always the one-job step")
    ap.add_argument("--force-sharded", action="store_true", help="debug: run the N>1 code path at N=1")
    ap.add_argument("--all-legs", action="store_true",
                    help="N > 1: also run the secondary sketch / multiset legs (sharded by input, no collectives in their data path); by default an "
                         "N > 1 run measures the all-pairs job only, so that nothing unrelated to it can delay or lose the scaling line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


def pmc_entry(kernel_substr):
    """the committed rocprofv3 --pmc summary of this same command (separate FETCH_SIZE / WRITE_SIZE passes,
    tools/pmc_round.sh + tools/pmc_summary.py), or None"""
    try:
        for k, e in json.load(open(PMC_FILE)).items():
            if kernel_substr in k and "hbm_write_bytes" in e:
                return e
    except (OSError, ValueError):
        return None
    return None


def pmc_traffic(kernel_substr, wide_loads):
    """HBM bytes of the LARGEST dispatch of the kernel in the PMC run of this command -- the launch of the headline
    workload: the same run also launches the kernel on small inputs (the compare half of configs[1], the config-1 CLI),
    and a mean over dispatches mixes those in (345 MB was (7 x 394 + 3) / 8).
    MI355X_MICROARCH.md (HBM): FETCH_SIZE/WRITE_SIZE are KiB; on gfx950 FETCH_SIZE tallies wide (16 B/lane) coalesced
    reads at half their bytes -> doubled for such kernels; narrow-load kernels are reported raw (uncalibrated per the guide)."""
    e = pmc_entry(kernel_substr)
    if e is None or "largest_dispatch_hbm_write_bytes" not in e:
        return None
    rd = e["largest_dispatch_hbm_read_bytes_raw"] * (2 if wide_loads else 1)
    return rd + e["largest_dispatch_hbm_write_bytes"]


def host_cores():
    """CPUs this process may actually use: the visible CPUs, cut by the affinity mask and by the cgroup CPU quota (a GPU box
    of this pool shows 256 CPUs and grants 16 of them; 256 threads under a 16-CPU quota only throttle each other)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return ""


def oracle_lib():
    """native build of the checker for the box's host CPU (falls back to the portable x86-64-v3 build)"""
    from oracle import oracle as O
    so = None
    try:
        so = O.build(march="native", out="/tmp/libd2oracle_native.so")
    except Exception:
        so = None
    return (O.load(so) if so else O.load()), ("-march=native" if so else "-march=x86-64-v3")


def cpu_baseline(sig_np, cards_np, S, seconds):
    """The oracle's OpenMP all-pairs (reference loop structure) on a bounded row sample."""
    lib, march = oracle_lib()
    import ctypes as C
    ncores = host_cores()
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
    return {"value": n / dt, "unit": "pairs/s", "cores": ncores, "kind": "port",
            "sample": f"rows [0,{min(rows, N)}) of the same {N}x{S} matrix = {n} pairs in {dt:.2f}s; "
                      f"oracle OpenMP restatement of emit_rectangular+compare, batch={bs}, {march}; cpu='{cpu_model()}'"}


def cpu_baseline_sketch(fastas, L, k, S, seconds):
    """K1's CPU leg.  (a) a slice of config 2: the oracle's restatement of the reference's per-file loop
    (fastxsketch.cpp:302: one OpenMP thread per input file) over in-memory FASTA buffers, all host cores,
    repeated until `seconds`/2 have passed; (b) BASELINE config 1 end to end: 32 x 1 Mbp FASTA files on disk
    -> sketch (k=31, S=1024) -> all-pairs --cmpout, the oracle beside the product's CLI."""
    from concurrent.futures import ThreadPoolExecutor
    import ctypes as C
    import subprocess
    import tempfile
    from dashing2_amd import synth
    lib, march = oracle_lib()
    ncores = host_cores()
    m = S + (S & 1)

    def one(buf):
        regs = np.empty(m, np.uint64)
        sig = np.empty(S, np.float64)
        card, nk = C.c_double(), C.c_uint64()
        lib.d2o_sketch_buffer(buf, len(buf), k, 1, 0, S, regs.ctypes.data_as(C.POINTER(C.c_uint64)),
                              sig.ctypes.data_as(C.POINTER(C.c_double)), C.byref(card), C.byref(nk))
        return nk.value

    done, t0 = 0, time.perf_counter()
    with ThreadPoolExecutor(ncores) as ex:
        while True:
            nks = list(ex.map(one, fastas))
            done += len(fastas)
            dt = time.perf_counter() - t0
            if dt >= seconds * 0.5 or done >= 64 * len(fastas):
                break
    assert all(x == L - k + 1 for x in nks)
    out = {"value": done * L / dt, "unit": "bases/s", "cores": ncores, "kind": "port",
           "sample": f"{done} sketches of in-memory {L} bp FASTA inputs ({len(fastas)} distinct genomes of config 2, one thread per input "
                     f"as fastxsketch.cpp:302) in {dt:.2f}s; oracle restatement of Encoder::for_each + maskfn + OPSetSketch::update + "
                     f"getcard/data, k={k}, S={S}, {march}; cpu='{cpu_model()}'"}
    # ---- config 1 end to end
    try:
        with tempfile.TemporaryDirectory(prefix="d2g_c1_") as td:
            paths = []
            for i in range(32):
                p = os.path.join(td, "g%05d.fa" % i)
                synth.write_fasta(p, "g%05d" % i, synth.random_genome(1000 + i, 1_000_000))
                paths.append(p)
            arr = (C.c_char_p * 32)(*[p.encode() for p in paths])
            sigs = np.empty((32, 1024), np.float64)
            cards = np.empty(32, np.float64)
            dist = np.empty(32 * 31 // 2, np.float32)
            nt = min(ncores, 32)
            PD = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                lib.d2o_sketch_files(arr, 32, 31, 1, 0, 1024, PD(sigs), PD(cards), nt)
                for i in range(32):
                    lib.d2o_densify(PD(sigs[i]), 1024)
                lib.d2o_allpairs_ut(PD(sigs), PD(cards), 32, 1024, 0, 31, dist.ctypes.data_as(C.POINTER(C.c_float)), nt, 0)
                dt1 = time.perf_counter() - t0
                best = dt1 if best is None else min(best, dt1)
            c1 = {"workload": "BASELINE config 1: 32 synthetic 1 Mbp FASTA files on disk, k=31, S=1024, sketch + all-pairs (--cmpout)",
                  "cpu_s": best, "cpu_threads": nt, "cpu_bases_per_s": 32e6 / best,
                  "cpu_note": "oracle d2o_sketch_files (file-parallel, fastxsketch.cpp:302) + densify + d2o_allpairs_ut; best of 3"}
            out["config1"] = c1
            exe = os.path.join(ROOT, "dashing2_amd", "bin", "dashing2")
            if os.path.exists(exe):
                lst = os.path.join(td, "files.txt")
                with open(lst, "w") as f:
                    f.write("".join(p + "\n" for p in paths))
                bestg = None
                for _ in range(3):
                    t0 = time.perf_counter()
                    r = subprocess.run([exe, "sketch", "-k", "31", "-S", "1024", "-F", lst, "--cmpout", os.path.join(td, "d.bin"),
                                        "--binary-output", "-o", os.path.join(td, "s.bin"), "-p", str(nt)],
                                       capture_output=True)
                    dtg = time.perf_counter() - t0
                    if r.returncode:
                        raise RuntimeError(r.stderr.decode()[-300:])
                    bestg = dtg if bestg is None else min(bestg, dtg)
                got = np.fromfile(os.path.join(td, "d.bin"), np.float32)
                same = got.size == dist.size and np.array_equal(got.view(np.uint32), dist.view(np.uint32))
                c1["gpu_cli_s"] = bestg
                c1["gpu_cli_note"] = ("`dashing2 sketch --cmpout` process wall time (start-up, GPU context, parse, K1, K2, output); "
                                      "best of 3; distances %s the oracle's" % ("bit-identical to" if same else "DIFFER from"))
    except Exception as e:                                       # noqa: BLE001 - reported in the line
        out.setdefault("config1", {})["error"] = f"{type(e).__name__}: {e}"[:400]
    return out


def cpu_baseline_multiset(L, k, S):
    """The oracle's --multiset restatement (sort + run-length Counter, time-ordered BagMinHash) on all
    host cores: one thread per input, like the reference's OpenMP loop over files (fastxsketch.cpp:302)."""
    from concurrent.futures import ThreadPoolExecutor
    from dashing2_amd import synth
    lib, march = oracle_lib()
    import ctypes as C
    ncores = host_cores()
    Ls = L                                                # config 5's own input size: ~20 s of CPU work on 16 cores
    buf = synth.fasta_bytes_fast("g", synth.random_genome(7, Ls))

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
                      f"Counter + BagMinHash (BMH-D2G spec), k={k}, S={S}, {march}"}


def pack_genomes(D, synth, first, count, L, k, keep=0, nthreads=None):
    """`count` synthetic genomes (indices first..) -> FASTA bytes -> d2g_seqpack (the product's ingest), on
    all host cores; returns one merged packed run stream + the first `keep` FASTA buffers."""
    from concurrent.futures import ThreadPoolExecutor
    nthreads = nthreads or min(host_cores(), 64)
    chunks = [list(range(first + c, min(first + count, first + c + 8))) for c in range(0, count, 8)]

    def work(idx):
        sp = D.SeqPack(k)
        kept = []
        for i in idx:
            fa = synth.fasta_bytes_fast("g%05d" % i, synth.random_genome(i, L))
            sp.add_fastx(fa)
            if i - first < keep:
                kept.append(fa)
        packed, run_start, run_len, goff = sp.arrays()
        res = (packed[:(sp.nbases + 3) // 4], run_start, run_len, goff, kept)
        sp.close()
        return res

    with ThreadPoolExecutor(nthreads) as ex:
        parts = list(ex.map(work, chunks))
    packed, rs, rl, go, kept = [], [], [], [np.zeros(1, np.uint64)], []
    byte_off, run_off = 0, 0
    for p, s, l, g, kp in parts:
        packed.append(p)
        rs.append(s + np.uint64(byte_off * 4))
        rl.append(l)
        go.append(g[1:] + np.uint64(run_off))
        byte_off += p.size
        run_off += s.size
        kept += kp
    packed.append(np.zeros(64, np.uint8))
    return np.concatenate(packed), np.concatenate(rs), np.concatenate(rl), np.concatenate(go), kept


def self_launch(args):
    """`python bench.py --gpus N` with no WORLD_SIZE: run N ranks under torch.distributed.run ourselves (one process per
    GPU, 127.0.0.1 rendezvous) -- after checking that N devices are there.  Returns the exit code."""
    import socket
    import subprocess
    import dashing2_amd as D
    have = int(D.lib().d2g_device_count())
    if have < args.gpus:
        print(json.dumps({"metric": "all-pairs sketch comparison throughput (pairs/s)", "value": None, "unit": "pairs/s",
                          "n_gpus": args.gpus, "error": f"--gpus {args.gpus} requested but {have} HIP device(s) visible: refusing to "
                                                         "measure a smaller job under that label"}), flush=True)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // args.gpus)))
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.force_sharded:
        raise SystemExit(self_launch(args))
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
    scaling = args.scaling or "strong"
    if world == 1:
        N = args.sketches or 10000
        workload = "BASELINE config 3" if (N, S) == (10000, 1024) else "custom"
    elif scaling == "strong":
        N = args.sketches or 50000
        workload = "BASELINE config 4" if (N, S) == (50000, 1024) else "custom"
    else:
        N = int(round((args.sketches or 10000) * math.sqrt(world)))
        workload = "BASELINE config 3 x sqrt(n_gpus) sketches (constant pairs per GPU)"
    if world > 1:
        N = (N + world - 1) // world * world                 # equal row blocks per rank
    pairs_total = N * (N - 1) // 2
    bounds = D.ut_partition(N, world)
    r0, r1 = bounds[rank], bounds[rank + 1]
    my_pairs = D.ut_count(N, r0, r1)
    sharded = (world > 1 and args.exchange == "alltoall") or args.force_sharded
    stream = torch.cuda.current_stream().cuda_stream
    ncores = host_cores()

    def make_sketches(n, seed=20260928):
        regs = synth.synthetic_registers(n, S, nclusters=max(8, n // 150), seed=seed)
        return D.oph_finalize(regs, S, nthreads=ncores)

    # ---- synthetic pre-built sketches, resident in HBM before the timed region.
    # N == 1 or --exchange broadcast: the whole matrix lives on rank 0.
    # N > 1 (default): rank r holds rows [r N/W, (r+1) N/W) -- what sharded sketching leaves behind.
    sig_np = cards_np = None
    if rank == 0:
        sig_np, cards_np = make_sketches(N)
        sig_dev = torch.from_numpy(sig_np.view(np.int64)).to(dev)
    else:
        sig_dev = torch.empty((N, S), dtype=torch.int64, device=dev)
    lut = torch.from_numpy(D.epilogue_lut(S, D.SIMILARITY, 31)).to(dev)
    out = torch.empty(max(my_pairs, 1), dtype=torch.float32, device=dev)
    if world > 1:
        dist.broadcast(sig_dev, 0)                           # untimed distribution of the synthetic input
    eng = cs = comm = None
    pipelined = False                 # the headline step is always ONE job's step; the stream form is measured afterwards
    exchange_fallback = None
    engine_kind = None
    if sharded:
        # The N > 1 data path: the row-sharded engine behind the C ABI (d2g_comm_* / d2g_allpairs_*: RCCL linked by
        # libd2g itself; torch.distributed only carries the 128-byte unique id, the barrier and the timing reductions).
        # First (untimed) pass under a guard: a failure on any rank (every rank learns it through one flag all-reduce)
        # falls back to the torch.distributed form of the same exchange, then to the whole-matrix broadcast, instead of
        # losing the measurement.
        def all_failed(err):
            if world > 1:
                flag = torch.tensor([1.0 if err else 0.0], dtype=torch.float64, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                if flag.item() > 0 and err is None:
                    err = "failed on another rank"
            return err

        # every collective below is entered by ALL ranks or by none: a rank that fails alone must not leave the others
        # blocked inside an RCCL call, so each local stage is followed by a flag all-reduce before the next collective
        err = None
        uid = [None]
        if rank == 0:
            try:
                uid = [D.comm_unique_id()]
            except Exception as e:                               # noqa: BLE001
                err = f"C-ABI engine: {type(e).__name__}: {e}"
        if world > 1:
            dist.broadcast_object_list(uid, src=0)
        if uid[0] is None and err is None:
            err = "C-ABI engine: no RCCL unique id from rank 0"
        if err is None:
            try:
                comm = D.Comm.create(ctx, rank, world, uid[0])   # ncclCommInitRank: collective, all ranks got the id
                import ctypes
                ctypes.CDLL(None).fflush(None)   # RCCL prints a version banner through C stdio: out now, not after the JSON line
                eng = D.AllPairs(ctx, comm, N, S)
                assert eng.rows_computed == (r0, r1)
                lo, hi = eng.rows_held
                my_rows = sig_dev[lo:hi].clone()
            except Exception as e:                               # noqa: BLE001 - reported in the JSON line
                err = f"C-ABI engine: {type(e).__name__}: {e}"
        err = all_failed(err)
        if err is None:
            try:
                eng.step_lut_dev(my_rows.data_ptr(), lut.data_ptr(), out.data_ptr(), stream)
                torch.cuda.synchronize()
            except Exception as e:                               # noqa: BLE001
                err = f"C-ABI engine: {type(e).__name__}: {e}"
        err = all_failed(err)
        if err is None:
            engine_kind = "libd2g (d2g_allpairs over d2g_comm: RCCL send/recv groups)"

            def step():
                # ONE job: all-to-all (rows -> column slices), prepare of S/W columns, all-gather of the planes, pair kernel --
                # chunk by chunk inside the step (the exchange of chunk c+1 under the prepare of chunk c), nothing carried over
                eng.step_lut_dev(my_rows.data_ptr(), lut.data_ptr(), out.data_ptr(), stream)

            def stream_step():
                # a STREAM of matrices: exchange + prepare of the next step under this step's pair kernel (two operand buffers)
                eng.enqueue_lut_dev(my_rows.data_ptr(), lut.data_ptr(), out.data_ptr(), stream, input_ready=True)
            plain_step = step
            cs = eng.operand()
        else:
            exchange_fallback = err
            eng = None
            err2 = None
            try:
                from dashing2_amd import dist as DD
                n_loc = N // world
                my_rows = sig_dev[rank * n_loc:(rank + 1) * n_loc].clone()
                teng = DD.RowShardedAllPairs(ctx, N, S, dev)
                assert (teng.r0, teng.r1) == (r0, r1)
                teng.step_lut(my_rows, lut, out, stream)
                torch.cuda.synchronize()
            except Exception as e:                               # noqa: BLE001
                err2 = f"torch engine: {type(e).__name__}: {e}"
            err2 = all_failed(err2)
            if err2 is None:
                engine_kind = "torch.distributed (dashing2_amd.dist.RowShardedAllPairs)"
                eng = teng

                def step():
                    teng.step_lut(my_rows, lut, out, stream)

                def stream_step():
                    teng.enqueue_lut(my_rows, lut, out, ready=False)    # my_rows was complete before the timed region
                plain_step = step
                cs = teng.full
            else:
                exchange_fallback += " | " + err2
                sharded = False
    slab_check = None
    if sharded:
        # The multi-GPU exchange has never run on hardware before the driver's scaling run: every rank checks the first and the
        # last rows of the slab the sharded step just wrote against a single-GPU computation over the WHOLE matrix (which it still
        # holds from the untimed distribution).  A mismatch anywhere is reported in the line ("valid": false), never hidden.
        try:
            ref = ctx.cmp_set_dev(sig_dev.data_ptr(), N, S, algo=algo, stream=stream)
            ok = True
            for a, z in ((r0, min(r0 + 2, r1)), (max(r0, r1 - 2), r1)):
                if z <= a:
                    continue
                n_chk = D.ut_count(N, a, z)
                want = torch.empty(n_chk, dtype=torch.float32, device=dev)
                ref.lut_ut_dev(lut.data_ptr(), want.data_ptr(), a, z, stream)
                torch.cuda.synchronize()
                o0 = D.ut_count(N, r0, a)
                ok = ok and bool(torch.equal(want.view(torch.int32), out[o0:o0 + n_chk].view(torch.int32)))
            ref.close()
            del ref
        except Exception as e:                                   # noqa: BLE001
            ok = False
            exchange_fallback = (exchange_fallback or "") + f" | slab check failed to run: {type(e).__name__}: {e}"
        flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        slab_check = bool(flag.item() == 1.0)
        del sig_dev
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

    def timed_run(step):
        # An event pair in the stream costs a few microseconds of device time per launch (measured: 0.530 ms per step with no
        # events, 0.549 with the pair kernel AND the prepare chain bracketed): the timed region brackets only the kernel the
        # roofline reports -- every launch of it --, the prepare chain is timed during the (untimed) warmup steps.
        ctx.set_timing(D.TIME_K2PREP)
        ctx.kernel_ms("k2prep")
        for _ in range(args.warmup if args.warmup > 0 else 1):       # --warmup 0: one untimed step still, for the prepare timing
            step()
        barrier()
        ctx.set_timing(D.TIME_K2)
        ctx.kernel_ms("k2")
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        ctx.set_timing(False)
        return dt

    dt = timed_run(step)
    nk2, k2_ms, _ = ctx.kernel_ms("k2")
    _, prep_ms, _ = ctx.kernel_ms("k2prep")
    # the gathered prepare status: a rank-table overflow on ANY rank invalidates the step everywhere (never silent)
    if sharded and engine_kind and engine_kind.startswith("libd2g"):
        eng.status(stream)
    stream_of_matrices = None
    if sharded and not args.no_pipeline:
        # secondary: the software-pipelined rate for a stream of matrices.  Probed on every rank first (a failure anywhere
        # skips it everywhere), timed like the headline, and its output checked against the plain step's.
        want = out.clone()
        perr = None
        try:
            stream_step(); stream_step()
            torch.cuda.synchronize()
        except Exception as e:                                   # noqa: BLE001
            perr = f"{type(e).__name__}: {e}"
        perr = all_failed(perr)
        if perr is None:
            sdt = timed_run(stream_step)
            ctx.kernel_ms("k2")
            if world > 1:
                t = torch.tensor([sdt], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                sdt = float(t.item())
            same = torch.tensor([1.0 if torch.equal(want, out) else 0.0], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(same, op=dist.ReduceOp.MIN)
            stream_of_matrices = {"value": pairs_total / (sdt / args.steps), "unit": "pairs/s", "ms_per_step": sdt / args.steps * 1e3,
                                  "outputs_identical_to_the_one_job_step": bool(same.item() == 1.0),
                                  "note": "NOT the headline: throughput over repeated matrices with the exchange + prepare of step i+1 hidden "
                                          "under the pair kernel of step i (d2g_allpairs_enqueue_lut_dev); BASELINE config 4 is one job"}
        else:
            stream_of_matrices = {"error": perr}
        del want
    max_distinct, nbits, mean_nbits = cs.planes(stream)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = pairs_total / (dt / args.steps)
    algo_used = cs.algo
    ops_extra = getattr(D, "BITSLICE_OPS_PER_GROUP_EXTRA", 2)   # VALU ops per pair and 32-register group beyond the id planes

    # ---- roofline of the dominant kernel (the pair kernel), rank 0's launch
    alg_bytes = 8 * S * N + 4 * my_pairs          # each sketch read once + one float per pair
    achieved = alg_bytes / (k2_ms * 1e-3) / 1e9 if k2_ms > 0 else 0.0
    kname = "k2_bitslice_kernel" if algo_used == D.CMP_BITSLICE else "k2_direct_kernel"
    pmc_ok = (algo_used == D.CMP_BITSLICE and world == 1 and N == 10000 and S == 1024)
    with_prep = alg_bytes / ((k2_ms + prep_ms) * 1e-3) / 1e9 if (k2_ms + prep_ms) > 0 else 0.0
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                # the 8 S N bytes of the sketches are read by the PREPARE chain, not by the pair kernel: the same algorithmic
                # bytes over pair kernel + prepare
                "frac_with_prepare": with_prep / HBM_PEAK_GBS,
                "traffic": pmc_traffic(kname, False) if pmc_ok else None,
                "traffic_note": "rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE (separate passes) of this command, the config-3-sized (largest) "
                                "dispatch, profiles/" + os.path.basename(PMC_FILE) + "; dword loads: read side raw/uncalibrated",
                "kernel": kname, "kernel_ms": k2_ms, "launches": nk2, "algorithmic_bytes": alg_bytes,
                "prep_ms": prep_ms,
                "note": "equality counting is VALU-bound, not HBM-bound (SURVEY 8d); see compute"}

    def valu(pairs, mean_planes, ms):
        if algo_used == D.CMP_BITSLICE:
            ops = pairs * ((S + 31) // 32) * (mean_planes + ops_extra)
        else:
            ops = pairs * S * 2
        a = ops / (ms * 1e-3) if ms > 0 else 0.0
        return a, a / VALU_PEAK_LANEOPS

    va, vf = valu(my_pairs, mean_nbits, k2_ms)
    compute = {"bound": "valu", "unit": "lane-ops/s", "achieved": va, "peak": VALU_PEAK_LANEOPS, "frac": vf,
               "bit_planes_max": nbits, "bit_planes_mean": mean_nbits, "max_shared_values_per_column_plus1": max_distinct}

    def measure_matrix(bits_np, n, steps=5):
        """prepare + pair kernel of an n x S matrix on this GPU, whole triangle; returns a small dict"""
        t_dev = torch.from_numpy(bits_np.view(np.int64)).to(dev)
        o = torch.empty(n * (n - 1) // 2, dtype=torch.float32, device=dev)
        c = ctx.cmp_set_dev(t_dev.data_ptr(), n, S, algo=algo, stream=stream)
        ctx.set_timing(D.TIME_K2PREP)                       # as in timed_run: the prepare chain is timed on the untimed steps
        ctx.kernel_ms("k2prep")
        for _ in range(2):
            c.update_dev(t_dev.data_ptr(), stream)
            c.lut_ut_dev(lut.data_ptr(), o.data_ptr(), 0, n, stream)
        torch.cuda.synchronize()
        ctx.set_timing(D.TIME_K2)
        ctx.kernel_ms("k2")
        t0 = time.perf_counter()
        for _ in range(steps):
            c.update_dev(t_dev.data_ptr(), stream)
            c.lut_ut_dev(lut.data_ptr(), o.data_ptr(), 0, n, stream)
        torch.cuda.synchronize()
        d = (time.perf_counter() - t0) / steps
        ctx.set_timing(False)
        _, kms, _ = ctx.kernel_ms("k2")
        _, pms, _ = ctx.kernel_ms("k2prep")
        md, nb, mean = c.planes(stream)
        c.close()
        npairs = n * (n - 1) // 2
        ab = 8 * S * n + 4 * npairs
        return {"sketches": n, "pairs_per_s": npairs / d, "ms_per_step": d * 1e3, "kernel_ms": kms, "prep_ms": pms,
                "bit_planes_max": nb, "bit_planes_mean": mean, "max_shared_values_per_column_plus1": md,
                "hbm_frac": ab / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS if kms > 0 else 0.0,
                "valu_frac": valu(npairs, mean, kms)[1]}

    config4 = None
    if world == 1 and not args.force_sharded:
        del out
        torch.cuda.empty_cache()
        if not args.no_matrices:
            # the plane count -- and with it the pair kernel's time -- depends on how many values a register
            # column shares among sketches: report the two extremes beside the stated matrix
            mats = {}
            try:
                mats["stated (clustered collection, %d clusters)" % max(8, N // 150)] = {
                    "sketches": N, "pairs_per_s": value, "ms_per_step": ms_per_step, "kernel_ms": k2_ms, "prep_ms": prep_ms,
                    "bit_planes_max": nbits, "bit_planes_mean": mean_nbits, "max_shared_values_per_column_plus1": max_distinct,
                    "hbm_frac": achieved / HBM_PEAK_GBS, "valu_frac": vf}
                mats["unrelated (no value shared by two sketches)"] = measure_matrix(synth.unrelated_registers(N, S), N)
                mats["adversarial (every value occurs exactly twice in its column)"] = measure_matrix(synth.paired_registers(N, S), N)
                # columns that share between 0 and 64 values (log-uniform): which column lands in which 32-register group matters,
                # since a group walks the MAXIMUM plane count of its columns.  Measured with the column plan (columns sorted by
                # plane class before grouping, the default) and with the caller's column order (D2G_BS_SORT=0).
                sk = synth.skewed_registers(N, S)
                mats["skewed (columns share 0..64 values, log-uniform)"] = measure_matrix(sk, N)
                os.environ["D2G_BS_SORT"] = "0"
                try:
                    mats["skewed, columns left in the caller's order (D2G_BS_SORT=0)"] = measure_matrix(sk, N)
                finally:
                    os.environ.pop("D2G_BS_SORT", None)
                del sk
            except Exception as e:                               # noqa: BLE001 - reported in the line
                mats["error"] = f"{type(e).__name__}: {e}"
            compute["matrices"] = mats
        if not args.no_config4 and S == 1024:
            try:
                s4, _ = make_sketches(50000, seed=20260929)
                r = measure_matrix(s4.view(np.uint64), 50000, steps=3)
                del s4
                r["workload"] = "BASELINE config 4 on ONE GPU: 50000 pre-built OPH sketches, S=1024, 1249975000 pairs, float32 Jaccard"
                config4 = r
            except Exception as e:                               # noqa: BLE001
                config4 = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()

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

    # ---- secondary: K1 sketch construction (config 2's shape).  Inputs: synthetic genomes rendered as FASTA
    # and ingested by the product's own parser/packer (d2g_seqpack); the packed run stream is resident in
    # HBM when the timed region starts.
    def sketch_leg():
        n_all, L, k = args.sketch_genomes, args.sketch_len, 31
        n_g = max(1, n_all // world * world) // world        # genomes are sharded one-per-rank, no collectives
        keep = min(n_g, max(16, min(2 * ncores, 256)))
        t0 = time.perf_counter()
        packed_np, run_start, run_len, goff, fastas = pack_genomes(D, synth, rank * n_g, n_g, L, k, keep=keep)
        gen_s = time.perf_counter() - t0
        packed = torch.from_numpy(packed_np).to(dev)
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
        # sanity: a sketch of random bases has no empty bucket and id % m == bucket, and genomes differ
        chk = regs_dev[:2].cpu().numpy().view(np.uint64)
        assert ((chk[0] & np.uint64(m - 1)) == np.arange(m, dtype=np.uint64)).all()
        assert n_g < 2 or (chk[0] != chk[1]).any()
        bases = int(plan.nbases)
        assert bases == n_g * L
        k1_bytes = n_g * ((L + 3) // 4 + 8 * m)
        ach = k1_bytes / (k1_ms * 1e-3) / 1e9 if k1_ms > 0 else 0.0

        # parse-inclusive: FASTA bytes in host memory -> parser threads -> packed runs -> H2D -> K1 -> registers
        # on the host (the library's persistent sketcher, one device thread), over the kept sample
        from concurrent.futures import ThreadPoolExecutor
        nthr = max(1, min(ncores, 64, len(fastas)))
        per = (len(fastas) + nthr - 1) // nthr
        sk = ctx.sketcher()
        pools = [D.SeqPack(k) for _ in range(nthr)]

        def parse_job(t):
            sp = pools[t]
            sp.clear()
            for fa in fastas[t * per:(t + 1) * per]:
                sp.add_fastx(fa)
            return sp

        def ingest_once():
            with ThreadPoolExecutor(nthr) as ex:
                for sp in ex.map(parse_job, range(nthr)):
                    if sp.ngenomes:
                        sk.run(sp, S)

        ingest_once()
        t0 = time.perf_counter()
        ingest_once()
        idt_host = time.perf_counter() - t0
        # the same sample through the DEVICE parser (K0): the FASTA bytes are copied into page-locked staging by the host threads
        # (what read() does in the CLI), cross PCIe raw, and are parsed + 2-bit-packed on the GPU; K1 runs on the device stream
        total_raw = sum((len(f) + 15) // 16 * 16 for f in fastas)
        pin = D.PinnedArray(ctx, total_raw + 64)
        offs = np.zeros(len(fastas), np.uint64)
        pos = 0
        for i, f in enumerate(fastas):
            offs[i] = pos
            pos += (len(f) + 15) // 16 * 16
        lens = np.array([len(f) for f in fastas], np.uint64)
        gfo = np.arange(len(fastas) + 1, dtype=np.uint64)

        def stage_job(t):
            for i in range(t * per, min(len(fastas), (t + 1) * per)):
                pin.array[int(offs[i]):int(offs[i]) + len(fastas[i])] = np.frombuffer(fastas[i], np.uint8)

        def ingest_once_k0():
            with ThreadPoolExecutor(nthr) as ex:
                list(ex.map(stage_job, range(nthr)))
            sk.ingest_raw(pin.array, pos, offs, lens, gfo, k)
            return sk.run_ingested(sk.ingested_runs(len(fastas)), S)

        r_dev = ingest_once_k0()
        ctx.set_timing(D.TIME_K0)
        ctx.kernel_ms("k0")
        t0 = time.perf_counter()
        r_dev = ingest_once_k0()
        idt = time.perf_counter() - t0
        ctx.set_timing(False)
        _, k0_ms, _ = ctx.kernel_ms("k0")
        # the device-parsed registers must be the host-parsed ones
        sp_chk = D.SeqPack(k)
        sp_chk.add_fastx(fastas[0])
        k0_same = bool(np.array_equal(sk.run(sp_chk, S)[0], r_dev[0]))
        sp_chk.close()
        pin.close()
        ingest_rate = len(fastas) * L / idt
        cpu = cpu_baseline_sketch(fastas[:min(len(fastas), 2 * ncores)], L, k, S, args.cpu_seconds) \
            if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
        pe = pmc_entry("k1_oph_kernel") if (world == 1 and n_g == 1000 and L == 5_000_000) else None
        # the compare half of BASELINE configs[1] on exactly these sketches: registers in HBM -> host x87 finalisation ->
        # all-pairs float32 Jaccard back on the host (upload, prepare, pair kernel, D2H), through the host-pointer C ABI
        c2 = None
        if rank == 0 and world == 1 and n_g >= 2:
            try:
                t0 = time.perf_counter()
                regs_h = regs_dev.cpu().numpy().view(np.uint64)
                sg, cd = D.oph_finalize(regs_h, S, nthreads=ncores)
                dm = ctx.cmp_dist_ut(sg.view(np.uint64), cd, nthreads=ncores)
                c2 = {"pairs": int(n_g * (n_g - 1) // 2), "seconds": time.perf_counter() - t0,
                      "note": "the compare half of configs[1] on the sketches just built: D2H of the registers, x87 finalisation (getcard/data), "
                              "d2g_cmp_dist_ut from host pointers (upload + prepare + pair kernel + D2H); values finite: "
                              + str(bool(np.isfinite(dm).all()))}
            except Exception as e:                               # noqa: BLE001
                c2 = {"error": f"{type(e).__name__}: {e}"}

        def build(sdt):
            o = {"metric": "sketch input bases/s (K1 kernel, packed bases resident in HBM)", "value": bases * world / (sdt / reps),
                 "unit": "bases/s", "ms_per_step": sdt / reps * 1e3,
                 "config": {"workload": f"BASELINE config 2 shape: {n_g * world} synthetic random genomes x {L} bp, k=31, S={S}, OPH, canonical",
                            "input": "splitmix64 genomes rendered as 80-column FASTA and ingested through d2g_seqpack (the product's parser + "
                                     f"2-bit packer; {gen_s:.1f}s, untimed)"},
                 "parse_inclusive": {"value": ingest_rate * world, "unit": "bases/s",
                                     "sample": f"{len(fastas)} in-memory FASTA inputs -> {nthr} host threads copy them into page-locked staging -> H2D of the raw bytes -> "
                                               f"K0 (device parser + 2-bit packer, {k0_ms:.3f} ms of kernels) -> K1 -> D2H of the registers in {idt:.3f}s; "
                                               f"registers identical to the host-parsed ones: {k0_same}",
                                     "host_parser": {"value": len(fastas) * L / idt_host * world, "unit": "bases/s",
                                                     "sample": f"the same inputs -> {nthr} parser threads (d2g_seqpack: AVX-512 packer) -> pinned staging -> H2D -> K1 -> D2H in "
                                                               f"{idt_host:.3f}s (the round-2 path; still used for gz / FASTQ inputs)"}},
                 "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": ach / HBM_PEAK_GBS,
                              "traffic": (2 * pe["largest_dispatch_hbm_read_bytes_raw"] + pe["largest_dispatch_hbm_write_bytes"]) if pe and "largest_dispatch_hbm_write_bytes" in pe else None,
                              "traffic_note": "rocprofv3 --pmc FETCH_SIZE (x2: 16-byte coalesced loads are tallied at half their bytes on gfx950) + WRITE_SIZE of the "
                                              "1000-genome launch, profiles/" + os.path.basename(PMC_FILE),
                              "kernel": "k1_oph_kernel",
                              "kernel_ms": k1_ms, "algorithmic_bytes": k1_bytes,
                              "note": "VALU-bound by the two mandated 64-bit Wang mixes per k-mer (~125 issue slots per base), not by HBM"}}
            if cpu is not None:
                o["cpu_baseline"] = cpu
            if c2 is not None:
                o["allpairs_of_these_sketches"] = c2
            return o
        return sdt, build

    secondary = world == 1 or args.all_legs
    sketch = None if (args.no_sketch or not secondary) else guarded(sketch_leg)

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

        reps = 2

        def k3_measure():
            k3_pass()
            torch.cuda.synchronize()
            ctx.set_timing(True)
            ctx.kernel_ms("k3")
            t0 = time.perf_counter()
            for _ in range(reps):
                k3_pass()
            torch.cuda.synchronize()
            d = time.perf_counter() - t0
            ctx.set_timing(False)
            nc, ms, _ = ctx.kernel_ms("k3")
            assert bool(torch.isfinite(sig3).all()) and bool((tw3 == float(L - k3 + 1)).all())
            return d, nc, ms

        def k3_traffic(compact):
            """HBM bytes per call of this variant's kernels from the committed PMC passes, or None.  A call launches some
            kernels once per genome range of its pipeline, so the figure is (bytes summed over all dispatches) / (calls); the PMC
            run makes the same number of calls of either variant, each with one k3_bmh_init_kernel launch."""
            if not (world == 1 and nb == 250 and L == 5_000_000):
                return None
            own = (("k3c_hist", "k3c_scan", "k3c_scatter", "k3_split_kernel", "k3_bmh_main_kernel<true, true>") if compact else
                   ("k3_hist_kernel", "k3_scan_kernel", "k3_scatter_kernel", "k3_refine_kernel", "k3_bmh_main_kernel<false, true>"))
            shared = ("k3_bmh_survivor", "k3_bmh_verify", "k3_bmh_init")
            try:
                d = json.load(open(PMC_FILE))
                ent = lambda w: [v for kk, v in d.items() if w in kk and "hbm_write_bytes" in v and "dispatches" in v]
                init = ent("k3_bmh_init")
                if not init:
                    return None
                calls = init[0]["dispatches"] / 2.0                          # per variant
                tot = lambda v: (v["hbm_read_bytes_raw"] + v["hbm_write_bytes"]) * v["dispatches"]
                t = 0.0
                for w in own:
                    e = ent(w)
                    if not e:
                        return None
                    t += sum(tot(v) for v in e) / calls
                for w in shared:
                    t += sum(tot(v) for v in ent(w)) / (2.0 * calls)
                return float(t)
            except (OSError, ValueError, KeyError, ZeroDivisionError):
                return None

        os.environ.pop("D2G_K3_COMPACT", None)
        mdt, ncalls, k3_ms = k3_measure()
        sig_default = sig3.clone()
        # the low-traffic variant of the same chain (4-byte stored words + tile-sorted split, D2G_K3_COMPACT=1): identical results
        os.environ["D2G_K3_COMPACT"] = "1"
        try:
            cdt, _, ck3_ms = k3_measure()
            same = bool(torch.equal(sig3.view(torch.int64), sig_default.view(torch.int64)))
        finally:
            os.environ.pop("D2G_K3_COMPACT", None)
        del sig_default
        k3_bytes = nb * ((L + 3) // 4 + 8 * S3 + 8)
        ach = k3_bytes / (k3_ms * 1e-3) / 1e9 if k3_ms > 0 else 0.0
        cpu_ms = cpu_baseline_multiset(L, k3, S3) if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
        traffic = k3_traffic(False)
        low_traffic = {"switch": "D2G_K3_COMPACT=1 (k <= 21)", "value": n_g * L * world / (cdt / reps), "unit": "bases/s", "kernel_ms": ck3_ms,
                       "traffic": k3_traffic(True), "registers_identical_to_default": same,
                       "note": "4-byte stored k-mer words, LDS tile sort, coalesced flush: about half the HBM traffic, more time (a Wang mix per "
                               "distinct k-mer moves into the issue-bound main pass, and every bucket is split once more) -- not the default"}

        def build(mdt):
            out = {"metric": "multiset sketch input bases/s (K3: exact k-mer counts + BagMinHash, packed bases resident in HBM)",
                   "value": n_g * L * world / (mdt / reps), "unit": "bases/s", "ms_per_step": mdt / reps * 1e3,
                   "config": {"workload": f"BASELINE config 5: {n_g * world} synthetic random genomes x {L} bp, k={k3}, S={S3}, "
                                          f"--multiset (BagMinHash), canonical, {nb} genomes per call"},
                   "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                "traffic": traffic,
                                "traffic_note": "sum over the chain's kernels of rocprofv3 --pmc FETCH_SIZE (raw) + WRITE_SIZE per call, same shape, profiles/" + os.path.basename(PMC_FILE),
                                "kernel": "k3 chain (hist, scan, scatter, refine, bmh_main, survivors, verify)",
                                "kernel_ms": k3_ms, "launches": ncalls, "algorithmic_bytes": k3_bytes,
                                "note": "per call of %d genomes; the chain also writes and re-reads the bucketed k-mer keys, "
                                        "which the compulsory-byte figure does not count" % nb},
                   "parity": "bit-exact vs oracle/d2_bmh_oracle.c (published BagMinHash under the BMH-D2G spec; the "
                             "reference's sketch/bmh.h is absent: parity unpinned against a real dashing2 binary)"}
            out["low_traffic_variant"] = low_traffic
            if cpu_ms is not None:
                out["cpu_baseline"] = cpu_ms
            return out
        return mdt, build

    multiset = None if (args.no_multiset or not secondary) else guarded(multiset_leg)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(sig_np, cards_np, S, args.cpu_seconds)

    if rank == 0:
        line = {
            "metric": "all-pairs sketch comparison throughput (pairs/s)", "value": value, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak" if (world > 1 and scaling == "weak") else "strong", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": f"{workload}: {N} pre-built OPH sketches, S={S}, all-pairs cmp only, {pairs_total} pairs, float32 Jaccard",
                       "sketches": N, "sketchsize": S, "pairs": pairs_total, "algo": "bitslice" if algo_used == D.CMP_BITSLICE else "direct",
                       "step": ("all-to-all rows->column slices + per-rank prepare of S/W columns + all-gather of bit planes + pair kernel w/ fused epilogue; row-sharded sketches resident in HBM"
                                if sharded else "RCCL broadcast (n_gpus>1) + prepare + pair kernel w/ fused epilogue; sketches resident in HBM"),
                       "parallelism": f"upper-triangle rows sharded over {world} GPU(s) by pair count",
                       **({"exchange_engine": engine_kind} if engine_kind else {}),
                       **({"exchange_fallback": exchange_fallback} if exchange_fallback else {}),
                       **({"exchange_chunks": eng.chunks} if (eng is not None and hasattr(eng, "chunks")) else {}),
                       **({"slab_check": "every rank's first and last slab rows equal a single-GPU computation" if slab_check
                           else "MISMATCH between the sharded step and a single-GPU computation: this line is NOT a valid measurement"}
                          if slab_check is not None else {})},
            "roofline": roofline, "compute": compute, "cpu_baseline": cpu, "config4_1gpu": config4,
            "sketch": sketch, "multiset_sketch": multiset,
        }
        if slab_check is False:
            line["valid"] = False
        if stream_of_matrices is not None:
            line["stream_of_matrices"] = stream_of_matrices
        # compact copies of the two secondary legs INSIDE roofline / cpu_baseline: these two objects are what the driver's
        # record keeps of the line (the full legs stay at top level)
        def brief(leg):
            if not isinstance(leg, dict) or "roofline" not in leg:
                return None, None
            r = leg["roofline"]
            b = {"value": leg.get("value"), "unit": leg.get("unit"), "kernel": r.get("kernel"), "kernel_ms": r.get("kernel_ms"),
                 "achieved": r.get("achieved"), "frac": r.get("frac"), "algorithmic_bytes": r.get("algorithmic_bytes"), "traffic": r.get("traffic")}
            c = leg.get("cpu_baseline")
            cb = {"value": c.get("value"), "unit": c.get("unit"), "cores": c.get("cores"), "kind": c.get("kind")} if isinstance(c, dict) else None
            return b, cb
        for key, leg in (("sketch", sketch), ("multiset_sketch", multiset)):
            b, cb = brief(leg)
            if b is not None:
                roofline[key] = b
            if cb is not None and isinstance(cpu, dict):
                cpu[key] = cb
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(line), flush=True)
    if eng is not None:
        eng.close()
        if comm is not None:
            comm.close()
    else:
        cs.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

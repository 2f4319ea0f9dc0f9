#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X dashing2 hot paths.

    python bench.py --gpus N --steps K --warmup W

Primary metric (BASELINE.json): all-pairs sketch comparison throughput, **pairs/s**.
N = 1: BASELINE config 3 -- 10 000 pre-built OPH sketches, S = 1024 (49 995 000 pairs), float32
Jaccard output.  One step = one whole pass of the path over sketches already resident in HBM:
prepare (transpose, per-column dense ids, bit planes) + the pair kernel with its fused epilogue.

N > 1 (default --scaling strong): BASELINE config 4 -- 50 000 sketches, S = 1024 (1 249 975 000 pairs),
the SAME total work at every N > 1; rank r holds rows [r N/W, (r+1) N/W) (what sharded sketching leaves
in HBM), one all-to-all + one all-gather of the compact bit-plane operand per step, every rank computes
its pair-balanced row range of the upper triangle.  `value` / `ms_per_step` are ONE JOB's step, the same
definition as at N = 1: exchange + prepare + pair kernel of one matrix, nothing carried over between steps.
Every N > 1 line carries its own base: the same job timed on ONE GPU in the same run (`scaling_base`), the
per-phase times of one step on every rank (`phases`) and the cost model's prediction for that world size
(`model`).  The N = 1 line also carries `config4_1gpu`.

**How N > 1 runs -- a supervisor with a watchdog ladder.**  The measuring processes are always CHILDREN of a
supervisor that touches no GPU: started by hand (`python bench.py --gpus N`) the supervisor is this process;
started by the driver under `torch.distributed.run` (RANK / WORLD_SIZE set) rank 0 is the supervisor and the
other ranks exit 0 at once.  The supervisor writes the synthetic matrix to a scratch file once, then tries, each
rung under a timeout, killing the rung's process group on a timeout or a non-zero exit:
  1. `cabi`      one process per GPU; libd2g's own RCCL communicator (ncclCommInitRank) + the row-sharded engine
                 (d2g_allpairs_*); torch.distributed (gloo, CPU) only carries the 128-byte id, barriers, reductions
  2. `inproc`    ONE process driving all N GPUs through d2g_comm_create_all + d2g_allpairs_step_all (ncclCommInitAll;
                 no rendezvous, no second communicator) -- the path the C++ `dashing2 cmp` CLI uses
  3. `torch`     one process per GPU, the same exchange written against torch collectives (backend nccl = RCCL)
  4. `broadcast` one process per GPU, the whole matrix broadcast per step (torch nccl), single-GPU prepare everywhere
and prints the first rung's line that succeeds (`launcher.ladder` says what happened on the way) -- or, if every
rung failed, a JSON line with "error" and exit code 2.  It never exits without a line.  Fewer than N devices
visible: a line with "error", exit code 2, never a smaller job under an `n_gpus: N` label.

Secondary objects in the same JSON line (N = 1 measures all of them):
  compute.matrices  the pair kernel on three matrices: unrelated sketches (1 id plane), the stated one,
                    and an adversarial one where every value occurs exactly twice per column
  sketch            K1 bases/s on BASELINE config 2's shape (1 000 x 5 Mbp, k=31, S=1024): synthetic
                    genomes -> FASTA bytes -> d2g_seqpack (the product's ingest) -> HBM; kernel-only and
                    parse-inclusive rates, and the oracle timed on the host cores beside it
  multiset_sketch   K3 bases/s on BASELINE config 5's shape (k=21, S=2048, --multiset)
`roofline.traffic` (and the legs') is measured IN THIS RUN when rocprofv3 is on the box: a child re-runs one launch
of each reported kernel under `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes); `traffic_source` says so,
or names the committed profile the figure was read from instead.

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
PMC_FILE = os.path.join(ROOT, "profiles", "r06_pmc.json")   # tools/pmc_round.sh -> tools/pmc_summary.py: the fallback when rocprofv3 cannot run here


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
    ap.add_argument("--engines", default=None,
                    help="N > 1: comma-separated rungs of the watchdog ladder to try, in order (default cabi,inproc,torch,broadcast)")
    ap.add_argument("--loopback", action="store_true",
                    help="N > 1 on ONE GPU (tests): all N contexts on device 0 through libd2g's loopback transport; only the `inproc` rung")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="`cabi` rung: skip the secondary stream-of-matrices measurement (exchange + prepare of step i+1 "
                         "under the pair kernel of step i); the headline is always the one-job step")
    ap.add_argument("--all-legs", action="store_true",
                    help="N > 1: after the all-pairs line is out, also run the K1 sketch leg sharded by input (no collectives) and print the line again with it")
    ap.add_argument("--no-traffic", action="store_true", help="do not re-run the reported kernels under rocprofv3 --pmc for roofline.traffic")
    # internal: set by the supervisor for its measuring children
    ap.add_argument("--worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--engine", default=None, choices=["cabi", "inproc", "torch", "broadcast"], help=argparse.SUPPRESS)
    ap.add_argument("--sig-file", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cards-file", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--pmc-child", default=None, help=argparse.SUPPRESS)
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


def set_switch(ctxs, name, value):
    """libd2g reads its D2G_* switches once per context: change one for an A/B leg and make the context(s) read it"""
    if value is None:
        os.environ.pop(name, None)
    else:
        os.environ[name] = value
    for c in (ctxs if isinstance(ctxs, (list, tuple)) else [ctxs]):
        c.reload_tuning()


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
    last = {}

    def run(r0, r1):
        n = sum(N - r - 1 for r in range(r0, r1))
        out = np.empty(n, np.float32)
        t0 = time.perf_counter()
        lib.d2o_allpairs_ut_rows(P(sig_np, C.c_double), P(cards_np, C.c_double), N, S, 0, 31, r0, r1,
                                 P(out, C.c_float), ncores, bs)
        last["out"] = out                                 # kept: the checker of the timed step's output (run_single)
        return n, time.perf_counter() - t0

    n, dt = run(0, min(N, 2 * ncores))                    # probe
    rate = n / max(dt, 1e-9)
    rows = int(min(N, max(2 * ncores, seconds * rate / max(N - 1, 1))))
    rows = max(ncores, rows // ncores * ncores)
    n, dt = run(0, min(rows, N))
    return {"value": n / dt, "unit": "pairs/s", "cores": ncores, "kind": "port",
            "sample": f"rows [0,{min(rows, N)}) of the same {N}x{S} matrix = {n} pairs in {dt:.2f}s; "
                      f"oracle OpenMP restatement of emit_rectangular+compare, batch={bs}, {march}; cpu='{cpu_model()}'"}, last["out"], min(rows, N)


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


def cross_family_stats(regs, fam):
    """What decides between the sparse path's regimes (VERDICT r5 #1c, reference src/cmp_core.cpp:461-465: only the equality count matters): the register
    values sketches of DIFFERENT families share, counted on the host from the u64 registers.  -> entries (= list entries an ideal family partition leaves:
    one per cross-family pair and shared register; per sketch: 2 x entries / N), the registers a sketch shares with at least one stranger (a family value one
    stranger joins counts for every member), entries per cross-family pair."""
    N, S = regs.shape
    entries = 0
    shared_regs = np.zeros(N, np.int64)
    for t in range(S):
        v = regs[:, t]
        order = np.lexsort((fam, v))
        vs, fs = v[order], fam[order]
        new_v = np.empty(N, bool); new_v[0] = True; np.not_equal(vs[1:], vs[:-1], out=new_v[1:])
        new_f = new_v.copy(); new_f[1:] |= fs[1:] != fs[:-1]
        vid = np.cumsum(new_v) - 1
        fid = np.cumsum(new_f) - 1
        h = np.bincount(vid)                      # holders per value
        hf = np.bincount(fid)                     # holders per (value, family)
        strangers = h[vid] - hf[fid]              # per holder: holders of its value in other families
        shared_regs[order] += strangers > 0
        entries += int(strangers.sum()) // 2
    same = np.bincount(fam.astype(np.int64)).astype(np.float64)
    cross_pairs_all = (N * (N - 1) - float((same * (same - 1)).sum())) / 2.0
    return {"cross_family_entries": entries, "cross_family_entries_per_sketch": 2.0 * entries / N, "registers_shared_with_a_stranger_per_sketch_mean": float(shared_regs.mean()),
            "registers_shared_with_a_stranger_per_sketch_max": int(shared_regs.max()), "cross_family_pairs": cross_pairs_all,
            "entries_per_cross_family_pair": entries / max(cross_pairs_all, 1.0)}


def k1_built_collection(D, synth, ctx, S, ncores, nfam, per_fam, L, k=31, **kw):
    """VERDICT r5 #1c: a collection whose sketches come from GENOMES through the product's own K1 (not planted registers): `nfam` families of `per_fam`
    mutated copies (synth.family_collection), optionally with conserved segments shared across families; FASTA -> d2g_seqpack -> k1_oph_kernel ->
    d2g_oph_finalize.  -> (signatures, cardinalities, registers, family of every sketch)"""
    regs, fam = [], []
    batch, sp, nb = 500, None, 0
    def flush():
        nonlocal sp, nb
        if sp is not None and nb:
            regs.append(ctx.oph_sketch_seqpack(sp, S))
            sp.close()
        sp, nb = None, 0
    for f, name, g in synth.family_collection(nfam, per_fam, L, **kw):
        if sp is None:
            sp = D.SeqPack(k)
        sp.add_fastx(synth.fasta_bytes_fast(name, g))
        fam.append(f)
        nb += 1
        if nb == batch:
            flush()
    flush()
    regs = np.concatenate(regs)
    sig, cards = D.oph_finalize(regs, S, nthreads=ncores)
    return sig, cards, regs, np.asarray(fam, np.int32)


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


# ======================================================================================================================
# N > 1: supervisor + watchdog ladder
# ======================================================================================================================
LADDER = ("cabi", "inproc", "torch", "broadcast")
VISIBILITY_VARS = ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "GPU_DEVICE_ORDINAL")
# seconds a rung may take before its process group is killed; the first rung also pays the first `import torch` of N processes on a
# fresh box (1-2 min).  D2G_BENCH_RUNG_TIMEOUT overrides all of them (tests).
RUNG_TIMEOUT = {"cabi": 420.0, "inproc": 300.0, "torch": 300.0, "broadcast": 240.0}
ENGINE_NAME = {"cabi": "libd2g (d2g_allpairs over d2g_comm: RCCL send/recv groups), one process per GPU",
               "inproc": "libd2g (d2g_comm_create_all + d2g_allpairs_step_all), ONE process driving every GPU",
               "torch": "torch.distributed (dashing2_amd.dist.RowShardedAllPairs), one process per GPU",
               "broadcast": "whole-matrix torch.distributed broadcast per step + single-GPU prepare on every rank"}
# profiles/r06_mgpu_model.txt (tools/mgpu_model.sh): ONE rank's step of BASELINE config 4 replayed from loopback kernel durations +
# every exchange at (bytes over the busiest link) / 50 GB/s + 6 us per enqueued operation.  ms per phase INSTANCE (a chunked phase
# runs `chunks` times; fill = the slab pre-filled at the start of the step, under the first exchange; order = the sparse path on the gathered
# operand -- ids from the planes, families, sort, pair list, sorted stream: REPLICATED on every rank --; pair = launch rows + listed tiles + pair list);
# step_ms = the replayed one-job step; speedup vs the model's own 1-rank engine step (3.22 ms; the plain single-GPU path: 2.7 ms);
# floor_ms (W = 8) = what no schedule of this design goes below (first exchange + one chunk's prepare + the plane exchange + order + pair).
MODEL_R06 = {
    2: {"chunks": 4, "pack": 0.088, "fill": 0.441, "x1": 0.512, "prepare": 0.185, "x2": 0.257, "derive": 0.015, "order": 0.450, "pair": 0.320, "step_ms": 4.109, "speedup": 0.78},
    4: {"chunks": 4, "pack": 0.048, "fill": 0.219, "x1": 0.128, "prepare": 0.101, "x2": 0.129, "derive": 0.015, "order": 0.454, "pair": 0.207, "step_ms": 1.911, "speedup": 1.69},
    8: {"chunks": 2, "pack": 0.025, "fill": 0.108, "x1": 0.064, "prepare": 0.101, "x2": 0.129, "derive": 0.023, "order": 0.451, "pair": 0.152, "step_ms": 1.270, "speedup": 2.54, "floor_ms": 1.072},
}
# the same replay at larger matrices (profiles/r06_mgpu_model_N<N>.txt; W = 8 against the model's own 1-rank engine step): the replicated
# order phase and the exchanges grow with N, the pair work with N^2, so the ratio rises with N -- and the REPLAYED step still does not
# reach 6x at N = 200000 (4.85x); only the design's floor does (6.45x), i.e. with every exchange after the first hidden.
MODEL_R06_AT_N = {
    50000:  {"W1_engine_step_ms": 3.223,  "W8_step_ms": 1.270, "W8_speedup": 2.54, "W8_floor_ms": 1.072, "W8_floor_speedup": 3.01},
    100000: {"W1_engine_step_ms": 10.909, "W8_step_ms": 2.869, "W8_speedup": 3.80, "W8_floor_ms": 2.381, "W8_floor_speedup": 4.58},
    200000: {"W1_engine_step_ms": 37.697, "W8_step_ms": 7.767, "W8_speedup": 4.85, "W8_floor_ms": 5.846, "W8_floor_speedup": 6.45},
}
MODEL_REACHES_6X_AT_N = None            # not at any N modelled (<= 200000: 160 GB of float32 output over 8 GPUs); the floor alone: at 200000


def multi_shape(args):
    """(N, S, scaling, workload) of an N > 1 run -- the same in the supervisor and in every worker"""
    S = args.sketchsize
    W = args.gpus
    scaling = args.scaling or "strong"
    if scaling == "strong":
        N = args.sketches or 50000
        workload = "BASELINE config 4" if (N, S) == (50000, 1024) else "custom"
    else:
        N = int(round((args.sketches or 10000) * math.sqrt(W)))
        workload = "BASELINE config 3 x sqrt(n_gpus) sketches (constant pairs per GPU)"
    N = (N + W - 1) // W * W                                   # equal row blocks per rank (the torch rung needs it; config 4 is unaffected)
    return N, S, scaling, workload


def error_line(args, msg, extra=None):
    line = {"metric": "all-pairs sketch comparison throughput (pairs/s)", "value": None, "unit": "pairs/s", "n_gpus": args.gpus,
            "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "error": msg}
    if extra:
        line.update(extra)
    return line


def last_json_line(path):
    """the last line of a worker's stdout that parses as a bench line (RCCL prints banners through C stdio)"""
    try:
        txt = open(path, "rb").read().decode("utf-8", "replace")
    except OSError:
        return None
    for ln in reversed(txt.splitlines()):
        ln = ln.strip()
        if ln.startswith("{") and '"metric"' in ln:
            try:
                return json.loads(ln)
            except ValueError:
                continue
    return None


def run_rung(engine, args, tmp, extra_argv, timeout):
    """one rung: spawn its measuring process(es) in their own session, wait under a watchdog, kill what is left.
    -> (line or None, outcome string, seconds)"""
    import signal
    import socket
    import subprocess
    nproc = 1 if engine == "inproc" else args.gpus
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    base_env = {k: v for k, v in os.environ.items()
                if not (k.startswith("TORCHELASTIC_") or k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE",
                                                                "ROLE_RANK", "ROLE_WORLD_SIZE", "ROLE_NAME", "MASTER_ADDR", "MASTER_PORT", "OMP_NUM_THREADS",
                                                                "TORCH_NCCL_ASYNC_ERROR_HANDLING", "NCCL_ASYNC_ERROR_HANDLING"))}
    base_env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if os.path.isdir("/sys/class/net/lo"):
        base_env.setdefault("GLOO_SOCKET_IFNAME", "lo")            # every rank is on this node: gloo must not depend on the hostname resolving
    base_env["OMP_NUM_THREADS"] = str(max(1, host_cores() // nproc))
    procs, files = [], []
    t0 = time.monotonic()
    for r in range(nproc):
        env = dict(base_env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nproc), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        fo = open(os.path.join(tmp, f"{engine}.{r}.out"), "wb")
        fe = open(os.path.join(tmp, f"{engine}.{r}.err"), "wb")
        files += [fo, fe]
        cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--worker", "--engine", engine] + extra_argv
        procs.append(subprocess.Popen(cmd, env=env, stdout=fo, stderr=fe, start_new_session=True))
    outcome = None
    while outcome is None:
        rcs = [p.poll() for p in procs]
        bad = [(i, rc) for i, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            outcome = "rank %d exited with code %d" % bad[0]
        elif all(rc == 0 for rc in rcs):
            outcome = "ok"
        elif time.monotonic() - t0 > timeout:
            outcome = "timeout after %.0f s (hung: ranks %s)" % (timeout, [i for i, rc in enumerate(rcs) if rc is None])
        else:
            time.sleep(0.1)
    for p in procs:                                             # whatever is still alive: the whole session of that worker
        if p.poll() is None:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except (ProcessLookupError, PermissionError):
                pass
    for p in procs:
        try:
            p.wait(timeout=30)
        except Exception:                                       # noqa: BLE001
            pass
    for f in files:
        f.close()
    line = last_json_line(os.path.join(tmp, f"{engine}.0.out"))
    if outcome != "ok":
        tails = []
        for r in range(nproc):
            try:
                t = open(os.path.join(tmp, f"{engine}.{r}.err"), "rb").read().decode("utf-8", "replace").strip().splitlines()
                if t:
                    tails.append(f"rank {r}: " + " | ".join(t[-3:])[-400:])
            except OSError:
                pass
        if tails:
            outcome += " :: " + " ;; ".join(tails[:3])
    for r in range(nproc):                                      # the workers' stderr is ours (RCCL warnings, tracebacks)
        try:
            sys.stderr.write(open(os.path.join(tmp, f"{engine}.{r}.err"), "rb").read().decode("utf-8", "replace")[-4000:])
        except OSError:
            pass
    return line, outcome, time.monotonic() - t0, [p.pid for p in procs]


def supervise(args):
    """N > 1: never measures anything itself.  Returns the exit code; prints exactly one JSON line."""
    import shutil
    import tempfile
    launched_by = "bench.py (no launcher)"
    if "WORLD_SIZE" in os.environ:
        launched_by = "torch.distributed.run ranks (rank 0 supervises, the others exit 0)"
        if int(os.environ["WORLD_SIZE"]) != args.gpus:
            print(json.dumps(error_line(args, f"--gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}")), flush=True)
            return 2
        if int(os.environ.get("RANK", "0")) != 0:
            return 0
    import dashing2_amd as D
    from dashing2_amd import synth
    skip_check = args.loopback or os.environ.get("D2G_BENCH_TEST_SKIP_DEVICE_CHECK") == "1"
    have = int(D.lib().d2g_device_count())
    if have < args.gpus and not skip_check and any(v in os.environ for v in VISIBILITY_VARS):
        # a launcher that pins each rank to its own GPU (HIP_VISIBLE_DEVICES = LOCAL_RANK and the like) hides the others from THIS rank, which
        # supervises for all of them: count again without the mask, in a throw-away process, and start the workers without it
        import subprocess
        env = {k: v for k, v in os.environ.items() if k not in VISIBILITY_VARS}
        code = "import ctypes,sys; print(ctypes.CDLL(sys.argv[1]).d2g_device_count())"
        try:
            unmasked = int(subprocess.run([sys.executable, "-c", code, D.LIB_PATH], env=env, capture_output=True, text=True, timeout=120).stdout.strip() or 0)
        except Exception:                                       # noqa: BLE001
            unmasked = 0
        if unmasked >= args.gpus:
            for v in VISIBILITY_VARS:
                os.environ.pop(v, None)
            have = unmasked
            launched_by += "; the launcher's device mask was dropped for the workers"
    if have < args.gpus and not skip_check:
        print(json.dumps(error_line(args, f"--gpus {args.gpus} requested but {have} HIP device(s) visible: refusing to "
                                          "measure a smaller job under that label")), flush=True)
        return 2
    engines = [e for e in (args.engines.split(",") if args.engines else (("inproc",) if args.loopback else LADDER)) if e]
    bad = [e for e in engines if e not in LADDER]
    if bad or (args.loopback and engines != ["inproc"]):
        print(json.dumps(error_line(args, f"unknown / unusable rung(s) {bad or engines} (--loopback runs the `inproc` rung only)")), flush=True)
        return 2
    N, S, _, _ = multi_shape(args)
    # scratch: the synthetic matrix, written ONCE for every rung (and the workers' stdout / stderr)
    where = None
    for cand in ("/dev/shm", tempfile.gettempdir()):
        try:
            st = os.statvfs(cand)
            if st.f_bavail * st.f_frsize > N * S * 8 + (64 << 20):
                where = cand
                break
        except OSError:
            continue
    tmp = tempfile.mkdtemp(prefix="d2g_bench_", dir=where)
    ladder, rc, out_line = [], 2, None
    try:
        t0 = time.monotonic()
        regs = synth.synthetic_registers(N, S, nclusters=max(8, N // 150), seed=20260929 if N == 50000 else 20260928)
        sig, cards = D.oph_finalize(regs, S, nthreads=host_cores())
        del regs
        sig_file, cards_file = os.path.join(tmp, "sig.npy"), os.path.join(tmp, "cards.npy")
        np.save(sig_file, sig.view(np.uint64))
        np.save(cards_file, cards)
        del sig
        gen_s = time.monotonic() - t0
        forced = os.environ.get("D2G_BENCH_RUNG_TIMEOUT")
        for engine in engines:
            timeout = float(forced) if forced else RUNG_TIMEOUT[engine]
            line, outcome, secs, pids = run_rung(engine, args, tmp, ["--sig-file", sig_file, "--cards-file", cards_file], timeout)
            usable = isinstance(line, dict) and line.get("value") is not None and line.get("n_gpus") == args.gpus
            ladder.append({"engine": engine, "outcome": outcome, "seconds": round(secs, 1), "line": bool(usable), "pids": pids})
            if usable:
                # a rung that was killed AFTER its headline line was out (a secondary leg hung) still delivered the measurement
                out_line, rc = line, 0
                break
        if out_line is None:
            out_line = error_line(args, "every rung of the multi-GPU ladder failed: " + " || ".join(f"{l['engine']}: {l['outcome']}" for l in ladder)[:3000])
        out_line["launcher"] = {"launched_by": launched_by, "ladder": ladder, "input_generation_s": round(gen_s, 1), "scratch": where,
                                "note": "the measuring processes are children of a supervisor that touches no GPU; each rung runs under a timeout "
                                        "and its whole process group is killed on a timeout or a non-zero exit"}
    except Exception as e:                                      # noqa: BLE001 - the supervisor itself must not lose the line
        out_line = error_line(args, f"supervisor: {type(e).__name__}: {e}", {"launcher": {"launched_by": launched_by, "ladder": ladder}})
        rc = 2
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print(json.dumps(out_line), flush=True)
    return rc


# ======================================================================================================================
# N > 1: one measuring process (a rank of `cabi` / `torch` / `broadcast`, or the single `inproc` process)
# ======================================================================================================================
def run_multi(args):
    engine = args.engine
    hooks = os.environ.get("D2G_BENCH_TEST_HANG", "").split(",")
    if engine in hooks:                                          # test hook: a rung that never comes back (before any import)
        while True:
            time.sleep(3600)
    if engine in os.environ.get("D2G_BENCH_TEST_FAIL", "").split(","):
        raise SystemExit(3)
    if engine in os.environ.get("D2G_BENCH_TEST_LINE_THEN_HANG", "").split(","):      # test hook: the headline is out, a secondary leg hangs
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps({"metric": "all-pairs sketch comparison throughput (pairs/s)", "value": 1.0, "unit": "pairs/s", "n_gpus": args.gpus,
                              "config": {"exchange_engine": "test hook"}}), flush=True)
        while True:
            time.sleep(3600)
    import ctypes
    import torch
    import torch.distributed as dist
    import dashing2_amd as D

    W = args.gpus
    ranked = engine != "inproc"
    N, S, scaling, workload = multi_shape(args)
    pairs_total = N * (N - 1) // 2
    bounds = D.ut_partition(N, W)
    algo = {"auto": D.CMP_AUTO, "direct": D.CMP_DIRECT, "bitslice": D.CMP_BITSLICE}[args.algo]
    ncores = host_cores()
    if ranked:
        rank0 = int(os.environ["RANK"])
        local_dev = int(os.environ.get("LOCAL_RANK", rank0))
        torch.cuda.set_device(local_dev)
        backend = "gloo" if engine == "cabi" else "nccl"
        kw = {"device_id": torch.device("cuda", local_dev)} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank0, world_size=W, **kw)
        ranks = [rank0]
        ctrl = torch.device("cpu") if backend == "gloo" else torch.device("cuda", local_dev)
    else:
        rank0, ranks, ctrl = 0, list(range(W)), torch.device("cpu")
    is_root = rank0 == 0

    def dev_of(r):
        return 0 if args.loopback else (r if not ranked else local_dev)

    # ---- control-plane reductions (gloo / CPU for the libd2g engines: nothing but the engine touches RCCL)
    def allreduce(x, op):
        if not ranked:
            return x
        t = torch.tensor([float(x)], dtype=torch.float64, device=ctrl)
        dist.all_reduce(t, op={"max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}[op])
        return float(t.item())

    def gather_obj(o):
        """per-rank python objects -> list indexed by rank (on every rank)"""
        if not ranked:
            return o
        res = [None] * W
        dist.all_gather_object(res, o[0])
        return res

    def sync_all():
        for r in ranks:
            torch.cuda.synchronize(dev_of(r))

    def barrier():
        sync_all()
        if ranked:
            dist.barrier()
        sync_all()

    # ---- input: the supervisor's scratch file (memory-mapped); rank r uploads the rows it HOLDS
    sig_np = np.load(args.sig_file, mmap_mode="r")
    assert sig_np.shape == (N, S) and sig_np.dtype == np.uint64
    lut_np = D.epilogue_lut(S, D.SIMILARITY, 31)
    L = {}                                                       # per local rank: everything that lives on its GPU
    for r in ranks:
        d = dev_of(r)
        torch.cuda.set_device(d)
        ctx = D.Context(d)
        tdev = torch.device("cuda", d)
        L[r] = {"ctx": ctx, "dev": tdev, "r0": bounds[r], "r1": bounds[r + 1], "pairs": D.ut_count(N, bounds[r], bounds[r + 1]),
                "lut": torch.from_numpy(lut_np).to(tdev)}
        L[r]["out"] = torch.empty(max(L[r]["pairs"], 1), dtype=torch.float32, device=tdev)
    stream = None                                                # the NULL stream of each engine's device
    comms = engs = None
    eng_of = {}
    if engine == "cabi":
        uid = [D.comm_unique_id() if is_root else None]
        dist.broadcast_object_list(uid, src=0)
        r = ranks[0]
        comm = D.Comm.create(L[r]["ctx"], r, W, uid[0])         # ncclCommInitRank: collective
        ctypes.CDLL(None).fflush(None)                          # RCCL's version banner (C stdio): out now, not after the JSON line
        eng_of[r] = D.AllPairs(L[r]["ctx"], comm, N, S)         # exchanges (N, S, world, chunks) with every rank: the first traffic
        comms = [comm]
    elif engine == "inproc":
        if args.loopback:
            os.environ["D2G_COMM_LOOPBACK"] = "1"
        comms = D.Comm.create_all([L[r]["ctx"] for r in ranks])  # ncclCommInitAll (or the loopback transport)
        ctypes.CDLL(None).fflush(None)
        for r in ranks:
            eng_of[r] = D.AllPairs(L[r]["ctx"], comms[r], N, S)
    for r in ranks:
        if r in eng_of:
            assert eng_of[r].rows_computed == (L[r]["r0"], L[r]["r1"])
            lo, hi = eng_of[r].rows_held
        else:
            lo, hi = r * (N // W), (r + 1) * (N // W)
        torch.cuda.set_device(dev_of(r))
        L[r]["rows"] = torch.from_numpy(np.ascontiguousarray(sig_np[lo:hi]).view(np.int64)).to(L[r]["dev"])
    teng = None
    if engine == "torch":
        from dashing2_amd import dist as DD
        r = ranks[0]
        teng = DD.RowShardedAllPairs(L[r]["ctx"], N, S, L[r]["dev"])
        assert (teng.r0, teng.r1) == (L[r]["r0"], L[r]["r1"])
    full_dev = {}
    if engine == "broadcast":
        r = ranks[0]
        full_dev[r] = (torch.from_numpy(np.ascontiguousarray(sig_np).view(np.int64)).to(L[r]["dev"]) if is_root
                       else torch.empty((N, S), dtype=torch.int64, device=L[r]["dev"]))
        dist.broadcast(full_dev[r], 0)
        L[r]["cs"] = L[r]["ctx"].cmp_set_dev(full_dev[r].data_ptr(), N, S, algo=algo, stream=stream)

    # ---- ONE job's step, enqueued for every local rank
    if engine == "cabi":
        r = ranks[0]

        def step():
            eng_of[r].step_lut_dev(L[r]["rows"].data_ptr(), L[r]["lut"].data_ptr(), L[r]["out"].data_ptr(), stream)
    elif engine == "inproc":
        e_list = [eng_of[r] for r in ranks]
        p_rows, p_lut, p_out = ([L[r][k].data_ptr() for r in ranks] for k in ("rows", "lut", "out"))

        def step():
            D.allpairs_step_all(e_list, p_rows, p_lut, p_out, None)
    elif engine == "torch":
        r = ranks[0]
        tstream = torch.cuda.current_stream().cuda_stream

        def step():
            teng.step_lut(L[r]["rows"], L[r]["lut"], L[r]["out"], tstream)
    else:
        r = ranks[0]
        tstream = torch.cuda.current_stream().cuda_stream

        def step():
            dist.broadcast(full_dev[r], 0)                       # the path's one exchange (RCCL over xGMI)
            L[r]["cs"].update_dev(full_dev[r].data_ptr(), tstream)
            L[r]["cs"].lut_ut_dev(L[r]["lut"].data_ptr(), L[r]["out"].data_ptr(), L[r]["r0"], L[r]["r1"], tstream)

    step()                                                       # first pass: allocations, code objects, the first exchange
    barrier()
    for r in ranks:
        if r in eng_of:
            eng_of[r].status(stream)                             # a rank-table overflow on ANY rank invalidates the step everywhere

    # ---- every rank's WHOLE slab against a single-GPU computation over the whole matrix (the exchange has never run on
    # hardware before the driver's scaling run); rank 0 times that single-GPU job: the base of this line's speedup
    base = None
    ok = True
    for r in ranks:
        d = L[r]
        torch.cuda.set_device(dev_of(r))
        full = full_dev.get(r)
        if full is None:
            full = torch.from_numpy(np.ascontiguousarray(sig_np).view(np.int64)).to(d["dev"])
        ref = d["ctx"].cmp_set_dev(full.data_ptr(), N, S, algo=algo, stream=stream)
        want = torch.empty(pairs_total, dtype=torch.float32, device=d["dev"])
        ref.lut_ut_dev(d["lut"].data_ptr(), want.data_ptr(), 0, N, stream)
        torch.cuda.synchronize(dev_of(r))
        o0 = D.ut_count(N, 0, d["r0"])
        ok = ok and bool(torch.equal(want[o0:o0 + d["pairs"]].view(torch.int32), d["out"][:d["pairs"]].view(torch.int32)))
        if r == 0:
            reps = 3

            def time_ref(cset):
                cset.update_dev(full.data_ptr(), stream)
                cset.lut_ut_dev(d["lut"].data_ptr(), want.data_ptr(), 0, N, stream)
                torch.cuda.synchronize(dev_of(r))
                t0 = time.perf_counter()
                for _ in range(reps):
                    cset.update_dev(full.data_ptr(), stream)
                    cset.lut_ut_dev(d["lut"].data_ptr(), want.data_ptr(), 0, N, stream)
                torch.cuda.synchronize(dev_of(r))
                return (time.perf_counter() - t0) / reps

            bdt = time_ref(ref)
            sp = ref.sparse_info(stream)
            # the single-GPU path held to the dense walk (D2G_BS_SPARSE=0: every tile) beside the default: the sharded engine's pair phase runs
            # whichever of the two the engine's gathered operand took (C-ABI engine: sparse tiles from N >= 8192, like a single GPU; the torch
            # and broadcast rungs: see engine_sparse below), and `speedup` is read against the base that runs the SAME algorithm
            set_switch(d["ctx"], "D2G_BS_SPARSE", "0")
            try:
                dense = d["ctx"].cmp_set_dev(full.data_ptr(), N, S, algo=algo, stream=stream)
                ddt = time_ref(dense)
                dense.close()
            finally:
                set_switch(d["ctx"], "D2G_BS_SPARSE", None)
            if eng_of:
                esp = eng_of[0].sparse_info()
            elif teng is not None:
                esp = teng.full.sparse_info(stream)
            else:
                esp = L[0]["cs"].sparse_info(stream)
            same = bdt if esp["sorted_operand"] else ddt
            base = {"base_1gpu_same_config_pairs_per_s": pairs_total / same, "base_1gpu_ms_per_step": same * 1e3,
                    "base_is": "best_1gpu (sparse tiles on both sides)" if esp["sorted_operand"] else "dense_walk_1gpu (every tile on both sides)",
                    "dense_walk_1gpu_pairs_per_s": pairs_total / ddt, "dense_walk_1gpu_ms_per_step": ddt * 1e3,
                    "best_1gpu_pairs_per_s": pairs_total / bdt, "best_1gpu_ms_per_step": bdt * 1e3, "best_1gpu_sparse": sp, "engine_sparse": esp,
                    "note": f"the SAME {N} x {S} job (prepare + compare over the whole triangle, sketches resident in HBM) on GPU 0 alone, mean of {reps} steps, "
                            "timed in this run before the sharded steps: best_1gpu = the single-GPU default (sparse tiles from N >= 8192: only tiles with a "
                            "shared register value are walked), dense_walk_1gpu = the same with D2G_BS_SPARSE=0 (every tile).  base_1gpu_same_config is the one "
                            "whose pair phase runs the algorithm rank 0's pair phase ran (engine_sparse); `speedup` is value / that, `speedup_vs_best_1gpu` "
                            "value / best_1gpu"}
        ref.close()
        del ref, want
        if r not in full_dev:
            del full
        torch.cuda.empty_cache()
    slab_check = allreduce(1.0 if ok else 0.0, "min") == 1.0

    # ---- per-phase times of ONE untimed step (libd2g engines), every rank
    phases = None
    if eng_of:
        for r in ranks:
            eng_of[r].set_phase_timing(True)
        barrier()
        step()
        sync_all()
        mine = [[[p["phase"], p["chunk"], round(p["start_ms"], 4), round(p["ms"], 4)] for p in eng_of[r].phase_times()] for r in ranks]
        for r in ranks:
            eng_of[r].set_phase_timing(False)
        allp = gather_obj(mine)
        worst = {}
        for rec in allp:
            for ph, c, st, ms in rec:
                worst.setdefault(ph, {}).setdefault(c, 0.0)
                worst[ph][c] = max(worst[ph][c], ms)
        phases = {"per_rank": allp, "fields": ["phase", "chunk", "start_ms (after the step's first enqueue reached the GPU)", "ms"],
                  "max_over_ranks_ms": {ph: [worst[ph][c] for c in sorted(worst[ph])] for ph in worst},
                  "note": "one untimed step with timing events around every phase on the stream it runs on (compute stream: pack, prepare, derive, order, pair; "
                          "exchange stream: x1 = rows -> column slices, x2 = bit-plane groups to everyone); an exchange's time includes waiting for "
                          "the slowest peer" + ("; ONE process enqueues all ranks here, so start_ms also carries the enqueue order" if not ranked else "")}

    # ---- the timed region
    def timed_run(fn):
        for r in ranks:
            L[r]["ctx"].set_timing(D.TIME_K2PREP)
            L[r]["ctx"].kernel_ms("k2prep")
        for _ in range(max(args.warmup, 1)):
            fn()
        barrier()
        for r in ranks:
            L[r]["ctx"].set_timing(D.TIME_K2)
            L[r]["ctx"].kernel_ms("k2")
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        barrier()
        dt = time.perf_counter() - t0
        for r in ranks:
            L[r]["ctx"].set_timing(False)
        return allreduce(dt, "max")

    dt = timed_run(step)
    per_rank = gather_obj([{"rank": r, "pair_kernel_ms": L[r]["ctx"].kernel_ms("k2")[1], "prepare_chain_ms": L[r]["ctx"].kernel_ms("k2prep")[1],
                            "pairs": L[r]["pairs"]} for r in ranks])
    for r in ranks:
        if r in eng_of:
            eng_of[r].status(stream)
    ms_per_step = dt / args.steps * 1e3
    value = pairs_total / (dt / args.steps)

    cs0 = None
    if is_root:
        cs0 = eng_of[0].operand() if eng_of else (teng.full if teng is not None else L[0]["cs"])
        max_distinct, nbits, mean_nbits = cs0.planes(stream)
        algo_used = cs0.algo
        k2_ms, prep_ms, my_pairs = per_rank[0]["pair_kernel_ms"], per_rank[0]["prepare_chain_ms"], per_rank[0]["pairs"]
        alg_bytes = 8 * S * N + 4 * my_pairs
        achieved = alg_bytes / (k2_ms * 1e-3) / 1e9 if k2_ms > 0 else 0.0
        kname = "k2_bitslice_kernel" if algo_used == D.CMP_BITSLICE else "k2_direct_kernel"
        if base is not None and base["engine_sparse"]["sorted_operand"] and not base["engine_sparse"]["dense_kernel_ran"]:
            kname = "k2 sparse chain (rows + gather + list + fill + k2_bitslice_sparse_kernel over %d listed tiles of rank 0's rows)" % base["engine_sparse"]["tiles_listed"]
        roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": None, "traffic_source": "not measured at N > 1 (the N = 1 line measures the same kernel's traffic in its own run)",
                    "kernel": kname, "kernel_ms": k2_ms, "algorithmic_bytes": alg_bytes, "prep_ms": prep_ms,
                    "note": "rank 0's launch: 8*S*N bytes of sketches + 4 bytes per pair of ITS slab; equality counting is VALU-bound, not HBM-bound (SURVEY 8d)"}
        ops = my_pairs * ((S + 31) // 32) * (mean_nbits + getattr(D, "BITSLICE_OPS_PER_GROUP_EXTRA", 2)) if algo_used == D.CMP_BITSLICE else my_pairs * S * 2
        va = ops / (k2_ms * 1e-3) if k2_ms > 0 else 0.0
        compute = {"bound": "valu", "unit": "lane-ops/s", "achieved": va, "peak": VALU_PEAK_LANEOPS, "frac": va / VALU_PEAK_LANEOPS,
                   "bit_planes_max": nbits, "bit_planes_mean": mean_nbits, "max_shared_values_per_column_plus1": max_distinct}
        if base is not None:
            base["speedup"] = value / base["base_1gpu_same_config_pairs_per_s"]
            base["speedup_vs_best_1gpu"] = value / base["best_1gpu_pairs_per_s"]
        model = None
        if (N, S) == (50000, 1024) and W in MODEL_R06 and eng_of:
            m = MODEL_R06[W]
            model = dict(m, source="profiles/r06_mgpu_model.txt (loopback kernel durations + exchanges at 50 GB/s per link + 6 us per enqueue)",
                         note="ms per phase instance; compare with phases.max_over_ranks_ms term by term",
                         reaches_6x_at_N=MODEL_REACHES_6X_AT_N,
                         at_N={str(k): v for k, v in MODEL_R06_AT_N.items()},
                         at_N_note="W = 8 replayed at larger matrices (profiles/r06_mgpu_model_N<N>.txt): 2.54x / 3.80x / 4.85x of the 1-rank engine step at "
                                   "N = 50000 / 100000 / 200000; the design's floor (every exchange but the first hidden) 3.01x / 4.58x / 6.45x -- "
                                   "the replicated order phase (0.45 / 1.05 / 2.75 ms) does not shrink with W")
        line = {
            "metric": "all-pairs sketch comparison throughput (pairs/s)", "value": value, "unit": "pairs/s",
            "n_gpus": W, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{workload}: {N} pre-built OPH sketches, S={S}, all-pairs cmp only, {pairs_total} pairs, float32 Jaccard",
                       "sketches": N, "sketchsize": S, "pairs": pairs_total, "algo": "bitslice" if algo_used == D.CMP_BITSLICE else "direct",
                       "step": ("all-to-all rows->column slices + per-rank prepare of S/W columns + all-gather of bit planes + (N >= 8192) ordering of the gathered operand and tile list on every rank + pair kernel w/ fused epilogue over this rank's rows; "
                                "row-sharded sketches resident in HBM" if engine != "broadcast" else
                                "RCCL broadcast of the whole matrix + prepare + pair kernel w/ fused epilogue on every rank"),
                       "parallelism": f"upper-triangle rows sharded over {W} GPU(s) by pair count",
                       "exchange_engine": ENGINE_NAME[engine],
                       **({"transport": "loopback (all contexts on ONE device: a functional run, not a scaling measurement)"} if args.loopback else {}),
                       **({"exchange_chunks": eng_of[0].chunks} if eng_of else {}),
                       "slab_check": ("every rank's WHOLE slab equals a single-GPU computation over the whole matrix" if slab_check else
                                      "MISMATCH between the sharded step and a single-GPU computation: this line is NOT a valid measurement")},
            "scaling_base": base, "phases": phases, "model": model, "per_rank": per_rank,
            "roofline": roofline, "compute": compute, "cpu_baseline": None,
        }
        if not slab_check:
            line["valid"] = False
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(line), flush=True)                      # the headline is OUT before any secondary leg can hang

    # ---- secondary (cabi): the software-pipelined rate for a stream of matrices
    if engine == "cabi" and not args.no_pipeline:
        r = ranks[0]
        want = L[r]["out"].clone()

        def stream_step():
            eng_of[r].enqueue_lut_dev(L[r]["rows"].data_ptr(), L[r]["lut"].data_ptr(), L[r]["out"].data_ptr(), stream, input_ready=True)
        stream_step(); stream_step()
        sync_all()
        sdt = timed_run(stream_step)
        same = allreduce(1.0 if torch.equal(want, L[r]["out"]) else 0.0, "min") == 1.0
        if is_root:
            line["stream_of_matrices"] = {"value": pairs_total / (sdt / args.steps), "unit": "pairs/s", "ms_per_step": sdt / args.steps * 1e3,
                                          "outputs_identical_to_the_one_job_step": same,
                                          "note": "NOT the headline: throughput over repeated matrices with the exchange + prepare of step i+1 hidden "
                                                  "under the pair kernel of step i (d2g_allpairs_enqueue_lut_dev); BASELINE config 4 is one job"}
            print(json.dumps(line), flush=True)
        del want

    # ---- secondary (--all-legs): K1 sketch construction sharded by input, no collectives
    if args.all_legs and not args.no_sketch:
        from dashing2_amd import synth
        n_all, Lg, k = args.sketch_genomes, args.sketch_len, 31
        n_g = max(1, n_all // W)
        secs, bases, k1 = 0.0, 0, []
        for r in ranks:
            d = L[r]
            torch.cuda.set_device(dev_of(r))
            packed_np, run_start, run_len, goff, _ = pack_genomes(D, synth, r * n_g, n_g, Lg, k)
            packed = torch.from_numpy(packed_np).to(d["dev"])
            plan = d["ctx"].oph_plan(run_start, run_len, goff, k)
            regs_dev = torch.empty((n_g, D.oph_m(S)), dtype=torch.int64, device=d["dev"])
            d["k1"] = (plan, packed, regs_dev)
            bases += int(plan.nbases)
        for rep in range(2):
            barrier()
            for r in ranks:
                L[r]["ctx"].set_timing(D.TIME_K1)
                L[r]["ctx"].kernel_ms("k1")
            t0 = time.perf_counter()
            for _ in range(5):
                for r in ranks:
                    plan, packed, regs_dev = L[r]["k1"]
                    L[r]["ctx"].oph_sketch_dev(plan, packed.data_ptr(), S, regs_dev.data_ptr(), stream=stream)
            barrier()
            secs = allreduce((time.perf_counter() - t0) / 5, "max")
        k1 = gather_obj([L[r]["ctx"].kernel_ms("k1")[1] for r in ranks])
        bases_all = bases * (W if ranked else 1)
        if is_root:
            line["sketch"] = {"metric": "sketch input bases/s (K1 kernel, packed bases resident in HBM), inputs sharded one block per GPU, no collectives",
                              "value": bases_all / secs, "unit": "bases/s", "ms_per_step": secs * 1e3, "k1_kernel_ms_per_rank": k1,
                              "config": {"workload": f"BASELINE config 2 shape: {n_g * W} synthetic random genomes x {Lg} bp, k=31, S={S}, OPH, canonical; {n_g} genomes per GPU"}}
            print(json.dumps(line), flush=True)

    for r in ranks:
        if r in eng_of:
            eng_of[r].close()
    if teng is not None:
        teng.close()
    for c in comms or []:
        c.close()
    for r in ranks:
        if "cs" in L[r]:
            L[r]["cs"].close()
        L[r]["ctx"].close()
    if ranked:
        dist.destroy_process_group()


# ======================================================================================================================
# roofline.traffic measured in THIS run: a child under rocprofv3 --pmc re-runs one launch of each reported kernel
# ======================================================================================================================
def pmc_child(args):
    """`bench.py --pmc-child k2,k1,k3` (run under rocprofv3 by measure_traffic): ONE launch of each reported kernel on the shapes
    the line reports -- the config-3 prepare chain + pair kernel, the 1000-genome K1 launch, one 250-genome K3 call of either key
    path.  K1 / K3 read random 2-bit bases generated on the device (the same bytes per base as packed genomes)."""
    import torch
    import dashing2_amd as D
    from dashing2_amd import synth
    which = set(args.pmc_child.split(","))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    ctx = D.Context(0)
    S = args.sketchsize
    if "k2" in which:
        N = args.sketches or 10000
        regs = synth.synthetic_registers(N, S, nclusters=max(8, N // 150), seed=20260928)
        sig, _ = D.oph_finalize(regs, S, nthreads=host_cores())
        sig_dev = torch.from_numpy(sig.view(np.int64)).to(dev)
        lut = torch.from_numpy(D.epilogue_lut(S, D.SIMILARITY, 31)).to(dev)
        out = torch.empty(N * (N - 1) // 2, dtype=torch.float32, device=dev)
        cs = ctx.cmp_set_dev(sig_dev.data_ptr(), N, S, algo=D.CMP_AUTO, stream=None)
        cs.lut_ut_dev(lut.data_ptr(), out.data_ptr(), 0, N, None)
        torch.cuda.synchronize()
        cs.close()
        del sig_dev, out
    L = args.sketch_len
    Lb = ((L + 3) // 4 + 63) // 64 * 64

    def plan_for(n, k):
        return ctx.oph_plan(np.arange(n, dtype=np.uint64) * np.uint64(Lb * 4), np.full(n, L, np.uint32), np.arange(n + 1, dtype=np.uint64), k)

    if "k1" in which:
        n = args.sketch_genomes
        packed = torch.randint(0, 256, (n * Lb + 64,), dtype=torch.uint8, device=dev)
        regs_dev = torch.empty((n, D.oph_m(S)), dtype=torch.int64, device=dev)
        ctx.oph_sketch_dev(plan_for(n, 31), packed.data_ptr(), S, regs_dev.data_ptr(), stream=None)
        torch.cuda.synchronize()
        del packed, regs_dev
    if "k3" in which:
        nb = max(1, min(args.multiset_batch, args.multiset_genomes))
        packed = torch.randint(0, 256, (nb * Lb + 64,), dtype=torch.uint8, device=dev)
        sig3 = torch.empty((nb, 2048), dtype=torch.float64, device=dev)
        tw3 = torch.empty((nb,), dtype=torch.float64, device=dev)
        plan = plan_for(nb, 21)
        for compact in ("0", "1"):
            set_switch(ctx, "D2G_K3_COMPACT", compact)
            ctx.bmh_sketch_dev(plan, packed.data_ptr(), 2048, sig3.data_ptr(), tw3.data_ptr(), stream=None)
            torch.cuda.synchronize()
            print("PMC-CHILD k3 compact=%s done" % compact, flush=True)     # a marker per call; the CSV keeps dispatch order
    ctx.close()


K3_DEFAULT = ("k3_hist_kernel", "k3_scan_kernel", "k3_scatter_kernel", "k3_refine_kernel", "k3_split_kernel", "k3_bmh_main_kernel", "k3_bmh_survivor", "k3_bmh_verify", "k3_bmh_init")
K3_COMPACT_ONLY = ("k3c_hist", "k3c_scan", "k3c_scatter")


def re_short(name):
    import re
    n = re.sub(r"^void\s+", "", name).replace("(anonymous namespace)::", "")
    return re.split(r"[(<]", n, 1)[0][:60] or n[:60]


def measure_traffic(args, which=("k2", "k1", "k3"), timeout=300.0):
    """-> {"k2": bytes, "k2_prepare": bytes, "k1": bytes, "k3": bytes, "k3_compact": bytes, "seconds": s} measured now, or
    {"error": why}.  FETCH_SIZE and WRITE_SIZE cannot share a pass (MI355X_MICROARCH.md: TCC has 4 slots, they cost 3 + 2): two
    passes of the same child, counters only beside --kernel-trace.  rocprofv3 reports both in KiB; on gfx950 FETCH_SIZE tallies wide
    (16 B per lane) coalesced reads at half their bytes: doubled for K1 (dwordx4 loads), raw for K2 / K3 (dword / dwordx2 loads)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return {"error": "rocprofv3 not found"}
    t0 = time.monotonic()
    tmp = tempfile.mkdtemp(prefix="d2g_pmc_", dir="/tmp")
    rows = {}                                                   # counter -> [(kernel name, dispatch id, value)]
    try:
        ctrs = ["FETCH_SIZE", "WRITE_SIZE"] + (["SQ_INSTS_VALU"] if "k1" in which else [])      # one counter per pass
        for ctr in ctrs:
            od = os.path.join(tmp, ctr)
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "-f", "csv", "-d", od, "--", sys.executable, os.path.abspath(__file__),
                   "--pmc-child", ",".join(which), "--sketches", str(args.sketches or 10000), "--sketchsize", str(args.sketchsize),
                   "--sketch-genomes", str(args.sketch_genomes), "--sketch-len", str(args.sketch_len),
                   "--multiset-genomes", str(args.multiset_genomes), "--multiset-batch", str(args.multiset_batch)]
            env = dict(os.environ, TMPDIR="/tmp")
            for kk in ("D2G_K3_COMPACT", "D2G_BS_SORT"):
                env.pop(kk, None)
            left = timeout - (time.monotonic() - t0)
            if left < 20:
                return {"error": "time budget for the counter passes used up"}
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=left, start_new_session=True)
            if r.returncode != 0:
                return {"error": f"rocprofv3 --pmc {ctr} exited with {r.returncode}: " + r.stdout.decode("utf-8", "replace")[-300:]}
            got = []
            for f in sorted(glob.glob(os.path.join(od, "**", "*counter_collection.csv"), recursive=True)):
                for rec in csv.DictReader(open(f)):
                    if rec.get("Counter_Name") == ctr:
                        got.append((rec["Kernel_Name"], int(rec.get("Dispatch_Id", 0) or 0), float(rec["Counter_Value"]) * (1.0 if ctr.startswith("SQ_") else 1024.0)))
            if not got:
                return {"error": f"no {ctr} rows in rocprofv3's output"}
            rows[ctr] = sorted(got, key=lambda x: x[1])
    except subprocess.TimeoutExpired:
        return {"error": f"rocprofv3 pass timed out ({timeout:.0f} s budget)"}
    except Exception as e:                                      # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)

    def total(ctr, pred, largest=False):
        v = [b for (k, _, b) in rows[ctr] if pred(k)]
        if not v:
            return None
        return max(v) if largest else sum(v)

    def both(pred, wide=False, largest=False):
        rd, wr = total("FETCH_SIZE", pred, largest), total("WRITE_SIZE", pred, largest)
        if rd is None or wr is None:
            return None
        return rd * (2 if wide else 1) + wr

    res = {"seconds": round(time.monotonic() - t0, 1)}
    if "k2" in which:
        # the child runs ONE prepare (set creation) and ONE compare launch of the config-3 matrix: every K2 kernel once
        COMPARE = ("k2_bitslice_kernel", "k2_direct_kernel", "k2_bitslice_sparse_kernel", "sp_fill", "sp_list", "sp_rows", "sp_gather", "sp_rowbm", "sp_patch")
        is_k2 = lambda k: any(x in k for x in ("k2_", "bs_", "sp_"))                          # noqa: E731
        is_cmp = lambda k: any(x in k for x in COMPARE)                                       # noqa: E731
        res["k2"] = both(is_cmp)
        res["k2_prepare"] = both(lambda k: is_k2(k) and not is_cmp(k))
        rd, wr = total("FETCH_SIZE", is_k2), total("WRITE_SIZE", is_k2)
        # calibration on the transpose kernel, whose traffic is known: it reads the N x S matrix once (8 S N bytes, 8-byte loads) and
        # writes the padded copy once (8 S Npad): FETCH_SIZE tallies some loads at half their bytes on gfx950 (MI355X_MICROARCH.md)
        N_, S_ = (args.sketches or 10000), args.sketchsize
        t_rd, t_wr = total("FETCH_SIZE", lambda k: "k2_transpose" in k), total("WRITE_SIZE", lambda k: "k2_transpose" in k)
        cal = None
        if t_rd and t_wr:
            cal = {"kernel": "k2_transpose_kernel", "known_read_bytes": 8 * S_ * N_, "counter_read_bytes_raw": t_rd, "read_scale": 8 * S_ * N_ / t_rd,
                   "known_write_bytes": 8 * S_ * ((N_ + 255) // 256 * 256), "counter_write_bytes": t_wr, "write_scale": 8 * S_ * ((N_ + 255) // 256 * 256) / t_wr}
        if rd is not None and wr is not None:
            res["k2_step"] = {"read_raw": rd, "read_x2": 2 * rd, "write": wr, "raw_total": rd + wr, "x2_total": 2 * rd + wr, "calibration": cal,
                              "per_kernel": {re_short(k): {"read_raw": total("FETCH_SIZE", lambda q, k=k: q == k), "write": total("WRITE_SIZE", lambda q, k=k: q == k)}
                                             for k in sorted({k for (k, _, _) in rows["FETCH_SIZE"] if is_k2(k)})}}
    if "k1" in which:
        res["k1"] = both(lambda k: "k1_oph_kernel" in k, wide=True, largest=True)
        res["k1_valu_wave_insts"] = total("SQ_INSTS_VALU", lambda k: "k1_oph_kernel" in k, largest=True) if "SQ_INSTS_VALU" in rows else None
    if "k3" in which:
        # the child makes ONE default call, then ONE compact call: split the k3 dispatches at the first compact-only kernel
        def split(ctr):
            d, c, seen_c = 0.0, 0.0, False
            for (k, _, b) in rows[ctr]:
                if not ("k3_" in k or "k3c_" in k):
                    continue
                if any(x in k for x in K3_COMPACT_ONLY):
                    seen_c = True
                # the second call starts with its init kernel; everything after the first call's verify belongs to it
                if seen_c:
                    c += b
                else:
                    d += b
            return d, c
        (rd_d, rd_c), (wr_d, wr_c) = split("FETCH_SIZE"), split("WRITE_SIZE")
        # k3_bmh_init of the second call precedes its first compact-only kernel: negligible (it writes n*S*8 bytes), left with the default call
        res["k3"] = rd_d + wr_d if (rd_d + wr_d) > 0 else None
        res["k3_compact"] = rd_c + wr_c if (rd_c + wr_c) > 0 else None
    return res


def run_single(args):
    import torch
    import dashing2_amd as D
    from dashing2_amd import synth

    rank, world = 0, 1
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    ctx = D.Context(0)
    algo = {"auto": D.CMP_AUTO, "direct": D.CMP_DIRECT, "bitslice": D.CMP_BITSLICE}[args.algo]
    S = args.sketchsize
    N = args.sketches or 10000
    workload = "BASELINE config 3" if (N, S) == (10000, 1024) else "custom"
    pairs_total = N * (N - 1) // 2
    r0, r1 = 0, N
    my_pairs = pairs_total
    stream = torch.cuda.current_stream().cuda_stream
    ncores = host_cores()

    def make_sketches(n, seed=20260928, collisions=0):
        regs = synth.synthetic_registers(n, S, nclusters=max(8, n // 150), seed=seed)
        if collisions:
            regs = synth.add_chance_collisions(regs, collisions, seed=seed + 1)
        return D.oph_finalize(regs, S, nthreads=ncores)

    # ---- synthetic pre-built sketches, resident in HBM before the timed region
    sig_np, cards_np = make_sketches(N)
    sig_dev = torch.from_numpy(sig_np.view(np.int64)).to(dev)
    lut = torch.from_numpy(D.epilogue_lut(S, D.SIMILARITY, 31)).to(dev)
    out = torch.empty(max(my_pairs, 1), dtype=torch.float32, device=dev)
    cs = ctx.cmp_set_dev(sig_dev.data_ptr(), N, S, algo=algo, stream=stream)

    def step():
        # the launch's output is announced ahead of the prepare: the prepare's latency-bound kernels carry the 200 MB fill of "no register
        # equal" as extra workgroups (d2g_cmp_ut_announce_dev; nothing is enqueued by the announcement itself, D2G_SP_RIDE=0 = off)
        cs.announce_ut_dev(out.data_ptr(), r0, r1, lut_dev_ptr=lut.data_ptr())
        cs.update_dev(sig_dev.data_ptr(), stream)           # transpose + ids + planes + families + pair list (+ the fill) (async)
        cs.lut_ut_dev(lut.data_ptr(), out.data_ptr(), r0, r1, stream)   # pair kernel + fused epilogue

    def barrier():
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
    max_distinct, nbits, mean_nbits = cs.planes(stream)
    sparse_info = cs.sparse_info(stream)
    ms_per_step = dt / args.steps * 1e3
    value = pairs_total / (dt / args.steps)
    algo_used = cs.algo
    ops_extra = getattr(D, "BITSLICE_OPS_PER_GROUP_EXTRA", 2)   # VALU ops per pair and 32-register group beyond the id planes

    # ---- roofline of the dominant kernel (the pair kernel), rank 0's launch
    alg_bytes = 8 * S * N + 4 * my_pairs          # each sketch read once + one float per pair
    compare_only = alg_bytes / (k2_ms * 1e-3) / 1e9 if k2_ms > 0 else 0.0
    sparse_ran = bool(sparse_info.get("sorted_operand")) and not sparse_info.get("dense_kernel_ran")
    kname = ("k2 step = prepare chain (transpose, rank, column plan, planes, families, pair list, sorted stream; its small kernels carry the announced "
             "output's fill) + compare launch (k2_bitslice_sparse_kernel over %d listed tiles, %d pair-list entries)" % (sparse_info.get("tiles_listed", 0), sparse_info.get("pairs_listed", 0)) if sparse_ran
             else "k2 step = prepare chain + k2_bitslice_kernel" if algo_used == D.CMP_BITSLICE else "k2 step = transpose + k2_direct_kernel")
    # the 8 S N bytes of the sketches are read by the PREPARE chain and the 4-byte outputs are written by the compare launch, so
    # the roofline figure of this job divides its algorithmic bytes by BOTH (hipEvents: the prepare chain bracketed on the warm-up steps, the
    # compare launch on every timed step); the compare launch on its own is the named secondary `compare_launch`
    step_ev_ms = k2_ms + prep_ms
    achieved = alg_bytes / (step_ev_ms * 1e-3) / 1e9 if step_ev_ms > 0 else 0.0
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "frac_of_wall_clock_step": alg_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "compare_launch": {"kernel_ms": k2_ms, "achieved": compare_only, "frac": compare_only / HBM_PEAK_GBS,
                                   "note": "the same algorithmic bytes over the compare launch alone (round 4's headline figure; the sketches' 8 S N bytes are NOT read by it, and since "
                                           "the output is announced ahead of the prepare the 4-byte fill is not written by it either: a secondary figure, not the roofline)"},
                "traffic": None, "traffic_source": None,                            # this run's own counter passes, below
                "traffic_note": "HBM bytes of ONE whole step (every kernel of the prepare chain + the compare launch): rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE "
                                "(separate passes).  `traffic` = read side x read_scale + write side, read_scale calibrated in the same run on the transpose "
                                "kernel (known 8 S N bytes in); traffic_detail has the raw and the x2 figures and the per-kernel table (the counter child prepares by CREATING its set, which "
                                "cannot be announced to: its 200 MB fill shows as sp_fill_kernel there, the same bytes the timed step's riders write)",
                "kernel": kname, "kernel_ms": step_ev_ms, "launches": nk2, "algorithmic_bytes": alg_bytes,
                "prep_ms": prep_ms, "compare_ms": k2_ms,
                "note": ("sparse tiles + pair list: only the 32 x 256 tiles a family's rows and columns meet in are walked, pairs of different families that share a value "
                         "come from a list, the rest of the output is a fill with the value of 0 equal registers; D2G_BS_SPARSE=0 walks every tile (dense_walk below)"
                         if sparse_ran else "equality counting is VALU-bound, not HBM-bound (SURVEY 8d); see compute")}

    def valu(pairs, mean_planes, ms, sp=None):
        """lane-ops per second of the launch.  The dense walk compares every pair; a sparse-tile launch only the pairs of its listed
        32 x 256 tiles (the rest of the output is a fill): its lane-ops are counted over those, so the fraction stays a fraction"""
        if sp and sp.get("sorted_operand") and not sp.get("dense_kernel_ran"):
            pairs = min(pairs, sp.get("tiles_listed", 0) * 32 * 256)
        if algo_used == D.CMP_BITSLICE:
            ops = pairs * ((S + 31) // 32) * (mean_planes + ops_extra)
        else:
            ops = pairs * S * 2
        a = ops / (ms * 1e-3) if ms > 0 else 0.0
        return a, a / VALU_PEAK_LANEOPS

    va, vf = valu(my_pairs, mean_nbits, k2_ms, sparse_info)
    compute = {"bound": "valu", "unit": "lane-ops/s", "achieved": va, "peak": VALU_PEAK_LANEOPS, "frac": vf,
               "bit_planes_max": nbits, "bit_planes_mean": mean_nbits, "max_shared_values_per_column_plus1": max_distinct, "sparse": sparse_info,
               "note": ("lane-ops of the pairs of the LISTED tiles (tiles_listed x 32 x 256) over the whole compare launch (fill, list and pair kernel): the sparse path "
                        "executes nothing for the other pairs" if sparse_ran else "lane-ops of every pair of the triangle over the pair kernel")}

    def measure_matrix(bits_np, n, steps=5):
        """prepare + pair kernel of an n x S matrix on this GPU, whole triangle; returns a small dict.  VERDICT r5 #5: every figure of a line comes from the SAME
        steady-state steps (both event brackets on in every one of them: kernel_ms + prep_ms <= ms_per_step, hbm_frac from ms_per_step); what a set WITHOUT history
        pays -- the CLI's one-shot cmp -- is `first_step_ms`: the set's remembered decisions forgotten (d2g_cmp_set_forget), one synchronised step by the wall clock,
        beside `single_step_ms`, one synchronised steady-state step measured the same way"""
        t_dev = torch.from_numpy(bits_np.view(np.int64)).to(dev)
        o = torch.empty(n * (n - 1) // 2, dtype=torch.float32, device=dev)
        c = ctx.cmp_set_dev(t_dev.data_ptr(), n, S, algo=algo, stream=stream)     # (creation = allocations + a first prepare: not timed)

        def one():
            c.announce_ut_dev(o.data_ptr(), 0, n, lut_dev_ptr=lut.data_ptr())
            c.update_dev(t_dev.data_ptr(), stream)
            c.lut_ut_dev(lut.data_ptr(), o.data_ptr(), 0, n, stream)

        def synced():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            one()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e3

        ctx.set_timing(False)
        one()
        c.forget()
        first_ms = synced()
        sp_first = c.sparse_info(stream)
        for _ in range(2):
            one()
        single_ms = min(synced() for _ in range(3))
        ctx.set_timing(D.TIME_K2 | D.TIME_K2PREP)
        ctx.kernel_ms("k2")
        ctx.kernel_ms("k2prep")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        torch.cuda.synchronize()
        d = (time.perf_counter() - t0) / steps
        ctx.set_timing(False)
        _, kms, _ = ctx.kernel_ms("k2")
        _, pms, _ = ctx.kernel_ms("k2prep")
        md, nb, mean = c.planes(stream)
        sp = c.sparse_info(stream)
        c.close()
        npairs = n * (n - 1) // 2
        ab = 8 * S * n + 4 * npairs
        return {"sketches": n, "pairs_per_s": npairs / d, "ms_per_step": d * 1e3, "kernel_ms": kms, "prep_ms": pms,
                "first_step_ms": first_ms, "single_step_ms": single_ms,
                "first_step_path": "dense walk" if (sp_first.get("dense_kernel_ran") or not sp_first.get("sorted_operand")) else "tiles + pair list",
                "bit_planes_max": nb, "bit_planes_mean": mean, "max_shared_values_per_column_plus1": md,
                "hbm_frac": ab / (d * 1e3 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "valu_frac": valu(npairs, mean, kms, sp)[1], "sparse": sp}

    # the output of the LAST timed step (announce + prepare + launch, exactly what the headline times) is kept and compared
    # bit for bit with what the cpu_baseline leg computes for the same matrix (the oracle, here only as the checker)
    out_host = None if args.no_cpu_baseline else out.cpu().numpy()
    config4 = None
    if True:
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
                    "hbm_frac": achieved / HBM_PEAK_GBS, "valu_frac": vf, "sparse": sparse_info}
                # the SAME matrix with every tile walked (D2G_BS_SPARSE=0: round 3's algorithm): what the headline owes to the matrix's block structure
                set_switch(ctx, "D2G_BS_SPARSE", "0")
                try:
                    dense_walk = measure_matrix(sig_np.view(np.uint64), N)
                finally:
                    set_switch(ctx, "D2G_BS_SPARSE", None)
                dense_walk["note"] = "the stated matrix, D2G_BS_SPARSE=0: every 32 x 256 tile walked by k2_bitslice_kernel"
                mats["stated, dense walk (D2G_BS_SPARSE=0)"] = dense_walk
                # the stated families + c chance collisions per sketch with random strangers (about half of them with values the stranger shares with
                # its whole family): the case real collections present (conserved k-mers across genera).  Round 4 listed nearly every tile at c = 10.
                def with_dense_reference(bits):
                    """the matrix on the default path, and beside it what D2G_BS_SPARSE=0 (every tile walked, no ordering) takes for its first and its steady
                    step: `first_step_over_dense` is what a one-shot CLI cmp pays for the sparse path's look at a matrix it cannot help"""
                    r = measure_matrix(bits, N)
                    set_switch(ctx, "D2G_BS_SPARSE", "0")
                    try:
                        d = measure_matrix(bits, N, steps=3)
                    finally:
                        set_switch(ctx, "D2G_BS_SPARSE", None)
                    r["dense_walk"] = {"ms_per_step": d["ms_per_step"], "first_step_ms": d["first_step_ms"], "single_step_ms": d["single_step_ms"]}
                    r["first_step_over_dense"] = r["first_step_ms"] / d["first_step_ms"] if d["first_step_ms"] else None
                    r["step_over_dense"] = r["ms_per_step"] / d["ms_per_step"] if d["ms_per_step"] else None
                    return r
                for c in (1, 3, 10, 30, 100):
                    sg, _ = make_sketches(N, collisions=c)
                    mats["stated + %d chance collisions per sketch" % c] = with_dense_reference(sg.view(np.uint64))
                    del sg
                # VERDICT r5 #1c: collections whose registers come from GENOMES through the product's K1: 66 families of 150 mutated copies of a 200 kbp genome
                # (substitution rate log-uniform in [0.0005, 0.03]); without conserved segments random 31-mers of different families never collide; with 48
                # conserved 1.5 kbp segments of which every family carries two, families that share a segment share the registers its k-mers win
                if not args.no_sketch and N == 10000:
                    for label, kw in (("K1-built: 66 families x 150 genomes of 200 kbp, no conserved segment", {}),
                                      ("K1-built: the same + 48 conserved 1.5 kbp segments, two per family", {"nseg": 48, "seg_len": 1500, "segs_per_family": 2})):
                        t0 = time.perf_counter()
                        sg, _, rg, fam = k1_built_collection(D, synth, ctx, S, ncores, 66, N // 66 + 1, 200_000, **kw)
                        sg, rg, fam = sg[:N], rg[:N], fam[:N]
                        r = with_dense_reference(np.ascontiguousarray(sg).view(np.uint64))
                        r["collection"] = dict(cross_family_stats(rg[:, :S], fam), build_s=time.perf_counter() - t0,
                                               note="registers from the product's own K1 over synthetic genomes (FASTA -> d2g_seqpack -> k1_oph_kernel); the cross-family "
                                                    "figures are counted on the host from the u64 registers and the generator's family labels")
                        mats[label] = r
                        del sg, rg
                mats["unrelated (no value shared by two sketches)"] = measure_matrix(synth.unrelated_registers(N, S), N)
                mats["adversarial (every value occurs exactly twice in its column)"] = with_dense_reference(synth.paired_registers(N, S))
                # columns that share between 0 and 64 values (log-uniform): which column lands in which 32-register group matters,
                # since a group walks the MAXIMUM plane count of its columns.  Measured with the column plan (columns sorted by
                # plane class before grouping, the default) and with the caller's column order (D2G_BS_SORT=0).
                sk = synth.skewed_registers(N, S)
                mats["skewed (columns share 0..64 values, log-uniform)"] = with_dense_reference(sk)
                set_switch(ctx, "D2G_BS_SORT", "0")
                try:
                    mats["skewed, columns left in the caller's order (D2G_BS_SORT=0)"] = measure_matrix(sk, N)
                finally:
                    set_switch(ctx, "D2G_BS_SORT", None)
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
        """a secondary leg returns (seconds, build(seconds) -> dict); an exception becomes {"error": ...} instead of losing the primary line"""
        try:
            secs, build = fn()
            return build(secs)
        except Exception as e:                                   # noqa: BLE001 - reported, not swallowed
            return {"error": f"{type(e).__name__}: {e}"}

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

        set_switch(ctx, "D2G_K3_COMPACT", None)
        mdt, ncalls, k3_ms = k3_measure()
        sig_default = sig3.clone()
        # the low-traffic variant of the same chain (4-byte stored words + tile-sorted split, D2G_K3_COMPACT=1): identical results
        set_switch(ctx, "D2G_K3_COMPACT", "1")
        try:
            cdt, _, ck3_ms = k3_measure()
            same = bool(torch.equal(sig3.view(torch.int64), sig_default.view(torch.int64)))
        finally:
            set_switch(ctx, "D2G_K3_COMPACT", None)
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

    multiset = None if args.no_multiset else guarded(multiset_leg)

    cpu, checked = None, None
    if not args.no_cpu_baseline:
        cpu, oracle_out, oracle_rows = cpu_baseline(sig_np, cards_np, S, args.cpu_seconds)
        got = out_host[:oracle_out.size].view(np.uint32)
        bad = np.flatnonzero(got != oracle_out.view(np.uint32))
        checked = {"pairs": int(oracle_out.size), "mismatches": int(bad.size), "rows": [0, int(oracle_rows)],
                   "of": "the float32 output of the last timed step (announce_ut_dev + update_dev + lut_ut_dev), bitwise, against the oracle's all-pairs of the same "
                         "matrix (the cpu_baseline leg's own output: rows [0, %d) of the condensed triangle)" % oracle_rows}
        if bad.size:
            checked["first_mismatch"] = {"index": int(bad[0]), "got": float(out_host[bad[0]]), "oracle": float(oracle_out[bad[0]])}
        del out_host, oracle_out

    # ---- HBM traffic of the reported kernels, measured NOW: a child under rocprofv3 --pmc (two passes) re-runs one launch of each
    # on the shapes reported above; the committed profile is only the fallback (and says so)
    if not args.no_traffic:
        del sig_dev
        torch.cuda.empty_cache()
        want = ["k2"] + (["k1"] if isinstance(sketch, dict) and "roofline" in sketch else []) + \
               (["k3"] if isinstance(multiset, dict) and "roofline" in multiset else [])
        tm = measure_traffic(args, which=want)
        src_ok = "this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two passes, %.0f s) over one launch of the same shape" % tm.get("seconds", 0.0)
        src_bad = "profiles/" + os.path.basename(PMC_FILE) + " (committed; this run's counter passes failed: %s)" % tm.get("error", "no rows for this kernel")

        def put(rf, key, shape_ok=True):
            if tm.get(key) and shape_ok:
                rf["traffic"], rf["traffic_source"] = tm[key], src_ok
            elif rf.get("traffic") is not None:
                rf["traffic_source"] = src_bad
        st = tm.get("k2_step")
        if isinstance(st, dict):
            cal = st.get("calibration") or {}
            scale = cal.get("read_scale") or 1.0
            roofline["traffic"] = st["read_raw"] * scale + st["write"]
            roofline["traffic_over_algorithmic"] = roofline["traffic"] / alg_bytes
            roofline["traffic_detail"] = dict(st, compare_launch_raw=tm.get("k2"), prepare_chain_raw=tm.get("k2_prepare"))
            roofline["traffic_source"] = src_ok
        else:
            roofline["traffic_source"] = "not measured: " + str(tm.get("error", "no K2 rows in the counter output"))
        if isinstance(sketch, dict) and "roofline" in sketch:
            put(sketch["roofline"], "k1")
            wi = tm.get("k1_valu_wave_insts")
            if wi and sketch["roofline"].get("kernel_ms"):
                # SQ_INSTS_VALU of the ONE 1000-genome dispatch (wave-instructions; x 64 lanes) over that launch's duration against the VALU
                # peak, and per base: the "VALU-bound at ~125 issue slots per base" claim, checkable from the line
                lane = wi * 64.0
                bases_l = args.sketch_genomes * args.sketch_len
                sketch["roofline"]["valu"] = {"wave_insts": wi, "lane_ops": lane, "achieved": lane / (sketch["roofline"]["kernel_ms"] * 1e-3), "peak": VALU_PEAK_LANEOPS,
                                              "unit": "lane-ops/s", "frac": lane / (sketch["roofline"]["kernel_ms"] * 1e-3) / VALU_PEAK_LANEOPS,
                                              "valu_wave_insts_per_base_per_lane": wi * 64.0 / bases_l,
                                              "source": "this run: rocprofv3 --pmc SQ_INSTS_VALU, the 1000-genome k1_oph_kernel dispatch"}
        if isinstance(multiset, dict) and "roofline" in multiset:
            put(multiset["roofline"], "k3")
            if isinstance(multiset.get("low_traffic_variant"), dict):
                put(multiset["low_traffic_variant"], "k3_compact")
    else:
        for leg in (sketch, multiset):
            if isinstance(leg, dict) and isinstance(leg.get("roofline"), dict) and leg["roofline"].get("traffic") is not None:
                leg["roofline"]["traffic_source"] = "profiles/" + os.path.basename(PMC_FILE) + " (committed; --no-traffic)"

    if True:
        line = {
            "metric": "all-pairs sketch comparison throughput (pairs/s)", "value": value, "unit": "pairs/s",
            "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": f"{workload}: {N} pre-built OPH sketches, S={S}, all-pairs cmp only, {pairs_total} pairs, float32 Jaccard",
                       "sketches": N, "sketchsize": S, "pairs": pairs_total, "algo": "bitslice" if algo_used == D.CMP_BITSLICE else "direct",
                       "step": "announce the output (host-side) + prepare (its small kernels write the output's fill) + pair kernel w/ fused epilogue; "
                               "sketches resident in HBM; every output word is written in every step",
                       "parallelism": "one GPU"},
            "roofline": roofline, "compute": compute, "cpu_baseline": cpu, "config4_1gpu": config4,
            "dense_walk": (compute.get("matrices") or {}).get("stated, dense walk (D2G_BS_SPARSE=0)"),
            "tuning": ctx.tuning(),
            "checked_vs_oracle": checked,
            "sketch": sketch, "multiset_sketch": multiset,
        }
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
    cs.close()
    ctx.close()
    if checked and checked["mismatches"]:
        print("bench.py: the timed step's output differs from the oracle in %d of %d pairs" % (checked["mismatches"], checked["pairs"]), file=sys.stderr, flush=True)
        sys.exit(3)


def main():
    args = parse_args()
    if args.pmc_child:
        return pmc_child(args)
    if args.worker:                      # a measuring child of the supervisor (tests also run one directly at world size 1)
        return run_multi(args)
    if args.gpus > 1:
        raise SystemExit(supervise(args))
    run_single(args)


if __name__ == "__main__":
    main()

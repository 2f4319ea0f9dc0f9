"""Per-rank cost of the multi-GPU bench step (BASELINE config 4, strong scaling, ONE JOB per step -- the definition
bench.py's headline uses at every N) measured on ONE GPU through the loopback transport, + a MODELLED xGMI term.

W contexts on device 0 run exactly the pack / exchange lists / sharded chunked prepare / gather / derive / pair
kernels that W GPUs run; on one device they run one after the other, so every kernel's duration (rocprofv3
--kernel-trace) is what it takes on a GPU of its own.  The report replays ONE rank's step as the engine schedules
it -- two in-order queues, the compute stream and the exchange stream, tied by the engine's events:
    compute:  pack | fill | prep(chunk 0) .. prep(chunk C-1) | derive(0) .. derive(C-1) | order | pair
              (fill = the rank's slab pre-filled with the value of "0 equal", since round 5 at the START of the step, under the first exchange;
               order = the sparse path on the gathered operand, N >= 8192: ids from the planes, families, sort, pair list, sorted stream;
               pair = launch rows, listed tiles, pair list applied)
    exchange: x1(0) .. x1(C-1) | x2(0) .. x2(C-1)            x1(c) before prep(c) before x2(c) before derive(c)
with the kernels at their measured durations and every exchange at (bytes a rank moves over its busiest link) /
(ASSUMED link rate).  A prediction to hold against the driver's SCALE run, not a measurement of it.

usage (GPU box):  tools/mgpu_model.sh            # runs the workers under rocprofv3 and prints the report
       worker:    mgpu_model.py --worker W [N S]  one loopback configuration, a few steps
       report:    mgpu_model.py --report DIR [N S]
"""
import csv
import glob
import os
import re
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

LINK_GBS = 50.0          # assumed sustained payload rate of one xGMI link and direction (peak 64 GB/s)
LAUNCH_US = 6.0          # assumed gap per enqueued kernel / exchange on an in-order queue (dispatch + event hand-off)
REPS = 4


def shape(argv):
    rest = [a for a in argv if not a.startswith("--")]
    N = int(rest[0]) if len(rest) > 0 else 50000
    S = int(rest[1]) if len(rest) > 1 else 1024
    return N, S


def worker(W, N, S):
    import dashing2_amd as D
    from dashing2_amd import synth
    regs = synth.synthetic_registers(N, S, nclusters=max(8, N // 150), seed=20260928)
    bits = D.oph_finalize(regs, S, nthreads=32)[0].view(np.uint64)
    lut = D.epilogue_lut(S, D.SIMILARITY, 31)

    def upload(ctx, arr):
        p = ctx.malloc(max(arr.nbytes, 8))
        ctx.h2d(p, np.ascontiguousarray(arr))
        return p

    ctxs = [D.Context(0) for _ in range(W)]
    comms = D.Comm.create_all(ctxs)
    engs = [D.AllPairs(ctxs[r], comms[r], N, S) for r in range(W)]
    held = [e.rows_held for e in engs]
    rows = [upload(ctxs[r], bits[held[r][0]:held[r][1]]) for r in range(W)]
    outs = [ctxs[r].malloc(max(D.ut_count(N, *engs[r].rows_computed), 1) * 4) for r in range(W)]
    luts = [upload(ctxs[r], lut) for r in range(W)]
    D.allpairs_step_all(engs, rows, luts, outs)              # warm-up (allocations, code objects)
    for c in ctxs:
        c.sync()
    t = time.perf_counter()
    for _ in range(REPS):
        D.allpairs_step_all(engs, rows, luts, outs)
    for c in ctxs:
        c.sync()
    wall = (time.perf_counter() - t) / REPS * 1e3
    md, nb, mean = engs[0].operand().planes()
    print(f"WORKER W={W} N={N} S={S} chunks={engs[0].chunks} loopback_step_ms={wall:.3f} planes_max={nb} planes_mean={mean:.2f}", flush=True)


def kernel_times(d):
    """per kernel short name: list of durations (us) in launch order"""
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
        for r in rows:
            n = re.sub(r"^void\s+", "", r["Kernel_Name"]).replace("(anonymous namespace)::", "")
            n = re.split(r"[(<]", n, 1)[0]
            out.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return out


def replay(W, C, t, x1_ms, x2_ms):
    """one rank's step: two in-order queues + the engine's event edges; returns (makespan, busy compute, busy exchange) in ms"""
    gap = LAUNCH_US / 1e3
    prep = t["prep_chunk"]
    comp = t["pack"] + gap
    ex = comp                                   # x1(0) waits for pack
    comp += t["fill"] + gap                     # the slab's fill right behind the pack, while x1 is on the links
    x1_done, prep_done, x2_done = [], [], []
    for c in range(C):
        ex += (x1_ms + gap) if W > 1 else gap
        x1_done.append(ex)
    for c in range(C):
        comp = max(comp, x1_done[c]) + prep + 4 * gap          # transpose, rank, column plan, planes
        prep_done.append(comp)
    for c in range(C):
        ex = max(ex, prep_done[c]) + ((x2_ms + gap) if W > 1 else gap)
        x2_done.append(ex)
    for c in range(C):
        comp = max(comp, x2_done[c]) + t["derive_chunk"] + gap
    comp += t["order"] + t["order_launches"] * gap + t["pair"] + t["pair_launches"] * gap
    return comp


def report(d, N, S):
    import dashing2_amd as D
    gw, ng = D.operand_layout(N, S)
    pairs = N * (N - 1) // 2
    base = None
    print(__doc__.split("usage")[0].strip())
    print()
    print(f"N = {N}, S = {S}; assumed {LINK_GBS:.0f} GB/s per xGMI link and direction, {LAUNCH_US:.0f} us per enqueued operation; kernel durations: rocprofv3, "
          f"mean over {REPS + 1} steps x W ranks (loopback on one MI355X)")
    for W in (1, 2, 4, 8):
        wd = os.path.join(d, f"W{W}")
        log = open(os.path.join(wd, "worker.log")).read()
        m = re.search(r"WORKER W=(\d+) N=(\d+) S=(\d+) chunks=(\d+) loopback_step_ms=([\d.]+) planes_max=(\d+) planes_mean=([\d.]+)", log)
        if not m:
            print(f"W={W}: worker failed:\n{log[-500:]}")
            continue
        C, wall = int(m.group(4)), float(m.group(5))
        kt = kernel_times(wd)
        steps = (REPS + 1) * W                                  # rank-steps in the trace

        def per_step(name):
            v = kt.get(name, [])
            return sum(v) / steps / 1e3 if v else 0.0           # ms per rank-step

        prep_kernels = ("k2_transpose_kernel", "bs_rank_kernel", "bs_colplan_kernel", "bs_planes_kernel")
        order_kernels = ("sp_unpack_kernel", "sp_link_kernel", "sp_flatten_kernel", "sp_attach_kernel", "sp_count_kernel", "sp_scan_kernel",
                         "sp_place_kernel", "sp_emit_kernel", "sp_pairs_kernel", "sp_permute_lds_kernel", "sp_bin_kernel")
        pair_kernels = ("sp_rows_kernel", "sp_gather_kernel", "sp_rowbm_kernel", "sp_list_kernel", "sp_compose_kernel", "k2_bitslice_sparse_kernel", "k2_bitslice_kernel")

        def launches(names):
            return sum(len(kt.get(k, [])) for k in names) / steps

        t = {"pack": per_step("mg_pack_kernel"), "prep_chunk": sum(per_step(k) for k in prep_kernels) / C,
             "derive_chunk": per_step("bs_derive_kernel") / C, "pair": sum(per_step(k) for k in pair_kernels), "pair_launches": max(1.0, launches(pair_kernels)),
             "order": sum(per_step(k) for k in order_kernels), "order_launches": launches(order_kernels), "fill": per_step("sp_fill_kernel")}
        a2a = (N / W) * S * 8 * (W - 1) / W                     # bytes a rank sends (= receives) in the row->column exchange
        gat = (ng - ng / W) * gw * 4                            # bytes a rank receives in the gather
        links = max(W - 1, 1)
        x1 = a2a / links / C / (LINK_GBS * 1e9) * 1e3 if W > 1 else 0.0
        x2 = gat / links / C / (LINK_GBS * 1e9) * 1e3 if W > 1 else 0.0
        step = replay(W, C, t, x1, x2)
        serial = t["pack"] + t["fill"] + C * (t["prep_chunk"] + t["derive_chunk"]) + t["order"] + t["pair"] + C * (x1 + x2)
        if base is None:
            base = step
        print(f"W={W} (C={C} chunks): per rank  pack {t['pack']:.3f}  fill {t['fill']:.3f} (under x1)  prepare {C}x{t['prep_chunk']:.3f}  derive {C}x{t['derive_chunk']:.3f}  order {t['order']:.3f} ({t['order_launches']:.0f} launches)  pair {t['pair']:.3f} ({t['pair_launches']:.0f} launches) ms;  "
              f"moves {a2a / 1e6:6.1f} MB out+in (rows->columns) + {gat / 1e6:6.1f} MB in (gather) over {links} link(s): {C}x{x1:.3f} + {C}x{x2:.3f} ms;  "
              f"ONE-JOB step {step:.3f} ms = {pairs / (step * 1e-3):.3e} pairs/s ({base / step:.2f}x of W=1)  [no overlap inside the step: {serial:.3f} ms, {base / serial:.2f}x];  "
              f"loopback wall {wall:.2f} ms for all {W} ranks on one device")
        print("      kernels (us per rank-step): " + "  ".join(f"{k.replace('_kernel', '')} {per_step(k) * 1e3:.0f}" for k in order_kernels + pair_kernels + ("sp_fill_kernel",) if kt.get(k)))
        if W == 8:
            # what no schedule can go below on this design at this size: the two exchanges at the assumed link rate (the second needs the first
            # chunk's prepare), the replicated part of the order phase (every rank needs the whole sorted operand: unpack + sort + pair list +
            # permute), the pair phase; the slab's output write (fill) hides under the first exchange
            floor = t["pack"] + x1 + t["prep_chunk"] + C * x2 + t["derive_chunk"] + t["order"] + t["pair"]
            print(f"      floor of this design at W=8: pack {t['pack']:.3f} + first x1 {x1:.3f} + one chunk's prepare {t['prep_chunk']:.3f} + x2 {C}x{x2:.3f} + last derive {t['derive_chunk']:.3f} + order {t['order']:.3f} "
                  f"+ pair {t['pair']:.3f} = {floor:.3f} ms ({base / floor:.2f}x of W=1): the order phase is REPLICATED (every rank needs the whole operand in family order) and does not shrink with W")
            print(f"SUMMARY N={N} W8_step_ms={step:.3f} W1_engine_step_ms={base:.3f} W8_speedup_vs_engine_W1={base / step:.2f} floor_ms={floor:.3f}")


if __name__ == "__main__":
    if "--worker" in sys.argv:
        i = sys.argv.index("--worker")
        W = int(sys.argv[i + 1])
        N, S = shape(sys.argv[i + 2:])
        worker(W, N, S)
    elif "--report" in sys.argv:
        i = sys.argv.index("--report")
        N, S = shape(sys.argv[i + 2:])
        report(sys.argv[i + 1], N, S)
    else:
        print(__doc__)

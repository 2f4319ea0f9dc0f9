"""Per-rank cost of the multi-GPU bench step (BASELINE config 4, strong scaling) measured on ONE GPU through the
loopback transport: W contexts on device 0 run the same pack / exchange lists / sharded prepare / gather / pair
kernels that W GPUs run, one after the other, so (time of a W-rank step) / W is one rank's device work with the
exchanges executed as HBM copies.  The xGMI time is then MODELLED from the bytes each rank moves (printed) --
this is a prediction to hold against the driver's SCALE run, not a measurement of it.
usage: mgpu_model.py [N=50000] [S=1024]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dashing2_amd as D
from dashing2_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
regs = synth.synthetic_registers(N, S, nclusters=max(8, N // 150), seed=20260928)
bits = D.oph_finalize(regs, S, nthreads=32)[0].view(np.uint64)
lut = D.epilogue_lut(S, D.SIMILARITY, 31)
LINK_GBS = 50.0          # assumed sustained payload rate of one xGMI link and direction (peak 64 GB/s)


def upload(ctx, arr):
    p = ctx.malloc(max(arr.nbytes, 8))
    ctx.h2d(p, np.ascontiguousarray(arr))
    return p


base = None
for W in (1, 2, 4, 8):
    ctxs = [D.Context(0) for _ in range(W)]
    comms = D.Comm.create_all(ctxs)
    engs = [D.AllPairs(ctxs[r], comms[r], N, S) for r in range(W)]
    held = [e.rows_held for e in engs]
    rows = [upload(ctxs[r], bits[held[r][0]:held[r][1]]) for r in range(W)]
    outs = [ctxs[r].malloc(max(D.ut_count(N, *engs[r].rows_computed), 1) * 4) for r in range(W)]
    luts = [upload(ctxs[r], lut) for r in range(W)]
    for _ in range(2):
        D.allpairs_step_all(engs, rows, luts, outs)
    for c in ctxs:
        c.sync()
    reps = 5
    t = time.perf_counter()
    for _ in range(reps):
        D.allpairs_step_all(engs, rows, luts, outs)
    for c in ctxs:
        c.sync()
    step = (time.perf_counter() - t) / reps * 1e3
    # prepare alone (exchange + sharded prepare + gather, no pair kernel)
    t = time.perf_counter()
    for _ in range(reps):
        D.allpairs_prepare_all(engs, rows)
    for c in ctxs:
        c.sync()
    prep = (time.perf_counter() - t) / reps * 1e3
    gw, ng = D.operand_layout(N, S)
    a2a = (N / W) * S * 8 * (W - 1) / W                 # bytes a rank sends (= receives) in the all-to-all
    gat = (ng - ng / W) * gw * 4                        # bytes a rank receives in the gather
    links = max(W - 1, 1)
    comm_ms = 0.0 if W == 1 else (a2a / links + gat / links) / (LINK_GBS * 1e9) * 1e3
    per_rank = step / W
    pairs = N * (N - 1) // 2
    if base is None:
        base = per_rank
    print(f"W={W}: loopback step {step:8.3f} ms = {per_rank:7.3f} ms per rank (exchange+prepare {prep / W:6.3f}, pair {(step - prep) / W:6.3f});  "
          f"rank moves {a2a / 1e6:6.1f} MB out+in (all-to-all) + {gat / 1e6:6.1f} MB in (gather) over {links} links: ~{comm_ms:5.3f} ms at {LINK_GBS:.0f} GB/s/link;  "
          f"serial model {pairs / ((per_rank + comm_ms) * 1e-3):.3e} pairs/s ({base / (per_rank + comm_ms):.2f}x), "
          f"pipelined model {pairs / (max((step - prep) / W, prep / W + comm_ms) * 1e-3):.3e} pairs/s ({base / max((step - prep) / W, prep / W + comm_ms):.2f}x)")
    for r in range(W):
        for p in (rows[r], outs[r], luts[r]):
            ctxs[r].free(p)
    for e in engs:
        e.close()

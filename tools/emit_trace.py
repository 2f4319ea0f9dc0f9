#!/usr/bin/env python3
"""Per-workgroup timeline of sp_emit_kernel's first step (variant build: tools/build_variant.sh trace -DD2G_SP_TRACE; D2G_LIB=dashing2_amd/libd2g_trace.so).
MATRIX / C / N as tools/k2_time.py.  Stamps (s_memrealtime, 100 MHz): 0 start, 1 LDS cleared, 2 pass A done, 3 counts read + mixed vote, 4 scan + places,
5 stream places reserved, 6 pass B done, 7 slots scanned + list places reserved, 8 records written."""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                   # noqa: E402
import dashing2_amd as D                       # noqa: E402
from dashing2_amd import synth                # noqa: E402
from dashing2_amd.capi import lib             # noqa: E402
N, S = int(os.environ.get("N", 10000)), 1024
which = os.environ.get("MATRIX", "noise")
regs = synth.synthetic_registers(N, S, nclusters=max(8, N // 150), seed=20260928)
if which == "noise": regs = synth.add_chance_collisions(regs, int(os.environ.get("C", 10)), seed=20260929)
sig = D.oph_finalize(regs, S, nthreads=32)[0]
ctx = D.Context(0)
t = torch.from_numpy(sig.view(np.int64)).to("cuda:0")
st = torch.cuda.current_stream().cuda_stream
cs = ctx.cmp_set_dev(t.data_ptr(), N, S, algo=D.CMP_BITSLICE, stream=st)
for _ in range(4): cs.update_dev(t.data_ptr(), st)
torch.cuda.synchronize()
buf = np.zeros(4096 * 16, dtype=np.uint64)
f = lib().d2g_debug_emit_trace; f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_size_t]
assert f(buf.ctypes.data, buf.size) == 0
tr = buf.reshape(-1, 16).astype(np.int64)[:S]
t0 = tr[:, 0].min()
us = (tr - t0) / 100.0
def stats(x): return f"mean {x.mean():7.2f}  p10 {np.percentile(x, 10):7.2f}  p50 {np.percentile(x, 50):7.2f}  p90 {np.percentile(x, 90):7.2f}  max {x.max():7.2f}"
names = ["start", "LDS cleared (0->1)", "pass A (1->2)", "counts + vote (2->3)", "scan + places (3->4)", "stream places: 2 global atomics (4->5)", "pass B (5->6)", "slot scan + list places: 1 global atomic (6->7)", "records (7->8)"]
print(f"{which} C={os.environ.get('C', '-')} N={N}: first step of {S} workgroups")
print(f"{names[0]:48s}", stats(us[:, 0]))
full = tr[:, 8] >= tr[:, 0]
print("workgroups that reached stamp 8 (a mixed value in the step):", int(full.sum()))
w = us[full] if full.any() else us
last = 8 if full.any() else 3
for k in range(1, last + 1):
    print(f"{names[k]:48s}", stats(w[:, k] - w[:, k - 1]))
print(f"{'end of the first step':48s}", stats(w[:, last]))

#!/usr/bin/env python3
"""Per-workgroup timeline of the sparse pair kernel (variant build -DD2G_SP_TRACE: tools/build_variant.sh trace -DD2G_SP_TRACE;
run with D2G_LIB=dashing2_amd/libd2g_trace.so).  Stamps (s_memrealtime, 100 MHz): 0 start, 1 sub-tile found + LDS cleared,
2 plane walk done, 3 LDS reduction done, 4 epilogue done, 5 loop left, 6 pair-list tail done."""
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
dev = torch.device("cuda", 0)
ctx = D.Context(0)
regs = synth.synthetic_registers(N, S, nclusters=max(8, N // 150), seed=20260928)
t = torch.from_numpy(regs.view(np.int64)).to(dev)
lut = torch.from_numpy(D.epilogue_lut(S, D.SIMILARITY, 31)).to(dev)
out = torch.empty(N * (N - 1) // 2, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
cs = ctx.cmp_set_dev(t.data_ptr(), N, S, algo=D.CMP_BITSLICE, stream=st)
for _ in range(4):
    cs.update_dev(t.data_ptr(), st)
    cs.lut_ut_dev(lut.data_ptr(), out.data_ptr(), 0, N, st)
torch.cuda.synchronize()
buf = np.zeros(16384 * 8, dtype=np.uint64)
f = lib().d2g_debug_sp_trace
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_size_t]
assert f(buf.ctypes.data, buf.size) == 0
tr = buf.reshape(-1, 8).astype(np.int64)
live = tr[:, 0] > 0
tr = tr[live]
t0 = tr[:, 0].min()
us = (tr - t0) / 100.0                          # 100 MHz -> us
worked = tr[:, 1] > 0                           # workgroups that found a sub-tile (stamp 1 written in this launch: stale stamps of earlier launches are older than t0 -> negative)
worked &= (us[:, 1] >= 0)
print(f"{live.sum()} workgroups stamped, {worked.sum()} walked a sub-tile; kernel span {us[:, [0, 5, 6]].max():.1f} us")
w = us[worked]
def stats(x): return f"mean {x.mean():6.2f}  p10 {np.percentile(x, 10):6.2f}  p50 {np.percentile(x, 50):6.2f}  p90 {np.percentile(x, 90):6.2f}  max {x.max():6.2f}"
print("start of the workgroup      ", stats(w[:, 0]))
print("prologue (0 -> 1)           ", stats(w[:, 1] - w[:, 0]))
print("plane walk (1 -> 2)         ", stats(w[:, 2] - w[:, 1]))
print("LDS reduction (2 -> 3)      ", stats(w[:, 3] - w[:, 2]))
print("epilogue (3 -> 4)           ", stats(w[:, 4] - w[:, 3]))
print("pair-list tail (5 -> 6)     ", stats(w[:, 6] - w[:, 5]))
print("end of the workgroup        ", stats(w[:, 6]))
idle = us[~worked]
if len(idle):
    print("workgroups without a sub-tile: start", stats(idle[:, 0]), " end", stats(idle[:, 6]))
ids = np.nonzero(live)[0][worked]
print("per XCD (blockIdx % 8): sub-tiles walked, mean / max plane walk, last end")
for q in range(8):
    m = (ids & 7) == q
    if m.any():
        print(f"  XCD {q}: {m.sum():4d}  walk mean {(w[m, 2] - w[m, 1]).mean():6.2f}  max {(w[m, 2] - w[m, 1]).max():6.2f}  prologue mean {(w[m, 1] - w[m, 0]).mean():5.2f}  last end {w[m, 6].max():6.2f}")
order = np.argsort(w[:, 2] - w[:, 1])[::-1][:12]
print("slowest plane walks: (blockIdx, start, walk us, end)", [(int(ids[k]), round(float(w[k, 0]), 2), round(float(w[k, 2] - w[k, 1]), 2), round(float(w[k, 6]), 2)) for k in order])

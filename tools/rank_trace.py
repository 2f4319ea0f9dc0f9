#!/usr/bin/env python3
"""Per-workgroup timeline of bs_rank_kernel (variant build: tools/build_variant.sh ranktrace -DD2G_RANK_TRACE; run with
D2G_LIB=dashing2_amd/libd2g_ranktrace.so).  N=10000 (config 3: the FAST kernel) or N=50000 (config 4: four partition passes per column).
Stamps (s_memrealtime, 100 MHz) -- FAST: 0 start, 1 values loaded + table cleared, 2 claims done, 3 confirms done, 4 compaction done, 5 ids stored;
general: 0 start, 1+2p walk of pass p done, 2+2p its compaction + final ids done."""
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
which = os.environ.get("MATRIX", "stated")
dev = torch.device("cuda", 0)
ctx = D.Context(0)
regs = synth.unrelated_registers(N, S) if which == "unrelated" else synth.synthetic_registers(N, S, nclusters=max(8, N // 150), seed=20260928)
t = torch.from_numpy(regs.view(np.int64)).to(dev)
st = torch.cuda.current_stream().cuda_stream
cs = ctx.cmp_set_dev(t.data_ptr(), N, S, algo=D.CMP_BITSLICE, stream=st)
for _ in range(4):
    cs.update_dev(t.data_ptr(), st)
torch.cuda.synchronize()
buf = np.zeros(8192 * 16, dtype=np.uint64)
f = lib().d2g_debug_rank_trace
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_size_t]
assert f(buf.ctypes.data, buf.size) == 0
tr = buf.reshape(-1, 16).astype(np.int64)[:S]
t0 = tr[:, 0].min()
us = (tr - t0) / 100.0
def stats(x): return f"mean {x.mean():7.2f}  p10 {np.percentile(x, 10):7.2f}  p50 {np.percentile(x, 50):7.2f}  p90 {np.percentile(x, 90):7.2f}  max {x.max():7.2f}"
print(f"{which} N={N}: {S} workgroups; kernel span {us[us > -1e6].max():.1f} us")
print("start of the workgroup        ", stats(us[:, 0]))
if N <= 12288:
    names = ["load values + clear table (0->1)", "claim: LDS compare-and-swap chains (1->2)", "confirm: owners' values fetched (2->3)", "compaction: ranks (3->4)", "ids stored (4->5)"]
    for k, nm in enumerate(names):
        print(f"{nm:44s}", stats(us[:, k + 1] - us[:, k]))
    print("workgroup life (0->5)                       ", stats(us[:, 5] - us[:, 0]))
    first = us[:, 0] < 2.0
    print(f"workgroups started in the first 2 us: {int(first.sum())}; their life:", stats((us[:, 5] - us[:, 0])[first]), "; the others':", stats((us[:, 5] - us[:, 0])[~first]) if (~first).any() else "")
else:
    npass = int(((tr[0] > 0).sum() - 1) // 2)
    for p in range(npass):
        print(f"pass {p}: walk (load, claim, confirm, pending ids)", stats(us[:, 1 + 2 * p] - us[:, 2 * p]))
        print(f"pass {p}: compaction + final ids               ", stats(us[:, 2 + 2 * p] - us[:, 1 + 2 * p]))
    print("workgroup life                              ", stats(us[:, 2 * npass] - us[:, 0]))

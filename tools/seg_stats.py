"""segment (family) sizes the ordering finds on the bench's stated matrix, and how many 16-row blocks / sub-tiles an exact listing would need (tools; GPU box)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dashing2_amd as D
from dashing2_amd import synth
N, S = int(os.environ.get("N", 10000)), 1024
ctx = D.Context(0)
regs = synth.synthetic_registers(N, S, nclusters=max(8, N // 150), seed=20260928)
sig = D.oph_finalize(regs, S, nthreads=32)[0]
t = torch.from_numpy(sig.view(np.int64)).to("cuda:0")
st = torch.cuda.current_stream().cuda_stream
cs = ctx.cmp_set_dev(t.data_ptr(), N, S, algo=D.CMP_BITSLICE, stream=st)
out = torch.empty(N * (N - 1) // 2, dtype=torch.uint32, device="cuda:0")
cs.eqcount_ut_dev(out.data_ptr(), 0, N, st) if hasattr(cs, "eqcount_ut_dev") else None
torch.cuda.synchronize()
pairs, roots = cs.debug_pairs()
u, cnt = np.unique(roots, return_counts=True)
print("nclusters planted", max(8, N // 150), "segments", len(u), "sizes: max", cnt.max(), "hist", {int(k): int(v) for k, v in zip(*np.unique(np.minimum(cnt, 400), return_counts=True))})
print("info", cs.sparse_info(st), "pairs listed", len(pairs))
# exact need: sorted order = by root; for every 16-row block the column span of same-segment pairs
order = np.argsort(roots, kind="stable")
r = roots[order]
start = np.r_[0, np.flatnonzero(r[1:] != r[:-1]) + 1]
end = np.r_[start[1:], N]
segend = np.repeat(end, end - start)          # per sorted position: its segment's end
for W in (64, 128):
    need = 0
    for k0 in range(0, N, 16):
        hi = int(segend[k0:k0 + 16].max())     # columns < hi hold same-segment partners of this block's rows
        lo = k0
        need += (hi - 1) // W - lo // W + 1
    print(f"sub-tiles of 16 x {W} an exact listing needs: {need}")

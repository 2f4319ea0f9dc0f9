"""one long chain (sketch i shares one register with i + 1 only): the worst case of the ordering's root walks (tools; GPU box)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dashing2_amd as D
N, S = int(os.environ.get("N", 12000)), 1024
L = int(os.environ.get("CHAIN", N))
rng = np.random.default_rng(5)
m = rng.random((N, S))
for i in range(N - 1):
    if (i + 1) % L:
        m[i + 1, i % S] = m[i, i % S]
ctx = D.Context(0)
dev = torch.device("cuda", 0)
t = torch.from_numpy(np.ascontiguousarray(m).view(np.int64)).to(dev)
out = torch.empty(N * (N - 1) // 2, dtype=torch.int32, device=dev)
ref = torch.empty_like(out)
st = torch.cuda.current_stream().cuda_stream
cs = ctx.cmp_set_dev(t.data_ptr(), N, S, algo=D.CMP_BITSLICE, stream=st)
cs.eqcount_ut_dev(out.data_ptr(), 0, N, st)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    cs.update_dev(t.data_ptr(), st)
    cs.eqcount_ut_dev(out.data_ptr(), 0, N, st)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
info = cs.sparse_info(st)
dr = ctx.cmp_set_dev(t.data_ptr(), N, S, algo=D.CMP_DIRECT, stream=st)
dr.eqcount_ut_dev(ref.data_ptr(), 0, N, st)
torch.cuda.synchronize()
print(f"N={N} chain length {L}: step {dt * 1e3:.3f} ms; equal to the direct kernel: {bool(torch.equal(out, ref))}; {info}")

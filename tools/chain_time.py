"""corner cases of the sparse-tile ordering against the direct kernel (tools; GPU box): CHAIN=L -- sketch i shares one register with i + 1 only, in
chains of L (L = N: one chain, the worst case of the root walks); PAIRS=1 -- families of two that share everything (N / 2 shared values per column);
FAMILIES=F -- families of F; N, S from the environment"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dashing2_amd as D
N, S = int(os.environ.get("N", 12000)), int(os.environ.get("S", 1024))
L = int(os.environ.get("CHAIN", N))
rng = np.random.default_rng(5)
m = rng.random((N, S))
if os.environ.get("PAIRS"):                                  # sketches 2i and 2i+1 share EVERY register: N / 2 shared values per column (past the
    m[1::2] = m[0:N - 1:2][: m[1::2].shape[0]]               # union pass's LDS table from N = 73 728 on), families of two
    L = 2
elif os.environ.get("FAMILIES"):                             # families of FAMILIES sketches (index j // F) that share ~70 % of their registers
    F = int(os.environ["FAMILIES"])
    base = rng.random(((N + F - 1) // F, S))
    m = np.where(rng.random((N, S)) < 0.7, base[np.arange(N) // F], m)
    L = F
else:
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
print(f"N={N} S={S} family / chain length {L}: step {dt * 1e3:.3f} ms; equal to the direct kernel: {bool(torch.equal(out, ref))}; {info}")

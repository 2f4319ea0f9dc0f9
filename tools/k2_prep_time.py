"""prepare-chain timing (transpose + rank + planes) on BASELINE config 3's shape (tools; GPU box)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dashing2_amd as D
from dashing2_amd import synth

N, S = int(os.environ.get("N", 10000)), 1024
ctx = D.Context(0)
dev = torch.device("cuda", 0)
which = os.environ.get("MATRIX", "stated")
regs = {"stated": lambda: synth.synthetic_registers(N, S, nclusters=max(8, N // 150), seed=20260928),
        "unrelated": lambda: synth.unrelated_registers(N, S), "paired": lambda: synth.paired_registers(N, S)}[which]()
sig = D.oph_finalize(regs, S, nthreads=32)[0] if which == "stated" else regs.view(np.float64)
t = torch.from_numpy(sig.view(np.int64)).to(dev)
st = torch.cuda.current_stream().cuda_stream
sets = []
for _ in range(3):
    sets.append(ctx.cmp_set_dev(t.data_ptr(), N, S, algo=D.CMP_BITSLICE, stream=st))
torch.cuda.synchronize()
ctx.set_timing(True); ctx.kernel_ms("k2prep")
for _ in range(20):
    cs = ctx.cmp_set_dev(t.data_ptr(), N, S, algo=D.CMP_BITSLICE, stream=st)
    del cs
torch.cuda.synchronize()
n, ms, _ = ctx.kernel_ms("k2prep")
print(f"{which} N={N} prep {ms:.4f} ms over {n} prepares")

"""Experiment: replay the N=1 step (memset + transpose + rank + planes + pair kernel) as a captured graph."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dashing2_amd as D
from dashing2_amd import synth
N, S = 10000, 1024
ctx = D.Context(0)
regs = synth.synthetic_registers(N, S, nclusters=N // 150, seed=20260928)
bits = D.oph_finalize(regs, S, nthreads=32)[0].view(np.int64)
dev = torch.device("cuda")
t_in = torch.from_numpy(bits).to(dev)
lut = torch.from_numpy(D.epilogue_lut(S, D.SIMILARITY, 31)).to(dev)
out = torch.empty(N * (N - 1) // 2, dtype=torch.float32, device=dev)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    cs = ctx.cmp_set_dev(t_in.data_ptr(), N, S, algo=D.CMP_BITSLICE, stream=s.cuda_stream)
    def step():
        cs.update_dev(t_in.data_ptr(), s.cuda_stream)
        cs.lut_ut_dev(lut.data_ptr(), out.data_ptr(), 0, N, s.cuda_stream)
    for _ in range(3): step()
    torch.cuda.synchronize()
    ref = out.clone()
    t = time.perf_counter()
    for _ in range(40): step()
    torch.cuda.synchronize()
    print("plain  %.4f ms/step" % ((time.perf_counter() - t) / 40 * 1e3))
    try:
        g = torch.cuda.CUDAGraph()
        out.zero_()
        with torch.cuda.graph(g, stream=s):
            step()
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(40): g.replay()
        torch.cuda.synchronize()
        print("graph  %.4f ms/step  same=%s" % ((time.perf_counter() - t) / 40 * 1e3, bool(torch.equal(out, ref))))
    except Exception as e:
        print("graph capture failed:", type(e).__name__, str(e)[:300])

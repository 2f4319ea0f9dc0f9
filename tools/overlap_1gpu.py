"""Does overlapping prepare(i+1) with the pair kernel(i) help on ONE GPU?  (experiment)"""
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
main = torch.cuda.current_stream()
xs = torch.cuda.Stream()
sets = [ctx.cmp_set_dev(t_in.data_ptr(), N, S, algo=D.CMP_BITSLICE, stream=main.cuda_stream) for _ in range(2)]
lut = torch.from_numpy(D.epilogue_lut(S, D.SIMILARITY, 31)).to(dev)
out = torch.empty(N * (N - 1) // 2, dtype=torch.float32, device=dev)
def plain(K):
    for _ in range(K):
        sets[0].update_dev(t_in.data_ptr(), main.cuda_stream)
        sets[0].lut_ut_dev(lut.data_ptr(), out.data_ptr(), 0, N, main.cuda_stream)
xdone = [torch.cuda.Event(), torch.cuda.Event()]
pdone = [None, None]
def piped(K):
    for i in range(K):
        b = i & 1
        if pdone[b] is not None: xs.wait_event(pdone[b])
        sets[b].update_dev(t_in.data_ptr(), xs.cuda_stream)
        xdone[b].record(xs)
        main.wait_event(xdone[b])
        sets[b].lut_ut_dev(lut.data_ptr(), out.data_ptr(), 0, N, main.cuda_stream)
        e = torch.cuda.Event(); e.record(main); pdone[b] = e
for name, fn in (("plain", plain), ("piped", piped), ("plain", plain), ("piped", piped)):
    fn(5); torch.cuda.synchronize(); t = time.perf_counter(); fn(40); torch.cuda.synchronize()
    print(name, "%.4f ms/step" % ((time.perf_counter() - t) / 40 * 1e3))

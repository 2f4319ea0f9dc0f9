"""first step of a set without history vs. the steady step (tools; GPU box): MATRIX / C / N as tools/k2_time.py; under kstats.sh the per-kernel table shows the sample"""
import os, sys, time
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
        "unrelated": lambda: synth.unrelated_registers(N, S), "paired": lambda: synth.paired_registers(N, S), "skewed": lambda: synth.skewed_registers(N, S),
        "noise": lambda: synth.add_chance_collisions(synth.synthetic_registers(N, S, nclusters=max(8, N // 150), seed=20260928), int(os.environ.get("C", 10)), seed=20260929)}[which]()
sig = D.oph_finalize(regs, S, nthreads=32)[0] if which in ("stated", "noise") else regs.view(np.float64)
t = torch.from_numpy(sig.view(np.int64)).to(dev)
lut = torch.from_numpy(D.epilogue_lut(S, D.SIMILARITY, 31)).to(dev)
out = torch.empty(N * (N - 1) // 2, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
cs = ctx.cmp_set_dev(t.data_ptr(), N, S, algo=D.CMP_BITSLICE, stream=st)
def step():
    cs.announce_ut_dev(out.data_ptr(), 0, N, lut_dev_ptr=lut.data_ptr()); cs.update_dev(t.data_ptr(), st); cs.lut_ut_dev(lut.data_ptr(), out.data_ptr(), 0, N, st)
def timed(forget):
    if forget: cs.forget()
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
for _ in range(3): step()
R = int(os.environ.get("REPS", 20))
first = sorted(timed(True) for _ in range(R)); 
for _ in range(3): step()
steady = sorted(timed(False) for _ in range(R))
print(f"{which} C={os.environ.get('C','-')} N={N} sparse={os.environ.get('D2G_BS_SPARSE','1')}: first step median {first[R//2]:.4f} ms (min {first[0]:.4f}), steady single step median {steady[R//2]:.4f} ms (min {steady[0]:.4f}); {cs.sparse_info(st)}")

"""Experiment: two independent K3 calls at once (two contexts, two streams, two host threads) against the same two calls
one after the other -- how much of the chain's memory-bound passes hide under the other call's counting pass?"""
import sys, time, threading, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dashing2_amd as d2
ng, L, k, S = 125, 5_000_000, 21, 2048
nb = (L + 3) // 4
stride = (nb + 63) // 64 * 64
run_start = np.arange(ng, dtype=np.uint64) * np.uint64(stride * 4)
run_len = np.full(ng, L, dtype=np.uint32)
off = np.arange(ng + 1, dtype=np.uint64)
work = []
for i in range(2):
    ctx = d2.Context(0)
    st = torch.cuda.Stream()
    packed = torch.randint(0, 256, (ng * stride + 64,), dtype=torch.uint8, device="cuda")
    plan = ctx.oph_plan(run_start, run_len, off, k)
    sig = torch.empty((ng, S), dtype=torch.float64, device="cuda")
    tw = torch.empty((ng,), dtype=torch.float64, device="cuda")
    work.append((ctx, st, packed, plan, sig, tw))
torch.cuda.synchronize()

def call(w, reps):
    ctx, st, packed, plan, sig, tw = w
    for _ in range(reps):
        ctx.bmh_sketch_dev(plan, packed.data_ptr(), S, sig.data_ptr(), tw.data_ptr(), stream=st.cuda_stream)

for w in work:
    call(w, 2)
torch.cuda.synchronize()
reps = 6
t = time.perf_counter()
for w in work:
    call(w, reps)
torch.cuda.synchronize()
seq = time.perf_counter() - t
t = time.perf_counter()
th = [threading.Thread(target=call, args=(w, reps)) for w in work]
for x in th: x.start()
for x in th: x.join()
torch.cuda.synchronize()
par = time.perf_counter() - t
print(f"2 x {reps} calls of {ng} genomes: one after the other {seq*1e3/reps/2:.2f} ms per call, two at once {par*1e3/reps/2:.2f} ms per call ({seq/par:.2f}x)")

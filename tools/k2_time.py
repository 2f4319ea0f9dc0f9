"""pair-kernel timing on BASELINE config 3's matrix (tools; GPU box): prints kernel ms for the current build/env"""
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
        "unrelated": lambda: synth.unrelated_registers(N, S), "paired": lambda: synth.paired_registers(N, S),
        "skewed": lambda: synth.skewed_registers(N, S),
        # the stated families + C chance collisions per sketch with random strangers
        "noise": lambda: synth.add_chance_collisions(synth.synthetic_registers(N, S, nclusters=max(8, N // 150), seed=20260928), int(os.environ.get("C", 10)), seed=20260929)}[which]()
sig = D.oph_finalize(regs, S, nthreads=32)[0] if which in ("stated", "noise") else regs.view(np.float64)
t = torch.from_numpy(sig.view(np.int64)).to(dev)
lut = torch.from_numpy(D.epilogue_lut(S, D.SIMILARITY, 31)).to(dev)
out = torch.empty(N * (N - 1) // 2, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
cs = ctx.cmp_set_dev(t.data_ptr(), N, S, algo=D.CMP_BITSLICE, stream=st)
PREFILL = os.environ.get("PREFILL", "0") == "1"       # the launch's fill on a second stream, beside the prepare chain (d2g_cmp_ut_prefill_dev)
ANNOUNCE = os.environ.get("ANNOUNCE", "1") == "1"     # the launch's output announced ahead of the prepare: its small kernels carry the fill (d2g_cmp_ut_announce_dev)
main = torch.cuda.current_stream()
side = torch.cuda.Stream() if PREFILL else None
ev_done, ev_fill = torch.cuda.Event(), torch.cuda.Event()


def step():
    if PREFILL:
        side.wait_event(ev_done)                       # the last step's launch has finished with `out`
        cs.prefill_ut_dev(out.data_ptr(), 0, N, lut_dev_ptr=lut.data_ptr(), stream=side.cuda_stream)
        ev_fill.record(side)
    if ANNOUNCE and not PREFILL:
        cs.announce_ut_dev(out.data_ptr(), 0, N, lut_dev_ptr=lut.data_ptr())
    cs.update_dev(t.data_ptr(), st)
    if PREFILL:
        main.wait_event(ev_fill)
    cs.lut_ut_dev(lut.data_ptr(), out.data_ptr(), 0, N, st)
    if PREFILL:
        ev_done.record(main)


for _ in range(3):
    step()
torch.cuda.synchronize()
ctx.set_timing(True); ctx.kernel_ms("k2")
t0 = time.perf_counter()
STEPS = int(os.environ.get("STEPS", 20))
for _ in range(STEPS):
    step()
torch.cuda.synchronize()
step_ms = (time.perf_counter() - t0) / STEPS * 1e3
n, ms, _ = ctx.kernel_ms("k2")
print(f"step (prepare + compare) {step_ms:.4f} ms announce={int(ANNOUNCE)} ride={os.environ.get('D2G_SP_RIDE','63')}; sparse: {cs.sparse_info(st)}")
print(f"{which} C={os.environ.get('C','-')} N={N} D2G_BS_EXP={os.environ.get('D2G_BS_EXP','0')} planes={cs.planes(st)} pair kernel {ms:.4f} ms over {n} launches")

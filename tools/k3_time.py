"""Time K3 (d2g_bmh_sketch_dev: k-mer counting + BagMinHash) on synthetic packed bases resident in
HBM: python tools/k3_time.py [ngenomes] [len] [k] [S] [reps]"""
import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dashing2_amd as d2
ng = int(sys.argv[1]) if len(sys.argv) > 1 else 100
L = int(float(sys.argv[2])) if len(sys.argv) > 2 else 5_000_000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 21
S = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
ctx = d2.Context(0)
ctx.set_timing(True)
nb = (L + 3) // 4
stride = (nb + 63) // 64 * 64
packed = torch.randint(0, 256, (ng * stride + 64,), dtype=torch.uint8, device="cuda")
run_start = np.arange(ng, dtype=np.uint64) * np.uint64(stride * 4)
run_len = np.full(ng, L, dtype=np.uint32)
off = np.arange(ng + 1, dtype=np.uint64)
plan = ctx.oph_plan(run_start, run_len, off, k)
sig = torch.empty((ng, S), dtype=torch.float64, device="cuda")
tw = torch.empty((ng,), dtype=torch.float64, device="cuda")
for i in range(reps + 1):
    if i == 1: ctx.kernel_ms("k3", reset=True)
    t = time.time()
    ctx.bmh_sketch_dev(plan, packed.data_ptr(), S, sig.data_ptr(), tw.data_ptr())
    torch.cuda.synchronize()
    wall = time.time() - t
cnt, avg, last = ctx.kernel_ms("k3")
print(f"K3 ng={ng} L={L} k={k} S={S}: kernels {avg:.2f} ms (wall {wall*1e3:.2f} ms)  {ng*L/avg/1e6:.2f} Gbase/s  "
      f"tw0={tw[0].item():.0f} finite={bool(torch.isfinite(sig).all())} mean_h={sig.mean().item():.4g}")

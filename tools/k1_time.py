"""Time K1 (d2g_oph_sketch_dev) on synthetic packed bases: python tools/k1_time.py [ngenomes] [len] [k] [S]"""
import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dashing2_amd as d2
ng = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
L = int(float(sys.argv[2])) if len(sys.argv) > 2 else 5_000_000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 31
S = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
ctx = d2.Context(0)
ctx.set_timing(True)
nb = (L + 3) // 4
stride = (nb + 63) // 64 * 64
packed = torch.randint(0, 256, (ng * stride + 64,), dtype=torch.uint8, device="cuda")
run_start = np.arange(ng, dtype=np.uint64) * np.uint64(stride * 4)
run_len = np.full(ng, L, dtype=np.uint32)
off = np.arange(ng + 1, dtype=np.uint64)
plan = ctx.oph_plan(run_start, run_len, off, k)
regs = torch.empty((ng, d2.oph_m(S)), dtype=torch.int64, device="cuda")
for i in range(6):
    if i == 2: ctx.kernel_ms("k1", reset=True)
    ctx.oph_sketch_dev(plan, packed.data_ptr(), S, regs.data_ptr())
    torch.cuda.synchronize()
cnt, avg, last = ctx.kernel_ms("k1")
print(f"K1 ng={ng} L={L} k={k} S={S}: {avg:.3f} ms  {ng*L/avg/1e6:.1f} Gbase/s  chk={int(regs.sum().item()) & 0xffffffff:08x}")

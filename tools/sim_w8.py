"""Per-rank compute cost of the 8-GPU bench step, measured on ONE GPU (no NCCL): pack, prepare of S/8 columns,
export, pair kernel over this rank's rows of the full operand."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dashing2_amd as D
from dashing2_amd import synth
N, S, W = 28288, 1024, 8
ctx = D.Context(0)
regs = synth.synthetic_registers(N, S, nclusters=N // 150, seed=8)
bits = D.oph_finalize(regs, S, nthreads=32)[0].view(np.int64)
dev = torch.device("cuda")
full_t = torch.from_numpy(bits).to(dev)
n_loc, S_loc = N // W, S // W
stream = torch.cuda.current_stream().cuda_stream
rows = full_t[:n_loc].clone()
send = torch.empty((W, n_loc, S_loc), dtype=torch.int64, device=dev)
recv = full_t[:, :S_loc].contiguous()
gw, ng = D.operand_layout(N, S)
gw_l, ng_l = D.operand_layout(N, S_loc)
planes_loc = torch.empty(ng_l * gw, dtype=torch.int32, device=dev)
meta_loc = torch.empty(ng_l, dtype=torch.int32, device=dev)
local = ctx.cmp_set_dev(recv.data_ptr(), N, S_loc, algo=D.CMP_BITSLICE, stream=stream)
ref = ctx.cmp_set_dev(full_t.data_ptr(), N, S, algo=D.CMP_BITSLICE, stream=stream)
b = D.ut_partition(N, W)
lut = torch.from_numpy(D.epilogue_lut(S, D.SIMILARITY, 31)).to(dev)
def tm(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
print("pack          %.3f ms" % tm(lambda: ctx.pack_column_slices_dev(rows.data_ptr(), n_loc, S, W, send.data_ptr(), stream)))
print("prepare S/8   %.3f ms" % tm(lambda: local.update_dev(recv.data_ptr(), stream)))
print("export        %.3f ms" % tm(lambda: local.export_operand_dev(planes_loc.data_ptr(), meta_loc.data_ptr(), stream)))
for r in (0, 3, 7):
    cnt = D.ut_count(N, b[r], b[r + 1])
    out = torch.empty(cnt, dtype=torch.float32, device=dev)
    print("pair kernel rank %d (%d pairs) %.3f ms" % (r, cnt, tm(lambda: ref.lut_ut_dev(lut.data_ptr(), out.data_ptr(), b[r], b[r + 1], stream))))
print("prepare full (1 GPU would) %.3f ms" % tm(lambda: ref.update_dev(full_t.data_ptr(), stream)))

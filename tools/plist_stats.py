"""Who is in the pair list?  (tools; GPU box)  Config-3's stated matrix (or MATRIX=noise C=c): the families the prepare found against the planted
clusters (sketch j belongs to cluster j % nclusters), and the sketches behind the list's entries: their planted share fraction, whether their
segment is their cluster's main segment, a fragment or a singleton."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dashing2_amd as D
from dashing2_amd import synth

N, S = int(os.environ.get("N", 10000)), 1024
ncl = max(8, N // 150)
regs = synth.synthetic_registers(N, S, nclusters=ncl, seed=20260928)
c = int(os.environ.get("C", 0))
if c:
    regs = synth.add_chance_collisions(regs, c, seed=20260929)
# the planted share fraction of every sketch, recomputed the way the generator drew it
rng = np.random.default_rng(20260928)
# (draws in synthetic_registers: parents, regs, f, take -- replay to get f)
def draw(shape):
    u = rng.exponential(1.0 / 4883, size=shape); return u
draw((ncl, S)); draw((N, S)); f = rng.uniform(0.0, 0.98, size=N)
sig = D.oph_finalize(regs, S, nthreads=32)[0]
ctx = D.Context(0)
dev = torch.device("cuda", 0)
t = torch.from_numpy(sig.view(np.int64)).to(dev)
st = torch.cuda.current_stream().cuda_stream
cs = ctx.cmp_set_dev(t.data_ptr(), N, S, algo=D.CMP_BITSLICE, stream=st)
out = torch.empty(N * (N - 1) // 2, dtype=torch.int32, device=dev)
cs.eqcount_ut_dev(out.data_ptr(), 0, N, st)
print("sparse:", cs.sparse_info(st))
pairs, roots = cs.debug_pairs(stream=st)
cl = np.arange(N) % ncl
# segments: size per root; main segment of a cluster = the root that holds most of its members
uniq, inv, cnt = np.unique(roots, return_inverse=True, return_counts=True)
print(f"{len(uniq)} segments for {ncl} planted clusters; sizes: max {cnt.max()}, singletons {int((cnt == 1).sum())}, of 2-20: {int(((cnt >= 2) & (cnt <= 20)).sum())}, > 200: {int((cnt > 200).sum())}")
main = {}
for k in range(ncl):
    r, n = np.unique(roots[cl == k], return_counts=True)
    main[k] = r[n.argmax()]
in_main = np.array([roots[j] == main[cl[j]] for j in range(N)])
mixed_seg = sum(1 for r in uniq if len(np.unique(cl[roots == r])) > 1)
print(f"sketches outside their cluster's main segment: {int((~in_main).sum())} (share fraction f: " + ", ".join(f"{x:.3f}" for x in np.sort(f[~in_main])[:40]) + (" ..." if (~in_main).sum() > 40 else "") + f"); segments that hold two or more clusters: {mixed_seg}")
print(f"pair list: {len(pairs)} entries")
if len(pairs):
    who = np.bincount(pairs.reshape(-1), minlength=N)
    top = np.argsort(-who)[:20]
    print("sketches with the most entries: " + "; ".join(f"j={j} n={who[j]} f={f[j]:.3f} segsize={cnt[inv[j]]} in_main={bool(in_main[j])}" for j in top))
    same_cluster = (cl[pairs[:, 0]] == cl[pairs[:, 1]]).mean()
    print(f"entries between sketches of ONE planted cluster: {same_cluster:.3f}")

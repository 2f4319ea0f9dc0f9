"""One large input through K1 and K3 (buckets far above one table round): exact weights, finite registers,
weighted-Jaccard self-consistency.  usage: big_genome_check.py [Mbp=100]"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dashing2_amd as D
from dashing2_amd import synth
L = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 100_000_000
ctx = D.Context(0)
g = synth.random_genome(5, L)
sp = D.SeqPack(21)
sp.add_fastx(synth.fasta_bytes("big", g))
sp.add_fastx(synth.fasta_bytes("big", g) + synth.fasta_bytes("half", g[:L // 2]))
t = time.time(); regs = ctx.oph_sketch_seqpack(sp, 1024); t1 = time.time() - t
assert (regs[0] != np.uint64(0xFFFFFFFFFFFFFFFF)).all()
t = time.time(); sig, tw = ctx.bmh_sketch_seqpack(sp, 1024); t3 = time.time() - t
nk = L - 20
assert tw[0] == nk and tw[1] == nk + L // 2 - 20, tw
assert np.isfinite(sig).all() and (sig[1] <= sig[0]).all()
est = (sig[0] == sig[1]).mean()
assert abs(est - 2 / 3) < 5 * np.sqrt((2 / 9) / 1024), est
nd = ctx.kmer_distinct_seqpack(sp)
print(f"L={L}: K1 {t1:.2f}s  K3 {t3:.2f}s (host arrays incl.)  distinct={nd.tolist()}  J_est={est:.4f}  OK")

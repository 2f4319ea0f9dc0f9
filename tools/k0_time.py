"""K0 (device FASTA parser) timing on config 2's inputs (tools; GPU box): N genomes x 5 Mbp of 80-column FASTA, staged in
page-locked memory, ingested in one call; per-kernel times with tools/kstats.sh.  usage: k0_time.py [N=200] [reps=5]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dashing2_amd as D
from dashing2_amd import synth
from concurrent.futures import ThreadPoolExecutor

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
L, k, S = 5_000_000, 31, 1024
ctx = D.Context(0)
with ThreadPoolExecutor(16) as ex:
    fastas = list(ex.map(lambda i: synth.fasta_bytes_fast("g%05d" % i, synth.random_genome(i, L)), range(N)))
lens = np.array([len(f) for f in fastas], np.uint64)
offs = np.zeros(N, np.uint64)
pos = 0
for i in range(N):
    offs[i] = pos
    pos += (len(fastas[i]) + 15) // 16 * 16
pin = D.PinnedArray(ctx, pos + 64)
for i, f in enumerate(fastas):
    pin.array[int(offs[i]):int(offs[i]) + len(f)] = np.frombuffer(f, np.uint8)
gfo = np.arange(N + 1, dtype=np.uint64)
sk = ctx.sketcher()
sk.ingest_raw(pin.array, pos, offs, lens, gfo, k)
runs = sk.ingested_runs(N)
regs = sk.run_ingested(runs, S)
ctx.set_timing(D.TIME_K0)
ctx.kernel_ms("k0")
t0 = time.perf_counter()
for _ in range(reps):
    sk.ingest_raw(pin.array, pos, offs, lens, gfo, k)
dt = (time.perf_counter() - t0) / reps
n, ms, _ = ctx.kernel_ms("k0")
print(f"K0: {N} x {L} bp ({pos / 1e9:.2f} GB of FASTA): ingest call {dt * 1e3:.2f} ms (H2D + kernels + run table), kernels {ms:.3f} ms "
      f"= {pos / (ms * 1e-3) / 1e9:.0f} GB/s of input, {N * L / (ms * 1e-3):.3e} bases/s")
sp = D.SeqPack(k)
sp.add_fastx(fastas[0])
assert np.array_equal(sk.run(sp, S)[0], regs[0])

#!/usr/bin/env python3
"""End-to-end timing of the drop-in CLI on synthetic data (GPU box):
   sketch: G genomes x L bp FASTA on local disk -> stacked sketch file
   cmp   : N presketched synthetic sketches -> binary / PHYLIP matrix
usage: e2e_cli.py [--genomes 200] [--len 5000000] [--sketches 10000] [--threads 64] [--workdir /tmp/d2e2e]"""
import argparse
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
EXE = os.path.join(ROOT, "dashing2_amd", "bin", "dashing2")


def run(args):
    t0 = time.perf_counter()
    r = subprocess.run([EXE] + args, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    if r.returncode:
        print(r.stderr[-2000:])
        raise SystemExit(1)
    info = [l for l in r.stderr.splitlines() if l.startswith("[d2g]")]
    return dt, info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genomes", type=int, default=200)
    ap.add_argument("--len", type=int, default=5_000_000)
    ap.add_argument("--sketches", type=int, default=10000)
    ap.add_argument("--threads", type=int, default=min(64, os.cpu_count() or 1))
    ap.add_argument("--workdir", default="/tmp/d2e2e")
    ap.add_argument("--big-sketches", type=int, default=0, help="also run cmp on this many presketched sketches (config 4: 50000 -> 5 GB of output)")
    ap.add_argument("--no-sketch", action="store_true")
    a = ap.parse_args()
    import dashing2_amd as D
    from dashing2_amd import synth
    os.makedirs(a.workdir, exist_ok=True)
    t0 = time.perf_counter()
    paths = []
    for i in range(a.genomes):
        p = os.path.join(a.workdir, f"g{i:05d}.fa")
        if not os.path.exists(p):
            synth.write_fasta(p, f"g{i:05d}", synth.random_genome(i, a.len))
        paths.append(p)
    lst = os.path.join(a.workdir, "files.txt")
    open(lst, "w").write("\n".join(paths) + "\n")
    print(f"generated {a.genomes} x {a.len} bp FASTA in {time.perf_counter() - t0:.1f}s")
    out = os.path.join(a.workdir, "stack.bin")
    bases = a.genomes * a.len
    for rep in range(0 if a.no_sketch else 2):
        dt, info = run(["sketch", "-v", "-k", "31", "-S", "1024", "-p", str(a.threads), "-F", lst, "-o", out])
        print(f"sketch run {rep}: {dt:.2f}s wall -> {bases / dt:.3e} bases/s end-to-end (FASTA on disk -> stacked sketches)")
        for l in info:
            print("   ", l)
    # BASELINE configs[1] as stated: the genomes sketched AND the full all-pairs PHYLIP matrix, one command
    for rep in range(0 if a.no_sketch else 2):
        ph = os.path.join(a.workdir, "all.phylip")
        dt, info = run(["sketch", "-v", "-k", "31", "-S", "1024", "-p", str(a.threads), "-F", lst, "-o", out, "--cmpout", ph, "--phylip"])
        npairs = a.genomes * (a.genomes - 1) // 2
        print(f"sketch + all-pairs PHYLIP run {rep}: {dt:.2f}s wall -> {bases / dt:.3e} bases/s, {npairs} pairs, {os.path.getsize(ph) / 1e6:.1f} MB of PHYLIP "
              f"(BASELINE configs[1]: FASTA on disk -> PHYLIP matrix)")
        for l in info:
            print("   ", l)
    for rep in range(0 if a.no_sketch else 2):
        dt, info = run(["sketch", "--multiset", "-v", "-k", "21", "-S", "2048", "-p", str(a.threads), "-F", lst, "-o", out + ".bmh"])
        print(f"sketch --multiset run {rep}: {dt:.2f}s wall -> {bases / dt:.3e} bases/s end-to-end (FASTA on disk -> stacked BagMinHash sketches)")
        for l in info:
            print("   ", l)
    # cmp on synthetic presketched collection
    N, S = a.sketches, 1024
    regs = synth.synthetic_registers(N, S, nclusters=max(8, N // 150), seed=20260928)
    sigs, cards = D.oph_finalize(regs, S, nthreads=min(os.cpu_count() or 1, 16))
    st = os.path.join(a.workdir, "syn.bin")
    with open(st, "wb") as f:
        np.array([N, S], np.uint64).tofile(f)
        cards.tofile(f)
        sigs.tofile(f)
    pairs = N * (N - 1) // 2
    for name, flags in [("binary", ["--binary-output"]), ("phylip", ["--phylip"]), ("binary mash", ["--binary-output", "--distance"]),
                        ("binary containment (host x87 epilogue)", ["--binary-output", "--containment"])]:
        o = os.path.join(a.workdir, "dist.out")
        dt, info = run(["cmp", "-v", "--presketched", "-k", "31", "-p", str(a.threads), "--cmpout", o] + flags + [st])
        print(f"cmp {name}: {dt:.2f}s wall -> {pairs / dt:.3e} pairs/s end-to-end ({os.path.getsize(o) / 1e6:.0f} MB out)")
        for l in info:
            print("   ", l)

    if a.big_sketches:
        N = a.big_sketches
        regs = synth.synthetic_registers(N, S, nclusters=max(8, N // 150), seed=20260929)
        sigs, cards = D.oph_finalize(regs, S, nthreads=os.cpu_count() or 1)
        st = os.path.join(a.workdir, "syn_big.bin")
        with open(st, "wb") as f:
            np.array([N, S], np.uint64).tofile(f)
            cards.tofile(f)
            sigs.tofile(f)
        del regs, sigs
        pairs = N * (N - 1) // 2
        for name, flags, env in [("binary", ["--binary-output"], {}), ("binary, all GPUs (D2G_DEVICES=all)", ["--binary-output"], {"D2G_DEVICES": "all"})]:
            o = os.path.join(a.workdir, "dist_big.out")
            os.environ.update(env)
            dt, info = run(["cmp", "-v", "--presketched", "-k", "31", "-p", str(a.threads), "--cmpout", o] + flags + [st])
            for kk in env:
                os.environ.pop(kk, None)
            print(f"cmp {N} sketches {name}: {dt:.2f}s wall -> {pairs / dt:.3e} pairs/s end-to-end ({os.path.getsize(o) / 1e6:.0f} MB out)")
            for l in info:
                print("   ", l)
            os.remove(o)


if __name__ == "__main__":
    main()
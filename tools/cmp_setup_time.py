#!/usr/bin/env python3
"""Where does the fixed set-up cost of `dashing2 cmp` go?  Fresh process, config-3-sized operand (10000 x 1024 u64 = 82 MB):
times context creation, device allocation, pageable vs page-locked H2D, the first and second d2g_cmp_set_create_dev (code-object
load + workspace allocations vs steady state) and the first pair-kernel launch.  usage: cmp_setup_time.py [N S]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("D2G_NO_TORCH_PRELOAD", "1")
import dashing2_amd as D  # noqa: E402
from dashing2_amd import synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
regs = synth.synthetic_registers(N, S, nclusters=max(8, N // 150), seed=1)
sig, cards = D.oph_finalize(regs, S, nthreads=8)
bits = np.ascontiguousarray(sig.view(np.uint64))
L = D.lib()
T = {}


def tick(name, t0):
    T[name] = (time.perf_counter() - t0) * 1e3


t0 = time.perf_counter(); ctx = D.Context(0); tick("ctx_create", t0)
t0 = time.perf_counter(); p = ctx.malloc(bits.nbytes); tick("malloc_82MB", t0)
t0 = time.perf_counter(); ctx.h2d(p, bits); ctx.sync(); tick("h2d_pageable", t0)
t0 = time.perf_counter(); ctx.h2d(p, bits); ctx.sync(); tick("h2d_pageable_again", t0)
t0 = time.perf_counter(); pin = D.PinnedArray(ctx, bits.nbytes); tick("malloc_host_82MB", t0)
t0 = time.perf_counter(); pin.array[:] = bits.view(np.uint8).reshape(-1); tick("memcpy_into_pinned", t0)
t0 = time.perf_counter(); ctx.h2d(p, pin.array); ctx.sync(); tick("h2d_pinned", t0)
t0 = time.perf_counter(); cs = ctx.cmp_set_dev(p, N, S); ctx.sync(); tick("set_create_dev_first", t0)
t0 = time.perf_counter(); cs.update_dev(p); ctx.sync(); tick("set_update_dev", t0)
t0 = time.perf_counter(); cs2 = ctx.cmp_set_dev(p, N, S); ctx.sync(); tick("set_create_dev_second", t0)
lut = D.epilogue_lut(S)
pl = ctx.malloc(lut.nbytes); ctx.h2d(pl, lut)
npairs = N * (N - 1) // 2
t0 = time.perf_counter(); po = ctx.malloc(npairs * 4); tick("malloc_out", t0)
t0 = time.perf_counter(); cs.lut_ut_dev(pl, po, 0, N); ctx.sync(); tick("pair_first", t0)
t0 = time.perf_counter(); cs.lut_ut_dev(pl, po, 0, N); ctx.sync(); tick("pair_second", t0)
t0 = time.perf_counter(); cs3 = ctx.cmp_set(bits); tick("set_create_hostptr_warm", t0)
for k, v in T.items():
    print(f"{k:28s} {v:9.3f} ms")

#!/usr/bin/env python3
"""variants of the first operations after context creation (each in a fresh process): usage cmp_setup_time2.py A|B|C|D"""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("D2G_NO_TORCH_PRELOAD", "1")
import dashing2_amd as D
v = sys.argv[1]
big = np.ones(82 * 1000 * 1000 // 8, np.uint64)
small = np.ones(512, np.uint64)
T = []
def tick(name, t0): T.append((name, (time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter(); ctx = D.Context(0); tick("ctx_create", t0)
p = ctx.malloc(big.nbytes)
if v == "A":      # pinned first
    t0 = time.perf_counter(); pin = D.PinnedArray(ctx, 16 << 20); tick("malloc_host_16MiB", t0)
    t0 = time.perf_counter(); ctx.h2d(p, pin.array); ctx.sync(); tick("h2d_pinned_16MiB_first", t0)
    t0 = time.perf_counter(); ctx.h2d(p, pin.array); ctx.sync(); tick("h2d_pinned_16MiB_again", t0)
    t0 = time.perf_counter(); ctx.h2d(p, big); ctx.sync(); tick("h2d_pageable_82MB_after", t0)
    t0 = time.perf_counter(); ctx.d2h(pin.array, p); ctx.sync(); tick("d2h_pinned_16MiB_first", t0)
elif v == "B":    # tiny pageable copy first
    t0 = time.perf_counter(); ctx.h2d(p, small); ctx.sync(); tick("h2d_pageable_4KB_first", t0)
    t0 = time.perf_counter(); ctx.h2d(p, big); ctx.sync(); tick("h2d_pageable_82MB_second", t0)
    t0 = time.perf_counter(); ctx.h2d(p, big); ctx.sync(); tick("h2d_pageable_82MB_third", t0)
    out = np.empty(16 << 17, np.uint64)
    t0 = time.perf_counter(); ctx.d2h(out, p); ctx.sync(); tick("d2h_pageable_16MiB_first", t0)
    t0 = time.perf_counter(); ctx.d2h(out, p); ctx.sync(); tick("d2h_pageable_16MiB_again", t0)
elif v == "C":    # 1 MB pageable copy first, then sizes of hipHostMalloc
    mid = np.ones(1 << 17, np.uint64)
    t0 = time.perf_counter(); ctx.h2d(p, mid); ctx.sync(); tick("h2d_pageable_1MB_first", t0)
    t0 = time.perf_counter(); ctx.h2d(p, big); ctx.sync(); tick("h2d_pageable_82MB_second", t0)
    for mb in (4, 16, 64, 64):
        t0 = time.perf_counter(); pin = D.PinnedArray(ctx, mb << 20); tick(f"malloc_host_{mb}MiB", t0)
elif v == "D":    # pageable H2D of the big buffer on the main thread while another thread page-locks 3 x 64 MiB
    ctx2 = ctx
    def side():
        t0 = time.perf_counter()
        pins = [D.PinnedArray(ctx2, 64 << 20) for _ in range(3)]
        tick("side: 3 x malloc_host_64MiB", t0)
    th = threading.Thread(target=side); th.start()
    t0 = time.perf_counter(); ctx.h2d(p, big); ctx.sync(); tick("main: h2d_pageable_82MB_first (concurrent)", t0)
    th.join()
for k, x in T: print(f"{v}  {k:44s} {x:9.3f} ms")

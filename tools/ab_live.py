#!/usr/bin/env python3
"""Overlapped HPLOC schedule (k_hploc_live beside the tile kernel) against the classic one (k_hploc_ext behind it): tree checksums over repeated builds,
wall-clock per build of back-to-back builds, stage times.  python tools/ab_live.py [N=10000000] [uniform|sponza|bunny] [REPS=50] [modes=block,live]"""
import os, sys, time
import numpy as np
import torch
torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bvh_pkg
pkg = bvh_pkg.load(); ctx = pkg.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "uniform"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
modes = (sys.argv[4] if len(sys.argv) > 4 else "block,live").split(",")
tris = pkg.meshgen.uniform(n, 1) if kind == "uniform" else pkg.meshgen.sponza_like(n, 3) if kind == "sponza" else pkg.meshgen.bunny_like(n, 2)
n = len(tris)
d_tris = torch.from_numpy(tris.view(np.uint8).reshape(-1)).cuda()
ctx.set_option("hploc", "async")
ref = pkg.HPLOC().build(ctx, d_tris, on_device=True, n=n).checksum()
for mode in modes:
    ctx.set_option("hploc", mode)
    b = pkg.HPLOC()
    bad = 0
    for _ in range(6):
        b.build(ctx, d_tris, on_device=True, n=n)
        bad += b.checksum() != ref
    ctx.set_profiling(0)
    for _ in range(5): b.build(ctx, d_tris, on_device=True, n=n)
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): b.build(ctx, d_tris, on_device=True, n=n)
    ctx.synchronize(); wall = (time.perf_counter() - t0) / reps * 1e3
    ctx.set_profiling(1)
    st = []
    for _ in range(10):
        b.build(ctx, d_tris, on_device=True, n=n); t = b.timings; st.append((t.ms_extents, t.ms_morton, t.ms_sort, t.ms_build))
    st = np.median(np.array(st), axis=0)
    ctx.set_profiling(0)
    extra = ""
    if mode == "live":
        try:
            extra = f"  [at tiles-done: items {ctx.get_option(1000 + 2047)} tickets {ctx.get_option(1000 + 2048)}]"
        except Exception:
            pass
    print(f"{kind} n={n} mode={mode}: checksum {'OK' if bad == 0 else f'MISMATCH x{bad}'}  wall {wall:.4f} ms/build  stages E {st[0]:.4f} M {st[1]:.4f} S {st[2]:.4f} B {st[3]:.4f}{extra}", flush=True)

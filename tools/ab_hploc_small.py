#!/usr/bin/env python3
"""HPLOC at the sizes where the one-launch kernel runs (< 800 k) and at 2 M / 10 M with the one-launch kernel forced: wall per build, kernel times, checksum.  python tools/ab_hploc_small.py"""
import os, sys, time
import numpy as np, torch
torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bvh_pkg
pkg = bvh_pkg.load(); ctx = pkg.Context(0)
for n, kind, mode in ((262144, "sponza", "auto"), (50000, "uniform", "auto"), (700000, "uniform", "auto"), (2_000_000, "uniform", "async"), (10_000_000, "uniform", "async")):
    tris = pkg.meshgen.uniform(n, 1) if kind == "uniform" else pkg.meshgen.sponza_like(n, 3)
    n = len(tris)
    d = torch.from_numpy(tris.view(np.uint8).reshape(-1)).cuda()
    ctx.set_option("hploc", mode)
    b = pkg.HPLOC()
    for _ in range(5): b.build(ctx, d, on_device=True, n=n)
    chk = b.checksum()
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(50): b.build(ctx, d, on_device=True, n=n)
    ctx.synchronize(); wall = (time.perf_counter() - t0) / 50 * 1e3
    ctx.set_profiling(2)
    for _ in range(20): b.build(ctx, d, on_device=True, n=n)
    kt = ctx.kernel_times(); ctx.set_profiling(0)
    print(f"{kind} {n} HPLOC {mode} wall {wall:.4f} ms | " + "  ".join(f"{k} {v[0] / 20:.4f}" for k, v in kt.items() if k.startswith("k_hploc")) + f"  checksum {chk:016x}", flush=True)
ctx.set_option("hploc", "auto")

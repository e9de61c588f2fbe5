#!/usr/bin/env python3
"""PLOC++: un-profiled wall time per build, per-kernel event times and the checksum at the config sizes (library from BVH_MI355X_LIB).  python tools/ab_ploc_wall.py"""
import os, sys, time
import numpy as np, torch
torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bvh_pkg
pkg = bvh_pkg.load(); ctx = pkg.Context(0)
for n, kind in ((262144, "sponza"), (50000, "uniform"), (2_000_000, "uniform"), (10_000_000, "uniform")):
    tris = pkg.meshgen.uniform(n, 1) if kind == "uniform" else pkg.meshgen.sponza_like(n, 3)
    n = len(tris)
    d = torch.from_numpy(tris.view(np.uint8).reshape(-1)).cuda()
    b = pkg.PLOCNew()
    for _ in range(5): b.build(ctx, d, on_device=True, n=n)
    chk = b.checksum()
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(50): b.build(ctx, d, on_device=True, n=n)
    ctx.synchronize(); wall = (time.perf_counter() - t0) / 50 * 1e3
    ctx.set_profiling(2)
    for _ in range(20): b.build(ctx, d, on_device=True, n=n)
    kt = ctx.kernel_times(); ctx.set_profiling(0)
    print(f"{kind} {n} PLOC++ wall {wall:.4f} ms | " + "  ".join(f"{k} {v[0] / 20:.4f} ({v[1] // 20})" for k, v in kt.items() if k.startswith("k_ploc")) + f"  checksum {chk:016x}", flush=True)

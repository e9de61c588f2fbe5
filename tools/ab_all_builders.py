#!/usr/bin/env python3
"""Every builder at the config sizes: un-profiled wall time per build (50 builds back to back) and the tree checksum (library from BVH_MI355X_LIB); for HPLOC at 10 M / 2 M also the
two emit kernels' event times.  python tools/ab_all_builders.py"""
import os, sys, time
import numpy as np, torch
torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bvh_pkg
pkg = bvh_pkg.load(); ctx = pkg.Context(0)
for n, kind in ((262144, "sponza"), (2_000_000, "uniform"), (10_000_000, "uniform")):
    tris = pkg.meshgen.uniform(n, 1) if kind == "uniform" else pkg.meshgen.sponza_like(n, 3)
    n = len(tris)
    d = torch.from_numpy(tris.view(np.uint8).reshape(-1)).cuda()
    out = []
    for algo, name in ((pkg.ALGO_HPLOC, "hploc"), (pkg.ALGO_SINGLEPASS, "lbvh1"), (pkg.ALGO_TWOPASS, "lbvh2"), (pkg.ALGO_PLOCPP, "ploc")):
        b = pkg.BUILDERS[algo]()
        for _ in range(5): b.build(ctx, d, on_device=True, n=n)
        chk = b.checksum()
        reps = 50 if n < 5_000_000 else 30
        ctx.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): b.build(ctx, d, on_device=True, n=n)
        ctx.synchronize(); wall = (time.perf_counter() - t0) / reps * 1e3
        extra = ""
        if algo == pkg.ALGO_HPLOC and n >= 1_000_000:
            ctx.set_profiling(2)
            for _ in range(20): b.build(ctx, d, on_device=True, n=n)
            kt = ctx.kernel_times(); ctx.set_profiling(0)
            extra = " (" + " ".join(f"{k[8:]} {v[0] / 20:.4f}" for k, v in kt.items() if "hploc" in k) + ")"
        out.append(f"{name} {wall:.4f}{extra} {chk & 0xffff:04x}")
    print(f"{kind} {n}: " + " | ".join(out), flush=True)

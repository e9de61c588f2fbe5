#!/usr/bin/env python3
"""whole-build wall time of the single-pass LBVH and HPLOC builders with either scheduler (one launch / tiles) around the thresholds.  python tools/ab_sched_small.py"""
import os, sys, time
import numpy as np
import torch
torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bvh_pkg
pkg = bvh_pkg.load(); ctx = pkg.Context(0)
for kind, n in (("bunny", 150_000), ("sponza", 200_000), ("sponza", 262_144), ("uniform", 262_144), ("sponza", 400_000), ("uniform", 600_000), ("uniform", 900_000)):
    tris = pkg.meshgen.uniform(n, 1) if kind == "uniform" else pkg.meshgen.sponza_like(n, 3) if kind == "sponza" else pkg.meshgen.bunny_like(n, 2)
    n = len(tris)
    d = torch.from_numpy(tris.view(np.uint8).reshape(-1)).cuda()
    row = []
    for name, cls, opt in (("lbvh_single", pkg.SinglePassLbvh, "lbvh"), ("hploc", pkg.HPLOC, "hploc")):
        for mode in ("single" if opt == "lbvh" else "async", "block"):
            try:
                ctx.set_option(opt, mode)
            except Exception as e:
                row.append(f"{name}/{mode}: n/a ({e})"); continue
            b = cls()
            best = 1e9
            for rep in range(3):
                for _ in range(10): b.build(ctx, d, on_device=True, n=n)
                ctx.synchronize(); t0 = time.perf_counter()
                for _ in range(200): b.build(ctx, d, on_device=True, n=n)
                ctx.synchronize(); best = min(best, (time.perf_counter() - t0) / 200 * 1e3)
            row.append(f"{name}/{mode} {best:.4f}")
        ctx.set_option(opt, "auto")
    print(f"{kind} n={n}: " + "  ".join(row), flush=True)

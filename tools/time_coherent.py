#!/usr/bin/env python3
"""stage times when the triangles arrive in spatially coherent order (a real mesh's usual state): the uniform mesh re-ordered by its own
Morton order, against the same mesh in random order.  python tools/time_coherent.py [N]"""
import os, sys
import numpy as np
import torch
torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bvh_pkg
pkg = bvh_pkg.load(); ctx = pkg.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
tris = pkg.meshgen.uniform(n, 1)
order = pkg.LBVH().build(ctx, tris).download()["sorted_vals"] if hasattr(pkg, "LBVH") else pkg.BUILDERS[1]().build(ctx, tris).download()["sorted_vals"]
coherent = np.ascontiguousarray(tris[order])
for label, t in (("random order", tris), ("morton order", coherent)):
    d = torch.from_numpy(t.view(np.uint8).reshape(-1)).cuda()
    for algo, name in ((pkg.ALGO_HPLOC, "hploc"), (pkg.ALGO_SINGLEPASS, "lbvh_single"), (pkg.ALGO_PLOCPP, "ploc")):
        b = pkg.BUILDERS[algo]()
        for _ in range(3): b.build(ctx, d, on_device=True, n=n)
        ctx.set_profiling(1); rows = []
        for _ in range(10):
            b.build(ctx, d, on_device=True, n=n); tm = b.timings; rows.append((tm.ms_total, tm.ms_extents, tm.ms_morton, tm.ms_sort, tm.ms_build))
        ctx.set_profiling(0)
        r = sorted(rows)[5]
        print(f"{label} {name} n={n}: total {r[0]:.3f} ms ({n / r[0] / 1e3:.0f} Mtris/s)  E {r[1]:.3f}  M {r[2]:.3f}  S {r[3]:.3f}  B {r[4]:.3f}", flush=True)
    del d

#!/usr/bin/env python3
"""per-kernel event times of a builder on the three mesh kinds of tools/meshgen (uniform = shuffled, sponza-like / bunny-like = generator order) — where does a stage
depend on the input's order or shape?  python tools/kernels_by_mesh.py [N ...]"""
import os, sys
import numpy as np
import torch
torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bvh_pkg
pkg = bvh_pkg.load(); ctx = pkg.Context(0)
sizes = [int(x) for x in sys.argv[1:]] or [10_000_000]
for n in sizes:
    for kind in ("uniform", "sponza", "bunny"):
        tris = pkg.meshgen.uniform(n, 1) if kind == "uniform" else pkg.meshgen.sponza_like(n, 3) if kind == "sponza" else pkg.meshgen.bunny_like(n, 2)
        ne = len(tris)
        d = torch.from_numpy(tris.view(np.uint8).reshape(-1)).cuda(); del tris
        for algo in (pkg.ALGO_HPLOC, pkg.ALGO_SINGLEPASS, pkg.ALGO_PLOCPP):
            b = pkg.BUILDERS[algo]()
            for _ in range(3): b.build(ctx, d, on_device=True, n=ne)
            ctx.set_profiling(2)
            for _ in range(10): b.build(ctx, d, on_device=True, n=ne)
            kt = ctx.kernel_times(); ctx.set_profiling(0)
            tot = sum(v[0] for v in kt.values()) / 10
            print(f"{kind:8s} n={ne} {pkg.ALGO_NAMES[algo]:15s} " + "  ".join(f"{k} {v[0] / 10:.4f}" for k, v in kt.items()) + f"  | sum {tot:.4f} ms ({ne / tot / 1e3:.0f} Mtris/s)", flush=True)
        del d
